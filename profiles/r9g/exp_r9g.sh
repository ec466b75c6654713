#!/bin/bash
# round 6, r9g: CUs left free by the forward's persistent kernels for the goal solver's field (option persistent_reserve_cus), and the
# PSP bottleneck's position GEMM on the persistent 256 x 128 kernel (pw256p_flush = 2048) so that it leaves them free too.
out=${1:-gpurun_out/r9g}
mkdir -p $out
P="python tools/bench_pipeline.py --episodes 2 --frames 40 --detector"
F='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(d["steps_per_s"], d["ms_per_step"], "pred+goal", d["prediction_plus_goal_ms_per_call"])'
for rep in 1 2; do
  echo "default: $($P 2>/dev/null | python -c "$F")" | tee -a $out/ab.txt
  echo "reserve 8: $(PEANUT_PERSISTENT_RESERVE_CUS=8 $P 2>/dev/null | python -c "$F")" | tee -a $out/ab.txt
  echo "reserve 16: $(PEANUT_PERSISTENT_RESERVE_CUS=16 $P 2>/dev/null | python -c "$F")" | tee -a $out/ab.txt
  echo "reserve 8 + pw256p_flush 2048: $(PEANUT_PERSISTENT_RESERVE_CUS=8 PEANUT_PW256P_FLUSH=2048 $P 2>/dev/null | python -c "$F")" | tee -a $out/ab.txt
  echo "reserve 16 + pw256p_flush 2048: $(PEANUT_PERSISTENT_RESERVE_CUS=16 PEANUT_PW256P_FLUSH=2048 $P 2>/dev/null | python -c "$F")" | tee -a $out/ab.txt
done
for v in 0 8; do
echo "pred720 alone reserve $v: $(PEANUT_PERSISTENT_RESERVE_CUS=$v python bench.py --batch 1 --size 720 --steps 50 --warmup 5 --no-cpu-baseline --traffic none --no-probe --configs '' | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')" | tee -a $out/ab.txt
done
