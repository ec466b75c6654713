#!/bin/bash
# A/B: issue priority (s_setprio) of the pyramid branch's kernels and stream priority of the side stream, one 720 x 720 / 240 x 240 map
out=gpurun_out/r9n; mkdir -p $out
for rep in 1 2; do
for cfg in "0 0" "1 0" "2 0" "3 0" "0 1" "3 1"; do
  set -- $cfg
  for S in 720 240; do
    r=$(PEANUT_SIDE_WAVE_PRIO=$1 PEANUT_SIDE_STREAM_PRIO=$2 python bench.py --batch 1 --size $S --steps 300 --warmup 30 --no-cpu-baseline --also "" --traffic none --no-probe --configs "" 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "wave_prio=$1 stream_prio=$2 size=$S rep=$rep ms=$r"
  done
done
done | tee $out/ab.txt
# the goal solver's round kernel at a higher issue priority, beside the forward (config 4, 2 x 40 frames with the detector)
for rep in 1 2; do
for fp in 0 1 3; do
  r=$(PEANUT_FMM_WAVE_PRIO=$fp python tools/bench_pipeline.py --episodes 2 --frames 40 --detector 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps_per_s'], d['prediction_plus_goal_ms_per_call'])")
  echo "fmm_wave_prio=$fp rep=$rep steps_per_s,pair_ms=$r"
done
done | tee $out/ab_goal.txt
