#!/bin/bash
# round 6, experiment r9b: the persistent 256 x 256 kernel on the SHORT-K layers the A-resident kernel carries today
# (layer2 / layer3 conv3: K = 128 / 256) and on layer3 conv1 (450 tiles: below its tile gate).  Run from the repo root.
out=${1:-gpurun_out/r9b}
mkdir -p $out
L="layer2.conv3,layer3.conv3,layer3.conv1"
for o in '{}' '{"pw256wp_mink": 128, "pw256wp_mintiles": 400}' '{"pw_ares": 0, "pw256wp_mink": 0}' '{}' '{"pw256wp_mink": 128, "pw256wp_mintiles": 400}'; do
  OPTS="$o" SHAPES=$L timeout 300 python tools/bench_gemm.py fp32 >> $out/gemm_ab.jsonl 2>> $out/gemm_ab.err
done
python -c "
import sys,json
for l in open('$out/gemm_ab.jsonl'):
    d=json.loads(l); print(d['shape'],d['kernel'],d['ms'],d['tflops'],d['opts'])" | tee $out/gemm_ab.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --configs '' --traffic none --no-probe"
for rep in 1 2; do
  echo "default: $(eval $B | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" | tee -a $out/e2e.txt
  echo "mink256: $(PEANUT_PW256WP_MINK=256 eval $B | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" | tee -a $out/e2e.txt
  echo "mink128: $(PEANUT_PW256WP_MINK=128 eval $B | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" | tee -a $out/e2e.txt
  echo "mink128+mintiles400: $(PEANUT_PW256WP_MINK=128 PEANUT_PW256WP_MINTILES=400 eval $B | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')" | tee -a $out/e2e.txt
done
