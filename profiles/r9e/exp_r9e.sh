#!/bin/bash
# round 6, r9e: with the transforms cheaper at batch 1, where does the Winograd threshold of the narrow (64-channel) layers belong?
out=${1:-gpurun_out/r9e}
mkdir -p $out
P="python bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --traffic none --no-probe --configs ''"
for rep in 1 2; do
for v in 100000 20000 2000; do
  for sz in 720 240; do
  echo "wino_narrow_minpix=$v pred$sz: $(PEANUT_WINO_NARROW_MINPIX=$v eval $P --size $sz | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')" | tee -a $out/ab.txt
  done
done
for v in 0 64; do
  echo "wino_min_cin=$v detector b1: $(PEANUT_WINO_MIN_CIN=$v python tools/bench_rcnn.py 1 2>/dev/null | grep '^{' | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:d[k] for k in d if "ms" in k or k in ("batch",)})' | head -1 | tr '\n' ' ')" | tee -a $out/ab.txt
done
done
