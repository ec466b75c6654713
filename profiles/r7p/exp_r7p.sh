#!/bin/bash
out=${1:-gpurun_out/r7p}
mkdir -p $out
for v in 0 400 1300 5000 0 400; do
  for b in 1 16; do
    PEANUT_RCNN_WINO_MINPIX=$v PRECS=fp32 timeout 300 python tools/bench_rcnn.py $b 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('minpix=$v B=$b', d['ms_per_batch'], d['front_end_ms'], d['images_per_s'])" | tee -a $out/rcnn_minpix.txt
  done
done
