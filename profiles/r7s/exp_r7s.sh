#!/bin/bash
out=${1:-gpurun_out/r7s}
mkdir -p $out
timeout 1200 python -m pytest tests/test_rcnn_gpu.py tests/test_agent_gpu.py tests/test_c_host_gpu.py -x -q -m gpu --timeout 400 2>&1 | tail -8 | tee $out/pytest_rcnn.txt
for v in 0 1 0 1; do
  for b in 1 16; do
    PEANUT_RCNN_RPN_FUSED=$v PRECS=fp32 timeout 300 python tools/bench_rcnn.py $b 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('fused=$v B=$b', d['ms_per_batch'], d['front_end_ms'], d['images_per_s'])" | tee -a $out/rcnn_fused.txt
  done
done
