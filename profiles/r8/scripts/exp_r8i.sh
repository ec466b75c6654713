#!/bin/bash
mkdir -p gpurun_out/r8i
timeout 1500 python -m pytest tests/test_rcnn_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -5
for o in 1 0; do
  PEANUT_RCNN_FPN_OVERLAP=$o bash tools/exp_r8g.sh | grep "per frame" | sed "s/^/overlap=$o /"
done
for o in 1 0; do
  PEANUT_RCNN_FPN_OVERLAP=$o timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 40 --detector 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fpn_overlap $o', d['steps_per_s'], d['ms_per_step'])"
done
