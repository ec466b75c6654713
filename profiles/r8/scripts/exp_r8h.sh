#!/bin/bash
mkdir -p gpurun_out/r8h
timeout 1200 python -m pytest tests/test_rcnn_gpu.py tests/test_rcnn_post_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -6
bash tools/exp_r8g.sh | head -24
