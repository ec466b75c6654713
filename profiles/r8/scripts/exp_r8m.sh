#!/bin/bash
timeout 1500 python -m pytest tests/test_rcnn_gpu.py tests/test_agent_gpu.py tests/test_c_host_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -4
timeout 800 python tools/exp_det_ab.py rcnn_nms_levels 1 2>&1 | tail -1
bash tools/exp_r8g.sh | grep -n "per frame\|rank_\|sort_keys\|nms_\|rpn_topk\|compact\|gather"
