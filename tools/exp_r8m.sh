#!/bin/bash
timeout 1500 python -m pytest tests/test_rcnn_gpu.py -x -q -m gpu -k "sliced or counting" 2>&1 | tail -2
timeout 800 python tools/exp_det_ab.py rcnn_topk_slice 20480 2>&1 | tail -1
bash tools/exp_r8g.sh | grep -n "per frame\|rank_se\|rank_sort\|sort_keys\|nms_scan\|rpn_topk"
