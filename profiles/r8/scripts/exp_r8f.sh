#!/bin/bash
mkdir -p gpurun_out/r8f
timeout 900 python -m pytest tests/test_goal_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -4
for i in 1 2; do
  timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 60 --serial-goal 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('trav', {k:d[k] for k in ('steps_per_s','goal_selection_ms_per_call','goal_selection_rounds_per_call','goal_selection_passes_per_call')})" | tee -a gpurun_out/r8f/inner.txt
done
timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 40 --detector 2>/dev/null | tail -1
