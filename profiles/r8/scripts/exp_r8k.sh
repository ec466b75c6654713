#!/bin/bash
mkdir -p gpurun_out/r8k
timeout 900 python -m pytest tests/test_goal_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -4
for rep in 1 2; do
  timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 40 --detector 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('det 2x40', d['steps_per_s'], d['ms_per_step'], d['prediction_plus_goal_ms_per_call'])"
  timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 60 --detector 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('det 2x60', d['steps_per_s'], d['ms_per_step'], d['prediction_plus_goal_ms_per_call'])"
done
timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 60 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('canned 2x60', d['steps_per_s'], d['ms_per_step'], d['prediction_plus_goal_ms_per_call'])"
