#!/bin/bash
mkdir -p gpurun_out/r8c
timeout 900 python -m pytest tests/test_goal_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r8c/tests.txt
for b in 1 0; do
for mode in "" "--serial-goal"; do
  PEANUT_FMM_BLOCKED=$b timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 60 $mode 2>/dev/null | tail -1 >> gpurun_out/r8c/pipeline_canned_blocked$b.jsonl
done
done
cat gpurun_out/r8c/tests.txt
python - <<'P'
import json
for b in (1,0):
    for l in open(f"gpurun_out/r8c/pipeline_canned_blocked{b}.jsonl"):
        d=json.loads(l); print("blocked",b, {k:d[k] for k in ("goal_overlap","steps_per_s","ms_per_step","predictions_rank0","prediction_plus_goal_ms_per_call","goal_selection_ms_per_call","goal_selection_rounds_per_call","goal_selection_passes_per_call")})
P
