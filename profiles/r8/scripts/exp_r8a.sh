#!/bin/bash
mkdir -p gpurun_out/r8a
timeout 900 python -m pytest tests/test_goal_gpu.py -x -q -m gpu -k "next_to or with_and_without" 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r8a/tests.txt
for rep in 1 2; do
for mode in "" "--serial-goal"; do
  timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 60 $mode 2>/dev/null | tail -1 >> gpurun_out/r8a/pipeline_canned2.jsonl
done
done
cat gpurun_out/r8a/tests.txt
python - <<'P'
import json
for f in ("canned2",):
    for l in open(f"gpurun_out/r8a/pipeline_{f}.jsonl"):
        d=json.loads(l); print(f, {k:d[k] for k in ("goal_overlap","steps_per_s","ms_per_step","predictions_rank0","prediction_plus_goal_ms_per_call","goal_selection_ms_per_call")})
P
