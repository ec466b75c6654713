#!/bin/bash
mkdir -p gpurun_out/r8d
timeout 900 python -m pytest tests/test_goal_gpu.py tests/test_agent_gpu.py tests/test_mapping_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r8d/tests.txt
for rep in; do
  timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 60 2>/dev/null | tail -1 >> gpurun_out/r8d/pipeline_canned.jsonl
  timeout 300 python tools/bench_pipeline.py --episodes 2 --frames 60 --detector 2>/dev/null | tail -1 >> gpurun_out/r8d/pipeline_detector.jsonl
done

cat gpurun_out/r8d/tests.txt
exit 0
python - <<'P'
import json
for f in ("canned","detector"):
    for l in open(f"gpurun_out/r8d/pipeline_{f}.jsonl"):
        d=json.loads(l); print(f, {k:d[k] for k in ("goal_overlap","steps_per_s","ms_per_step","predictions_rank0","prediction_plus_goal_ms_per_call")})
print(open("gpurun_out/r8d/step_profile.json").read())
P
