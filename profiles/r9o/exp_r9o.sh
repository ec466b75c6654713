#!/bin/bash
# goal solver: block -> SIMD map of the blocked round kernel (Gray-code Latin square vs GF(4) square), config 4 with the field on its own
# (--serial-goal: goal_selection_ms_per_call) and beside the forward
out=gpurun_out/r9o; mkdir -p $out
for rep in 1 2; do
for m in 0 1; do
  r=$(PEANUT_FMM_SIMD_MAP=$m python tools/bench_pipeline.py --episodes 2 --frames 40 --detector --serial-goal 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps_per_s'], d['goal_selection_ms_per_call'], d['goal_selection_rounds_per_call'])")
  echo "serial simd_map=$m rep=$rep steps_per_s,goal_ms,rounds=$r"
  r=$(PEANUT_FMM_SIMD_MAP=$m python tools/bench_pipeline.py --episodes 2 --frames 40 --detector 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps_per_s'], d['prediction_plus_goal_ms_per_call'])")
  echo "beside simd_map=$m rep=$rep steps_per_s,pair_ms=$r"
done
done | tee $out/ab.txt
