#!/bin/bash
out=gpurun_out/r9q; mkdir -p $out
for rep in 1 2; do
for flag in "" "--no-early-goal"; do
  r=$(python tools/bench_pipeline.py --episodes 2 --frames 40 --detector $flag 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps_per_s'], d['prediction_plus_goal_ms_per_call'], d['predictions_rank0'], d['goal_fields_begun_before_segmentation'])")
  echo "40f early='$flag' rep=$rep steps_per_s,pair_ms,preds,early=$r"
done
done | tee $out/ab.txt
for flag in "" "--no-early-goal"; do
  r=$(python tools/bench_pipeline.py --episodes 2 --frames 60 --detector $flag 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps_per_s'], d['prediction_plus_goal_ms_per_call'], d['predictions_rank0'], d['goal_fields_begun_before_segmentation'])")
  echo "60f early='$flag' steps_per_s,pair_ms,preds,early=$r"
done | tee -a $out/ab.txt
