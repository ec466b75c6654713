#!/bin/bash
out=${1:-gpurun_out/r7k}
mkdir -p $out
for cfg in "PEANUT_MAP_FUSED=0 PEANUT_MAP_VOXELS_FB=0" "PEANUT_MAP_FUSED=0 PEANUT_MAP_VOXELS_FB=1" "PEANUT_MAP_FUSED=1"; do
  echo "== $cfg" | tee -a $out/mapping_ab.txt
  env $cfg timeout 300 python tools/measure_mapping.py 2>&1 | tail -1 | tee -a $out/mapping_ab.txt
done
timeout 900 python -m pytest tests/test_mapping_gpu.py tests/test_goal_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $out/pytest_map_goal.txt
timeout 600 python -m pytest tests/test_rcnn_gpu.py -x -q -m gpu -k "semantic or args_constructor" 2>&1 | tail -5 | tee $out/pytest_rcnn.txt
timeout 600 python tools/configs_bench.py 3,4,mapping > $out/configs_3_4.json 2> $out/configs_err.txt; tail -3 $out/configs_err.txt; python -c "
import json; d=json.load(open('$out/configs_3_4.json'))
print(json.dumps(d['3']['post'], indent=0)[:3000]); print(json.dumps(d['3']['batch1']['post']['stages'], indent=0)[:2500]); print(json.dumps({k: d['4'][k] for k in ('value','ms_per_step','stages','roofline')}, indent=0)); print(d['mapping']['ms_per_step'], d['mapping']['roofline']['frac'], d['3']['cpu_baseline'])"
