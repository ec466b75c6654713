#!/bin/bash
mkdir -p gpurun_out/r8l; : > gpurun_out/r8l/gates.txt
run() { echo "== $*" >> gpurun_out/r8l/gates.txt; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --also "" --traffic none --configs "" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/r8l/gates.txt; }
run A=1
run PEANUT_PW256WP_MINTILES=400
run PEANUT_PW256WP_MINTILES=256
run PEANUT_PW256WP_MINTILES=400 PEANUT_PW256WP_MINK=256
run A=1
cat gpurun_out/r8l/gates.txt
