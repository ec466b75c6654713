#!/bin/bash
mkdir -p gpurun_out/r8j
timeout 900 python -m pytest tests/test_mapping_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -4
R=$PWD; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/fm_trace2
rocprofv3 --kernel-trace -d /tmp/fm_trace2 -- python $R/tools/measure_mapping.py > $R/gpurun_out/r8j/mapping_measure.json 2>/dev/null
db=$(find /tmp/fm_trace2 -name '*.db' | head -1); [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $R/gpurun_out/r8j/mapping_trace.txt
cd $R; cat gpurun_out/r8j/mapping_measure.json; head -12 gpurun_out/r8j/mapping_trace.txt
