#!/bin/bash
out=${1:-gpurun_out/r7r}
mkdir -p $out
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "pw256p or bit_identical or stream_k" --timeout 300 2>&1 | tail -4 | tee $out/pytest_256p.txt
SHAPES="layer3.conv1" timeout 300 python tools/bench_gemm.py fp32 2>/dev/null | tee -a $out/gemm.txt
bash tools/exp_e2e.sh $out "X=1"
timeout 300 python bench.py --batch 1 --size 720 --steps 50 --warmup 5 --no-cpu-baseline --also "" --traffic none --no-probe --configs "" 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b1 720', d['ms_per_step'])" | tee -a $out/e2e.txt
