#!/bin/bash
# round 5: first run of the persistent 256 x 256 kernel -- its tests, then the per-layer A/B against the kernels it replaces
out=${1:-gpurun_out/r7a}
mkdir -p $out
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "pw256wp or bit_identical" > $out/pytest_wp.txt 2>&1
tail -15 $out/pytest_wp.txt
L="layer3.conv1,layer4.0.conv1,layer4.conv1,layer4.conv3,layer3.0.conv3ds,layer4.0.conv3ds"
for o in '{"pw256wp_mink": 0}' '{}' '{"pw256wp_mintiles": 256}' '{"pw256wp_npre": 4}' '{"pw256wp_npre": 2}' '{"pw256wp_mink": 0}' '{}'; do
  OPTS="$o" SHAPES=$L timeout 300 python tools/bench_gemm.py fp32 >> $out/gemm_ab.jsonl 2>> $out/gemm_ab.err
done
cat $out/gemm_ab.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'],d['kernel'],d['ms'],d['tflops'],d['opts'])"
