#!/bin/bash
out=${1:-gpurun_out/r7b}
mkdir -p $out
L="layer4.0.conv1,layer4.conv1,layer4.conv3,layer3.0.conv3ds,layer4.0.conv3ds"
for st in 0 4 8 12 16 24 32 0; do
  OPTS="{\"pw256wp_stagger\": $st}" SHAPES=$L timeout 300 python tools/bench_gemm.py fp32 >> $out/gemm_stagger.jsonl 2>> $out/gemm.err
done
python -c "
import sys,json
for l in open('$out/gemm_stagger.jsonl'):
    d=json.loads(l); print(d['shape'],d['kernel'],d['ms'],d['tflops'],d['opts'])"
