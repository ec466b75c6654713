#!/bin/bash
out=${1:-gpurun_out/r7e}
mkdir -p $out
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k "pw256wp or bit_identical" > $out/pytest_wp.txt 2>&1
tail -4 $out/pytest_wp.txt
bash tools/exp_probe.sh $out | grep -v "wg0 iter"
