#!/bin/bash
out=${1:-gpurun_out/r7d}
mkdir -p $out
for b in tools/micro/build/wp_probe*; do
  echo "== $b" >> $out/probe.txt
  timeout 120 $b 115200 512 2048 1 20 >> $out/probe.txt 2>&1
  timeout 120 $b 115200 2048 512 0 20 >> $out/probe.txt 2>&1
  timeout 120 $b 115200 1536 2048 0 10 1024 >> $out/probe.txt 2>&1
done
cat $out/probe.txt
