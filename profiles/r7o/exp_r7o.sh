#!/bin/bash
out=${1:-gpurun_out/r7o}
mkdir -p $out
R=$PWD
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -- python $R/bench.py --batch 1 --size 720 --steps 30 --warmup 5 --no-cpu-baseline --no-probe --also "" --traffic none --configs "" > /tmp/tl.log 2>&1
db=$(find /tmp/tl -name '*.db' | head -1)
cd $R
timeout 600 python tools/gap_b1.py $out/gap_b1.json --timeline-db $db --timeline-forwards 20 2>&1 | tail -30 | tee $out/gap_b1_summary.txt

