#!/bin/bash
# kernel timeline of one far-reaching and one short goal selection (serial order), tools/bench_pipeline.py --serial-goal
out=${1:-gpurun_out/r8b}; mkdir -p $out
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl_serial
rocprofv3 --kernel-trace -d /tmp/tl_serial -- python $R/tools/bench_pipeline.py --episodes 1 --frames 40 --serial-goal > /tmp/tl_serial.log 2>&1
db=$(find /tmp/tl_serial -name '*.db' | head -1)
python - $db <<'P' > $R/$out/goal_select_timeline_final.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tl = db.execute("select name,start,end from kernels order by start").fetchall()
sel = [i for i, r in enumerate(tl) if "fmm_init" in r[0]]
fin = [i for i, r in enumerate(tl) if "goal_argmax_final" in r[0]]
for a in (sel[-10], sel[-3]):
    b = min(j for j in fin if j > a)
    t0 = tl[a - 2][1]
    print("---- select starting at dispatch", a, ":", b - a + 3, "dispatches,", round((tl[b][2] - t0) / 1e3, 1), "us")
    for r in tl[a - 2:b + 1]:
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f}  {r[0].replace('(anonymous namespace)::','').replace('peanut::','')[:80]}")
P
cd $R
