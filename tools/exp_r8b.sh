#!/bin/bash
out=gpurun_out/r8b; mkdir -p $out
R=$PWD
cd /tmp; export TMPDIR=/tmp
for mode in serial; do
  rm -rf /tmp/tl_$mode
  rocprofv3 --kernel-trace -d /tmp/tl_$mode -- python $R/tools/bench_pipeline.py --episodes 1 --frames 40 --serial-goal > /tmp/tl_$mode.log 2>&1
  db=$(find /tmp/tl_$mode -name '*.db' | head -1)
  python - $db $mode <<'P' > $R/$out/window2_$mode.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tl = db.execute("select name,start,end,grid_x from kernels order by start").fetchall()
sel = [i for i, r in enumerate(tl) if "fmm_init" in r[0]]
fin = [i for i, r in enumerate(tl) if "goal_argmax_final" in r[0]]
for a in (sel[-10], sel[-3]):
    b = min(j for j in fin if j > a)
    t0 = tl[a - 2][1]
    print("---- select", a)
    for r in tl[a - 2:b + 1]:
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f}  {r[0][:90]}")
P
done
