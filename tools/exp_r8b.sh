#!/bin/bash
out=gpurun_out/r8b; mkdir -p $out
R=$PWD
cd /tmp; export TMPDIR=/tmp
for mode in overlap; do
  rm -rf /tmp/tl_$mode
  rocprofv3 --kernel-trace -d /tmp/tl_$mode -- python $R/tools/bench_pipeline.py --episodes 1 --frames 40 $flag > /tmp/tl_$mode.log 2>&1
  db=$(find /tmp/tl_$mode -name '*.db' | head -1)
  python - $db $mode <<'P' > $R/$out/window_$mode.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
print(cols)
want = [c for c in ("name", "start", "end", "queue_id", "stream_id", "queue", "stream", "grid_size", "workgroup_size", "grid_x", "workgroup_x") if c in cols]
tl = db.execute(f"select {','.join(want)} from kernels order by start").fetchall()
sel = [i for i, r in enumerate(tl) if "fmm_init" in r[0]]
a = sel[-10]
# from 60 kernels before the select's init to 40 after
t0 = tl[a - 60][1]
for r in tl[a - 60:a + 60]:
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f}  {r[0][:70]:70s} {r[3:]}")
P
done
