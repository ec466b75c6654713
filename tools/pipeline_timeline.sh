#!/bin/bash
# one prediction step of config 4 (update_prediction + update_global_goal) under rocprofv3 --kernel-trace: every launch from the
# goal solver's traversible-map kernel to its final argmax, with start offset, duration and queue, so that what runs beside the
# forward and what waits is visible.     tools/pipeline_timeline.sh [outdir]
out=${1:-gpurun_out/pipe_tl}
R=$PWD; mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl_pipe
rocprofv3 --kernel-trace -d /tmp/tl_pipe -- python $R/tools/bench_pipeline.py --episodes 1 --frames 30 --detector > /tmp/tl_pipe.log 2>&1
db=$(find /tmp/tl_pipe -name '*.db' | head -1)
python - $db <<'P' | tee $R/$out/pred_step_timeline.txt
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
tl = db.execute(f"select name,start,end,grid_x,workgroup_x,{q} from kernels order by start").fetchall()
def short(nm):
    return nm.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("peanut::", "")[:60]
starts = [i for i, r in enumerate(tl) if "goal_trav" in r[0]]
ends = [i for i, r in enumerate(tl) if "goal_argmax_final" in r[0]]
if not starts or not ends:
    print("no prediction step found;", len(tl), "dispatches"); print(open("/tmp/tl_pipe.log").read()[-1500:]); sys.exit(1)
# the last complete prediction step, or the one asked for (PIPE_TL_STEP = index into the list of spans printed at the end)
import os
e = ends[int(os.environ.get("PIPE_TL_STEP", "-1"))]; s = max(i for i in starts if i < e)
seg = tl[s:e + 1]
t0 = seg[0][1]
qs = {}
print(f"# prediction step: {len(seg)} launches, span {(seg[-1][2] - t0) / 1e3:.1f} us; columns: start_us dur_us queue workgroups kernel")
per_q = collections.defaultdict(float)
for nm, st, en, gx, wx, qq in seg:
    qi = qs.setdefault(qq, len(qs))
    per_q[qi] += (en - st) / 1e3
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{qi} {gx // max(wx, 1):6d}  {short(nm)}")
print("# kernel time per queue (us):", dict(per_q))
# all prediction steps: span from goal_trav to argmax_final
sp = []
for e in ends:
    ss = [i for i in starts if i < e]
    if ss: sp.append((tl[e][2] - tl[max(ss)][1]) / 1e3)
print("# spans of all prediction steps (us):", [round(x) for x in sp])
P
