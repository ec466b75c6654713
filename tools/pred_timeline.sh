#!/bin/bash
# one prediction forward under rocprofv3 --kernel-trace: launches, span, per-kernel totals and the launch sequence of the last forward.
#   tools/pred_timeline.sh [outdir] [batch] [size]
out=${1:-gpurun_out/pred_tl}; B=${2:-1}; S=${3:-720}
R=$PWD; mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl_pred
rocprofv3 --kernel-trace -d /tmp/tl_pred -- python $R/bench.py --batch $B --size $S --steps 20 --warmup 5 --no-cpu-baseline --no-probe --traffic none --configs "" > /tmp/tl_pred.log 2>&1
db=$(find /tmp/tl_pred -name '*.db' | head -1)
python - $db <<'P' | tee $R/$out/pred_b${B}_${S}_timeline.txt
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tl = db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
def short(nm):
    return nm.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("peanut::", "")[:64]
# forwards: split at the first kernel of a forward (the stem's NCHW conv or the layout pass)
idx = [i for i, r in enumerate(tl) if "conv_patch_kernel<16, 32, 2, 8, 1, 1, true>" in r[0] or "nchw_to_nhwc" in r[0]]      # a forward's first launch
segs = [tl[idx[k]:idx[k + 1]] for k in range(max(len(idx) - 11, 0), len(idx) - 1)]
n = len(segs)
if n == 0:
    print("no forward found;", len(tl), "dispatches; kernel names:", sorted({short(r[0]) for r in tl})[:40])
    print(open("/tmp/tl_pred.log").read()[-1500:])
    sys.exit(1)
fam = collections.defaultdict(lambda: [0, 0.0])
span = busy = 0.0
for seg in segs:
    span += (max(r[2] for r in seg) - seg[0][1]) / 1e3
    for nm, s, e, gx, wx in seg:
        fam[short(nm)][0] += 1; fam[short(nm)][1] += (e - s) / 1e3; busy += (e - s) / 1e3
print(f"forwards {n}; per forward: launches {sum(v[0] for v in fam.values())/n:.0f} span {span/n:.1f} us kernel time {busy/n:.1f} us")
for k, v in sorted(fam.items(), key=lambda x: -x[1][1]):
    print(f"{v[1]/n:9.1f} us {v[0]/n:6.1f} x {v[1]/v[0]:8.1f} us each  {k}")
print("# launch sequence of the last forward: dur_us gap_before_us workgroups kernel")
prev = None
for nm, st, en, gx, wx in segs[-1]:
    print(f"{(en - st) / 1e3:8.1f} {((st - prev) / 1e3 if prev else 0):7.1f} {gx // max(wx, 1):6d}  {short(nm)}")
    prev = en
P
