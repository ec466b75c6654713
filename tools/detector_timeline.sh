#!/bin/bash
# one detector frame at batch 1 under rocprofv3 --kernel-trace: launches, span, per-kernel totals, and the launch sequence of the
# middle of the frame (a few res4 blocks).   tools/detector_timeline.sh [outdir]
out=${1:-gpurun_out/r8g}; mkdir -p $out
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl_det
cat > /tmp/det_loop.py <<'P'
import sys, torch
sys.path.insert(0, sys.argv[1])
from peanut_amd.rcnn import MaskRCNN
from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict
dev = torch.device("cuda", 0)
rcfg = RcnnCfg(score_thresh_test=0.5)
net = MaskRCNN(rcfg, make_seeded_rcnn_state_dict(rcfg, 0), device=dev)
g = torch.Generator().manual_seed(3)
img = torch.randint(0, 256, (1, 480, 640, 3), generator=g, dtype=torch.uint8).to(dev)
for _ in range(25):
    net.semantic(img, rcfg.num_classes, 0.5, 0.5, None)
torch.cuda.synchronize()
P
rocprofv3 --kernel-trace -d /tmp/tl_det -- python /tmp/det_loop.py $R > /tmp/tl_det.log 2>&1
db=$(find /tmp/tl_det -name '*.db' | head -1)
python - $db <<'P' | tee $R/$out/detector_b1_timeline.txt
import sqlite3, sys, collections, re
db = sqlite3.connect(sys.argv[1])
tl = db.execute("select name,start,end from kernels order by start").fetchall()
# frames: split at rcnn_preprocess
idx = [i for i, r in enumerate(tl) if "rcnn_preprocess" in r[0]]
print("frames", len(idx))
segs = [tl[idx[k]:idx[k + 1]] for k in range(len(idx) - 11, len(idx) - 1)]
n = len(segs)
fam = collections.defaultdict(lambda: [0, 0.0])
gaps = 0.0; span = 0.0; busy = 0.0
for seg in segs:
    span += (seg[-1][2] - seg[0][1]) / 1e3
    for i, (nm, s, e) in enumerate(seg):
        key = nm.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("peanut::", "")[:70]
        fam[key][0] += 1; fam[key][1] += (e - s) / 1e3
        busy += (e - s) / 1e3
        if i + 1 < len(seg): gaps += max(seg[i + 1][1] - e, 0) / 1e3
print(f"per frame: launches {sum(v[0] for v in fam.values())/n:.0f} span {span/n:.1f} us busy {busy/n:.1f} gaps {gaps/n:.1f}")
for k, v in sorted(fam.items(), key=lambda x: -x[1][1])[:32]:
    print(f"{v[1]/n:9.1f} us {v[0]/n:6.1f} x  {k}")
seg = segs[-1]
print("# launch sequence of the last frame: dur_us gap_before_us workgroups(x) kernel   (res4's 23 blocks collapsed to the first two)")
rows = db.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
first = [i for i, r in enumerate(rows) if r[1] == seg[0][1]][0]
prev = None
seq = rows[first:first + len(seg)]
blk = 0
for i, (nm, st, en, gx, wx) in enumerate(seq):
    if 75 < i < 75 + 21 * 5 + 3:          # the repeated res4 blocks
        prev = en
        continue
    key = nm.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("peanut::", "")[:60]
    print(f"{i:4d} {(en - st) / 1e3:8.1f} {((st - prev) / 1e3 if prev else 0):7.1f} {gx // max(wx, 1):6d}  {key}")
    prev = en
P
