#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel call count / total / average duration
(the `--kernel-trace --stats` view) and, when the run carried --pmc counters, per-kernel counter
sums and per-dispatch averages.  Usage: rocpd_summary.py results.db [--skip-first N]"""
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("peanut::", "").replace("void ", "")
    if "(" in name:
        name = name[:name.index("(")]
    return name[:90]


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, duration, dispatch_id, grid_x, workgroup_x, lds_size, vgpr_count, "
                       "accum_vgpr_count, sgpr_count from kernels order by start").fetchall()
    if "--timeline" in sys.argv:   # the last N dispatches in start order: duration and idle gap before each
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        tl = cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()[-n:]
        t0, prev_end, busy = tl[0][1], tl[0][1], 0
        print(f"# timeline of the last {len(tl)} dispatches: start_us  dur_us  gap_before_us  workgroups  kernel")
        for name, st, en, gx, wx in tl:
            print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} {(st - prev_end) / 1e3:8.1f} {gx // max(wx, 1):7d}  {short(name)}")
            busy += en - st
            prev_end = max(prev_end, en)
        print(f"# span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
        return
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0, None])
    for name, dur, did, gx, wx, lds, vg, ag, sg in rows:
        a = agg[short(name)]
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
        a[4] = (lds, vg, ag, sg)
    total = sum(a[1] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, {total / 1e6:.3f} ms of kernel time")
    print(f"{'kernel':92s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}  lds/vgpr/agpr/sgpr")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:92s} {a[0]:6d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} "
              f"{100 * a[1] / total:6.2f}  {a[4]}")
    try:
        crow = cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
    except sqlite3.Error:
        crow = []
    if crow:
        cagg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for kname, cname, val, did in crow:
            c = cagg[short(kname)][cname]
            c[0] += 1
            c[1] += val
        print("\n# counters: per kernel, sum over dispatches (and mean per dispatch)")
        for k, cs in sorted(cagg.items(), key=lambda kv: -agg.get(kv[0], [0, 0])[1]):
            print(k)
            for cname, (n, s) in sorted(cs.items()):
                print(f"    {cname:32s} n={n:5d} sum={s:.6g} mean={s / n:.6g}")


if __name__ == "__main__":
    main()
