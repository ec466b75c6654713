#!/usr/bin/env python3
"""Batch-1 gap table (round 5): where one 720 x 720 map -- the agent's own operating point, nav/agent/agent_state.py:345-373 -- and
one detector frame (nav/agent/agent_helper.py:220-225) lose against the same layers at the benchmark batch.

    python tools/gap_b1.py out.json [--timeline-db results.db --timeline-forwards N]

Per op of the 720 x 720 forward at batch 1: executed TFLOP/s, the SAME layer's rate at batch 32 of 480 x 480 (the headline), their
ratio, the launch's workgroup tiles against the 256 CUs (executed FLOPs / (2 K BM BN): exact for the GEMM families, whose rows and
columns are padded to whole tiles in the FLOP count), ops sorted by the time they would save at the batch-32 rate.  The same for a
detector frame (front end, batch 1 against batch 16; back half stage by stage from peanut_rcnn_stage_times).  With --timeline-db (a
rocprofv3 --kernel-trace sqlite of `bench.py --batch 1 --size 720 ...`): kernel time against the idle gaps between kernels per forward."""
import json
import os
import re
import sqlite3
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TILE = {"256x256p": (256, 256), "256x256": (256, 256), "256x128p": (256, 128), "256x128": (256, 128), "ares_128x128": (128, 128),
        "128x128": (128, 128), "128x64": (128, 64), "128x32": (128, 32)}


def tile_of(kernel):
    for k, t in TILE.items():
        if kernel.endswith(k) or ("_" + k) in kernel:
            return t
    return None


def k_of_pred(op):
    """input channels (GEMM K) of a prediction-forward op, by its name (ResNet-50-V1c-D8 + PSP head)"""
    m = re.search(r"layer(\d)\.(\d+)\.(conv\d)(\+downsample)?(\[wino\d_gemm\])?", op)
    if not m:
        if "bottleneck.conv[x][wino" in op and "gemm" in op:
            return 2048
        return None
    L, blk, conv, ds, wino = int(m.group(1)), int(m.group(2)), m.group(3), m.group(4), m.group(5)
    planes = 64 << (L - 1)
    cin_block = (64 if L == 1 else planes * 2) if blk == 0 else planes * 4
    if conv == "conv1":
        return cin_block
    if conv == "conv2":
        return planes if wino else None          # direct 3x3: K = 9 * planes, not a pointwise tile count
    if conv == "conv3":
        return planes + (cin_block if ds else 0)
    return None


def uniq(rows):
    """op names made unique in launch order (the RPN head runs the same three layers on five pyramid levels)"""
    seen, out = {}, []
    for r in rows:
        n = r[0]
        seen[n] = seen.get(n, 0) + 1
        out.append((n if seen[n] == 1 else f"{n}#{seen[n]}",) + tuple(r[1:]))
    return out


def table(rows_small, rows_big, k_of, label_small, label_big):
    rows_small, rows_big = uniq(rows_small), uniq(rows_big)
    big = {n: (fl / (ms * 1e-3) / 1e12 if ms > 0 and fl > 0 else None) for n, k, ms, fl, *_ in rows_big}
    out = []
    for n, kern, ms, fl, *_ in rows_small:
        e = {"op": n, "kernel": kern, "ms": round(ms, 4)}
        if fl > 0 and ms > 0:
            tf = fl / (ms * 1e-3) / 1e12
            e["tflops"] = round(tf, 1)
            b = big.get(n)
            if b:
                e["tflops_" + label_big] = round(b, 1)
                e["ratio"] = round(tf / b, 3)
                e["ms_at_" + label_big + "_rate"] = round(fl / (b * 1e12) * 1e3, 4)
                e["ms_lost"] = round(ms - fl / (b * 1e12) * 1e3, 4)
            t, K = tile_of(kern), k_of(n)
            if t and K:
                tiles = fl / (2.0 * K * t[0] * t[1])
                e["tiles"] = round(tiles, 1)
                e["tiles_per_cu"] = round(tiles / 256.0, 2)
        out.append(e)
    return out


def timeline(db_path, forwards):
    db = sqlite3.connect(db_path)
    tl = db.execute("select name, start, end from kernels order by start").fetchall()
    # one forward = the launches after one final resize (upsample_logits) up to and including the next: the last `forwards` of them
    ends = [i for i, (n, s, e) in enumerate(tl) if "upsample_logits" in n]
    if len(ends) < forwards + 1:
        return {"error": f"only {len(ends)} forwards in the trace", "kernels_seen": sorted({n[:60] for n, _, _ in tl})[:40]}
    per = []
    for a, b in zip(ends[-forwards - 1:-1], ends[-forwards:]):
        seg = tl[a + 1:b + 1]
        busy = sum(e - s for _, s, e in seg)
        gaps = sum(max(seg[i + 1][1] - seg[i][2], 0) for i in range(len(seg) - 1))
        per.append((len(seg), busy / 1e3, gaps / 1e3, (seg[-1][2] - seg[0][1]) / 1e3))
    n = len(per)
    return {"forwards": n, "launches_per_forward": per[0][0], "kernel_time_us": round(sum(p[1] for p in per) / n, 1),
            "inter_kernel_gaps_us": round(sum(p[2] for p in per) / n, 1), "span_us": round(sum(p[3] for p in per) / n, 1),
            "note": "means over the last forwards of a rocprofv3 --kernel-trace run (first launch of a forward to its last launch's end; "
                    "gaps = idle time between consecutive dispatches on the stream)"}


def main():
    out_path = sys.argv[1]
    from bench import synth_maps
    from peanut_amd.prediction import PEANUT_Prediction_Model
    from peanut_amd.weights import PredCfg, make_seeded_state_dict
    dev = torch.device("cuda", 0)
    cfg = PredCfg()
    m = PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=0), state_dict=make_seeded_state_dict(cfg, 0), cfg=cfg)
    x32 = synth_maps(32, cfg.in_channels, 480, dev, seed0=0)
    m.model.profile(x32, repeats=1)
    big = m.model.profile(x32, repeats=3)
    del x32
    x1 = synth_maps(1, cfg.in_channels, 720, dev, seed0=5)
    m.model.profile(x1, repeats=2)
    small = m.model.profile(x1, repeats=10)
    pred = table(small, big, k_of_pred, "b1_720", "b32_480")
    tot = sum(e["ms"] for e in pred)
    lost = sorted((e for e in pred if e.get("ms_lost")), key=lambda e: -e["ms_lost"])
    res = {"prediction_720_b1": {"sum_op_ms": round(tot, 3), "ms_lost_vs_b32_rate": round(sum(e["ms_lost"] for e in lost if e["ms_lost"] > 0), 3),
                                 "ms_gained_vs_b32_rate": round(-sum(e["ms_lost"] for e in lost if e["ms_lost"] < 0), 3),
                                 "ops_within_10pct_of_b32_rate": sum(1 for e in pred if e.get("ratio", 0) >= 0.9),
                                 "ops_with_a_rate": sum(1 for e in pred if "ratio" in e),
                                 "top_losses": [{k: e[k] for k in ("op", "kernel", "ms", "ratio", "ms_lost", "tiles") if k in e} for e in lost[:15]],
                                 "ops": pred}}
    del m
    torch.cuda.empty_cache()
    # detector frame
    from peanut_amd.rcnn import MaskRCNN
    from peanut_amd.rcnn_weights import RcnnCfg, make_seeded_rcnn_state_dict
    rcfg = RcnnCfg(score_thresh_test=0.5)
    net = MaskRCNN(rcfg, make_seeded_rcnn_state_dict(rcfg, 0), device=dev)
    g = torch.Generator().manual_seed(3)
    img16 = torch.randint(0, 256, (16, 480, 640, 3), generator=g, dtype=torch.uint8).to(dev)
    img1 = img16[:1].contiguous()
    net.probe_front(img16, reps=1)
    f16 = [(n, k, ms, fl) for n, k, ms, fl in net.probe_front(img16, reps=3)]
    net.probe_front(img1, reps=2)
    f1 = [(n, k, ms, fl) for n, k, ms, fl in net.probe_front(img1, reps=10)]
    det = table(f1, f16, lambda n: None, "b1", "b16")
    dl = sorted((e for e in det if e.get("ms_lost")), key=lambda e: -e["ms_lost"])
    net.set_stage_timing(True)
    stages = {}
    for _ in range(5):
        net.semantic(img1, rcfg.num_classes, 0.5, 0.5, None)
        for name, bound, ms, work in net.stage_times():
            stages.setdefault(name, []).append(ms)
    net.set_stage_timing(False)
    res["detector_frame_b1"] = {"front_end_sum_op_ms": round(sum(e["ms"] for e in det), 3), "front_end_launches": len(det),
                                "front_end_ms_lost_vs_b16_rate": round(sum(e["ms_lost"] for e in dl if e["ms_lost"] > 0), 3),
                                "ops_within_10pct_of_b16_rate": sum(1 for e in det if e.get("ratio", 0) >= 0.9),
                                "ops_with_a_rate": sum(1 for e in det if "ratio" in e),
                                "top_losses": [{k: e[k] for k in ("op", "kernel", "ms", "ratio", "ms_lost") if k in e} for e in dl[:15]],
                                "stages_ms": {k: round(sum(v[1:]) / max(len(v) - 1, 1), 4) for k, v in stages.items()},
                                "ops": det}
    if "--timeline-db" in sys.argv:
        dbp = sys.argv[sys.argv.index("--timeline-db") + 1]
        n = int(sys.argv[sys.argv.index("--timeline-forwards") + 1]) if "--timeline-forwards" in sys.argv else 10
        res["prediction_720_b1"]["timeline"] = timeline(dbp, n)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    p = res["prediction_720_b1"]
    print(json.dumps({"pred720_sum_ms": p["sum_op_ms"], "lost_ms": p["ms_lost_vs_b32_rate"], "within10pct": p["ops_within_10pct_of_b32_rate"],
                      "of": p["ops_with_a_rate"], "timeline": p.get("timeline")}))
    for e in p["top_losses"][:12]:
        print(e)
    d = res["detector_frame_b1"]
    print(json.dumps({k: d[k] for k in ("front_end_sum_op_ms", "front_end_launches", "front_end_ms_lost_vs_b16_rate", "ops_within_10pct_of_b16_rate",
                                        "ops_with_a_rate", "stages_ms")}))
    for e in d["top_losses"][:10]:
        print(e)


if __name__ == "__main__":
    main()
