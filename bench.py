#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): maps/sec of the 480x480x(4+N_cat) map-prediction forward at
batch 32 per GPU, data-parallel over N GPUs of one node (weak scaling, no data-path collective).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (NCHW map batch resident in HBM -> NCHW probabilities in HBM)
over one batch of synthetic maps.  Rank 0 prints ONE JSON line -- the compact contract object
(contract_line(): under 4 KB; the driver could not parse round 5's 23 KB line) -- and writes everything else it measured
(per-family tables, extra precision modes, the other configurations with their stage tables) to bench_detail.json.
`roofline` is measured live with HIP events recorded inside the timed region on the launch stream
(peanut_pred_probe_*); `roofline.traffic` comes from a rocprofv3 PMC pass over this very command that
bench.py runs on itself as a child (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only);
`cpu_baseline` times the oracle restatement of the reference's fp32 PyTorch path on this box's host
cores (rank 0, N=1 only, bounded sample: BASELINE.md sec. 4 -- 2 warm-ups, best of 5, B=1 and B=4).

Without a launcher (`WORLD_SIZE` unset) `--gpus N` with N > 1 spawns the N ranks itself through
torch.distributed.run; it exits non-zero when fewer than N GPUs are visible -- it never prints an
`n_gpus: 1` line for `--gpus 8`.  `--config 5` selects SURVEY.md sec. 8d config 5 (960x960, 25 channels,
8 maps per GPU, all-gather of the predicted maps timed); `--maps file.npz` feeds a map sequence in the
reference's on-disk format (nav/collect_maps.py:80-87) instead of synthetic maps.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from peanut_amd import dist as pdist  # noqa: E402
from peanut_amd.weights import PredCfg, conv_flops_per_map, make_seeded_state_dict  # noqa: E402

# MI355X_MICROARCH.md: fp32 MFMA 157.3 TF (v_mfma_f32_32x32x2_f32); dense bf16/f16 MFMA 2.5 PF, of which a
# split-product mode can deliver at most one third as fp32-equivalent FLOPs (3 MFMAs per product).
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0 / 3, "fp16x3": 2500.0 / 3, "bf16x6": 2500.0 / 6}
DTYPE = {"fp32": "f32",
         "bf16x3": "f32 tensors; conv products from 2 bf16 pieces per value (3 MFMA products), f32 accumulate",
         "fp16x3": "f32 tensors; conv products emulated from 2 fp16 pieces per value (3 MFMA products), f32 accumulate",
         "bf16x6": "f32 tensors; conv products emulated from 3 bf16 pieces per value (6 MFMA products), f32 accumulate"}
MODE_NOTES = {
    "fp16x3": "opt-in: 2 fp16 pieces per value (22 significand bits; activations split in registers, weights pre-split "
              "after a per-layer power-of-two scale), 3 MFMA products per fp32 product, fp32 accumulate; needs the "
              "activations of the emulated layers below 65504 in magnitude; 9.5e-6 max-abs on the logits vs the reference "
              "golden vectors, 5-7e-6 from the float64 run",
    "bf16x3": "opt-in speed mode: 2 bf16 pieces per value, 3 MFMA products per fp32 product, fp32 accumulate; ~9e-5 max-abs "
              "on the logits vs the reference golden vectors (bound 1e-3); not fp32-class, not the headline value",
    "bf16x6": "fp32 emulation on the bf16 matrix cores (csrc/gemm_rs.hip, conv_rs.hip): activations stay fp32 in HBM / LDS and are split "
              "into 3 bf16 pieces in registers (exact split), weights pre-split, 6 MFMA products per fp32 product, fp32 "
              "accumulate; 9.1e-6 max-abs on the logits vs the reference golden vectors, 5-9e-6 from a float64 run of the "
              "reference model (8.9e-6 at 480x480) -- the level of the fp32 MFMA path (7.9e-6 / 5-7e-6) and of the reference's own "
              "fp32 CPU path (5.7e-6 / 7.7e-6); reported next to the headline, which stays on fp32 MFMA instructions",
}
METRIC = "maps/sec for 480x480x(4+N_cat) prediction fwd, batch 32"
ALGORITHMS = ("direct implicit GEMM on fp32 MFMA; stride-1 3x3 convs with >= 64 input channels as Winograd with fp32 "
              "transforms, form chosen per shape -- at this size F(6x6,3x3) up to dilation 2 (PSP bottleneck included "
              "in the fp32 mode, whose position GEMMs accumulate in two levels: partial sums of 64 channels), "
              "F(5x5,3x3) for the dilation-4 layers (15x15 sub-grids), F(4x4,3x3) in the bottleneck of the emulated "
              "modes; pyramid half of the PSP bottleneck folded through linearity "
              "(nominal GFLOP/map counts the reference's 61 direct convs, so nominal TFLOP/s can exceed the MFMA peak)")

# The driver parses the LAST stdout line; round 5's 23 KB line was not parsed.  The contract line is held under this many bytes,
# everything else (op families, modes, the other configurations with their stage tables, notes) goes to bench_detail.json.
CONTRACT_LINE_MAX_BYTES = 4096
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                 "traffic_measured_by_this_run", "launches_per_step", "avg_launch_ms", "mfma_busy", "flops_per_launch", "share_of_step_time",
                 "gflop_per_map_executed", "whole_forward_tflops_executed", "effective_clock_ghz")
CPU_KEYS = ("value", "unit", "cores", "kind", "cpu_model", "sample")


def write_detail(detail: dict, path: str = "") -> str:
    """Everything bench.py measured, as indented JSON next to the script (and under gpurun_out/ when that directory exists,
    so that a gpurun call brings it back); returns the path written ('' when neither location is writable)."""
    written = ""
    targets = [path] if path else [os.path.join(ROOT, "bench_detail.json")]
    if not path and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        targets.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    for t in targets:
        try:
            with open(t, "w") as fh:
                json.dump(detail, fh, indent=1)
            written = written or t
        except OSError:
            pass
    return written


def configs_summary(configs: dict) -> dict:
    """Flat numbers of the other BASELINE.json configurations (tools/configs_bench.py) for the contract line."""
    flat = {}
    if not isinstance(configs, dict):
        return flat
    if "error" in configs:
        flat["error"] = str(configs["error"])[:200]
    for key, stem, unit in (("1", "config1", "maps_s"), ("3", "config3", "images_s"), ("4", "config4", "steps_s"),
                            ("5", "config5", "maps_s"), ("mapping", "mapping", "steps_s")):
        c = configs.get(key)
        if isinstance(c, dict) and "value" in c:
            flat[f"{stem}_{unit}"] = c["value"]
            if (c.get("roofline") or {}).get("frac") is not None:
                flat[f"{stem}_roofline_frac"] = c["roofline"]["frac"]
            if (c.get("cpu_baseline") or {}).get("value") is not None:
                flat[f"{stem}_cpu_{unit}"] = c["cpu_baseline"]["value"]
    c3 = configs.get("3")
    if isinstance(c3, dict) and isinstance(c3.get("batch1"), dict):
        flat["detector_b1_ms_per_frame"] = c3["batch1"].get("ms_per_frame")
    c4 = configs.get("4")
    if isinstance(c4, dict):
        st = c4.get("stages") or {}
        if "prediction_720_per_step" in st:
            flat["pred720_b1_ms_per_map"] = st["prediction_720_per_step"].get("ms_per_call")
            flat["pred720_b1_frac"] = st["prediction_720_per_step"].get("frac")
        if "mapping" in st:
            flat["mapping_ms_per_step"] = st["mapping"].get("ms")
        if "goal_selection_per_step" in st:
            flat["goal_ms_added_per_call"] = st["goal_selection_per_step"].get("ms_added_per_call_next_to_the_forward")
        flat["config4_ms_per_step"] = c4.get("ms_per_step")
    if "seconds" in configs:
        flat["seconds"] = configs["seconds"]
    return flat


def contract_line(detail: dict, detail_path: str = "") -> dict:
    """The driver's contract object: the fields the bench contract names plus `roofline` and `cpu_baseline` reduced to their
    numbers, flat summaries of the extra modes / configurations, and the name of the detail file -- never above
    CONTRACT_LINE_MAX_BYTES as compact JSON (optional keys are dropped, least important first, if it ever would be)."""
    line = {k: detail[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                   "scaling", "vs_baseline", "dtype", "data", "config")}
    roof, cpu = detail.get("roofline"), detail.get("cpu_baseline")
    line["roofline"] = {k: roof[k] for k in ROOFLINE_KEYS if k in roof} if roof else None
    line["cpu_baseline"] = {k: cpu[k] for k in CPU_KEYS if k in cpu} if cpu else None
    optional = []                   # dropped from the end if the line would not fit
    for k in ("speedup_vs_cpu_baseline", "gflop_per_map_nominal", "allgather_maps_ms", "allgather_maps_bytes_per_rank",
              "allgather_maps_path", "rccl_ranks_seen"):
        if k in detail:
            line[k] = detail[k]
    if detail.get("configs") is not None:
        line["configs_summary"] = configs_summary(detail["configs"])
        optional.append("configs_summary")
    if detail.get("modes"):
        line["modes_summary"] = {m: {"value": v["value"], "ms_per_step": v["ms_per_step"],
                                     "frac": (v.get("roofline") or {}).get("frac")} for m, v in detail["modes"].items()}
        optional.append("modes_summary")
    if roof and roof.get("hbm_bound_kernels"):
        line["hbm_bound_frac_of_8tbs"] = {k: v["frac_of_8tbs"] for k, v in roof["hbm_bound_kernels"].items()}
        optional.append("hbm_bound_frac_of_8tbs")
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None

    def size(o):
        return len(json.dumps(o, separators=(",", ":")).encode())
    while size(line) >= CONTRACT_LINE_MAX_BYTES and optional:
        line.pop(optional.pop())
    if size(line) >= CONTRACT_LINE_MAX_BYTES:      # free-text fields are the only thing left that can be long
        for obj, k in ((line.get("cpu_baseline") or {}, "sample"), (line["config"], "workload"), (line, "allgather_maps_path")):
            if isinstance(obj.get(k), str):
                obj[k] = obj[k][:160]
    assert size(line) < CONTRACT_LINE_MAX_BYTES
    return line


def synth_maps(b: int, c: int, s: int, device, seed0: int = 0) -> torch.Tensor:
    """SURVEY.md sec. 8d config 2, literally (values in {0,1} like the reference's thresholded maps):
    ch 0/1 (obstacle / explored) = Bernoulli(0.3) seeds dilated by a 5x5 max-pool, ch 2-3 = one 5x5 square
    (agent location), ch 4.. = category blobs from Bernoulli(0.02) seeds dilated 5x5; seed = global map index."""
    out = torch.zeros((b, c, s, s), dtype=torch.float32, device=device)
    for i in range(b):
        g = torch.Generator(device="cpu").manual_seed(seed0 + i)
        occ = (torch.rand((1, 2, s, s), generator=g) < 0.3).float()
        out[i, 0:2] = torch.nn.functional.max_pool2d(occ, 5, 1, 2)[0].to(device)
        cy, cx = (int(v) for v in torch.randint(8, s - 8, (2,), generator=g))
        out[i, 2:4, cy - 2:cy + 3, cx - 2:cx + 3] = 1.0
        if c > 4:
            cat = (torch.rand((1, c - 4, s, s), generator=g) < 0.02).float()
            out[i, 4:] = torch.nn.functional.max_pool2d(cat, 5, 1, 2)[0].to(device)
    return out


def maps_from_file(path: str, b: int, c: int, s: int, device, first: int = 0) -> torch.Tensor:
    """[b,c,s,s] model inputs from a map sequence in the reference's .npz format (key 'maps', uint8 [T,C,W,H],
    nav/collect_maps.py:80-87), scaled by /255 as LoadMapFromFile does (train_prediction_model.py:63-68) and
    centre-cropped to the prediction window like Agent_State.update_prediction (agent_state.py:357-361).
    Snapshots are taken round-robin starting at `first` when the file holds fewer than b of them."""
    from peanut_amd import mapio
    maps = mapio.load_map_sequence(path)
    if maps.ndim != 4 or maps.shape[1] != c:
        raise SystemExit(f"--maps: expected uint8 [T,{c},W,H], got {maps.shape}")
    if maps.shape[2] < s or maps.shape[3] < s:
        raise SystemExit(f"--maps: maps of {maps.shape[2]}x{maps.shape[3]} are smaller than --size {s}")
    x1, y1 = maps.shape[2] // 2 - s // 2, maps.shape[3] // 2 - s // 2
    xs = [mapio.model_input(maps, (first + i) % maps.shape[0])[:, :, x1:x1 + s, y1:y1 + s] for i in range(b)]
    return torch.cat(xs, 0).contiguous().to(device)


def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg: PredCfg, sd, s: int, budget_s: float = 40.0):
    """Oracle (= bit-exact restatement of the reference's CPU PyTorch path) on the host cores, timed as
    BASELINE.md sec. 4 plans: fp32, no_grad, torch.set_num_threads(N) with N and the CPU model reported,
    2 warm-ups then best of 5, at B=1 and B=4 (a batch of 32 is 8 x 4).  `value` is the best per-map rate of the
    legs (the most favourable reading for the CPU).  The thread count defaults to 16: the sweep on the MI355X
    box's 2x EPYC 9575F (profiles/cpu_baseline_sweep_r1.json) found more threads slower for these convs; a
    second B=4 leg with 64 threads is tried while the time budget lasts."""
    from oracle import pspnet_ref
    threads = int(os.environ.get("PEANUT_CPU_THREADS", min(16, os.cpu_count() or 1)))
    legs, t_start = [], time.perf_counter()
    plan = [(1, threads), (4, threads)]
    if (os.cpu_count() or 1) >= 64 and "PEANUT_CPU_THREADS" not in os.environ:
        plan.append((4, 64))
    for b, n in plan:
        if legs and time.perf_counter() - t_start > budget_s:
            break
        torch.set_num_threads(n)
        x = synth_maps(b, cfg.in_channels, s, "cpu", seed0=10_000)
        for _ in range(2):
            pspnet_ref.forward_batch(sd, x, cfg)                  # warm-ups (oneDNN primitive cache)
        best = float("inf")
        for _ in range(5):
            t0 = time.perf_counter()
            pspnet_ref.forward_batch(sd, x, cfg)
            best = min(best, time.perf_counter() - t0)
        legs.append({"batch": b, "threads": n, "maps_per_s": round(b / best, 4), "best_s": round(best, 4)})
    top = max(legs, key=lambda l: l["maps_per_s"])
    return {"value": top["maps_per_s"], "unit": "maps/s", "cores": int(top["threads"]), "kind": "port",
            "cpu_model": cpu_model_name(), "host_logical_cpus": os.cpu_count(), "legs": legs,
            "sample": f"oracle/pspnet_ref.py forward of [B,{cfg.in_channels},{s},{s}] fp32 maps, 2 warm-ups + best of 5 "
                      f"per leg (BASELINE.md sec. 4), torch {torch.__version__} CPU kernels; value = best leg "
                      f"(B={top['batch']}, {top['threads']} threads)"}


# the family that carries most of the fp32 headline's step time (round 5: the persistent 256 x 256 kernel); profiles/hbm_traffic.json
# must hold its traffic for the N > 1 lines (tests/test_dist_cpu.py checks the file, tests/test_pred_gpu.py the family)
DOMINANT_FAMILY_FP32 = "conv_pw_glds_256x256p"

# bench kernel family -> substring of the rocprofv3 kernel name
FAMILY_KERNEL = {
    "conv_pw_glds_256x128": "conv_pw_glds256_kernel(",
    "conv_pw_glds_256x256": "conv_pw_glds256w_kernel",
    "conv_pw_glds_256x128p": "conv_pw_glds256p_kernel",
    "conv_pw_glds_256x256p": "conv_pw_glds256wp_kernel",
    "conv_pw_ares_128x128": "conv_pw_ares_kernel",
    "conv_pw_glds_128x128": "conv_pw_glds_kernel<128, 2, 2>",
    "conv_pw_glds_128x64": "conv_pw_glds_kernel<64, 2, 2>",
    "conv_pw_glds_128x32": "conv_pw_glds_kernel<32, 4, 1>",
    # gemm_rs_kernel<BM, BN, WM, WN, KIND, PACK>: matched up to the emulation kind (3 = bf16x6, 2 = bf16x3, 4 = fp16x3)
    "gemm_rs6_256x256": "gemm_rs_kernel<256, 256, 4, 2, 3,",
    "gemm_rs3_256x256": "gemm_rs_kernel<256, 256, 4, 2, 2,",
    "gemm_rs6_128x128": "gemm_rs_kernel<128, 128, 2, 2, 3,",
    "gemm_rs3_128x128": "gemm_rs_kernel<128, 128, 2, 2, 2,",
    "gemm_rs3h_256x256": "gemm_rs_kernel<256, 256, 4, 2, 4,",
    "gemm_rs3h_128x128": "gemm_rs_kernel<128, 128, 2, 2, 4,",
}


def _pmc_pass(counter: str, child_args, timeout_s: float, want=None):
    """One `rocprofv3 --kernel-trace --pmc <counters>` pass over a short child run of this script; returns
    {kernel name: (dispatches, sum)} from the rocpd database -- {(kernel name, counter): ...} when `want` names several
    counters -- or None."""
    import sqlite3
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = tempfile.mkdtemp(prefix="peanut_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--pmc"] + counter.split() + ["-d", out, "--", sys.executable, os.path.abspath(__file__)] + child_args
    try:
        r = subprocess.run(cmd, cwd=out, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
        if r.returncode != 0:
            return None
        dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if not dbs:
            return None
        agg = {}
        cur = sqlite3.connect(dbs[0]).cursor()
        if want is not None:      # kernel durations of the same pass (ns): the effective clock is GRBM_GUI_ACTIVE / 8 / duration
            try:
                for kname, dur in cur.execute("select name, duration from kernels"):
                    a = agg.setdefault((kname, "__duration_ns"), [0, 0.0])
                    a[0] += 1
                    a[1] += dur
            except sqlite3.Error:
                pass
        for kname, cname, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            if want is not None:
                if cname in want:
                    a = agg.setdefault((kname, cname), [0, 0.0])
                    a[0] += 1
                    a[1] += val
            elif cname == counter:
                a = agg.setdefault(kname, [0, 0.0])
                a[0] += 1
                a[1] += val
        return agg
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error):
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def measure_hbm_traffic(family: str, child_args, timeout_s: float = 150.0):
    """HBM bytes per launch of `family`'s kernel, measured on THIS command as MI355X_MICROARCH.md's HBM section
    prescribes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (with --kernel-trace only), values in KiB,
    FETCH_SIZE doubled on gfx950 (16-byte-per-lane streaming reads are tallied at half their size)."""
    pat = FAMILY_KERNEL.get(family)
    if pat is None:
        return {"traffic": None, "traffic_source": f"no rocprofv3 kernel name known for family {family}"}
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = _pmc_pass(counter, child_args, timeout_s)
        if not agg:
            return {"traffic": None, "traffic_source": f"rocprofv3 --pmc {counter} pass unavailable or failed"}
        n = sum(a[0] for k, a in agg.items() if pat in k)
        s = sum(a[1] for k, a in agg.items() if pat in k)
        if n == 0:
            return {"traffic": None, "traffic_source": f"kernel {pat} not found in the --pmc {counter} pass"}
        res[counter] = (n, s / n)
    fetch_kib, write_kib = res["FETCH_SIZE"][1], res["WRITE_SIZE"][1]
    busy = {}
    both = _pmc_pass("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE", child_args, timeout_s, want=("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
    if both:
        mf = sum(a[1] for (k, c), a in both.items() if pat in k and c == "SQ_VALU_MFMA_BUSY_CYCLES")
        ga = sum(a[1] for (k, c), a in both.items() if pat in k and c == "GRBM_GUI_ACTIVE")
        nd = sum(a[0] for (k, c), a in both.items() if pat in k and c == "GRBM_GUI_ACTIVE")
        if ga > 0:
            # GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles; 256 CUs x 4 SIMDs have a matrix pipe each
            busy = {"mfma_busy": round(mf / (ga / 8.0 * 1024.0), 4), "mfma_busy_dispatches": nd,
                    "mfma_busy_source": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), third PMC child pass "
                                        "(MFMA utilisation in cycles: independent of the clock the governor picks)"}
            dur_ns = sum(a[1] for (k, c), a in both.items() if pat in k and c == "__duration_ns")
            if dur_ns > 0:      # what the power / clock governor left of the 2.4 GHz nameplate while this kernel ran
                busy["effective_clock_ghz"] = round(ga / 8.0 / dur_ns, 3)
                busy["effective_clock_source"] = "GRBM_GUI_ACTIVE / 8 / kernel duration of the same (profiled) pass"
    return {**busy, "traffic": round((2.0 * fetch_kib + write_kib) * 1024.0),
            "traffic_fetch_kib_raw": round(fetch_kib, 1), "traffic_write_kib": round(write_kib, 1),
            "traffic_dispatches": res["FETCH_SIZE"][0],
            "traffic_source": "measured by this run: child passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and "
                              "`--pmc WRITE_SIZE` over `bench.py " + " ".join(child_args) + "` (per-dispatch mean of "
                              f"{pat}; FETCH_SIZE x2 per MI355X_MICROARCH.md)"}


def hbm_traffic_from_file(precision, family):
    """Fallback: per-dispatch means committed under profiles/hbm_traffic.json by an earlier PMC run."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            e = json.load(fh)[precision][family]
        return {"traffic": round((2.0 * e["fetch_size_kib_mean"] + e["write_size_kib_mean"]) * 1024.0),
                "traffic_source": "NOT measured by this run -- " + e["source"]}
    except (OSError, KeyError, ValueError):
        return {"traffic": None}


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} HIP device(s) are visible; refusing to report a smaller job "
              f"under that flag", file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class HipBackend:
    """What main() needs from the device side: the HIP device of this rank, the model factory (the product path through
    the C ABI) and a synchronise.  tests/test_dist_cpu.py substitutes a host-side stand-in to run main()'s N > 1 control
    flow (barriers, max-over-ranks timing, rank-0-only printing, the all-gather fallback and its exit status) under gloo."""
    name = "hip"

    def __init__(self):
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())

    def synchronize(self):
        torch.cuda.synchronize()

    def make_model(self, cfg, sd, precision):
        from peanut_amd.prediction import PEANUT_Prediction_Model
        return PEANUT_Prediction_Model(SimpleNamespace(sem_gpu_id=self.device.index), state_dict=sd, cfg=cfg, precision=precision)

    def state_dict(self, cfg):
        return make_seeded_state_dict(cfg, seed=0)


PRESETS = {
    2: dict(size=480, channels=14, batch=32),
    5: dict(size=960, channels=25, batch=8),
}


def main(argv=None, backend_factory=HipBackend):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(PRESETS),
                    help="SURVEY.md sec. 8d configuration: 2 = headline (480x480, 14 ch, 32 maps/GPU), "
                         "5 = 960x960, 25 ch, 8 maps/GPU + all-gather of the predicted maps")
    ap.add_argument("--batch", type=int, default=None, help="maps per GPU per step (overrides the preset)")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--channels", type=int, default=None, help="4 + N_cat input channels")
    ap.add_argument("--maps", default="", help="map sequence (.npz, key 'maps', uint8 [T,C,W,H]: the reference's "
                                               "collect_maps.py format) used as input instead of synthetic maps")
    ap.add_argument("--precision", default=os.environ.get("PEANUT_PRECISION", "fp32"),
                    choices=sorted(PEAK_TFLOPS), help="conv arithmetic (include/peanut_hip.h PEANUT_PREC_*)")
    ap.add_argument("--also", default=os.environ.get("PEANUT_BENCH_ALSO"),
                    help="comma list of extra precision modes measured after the main run and reported under "
                         "'modes' of bench_detail.json ('all' = bf16x6,fp16x3,bf16x3); default: none -- the driver form measures "
                         "the fp32 headline only (tools/final_measure.sh asks for all of them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the per-op HIP-event probe")
    ap.add_argument("--traffic", default="auto", choices=["auto", "measure", "file", "none"],
                    help="roofline.traffic: 'measure' profiles a short child run of this command with rocprofv3 PMC "
                         "counters; 'auto' = measure at N=1 when rocprofv3 is installed, else the committed file")
    ap.add_argument("--op-table", default="", help="write the per-op timing table (JSON) here")
    ap.add_argument("--detail", default="", help="where the full measurement record goes (default: bench_detail.json next to this script)")
    ap.add_argument("--configs", default=os.environ.get("PEANUT_BENCH_CONFIGS", "4"),
                    help="the other BASELINE.json configurations measured after the headline (N = 1 only) and reported under "
                         "'configs' of bench_detail.json, each with its own roofline and cpu_baseline, flat numbers in the contract "
                         "line's configs_summary (tools/configs_bench.py); default 4 (the per-step pipeline, which also measures the "
                         "detector and the mapping stage), 'all' = 1,3,4,5,mapping, empty string to skip")
    args = ap.parse_args(argv)
    preset = PRESETS[args.config]
    for k, v in preset.items():
        if getattr(args, k) is None:
            setattr(args, k, v)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    rank, local_rank, world = pdist.init_process_group()
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    backend = backend_factory()
    dev = backend.device
    if args.also is None:
        args.also = ""
    if args.also == "all":
        args.also = "bf16x6,fp16x3,bf16x3"
    if args.configs == "all":
        args.configs = "1,3,4,5,mapping"

    cfg = PredCfg(in_channels=args.channels)
    sd = backend.state_dict(cfg)
    B, S = args.batch, args.size
    # this rank's shard of the global batch (weak scaling: B maps per GPU)
    if args.maps:
        x = maps_from_file(args.maps, B, cfg.in_channels, S, dev, first=rank * B)
    else:
        x = synth_maps(B, cfg.in_channels, S, dev, seed0=rank * B)
    out = torch.empty((B, cfg.num_classes, S, S), dtype=torch.float32, device=dev)

    def run_mode(precision, steps, warmup, op_table=""):
        """W untimed warm-up steps, then exactly `steps` timed steps bracketed by barrier + synchronize;
        returns (max-over-ranks seconds, roofline dict)."""
        model = backend.make_model(cfg, sd, precision)
        for _ in range(warmup):
            model.get_prediction_batch(x, apply_sigmoid=True, out=out)
        backend.synchronize()
        pdist.barrier()
        backend.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):                 # the timed region: the product path as a caller runs it, no probe
            model.get_prediction_batch(x, apply_sigmoid=True, out=out)
        backend.synchronize()
        pdist.barrier()
        elapsed = pdist.max_over_ranks(time.perf_counter() - t0, device=dev)
        roof = None
        if not args.no_probe:
            # second pass, after the timed steps: per-op HIP events on the launch stream (the probe serialises the two-stream
            # PSP head and costs ~1.5 %, so it no longer sits inside the timed region)
            model.model.probe_enable(True)
            for _ in range(min(steps, 10)):
                model.get_prediction_batch(x, apply_sigmoid=True, out=out)
            backend.synchronize()
            nf, rows = model.model.probe_collect()
            model.model.probe_enable(False)
            fam = {}
            for name, kern, ms, fl, by in rows:
                f = fam.setdefault(kern, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
                f["ms"] += ms
                f["flops"] += fl * nf
                f["bytes"] += by * nf
                f["launches"] += nf
            k, f = max(fam.items(), key=lambda kv: kv[1]["ms"])
            ach = f["flops"] / (f["ms"] * 1e-3) / 1e12
            peak = PEAK_TFLOPS[precision]
            executed = sum(r[3] for r in rows) / B / 1e9
            roof = {"bound": "mfma", "kernel": k, "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": round(f["bytes"] / max(f["launches"], 1)),
                    "launches_per_step": f["launches"] // max(nf, 1),
                    "avg_launch_ms": round(f["ms"] / max(f["launches"], 1), 4),
                    "flops_per_launch": round(f["flops"] / max(f["launches"], 1)),
                    "share_of_step_time": round(f["ms"] / max(sum(v["ms"] for v in fam.values()), 1e-9), 4),
                    "note": "achieved = FLOPs this kernel family EXECUTES per launch (Winograd GEMMs at their transformed "
                            "size incl. tile padding, folded pyramid excluded) / its mean launch time from HIP events on the "
                            "launch stream, recorded over a second pass of the same steps right after the timed region",
                    "gflop_per_map_executed": round(executed, 3),
                    "whole_forward_tflops_executed": round(executed * 1e9 * B * steps / elapsed / 1e12, 2)}
            # the HBM-bound sub-kernels are reported against bandwidth -- algorithmic bytes (operands read
            # once, result written once) / summed launch time of the family; peak ~8 TB/s (MI355X_MICROARCH.md)
            hbm = {kk: {"gb_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9), "frac_of_8tbs": round(v["bytes"] / (v["ms"] * 1e-3) / 8e12, 3),
                        "ms_per_step": round(v["ms"] / max(nf, 1), 3)}
                   for kk, v in fam.items() if v["flops"] == 0 and v["bytes"] > 0 and v["ms"] > 0}
            if hbm:
                roof["hbm_bound_kernels"] = hbm
            if op_table and rank == 0:
                with open(op_table, "w") as fh:
                    json.dump({"forwards": nf, "B": B, "S": S, "precision": precision,
                               "ops": [{"op": n_, "kernel": k_, "ms": ms / nf, "gflop": fl / 1e9, "mbytes": by / 1e6,
                                        "tflops": (fl / (ms / nf * 1e-3) / 1e12) if ms > 0 and fl > 0 else None,
                                        "algorithmic_gb_s": (by / (ms / nf * 1e-3) / 1e9) if ms > 0 and by > 0 else None}
                                       for n_, k_, ms, fl, by in rows],
                               "families": {k_: {"ms_per_step": v["ms"] / nf,
                                                 "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] else None}
                                            for k_, v in fam.items()}}, fh, indent=1)
        del model
        return elapsed, roof

    elapsed, roof = run_mode(args.precision, args.steps, args.warmup, args.op_table)
    modes = {}
    for extra in [m for m in args.also.split(",") if m and m != args.precision]:
        st = max(3, args.steps // 2)
        e_s, e_roof = run_mode(extra, st, 2)
        if rank == 0 and world == 1 and e_roof is not None and args.traffic in ("auto", "measure") and not args.maps:
            # the same three PMC child passes as for the headline: HBM traffic, matrix-pipe busy share and effective clock of
            # this mode's dominant kernel
            child = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-probe", "--also", "", "--traffic", "none", "--configs", "",
                     "--config", str(args.config), "--batch", str(B), "--size", str(S), "--channels", str(cfg.in_channels),
                     "--precision", extra]
            got = measure_hbm_traffic(e_roof["kernel"], child)
            if got:
                e_roof.update(got)
        modes[extra] = {"value": round(world * B * st / e_s, 3), "unit": "maps/s", "ms_per_step": round(e_s / st * 1e3, 3),
                        "dtype": DTYPE[extra], "steps": st, "roofline": e_roof,
                        "note": MODE_NOTES.get(extra, "")}

    # logging-only collective: collate the predicted maps of the last step (outside the timed steps; timed on its own)
    gather_ms = None
    gather_failed = False
    if world > 1 or args.config == 5:
        gather = pdist.allgather_maps                  # peanut_allgather_maps (the library's RCCL entry point)
        # one rank (--config 5 at N = 1): peanut_amd.dist.allgather_maps returns its input untouched -- nothing goes through RCCL
        gather_path = "peanut_allgather_maps (RCCL)" if world > 1 else "identity (1 rank: RCCL not used)"
        try:
            gather(out)                                # warm-up (communicator set-up, collective)
        except Exception as e:                         # reporting only: never lose the bench line over the logging collective
            import torch.distributed as tdist

            def gather(t, _e=e):
                full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                if world > 1:
                    tdist.all_gather_into_tensor(full, t.contiguous())
                else:
                    full.copy_(t)
                return full
            gather_path = f"torch.distributed all_gather_into_tensor (library path failed: {e})"
            gather_failed = True
            gather(out)
        backend.synchronize()
        pdist.barrier()
        tg = time.perf_counter()
        allmaps = gather(out)
        backend.synchronize()
        gather_ms = pdist.max_over_ranks((time.perf_counter() - tg) * 1e3, device=dev)
        assert allmaps.shape[0] == world * B

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and backend.name == "hip":
        cpu = cpu_baseline(cfg, sd, S)

    # every other BASELINE.json configuration, at N = 1 only (they are per-GPU shares; episodes / maps shard without a collective)
    configs = None
    which = tuple(c for c in args.configs.split(",") if c)
    if rank == 0 and world == 1 and which and backend.name == "hip" and args.config == 2 and not args.maps and args.batch == PRESETS[2]["batch"] and \
            args.size == PRESETS[2]["size"]:
        x = out = None                  # the headline's tensors are no longer needed: free them for the other configurations
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import configs_bench
        try:
            configs = configs_bench.measure_configs(which, dev, with_cpu=not args.no_cpu_baseline)
        except Exception as e:     # noqa: BLE001 -- never lose the headline line over a secondary configuration
            configs = {"error": f"{type(e).__name__}: {e}"}
        out = torch.empty((B, cfg.num_classes, S, S), dtype=torch.float32, device=dev)

    if rank == 0 and roof is not None and args.traffic != "none":
        measured = None
        if world == 1 and args.traffic in ("auto", "measure"):
            child = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-probe", "--also", "", "--traffic", "none", "--configs", "",
                     "--config", str(args.config), "--batch", str(B), "--size", str(S), "--channels", str(cfg.in_channels),
                     "--precision", args.precision] + (["--maps", os.path.abspath(args.maps)] if args.maps else [])
            measured = measure_hbm_traffic(roof["kernel"], child)
        roof["traffic_measured_by_this_run"] = bool(measured is not None and measured.get("traffic") is not None)
        if roof["traffic_measured_by_this_run"]:
            roof.update(measured)
        else:
            fb = hbm_traffic_from_file(args.precision, roof["kernel"]) if (args.config == 2 and not args.maps) else {"traffic": None}
            if measured is not None and fb.get("traffic") is None:
                fb = measured
            elif measured is not None:
                fb["traffic_source"] += " (" + measured.get("traffic_source", "") + ")"
            roof.update(fb)

    if rank == 0:
        total_maps = world * B * args.steps
        value = total_maps / elapsed
        detail = {
            "metric": METRIC if args.config == 2 else f"maps/sec for {S}x{S}x(4+N_cat) prediction fwd, batch {B} per GPU "
                                                        f"(SURVEY.md sec. 8d config {args.config})",
            "value": round(value, 3), "unit": "maps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.precision],
            "data": ("map sequence " + os.path.basename(args.maps)) if args.maps else "synthetic",
            "config": {"workload": f"{S}x{S}, {cfg.in_channels}-channel (4+{cfg.in_channels - 4}) partial maps -> "
                                   f"{cfg.num_classes}-class prediction forward (PSPNet R50-V1c-D8), "
                                   f"batch {B} per GPU, seeded random-init weights",
                       "survey_config": args.config, "global_batch": world * B,
                       "parallelism": f"dp{world} (map shards, no data-path collective)"},
            "gflop_per_map_nominal": round(conv_flops_per_map(cfg, S, S) / 1e9, 3),
            "whole_forward_tflops_nominal": round(value * conv_flops_per_map(cfg, S, S) / 1e12, 2),
            "algorithms": ALGORITHMS,
            "roofline": roof, "cpu_baseline": cpu,
        }
        if cpu is not None:
            detail["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
            for mname, m in modes.items():
                m["speedup_vs_cpu_baseline"] = round(m["value"] / cpu["value"], 1)
        if modes:
            detail["modes"] = modes
        if configs is not None:
            detail["configs"] = configs
        if gather_ms is not None:
            detail["allgather_maps_ms"] = round(gather_ms, 3)
            detail["allgather_maps_bytes_per_rank"] = int(out.numel() * 4)
            detail["allgather_maps_path"] = gather_path
            detail["rccl_ranks_seen"] = pdist.rccl_ranks_seen()       # ranks of the library's own communicator (peanut_comm_info); 0: none was built
        detail_path = write_detail(detail, args.detail)
        print(json.dumps(contract_line(detail, detail_path), separators=(",", ":")), flush=True)
    pdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if gather_failed and world > 1:
        # the line above is complete (the collation ran through torch.distributed), but peanut_allgather_maps -- the library's
        # own RCCL entry -- did not work: a multi-GPU run must not pass silently on the fallback
        print("bench.py: peanut_allgather_maps failed on this run (see allgather_maps_path)", file=sys.stderr)
        raise SystemExit(4)


if __name__ == "__main__":
    main()
