"""RCCL on the GPU box: the collectives bench.py / peanut_amd.dist use for N > 1, with min(2, visible GPUs) ranks
(one process per GPU, launched like the driver launches bench.py), the C-ABI communicator with one rank, and
bench.py's own rank spawning."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_ranks_allgather_through_the_c_abi():
    n = min(2, torch.cuda.device_count())
    assert n >= 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_rccl_worker.py")]
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert f"RCCL_WORKER_OK world={n}" in r.stdout, r.stdout[-3000:]


def test_single_rank_communicator_and_rccl_binding():
    """peanut_comm_* with one rank (no id needed) copies the shard; the unique id comes from the RCCL copy torch
    already loaded (no second RCCL in the process)."""
    from peanut_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.peanut_comm_create(C.byref(h), 1, 0, None), "peanut_comm_create")
    n, r = C.c_int(), C.c_int()
    _lib.check(lib.peanut_comm_info(h, C.byref(n), C.byref(r)), "peanut_comm_info")
    assert (n.value, r.value) == (1, 0)
    local = torch.rand(2, 6, 16, 16, device="cuda")
    out = torch.zeros_like(local)
    _lib.check(lib.peanut_allgather_maps(h, local.data_ptr(), out.data_ptr(), local.numel(), _lib.current_stream_ptr()),
               "peanut_allgather_maps")
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    lib.peanut_comm_destroy(h)
    ident = (C.c_ubyte * 128)()
    _lib.check(lib.peanut_comm_unique_id(C.byref(ident)), "peanut_comm_unique_id")
    assert any(ident) and b"rccl" in lib.peanut_comm_backend()
    assert lib.peanut_comm_create(C.byref(h), 2, 5, C.byref(ident)) != 0        # rank out of range


def test_bench_spawns_its_own_ranks_or_refuses():
    """`python bench.py --gpus N` without a launcher must run N ranks (n_gpus == N in the JSON line) or exit
    non-zero -- never report a smaller job under that flag."""
    have = torch.cuda.device_count()
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "96",
            "--no-cpu-baseline", "--also", "", "--traffic", "none"]
    r = subprocess.run(base + ["--gpus", str(have + 1)], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout
    n = min(2, have)
    if n > 1:
        r = subprocess.run(base + ["--gpus", str(n)], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == n and line["config"]["global_batch"] == 2 * n and "allgather_maps_ms" in line
    # config-5 preset (per-GPU shard shrunk to keep the test short): the all-gather is timed and reported
    r = subprocess.run(base[:2] + ["--config", "5", "--steps", "1", "--warmup", "1", "--batch", "1", "--size", "192", "--no-cpu-baseline",
                                   "--also", "", "--traffic", "none"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["survey_config"] == 5 and "25-channel" in line["config"]["workload"]
    assert "allgather_maps_ms" in line


def _run_bench(args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = r.stdout.splitlines()
    lines = [ln for ln in out_lines if ln.startswith("{")]
    # the driver parses the LAST stdout line: exactly one JSON line, last, strict JSON, under 4 KB
    assert len(lines) == 1 and out_lines[-1] == lines[0]
    assert len(lines[0].encode()) < 4096, len(lines[0])
    return json.loads(lines[0])


def _check_contract(line, steps, warmup, batch):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in line, k
    assert line["unit"] == "maps/s" and line["n_gpus"] == 1 and line["steps"] == steps and line["warmup"] == warmup
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - batch * steps / (line["ms_per_step"] * steps * 1e-3)) <= 0.01 * line["value"]
    roof = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "launches_per_step",
              "avg_launch_ms"):
        assert k in roof, k
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 157.3
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = line["cpu_baseline"]
    assert set(cpu) == {"value", "unit", "cores", "kind", "cpu_model", "sample"}
    assert cpu["kind"] == "port" and cpu["unit"] == "maps/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]


@pytest.mark.slow
def test_bench_line_carries_the_contract_fields(tmp_path):
    """The ONE JSON line of bench.py (small shape, short run): the driver's fields, the roofline and cpu_baseline objects reduced
    to their numbers, flat summaries of the requested modes; the per-family tables and the modes' own rooflines are in the detail
    file the line names."""
    detail_path = str(tmp_path / "detail.json")
    line = _run_bench(["--steps", "3", "--warmup", "1", "--batch", "2", "--size", "96", "--also", "bf16x6,fp16x3", "--traffic", "none",
                       "--detail", detail_path])
    _check_contract(line, 3, 1, 2)
    assert set(line["modes_summary"]) == {"bf16x6", "fp16x3"} and all(m["value"] > 0 for m in line["modes_summary"].values())
    assert {"nchw_to_nhwc", "maxpool", "ppm_pool", "upsample_logits"} <= set(line["hbm_bound_frac_of_8tbs"])
    with open(detail_path) as fh:
        detail = json.load(fh)
    assert detail["value"] == line["value"] and detail["roofline"]["frac"] == line["roofline"]["frac"]
    hbm = detail["roofline"]["hbm_bound_kernels"]
    assert {"nchw_to_nhwc", "maxpool", "ppm_pool", "upsample_logits"} <= set(hbm) and all(v["gb_s"] > 0 for v in hbm.values())
    assert set(detail["modes"]) == {"bf16x6", "fp16x3"}
    for m in detail["modes"].values():
        assert m["value"] > 0 and m["roofline"]["peak"] in (416.7, 833.3) and "speedup_vs_cpu_baseline" in m


@pytest.mark.slow
def test_bench_default_command_form_prints_a_parseable_contract_line():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the driver's own command form, every default in force (headline at
    480 x 480 x 14, batch 32, PMC traffic passes, CPU baseline, config 4 with the detector and the mapping stage): the last
    stdout line is the contract object, strict JSON under 4 KB, with a measured roofline, a cpu_baseline and the flat
    configs_summary (round 5's line was 23 KB and the driver recorded parsed = null)."""
    t0 = __import__("time").perf_counter()
    line = _run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], timeout=1500)
    wall = __import__("time").perf_counter() - t0
    _check_contract(line, 20, 5, 32)
    assert line["config"]["survey_config"] == 2 and line["config"]["global_batch"] == 32
    import bench
    assert line["roofline"]["kernel"] == bench.DOMINANT_FAMILY_FP32 and line["roofline"]["frac"] > 0.5
    cs = line["configs_summary"]
    assert "error" not in cs, cs
    for k in ("config4_steps_s", "detector_b1_ms_per_frame", "pred720_b1_ms_per_map", "mapping_steps_s", "config3_images_s"):
        assert cs.get(k) and cs[k] > 0, (k, cs)
    assert os.path.exists(os.path.join(ROOT, line["detail"]))
    print(f"default bench form: {wall:.0f} s wall, line {len(json.dumps(line, separators=(',', ':')))} bytes, "
          f"{line['value']} maps/s, frac {line['roofline']['frac']}, traffic {line['roofline']['traffic']}")
