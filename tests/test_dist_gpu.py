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


@pytest.mark.slow
def test_bench_line_carries_the_contract_fields():
    """The ONE JSON line of bench.py (small shape, short run): the driver's fields, the roofline object with the HBM-bound
    families reported against bandwidth, a cpu_baseline object (bounded sample), and every requested mode with its own
    roofline."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "2", "--size", "96",
           "--also", "bf16x6,fp16x3", "--traffic", "none"]
    r = subprocess.run(cmd, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "modes"):
        assert k in line, k
    assert line["unit"] == "maps/s" and line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 2 * 3 / (line["ms_per_step"] * 3e-3)) <= 0.01 * line["value"]
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 157.3 and "traffic" in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    hbm = roof["hbm_bound_kernels"]
    assert {"nchw_to_nhwc", "maxpool", "ppm_pool", "upsample_logits"} <= set(hbm) and all(v["gb_s"] > 0 for v in hbm.values())
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "maps/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert set(line["modes"]) == {"bf16x6", "fp16x3"}
    for m in line["modes"].values():
        assert m["value"] > 0 and m["roofline"]["peak"] in (416.7, 833.3) and "speedup_vs_cpu_baseline" in m
