"""RCCL smoke on the GPU box: the collectives bench.py / peanut_amd.dist use for N > 1 (barrier,
all_reduce(MAX), all_gather_into_tensor) run on the `nccl` (= RCCL) backend with a single rank.  The
multi-rank logic itself is covered by the world-size-2 gloo tests (tests/test_dist_cpu.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_collectives():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        torch.cuda.set_device(0)
        dist.barrier()
        t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 3.5
        local = torch.rand(2, 6, 16, 16, device="cuda")
        out = torch.empty_like(local)
        dist.all_gather_into_tensor(out, local)
        assert torch.equal(out, local)
    finally:
        dist.destroy_process_group()
