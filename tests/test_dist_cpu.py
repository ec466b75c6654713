"""N>1 path on CPU: world_size-2 gloo processes exercising the shard partition and the
logging-only all-gather (peanut_amd/dist.py).  No GPU, no HIP library needed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from peanut_amd import dist as pdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        r, lr, w = pdist.init_process_group(backend="gloo")
        assert (r, w) == (rank, world)
        # equal shards: rank-major collation of [B_local,K,H,W]
        local = torch.full((3, 6, 8, 8), float(rank)) + torch.arange(3.0)[:, None, None, None] * 0.1
        full = pdist.allgather_maps(local)
        assert full.shape == (world * 3, 6, 8, 8)
        for rr in range(world):
            assert torch.allclose(full[rr * 3:(rr + 1) * 3, 0, 0, 0], rr + torch.arange(3.0) * 0.1)
        # ragged shards of a 5-map global batch
        n = 5
        s, e = pdist.shard_range(n, rank, world)
        glob = torch.arange(float(n))[:, None, None, None].expand(n, 6, 4, 4).contiguous()
        counts = [pdist.shard_range(n, i, world)[1] - pdist.shard_range(n, i, world)[0] for i in range(world)]
        got = pdist.allgather_ragged(glob[s:e].contiguous(), counts)
        assert torch.equal(got, glob)
        # timing reduction used by bench.py
        assert pdist.max_over_ranks(float(rank + 1)) == float(world)
        pdist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        q.put((rank, repr(ex)))


def test_world2_gloo_allgather_and_shards():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_range_partition_properties():
    for n in (0, 1, 5, 8, 32, 100):
        for world in (1, 2, 3, 8):
            spans = [pdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pdist.shard_range(4, 2, 2)


def test_single_process_identity():
    t = torch.randn(2, 6, 4, 4)
    assert pdist.allgather_maps(t) is t
    assert pdist.max_over_ranks(3.5) == 3.5
