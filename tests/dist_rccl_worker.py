"""Worker of tests/test_dist_gpu.py: one rank of an N-process RCCL job (launched through torch.distributed.run,
one process per GPU).  Exercises what bench.py / peanut_amd.dist use for N > 1: barrier, all_reduce(MAX), and the
library's own ``peanut_allgather_maps`` (C ABI, RCCL bound at run time) against torch's all_gather_into_tensor."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from peanut_amd import dist as pdist  # noqa: E402


def main():
    rank, local_rank, world = pdist.init_process_group()
    if not dist.is_initialized():            # a 1-GPU box: still go through RCCL, with a single rank
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    assert world == int(os.environ["WORLD_SIZE"]) and dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    pdist.barrier()
    assert pdist.max_over_ranks(float(rank + 1), device=dev) == float(world)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.rand((3, 6, 40, 40), generator=g).to(dev)
    full = pdist.allgather_maps(local)                      # peanut_allgather_maps
    want = torch.empty((world * 3, 6, 40, 40), device=dev)
    dist.all_gather_into_tensor(want, local)
    assert full.shape == want.shape and torch.equal(full, want)
    for r in range(world):
        ref = torch.rand((3, 6, 40, 40), generator=torch.Generator().manual_seed(100 + r)).to(dev)
        assert torch.equal(full[r * 3:(r + 1) * 3], ref), f"shard of rank {r} misplaced"
    comm = pdist.map_comm()
    assert (comm.rank, comm.world) == (rank, world) and "rccl" in comm.backend
    # ragged shards of a 5-map global batch
    n = 5
    s, e = pdist.shard_range(n, rank, world)
    glob = torch.arange(float(n), device=dev)[:, None, None, None].expand(n, 6, 4, 4).contiguous()
    counts = [pdist.shard_range(n, i, world)[1] - pdist.shard_range(n, i, world)[0] for i in range(world)]
    assert torch.equal(pdist.allgather_ragged(glob[s:e].contiguous(), counts), glob)
    torch.cuda.synchronize()
    pdist.barrier()
    pdist.close_map_comm()
    dist.destroy_process_group()
    if rank == 0:
        print(f"RCCL_WORKER_OK world={world} backend={comm.backend}", flush=True)


if __name__ == "__main__":
    main()
