"""Multi-GPU layer of the hot path: one process per GPU, maps / episodes sharded data-parallel,
no collective on the data path.  The only collective is an all-gather of predicted maps to
collate them for logging (BASELINE.json north_star), over RCCL/xGMI (``backend='nccl'`` on ROCm)
or gloo on CPU for tests.

The reference shards by hand with ``--start_ep/--end_ep/--sem_gpu_id`` (nav/arguments.py:15-20,
nav/collect.py:37-39,50) and never communicates; :func:`shard_range` reproduces that partition.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from the env if WORLD_SIZE > 1.  ``nccl`` (= RCCL) when a GPU
    is visible, else ``gloo``.  Binds this process to GPU ``LOCAL_RANK``."""
    rank, local_rank, world = env_rank_world()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of ``n_items`` units (maps of a batch, or episodes) owned by
    ``rank`` -- the ``--start_ep/--end_ep`` partition; the first ``n_items % world`` ranks get
    one extra unit, empty shards are allowed when world > n_items."""
    if n_items < 0 or world < 1 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


class MapComm:
    """``peanut_comm_t`` of the C ABI (include/peanut_hip.h, csrc/comm.hip) spanning the torch.distributed world:
    rank 0 creates the RCCL unique id, the 128 bytes travel through the process group that already exists, every
    rank builds its communicator on its own GPU.  The collective itself is then one ``peanut_allgather_maps`` call
    on torch's current stream -- the same entry point a torch-free host uses."""

    def __init__(self, device=None):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # Every step that can fail on one rank only is followed by a collective exchange of its status, so that all ranks
        # raise (or carry on) TOGETHER: a rank that raised alone would leave the others blocked in the next collective.
        ident = (C.c_ubyte * 128)()
        status, err = 0, ""
        if self.rank == 0:
            try:
                _lib.check(self._lib.peanut_comm_unique_id(C.byref(ident)), "peanut_comm_unique_id")
            except Exception as e:  # noqa: BLE001 - reported to every rank below
                status, err = 1, str(e)
        t = torch.tensor([status] + list(ident), dtype=torch.uint8)
        on_gpu = dist.get_backend() == "nccl"
        if on_gpu:
            t = t.to(self.device)
        dist.broadcast(t, src=0)
        vals = t.cpu().tolist()
        if vals[0] != 0:
            raise _lib.PeanutHipError("MapComm: rank 0 could not create the communicator id" + (f": {err}" if err else ""))
        for i, v in enumerate(vals[1:]):
            ident[i] = v
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._lib.peanut_comm_create(C.byref(self._h), self.world, self.rank, C.byref(ident))
        msg = (self._lib.peanut_last_error() or b"").decode() if rc else ""
        bad = torch.tensor([1 if rc else 0], dtype=torch.int32, device=self.device if on_gpu else "cpu")
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            if not rc:                       # this rank's communicator is fine, a peer's is not: give it back
                self.close()
            raise _lib.PeanutHipError("MapComm: peanut_comm_create failed on " + ("this rank: " + msg if rc else "another rank"))
        self.backend = (self._lib.peanut_comm_backend() or b"").decode()

    def allgather_maps(self, local: torch.Tensor) -> torch.Tensor:
        from . import _lib
        if not local.is_cuda or local.dtype != torch.float32:
            raise ValueError("MapComm.allgather_maps needs a float32 tensor on the HIP device")
        local = local.contiguous()
        out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=torch.float32, device=local.device)
        with torch.cuda.device(local.device):
            rc = self._lib.peanut_allgather_maps(self._h, local.data_ptr(), out.data_ptr(), local.numel(),
                                                 _lib.current_stream_ptr(local.device))
        _lib.check(rc, "peanut_allgather_maps")
        return out

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.peanut_comm_destroy(h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover - interpreter shutdown
            pass


_MAP_COMM: Optional[MapComm] = None


def map_comm() -> MapComm:
    """The process-wide communicator (created collectively on first use: every rank must reach this call)."""
    global _MAP_COMM
    if _MAP_COMM is None:
        _MAP_COMM = MapComm()
    return _MAP_COMM


def rccl_ranks_seen() -> int:
    """Ranks of the library's own RCCL communicator as ``peanut_comm_info`` reports them (0 when none has been built in
    this process: one rank, or a host-tensor all-gather) -- bench.py prints it so that a multi-GPU record shows how many
    ranks ``peanut_allgather_maps`` really spanned."""
    if _MAP_COMM is None or not getattr(_MAP_COMM, "_h", None):
        return 0
    import ctypes as C
    n, r = C.c_int(0), C.c_int(0)
    if _MAP_COMM._lib.peanut_comm_info(_MAP_COMM._h, C.byref(n), C.byref(r)) != 0:
        return 0
    return int(n.value)


def close_map_comm():
    global _MAP_COMM
    if _MAP_COMM is not None:
        _MAP_COMM.close()
        _MAP_COMM = None


def allgather_maps(local: torch.Tensor, world: Optional[int] = None) -> torch.Tensor:
    """Collate equally-shaped predicted-map shards [B_local,K,H,W] from every rank into
    [world*B_local,K,H,W] (rank-major) with ONE all-gather; logging only, off the data path.
    With one process it is the identity.  HIP tensors go through the library's own RCCL entry point
    (``peanut_allgather_maps``); host tensors (the gloo world-size-2 tests of the sharding logic, no GPU) through
    ``torch.distributed``."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    if local.is_cuda and local.dtype == torch.float32:
        return map_comm().allgather_maps(local)
    world = world or dist.get_world_size()
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


def allgather_ragged(local: torch.Tensor, counts: List[int]) -> torch.Tensor:
    """Ragged variant (shards of different B_local, e.g. 10 maps over 4 ranks): pads to the
    largest shard, gathers once, strips the padding."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if len(counts) != world:
        raise ValueError("counts must have one entry per rank")
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    full = allgather_maps(pad, world)
    parts = [full[r * mx: r * mx + counts[r]] for r in range(world)]
    return torch.cat(parts, 0)


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
