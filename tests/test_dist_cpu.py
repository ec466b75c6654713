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


class _CountingAgent:
    """Stands in for PEANUT_Agent in the outer-loop test: counts resets and frames, 'predicts' every second frame."""

    def __init__(self):
        self.resets, self.frames = 0, 0

    def reset(self):
        self.resets += 1

    def act(self, observations):
        self.frames += 1
        assert set(observations) >= {"rgb", "depth", "gps", "compass", "objectgoal"}
        return {"predicted": self.frames % 2 == 0}


def _worker8(rank, world, port, ep_dir, n_eps, q):
    try:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if root not in sys.path:
            sys.path.insert(0, root)
        import bench
        from peanut_amd import replay
        r, lr, w = pdist.init_process_group(backend="gloo")
        assert (r, w) == (rank, world)
        # (1) bench.py's map shards: rank r owns global maps [r * B, (r + 1) * B) -- the seeds make the union of the shards
        #     exactly the global batch a single process would generate
        B, C, S = 2, 6, 32
        mine = bench.synth_maps(B, C, S, "cpu", seed0=rank * B)
        allmaps = pdist.allgather_maps(mine)
        assert allmaps.shape == (world * B, C, S, S)
        assert torch.equal(allmaps, bench.synth_maps(world * B, C, S, "cpu", seed0=0))
        # (2) episode shards (nav/arguments.py:15-20: --start_ep / --end_ep per process): contiguous, disjoint, complete
        paths = [os.path.join(ep_dir, f"ep{i:02d}.npz") for i in range(n_eps)]
        ids = replay.episode_shard(n_eps)
        s, e = pdist.shard_range(n_eps, rank, world)
        assert ids == list(range(s, e))
        agent = _CountingAgent()
        done = replay.run_recorded_shard(agent, paths, start_ep=s, end_ep=e) if e > s else {}
        assert sorted(done) == ids and agent.resets == len(ids)
        gathered = [None] * world
        dist.all_gather_object(gathered, (ids, dict(done), agent.frames))
        owners = [i for g in gathered for i in g[0]]
        assert owners == list(range(n_eps))                           # every episode exactly once, in rank order
        assert sum(g[2] for g in gathered) == sum(3 + (i % 3) for i in range(n_eps))
        # (3) the bench's timing reduction
        assert pdist.max_over_ranks(float(rank + 1)) == float(world)
        pdist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, repr(ex) + traceback.format_exc()[-600:]))


@pytest.mark.parametrize("n_eps", [8, 11])
def test_world8_gloo_map_and_episode_shards(tmp_path, n_eps):
    """SURVEY.md sec. 8d config 4 / 5 at the driver's scale: EIGHT ranks (gloo, CPU).  bench.py's shard seeds, the episode
    windows of the reference's outer loop (8 episodes over 8 GPUs; 11 for a ragged split) and the rank-major collation."""
    import numpy as np
    from peanut_amd import episodes as E
    for i in range(n_eps):
        frames = [{"rgb": np.zeros((4, 4, 3), np.uint8), "depth": np.zeros((4, 4, 1), np.float32),
                   "gps": np.zeros(2, np.float32), "compass": np.zeros(1, np.float32), "objectgoal": np.array([i % 6])}
                  for _ in range(3 + (i % 3))]
        E.save_episode(str(tmp_path / f"ep{i:02d}.npz"), frames)
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, str(tmp_path), n_eps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


# ---- bench.py's own main() at N > 1, on the CPU: the product model replaced by a host-side stand-in at the point where main()
# asks its backend for one (bench.HipBackend.make_model -> PEANUT_Prediction_Model -> the C ABI) ----
class _StubSegmentor:
    def probe_enable(self, on):
        pass

    def probe_collect(self):          # (forwards, [(op, kernel family, summed ms, flops, bytes)])
        import bench
        return 2, [("backbone.layer4.0.conv1", bench.DOMINANT_FAMILY_FP32, 2.0, 1.0e9, 1.0e6), ("upsample_logits", "upsample_logits", 0.2, 0.0, 1.0e6)]


class _StubModel:
    def __init__(self, rank):
        self.rank, self.model = rank, _StubSegmentor()

    def get_prediction_batch(self, x, apply_sigmoid=True, out=None):
        out.fill_(float(self.rank))
        return out


def _bench_worker(rank, world, port, q, break_gather):
    import contextlib
    import io
    import json
    try:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import bench

        class StubBackend:
            name = "stub"
            device = torch.device("cpu")

            def synchronize(self):
                pass

            def make_model(self, cfg, sd, precision):
                return _StubModel(rank)

            def state_dict(self, cfg):
                return {}

        if break_gather:               # the library's all-gather entry fails: main() must fall back, finish the line and exit 4
            real, calls = bench.pdist.allgather_maps, []

            def flaky(t):
                calls.append(1)
                if len(calls) == 1:
                    raise RuntimeError("simulated peanut_allgather_maps failure")
                return real(t)
            bench.pdist.allgather_maps = flaky
        buf, code = io.StringIO(), 0
        with contextlib.redirect_stdout(buf):
            try:
                import tempfile
                detail = os.path.join(tempfile.gettempdir(), f"peanut_bench_detail_{port}.json")
                bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "2", "--size", "32", "--detail", detail],
                           backend_factory=StubBackend)
            except SystemExit as e:
                code = e.code if isinstance(e.code, int) else 1
        out_lines = buf.getvalue().splitlines()
        lines = [l for l in out_lines if l.startswith("{")]
        # the driver reads the LAST stdout line: it must be the (only) JSON line, strict JSON, under the contract's size bound
        assert not lines or (out_lines[-1] == lines[-1] and len(lines[-1].encode()) < bench.CONTRACT_LINE_MAX_BYTES), out_lines[-1][:200]
        full = []
        if lines and os.path.exists(detail):
            with open(detail) as fh:
                full = [json.load(fh)]
            os.remove(detail)
        q.put((rank, code, [json.loads(l) for l in lines], full))
    except Exception as ex:  # pragma: no cover
        import traceback
        q.put((rank, -1, traceback.format_exc() + repr(ex), []))


@pytest.mark.parametrize("break_gather", [False, True], ids=["gather_ok", "gather_falls_back"])
def test_bench_main_runs_its_multi_rank_control_flow_under_gloo(break_gather):
    """bench.py's main() with WORLD_SIZE = 2 on gloo and a stand-in model: both ranks pass the barriers, the step time is the
    max over ranks, ONLY rank 0 prints the JSON line, the line carries n_gpus = 2, weak scaling, the whole-job value, the
    all-gather time, rccl_ranks_seen and a roofline whose traffic came from the committed file (no PMC child pass at
    N > 1), the extra precision modes are skipped -- and when the library's all-gather entry fails, the run still prints
    its complete line through the torch.distributed fallback but exits with status 4 on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, break_gather)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, code, lines, _full in res:
        assert code == (4 if break_gather else 0), (rank, code, lines)
    assert len(res[0][2]) == 1 and res[1][2] == [], res
    line, detail = res[0][2][0], res[0][3][0]
    assert line["detail"] and detail["value"] == line["value"] and detail["roofline"]["kernel"] == line["roofline"]["kernel"]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert line["config"]["global_batch"] == 4 and line["steps"] == 3 and line["warmup"] == 1
    assert abs(line["value"] - 4 * 3 / (line["ms_per_step"] * 3e-3)) <= 0.01 * line["value"] + 1e-3
    assert "modes_summary" not in line and "configs_summary" not in line and line["cpu_baseline"] is None
    assert "modes" not in detail and "configs" not in detail
    assert line["allgather_maps_ms"] >= 0 and line["allgather_maps_bytes_per_rank"] == 2 * 6 * 32 * 32 * 4
    assert line["rccl_ranks_seen"] == 0                      # host tensors: the library's RCCL communicator was not built
    assert ("failed" in line["allgather_maps_path"]) == break_gather
    roof = line["roofline"]
    import bench
    # the headline's CURRENT dominant family (bench.DOMINANT_FAMILY_FP32; tests/test_pred_gpu.py checks on the GPU that it still is)
    # must be in profiles/hbm_traffic.json: at N > 1 no PMC child pass runs and the line's traffic comes from that file
    assert roof["kernel"] == bench.DOMINANT_FAMILY_FP32 and roof["bound"] == "mfma"
    assert roof["traffic"] is not None and roof["traffic"] > 0 and roof["traffic_measured_by_this_run"] is False
    assert "NOT measured by this run" in detail["roofline"]["traffic_source"]
