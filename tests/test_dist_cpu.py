"""world_size-2 gloo tests (CPU) of the multi-GPU plan: segment sharding + the visual-token all-gather."""
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa: E402,F401  (spawned workers re-import this module without conftest)

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from grounded_video_llm_amd import dist as gdist


def test_shard_bounds():
    assert gdist.shard_bounds(12, 8) == [(0, 2), (2, 4), (4, 6), (6, 8), (8, 9), (9, 10), (10, 11), (11, 12)]
    assert gdist.shard_bounds(12, 1) == [(0, 12)]
    assert gdist.shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    for n in range(0, 40):
        for w in range(1, 9):
            b = gdist.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _worker(rank, world, port, n_units, rows, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_units * rows * 4, dtype=torch.float32).view(n_units * rows, 4)
        lo, hi = gdist.my_shard(n_units, rank, world)
        got = gdist.allgather_visual(full[lo * rows: hi * rows].clone(), n_units, rows)
        q.put((rank, bool(torch.equal(got, full))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_units,rows", [(12, 5), (3, 7), (1, 4)])
def test_allgather_visual_gloo_world2(n_units, rows):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_units, rows, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert res == [(0, True), (1, True)]


def _worker_v(rank, world, port, n_units, rows, q):
    """dist.allgather_visual(gatherv=...): the routing of the uneven-block exchange (Engine.allgatherv_visual = gvl_allgatherv_visual on a GPU box) -- the
    callable gets this rank's block and the per-rank row counts in rank order and must return the segment-ordered whole; emulated with gloo here."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_units * rows * 4, dtype=torch.float32).view(n_units * rows, 4)
        lo, hi = gdist.my_shard(n_units, rank, world)
        seen = {}

        def fake_gatherv(local, rows_per_rank):
            seen["rows"] = list(rows_per_rank)
            parts = [None] * world
            dist.all_gather_object(parts, local)
            assert [p.shape[0] for p in parts] == seen["rows"]
            return torch.cat(parts, 0)
        got = gdist.allgather_visual(full[lo * rows: hi * rows].clone(), n_units, rows, gatherv=fake_gatherv)
        want_rows = [(h - l) * rows for l, h in gdist.shard_bounds(n_units, world)]
        q.put((rank, bool(torch.equal(got, full)) and seen["rows"] == want_rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_units,rows", [(3, 7), (12, 5)])
def test_allgatherv_routing_gloo_world2(n_units, rows):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_v, args=(r, 2, port, n_units, rows, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(60) for p in ps]
    assert res == [(0, True), (1, True)]


def test_rotated_plan_balances_and_reassembles():
    """bench.py's N-clips-in-flight plan: every rank encodes exactly n units, and after the all-gather every rank can
    rebuild ITS clip's units in order (simulated without processes)."""
    n = 12
    for world in (1, 2, 3, 4, 8):
        enc = {}
        for r in range(world):
            plan = gdist.rotated_encode_plan(n, r, world)
            assert sum(h - l for _, l, h in plan) == n                      # balanced: 12 segments per rank
            enc[r] = [(c, u) for c, lo, hi in plan for u in range(lo, hi)]  # what rank r holds, in encode order
        covered = sorted(x for r in range(world) for x in enc[r])
        assert covered == [(c, u) for c in range(world) for u in range(n)]  # every (clip, unit) encoded exactly once
        for r in range(world):
            got = []
            for src, off, cnt in gdist.rotated_gather_index(n, r, world):
                got += enc[src][off: off + cnt]
            assert got == [(r, u) for u in range(n)], (world, r, got)


def _bench_exchange_worker(rank, world, port, cps, q):
    """bench.py's per-STEP exchange on gloo: each rank 'encodes' the 12 segment blocks of its rotated plan for `cps` clip rounds (block
    content = a code of (round, clip, segment, row)), ONE all-gather, and must end up with clip == rank's 12 segments in order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GVL_BENCH_BACKEND="gloo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        L, H = 3, 8
        st = bench.Stepper.__new__(bench.Stepper)
        st.world, st.rank, st.dev, st.L = world, rank, torch.device("cpu"), L
        st.mine = gdist.rotated_encode_plan(12, rank, world)
        st.gather = gdist.rotated_gather_index(12, rank, world)

        def block(rnd, clip, seg):
            idx = (rnd * world + clip) * 12 + seg                       # < 72 for cps <= 3: every (block, row) code is an integer <= 216,
            return (torch.arange(L, dtype=torch.float32)[:, None] + 3.0 * idx).expand(L, H).to(torch.bfloat16)   # exact in bf16, all distinct

        vis_list = [torch.cat([block(rnd, c, u) for c, lo, hi in st.mine for u in range(lo, hi)], 0) for rnd in range(cps)]
        got = st._exchange_multi(vis_list)
        ok = len(got) == cps and all(torch.equal(got[rnd], torch.cat([block(rnd, rank, u) for u in range(12)], 0)) for rnd in range(cps))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cps", [1, 3])
def test_bench_step_exchange_gloo_world2(cps):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bench_exchange_worker, args=(r, 2, port, cps, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in ps)
    [p.join(60) for p in ps]
    assert res == [(0, True), (1, True)]
