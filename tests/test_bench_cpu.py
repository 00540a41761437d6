"""bench.py's launch / argument / exchange / reporting plumbing at world_size 2 on CPU (gloo) with a stub engine (tests/bench_stub.py):
what the driver's N > 1 run exercises besides the kernels -- rendezvous from the environment, the rotated frame-batch plan and its ONE
all-gather per step, ragged prompts, max-over-ranks timing, the per-rank stage table, the sharded single-clip plan, the one JSON line --
and the watchdog that turns a dead peer into a diagnosable line instead of a silent time-out."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, extra_args=(), extra_env=None, timeout=300):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GVL_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2", **(extra_env or {}))
        env.pop("GVL_BENCH_SAME_DEVICE", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "bench_stub.py"), "--gpus", str(world), *extra_args],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill(); o, e = p.communicate()
        outs.append((p.returncode, o, e))
    return outs


@pytest.mark.parametrize("world,mode_args", [(2, ()), (2, ("--clips-per-step", "1")), (1, ("--mode", "serial"))])
def test_bench_main_end_to_end_on_cpu(world, mode_args):
    if __import__("torch").cuda.is_available():
        pytest.skip("plumbing test is for the GPU-less container")
    outs = _run(world, ("--steps", "3", "--warmup", "1", "--new-tokens", "6", *mode_args))
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][1]                          # ONE JSON line, rank 0 only
    assert all(not l.startswith("{") for rc, o, e in outs[1:] for l in o.splitlines())
    d = json.loads(lines[0])
    cps = 1 if ("--mode" in mode_args or "1" in mode_args) else 8
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["unit"] == "clips/s" and d["scaling"] == "weak"
    # whole-job clips / max-over-ranks time (ms_per_step is printed with two decimals: a sub-millisecond stub step needs the slack)
    assert abs(d["value"] - world * 3 * cps / (d["ms_per_step"] * 3e-3)) / d["value"] < max(1e-2, 0.011 / d["ms_per_step"])
    assert d["ids_match_serial"] is True                         # the last timed clip, alone and un-batched, gives the same ids: pixels, prompt and exchange all landed
    assert d["single_clip_latency_ms_sharded"] > 0 and d["config"]["clips_per_step"] == cps
    if world > 1:
        assert [p["rank"] for p in d["per_rank_stage_ms"]] == list(range(world))
        assert all(set(p) >= {"vision_ms", "exchange_ms", "prefill_ms", "decode_ms"} for p in d["per_rank_stage_ms"])
        assert d["n_ranks_seen_by_rccl"] is None                 # the stub has no RCCL communicator; a real engine reports ncclCommCount
    assert "clips_in_profiled_pass" in d["roofline"] and not any(k.endswith("_per_step") for k in d["stages"])


def test_bench_watchdog_names_the_hung_stage():
    if __import__("torch").cuda.is_available():
        pytest.skip("plumbing test is for the GPU-less container")
    outs = _run(2, ("--steps", "2", "--warmup", "1", "--new-tokens", "4", "--watchdog-s", "6"), {"GVL_STUB_HANG_RANK": "1", "GVL_STUB_HANG_AFTER": "3"}, timeout=120)
    rc0, o0, e0 = outs[0]
    assert rc0 == 3, (rc0, e0[-1500:])
    lines = [l for l in o0.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["hang"]["rank"] == 0 and d["hang"]["seconds_in_stage"] >= 6
    assert "step" in d["hang"]["stage"] or "barrier" in d["hang"]["stage"], d["hang"]
    assert len(d["per_rank_stage_ms"]) == 2                      # the diagnostics gathered BEFORE the timed region survive the hang
    assert "WATCHDOG rank 0/2 stuck in stage" in e0
    assert outs[1][0] == 3 and "WATCHDOG rank 1/2" in outs[1][2]


def test_bench_under_the_drivers_own_launcher():
    """The driver's N > 1 command line verbatim (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P <script> --gpus N --steps K --warmup W), with the stub engine on gloo: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come
    from the launcher itself, not from this test."""
    if __import__("torch").cuda.is_available():
        pytest.skip("plumbing test is for the GPU-less container")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GVL_BENCH_SAME_DEVICE")}
    env.update(GVL_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "bench_stub.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--new-tokens", "4"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["ids_match_serial"] is True
    assert [q["rank"] for q in d["per_rank_stage_ms"]] == [0, 1]


def test_a_hung_libgvl_communicator_check_does_not_cost_the_line():
    """The ncclCommCount / gvl_allgather_visual check of libgvl's own communicator runs last, in a thread with its own bound: if its bootstrap
    hangs, rank 0 still prints the complete line (value, roofline, per-rank stages; n_ranks_seen_by_rccl = "timed out") and every rank exits 0."""
    if __import__("torch").cuda.is_available():
        pytest.skip("plumbing test is for the GPU-less container")
    outs = _run(2, ("--steps", "2", "--warmup", "1", "--new-tokens", "4"), {"GVL_STUB_COMM_HANG": "1", "GVL_BENCH_DIAG_S": "3"}, timeout=180)
    for rc, o, e in outs:
        assert rc == 0, e[-1500:]
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["n_ranks_seen_by_rccl"] == "timed out" and "hang" not in d and len(d["per_rank_stage_ms"]) == 2 and "roofline" in d


@pytest.mark.parametrize("exchange", ["gvl", "torch"])
def test_bench_exchange_through_the_engines_own_communicator_and_rank0_llm_mode(exchange):
    """--exchange gvl: the all-gather of the visual tokens goes through the ENGINE's communicator (gvl_comm_init + gvl_allgather_visual on a real
    engine; the stub plays it with gloo) in the timed region, the sharded single clip and the rank-0-LLM mode; the other collective is timed as an
    extra.  The rank-0-LLM pipelined mode (every rank encodes, rank 0 alone runs the LLM of the round's clips) must give, clip by clip, the ids
    the clip's own rank produces in the weak-scaling plan."""
    if __import__("torch").cuda.is_available():
        pytest.skip("plumbing test is for the GPU-less container")
    outs = _run(2, ("--steps", "3", "--warmup", "1", "--new-tokens", "6", "--exchange", exchange), {"GVL_STUB_COMM": "1"})
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][1]
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["ids_match_serial"] is True and d["exchange"].startswith(exchange)
    other = d["clips_per_s_other_exchange"]
    assert other["exchange"] == ("torch" if exchange == "gvl" else "gvl") and other["clips_per_s"] > 0, other
    r0 = d["rank0_llm_pipelined"]
    assert r0["clips_per_s"] > 0 and r0["clips_per_round"] == 2 and r0["ids_match_the_per_rank_llm"] is True, r0
    assert d["n_ranks_seen_by_rccl"] == 2 and d["gvl_allgather_matches_rank_order"] is True


@pytest.mark.parametrize("segs", [12, 32])
def test_bench_main_at_world_8_through_the_rotated_plan_and_the_rank0_llm_mode(segs):
    """the configurations nobody has run on 8 devices yet, at WORLD 8 on CPU (gloo, stub engine): 12 segments = BASELINE configs[2] (blocks of
    2,2,2,2,1,1,1,1 rotated per clip) and 32 segments = configs[4] (4 per rank), through the rotated encode plan, the one all-gather per step over all clips of the
    round, the sharded single clip and the rank-0-LLM pipelined mode; ids must match the per-rank plan clip by clip and the line must carry the 8 per-rank stage
    records a first hardware failure would be diagnosed from."""
    if __import__("torch").cuda.is_available():
        pytest.skip("plumbing test is for the GPU-less container")
    outs = _run(8, ("--steps", "2", "--warmup", "1", "--new-tokens", "4", "--exchange", "gvl"), {"GVL_STUB_COMM": "1", "GVL_BENCH_SEGS": str(segs)}, timeout=600)
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][1]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["ids_match_serial"] is True and d["scaling"] == "weak"
    assert [p["rank"] for p in d["per_rank_stage_ms"]] == list(range(8))
    r0 = d["rank0_llm_pipelined"]
    assert r0["clips_per_s"] > 0 and r0["clips_per_round"] == 8 and r0["ids_match_the_per_rank_llm"] is True, r0
    assert d["n_ranks_seen_by_rccl"] == 8 and d["gvl_allgather_matches_rank_order"] is True
    assert d["single_clip_latency_ms_sharded"] > 0
