"""-m gpu: first contact with RCCL on a 1-GPU box (missing #5): the nccl backend of torch.distributed at world_size 1
and libgvl's own communicator (gvl_comm_init / gvl_allgather_visual through the C ABI) both run a REAL (degenerate) all-gather on the
device -- library load, communicator init, bf16 dtype, stream ordering -- before the driver's N > 1 runs do.  The multi-rank
arithmetic of the plan is covered on CPU by tests/test_dist_cpu.py (gloo, world_size 2)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, bf, tiny_geo  # noqa: E402
from grounded_video_llm_amd import dist as gdist, engine as E  # noqa: E402


@pytest.fixture(scope="module")
def nccl_world1():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    torch.distributed.destroy_process_group()


def test_torch_nccl_allgather_visual_world1(nccl_world1):
    L, H, n = 285, 3072, 12
    g = torch.Generator(device=DEV); g.manual_seed(1)
    local = torch.randn((n * L, H), device=DEV, generator=g).to(bf)
    out = gdist.allgather_visual(local, n, L, None, force_collective=True)        # all_gather_into_tensor on RCCL, bf16, padded blocks
    assert out.dtype == bf and torch.equal(out, local)


def test_bench_exchange_path_world1(nccl_world1):
    """bench.py's per-step exchange (ONE all-gather for the step's clip rounds + segment-order re-assembly) on the nccl backend."""
    import bench
    st = bench.Stepper.__new__(bench.Stepper)
    st.world, st.rank, st.dev, st.L = 1, 0, torch.device(DEV), 285
    g = torch.Generator(device=DEV); g.manual_seed(2)
    vis = [torch.randn((12 * 285, 3072), device=DEV, generator=g).to(bf) for _ in range(4)]
    side = torch.cuda.Stream(DEV)                                                    # the bench runs the exchange on its vision stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = st._exchange_multi(vis, force=True)
    torch.cuda.current_stream().wait_stream(side)
    assert len(got) == 4 and all(torch.equal(a, b) for a, b in zip(got, vis))
    # --exchange gvl: the same step exchange through libgvl's own communicator (gvl_comm_init over the process group + gvl_allgather_visual)
    eng = E.Engine(tiny_geo(), DEV, towers=())
    gdist.init_gvl_comm(eng)
    st.eng, st.exchange = eng, "gvl"
    with torch.cuda.stream(side):
        got = st._exchange_multi(vis, force=True)
        allc = st._exchange_multi(vis[:1], force=True, all_clips=True)              # the rank-0-LLM plan's view: every clip of the round
    torch.cuda.current_stream().wait_stream(side)
    assert len(got) == 4 and all(torch.equal(a, b) for a, b in zip(got, vis))
    assert len(allc) == 1 and len(allc[0]) == 1 and torch.equal(allc[0][0], vis[0])
    assert eng.comm_count() == 1
    eng.close()


def test_c_abi_rccl_communicator_world1():
    """gvl_comm_unique_id / gvl_comm_init / gvl_allgather_visual: what a non-Python host calls (include/gvl.h)."""
    eng = E.Engine(tiny_geo(), DEV, towers=())
    local = (torch.arange(7 * 64, device=DEV, dtype=torch.float32).view(7, 64) / 17).to(bf)
    assert torch.equal(eng.allgather_visual(local), local)          # no communicator yet: single-rank copy
    uid = eng.comm_unique_id()
    assert len(uid) == 128 and uid != bytes(128)
    eng.comm_init(uid, 0, 1)
    s = torch.cuda.Stream(DEV)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = eng.allgather_visual(local)                             # ncclAllGather on the caller's (side) stream
    torch.cuda.current_stream().wait_stream(s)
    assert out.shape == local.shape and torch.equal(out, local)
    with pytest.raises(E.L.GvlError):
        eng.comm_init(uid, 0, 1)                                      # already initialised
    # gvl_allgatherv_visual (uneven blocks straight into the segment-ordered buffer): world 1 = one ncclBroadcast inside a group, into a caller-owned prefix
    pre = torch.zeros((9, 64), device=DEV, dtype=bf)
    with torch.cuda.stream(s):
        got = eng.allgatherv_visual(local, [7], out=pre[:7])
    torch.cuda.current_stream().wait_stream(s)
    assert got.data_ptr() == pre.data_ptr() and torch.equal(pre[:7], local) and float(pre[7:].abs().max()) == 0.0
    eng.close()


def test_two_contexts_in_one_process():
    """a kernel's MaxDynamicSharedMemorySize is a per-DEVICE attribute; the launchers used to remember "set" in a process-global
    flag, so a host holding one gvl_ctx per device in one process never raised it on the second device.  It is a per-device-ordinal bit now
    (gvl_set_max_lds).  One GPU here: two engines on it, both driving the big-LDS kernels (256^2 ping-pong GEMM: 128 KB of LDS; attention), interleaved,
    must agree with each other -- and on a multi-GPU box the second engine goes to the last device."""
    dev2 = f"cuda:{torch.cuda.device_count() - 1}"
    engs = [E.Engine(tiny_geo(), d, towers=()) for d in (DEV, dev2)]
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    A = (torch.randn((40960, 2048), generator=g) * 0.5).to(bf)                      # 160 x 4 tiles of 256^2 and K >= 1408: the persistent ping-pong kernel (128 KB LDS)
    W = (torch.randn((1024, 2048), generator=g) * 0.02).to(bf)
    q = (torch.randn((200, 3 * 4 * 64), generator=g) * 0.5).to(bf)                 # fused qkv rows of one sequence: 4 heads of 64, S = 200
    outs = []
    for rnd in range(2):
        for eng, d in zip(engs, (DEV, dev2)):
            with torch.cuda.device(d):
                c = eng.op_gemm(A.to(d), W.to(d))
                o = eng.op_attention(q.to(d), 1, 200, 4, 4, 64, 0.125, False)
                torch.cuda.synchronize(d)
                outs.append((c.cpu(), o.cpu()))
    for c, o in outs[1:]:
        assert torch.equal(c, outs[0][0]) and torch.equal(o, outs[0][1])
    for eng in engs:
        eng.close()


def test_model_encode_images_through_the_gvl_exchange_world1(nccl_world1):
    """LLAVA_NEXT_VIDEO(group = the default process group, exchange = "gvl"): encode_images shards the segments (world 1: all of them), builds libgvl's own
    communicator over the group (dist.init_gvl_comm) and all-gathers the token blocks with gvl_allgather_visual -- the product path of the C-ABI exchange.
    Must equal the same model's un-distributed encode bit for bit, and exchange = "torch" likewise."""
    from grounded_video_llm_amd import synth
    from grounded_video_llm_amd.model import LLAVA_NEXT_VIDEO, SyntheticTokenizer
    hid, vocab = 128, 640
    short, long = synth.longrope_factors(32)
    geo = E.TowerGeometry(llm="phi3.5", clip_hidden=64, clip_inter=128, clip_layers=3, clip_heads=4, iv2_dim=64, iv2_inter=128, iv2_depth=3, iv2_heads=4, hidden=hid,
                          inter=256, layers=2, heads=4, kv_heads=4, vocab=vocab, rope_short=short, rope_long=long, rope_theta=10000.0, max_seq=2048, max_segs=6,
                          kv_pages=40, max_prefill=1024)
    sd = {"vision_tower": synth.clip_weights(64, 128, 3, seed="gen.clip"), "video_encoder": synth.iv2_weights(64, 128, 3, 2, seed="gen.iv2"),
          "projectors": synth.projector_weights("phi3.5", hid, 64, 64, seed="gen.proj"), "language_model": synth.llm_weights("phi3", hid, 256, 2, 4, 4, vocab, True, seed="gen.llm")}
    tok = SyntheticTokenizer(vocab, 300)
    samples = {"spatial_pixel_values": synth.det_tensor("gen.sp", (1, 2, 3, 336, 336)).to(DEV), "temporal_pixel_values": synth.det_tensor("gen.tp", (1, 4, 3, 224, 224)).to(DEV)}
    outs = {}
    for exchange in ("gvl", "torch"):
        m = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=64, num_frames=4, num_segs=2, num_temporal_tokens=300, lora=False, llm="phi3.5", geometry=geo, tokenizer=tok,
                             state_dicts=sd, device=DEV, group=torch.distributed.group.WORLD, exchange=exchange)
        outs[exchange] = m.encode_images(samples).clone()
        if exchange == "gvl":
            assert m.engine.comm_count() == 1 and getattr(m.engine, "comm_world", 0) == 1
            m.group = None                                        # the un-distributed path of the same model
            plain = m.engine.encode_segments(samples["spatial_pixel_values"][0], samples["temporal_pixel_values"].reshape(1, 2, 2, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1).contiguous())
            assert torch.equal(outs["gvl"][0], plain)
        m.engine.close()
    assert torch.equal(outs["gvl"], outs["torch"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs: the first N > 1 run of RCCL on real devices happens on the driver's multi-GPU lease")
def test_two_ranks_over_rccl():
    """torchrun-spawned world of 2 on backend nccl (= RCCL over xGMI): tests/nccl_worker.py shards a 3-segment clip 2 + 1, exchanges the token
    blocks through libgvl's own communicator (gvl_allgatherv_visual, straight into the segment-ordered prefix), through torch.distributed, and through
    bench.py's per-step exchange; every result must equal the un-distributed encode bit for bit and ncclCommCount must say 2.  Auto-skipped on a 1-GPU box."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(os.path.dirname(os.path.abspath(__file__)), "nccl_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "NCCL2_OK rank 0 of 2" in r.stdout and "NCCL2_OK rank 1 of 2" in r.stdout, r.stdout[-2000:]
