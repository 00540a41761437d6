"""Worker of tests/test_gpu_dist.py::test_two_ranks_over_rccl (launched by torch.distributed.run, one process per GPU, backend nccl = RCCL).

Each rank builds the same tiny model, encodes ITS shard of a 3-segment clip (uneven: 2 + 1), and the token blocks travel
  (a) through libgvl's own communicator behind the C ABI -- gvl_comm_init + gvl_allgatherv_visual: straight into the segment-ordered prefix --
  (b) through torch.distributed.all_gather_into_tensor (padded blocks + re-assembly),
  (c) through bench.py's per-step exchange (`--exchange gvl` and `torch`) on the rotated plan;
every result must equal the same rank's UN-distributed encode of all segments bit for bit, and ncclCommCount must report 2 ranks."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gvl_bootstrap  # noqa: E402,F401


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    torch.distributed.init_process_group("nccl", device_id=torch.device(dev))
    from grounded_video_llm_amd import dist as gdist, engine as E, synth
    from grounded_video_llm_amd.model import LLAVA_NEXT_VIDEO, SyntheticTokenizer
    hid, vocab, n_segs, fps = 128, 640, 3, 2
    short, long = synth.longrope_factors(32)
    geo = E.TowerGeometry(llm="phi3.5", clip_hidden=64, clip_inter=128, clip_layers=3, clip_heads=4, iv2_dim=64, iv2_inter=128, iv2_depth=3, iv2_heads=4, hidden=hid,
                          inter=256, layers=2, heads=4, kv_heads=4, vocab=vocab, rope_short=short, rope_long=long, rope_theta=10000.0, max_seq=2048, max_segs=6,
                          kv_pages=40, max_prefill=1024)
    sd = {"vision_tower": synth.clip_weights(64, 128, 3, seed="gen.clip"), "video_encoder": synth.iv2_weights(64, 128, 3, 2, seed="gen.iv2"),
          "projectors": synth.projector_weights("phi3.5", hid, 64, 64, seed="gen.proj"), "language_model": synth.llm_weights("phi3", hid, 256, 2, 4, 4, vocab, True, seed="gen.llm")}
    tok = SyntheticTokenizer(vocab, 300)
    sp = synth.det_tensor("nccl2.sp", (1, n_segs, 3, 336, 336)).to(dev)
    tp = synth.det_tensor("nccl2.tp", (1, n_segs * fps, 3, 224, 224)).to(dev)
    samples = {"spatial_pixel_values": sp, "temporal_pixel_values": tp}
    outs = {}
    for exchange in ("gvl", "torch"):
        m = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=64, num_frames=n_segs * fps, num_segs=n_segs, num_temporal_tokens=300, lora=False, llm="phi3.5", geometry=geo,
                             tokenizer=tok, state_dicts=sd, device=dev, group=torch.distributed.group.WORLD, exchange=exchange)
        outs[exchange] = m.encode_images(samples).clone()
        if exchange == "gvl":
            assert m.engine.comm_count() == world, f"ncclCommCount says {m.engine.comm_count()} ranks"
            plain = m.engine.encode_segments(sp[0], tp.reshape(1, n_segs, fps, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1).contiguous())
            assert torch.equal(outs["gvl"][0], plain), "sharded encode + gvl_allgatherv_visual differs from the un-distributed encode"
            # bench.py's per-step exchange on the rotated plan, through the same communicator and through torch.distributed
            import bench
            st = bench.Stepper.__new__(bench.Stepper)
            st.world, st.rank, st.dev, st.L, st.eng = world, rank, torch.device(dev), 5, m.engine
            st.mine = gdist.rotated_encode_plan(12, rank, world)
            st.gather = gdist.rotated_gather_index(12, rank, world)

            def block(rnd, clip, seg):
                idx = (rnd * world + clip) * 12 + seg
                return (torch.arange(5, dtype=torch.float32, device=dev)[:, None] + 3.0 * idx).expand(5, 16).to(torch.bfloat16)
            vis_list = [torch.cat([block(rnd, c, u) for c, lo, hi in st.mine for u in range(lo, hi)], 0) for rnd in range(2)]
            for ex in ("gvl", "torch"):
                st.exchange = ex
                got = st._exchange_multi(vis_list)
                torch.cuda.synchronize()
                assert all(torch.equal(got[rnd], torch.cat([block(rnd, rank, u) for u in range(12)], 0)) for rnd in range(2)), f"bench exchange ({ex})"
        m.engine.close()
    assert torch.equal(outs["gvl"], outs["torch"])
    torch.distributed.barrier()
    print(f"NCCL2_OK rank {rank} of {world}", flush=True)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
