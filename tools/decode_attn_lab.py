"""Per-launch time of the decode-attention kernel and of the decode projections, by group size and context (full-size Phi-3.5 through the
C ABI, eager steps bracketed by the library's per-family hipEvents).   python tools/decode_attn_lab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
import torch
from grounded_video_llm_amd import engine as E, lib as L, synth, weights as Wt

dev = "cuda:0"
geo = E.TowerGeometry(llm="phi3.5", max_seq=4096, max_prefill=3712, kv_pages=0, max_segs=1)
geo.rope_short, geo.rope_long = synth.longrope_factors(96)
eng = E.Engine(geo, dev, towers=("llm",))
W = synth.llm_weights("phi3", geo.hidden, geo.inter, geo.layers, geo.heads, geo.kv_heads, geo.vocab, True, seed="d2e", device=dev)
eng.load_packed(Wt.pack_llm(W, "phi3", geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta, geo.rope_short, geo.rope_long)); del W
torch.cuda.empty_cache()
eng.finalize()
eng.debug_set("decode_graph", 0)
for k in [x for x in os.environ.get("GVL_LAB_SET", "").split(",") if x]:
    kk, vv = k.split("="); eng.debug_set(kk, int(vv))
g = torch.Generator(device=dev); g.manual_seed(1)
new = 9
for S in [int(x) for x in os.environ.get("GVL_E2E_S", "1760,3519").split(",")]:
    emb = (torch.randn((S, geo.hidden), device=dev, generator=g) * 0.5).to(torch.bfloat16)
    for B in [int(x) for x in os.environ.get("GVL_E2E_B", "1,2,4,8,16").split(",")]:
        seqs = [eng.seq_alloc(S + new + 1) for _ in range(B)]
        for s in seqs:
            eng.prefill(s, emb)
        eng.decode_greedy_batch(seqs, 2, None)           # warm
        torch.cuda.synchronize()
        eng.prof_enable(True)
        eng.decode_greedy_batch(seqs, new, None)
        torch.cuda.synchronize()
        da_ms, da_n, _ = eng.prof_read(L.PROF_DECODE_ATTN); gv_ms, gv_n, _ = eng.prof_read(L.PROF_GEMV)
        eng.prof_enable(False)
        for s in seqs:
            eng.seq_free(s)
        kvb = B * 393216 / 32 * (S + 4)                  # K + V bytes of one layer's launch
        print(f"[decode_attn_lab] S={S:5d} B={B:2d}: decode attention {1e3 * da_ms / max(da_n, 1):7.2f} us/launch ({kvb / 1e6:6.1f} MB -> {kvb / (da_ms / max(da_n, 1) * 1e-3) / 1e12:5.2f} TB/s)"
              f"   projections {1e3 * gv_ms / max(gv_n, 1):6.2f} us/launch x {gv_n // max(da_n, 1)} per layer", flush=True)
eng.close()
