"""Decode-only throughput of the full-size LLM through the C ABI: tokens/s for groups of 1 / 4 / 16 sequences at a short and at the
BASELINE context (3.5 k), and the fraction of the HBM roofline (weights 7.45 GB + KV 384 KB x context per Phi-3.5 token).
  GVL_E2E_GRAPH=0 GVL_E2E_S=64,3519 GVL_E2E_B=1,16 python tools/decode_e2e.py [phi|llama]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
import torch
from grounded_video_llm_amd import engine as E, synth, weights as Wt

which = sys.argv[1] if len(sys.argv) > 1 else "phi"
dev = "cuda:0"
if which == "phi":
    geo = E.TowerGeometry(llm="phi3.5", max_seq=4096, max_prefill=3712, kv_pages=0, max_segs=1)
    geo.rope_short, geo.rope_long = synth.longrope_factors(96)
    kind, wbytes, kv_per_tok = "phi3", 7.45e9, 393216
else:
    geo = E.TowerGeometry.llama3_8b(max_seq=4096, max_prefill=3712, kv_pages=0, max_segs=1)
    kind, wbytes, kv_per_tok = "llama", 15.0e9, 131072
geo.decode_fp8 = int(os.environ.get("GVL_FP8", "0"))           # quantised weight variant (cfg.decode_fp8): 1 = FP8 (half the weight bytes per step), 2 = MXFP4 (a quarter + 1/32 of scales)
if geo.decode_fp8 == 1:
    wbytes /= 2
elif geo.decode_fp8 == 2:
    wbytes *= (0.5 + 1.0 / 32) / 2
eng = E.Engine(geo, dev, towers=("llm",))
W = synth.llm_weights(kind, geo.hidden, geo.inter, geo.layers, geo.heads, geo.kv_heads, geo.vocab, True, seed="d2e", device=dev)
eng.load_packed(Wt.pack_llm(W, kind, geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta, geo.rope_short, geo.rope_long)); del W
torch.cuda.empty_cache()
eng.finalize()
graph = int(os.environ.get("GVL_E2E_GRAPH", "1"))     # 1 (library default): a group's decode step is captured once and replayed
eng.debug_set("decode_graph", graph)
for kv in [x for x in os.environ.get("GVL_LAB_SET", "").split(",") if x]:      # gvl_debug_set KEY=INT pairs (A/B of result-neutral launch parameters)
    k_, v_ = kv.split("="); eng.debug_set(k_, int(v_))
print("kv pool", eng.kv_info(), "graph", graph, "fp8", geo.decode_fp8, flush=True)
g = torch.Generator(device=dev); g.manual_seed(1)
new = 33
SS = [int(x) for x in os.environ.get("GVL_E2E_S", "64,3519").split(",")]
BS = [int(x) for x in os.environ.get("GVL_E2E_B", "1,2,4,8,16").split(",")]
for S in SS:
    emb = (torch.randn((S, geo.hidden), device=dev, generator=g) * 0.5).to(torch.bfloat16)
    for B in BS:
        seqs = [eng.seq_alloc(S + new + 1) for _ in range(B)]
        for s in seqs:
            eng.prefill(s, emb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.decode_greedy_batch(seqs, new, None)
        dt = time.perf_counter() - t0
        for s in seqs:
            eng.seq_free(s)
        steps = new - 1
        byt = steps * (wbytes + B * kv_per_tok * (S + steps / 2))
        print(f"{which} S={S:5d} B={B:2d}: {B * steps / dt:8.1f} tok/s  {1e3 * dt / steps:6.3f} ms/step  {byt / dt / 1e12:5.2f} TB/s = {byt / dt / 8e12:4.2f} of HBM peak", flush=True)
eng.close()
