"""InternVideo2 tower alone on 12 segments: wall ms per encode (events, un-profiled) and the per-family sums of a profiled encode
(hipEvent pairs around every launch).  LAB driver: the environment selects kernel variants.   python tools/iv2_prof.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E, lib as L, synth, weights as Wt

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
geo = E.TowerGeometry(max_segs=12)
eng = E.Engine(geo, "cuda:0", towers=("iv2",))
W = synth.iv2_weights(geo.iv2_dim, geo.iv2_inter, geo.iv2_depth, geo.frames_per_seg, seed="iv2.one", device="cuda:0")
eng.load_packed(Wt.pack_iv2(W, geo.iv2_depth - 1, geo.frames_per_seg)); del W
eng.finalize()
tp = torch.randn((12, 3, geo.frames_per_seg, 224, 224), device="cuda:0")
out = eng.iv2_encode(tp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = eng.iv2_encode(tp)
e1.record(); torch.cuda.synchronize()
wall = e0.elapsed_time(e1) / iters
eng.prof_enable(True)
out2 = eng.iv2_encode(tp)
torch.cuda.synchronize()
res = {}
for name, cat in (("gemm", L.PROF_GEMM), ("attn", L.PROF_ATTN), ("other", L.PROF_OTHER)):
    ms, n, work = eng.prof_read(cat)
    res[name] = (round(ms, 3), n, round(work / max(ms, 1e-9) / 1e9, 1))
eng.prof_enable(False)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("GVL_LAB"))
print(f"[iv2_prof] {tag or 'base'}: wall {wall:.3f} ms/encode; profiled: " + "  ".join(f"{k} {v[0]} ms ({v[1]} launches, {v[2]} TFLOP/s)" for k, v in res.items())
      + f"; checksum {float(out.float().abs().mean()):.6f} same {bool(torch.equal(out, out2))}", flush=True)
