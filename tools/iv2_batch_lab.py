"""LAB: InternVideo2 tower, 24 segments as 2 x 12 vs 1 x 24 (does a bigger M fill the rounds better than it thrashes the Infinity Cache?)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E, lib as L, synth, weights as Wt
geo = E.TowerGeometry(max_segs=96)
eng = E.Engine(geo, "cuda:0", towers=("iv2",))
W = synth.iv2_weights(geo.iv2_dim, geo.iv2_inter, geo.iv2_depth, geo.frames_per_seg, seed="iv2.one", device="cuda:0")
eng.load_packed(Wt.pack_iv2(W, geo.iv2_depth - 1, geo.frames_per_seg)); del W
eng.finalize()
tp = torch.randn((96, 3, geo.frames_per_seg, 224, 224), device="cuda:0")
def run(chunk, total=96):
    for i in range(0, total, chunk):
        eng.iv2_encode(tp[i:i + chunk])
for chunk in (12, 24, 32, 48, 96, 12, 24, 32, 48, 96):
    run(chunk); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(chunk); run(chunk); e1.record(); torch.cuda.synchronize()
    print(f"[iv2 batch] {chunk:2d} segments per call: {e0.elapsed_time(e1) / 16:.3f} ms per 12 segments", flush=True)
