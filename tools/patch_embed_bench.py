"""Whole-op time of the patch embedding at the sizes the bench runs it (CLIP: 96 key frames -> M = 55 296 patch rows; InternVideo2: 96 segments x 8
frames -> M = 196 608), fused kernel (gvl_patch.hip) vs the three-pass path (gvl_debug_set patch_fused = 1 | 0): the towers are run with ZERO layers so
that the encode call is the embedding plus the CLS strip.   python tools/patch_embed_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E, lib as L, synth, weights as Wt

NSEG = 96
geo = E.TowerGeometry(max_segs=NSEG)
geo.clip_layers, geo.iv2_depth = 1, 1            # layers run = layers - 1 = 0 for both towers
eng = E.Engine(geo, "cuda:0", towers=("clip", "iv2"))
W = synth.iv2_weights(geo.iv2_dim, geo.iv2_inter, 1, geo.frames_per_seg, seed="pe.iv2", device="cuda:0")
eng.load_packed(Wt.pack_iv2(W, 0, geo.frames_per_seg)); del W
W = synth.clip_weights(geo.clip_hidden, geo.clip_inter, 1, geo.clip_image, geo.clip_patch, seed="pe.clip", device="cuda:0")
eng.load_packed(Wt.pack_clip(W, 0)); del W
eng.finalize()
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
tp = torch.randn((NSEG, 3, geo.frames_per_seg, 224, 224), device="cuda:0", generator=g)
sp = torch.randn((NSEG, 3, geo.clip_image, geo.clip_image), device="cuda:0", generator=g)
base_other = {}
flops = {"iv2": 2.0 * NSEG * geo.frames_per_seg * 256 * 1408 * 588, "clip": 2.0 * NSEG * 576 * 1024 * 588}
for name, fn in (("iv2", lambda: eng.iv2_encode(tp)), ("clip", lambda: eng.clip_encode(sp))):
    for mode in (1, 0, 1, 0):
        eng.debug_set("patch_fused", mode)
        fn(); torch.cuda.synchronize()
        eng.prof_enable(True); fn(); torch.cuda.synchronize()
        fam = {k: eng.prof_read(c) for k, c in (("gemm", L.PROF_GEMM), ("other", L.PROF_OTHER))}
        eng.prof_enable(False)
        base = base_other.setdefault(name, fam["other"][0]) if mode else base_other.get(name, 0.0)     # fused: `other` = the CLS strip alone
        whole = fam["gemm"][0] + (fam["other"][0] - base_other.get(name, fam["other"][0]) if not mode else 0.0)
        print(f"[patch_embed] {name:4s} patch_fused={mode}: gemm-family {fam['gemm'][0]*1e3:8.1f} us in {fam['gemm'][1]} launches, other {fam['other'][0]*1e3:8.1f} us in {fam['other'][1]} launches"
              f"  -> whole op {whole*1e3:7.1f} us = {flops[name]/(whole*1e-3)/1e12:.0f} TFLOP/s ({flops[name]/1e9:.1f} GFLOP of convolution)", flush=True)
