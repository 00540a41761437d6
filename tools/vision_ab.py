"""A/B of a result-neutral launch parameter of the vision towers in ONE process, interleaved rounds (default: the attention operand path,
gvl_debug_set("vision_in_place", 1 | 2 | 0)).
   python tools/vision_ab.py [rounds] [key] [values...]   e.g.  tools/vision_ab.py 3 attn_pipe 1 0
   -> wall ms per encode (96 segments = 8 clips: IV2 on 96 x 8 frames, CLIP on 96 frames) and the per-family sums"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E, lib as L, synth, weights as Wt

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
KEY = sys.argv[2] if len(sys.argv) > 2 else "vision_in_place"
VALUES = [int(v) for v in sys.argv[3:]] or [1, 2, 0]
NSEG = 96      # 8 clips of 12 segments: the batch one bench step encodes
geo = E.TowerGeometry(max_segs=NSEG)
eng = E.Engine(geo, "cuda:0", towers=("clip", "iv2"))
W = synth.iv2_weights(geo.iv2_dim, geo.iv2_inter, geo.iv2_depth, geo.frames_per_seg, seed="iv2.one", device="cuda:0")
eng.load_packed(Wt.pack_iv2(W, geo.iv2_depth - 1, geo.frames_per_seg)); del W
W = synth.clip_weights(geo.clip_hidden, geo.clip_inter, geo.clip_layers, geo.clip_image, geo.clip_patch, seed="clip.one", device="cuda:0")
eng.load_packed(Wt.pack_clip(W, geo.clip_layers - 1)); del W
eng.finalize()
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
tp = torch.randn((NSEG, 3, geo.frames_per_seg, 224, 224), device="cuda:0", generator=g)
sp = torch.randn((NSEG, 3, geo.clip_image, geo.clip_image), device="cuda:0", generator=g)
runs = {"iv2": lambda: eng.iv2_encode(tp), "clip": lambda: eng.clip_encode(sp)}
outs = {}
for rnd in range(rounds):
    for mode in VALUES:
        eng.debug_set(KEY, mode)
        for name, fn in runs.items():
            if KEY.startswith("attn_pipe") and name == "clip":
                continue
            out = fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): out = fn()
            e1.record(); torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) / 3
            eng.prof_enable(True); fn(); torch.cuda.synchronize()
            fam = {k: eng.prof_read(c)[0] for k, c in (("gemm", L.PROF_GEMM), ("attn", L.PROF_ATTN), ("other", L.PROF_OTHER))}
            eng.prof_enable(False)
            first = outs.setdefault(name, out)
            same = "yes" if first is out or bool(torch.equal(first, out)) else f"no (max diff {float((first.float() - out.float()).abs().max() / first.float().abs().max()):.2e} of scale)"
            print(f"[vision_ab] round {rnd} {name:4s} {KEY}={mode}: wall {wall:7.3f} ms  gemm {fam['gemm']:7.3f}  attn {fam['attn']:7.3f}  other {fam['other']:6.3f}  bit-identical to first: {same}", flush=True)
