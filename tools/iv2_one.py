"""Run the InternVideo2 tower alone on 12 segments a few times (target for rocprofv3 --pmc: the in-model attention kernel with the ones-row
row sum, attn_fwd_kernel<96, 4, 2, 1>, and the in-model GEMM epilogues).   python tools/iv2_one.py [iters] [in_place]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E, synth, weights as Wt

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
in_place = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # 0: the round-2 operand path (gvl_debug_set vision_in_place)
geo = E.TowerGeometry(max_segs=12)
eng = E.Engine(geo, "cuda:0", towers=("iv2",))
W = synth.iv2_weights(geo.iv2_dim, geo.iv2_inter, geo.iv2_depth, geo.frames_per_seg, seed="iv2.one", device="cuda:0")
eng.load_packed(Wt.pack_iv2(W, geo.iv2_depth - 1, geo.frames_per_seg)); del W
eng.finalize()
eng.debug_set("vision_in_place", in_place)
tp = torch.randn((12, 3, geo.frames_per_seg, 224, 224), device="cuda:0")
for _ in range(iters):
    out = eng.iv2_encode(tp)
torch.cuda.synchronize()
print("done", tuple(out.shape))
