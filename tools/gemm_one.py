"""Run ONE GEMM shape/config a few times (target for rocprofv3 --pmc)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

M, N, K, cfg = (int(x) for x in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
W = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
for _ in range(iters):
    eng.op_gemm(A, W, tile_cfg=cfg)
torch.cuda.synchronize()
print("done", M, N, K, cfg)
