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
mode = sys.argv[6] if len(sys.argv) > 6 else "plain"      # e.g. bias_gelu, bias_gamma_resid, resid, silu, bias_resid32; rs_... / ..._sq = the fused-RMSNorm forms
kw = {}
parts = mode.split("_")
rows = "rs" in parts or "sq" in parts
if "rs" in parts:
    kw["rowscale"] = torch.rand((M,), device="cuda") + 0.5
if "sq" in parts:
    kw["want_rowsq"] = True
if "bias" in mode:
    kw["bias"] = torch.randn((N,), device="cuda")
if "gamma" in mode:
    kw["gamma"] = torch.randn((N,), device="cuda") * 0.1
if "resid32" in mode:
    kw["resid"] = torch.randn((M, N), device="cuda"); kw["out_f32"] = True
elif "resid" in mode:
    kw["resid"] = torch.randn((M, N), device="cuda").to(torch.bfloat16)
if "qgelu" in mode:
    kw["act"] = 1
elif "gelu" in mode:
    kw["act"] = 2
elif "silu" in mode:
    kw["act"] = 3
for _ in range(iters):
    (eng.op_gemm_rows if rows else eng.op_gemm)(A, W, tile_cfg=cfg, **kw)
torch.cuda.synchronize()
print("done", M, N, K, cfg)
