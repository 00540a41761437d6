"""GEMM micro-benchmark over the hot-path shapes (run on the GPU box): TFLOP/s per tile configuration."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

SHAPES = [  # (name, M, N, K)
    ("clip.patch", 6912, 1024, 640), ("iv2.patch", 24576, 1408, 640),      # patch-embedding GEMMs (K = 588 padded to 640)
    ("clip.qkv", 6924, 3072, 1024), ("clip.out", 6924, 1024, 1024), ("clip.fc1", 6924, 4096, 1024), ("clip.fc2", 6924, 1024, 4096),
    ("iv2.qkv", 24588, 4224, 1408), ("iv2.proj", 24588, 1408, 1408), ("iv2.fc1", 24588, 6144, 1408), ("iv2.fc2", 24588, 1408, 6144),
    ("phi.qkv", 3519, 9216, 3072), ("phi.o", 3519, 3072, 3072), ("phi.gu", 3519, 16384, 3072), ("phi.down", 3519, 3072, 8192),
    ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192),
]
# the fused epilogue each shape runs with in the model (GVL_BENCH_EPI=model); default: plain bf16 store
EPI = {"iv2.patch": "bias", "clip.qkv": "bias", "clip.out": "bias_resid32", "clip.fc1": "bias_qgelu", "clip.fc2": "bias_resid32",
       "iv2.qkv": "plain", "iv2.proj": "bias_gamma_resid", "iv2.fc1": "bias_gelu", "iv2.fc2": "bias_gamma_resid",
       "phi.qkv": "plain", "phi.o": "resid", "phi.gu": "silu", "phi.down": "resid"}
CFGS = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,21,82".split(","))]


def main():
    eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
    res = {}
    for name, M, N, K in SHAPES:
        A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        W = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
        row = {}
        kw = {}
        mode = EPI.get(name, "plain") if os.environ.get("GVL_BENCH_EPI") == "model" else "plain"
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
        for cfg in CFGS:
            for _ in range(2):
                eng.op_gemm(A, W, tile_cfg=cfg, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                eng.op_gemm(A, W, tile_cfg=cfg, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            row[cfg] = round(2.0 * M * N * K / (ms * 1e-3) / 1e12, 1)
        res[name] = row
        print(name, (M, N, K), mode, row, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
