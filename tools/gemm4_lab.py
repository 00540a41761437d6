"""LAB: interleaved A/B of GEMM tile configurations (82 = 8-wave ping-pong, 84 / 86 / 87 = 4-wave AGPR kernel, loop schedule 0 / 1 / 2) with the
epilogues the MODEL runs (fused RMSNorm forms included), several rounds per configuration in ONE process, medians.
  python tools/gemm4_lab.py 82,86 [model|epi|sq] [rounds]
  model: the bench's GEMM shapes (one clip's M);  epi: every epilogue at K = 192 and K = 1408 (tile time difference = what the epilogue costs);  sq: squares"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

CFGS = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "82,86").split(",")]
WHAT = sys.argv[2] if len(sys.argv) > 2 else "model"
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 5

MSCALE = int(os.environ.get("GVL_LAB_MSCALE", "1"))       # 8: the bench's M (8 clips per step)
MODEL = [("clip.qkv", 27696, 3072, 1024, "bias"), ("clip.fc1", 27696, 4096, 1024, "bias_qgelu"),
         ("iv2.qkv", 24588, 4224, 1408, "rs"), ("iv2.proj", 24588, 1408, 1408, "bias_gamma_resid_sq"), ("iv2.fc1", 24588, 6144, 1408, "rs_bias_gelu"),
         ("iv2.fc2", 24588, 1408, 6144, "bias_gamma_resid_sq"),
         ("phi.qkv", 14076, 9216, 3072, "rs"), ("phi.o", 14076, 3072, 3072, "resid_sq"), ("phi.gu", 14076, 16384, 3072, "rs_silu"), ("phi.down", 14076, 3072, 8192, "resid_sq")]
EPIS = ["plain", "rs", "bias", "bias_qgelu", "rs_bias_gelu", "rs_silu", "resid_sq", "bias_gamma_resid_sq"]
if WHAT == "model":
    SHAPES = [(n, m * MSCALE if n.startswith("iv2") else m, N, K, mode) for n, m, N, K, mode in MODEL]
elif WHAT == "p":          # the shapes the pipelined kernel serves so far
    SHAPES = [("clip.qkv", 27696, 3072, 1024, "bias"), ("iv2.qkv", 24588, 4224, 1408, "rs"), ("phi.qkv", 14076, 9216, 3072, "rs"), ("sq8192", 8192, 8192, 8192, "plain")]
elif WHAT == "p4":
    SHAPES = [("iv2.qkv", 24588, 4224, 1408, "rs"), ("iv2.fc1", 24588, 6144, 1408, "rs_bias_gelu"), ("phi.gu", 14076, 16384, 3072, "rs_silu"), ("clip.qkv", 27696, 3072, 1024, "bias")]
elif WHAT == "p5":
    SHAPES = [("iv2.fc1", 24588, 6144, 1408, "rs_bias_gelu"), ("iv2.fc2", 24588, 1408, 6144, "bias_gamma_resid_sq"), ("phi.o", 14076, 3072, 3072, "resid_sq"),
              ("phi.gu", 14076, 16384, 3072, "rs_silu"), ("phi.down", 14076, 3072, 8192, "resid_sq")]
elif WHAT == "p3":
    SHAPES = [("iv2.proj", 24588, 1408, 1408, "bias_gamma_resid_sq"), ("iv2.fc2", 24588, 1408, 6144, "bias_gamma_resid_sq"), ("phi.o", 14076, 3072, 3072, "resid_sq"), ("phi.down", 14076, 3072, 8192, "resid_sq")]
elif WHAT == "p2":
    SHAPES = [("clip.qkv", 27696, 3072, 1024, "bias"), ("iv2.qkv", 24588, 4224, 1408, "rs")]
elif WHAT == "epi":
    SHAPES = [(f"{e}@K{k}", 24576, 2048, k, e) for e in EPIS for k in (192, 1408)]
else:
    SHAPES = [("sq4096", 4096, 4096, 4096, "plain"), ("sq8192", 8192, 8192, 8192, "plain")]


def main():
    eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
    # "cfg" entries >= 1000 mean: automatic configuration (tile_cfg 0) under gvl_debug_set("gemm_band", cfg - 1000) -- per-shape A/B of the rasterisation band
    tot = {c: 0.0 for c in CFGS}
    for name, M, N, K, mode in SHAPES:
        A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        W = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
        parts = mode.split("_")
        kw = {}
        if "bias" in parts:
            kw["bias"] = torch.randn((N,), device="cuda")
        if "gamma" in parts:
            kw["gamma"] = torch.randn((N,), device="cuda") * 0.1
        if "resid" in parts:
            kw["resid"] = torch.randn((M, N), device="cuda").to(torch.bfloat16)
        kw["act"] = 1 if "qgelu" in parts else (2 if "gelu" in parts else (3 if "silu" in parts else 0))
        rows = "rs" in parts or "sq" in parts
        if "rs" in parts:
            kw["rowscale"] = torch.rand((M,), device="cuda") + 0.5
        if "sq" in parts:
            kw["want_rowsq"] = True

        def run(cfg, n):
            # cfg 2000 / 2001 / 2002: automatic configuration under gvl_debug_set("gemm_narrow", 0 / 1 / 2) -- A/B of the narrow column tiles
            if cfg >= 2000:
                eng.debug_set("gemm_narrow", cfg - 2000)
            elif cfg >= 1000:
                eng.debug_set("gemm_band", cfg - 1000)
            for _ in range(n):
                (eng.op_gemm_rows if rows else eng.op_gemm)(A, W, tile_cfg=0 if cfg >= 1000 else cfg, **kw)
            if cfg >= 2000:
                eng.debug_set("gemm_narrow", 1)
            elif cfg >= 1000:
                eng.debug_set("gemm_band", 0)
        for c in CFGS:
            run(c, 2)
        torch.cuda.synchronize()
        res = {c: [] for c in CFGS}
        n = 6
        for rnd in range(ROUNDS):
            for c in (CFGS if rnd % 2 == 0 else CFGS[::-1]):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(c, n); e1.record(); torch.cuda.synchronize()
                res[c].append(e0.elapsed_time(e1) / n * 1e3)
        line = f"{name:26s} ({M},{N},{K})"
        for c in CFGS:
            us = sorted(res[c])[len(res[c]) // 2]
            tot[c] += us
            line += f"  cfg{c} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
        if len(CFGS) > 1:
            b = sorted(res[CFGS[0]])[len(res[CFGS[0]]) // 2]
            line += "  | vs first: " + " ".join(f"{b / sorted(res[c])[len(res[c]) // 2]:.3f}" for c in CFGS[1:])
        print(line, flush=True)
    print("sum of us:", {k: round(v, 1) for k, v in tot.items()}, flush=True)


if __name__ == "__main__":
    main()
