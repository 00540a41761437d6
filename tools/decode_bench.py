"""Microbenchmark of the decode projections (HBM-bound): skinny MFMA GEMM variants vs the round-1 VALU GEMV, per shape and batch.
  python tools/decode_bench.py            -> table of us / launch and achieved weight-stream GB/s
Variant code = rb*1000 + nw*100 + u*10 + nt (gvl_decode.hip)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
import torch
from grounded_video_llm_amd import engine as E

eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
shapes = [("phi.o", 3072, 3072), ("phi.down", 3072, 8192), ("phi.qkv", 9216, 3072), ("phi.gate_up", 16384, 3072), ("phi.lm_head", 32366, 3072),
          ("llama.down", 4096, 14336), ("floor", 256, 3072)]
variants = [int(v) for v in os.environ.get("VARIANTS", "0,1841,2841").split(",")]
batches = [int(v) for v in os.environ.get("BATCHES", "1,2,4,8,16").split(",")]


def run(N, K, B, mode, variant):
    us = C.c_double(0)
    rounds = max(1, min(16, int(2e9 // (N * K * 2))))          # > 1 GB of distinct weights: the 256 MB Infinity Cache cannot serve the stream
    rc = eng.lib.gvl_op_decode_bench(eng.ctx, N, K, B, mode, variant, rounds, 200, C.byref(us), eng.stream)
    return us.value if rc == 0 else None


for name, N, K in shapes:
    for B in batches:
        row = [f"{name:12s} N={N:6d} K={K:6d} B={B:2d}"]
        if B in (1, 2, 4):
            us = run(N, K, B, 1, 0)
            row.append(f"valu {us:7.2f}us {N * K * 2 / us / 1e3:6.0f}GB/s" if us else "valu   n/a")
        for v in variants:
            us = run(N, K, B, 0, v)
            row.append(f"v{v:04d} {us:7.2f}us {N * K * 2 / us / 1e3:6.0f}GB/s" if us else f"v{v:04d}   n/a")
        if K <= 4096 and B <= 4:           # fused RMSNorm prologue (qkv / gate_up / lm_head): skinny GEMM (LDS) vs VALU GEMV
            for mode, tag in ((2, "norm+mfma"), (3, "norm+valu")):
                us = run(N, K, B, mode, 0)
                row.append(f"{tag} {us:7.2f}us" if us else f"{tag} n/a")
        print(" | ".join(row), flush=True)
