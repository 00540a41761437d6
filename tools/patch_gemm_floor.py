"""The patch-GEMM of the two towers (north_star: ">= 50 % MFMA peak on the ViT patch-GEMM") and what bounds it: hipEvent times of the GEMM alone on pre-patchified
bf16 operands (K = 588 padded to 640: ten k-tiles per output tile), under whatever libgvl build GVL_LIB_PATH names.  Run under the shipped library and under
LAB builds that remove ONE activity from the ping-pong main loop (wrong results on purpose; build: GVL_BUILD_TAG=x GVL_BUILD_DEFS="-DGVL_LAB -DGVL_PP_ENERGY_LAB=n"):
  n = 2  no global->LDS DMA after a tile's first k-tile      n = 1  fragment reads in phase 0 only (a quarter of the ds_read_b128)
and with the plain / bias epilogue, plus the long-K control (same M, N with K = 4096) that shows the rate the SAME kernel reaches when the main loop dominates.
Prints TFLOP/s on the padded K actually executed and on the algorithmic K = 588, and the fraction of the 2.5 PF dense bf16 peak."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
tag = os.path.basename(os.environ.get("GVL_LIB_PATH", "libgvl.so"))
for name, M, N, K, Kalg, mode in (("iv2.patch", 196608, 1408, 640, 588, "bias"), ("iv2.patch", 196608, 1408, 640, 588, "plain"), ("clip.patch", 55296, 1024, 640, 588, "plain"),
                                   ("iv2.patch K=4096 control", 196608, 1408, 4096, 4096, "bias"), ("clip.patch K=4096 control", 55296, 1024, 4096, 4096, "plain")):
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    W = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
    kw = {"bias": torch.randn((N,), device="cuda")} if mode == "bias" else {}
    for cfg in (82, 21):                                  # 82: the persistent 256x256 ping-pong kernel (one block per CU); 21: 128x128, two blocks per CU (what the launcher picks at K < 1024)
        for _ in range(3):
            eng.op_gemm(A, W, tile_cfg=cfg, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            eng.op_gemm(A, W, tile_cfg=cfg, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        tf_pad, tf_alg = 2.0 * M * N * K / us / 1e6, 2.0 * M * N * Kalg / us / 1e6
        hbm = (M * K * 2 + M * N * 2 + N * K * 2) / us / 1e6      # TB/s of algorithmic operand + output bytes
        print(f"{tag:<18} cfg {cfg} {name:<26} M={M:>6} N={N:>5} K={K:>5} {mode:<5} {us:8.1f} us  {tf_pad:7.1f} TF/s executed = {tf_pad / 2500:.3f} of peak | {tf_alg:7.1f} TF/s algorithmic = {tf_alg / 2500:.3f} | "
              f"operand+output bytes at {hbm:.2f} TB/s", flush=True)
