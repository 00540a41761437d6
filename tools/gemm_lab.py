"""LAB: operand row-pitch A/B of the GEMM (does a K stride of 8/12/16 KiB hurt the DMA?).  Run on the GPU box.
usage: gemm_lab.py  -> table of TFLOP/s per (shape, pad), interleaved rounds"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

SHAPES = [("iv2.fc2", 24588, 1408, 6144, "bias_gamma_resid"), ("clip.fc2", 27696, 1024, 4096, "bias_resid32"), ("phi.down", 3519, 3072, 8192, "resid"),
          ("sq8192", 8192, 8192, 8192, "plain"), ("iv2.fc1", 24588, 6144, 1408, "bias_gelu"), ("phi.gu", 3519, 16384, 3072, "silu")]
PADS = [0, 64, 192]


def main():
    eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
    lib = eng.lib
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    for name, M, N, K, mode in SHAPES:
        res = {}
        bufs = {}
        for pad in PADS:
            A = torch.randn((M, K + pad), device="cuda").to(torch.bfloat16)
            W = (torch.randn((N, K + pad), device="cuda") * K ** -0.5).to(torch.bfloat16)
            bufs[pad] = (A, W)
        bias = torch.randn((N,), device="cuda") if "bias" in mode else None
        gamma = torch.randn((N,), device="cuda") * 0.1 if "gamma" in mode else None
        out_f32 = "resid32" in mode
        resid = None
        if "resid32" in mode:
            resid = torch.randn((M, N), device="cuda")
        elif "resid" in mode:
            resid = torch.randn((M, N), device="cuda").to(torch.bfloat16)
        act = 1 if "qgelu" in mode else (2 if "gelu" in mode else (3 if "silu" in mode else 0))
        Cc = torch.empty((M, N // 2 if act == 3 else N), device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)

        def run(pad, n):
            A, W = bufs[pad]
            os.environ["GVL_LAB_LD"] = f"{K + pad},{K + pad}"
            for _ in range(n):
                rc = lib.gvl_op_gemm(eng.ctx, p(A), p(W), p(Cc), M, N, K, p(bias), p(gamma), p(resid), act, 1 if out_f32 else 0, 0, eng.stream)
                assert rc == 0, rc
        for pad in PADS:
            print("# pad", pad, file=sys.stderr, flush=True)
            run(pad, 2)
        torch.cuda.synchronize()
        if os.environ.get("GVL_GEMM_TIMING"):
            continue
        for rnd in range(3):
            for pad in PADS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(pad, 10); e1.record(); torch.cuda.synchronize()
                res.setdefault(pad, []).append(round(2.0 * M * N * K / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12, 1))
        print(name, (M, N, K), mode, res, flush=True)
    os.environ.pop("GVL_LAB_LD", None)


if __name__ == "__main__":
    main()
