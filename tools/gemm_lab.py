"""LAB: interleaved A/B of GEMM launcher variants (environment switches re-read per launch) over the hot-path shapes, model epilogues.
usage: gemm_lab.py "name=ENV1=1,ENV2=1;name2=..."   (an empty env list = default)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

SHAPES = [("clip.qkv", 27696, 3072, 1024, "bias"), ("clip.out", 27696, 1024, 1024, "bias_resid32"), ("clip.fc1", 27696, 4096, 1024, "bias_qgelu"), ("clip.fc2", 27696, 1024, 4096, "bias_resid32"),
          ("iv2.qkv", 24588, 4224, 1408, "plain"), ("iv2.proj", 24588, 1408, 1408, "bias_gamma_resid"), ("iv2.fc1", 24588, 6144, 1408, "bias_gelu"), ("iv2.fc2", 24588, 1408, 6144, "bias_gamma_resid"),
          ("phi.qkv", 3519, 9216, 3072, "plain"), ("phi.o", 3519, 3072, 3072, "resid"), ("phi.gu", 3519, 16384, 3072, "silu"), ("phi.down", 3519, 3072, 8192, "resid"),
          ("sq8192", 8192, 8192, 8192, "plain")]
if os.environ.get("GVL_LAB_PHI_M"):       # the decoder GEMMs at other prefill-group sizes: GVL_LAB_PHI_M=1790,3580,7160,14320
    SHAPES = [(f"{n}@{m}", m, N, K, mode) for m in map(int, os.environ["GVL_LAB_PHI_M"].split(",")) for n, _, N, K, mode in SHAPES if n.startswith("phi.")]
VARS = []
for item in (sys.argv[1] if len(sys.argv) > 1 else "base=").split(";"):
    name, envs = item.split("=", 1)
    VARS.append((name, dict(e.split("=") for e in envs.split(",") if e)))
ALL_KEYS = sorted({k for _, d in VARS for k in d})


def main():
    eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
    print("variants:", VARS, flush=True)
    tot = {n: 0.0 for n, _ in VARS}
    for name, M, N, K, mode in SHAPES:
        A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
        W = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
        kw = {}
        if "bias" in mode:
            kw["bias"] = torch.randn((N,), device="cuda")
        if "gamma" in mode:
            kw["gamma"] = torch.randn((N,), device="cuda") * 0.1
        if "resid32" in mode:
            kw["resid"] = torch.randn((M, N), device="cuda"); kw["out_f32"] = True
        elif "resid" in mode:
            kw["resid"] = torch.randn((M, N), device="cuda").to(torch.bfloat16)
        kw["act"] = 1 if "qgelu" in mode else (2 if "gelu" in mode else (3 if "silu" in mode else 0))

        def run(env, n):
            for k in ALL_KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            for _ in range(n):
                eng.op_gemm(A, W, **kw)
        res = {}
        for vn, env in VARS:
            run(env, 2)
        torch.cuda.synchronize()
        for rnd in range(4):
            for vn, env in VARS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(env, 8); e1.record(); torch.cuda.synchronize()
                res.setdefault(vn, []).append(e0.elapsed_time(e1) / 8 * 1e3)
        line = f"{name:9s} ({M},{N},{K}) {mode:17s}"
        for vn, _ in VARS:
            us = sorted(res[vn])[1]                    # second best of 4 rounds
            tot[vn] += us
            line += f"  {vn} {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
        print(line, flush=True)
    print("sum of us:", {k: round(v, 1) for k, v in tot.items()}, flush=True)


if __name__ == "__main__":
    main()
