"""Build libgvl.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# LAB builds (A/B of two kernel variants on one GPU box, loaded through GVL_LIB_PATH): GVL_BUILD_TAG=x GVL_BUILD_DEFS="-DFOO=1" -> libgvl_x.so
TAG = os.environ.get("GVL_BUILD_TAG", "")
OUT = os.path.join(HERE, f"libgvl_{TAG}.so" if TAG else "libgvl.so")
SOURCES = ["gvl_gemm.hip", "gvl_gemm4.hip", "gvl_gemm4p.hip", "gvl_attn.hip", "gvl_elem.hip", "gvl_decode.hip", "gvl_model.hip", "gvl_vision.hip", "gvl_llm.hip", "gvl_host.hip", "gvl_pre.hip",
           "gvl_probe.hip", "gvl_patch.hip"]
HEADERS = ["gvl_internal.h", "gvl_ctx.h", "gvl_model.h", "gvl_gemm_epi.h", "gvl_gemm4_loop.inc", "gvl_gemm4p_loop.inc", os.path.join("..", "..", "include", "gvl.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", "-Wno-cuda-compat",
         # MFMA accumulators in VGPRs (gfx950 has a unified file): no v_accvgpr_read/write copies around the softmax / epilogue VALU
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def source_sha16() -> str:
    """Fingerprint of what decides the GEMM family's memory traffic: every kernel source / header of libgvl plus bench.py (launch shapes).
    tools/pmc_traffic.py stamps it into profiles/rNN_pmc_traffic.json; bench.py reports `roofline.traffic` from that file only while the
    stamp equals the tree it runs from -- a PMC figure of other code is not this run's figure."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(SOURCES + ["gvl_internal.h", "gvl_ctx.h", "gvl_model.h", "gvl_gemm_epi.h", "gvl_gemm4_loop.inc", "gvl_gemm4p_loop.inc"])] + [os.path.join(HERE, "..", "bench.py")]
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build" + ("_" + TAG if TAG else ""))
    flags = FLAGS + os.environ.get("GVL_BUILD_DEFS", "").split()
    if os.environ.get("GVL_BUILD_NO_VGPR_FORM"):          # LAB: MFMA accumulators in AGPRs (hipcc's default) for the listed sources, e.g. "gvl_attn.hip"
        no_vf = os.environ["GVL_BUILD_NO_VGPR_FORM"].split(",")
    else:
        no_vf = []
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            f_ = [x for x in flags if x not in ("-mllvm", "-amdgpu-mfma-vgpr-form")] if s in no_vf else flags
            cmd = [hipcc] + f_ + ["-c", src, "-o", obj]
            if verbose:
                print("[gvl build]", " ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
    if force or procs or _newer(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print("[gvl build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
