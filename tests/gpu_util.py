"""Helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np
import torch

import _gvl_bootstrap  # noqa: F401
from grounded_video_llm_amd import engine as E, weights as Wt, synth

bf = torch.bfloat16
DEV = "cuda:0"


def rel_err(got, ref):
    got = got.detach().float().cpu().double()
    ref = torch.as_tensor(np.asarray(ref)).double() if not isinstance(ref, torch.Tensor) else ref.detach().float().cpu().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), "non-finite values in HIP output"
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-12))


def check(got, ref, tol, what=""):
    e = rel_err(got, ref)
    print(f"[parity] {what}: max-abs err / max-abs ref = {e:.3e} (tol {tol:.1e})")
    assert e <= tol, f"{what}: {e:.3e} > {tol:.1e}"
    return e


def tiny_geo(**kw):
    base = dict(clip_hidden=64, clip_inter=128, clip_layers=3, clip_heads=4, clip_image=28, clip_patch=14,
                iv2_dim=64, iv2_inter=128, iv2_depth=4, iv2_heads=4, iv2_image=28, iv2_patch=14, frames_per_seg=2,
                hidden=64, inter=128, layers=2, heads=4, kv_heads=4, vocab=100, max_seq=512, max_segs=2, kv_pages=16, max_prefill=256)
    base.update(kw)
    return E.TowerGeometry(**base)


def llm_engine(geo, W, towers=("llm",)):
    eng = E.Engine(geo, DEV, towers=towers)
    packed = Wt.pack_llm(W, geo.kind, geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta, geo.rope_short, geo.rope_long,
                         geo.rope_max_pos, geo.rope_orig_max_pos)
    eng.load_packed(packed)
    eng.finalize()
    return eng


def check_bf16_class(got, ref32, ref_emu, floor, what=""):
    """The HIP path must be as close to the fp32 truth as a bf16 implementation of the reference can be:
    err(HIP, fp32) <= max(floor, 2.5 * err(bf16-emulating oracle, fp32))."""
    e_emu = rel_err(ref_emu, ref32)
    tol = max(floor, 2.5 * e_emu)
    return check(got, ref32, tol, what + f" [bf16-emulation itself is {e_emu:.2e} from fp32]")
