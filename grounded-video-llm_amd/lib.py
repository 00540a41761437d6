"""ctypes binding of libgvl.so (the C ABI declared in include/gvl.h).

There is NO fallback: if the shared library is missing or cannot be loaded, importing callers get a
RuntimeError telling them to run `python __graft_entry__.py build` (hipcc, gfx950).  PyTorch is only
used by callers for device memory and streams; this module has no torch dependency.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GVL_LIB_PATH") or os.path.join(_HERE, "libgvl.so")   # override: A/B of two builds on one GPU box (tools/)

F32, BF16, I32, I64 = 0, 1, 2, 3
LLM_PHI3, LLM_LLAMA = 0, 1
ACT_NONE, ACT_QUICK_GELU, ACT_GELU, ACT_SILU_MUL = 0, 1, 2, 3
PROF_GEMM, PROF_ATTN, PROF_GEMV, PROF_DECODE_ATTN, PROF_OTHER = 0, 1, 2, 3, 4


class GvlConfig(C.Structure):
    _fields_ = [
        ("llm_kind", C.c_int32),
        ("clip_hidden", C.c_int32), ("clip_inter", C.c_int32), ("clip_layers_run", C.c_int32), ("clip_heads", C.c_int32),
        ("clip_image", C.c_int32), ("clip_patch", C.c_int32),
        ("iv2_dim", C.c_int32), ("iv2_inter", C.c_int32), ("iv2_blocks_run", C.c_int32), ("iv2_heads", C.c_int32),
        ("iv2_image", C.c_int32), ("iv2_patch", C.c_int32), ("iv2_frames_per_seg", C.c_int32),
        ("hidden", C.c_int32), ("inter", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("kv_heads", C.c_int32),
        ("vocab", C.c_int32),
        ("rms_eps", C.c_float),
        ("lm_head_bias", C.c_int32), ("rope_orig_max_pos", C.c_int32), ("max_seq", C.c_int32),
        ("max_segs", C.c_int32), ("kv_pages", C.c_int32), ("max_prefill", C.c_int32), ("decode_fp8", C.c_int32),
    ]


_SIGS = {
    # name: (restype, argtypes)
    "gvl_create": (C.c_int, [C.POINTER(GvlConfig), C.POINTER(C.c_void_p)]),
    "gvl_destroy": (C.c_int, [C.c_void_p]),
    "gvl_last_error": (C.c_char_p, [C.c_void_p]),
    "gvl_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    "gvl_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "gvl_finalize_weights": (C.c_int, [C.c_void_p]),
    "gvl_load_packed": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "gvl_kv_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "gvl_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gvl_comm_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "gvl_comm_destroy": (C.c_int, [C.c_void_p]),
    "gvl_decode_group_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gvl_comm_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "gvl_allgather_visual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_allgatherv_visual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_clip_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_iv2_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_build_visual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_tokens_per_seg": (C.c_int, [C.c_void_p]),
    "gvl_encode_segments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_splice": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "gvl_seq_alloc": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "gvl_seq_free": (C.c_int, [C.c_void_p, C.c_int]),
    "gvl_seq_fork": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "gvl_seq_clone": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "gvl_prefill_extend": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_prefill": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_decode_greedy": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int), C.c_void_p]),
    "gvl_preprocess_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    "gvl_prefill_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "gvl_decode_steps": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p]),
    "gvl_seq_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "gvl_forward_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p]),
    "gvl_prefill_varlen": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p]),
    "gvl_decode_greedy_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int), C.c_void_p]),
    "gvl_decode_step_logits": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gvl_decode_step_logits_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]),
    "gvl_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gvl_prof_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "gvl_op_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gvl_op_gemm_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gvl_op_fold_gamma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "gvl_op_rowsq_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "gvl_op_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "gvl_op_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "gvl_op_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "gvl_debug_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gvl_set_sampling": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64]),
    "gvl_op_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gvl_op_dgemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gvl_op_decode_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "gvl_probe_mfma": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "gvl_trace_marker": (C.c_int, [C.c_int, C.c_void_p]),
    "gvl_op_gemv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """dlopen libgvl.so and bind every symbol of include/gvl.h.  Raises RuntimeError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU / PyTorch fallback exists). "
            "Build it with `python __graft_entry__.py build` (hipcc --offload-arch=gfx950).")
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)   # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


ERR_ARG, ERR_STATE, ERR_HIP, ERR_OOM, ERR_NOGPU = -1, -2, -3, -4, -5   # include/gvl.h gvl_status


class GvlError(RuntimeError):
    status = 0


def check(lib, ctx, rc, what=""):
    if rc != 0:
        msg = lib.gvl_last_error(ctx)
        err = GvlError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
        err.status = rc
        raise err
