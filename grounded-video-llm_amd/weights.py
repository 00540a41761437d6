"""Offline weight packer: reference state-dict keys -> packed tensors of libgvl.

Replaces the load_state_dict() calls of inference.py:156-162 / models/llava_next_video.py:117-151.
What it does (all once, at load time):
  * CLIP q/k/v projections fused into one [3C, C] matrix; patch conv flattened to [C, 640]
    (K = 588 zero-padded to the GEMM's 64-element K tile);
  * InternVideo2 temporal pos-embed interpolation (models/internvideo2.py:290-303) and bf16 cast
    (models/llava_next_video.py:134);
  * Phi-3 gate_up rows interleaved (gate_j, up_j) so the SwiGLU product is a GEMM epilogue; Llama
    q/k/v and gate/up fused the same way;
  * LoRA (peft keys `...lora_A.default.weight` / `lora_B`) merged in fp32:  W' = W + (alpha/r) B A
    (models/llava_next_video.py:212-224);
  * RoPE / LongRoPE cos-sin tables computed with the reference's own fp32 formulas and bf16-rounded
    (models/modeling_phi3.py:380-409), one table per factor set (App. C #6).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F


bf = torch.bfloat16


def _pad_k(w2d: torch.Tensor, kp: int) -> torch.Tensor:
    out = torch.zeros((w2d.shape[0], kp), dtype=w2d.dtype, device=w2d.device)
    out[:, : w2d.shape[1]] = w2d
    return out


def interpolate_pos_embed_t(pos: torch.Tensor, orig_t: int, new_t: int, n_extra: int = 1) -> torch.Tensor:
    """Temporal linear interpolation of the InternVideo2 pos_embed (internvideo2.py:290-303)."""
    if orig_t == new_t:
        return pos
    Cd = pos.shape[-1]
    extra, tok = pos[:, :n_extra], pos[:, n_extra:]
    tok = tok.reshape(1, orig_t, -1, Cd).permute(0, 2, 3, 1).reshape(-1, Cd, orig_t)
    tok = F.interpolate(tok.float(), size=new_t, mode="linear")
    tok = tok.reshape(1, -1, Cd, new_t).permute(0, 3, 1, 2).reshape(1, -1, Cd)
    return torch.cat((extra.float(), tok), dim=1)


def pack_clip(W: Dict[str, torch.Tensor], layers_run: int, prefix: str = "vision_model.") -> Dict[str, torch.Tensor]:
    out = {}
    pw = W[prefix + "embeddings.patch_embedding.weight"]
    Cd = pw.shape[0]
    k = pw[0].numel()
    kp = (k + 63) // 64 * 64
    out["clip.patch.w"] = _pad_k(pw.reshape(Cd, k).float(), kp).to(bf)
    out["clip.cls"] = W[prefix + "embeddings.class_embedding"].float().reshape(-1)
    out["clip.pos"] = W[prefix + "embeddings.position_embedding.weight"].float()
    out["clip.preln.w"] = W[prefix + "pre_layrnorm.weight"].float()
    out["clip.preln.b"] = W[prefix + "pre_layrnorm.bias"].float()
    for i in range(layers_run):
        p = f"{prefix}encoder.layers.{i}."
        o = f"clip.L{i}."
        out[o + "qkv.w"] = torch.cat([W[p + f"self_attn.{n}_proj.weight"].float() for n in "qkv"], 0).to(bf)
        out[o + "qkv.b"] = torch.cat([W[p + f"self_attn.{n}_proj.bias"].float() for n in "qkv"], 0).to(bf).float()  # autocast casts the bias
        out[o + "out.w"] = W[p + "self_attn.out_proj.weight"].to(bf)
        out[o + "out.b"] = W[p + "self_attn.out_proj.bias"].to(bf).float()
        out[o + "ln1.w"], out[o + "ln1.b"] = W[p + "layer_norm1.weight"].float(), W[p + "layer_norm1.bias"].float()
        out[o + "ln2.w"], out[o + "ln2.b"] = W[p + "layer_norm2.weight"].float(), W[p + "layer_norm2.bias"].float()
        out[o + "fc1.w"], out[o + "fc1.b"] = W[p + "mlp.fc1.weight"].to(bf), W[p + "mlp.fc1.bias"].to(bf).float()
        out[o + "fc2.w"], out[o + "fc2.b"] = W[p + "mlp.fc2.weight"].to(bf), W[p + "mlp.fc2.bias"].to(bf).float()
    return out


def iv2_ckpt_frames(pos_rows: int, frames: int, tokens_per_frame: Optional[int] = None) -> int:
    """Temporal size of a checkpoint's pos_embed [1, 1 + T*L, C].  With L known it is (rows-1)/L; otherwise the grid is square,
    so T is the candidate -- the target `frames` first, then the reference's orig_t_size=4
    (interpolate_pos_embed_internvideo2_new(..., orig_t_size=4), models/llava_next_video.py:131) -- whose L is a perfect square."""
    n = pos_rows - 1
    if tokens_per_frame:
        if n % tokens_per_frame:
            raise ValueError(f"iv2 pos_embed has {pos_rows} rows: not 1 + T*{tokens_per_frame}")
        return n // tokens_per_frame
    for t in (frames, 4):
        if t > 0 and n % t == 0 and math.isqrt(n // t) ** 2 == n // t:
            return t
    raise ValueError(f"cannot infer the temporal size of an iv2 pos_embed with {pos_rows} rows; pass ckpt_frames / tokens_per_frame")


def pack_iv2(W: Dict[str, torch.Tensor], blocks_run: int, frames: int, ckpt_frames: Optional[int] = None,
             tokens_per_frame: Optional[int] = None) -> Dict[str, torch.Tensor]:
    out = {}
    pw = W["patch_embed.proj.weight"]
    Cd = pw.shape[0]
    k = pw[0].numel()
    kp = (k + 63) // 64 * 64
    out["iv2.patch.w"] = _pad_k(pw.reshape(Cd, k).float(), kp).to(bf)
    out["iv2.patch.b"] = W["patch_embed.proj.bias"].to(bf).float()      # bf16 module: the conv adds a bf16 bias
    out["iv2.cls"] = W["cls_token"].reshape(-1).to(bf)
    pos = W["pos_embed"]
    if ckpt_frames is None:      # the released checkpoint is `-f4`: 4 temporal positions, interpolated to frames_per_seg at load
        ckpt_frames = iv2_ckpt_frames(pos.shape[1], frames, tokens_per_frame)
    if ckpt_frames != frames:
        pos = interpolate_pos_embed_t(pos, ckpt_frames, frames)
    out["iv2.pos"] = pos.reshape(-1, Cd).to(bf)
    for i in range(blocks_run):
        p, o = f"blocks.{i}.", f"iv2.B{i}."
        out[o + "n1.w"], out[o + "n2.w"] = W[p + "norm1.weight"].to(bf), W[p + "norm2.weight"].to(bf)
        out[o + "qkv.w"] = W[p + "attn.qkv.weight"].to(bf)
        out[o + "qn.w"], out[o + "kn.w"] = W[p + "attn.q_norm.weight"].to(bf), W[p + "attn.k_norm.weight"].to(bf)
        out[o + "proj.w"], out[o + "proj.b"] = W[p + "attn.proj.weight"].to(bf), W[p + "attn.proj.bias"].to(bf).float()
        out[o + "ls1"], out[o + "ls2"] = W[p + "ls1.gamma"].to(bf).float(), W[p + "ls2.gamma"].to(bf).float()
        out[o + "fc1.w"], out[o + "fc1.b"] = W[p + "mlp.fc1.weight"].to(bf), W[p + "mlp.fc1.bias"].to(bf).float()
        out[o + "fc2.w"], out[o + "fc2.b"] = W[p + "mlp.fc2.weight"].to(bf), W[p + "mlp.fc2.bias"].to(bf).float()
    return out


def pack_projectors(W: Dict[str, torch.Tensor], llm: str) -> Dict[str, torch.Tensor]:
    out = {}
    a, b = ("linear_0", "linear_1") if llm == "phi3.5" else ("linear_1", "linear_2")
    out["mm.0.w"], out["mm.0.b"] = W[f"multi_modal_projector.{a}.weight"].to(bf), W[f"multi_modal_projector.{a}.bias"].to(bf).float()
    out["mm.1.w"], out["mm.1.b"] = W[f"multi_modal_projector.{b}.weight"].to(bf), W[f"multi_modal_projector.{b}.bias"].to(bf).float()
    out["vp.0.w"], out["vp.0.b"] = W["video_projecter.up_proj.weight"].to(bf), W["video_projecter.up_proj.bias"].to(bf).float()
    out["vp.1.w"], out["vp.1.b"] = W["video_projecter.down_proj.weight"].to(bf), W["video_projecter.down_proj.bias"].to(bf).float()
    if llm == "phi3.5":
        out["sub_gn"] = W["sub_GN"].reshape(-1).float()
        out["glb_gn"] = W["glb_GN"].reshape(-1).to(bf)
    else:
        out["newline"] = W["image_newline"].reshape(-1).to(bf)
    return out


def _strip_peft(W: Dict[str, torch.Tensor], alpha: float, r: int) -> Dict[str, torch.Tensor]:
    """Undo peft's key wrapping and merge LoRA pairs.  `base_model.model.X.weight` -> `X.weight`."""
    if not any("lora_A" in k for k in W):
        return {k[len("base_model.model."):] if k.startswith("base_model.model.") else k: v for k, v in W.items()}
    out = {}
    for k, v in W.items():
        kk = k[len("base_model.model."):] if k.startswith("base_model.model.") else k
        if "lora_" in kk:
            continue
        out[kk.replace(".base_layer", "")] = v
    for k, v in W.items():
        if "lora_A" in k:
            kk = k[len("base_model.model."):] if k.startswith("base_model.model.") else k
            base = kk.split(".lora_A")[0] + ".weight"
            Bm = W[k.replace("lora_A", "lora_B")]
            out[base] = out[base].float() + (alpha / r) * (Bm.float() @ v.float())
    return out


def reset_embeddings(W: Dict[str, torch.Tensor], n_new: int, lm_head_bias: bool = True) -> Dict[str, torch.Tensor]:
    """LLAVA_NEXT_VIDEO.reset_embeddings (models/llava_next_video.py:231-268) on a BASE language-model state dict: the embedding and
    the lm_head grow by n_new rows filled with the mean row; the new lm_head is an nn.Linear with a bias (its random default init is
    replaced by zeros here -- the fine-tuned checkpoint overwrites it).  No-op when the dict already has the grown vocabulary."""
    W = dict(W)
    ek = next(k for k in W if k.endswith("embed_tokens.weight"))
    hk = next(k for k in W if k.endswith("lm_head.weight"))
    for k in (ek, hk):
        w = W[k].float()
        W[k] = torch.cat([w, w.mean(0, keepdim=True).expand(n_new, -1)], 0).to(W[k].dtype)
    bk = hk[: -len("weight")] + "bias"
    if lm_head_bias and bk not in W:
        W[bk] = torch.zeros(W[hk].shape[0], dtype=torch.float32, device=W[hk].device)
    return W


def rope_tables(head_dim: int, max_seq: int, theta: float, factors: Optional[Sequence[float]], max_pos: int, orig_max_pos: int,
                device="cpu"):
    """cos/sin [max_seq, head_dim/2] as float32 holding bf16-rounded values (modeling_phi3.py:380-409 /
    modeling_llama.py:119-133: fp32 math, then `.to(x.dtype)`)."""
    ar = torch.arange(0, head_dim, 2, dtype=torch.int64, device=device).float() / head_dim
    if factors is not None:
        ext = torch.tensor(list(factors), dtype=torch.float32, device=device)
        inv = 1.0 / (ext * theta ** ar)
        scale = max_pos / orig_max_pos
        sf = 1.0 if scale <= 1.0 else math.sqrt(1 + math.log(scale) / math.log(orig_max_pos))
    else:
        inv = 1.0 / (theta ** ar)
        sf = 1.0
    fr = torch.arange(max_seq, device=device).float()[:, None] * inv[None, :]
    return (fr.cos() * sf).to(bf).float().contiguous(), (fr.sin() * sf).to(bf).float().contiguous()


def pack_llm(W: Dict[str, torch.Tensor], kind: str, layers: int, heads: int, kv_heads: int, max_seq: int, rope_theta: float,
             short_factor=None, long_factor=None, max_pos: int = 131072, orig_max_pos: int = 4096,
             lora_alpha: float = 256.0, lora_r: int = 128) -> Dict[str, torch.Tensor]:
    W = _strip_peft(W, lora_alpha, lora_r)
    out = {}
    emb = W["model.embed_tokens.weight"]
    hidden = emb.shape[1]
    d = hidden // heads
    dev = emb.device
    out["llm.embed"] = emb.to(bf)
    out["llm.norm.w"] = W["model.norm.weight"].to(bf)
    out["llm.head.w"] = W["lm_head.weight"].to(bf)
    if "lm_head.bias" in W:
        out["llm.head.b"] = W["lm_head.bias"].to(bf).float()
    for i in range(layers):
        p, o = f"model.layers.{i}.", f"llm.L{i}."
        out[o + "ln1.w"] = W[p + "input_layernorm.weight"].to(bf)
        out[o + "ln2.w"] = W[p + "post_attention_layernorm.weight"].to(bf)
        out[o + "o.w"] = W[p + "self_attn.o_proj.weight"].to(bf)
        out[o + "down.w"] = W[p + "mlp.down_proj.weight"].to(bf)
        if kind == "phi3":
            out[o + "qkv.w"] = W[p + "self_attn.qkv_proj.weight"].to(bf)
            gu = W[p + "mlp.gate_up_proj.weight"]
            g, u = gu[: gu.shape[0] // 2], gu[gu.shape[0] // 2:]
        else:
            out[o + "qkv.w"] = torch.cat([W[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0).to(bf)
            g, u = W[p + "mlp.gate_proj.weight"], W[p + "mlp.up_proj.weight"]
        out[o + "gu.w"] = torch.stack([g, u], dim=1).reshape(2 * g.shape[0], g.shape[1]).to(bf)
    cs, sn = rope_tables(d, max_seq, rope_theta, short_factor, max_pos, orig_max_pos, dev)
    out["rope.cos_s"], out["rope.sin_s"] = cs, sn
    if long_factor is not None:
        cl, sl = rope_tables(d, max_seq, rope_theta, long_factor, max_pos, orig_max_pos, dev)
        out["rope.cos_l"], out["rope.sin_l"] = cl, sl
    return out


# ---- packed file: what `tools/pack_checkpoint.py` writes once and the serving process maps at start ------------------
def save_packed(path: str, packed: Dict[str, torch.Tensor], meta: Optional[Dict[str, str]] = None) -> None:
    """All packed tensors (the names / layouts `gvl_load_weight` expects) in ONE safetensors file: LoRA already merged, q/k/v fused,
    K padded, RoPE tables built -- the per-start cost drops to a file map + one host-to-device copy per tensor."""
    from safetensors.torch import save_file
    save_file({k: v.detach().cpu().contiguous() for k, v in packed.items()}, path, metadata={"format": "gvl-packed-1", **(meta or {})})


def load_packed_file(path: str) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    out = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        if (f.metadata() or {}).get("format") != "gvl-packed-1":
            raise ValueError(f"{path}: not a gvl packed weight file")
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out
