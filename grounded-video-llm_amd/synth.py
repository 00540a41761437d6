"""Deterministic synthetic tensors (weights / pixels) that do not depend on torch's RNG streams.

There are no real checkpoints on the build or GPU boxes, so parity tests, smoke() and the bench
run on seeded synthetic weights at the reference's shapes.  `det_tensor` is a
counter-based generator (numpy PCG64 raw stream -> uniform), stable across torch/numpy versions,
so fixtures only need to store seeds + outputs.
"""
from __future__ import annotations

import hashlib

import numpy as np
import torch


def _seed_of(name: str) -> int:
    return int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], "little")


def det_numpy(name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Uniform tensor with the given std/mean, fully determined by `name` and `shape`."""
    n = int(np.prod(shape)) if len(shape) else 1
    raw = np.random.PCG64(_seed_of(name)).random_raw(n)
    u = (raw >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)  # [0,1)
    x = (u - 0.5) * (2.0 * np.sqrt(3.0) * std) + mean
    return x.astype(np.float32).reshape(shape)


def det_tensor(name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    return torch.from_numpy(det_numpy(name, tuple(shape), std, mean))


_M32 = 0xFFFFFFFF


def _fmix32(x: torch.Tensor) -> torch.Tensor:
    """murmur3 finaliser on 32-bit values held in int64 lanes (integer ops only: bit-identical on CPU and on the GPU)."""
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & _M32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & _M32
    return x ^ (x >> 16)


def exact_tensor(name: str, shape, std: float = 1.0, mean: float = 0.0, device="cpu", chunk: int = 1 << 26) -> torch.Tensor:
    """Counter-based uniform tensor that is BIT-IDENTICAL on every device: element i = f(sha256(name), i) through two
    murmur3 finalisers in integer arithmetic, 24 random bits -> f32 exactly, then two separately rounded IEEE f32
    operations (scale, shift).  Used for the full-size goldens: the reference runs on CPU weights in the build container,
    the HIP path regenerates the same weights on the GPU (no multi-GB fixture, no host->device copy)."""
    dev = torch.device(device)
    n = 1
    for d in shape:
        n *= int(d)
    assert n < (1 << 32), "exact_tensor: one tensor holds fewer than 2^32 elements"
    s = _seed_of(name)
    s0, s1 = s & _M32, (s >> 32) & _M32
    scale = float(np.float32(2.0 * np.sqrt(3.0) * std))
    shift = float(np.float32(mean))
    out = torch.empty(n, dtype=torch.float32, device=dev)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        i = torch.arange(lo, hi, dtype=torch.int64, device=dev)
        h = _fmix32(i ^ s0)
        h = _fmix32((h + s1) & _M32)
        u = (h >> 8).to(torch.float32) * (1.0 / 16777216.0) - 0.5          # exact: 24-bit fraction, |u| < 0.5
        u.mul_(scale)
        if shift != 0.0:
            u.add_(shift)
        out[lo:hi] = u
    return out.reshape(tuple(int(d) for d in shape))


class _Gen:
    """name -> tensor factory.
    exact=True : exact_tensor on `device` (bit-identical on CPU and GPU; full-size parity goldens);
    device='cpu': det_tensor (numpy PCG64 stream; the small round-1 goldens);
    device='cuda': torch.randn on the GPU (fast full-size bench weights; not used for parity)."""

    def __init__(self, seed: str, device="cpu", exact=False):
        self.seed, self.device, self.exact = seed, torch.device(device), exact
        if self.device.type != "cpu" and not exact:
            self.g = torch.Generator(device=self.device)
            self.g.manual_seed(_seed_of(seed) % (2 ** 31))

    def __call__(self, name, shape, std=0.02, mean=0.0):
        if self.exact:
            return exact_tensor(self.seed + "/" + name, shape, std, mean, self.device)
        if self.device.type == "cpu":
            return det_tensor(self.seed + "/" + name, shape, std, mean)
        t = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        t.normal_(mean, std, generator=self.g)
        return t


def clip_weights(hidden=1024, inter=4096, layers=24, image=336, patch=14, seed="clip", device="cpu", prefix="vision_model.", exact=False):
    """State-dict keys of CLIPVisionModel (models/modeling_clip.py; SURVEY §8b weight contract)."""
    g = _Gen(seed, device, exact)
    P = (image // patch) ** 2 + 1
    W = {
        prefix + "embeddings.class_embedding": g("cls", (hidden,), 0.5),
        prefix + "embeddings.patch_embedding.weight": g("patch", (hidden, 3, patch, patch), 0.03),
        prefix + "embeddings.position_embedding.weight": g("pos", (P, hidden), 0.3),
        prefix + "pre_layrnorm.weight": g("preln.w", (hidden,), 0.1, 1.0),
        prefix + "pre_layrnorm.bias": g("preln.b", (hidden,), 0.05),
        prefix + "post_layernorm.weight": g("postln.w", (hidden,), 0.1, 1.0),
        prefix + "post_layernorm.bias": g("postln.b", (hidden,), 0.05),
    }
    sw = hidden ** -0.5
    for i in range(layers):
        p = f"{prefix}encoder.layers.{i}."
        for n in ("q", "k", "v", "out"):
            W[p + f"self_attn.{n}_proj.weight"] = g(f"{i}.{n}.w", (hidden, hidden), sw)
            W[p + f"self_attn.{n}_proj.bias"] = g(f"{i}.{n}.b", (hidden,), 0.05)
        for n in ("layer_norm1", "layer_norm2"):
            W[p + n + ".weight"] = g(f"{i}.{n}.w", (hidden,), 0.1, 1.0)
            W[p + n + ".bias"] = g(f"{i}.{n}.b", (hidden,), 0.05)
        W[p + "mlp.fc1.weight"] = g(f"{i}.fc1.w", (inter, hidden), sw)
        W[p + "mlp.fc1.bias"] = g(f"{i}.fc1.b", (inter,), 0.05)
        W[p + "mlp.fc2.weight"] = g(f"{i}.fc2.w", (hidden, inter), inter ** -0.5)
        W[p + "mlp.fc2.bias"] = g(f"{i}.fc2.b", (hidden,), 0.05)
    return W


def iv2_weights(dim=1408, inter=6144, depth=40, frames=8, image=224, patch=14, seed="iv2", device="cpu", exact=False):
    """State-dict keys of PretrainInternVideo2 that the hot path reads (models/internvideo2.py:766-1040).
    `depth` = number of block weight sets generated (the forward uses depth-1 of them)."""
    g = _Gen(seed, device, exact)
    L = (image // patch) ** 2
    W = {
        "cls_token": g("cls", (1, 1, dim), 0.5),
        "pos_embed": g("pos", (1, 1 + frames * L, dim), 0.3),
        "patch_embed.proj.weight": g("patch.w", (dim, 3, 1, patch, patch), 0.03),
        "patch_embed.proj.bias": g("patch.b", (dim,), 0.05),
    }
    sw = dim ** -0.5
    for i in range(depth):
        p = f"blocks.{i}."
        W[p + "norm1.weight"] = g(f"{i}.n1", (dim,), 0.1, 1.0)
        W[p + "norm2.weight"] = g(f"{i}.n2", (dim,), 0.1, 1.0)
        W[p + "attn.qkv.weight"] = g(f"{i}.qkv", (3 * dim, dim), sw)
        W[p + "attn.q_norm.weight"] = g(f"{i}.qn", (dim,), 0.1, 1.0)
        W[p + "attn.k_norm.weight"] = g(f"{i}.kn", (dim,), 0.1, 1.0)
        W[p + "attn.proj.weight"] = g(f"{i}.proj.w", (dim, dim), sw)
        W[p + "attn.proj.bias"] = g(f"{i}.proj.b", (dim,), 0.05)
        W[p + "ls1.gamma"] = g(f"{i}.ls1", (dim,), 0.02, 0.1)
        W[p + "ls2.gamma"] = g(f"{i}.ls2", (dim,), 0.02, 0.1)
        W[p + "mlp.fc1.weight"] = g(f"{i}.fc1.w", (inter, dim), sw)
        W[p + "mlp.fc1.bias"] = g(f"{i}.fc1.b", (inter,), 0.05)
        W[p + "mlp.fc2.weight"] = g(f"{i}.fc2.w", (dim, inter), inter ** -0.5)
        W[p + "mlp.fc2.bias"] = g(f"{i}.fc2.b", (dim,), 0.05)
    return W


def projector_weights(llm="phi3.5", llm_hidden=3072, clip_hidden=1024, iv2_dim=1408, seed="proj", device="cpu", exact=False):
    """multi_modal_projector / video_projecter / newline tensors (models/llava_next_video.py:26-54,122-145)."""
    g = _Gen(seed, device, exact)
    W = {}
    if llm == "phi3.5":
        cin = 4 * clip_hidden
        W["multi_modal_projector.linear_0.weight"] = g("mm0.w", (llm_hidden, cin), cin ** -0.5)
        W["multi_modal_projector.linear_0.bias"] = g("mm0.b", (llm_hidden,), 0.05)
        W["multi_modal_projector.linear_1.weight"] = g("mm1.w", (llm_hidden, llm_hidden), llm_hidden ** -0.5)
        W["multi_modal_projector.linear_1.bias"] = g("mm1.b", (llm_hidden,), 0.05)
        W["glb_GN"] = g("glb", (1, 1, cin), 0.5)
        W["sub_GN"] = g("sub", (1, 1, 1, cin), 0.5)
    else:
        W["multi_modal_projector.linear_1.weight"] = g("mm1.w", (llm_hidden, clip_hidden), clip_hidden ** -0.5)
        W["multi_modal_projector.linear_1.bias"] = g("mm1.b", (llm_hidden,), 0.05)
        W["multi_modal_projector.linear_2.weight"] = g("mm2.w", (llm_hidden, llm_hidden), llm_hidden ** -0.5)
        W["multi_modal_projector.linear_2.bias"] = g("mm2.b", (llm_hidden,), 0.05)
        W["image_newline"] = g("nl", (llm_hidden,), 0.5)
    W["video_projecter.up_proj.weight"] = g("vp0.w", (llm_hidden, iv2_dim), iv2_dim ** -0.5)
    W["video_projecter.up_proj.bias"] = g("vp0.b", (llm_hidden,), 0.05)
    W["video_projecter.down_proj.weight"] = g("vp1.w", (llm_hidden, llm_hidden), llm_hidden ** -0.5)
    W["video_projecter.down_proj.bias"] = g("vp1.b", (llm_hidden,), 0.05)
    return W


def llm_weight_specs(kind="phi3", hidden=3072, inter=8192, layers=32, heads=32, kv_heads=32, vocab=32366, lm_head_bias=True):
    """(state-dict key, generator name, shape, std, mean) of every tensor of Phi3ForCausalLM / LlamaForCausalLM, in the
    order llm_weights() generates them; lm_head has a bias after reset_embeddings (models/llava_next_video.py:263).  A caller that
    cannot hold a second copy of an 8 B-parameter model (the C3 golden generator) walks the list one tensor at a time."""
    d = hidden // heads
    sw = hidden ** -0.5
    specs = [("model.embed_tokens.weight", "embed", (vocab, hidden), 0.5, 0.0),
             ("model.norm.weight", "norm", (hidden,), 0.1, 1.0),
             ("lm_head.weight", "head.w", (vocab, hidden), sw, 0.0)]
    if lm_head_bias:
        specs.append(("lm_head.bias", "head.b", (vocab,), 0.05, 0.0))
    for i in range(layers):
        p = f"model.layers.{i}."
        specs.append((p + "input_layernorm.weight", f"{i}.ln1", (hidden,), 0.1, 1.0))
        specs.append((p + "post_attention_layernorm.weight", f"{i}.ln2", (hidden,), 0.1, 1.0))
        specs.append((p + "self_attn.o_proj.weight", f"{i}.o", (hidden, heads * d), sw, 0.0))
        specs.append((p + "mlp.down_proj.weight", f"{i}.down", (hidden, inter), inter ** -0.5, 0.0))
        if kind == "phi3":
            specs.append((p + "self_attn.qkv_proj.weight", f"{i}.qkv", ((heads + 2 * kv_heads) * d, hidden), sw, 0.0))
            specs.append((p + "mlp.gate_up_proj.weight", f"{i}.gu", (2 * inter, hidden), sw, 0.0))
        else:
            specs.append((p + "self_attn.q_proj.weight", f"{i}.q", (heads * d, hidden), sw, 0.0))
            specs.append((p + "self_attn.k_proj.weight", f"{i}.k", (kv_heads * d, hidden), sw, 0.0))
            specs.append((p + "self_attn.v_proj.weight", f"{i}.v", (kv_heads * d, hidden), sw, 0.0))
            specs.append((p + "mlp.gate_proj.weight", f"{i}.gate", (inter, hidden), sw, 0.0))
            specs.append((p + "mlp.up_proj.weight", f"{i}.up", (inter, hidden), sw, 0.0))
    return specs


def llm_weights(kind="phi3", hidden=3072, inter=8192, layers=32, heads=32, kv_heads=32, vocab=32366,
                lm_head_bias=True, seed="llm", device="cpu", exact=False):
    """State-dict keys of Phi3ForCausalLM / LlamaForCausalLM, all tensors materialised (see llm_weight_specs; its order
    is the generation order -- it matters for the sequential torch RNG stream of device='cuda', exact=False)."""
    g = _Gen(seed, device, exact)
    return {k: g(n, shp, std, mean) for k, n, shp, std, mean in llm_weight_specs(kind, hidden, inter, layers, heads, kv_heads, vocab, lm_head_bias)}


def longrope_factors(head_dim=96):
    """Placeholder LongRoPE factor vectors (the real ones live in HF config.json, absent here)."""
    n = head_dim // 2
    short = [1.0 + 0.2 * i / max(n - 1, 1) for i in range(n)]
    long = [1.0 + 63.0 * i / max(n - 1, 1) for i in range(n)]
    return short, long


LORA_TARGETS = {"phi3": ("self_attn.qkv_proj", "self_attn.o_proj", "mlp.gate_up_proj", "mlp.down_proj"),
                "llama": ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.up_proj", "mlp.down_proj",
                          "mlp.gate_proj")}       # models/llava_next_video.py:219-222


def lora_wrap(W, kind="phi3", r=128, seed="lora", device="cpu", exact=False, std=0.02):
    """A plain LLM state dict -> the key layout of the reference's fine-tuned checkpoints: peft==0.3.0 wraps the model, so every key
    gains the `base_model.model.` prefix and each target projection gains `lora_A.default.weight` [r, in] / `lora_B.default.weight`
    [out, r] (models/llava_next_video.py:212-229; SURVEY §8b weight contract [ext])."""
    g = _Gen(seed, device, exact)
    out = {"base_model.model." + k: v for k, v in W.items()}
    for k, v in W.items():
        if k.endswith(".weight") and any(k.endswith(t + ".weight") for t in LORA_TARGETS[kind]):
            base = "base_model.model." + k[: -len(".weight")]
            out[base + ".lora_A.default.weight"] = g(k + ".A", (r, v.shape[1]), std)
            out[base + ".lora_B.default.weight"] = g(k + ".B", (v.shape[0], r), std)
    return out
