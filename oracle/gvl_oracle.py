"""CPU ORACLE for the Grounded-VideoLLM inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-on-CPU (fp32) restatement of the reference algorithm.  It is the
*checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``grounded-video-llm_amd/``) never imports, calls or falls back to anything in here.

Parity status: PINNED against outputs of the reference's own modules run in this container
(``oracle/make_golden.py`` imports /root/reference through ``oracle/ref_shims.py`` and writes
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).  Third-party
arithmetic that is not under /root/reference (transformers GenerationMixin greedy loop,
DynamicCache, peft LoRA, flash-attn) is restated from its published semantics: "parity
unpinned" for the LoRA merge (peft 0.3.0 is absent here) and for the HF generate() loop
beyond what the full-sequence forward pins.

Every function cites the reference file:line it follows (paths relative to /root/reference).

Numerics: ``emu=False`` is exact fp32 (what the reference computes on a CPU host, SURVEY
App. A last paragraph).  ``emu=True`` inserts bf16 roundings at the points where the
reference *GPU* path (autocast bf16 / bf16 weights) rounds, following SURVEY Appendix A; the
HIP path is compared against both.
"""
from __future__ import annotations

import math
import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IMAGE_TOKEN_INDEX = -200          # datasets/chat/base_template.py:14
IGNORE_INDEX = -100               # datasets/chat/base_template.py:13
DEFAULT_IMAGE_TOKEN = "<image>"   # datasets/chat/base_template.py:15
GROUNDING_TOKEN = "<timestamp_grounding>"  # datasets/chat/base_template.py:16

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # mm_utils/utils.py:147
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)   # mm_utils/utils.py:148
INTERNVIDEO_MEAN = (0.485, 0.456, 0.406)                    # mm_utils/utils.py:150
INTERNVIDEO_STD = (0.229, 0.224, 0.225)                     # mm_utils/utils.py:151


# --------------------------------------------------------------------------------------
# rounding helper
# --------------------------------------------------------------------------------------
def _r(x: torch.Tensor, emu: bool) -> torch.Tensor:
    """Round to bf16 and come back to fp32 when emulating the reference GPU numerics."""
    return x.to(torch.bfloat16).to(torch.float32) if emu else x


def _lin(x, w, b, emu):
    """nn.Linear under autocast: operands cast to bf16, fp32 accumulate, bf16 result."""
    y = F.linear(_r(x, emu), _r(w, emu), None if b is None else _r(b, emu))
    return _r(y, emu)


# --------------------------------------------------------------------------------------
# a1 / a14 -- integer paths
# --------------------------------------------------------------------------------------
def get_frame_indices(num_frames: int, vlen: int, sample: str = "middle") -> List[int]:
    """mm_utils/video_utils.py:13-38 ('middle' branch: midpoint of each linspace interval)."""
    if sample != "middle":
        raise NotImplementedError("only the inference path's 'middle' sampling is on the hot path")
    acc_samples = min(num_frames, vlen)
    intervals = np.linspace(start=0, stop=vlen, num=acc_samples + 1).astype(int)
    ranges = [(int(intervals[i]), int(intervals[i + 1]) - 1) for i in range(len(intervals) - 1)]
    frame_indices = [(a + b) // 2 for a, b in ranges]
    if len(frame_indices) < num_frames:  # padded with last frame, :33-36
        padded = [frame_indices[-1]] * num_frames
        padded[: len(frame_indices)] = frame_indices
        frame_indices = padded
    return frame_indices


def spatial_frame_indices(num_frames: int, num_segs: int) -> List[int]:
    """inference.py:82-83."""
    per = int(num_frames // num_segs)
    return [(i * per) + int(per / 2) for i in range(num_segs)]


def seconds_to_temporal_tokens(query: str, duration: float, num_temporal_tokens: int = 300) -> str:
    """inference.py:107 -- '(\\d+) seconds' -> '<k>' with k = int(float(sec)/duration*N)."""
    return re.sub(r"(\d+) seconds",
                  lambda m: f"<{int(float(m.group(1)) / duration * num_temporal_tokens)}>", query)


def quantize_timestamp(time: float, duration: float, num_temporal_tokens: int = 300) -> int:
    """datasets/mix_grounded.py:84-85 (training-side order of operations, clipped to N)."""
    return min(int(num_temporal_tokens * time / duration), num_temporal_tokens)


def parse_time_interval(text: str, duration: float, num_temporal_tokens: int = 300, llm: str = "phi3.5") -> str:
    """inference.py:125-134 -- '<k>' -> seconds string (Phi keeps the leading space)."""
    def rep(m):
        x = int(m.group(1))
        t = duration * x / num_temporal_tokens
        if llm == "phi3.5":
            return f" {t:.2f} seconds"
        elif llm == "llama3":
            return f"{t:.2f} seconds"
        return None  # reference returns None for other llm values -> re.sub raises; keep it loud
    return re.sub(r"<(\d+)>", rep, text)


# prompt templates -- datasets/chat/base_template.py:86-134
_SYSTEM = {
    "phi3.5": "<|system|>\nYou are a helpful AI assistant that can generate responses based on visual inputs.",
    "llama3": "<|start_header_id|>system<|end_header_id|>You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, and assist the user with a variety of tasks using natural language.",
    "vicuna": "You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, and assist the user with a variety of tasks using natural language.",
}
_USER = {"phi3.5": "\n<|user|>\n", "llama3": "<|start_header_id|>user<|end_header_id|>", "vicuna": "\nUSER: "}
_ASSIST = {"phi3.5": ("\n<|assistant|>\n", "<|endoftext|>"),
           "llama3": ("<|start_header_id|>assistant<|end_header_id|>", "<|eot_id|>"),
           "vicuna": ("\nASSISTANT: ", "</s>")}


def template_encode(llm: str, conv: Sequence[Dict[str, str]]) -> str:
    """Template.encode -> _prompt, datasets/chat/base_template.py:49-112."""
    qs, ans = [], []
    first_is_not_question = 0
    for i, m in enumerate(conv):
        if i == 0 and m["from"] != "human":
            first_is_not_question = 1
            continue
        (qs if i % 2 == first_is_not_question else ans).append(m["value"])
    assert len(qs) == len(ans)
    msg = ""
    for i, (q, a) in enumerate(zip(qs, ans)):
        if i == 0:
            msg += _SYSTEM[llm]
        if DEFAULT_IMAGE_TOKEN in q and GROUNDING_TOKEN not in q:
            q = q.replace(DEFAULT_IMAGE_TOKEN, "").strip()
            q = (DEFAULT_IMAGE_TOKEN + "\n" + q).strip()
        msg += _USER[llm] + q
        msg += _ASSIST[llm][0] + a + _ASSIST[llm][1]
    return msg


def build_prompt(llm: str, mode: str, text: str, duration: float = 0.0, num_temporal_tokens: int = 300) -> str:
    """inference.py:93-113 -- the three prompt modes; trailing eos removed."""
    if mode == "grounding":
        value = DEFAULT_IMAGE_TOKEN + " " + GROUNDING_TOKEN + "\n" + text
    elif mode == "qa":
        value = DEFAULT_IMAGE_TOKEN + "\n" + text
    elif mode == "referring":
        value = DEFAULT_IMAGE_TOKEN + "\n" + seconds_to_temporal_tokens(text, duration, num_temporal_tokens)
    else:
        raise AssertionError(mode)
    conv = [{"from": "human", "value": value}, {"from": "gpt", "value": ""}]
    return template_encode(llm, conv).replace(_ASSIST[llm][1], "")


def tokenizer_image_token(prompt: str, tokenize: Callable[[str], List[int]], bos_token_id: Optional[int],
                          image_token_index: int = IMAGE_TOKEN_INDEX) -> List[int]:
    """models/llava_next_video.py:409-426."""
    chunks = [tokenize(c) for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    ids: List[int] = []
    offset = 0
    if len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    inter = [e for sub in zip(chunks, [sep] * len(chunks)) for e in sub][:-1]
    for x in inter:
        ids.extend(x[offset:])
    return ids


def left_pad_truncate(batch_ids: Sequence[Sequence[int]], pad_id: int, max_txt_len: int):
    """models/llava_next_video.py:626-647 -- flip / pad_sequence / truncate / flip back."""
    L = max(len(x) for x in batch_ids)
    ids = torch.full((len(batch_ids), L), pad_id, dtype=torch.long)
    mask = torch.zeros((len(batch_ids), L), dtype=torch.long)
    for i, x in enumerate(batch_ids):
        ids[i, L - len(x):] = torch.tensor(list(x), dtype=torch.long)
        mask[i, L - len(x):] = 1
    if L > max_txt_len:      # keeps the LAST max_txt_len tokens (truncation happens on the flipped tensor)
        ids, mask = ids[:, L - max_txt_len:], mask[:, L - max_txt_len:]
    return ids, mask


# --------------------------------------------------------------------------------------
# f4 -- training forward: label masking (integers) and the causal-LM loss
# --------------------------------------------------------------------------------------
def make_labels(llm: str, input_ids: Sequence[int], prompt: str, tokenize: Callable[[str], List[int]], bos_token_id: Optional[int]) -> torch.Tensor:
    """models/llava_next_video.py:325-407 (make_labels + _make_masks_{llama3,vicuna,phi3}): everything but the assistant
    answers (and their eos) is IGNORE_INDEX.  Tensor slicing semantics (negative lengths give empty slices) are the reference's."""
    labels = torch.tensor(list(input_ids), dtype=torch.long).clone()
    sep, eos_token = _ASSIST[llm]
    rounds = prompt.split(eos_token)
    cur_len = 1                      # bos
    labels[:cur_len] = IGNORE_INDEX
    for i, rou in enumerate(rounds):
        if rou == "":
            break
        parts = rou.split(sep)
        if len(parts) != 2:
            break
        parts[0] += sep
        round_len = len(tokenizer_image_token(rou, tokenize, bos_token_id)) + 1 - 1          # + eos - bos
        if llm == "llama3":
            instruction_len = len(tokenizer_image_token(parts[0], tokenize, bos_token_id)) - 1
        elif llm == "vicuna":
            instruction_len = len(tokenizer_image_token(parts[0], tokenize, bos_token_id)) - 1 - 1
            if i >= 1:
                instruction_len -= 1
                round_len -= 1
        elif llm == "phi3.5":
            instruction_len = len(tokenizer_image_token(parts[0], tokenize, bos_token_id)) - 1 - 1
            if i >= 1:
                instruction_len += 1
                round_len += 1
        else:
            raise ValueError(llm)
        labels[cur_len: cur_len + instruction_len] = IGNORE_INDEX
        cur_len += round_len
    labels[cur_len:] = IGNORE_INDEX
    return labels


def prepare_batch(llm: str, texts: Sequence[str], tokenize, bos_token_id, pad_token_id: int, eos_token_id: int, max_txt_len: int):
    """models/llava_next_video.py:428-452: per-text ids / labels / ones-mask, RIGHT pad (pad id / -100 / 0), truncate to
    max_txt_len columns and overwrite the last label column with eos (for EVERY row, also the padded ones)."""
    ids = [tokenizer_image_token(t, tokenize, bos_token_id) for t in texts]
    labs = [make_labels(llm, x, t, tokenize, bos_token_id) for x, t in zip(ids, texts)]
    W = max(len(x) for x in ids)
    bi = torch.full((len(ids), W), pad_token_id, dtype=torch.long)
    bl = torch.full((len(ids), W), IGNORE_INDEX, dtype=torch.long)
    bm = torch.zeros((len(ids), W), dtype=torch.long)
    for r, (x, l) in enumerate(zip(ids, labs)):
        bi[r, :len(x)] = torch.tensor(x, dtype=torch.long)
        bl[r, :len(x)] = l
        bm[r, :len(x)] = 1
    if W > max_txt_len:
        bi, bl, bm = bi[:, :max_txt_len], bl[:, :max_txt_len].clone(), bm[:, :max_txt_len]
        bl[:, -1] = eos_token_id
    return bi, bl, bm


def splice_labels(input_ids: torch.Tensor, labels: torch.Tensor, mask: torch.Tensor, n_visual: int, is_text: bool):
    """Label / mask half of prepare_multimodal_inputs (models/llava_next_video.py:568-596): visual rows are IGNORE_INDEX;
    a 'text' sample gets its (dummy) visual rows appended at the END with mask 0."""
    k = int(torch.where(input_ids == IMAGE_TOKEN_INDEX)[0])
    ign = torch.full((n_visual,), IGNORE_INDEX, dtype=torch.long)
    if is_text:
        return torch.cat([labels[:k], labels[k + 1:], ign]), torch.cat([mask[:k], mask[k + 1:], torch.zeros(n_visual, dtype=torch.long)])
    return torch.cat([labels[:k], ign, labels[k + 1:]]), torch.cat([mask[:k], torch.ones(n_visual, dtype=torch.long), mask[k + 1:]])


def causal_lm_loss_terms(logits: torch.Tensor, labels: torch.Tensor):
    """Phi3ForCausalLM.forward labels branch (models/modeling_phi3.py:1527-1539): logits.float(); logits[:-1] vs labels[1:];
    CrossEntropyLoss(ignore_index=-100).  Returns (sum of token nll, number of labelled tokens); loss = sum / count."""
    lg = logits.float()[:-1]
    y = labels[1:]
    keep = y != IGNORE_INDEX
    if int(keep.sum()) == 0:
        return 0.0, 0
    lp = torch.log_softmax(lg[keep], dim=-1)
    nll = -lp[torch.arange(lp.shape[0]), y[keep]]
    return float(nll.double().sum()), int(keep.sum())


# --------------------------------------------------------------------------------------
# a2 / a3 -- CLIP ViT-L/14-336 (models/modeling_clip.py)
# --------------------------------------------------------------------------------------
def clip_embeddings(px: torch.Tensor, W: Dict[str, torch.Tensor], emu=False, prefix="vision_model.") -> torch.Tensor:
    """CLIPVisionEmbeddings.forward :182-191 + pre_layrnorm :851.  px [N,3,H,W] -> [N,1+P,C] fp32."""
    w = W[prefix + "embeddings.patch_embedding.weight"]
    ps = w.shape[-1]
    pe = _r(F.conv2d(_r(px, emu), _r(w, emu), None, stride=ps), emu)         # conv is autocast->bf16
    pe = pe.flatten(2).transpose(1, 2)
    cls = W[prefix + "embeddings.class_embedding"].expand(px.shape[0], 1, -1)
    x = torch.cat([cls, pe], dim=1) + W[prefix + "embeddings.position_embedding.weight"][None]
    C = x.shape[-1]
    return F.layer_norm(x, (C,), W[prefix + "pre_layrnorm.weight"], W[prefix + "pre_layrnorm.bias"], 1e-5)


def clip_layer(x: torch.Tensor, W: Dict[str, torch.Tensor], i: int, num_heads: int, emu=False,
               prefix="vision_model.", eps=1e-5, round_scores=False) -> torch.Tensor:
    """CLIPEncoderLayer.forward :355-393, CLIPAttention.forward :252-328, CLIPMLP :339-343."""
    p = f"{prefix}encoder.layers.{i}."
    N, S, C = x.shape
    hd = C // num_heads
    h = F.layer_norm(x, (C,), W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], eps)
    q = _lin(h, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"], emu) * (hd ** -0.5)
    k = _lin(h, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"], emu)
    v = _lin(h, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"], emu)
    q = _r(q, emu)
    sh = lambda t: t.view(N, S, num_heads, hd).transpose(1, 2)
    q, k, v = sh(q), sh(k), sh(v)
    s = q @ k.transpose(-1, -2)
    if round_scores:                 # eager bmm under autocast returns bf16 scores (:274)
        s = _r(s, emu)
    pr = torch.softmax(s.float(), dim=-1)
    o = _r(_r(pr, emu) @ v, emu)                                            # bmm(attn_probs, v) in bf16 :314
    o = o.transpose(1, 2).reshape(N, S, C)
    o = _lin(o, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"], emu)
    x = x + o
    h = F.layer_norm(x, (C,), W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], eps)
    h = _lin(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"], emu)
    h = _r(h * torch.sigmoid(1.702 * h), emu)                               # quick_gelu (transformers ACT2FN)
    h = _lin(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"], emu)
    return x + h


def clip_penultimate(px, W, num_layers: int, num_heads: int, emu=False, prefix="vision_model.") -> torch.Tensor:
    """vision_tower(...).hidden_states[-2][:, 1:]  (models/llava_next_video.py:504-505).

    hidden_states has num_layers+1 entries (modeling_clip.py:626-651); [-2] is the output of layer
    num_layers-1, so only num_layers-1 layers are needed."""
    x = clip_embeddings(px, W, emu, prefix)
    for i in range(num_layers - 1):
        x = clip_layer(x, W, i, num_heads, emu, prefix)
    return x[:, 1:]


# --------------------------------------------------------------------------------------
# a6 / a7 -- InternVideo2 (models/internvideo2.py)
# --------------------------------------------------------------------------------------
def interpolate_pos_embed_t(pos: torch.Tensor, orig_t: int, new_t: int, n_extra: int = 1) -> torch.Tensor:
    """interpolate_pos_embed_internvideo2_new :290-303 (temporal, linear).  pos [1, n_extra+T*HW, C]."""
    if orig_t == new_t:
        return pos
    C = pos.shape[-1]
    extra, tok = pos[:, :n_extra], pos[:, n_extra:]
    tok = tok.view(1, orig_t, -1, C).permute(0, 2, 3, 1).reshape(-1, C, orig_t)
    tok = F.interpolate(tok, size=new_t, mode="linear")
    tok = tok.view(1, -1, C, new_t).permute(0, 3, 1, 2).reshape(1, -1, C)
    return torch.cat((extra, tok), dim=1)


def _rmsnorm(x, w, eps, emu):
    """RMSNorm.forward internvideo2.py:443-448 == Phi3RMSNorm modeling_phi3.py:319-324."""
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    xn = xf * torch.rsqrt(var + eps)
    return _r(_r(w, emu) * _r(xn, emu), emu)


def rmsnorm_linear_fused(x, norm_w, w, b, eps):
    """The HIP build's FUSED form of `Linear(RMSNorm(x))` (round 5; csrc/gvl_gemm.hip GemmArgs.rowsq / rowscale, csrc/gvl_elem.hip fold_gamma / rowsq_finish),
    restated with its rounding points so that the CPU suite can bound it against the reference-order form `_lin(_rmsnorm(x, ...), ...)`
    (internvideo2.py:443-448 + :587,631; modeling_phi3.py:319-324 + :459,659):
        W'   = bf16(W * gamma)                                  one rounding per weight (fold_gamma, at gvl_finalize_weights)
        rs_m = rsqrt(sum_blocks(sum_{64 cols} bf16(x)^2) / C + eps)   fp32; the blocks are added in index order (rowsq_finish)
        y    = bf16(rs_m * (bf16(x) . W'^T) [+ b])              fp32 accumulate, the row scale on the accumulator, ONE rounding of the output
    The reference rounds x * rs and then gamma * (x * rs) to bf16 before the GEMM; here those two activation roundings are gone."""
    xb = x.to(torch.bfloat16).float()
    C = xb.shape[-1]
    assert C % 64 == 0
    wf = (w.to(torch.bfloat16).float() * norm_w.to(torch.bfloat16).float()).to(torch.bfloat16).float()
    blocks = xb.pow(2).reshape(*xb.shape[:-1], C // 64, 64).sum(-1)
    tot = torch.zeros_like(blocks[..., 0])
    for j in range(C // 64):
        tot = tot + blocks[..., j]
    rs = torch.rsqrt(tot / C + eps)
    y = rs[..., None] * F.linear(xb, wf)
    if b is not None:
        y = y + b.float()
    return y.to(torch.bfloat16).float()


def iv2_embed(px: torch.Tensor, W, emu=False) -> torch.Tensor:
    """PatchEmbed.forward :721-725 + cls/pos :972-1011.  px [B,3,T,H,W] -> [B,1+T*L,C]."""
    w, b = W["patch_embed.proj.weight"], W["patch_embed.proj.bias"]
    ps = w.shape[-1]
    x = F.conv3d(_r(px, emu), _r(w, emu), _r(b, emu), stride=(w.shape[2], ps, ps))
    x = _r(x, emu).flatten(3).permute(0, 2, 3, 1)          # B T HW C
    B, T, L, C = x.shape
    x = x.reshape(B, T * L, C)
    x = torch.cat((_r(W["cls_token"], emu).expand(B, -1, -1), x), dim=1)
    return _r(x + _r(W["pos_embed"], emu), emu)


def iv2_block(x, W, i: int, num_heads: int, emu=False, eps=1e-6) -> torch.Tensor:
    """Block._inner_forward :680-684; Attention._naive_attn :564-583 (== _flash_attn :585-605)."""
    p = f"blocks.{i}."
    B, S, C = x.shape
    hd = C // num_heads
    h = _rmsnorm(x, W[p + "norm1.weight"], eps, emu)
    qkv = _lin(h, W[p + "attn.qkv.weight"], None, emu)
    q, k, v = qkv.reshape(B, S, 3, C).unbind(2)
    q = _rmsnorm(q, W[p + "attn.q_norm.weight"], eps, emu)   # over the FULL width, all heads jointly :572
    k = _rmsnorm(k, W[p + "attn.k_norm.weight"], eps, emu)
    sh = lambda t: t.view(B, S, num_heads, hd).transpose(1, 2)
    q, k, v = sh(q), sh(k), sh(v)
    s = (q * (hd ** -0.5)) @ k.transpose(-1, -2)
    o = _r(torch.softmax(s, dim=-1) @ v, emu)
    o = o.transpose(1, 2).reshape(B, S, C)
    o = _lin(o, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"], emu)
    o = _r(o.float() * _r(W[p + "ls1.gamma"].float(), emu), emu)      # LayerScale force_fp32 :458-463
    x = _r(x + o, emu)
    h = _rmsnorm(x, W[p + "norm2.weight"], eps, emu)
    h = _lin(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"], emu)
    h = _r(F.gelu(h), emu)                                   # nn.GELU (erf) :616
    h = _lin(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"], emu)
    h = _r(h.float() * _r(W[p + "ls2.gamma"].float(), emu), emu)
    return _r(x + h, emu)


def iv2_encode(px, W, depth: int, num_heads: int, emu=False, x_vis_return_idx: int = -2) -> torch.Tensor:
    """PretrainInternVideo2.forward(x, None, False, x_vis_return_idx=-2, x_vis_only=True)[:, 1:, :]
    :970-1040 -- runs blocks 0..depth-2 (break at idx == depth + x_vis_return_idx :1028)."""
    x = iv2_embed(px, W, emu)
    for i in range(depth):
        x = iv2_block(x, W, i, num_heads, emu)
        if i == depth + x_vis_return_idx:
            break
    return x[:, 1:, :]


# --------------------------------------------------------------------------------------
# a4 / a5 / a8 / a9 -- glue + projectors (models/llava_next_video.py)
# --------------------------------------------------------------------------------------
def hd_merge_2x2_phi3(f: torch.Tensor) -> torch.Tensor:
    """reshape_hd_patches_2x2merge_phi3(f, 1, 1) :454-476.  [N,576,1024] -> [N,12,12,4096]."""
    N, L, C = f.shape
    H = int(L ** 0.5)
    return (f.reshape(N, H // 2, 2, H // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, H // 2, H // 2, 4 * C))


def add_image_newline_phi3(fhd: torch.Tensor, sub_GN: torch.Tensor) -> torch.Tensor:
    """add_image_newline_phi3 :478-489.  [N,h,w,4096] -> [N,h*(w+1),4096]."""
    N, h, w, D = fhd.shape
    nl = sub_GN.reshape(1, 1, 1, D).expand(N, h, 1, D)
    return torch.cat([fhd, nl], dim=2).reshape(N, -1, D)


def pool_spatial_llama(f: torch.Tensor, out_hw: int = 8) -> torch.Tensor:
    """AdaptiveAvgPool3d([S, 8, 8]) on [bs,C,S,24,24] :509-517 == block mean.  [N,576,C] -> [N,64,C]."""
    N, L, C = f.shape
    H = int(math.sqrt(L))
    k = H // out_hw
    return f.reshape(N, out_hw, k, out_hw, k, C).mean(dim=(2, 4)).reshape(N, out_hw * out_hw, C)


def pool_temporal(seg: torch.Tensor, T: int, pool: int = 4, emu=False) -> torch.Tensor:
    """AdaptiveAvgPool3d([T,4,4]) on [B,1408,T,16,16] :543-549.  [B,T*256,C] -> [B,T*16,C]."""
    B, TL, C = seg.shape
    H = int(math.sqrt(TL // T))
    k = H // pool
    y = seg.reshape(B, T, pool, k, pool, k, C).mean(dim=(3, 5)).reshape(B, T * pool * pool, C)
    return _r(y, emu)


def mlp_projector(x, w0, b0, w1, b1, emu=False):
    """Phi3_5_Projecter :50-54 / Video_Projecter :35-39 / LlavaMultiModalProjector: Linear-GELU(erf)-Linear."""
    h = _r(F.gelu(_lin(x, w0, b0, emu)), emu)
    return _lin(h, w1, b1, emu)


def encode_images(spatial_px, temporal_px, Wclip, Wiv2, Wproj, llm: str, *, clip_layers=24, clip_heads=16,
                  iv2_depth=40, iv2_heads=16, emu=False) -> torch.Tensor:
    """LLAVA_NEXT_VIDEO.encode_images :491-566.

    spatial_px [bs,S,3,336,336], temporal_px [bs,F,3,224,224] -> [bs, S*(img+seg+1), D].
    Wproj keys: 'multi_modal_projector.*', 'video_projecter.*', and 'glb_GN','sub_GN' (phi3.5) or
    'image_newline' (llama3)."""
    bs, S = spatial_px.shape[:2]
    F_ = temporal_px.shape[1]
    fps = F_ // S
    img = clip_penultimate(spatial_px.flatten(0, 1), Wclip, clip_layers, clip_heads, emu)   # [bs*S,576,1024]
    if llm == "phi3.5":
        img = add_image_newline_phi3(hd_merge_2x2_phi3(img), Wproj["sub_GN"])
        img = mlp_projector(img, Wproj["multi_modal_projector.linear_0.weight"], Wproj["multi_modal_projector.linear_0.bias"],
                            Wproj["multi_modal_projector.linear_1.weight"], Wproj["multi_modal_projector.linear_1.bias"], emu)
    else:
        img = pool_spatial_llama(img)
        img = mlp_projector(img, Wproj["multi_modal_projector.linear_1.weight"], Wproj["multi_modal_projector.linear_1.bias"],
                            Wproj["multi_modal_projector.linear_2.weight"], Wproj["multi_modal_projector.linear_2.bias"], emu)
    img = img.reshape(bs, S, img.shape[1], img.shape[2])
    t = temporal_px.reshape(bs, S, fps, *temporal_px.shape[2:]).permute(0, 1, 3, 2, 4, 5).flatten(0, 1)  # (bs S) c f h w
    seg = iv2_encode(t, Wiv2, iv2_depth, iv2_heads, emu)                                                # [(bs S), f*256, 1408]
    seg = pool_temporal(seg, fps, 4, emu)
    seg = mlp_projector(seg, Wproj["video_projecter.up_proj.weight"], Wproj["video_projecter.up_proj.bias"],
                        Wproj["video_projecter.down_proj.weight"], Wproj["video_projecter.down_proj.bias"], emu)
    seg = seg.reshape(bs, S, seg.shape[1], seg.shape[2])
    D = seg.shape[-1]
    if llm == "phi3.5":
        nl = Wproj["glb_GN"].reshape(-1)[None, :]
        nl = mlp_projector(nl, Wproj["multi_modal_projector.linear_0.weight"], Wproj["multi_modal_projector.linear_0.bias"],
                           Wproj["multi_modal_projector.linear_1.weight"], Wproj["multi_modal_projector.linear_1.bias"], emu)
    else:
        nl = _r(Wproj["image_newline"].reshape(1, -1), emu)
    nl = nl.reshape(1, 1, 1, D).expand(bs, S, 1, D)
    return torch.cat([img, seg, nl], dim=2).reshape(bs, -1, D)


def splice(input_ids: torch.Tensor, visual: torch.Tensor, embed_w: torch.Tensor, emu=False) -> torch.Tensor:
    """prepare_multimodal_inputs (non-'text' branch) :579-590 for ONE sample: embed(pre) | visual | embed(post)."""
    idx = int(torch.where(input_ids == IMAGE_TOKEN_INDEX)[0][0])
    e = _r(embed_w, emu)
    return torch.cat([e[input_ids[:idx]], _r(visual, emu), e[input_ids[idx + 1:]]], dim=0)


# --------------------------------------------------------------------------------------
# a11 / a12 / a13 -- decoder LLMs (models/modeling_phi3.py, models/modeling_llama.py)
# --------------------------------------------------------------------------------------
class LLMConfig:
    def __init__(self, kind: str, hidden: int, inter: int, layers: int, heads: int, kv_heads: int, vocab: int,
                 rms_eps: float = 1e-5, rope_theta: float = 10000.0, max_pos: int = 131072, orig_max_pos: int = 4096,
                 short_factor: Optional[Sequence[float]] = None, long_factor: Optional[Sequence[float]] = None):
        self.kind, self.hidden, self.inter, self.layers = kind, hidden, inter, layers
        self.heads, self.kv_heads, self.vocab = heads, kv_heads, vocab
        self.rms_eps, self.rope_theta, self.max_pos, self.orig_max_pos = rms_eps, rope_theta, max_pos, orig_max_pos
        self.short_factor, self.long_factor = short_factor, long_factor
        self.head_dim = hidden // heads


def rope_cos_sin(cfg: LLMConfig, positions: torch.Tensor, kv_seq_len: int, emu=False):
    """Phi3LongRoPEScaledRotaryEmbedding.forward modeling_phi3.py:380-409 (short/long chosen by
    kv_seq_len > original_max_position_embeddings) / Phi3RotaryEmbedding :347-366 /
    LlamaRotaryEmbedding modeling_llama.py:119-133.  Returns cos, sin [S, head_dim] (bf16-rounded if emu)."""
    d = cfg.head_dim
    ar = torch.arange(0, d, 2, dtype=torch.int64).float() / d
    if cfg.short_factor is not None:
        fac = cfg.long_factor if kv_seq_len > cfg.orig_max_pos else cfg.short_factor
        ext = torch.tensor(list(fac), dtype=torch.float32)
        inv = 1.0 / (ext * cfg.rope_theta ** ar)
        scale = cfg.max_pos / cfg.orig_max_pos
        sf = 1.0 if scale <= 1.0 else math.sqrt(1 + math.log(scale) / math.log(cfg.orig_max_pos))
    else:
        inv = 1.0 / (cfg.rope_theta ** ar)
        sf = 1.0
    fr = positions.float()[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return _r(emb.cos() * sf, emu), _r(emb.sin() * sf, emu)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _apply_rope(t, cos, sin, emu):
    """apply_rotary_pos_emb modeling_phi3.py:421-445 (each product and the sum round to bf16 under emu)."""
    return _r(_r(t * cos, emu) + _r(_rot_half(t) * sin, emu), emu)


LORA_SCALE = 256.0 / 128.0     # lora_alpha / r, models/llava_next_video.py:217-218


def _wlin(W, name: str, x, emu):
    """One decoder projection `name` (no bias).  If the state dict carries peft LoRA factors for it
    (`<name>.lora_A.default.weight` [r, in], `<name>.lora_B.default.weight` [out, r]) the UN-merged
    peft==0.3.0 forward [ext, restated from its published algorithm -- parity unpinned] is evaluated:
        result = F.linear(x, W);  result += lora_B(lora_A(dropout(x))) * scaling        (dropout = identity in eval)
    exactly as the reference runs it at inference (the adapters are never merged, models/llava_next_video.py:212-224)."""
    y = _lin(x, W[name + ".weight"], None, emu)
    a = W.get(name + ".lora_A.default.weight")
    if a is not None:
        d = _lin(_lin(x, a, None, emu), W[name + ".lora_B.default.weight"], None, emu)
        y = _r(y + _r(d * LORA_SCALE, emu), emu)
    return y


def _qkv(cfg: LLMConfig, W, p, h, emu):
    H, KV, d = cfg.heads, cfg.kv_heads, cfg.head_dim
    if cfg.kind == "phi3":
        qkv = _wlin(W, p + "self_attn.qkv_proj", h, emu)                  # fused :659-663
        q, k, v = qkv[..., : H * d], qkv[..., H * d: H * d + KV * d], qkv[..., H * d + KV * d:]
    else:
        q = _wlin(W, p + "self_attn.q_proj", h, emu)                       # modeling_llama.py:432-434
        k = _wlin(W, p + "self_attn.k_proj", h, emu)
        v = _wlin(W, p + "self_attn.v_proj", h, emu)
    return q, k, v


def _mlp(cfg: LLMConfig, W, p, h, emu):
    if cfg.kind == "phi3":
        gu = _wlin(W, p + "mlp.gate_up_proj", h, emu)                     # :459-464, gate = first half
        g, u = gu.chunk(2, dim=-1)
        a = _r(u * _r(F.silu(g), emu), emu)
        return _wlin(W, p + "mlp.down_proj", a, emu)
    g = _wlin(W, p + "mlp.gate_proj", h, emu)                             # modeling_llama.py:236
    u = _wlin(W, p + "mlp.up_proj", h, emu)
    a = _r(_r(F.silu(g), emu) * u, emu)
    return _wlin(W, p + "mlp.down_proj", a, emu)


def llm_forward(cfg: LLMConfig, W, x: torch.Tensor, emu=False, cache=None, pos0: int = 0, last_only=False):
    """Phi3Model.forward :1249-1383 + Phi3ForCausalLM.forward :1512-1526 (or the Llama twins
    modeling_llama.py:934-1087, :1165-1231) for ONE un-padded sequence.

    x [S, hidden] input embeddings at positions pos0..pos0+S-1.  cache: list (per layer) of
    [K, V] tensors [KV, S_past, d] which is appended in place (DynamicCache.update semantics),
    or None for a full forward.  Returns fp32 logits [S, V] (or [1, V] if last_only)."""
    S = x.shape[0]
    H, KV, d = cfg.heads, cfg.kv_heads, cfg.head_dim
    pos = torch.arange(pos0, pos0 + S)
    cos, sin = rope_cos_sin(cfg, pos, pos0 + S, emu)
    x = _r(x, emu)
    for li in range(cfg.layers):
        p = f"model.layers.{li}."
        h = _rmsnorm(x, W[p + "input_layernorm.weight"], cfg.rms_eps, emu)
        q, k, v = _qkv(cfg, W, p, h, emu)
        q = q.view(S, H, d).transpose(0, 1)
        k = k.view(S, KV, d).transpose(0, 1)
        v = v.view(S, KV, d).transpose(0, 1)
        q, k = _apply_rope(q, cos[None], sin[None], emu), _apply_rope(k, cos[None], sin[None], emu)
        if cache is not None:
            if cache[li] is None:
                cache[li] = [k, v]
            else:
                cache[li] = [torch.cat([cache[li][0], k], dim=1), torch.cat([cache[li][1], v], dim=1)]
            k, v = cache[li]
        Sk = k.shape[1]
        kk = k.repeat_interleave(H // KV, dim=0)          # repeat_kv modeling_llama.py:241-250
        vv = v.repeat_interleave(H // KV, dim=0)
        s = (q @ kk.transpose(-1, -2)) / math.sqrt(d)
        qi = torch.arange(Sk - S, Sk)[:, None]
        ki = torch.arange(Sk)[None, :]
        s = s.masked_fill((ki > qi)[None], float("-inf"))
        o = _r(_r(torch.softmax(s.float(), dim=-1), emu) @ vv, emu)
        o = o.transpose(0, 1).reshape(S, H * d)
        x = _r(x + _wlin(W, p + "self_attn.o_proj", o, emu), emu)
        h = _rmsnorm(x, W[p + "post_attention_layernorm.weight"], cfg.rms_eps, emu)
        x = _r(x + _mlp(cfg, W, p, h, emu), emu)
    if last_only:
        x = x[-1:]
    h = _rmsnorm(x, W["model.norm.weight"], cfg.rms_eps, emu)
    logits = _lin(h, W["lm_head.weight"], W.get("lm_head.bias"), emu)       # bias: SURVEY App. C #3
    return logits.float()


def greedy_generate(cfg: LLMConfig, W, inputs_embeds: torch.Tensor, max_new_tokens: int, eos_token_id: Optional[int],
                    emu=False, use_cache=True, return_margins=False, return_scales=False):
    """language_model.generate(inputs_embeds=..., do_sample=False, num_beams=1) semantics
    (models/llava_next_video.py:655-661; transformers GenerationMixin [ext]): only NEW ids are
    returned, the step that emits eos is included, generation stops after it.

    use_cache=False is the O(n^2) definition (SURVEY §8c 'Greedy-decode oracle'); use_cache=True is
    the mathematically identical KV-cached loop used for the CPU baseline.

    return_margins: also the top-1 minus top-2 logit of every step; return_scales: also max|logit| of every step, so that a test can
    state "ids must agree wherever the margin exceeds x of the logit scale" instead of an absolute margin."""
    e = _r(W["model.embed_tokens.weight"], emu)
    out: List[int] = []
    margins: List[float] = []
    scales: List[float] = []
    if use_cache:
        cache = [None] * cfg.layers
        logits = llm_forward(cfg, W, inputs_embeds, emu, cache, 0, last_only=True)
        n = inputs_embeds.shape[0]
    else:
        seq = inputs_embeds
        logits = llm_forward(cfg, W, seq, emu, None, 0, last_only=True)
    for _ in range(max_new_tokens):
        top2 = torch.topk(logits[-1], 2)
        tok = int(top2.indices[0])
        margins.append(float(top2.values[0] - top2.values[1]))
        scales.append(float(logits[-1].abs().max()))
        out.append(tok)
        if eos_token_id is not None and tok == eos_token_id:
            break
        if len(out) == max_new_tokens:
            break
        if use_cache:
            logits = llm_forward(cfg, W, e[tok][None], emu, cache, n, last_only=True)
            n += 1
        else:
            seq = torch.cat([seq, e[tok][None]], dim=0)
            logits = llm_forward(cfg, W, seq, emu, None, 0, last_only=True)
    if return_scales:
        return out, margins, scales
    return (out, margins) if return_margins else out


def lora_merge(w: torch.Tensor, A: torch.Tensor, B: torch.Tensor, alpha: float = 256.0, r: int = 128) -> torch.Tensor:
    """peft 0.3.0 LoRA linear [ext, parity unpinned]: y = W x + (alpha/r) B(A x)  ==  (W + (alpha/r) B A) x.
    models/llava_next_video.py:212-224 (r=128, alpha=256 -> x2.0)."""
    return w.float() + (alpha / r) * (B.float() @ A.float())


# --------------------------------------------------------------------------------------
# synthetic weights (shared by tests / bench so HIP path and oracle see identical tensors)
# --------------------------------------------------------------------------------------
def frame_normalize(frames_u8: torch.Tensor, mean, std) -> torch.Tensor:
    """ToTensor + Normalize tail of frame_transform (mm_utils/utils.py:177-181) for already-sized frames."""
    x = frames_u8.float() / 255.0
    m = torch.tensor(mean).view(1, 3, 1, 1)
    s = torch.tensor(std).view(1, 3, 1, 1)
    return (x - m) / s


# --------------------------------------------------------------------------------------
# frame pre-processing: frame_transform = ToPILImage -> Resize(S, BICUBIC) -> CenterCrop(S) -> ToTensor -> Normalize
# (mm_utils/utils.py:153-183; call sites inference.py:69-88).  The arithmetic lives in two third-party packages that are NOT
# under /root/reference: torchvision==0.16.2 (requirements.txt:18; size / crop rules, ToTensor, Normalize) and Pillow==11.1.0
# (requirements.txt:13; Image.resize -> src/libImaging/Resample.c).  Both are restated here from their published algorithms.
# PINNED against Pillow itself: oracle/make_golden.py ("pre") resizes seeded images with the Pillow installed in the build
# container (12.2.0 -- the 8-bit two-pass resampler has been stable across these releases) and tests/golden/preprocess.npz holds
# its outputs.  torchvision is absent here: its three integer rules (shortest-edge size, crop offsets with Python round(),
# ToTensor /255) are "parity unpinned", restated from the 0.16.2 sources.
# --------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2      # Resample.c: #define PRECISION_BITS (32 - 8 - 2)


def _pil_bicubic(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box: (bounds [out,2] = (xmin, count), kk [out,ksize] int32).
    Every operation is a separately rounded IEEE double operation, as in the C source compiled without FMA contraction."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale                       # bicubic support = 2.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PIL_PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pil_pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One 8-bit pass (ImagingResampleHorizontal_8bpc / Vertical_8bpc): int32 accumulator starting at 1 << (PRECISION_BITS - 1),
    arithmetic shift, clip8.  img uint8 [H, W, C]."""
    src = img.astype(np.int64)
    n_out = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = n_out
    out = np.empty(shape, np.uint8)
    for o in range(n_out):
        x0, n = int(bounds[o, 0]), int(bounds[o, 1])
        k = kk[o, :n].astype(np.int64)
        if axis == 1:
            acc = (1 << (PIL_PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * k[None, :, None]).sum(1)
            out[:, o, :] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
        else:
            acc = (1 << (PIL_PRECISION_BITS - 1)) + (src[x0:x0 + n, :, :] * k[:, None, None]).sum(0)
            out[o, :, :] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255)
    return out


def pil_resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), BICUBIC) for an RGB uint8 image [H, W, 3] (Resample.c ImagingResampleInner: horizontal pass,
    then vertical pass on its 8-bit result; a pass is skipped when the size does not change)."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        out = _pil_pass(out, *pil_resample_coeffs(w, out_w), axis=1)
    if out_h != h:
        out = _pil_pass(out, *pil_resample_coeffs(h, out_h), axis=0)
    return out


def tv_resized_size(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision 0.16.2 transforms/functional.py _compute_resized_output_size for an int size (shortest edge -> size): (new_h, new_w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def tv_center_crop_offsets(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision 0.16.2 functional.center_crop: (top, left) = int(round((h - S) / 2.0)), int(round((w - S) / 2.0)) -- Python round()."""
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def frame_transform(frame_chw_u8: np.ndarray, size: int, mean, std) -> np.ndarray:
    """mm_utils/utils.py:153-183 for one uint8 frame [3, H, W] (as read_frames_decord hands it over, mm_utils/video_utils.py:91) ->
    float32 [3, size, size].  ToTensor: uint8 -> float32, / 255; Normalize: (x - mean) / std, all in float32."""
    img = np.ascontiguousarray(np.transpose(frame_chw_u8, (1, 2, 0)))          # ToPILImage: HWC RGB
    h, w = img.shape[:2]
    nh, nw = tv_resized_size(h, w, size)
    if (nh, nw) != (h, w):
        img = pil_resize_bicubic(img, nw, nh)
    top, left = tv_center_crop_offsets(nh, nw, size)
    img = img[top:top + size, left:left + size]
    x = np.transpose(img, (2, 0, 1)).astype(np.float32) / np.float32(255)
    m = np.asarray(mean, np.float32).reshape(3, 1, 1)
    s = np.asarray(std, np.float32).reshape(3, 1, 1)
    return (x - m) / s


def synthetic_frame(name: str, h: int, w: int) -> np.ndarray:
    """Seeded uint8 RGB test frame [h, w, 3] (smooth pattern + strong noise: exercises both the antialiasing taps and the clip8
    saturation of the resampler).  Shared by oracle/make_golden.py and the tests so that fixtures hold outputs only."""
    rs = np.random.RandomState(sum(map(ord, name)))
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(xx / 7.0 + c) * np.cos(yy / 11.0 - c) for c in range(3)], -1)
    return np.clip(base + rs.randint(-90, 91, (h, w, 3)), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------
# f3 -- FP8 weight variant (no reference counterpart: SURVEY §8 f3 lists it as a tooling option).  The library quantises every
# decoder projection and lm_head to OCP e4m3 with a per-row POWER-OF-TWO scale and evaluates the model whose weights are the
# de-quantised values; this restates that quantiser so the tests can run the ordinary oracle forward on W_q.
# --------------------------------------------------------------------------------------
_FP8_PROJ = ("qkv_proj", "o_proj", "gate_up_proj", "down_proj", "q_proj", "k_proj", "v_proj", "gate_proj", "up_proj")


def fp8_pow2_dequant(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] (bf16 values) -> fp8_e4m3(w / s) * s with s = smallest power of two >= max|row| / 448 (exact in bf16)."""
    w = w.to(torch.bfloat16).float()
    amax = w.abs().amax(dim=1, keepdim=True)
    m, e = torch.frexp(amax / 448.0)                        # amax / 448 = m * 2^e, m in [0.5, 1)
    e = torch.where(m == 0.5, e - 1, e)
    s = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), e), torch.ones_like(amax))
    return (w / s).to(torch.float8_e4m3fn).float() * s


def fp8_weight_model(W: Dict[str, torch.Tensor], dequant=None) -> Dict[str, torch.Tensor]:
    """State dict of the FP8-variant model: decoder projections and lm_head.weight de-quantised, everything else untouched."""
    dequant = dequant or fp8_pow2_dequant
    out = {}
    for k, v in W.items():
        if k == "lm_head.weight" or (k.endswith(".weight") and any(k.endswith(p + ".weight") for p in _FP8_PROJ)):
            out[k] = dequant(v)
        else:
            out[k] = v
    return out


_E2M1 = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)


def mxfp4_dequant(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] (bf16 values, K % 32 == 0) -> MXFP4 round trip, OCP Microscaling Formats v1.0 [ext]: blocks of 32 consecutive k of one
    row share an E8M0 scale X = 2^(floor(log2(max|w|)) - 2) (emax of E2M1 = 2; an all-zero block stays zero); elements = w / X rounded
    to the nearest E2M1 value {0, .5, 1, 1.5, 2, 3, 4, 6}, ties to the even code, saturating at 6.  The result is exact in bf16."""
    w = w.to(torch.bfloat16).float()
    N, K = w.shape
    b = w.view(N, K // 32, 32)
    amax = b.abs().amax(dim=2, keepdim=True)
    _, ex = torch.frexp(amax)                                   # amax = m * 2^ex, m in [0.5, 1)  =>  floor(log2 amax) = ex - 1
    X = torch.ldexp(torch.ones_like(amax), ex - 1 - 2)
    t = torch.where(amax > 0, b.abs() / X, torch.zeros_like(b))
    grid = torch.tensor(_E2M1)
    # ties go to the even CODE (index): 0.25 -> 0, 0.75 -> 1.0, 1.25 -> 1.0, 1.75 -> 2.0, 2.5 -> 2.0, 3.5 -> 4.0, 5.0 -> 4.0
    code = (t > 0.25).long() + (t >= 0.75).long() + (t > 1.25).long() + (t >= 1.75).long() + (t > 2.5).long() + (t >= 3.5).long() + (t > 5.0).long()
    q = grid[code] * X * torch.sign(b)
    return torch.where(amax > 0, q, torch.zeros_like(q)).view(N, K)


def mxfp4_weight_model(W: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return fp8_weight_model(W, mxfp4_dequant)


# ---------------------------------------------------------------------------------------------------------------------
# do_sample=True token selection.  The reference forwards do_sample / temperature / top_p to HF generate
# (models/llava_next_video.py:655-661; inference.py:45-49: do_sample True, temperature 0.2, top_p None); the arithmetic lives in
# transformers [ext, pinned 4.40.1 in requirements.txt:20]: generation/logits_process.py TemperatureLogitsWarper (scores / T),
# TopKLogitsWarper (remove scores < the k-th largest; GenerationConfig default top_k = 50), TopPLogitsWarper (sort ascending, remove
# while cumsum(softmax) <= 1 - top_p, keep >= 1) and then torch.multinomial(softmax(scores)).  `sample_keep_mask` restates the three
# warpers and is checked against the INSTALLED transformers' classes in tests/test_oracle_golden.py; the draw itself cannot follow
# torch.multinomial's Philox stream, so `sample_token` defines it as Gumbel-max over a counter hash -- a sample of exactly
# softmax(scores) on the kept set -- and the HIP sampler must reproduce kept set and token.
# ---------------------------------------------------------------------------------------------------------------------
def sample_keep_mask(logits: np.ndarray, temperature: float, top_k: int, top_p: Optional[float]) -> np.ndarray:
    s = np.asarray(logits, dtype=np.float64) / float(temperature)
    keep = np.ones(s.shape, dtype=bool)
    if top_k and 0 < top_k < s.size:
        kth = np.sort(s)[-top_k]
        keep &= s >= kth                                          # TopKLogitsWarper: scores < topk(...)[-1] are removed (ties stay)
    if top_p is not None and 0.0 < top_p < 1.0:
        sm = np.where(keep, s, -np.inf)
        p = np.exp(sm - sm.max()); p /= p.sum()
        order = np.argsort(sm, kind="stable")                      # ascending, as torch.sort(descending=False)
        cum = np.cumsum(p[order])
        rem = cum <= (1.0 - top_p)
        rem[-1] = False                                            # min_tokens_to_keep = 1
        keep2 = np.ones_like(keep); keep2[order[rem]] = False
        keep &= keep2
    return keep


def _fmix32_np(h):
    h = np.asarray(h, dtype=np.uint64) & 0xFFFFFFFF
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85EBCA6B)) & 0xFFFFFFFF
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xC2B2AE35)) & 0xFFFFFFFF
    h ^= h >> np.uint64(16)
    return h


def sample_uniforms(n: int, seed: int, stream: int, step: int) -> np.ndarray:
    """u_i in (0, 1), i < n: the counter hash the device sampler uses (a pure function of seed, stream, step, i)."""
    M = 0xFFFFFFFF
    k0 = int(_fmix32_np((seed & M) ^ 0x9E3779B9)); k1 = int(_fmix32_np(((seed >> 32) & M) ^ k0 ^ 0x85EBCA77))
    kk = int(_fmix32_np(k1 ^ int(_fmix32_np((stream * 0x9E3779B1 + 0x7F4A7C15) & M)) ^ int(_fmix32_np((step * 0x85EBCA77 + 0x165667B1) & M))))
    kk2 = int(_fmix32_np((kk + 0x632BE5AB) & M))
    i = np.arange(n, dtype=np.uint64)
    h = _fmix32_np(_fmix32_np((i + np.uint64(kk)) & M) ^ np.uint64(kk2))
    return ((h >> np.uint64(8)).astype(np.float64) + 0.5) / 16777216.0


def sample_token(logits: np.ndarray, temperature: float, top_k: int, top_p: Optional[float], seed: int, stream: int, step: int):
    """-> (token, margin, keep): Gumbel-max draw from softmax(logits / T) restricted to the HF-kept set; margin = perturbed score of the
    winner minus the runner-up's (how far the draw is from flipping under float rounding)."""
    keep = sample_keep_mask(logits, temperature, top_k, top_p)
    s = np.asarray(logits, dtype=np.float64)
    u = sample_uniforms(s.size, seed, stream, step)
    sc = (s - s.max()) / float(temperature) - np.log(-np.log(u))
    sc = np.where(keep, sc, -np.inf)
    o = np.argsort(-sc, kind="stable")
    return int(o[0]), float(sc[o[0]] - sc[o[1]]) if keep.sum() > 1 else float("inf"), keep
