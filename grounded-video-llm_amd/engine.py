"""Engine: one gvl_ctx on one GPU, with torch tensors as the caller-owned I/O buffers.

PyTorch here is plumbing only (device allocations, the current HIP stream); all arithmetic runs
inside libgvl.so.  If the library cannot be loaded this module raises -- there is no eager fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as L

bf = torch.bfloat16


@dataclass
class TowerGeometry:
    """Mirrors gvl_config (include/gvl.h).  Defaults = the reference's Phi-3.5 / 96-frame configuration
    (models/llava_next_video.py:56-71, models/internvideo2.py:1089-1114, HF Phi-3.5 config [ext])."""
    llm: str = "phi3.5"                      # 'phi3.5' | 'llama3' | 'vicuna'
    clip_hidden: int = 1024
    clip_inter: int = 4096
    clip_layers: int = 24                    # layers in the checkpoint; layers-1 are executed (hidden_states[-2])
    clip_heads: int = 16
    clip_image: int = 336
    clip_patch: int = 14
    iv2_dim: int = 1408
    iv2_inter: int = 6144
    iv2_depth: int = 40                      # blocks in the checkpoint; depth-1 are executed (x_vis_return_idx=-2)
    iv2_heads: int = 16
    iv2_image: int = 224
    iv2_patch: int = 14
    frames_per_seg: int = 8
    hidden: int = 3072
    inter: int = 8192
    layers: int = 32
    heads: int = 32
    kv_heads: int = 32
    vocab: int = 32366                       # 32064 + 302 temporal tokens (llava_next_video.py:235-268)
    rms_eps: float = 1e-5
    lm_head_bias: bool = True
    rope_theta: float = 10000.0
    rope_short: Optional[Sequence[float]] = None
    rope_long: Optional[Sequence[float]] = None
    rope_max_pos: int = 131072
    rope_orig_max_pos: int = 4096
    max_seq: int = 8192
    max_segs: int = 12
    kv_pages: int = 128
    max_prefill: int = 4096
    decode_fp8: int = 0                      # quantised weight variant of the LLM (opt-in, not the reference's numerics): 1 / True = FP8 e4m3 with per-row
                                             # power-of-two scales, 2 = MXFP4 (E2M1 elements, E8M0 scale per 32 k); 0 = bf16

    @property
    def kind(self) -> str:
        return "phi3" if self.llm == "phi3.5" else "llama"

    @staticmethod
    def vicuna_7b(**kw) -> "TowerGeometry":
        """LLaVA-Next vicuna-7b-v1.5 base [ext]: Llama-2 architecture -- MHA 32 x 128, inter 11008, theta 1e4, vocab 32000 + 302."""
        base = dict(llm="vicuna", hidden=4096, inter=11008, layers=32, heads=32, kv_heads=32, vocab=32000 + 302, rope_theta=10000.0,
                    rope_short=None, rope_long=None, rope_orig_max_pos=0, max_seq=4096)
        base.update(kw)
        return TowerGeometry(**base)

    def apply_hf_config(self, cfg: dict) -> "TowerGeometry":
        """Fill the LLM fields from an HF config.json dict (the `text`/top-level keys of Phi-3.5(-vision) / Llama configs [ext]):
        hidden sizes, rms_norm_eps, rope_theta and -- for Phi-3.5 -- the LongRoPE short/long factors and both context limits that
        Phi3LongRoPEScaledRotaryEmbedding reads (models/modeling_phi3.py:369-409).  Returns self."""
        cfg = cfg.get("text_config", cfg)
        m = {"hidden_size": "hidden", "intermediate_size": "inter", "num_hidden_layers": "layers", "num_attention_heads": "heads",
             "num_key_value_heads": "kv_heads", "rms_norm_eps": "rms_eps", "rope_theta": "rope_theta"}
        for k, a in m.items():
            if cfg.get(k) is not None:
                setattr(self, a, type(getattr(self, a))(cfg[k]))
        rs = cfg.get("rope_scaling")
        if rs:
            kind = rs.get("type", rs.get("rope_type"))
            # Phi3Config._rope_scaling_adjustment renames the legacy types "su" / "yarn" to "longrope" (modeling_phi3.py:191-202)
            if kind not in ("longrope", "su", "yarn") or "short_factor" not in rs or "long_factor" not in rs:
                raise ValueError(f"unsupported rope_scaling {rs!r}: need type longrope with short_factor and long_factor (modeling_phi3.py:204-240)")
            self.rope_short, self.rope_long = [float(v) for v in rs["short_factor"]], [float(v) for v in rs["long_factor"]]
            self.rope_max_pos = int(cfg.get("max_position_embeddings", self.rope_max_pos))
            self.rope_orig_max_pos = int(cfg.get("original_max_position_embeddings", self.rope_orig_max_pos))
        return self

    @staticmethod
    def llama3_8b(**kw) -> "TowerGeometry":
        base = dict(llm="llama3", hidden=4096, inter=14336, layers=32, heads=32, kv_heads=8, vocab=128558, rope_theta=500000.0,
                    rope_short=None, rope_long=None, rope_orig_max_pos=0, max_seq=8192)
        base.update(kw)
        return TowerGeometry(**base)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    def __init__(self, geo: TowerGeometry, device: str = "cuda:0", towers: Sequence[str] = ("clip", "iv2", "llm")):
        self.lib = L.load()
        self.geo = geo
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("grounded_video_llm_amd.Engine needs a HIP device (cuda:N); there is no CPU path")
        torch.cuda.set_device(self.device)
        cfg = L.GvlConfig()
        cfg.llm_kind = L.LLM_PHI3 if geo.kind == "phi3" else L.LLM_LLAMA
        if "clip" in towers:
            cfg.clip_hidden, cfg.clip_inter, cfg.clip_layers_run = geo.clip_hidden, geo.clip_inter, geo.clip_layers - 1
            cfg.clip_heads, cfg.clip_image, cfg.clip_patch = geo.clip_heads, geo.clip_image, geo.clip_patch
        if "iv2" in towers:
            cfg.iv2_dim, cfg.iv2_inter, cfg.iv2_blocks_run = geo.iv2_dim, geo.iv2_inter, geo.iv2_depth - 1
            cfg.iv2_heads, cfg.iv2_image, cfg.iv2_patch, cfg.iv2_frames_per_seg = geo.iv2_heads, geo.iv2_image, geo.iv2_patch, geo.frames_per_seg
        cfg.hidden = geo.hidden if ("llm" in towers or "proj" in towers) else 0
        if "llm" in towers:
            cfg.inter, cfg.layers, cfg.heads, cfg.kv_heads, cfg.vocab = geo.inter, geo.layers, geo.heads, geo.kv_heads, geo.vocab
            cfg.rms_eps = geo.rms_eps
            cfg.lm_head_bias = 1 if geo.lm_head_bias else 0
            cfg.rope_orig_max_pos = geo.rope_orig_max_pos if geo.rope_long is not None else 0
            cfg.max_seq = geo.max_seq
            cfg.kv_pages, cfg.max_prefill = geo.kv_pages, geo.max_prefill
            cfg.decode_fp8 = int(geo.decode_fp8)
        cfg.max_segs = geo.max_segs
        self.cfg = cfg
        self.towers = tuple(towers)
        h = C.c_void_p()
        rc = self.lib.gvl_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise L.GvlError(f"gvl_create failed ({rc}): {self.lib.gvl_last_error(None).decode()}")
        self.ctx = h
        self._finalized = False

    # ---- lifecycle -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            self.lib.gvl_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        L.check(self.lib, self.ctx, rc, what)

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- weights ---------------------------------------------------------------------------------
    def load_packed(self, packed: Dict[str, torch.Tensor]):
        for name, t in packed.items():
            if t.dtype == torch.float32:
                dt = L.F32
            elif t.dtype == bf:
                dt = L.BF16
            else:
                raise TypeError(f"{name}: packed tensors must be float32 or bfloat16, got {t.dtype}")
            t = t.contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*(list(t.shape) or [1]))
            self._chk(self.lib.gvl_load_weight(self.ctx, name.encode(), C.c_void_p(t.data_ptr()), dt, shape, max(t.dim(), 1),
                                               1 if t.is_cuda else 0), f"gvl_load_weight({name})")

    def load_packed_file(self, path: str) -> int:
        """gvl_load_packed: the C++ loader of a `gvl-packed-1` file (tools/pack_checkpoint.py); no torch / safetensors involved."""
        n = C.c_int(0)
        self._chk(self.lib.gvl_load_packed(self.ctx, os.fsencode(path), C.byref(n)), "gvl_load_packed")
        return n.value

    def finalize(self):
        self._chk(self.lib.gvl_finalize_weights(self.ctx), "gvl_finalize_weights")
        self._finalized = True

    def kv_info(self) -> Dict[str, int]:
        t, f, b, m = C.c_int(0), C.c_int(0), C.c_int64(0), C.c_int(0)
        self._chk(self.lib.gvl_kv_info(self.ctx, C.byref(t), C.byref(f), C.byref(b), C.byref(m)), "gvl_kv_info")
        return {"total_pages": t.value, "free_pages": f.value, "pool_bytes": b.value, "max_live_seqs": m.value, "tokens": t.value * 64}

    def decode_group_info(self) -> Dict[str, int]:
        """gvl_decode_group_info: the group sizes one batched decode step takes ({max_group: 16, any_size: 1} on the skinny-MFMA path;
        {4, 0} = sizes 1 / 2 / 4 on the VALU fallback)."""
        m, a = C.c_int(0), C.c_int(0)
        self._chk(self.lib.gvl_decode_group_info(self.ctx, C.byref(m), C.byref(a)), "gvl_decode_group_info")
        return {"max_group": m.value, "any_size": a.value}

    # ---- multi-GPU exchange through the C ABI (RCCL dlopen'ed by libgvl) -----------------------------
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.lib.gvl_comm_unique_id(buf)
        if rc != 0:
            raise L.GvlError(f"gvl_comm_unique_id failed ({rc}): {self.lib.gvl_last_error(None).decode()}")
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, world: int):
        assert len(uid) == 128
        self._chk(self.lib.gvl_comm_init(self.ctx, uid, int(rank), int(world)), "gvl_comm_init")
        self.comm_world = world

    def comm_count(self) -> int:
        """Ranks RCCL itself reports for the ctx's communicator (ncclCommCount); 1 without a communicator."""
        n = C.c_int(0)
        self._chk(self.lib.gvl_comm_count(self.ctx, C.byref(n)), "gvl_comm_count")
        return n.value

    def allgather_visual(self, local: torch.Tensor) -> torch.Tensor:
        """local bf16 [rows, hidden] (same rows on every rank) -> [world * rows, hidden] in rank order (ncclAllGather on the current stream)."""
        local = local.contiguous()
        world = getattr(self, "comm_world", 1)
        out = torch.empty((world * local.shape[0], local.shape[1]), dtype=bf, device=self.device)
        self._chk(self.lib.gvl_allgather_visual(self.ctx, None, _ptr(local), local.shape[0], local.shape[1], _ptr(out), self.stream), "gvl_allgather_visual")
        return out

    def allgatherv_visual(self, local: torch.Tensor, rows_per_rank, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Uneven blocks straight into the segment-ordered prefix: rank r's `rows_per_rank[r]` rows land at row offset sum(rows_per_rank[:r]) of the result on
        every rank (gvl_allgatherv_visual: one ncclGroup of per-rank broadcasts on the current stream; no padding, no re-assembly copy).  `out`: a
        preallocated [sum(rows), hidden] bf16 buffer (the step's prefix) to gather into."""
        local = local.contiguous()
        rows = [int(r) for r in rows_per_rank]
        hidden = local.shape[1]
        if out is None:
            out = torch.empty((sum(rows), hidden), dtype=bf, device=self.device)
        assert out.is_contiguous() and out.shape == (sum(rows), hidden) and out.dtype == bf
        arr = (C.c_int * len(rows))(*rows)
        self._chk(self.lib.gvl_allgatherv_visual(self.ctx, None, _ptr(local) if local.numel() else None, arr, hidden, _ptr(out), self.stream), "gvl_allgatherv_visual")
        return out

    # ---- vision ----------------------------------------------------------------------------------
    def clip_encode(self, px: torch.Tensor) -> torch.Tensor:
        """vision_tower(px, output_hidden_states=True).hidden_states[-2][:, 1:] (llava_next_video.py:504-505)."""
        px = px.to(self.device, torch.float32).contiguous()
        n = px.shape[0]
        P = (self.geo.clip_image // self.geo.clip_patch) ** 2
        out = torch.empty((n, P, self.geo.clip_hidden), dtype=torch.float32, device=self.device)
        self._chk(self.lib.gvl_clip_encode(self.ctx, _ptr(px), n, _ptr(out), self.stream), "gvl_clip_encode")
        return out

    def iv2_encode(self, px: torch.Tensor) -> torch.Tensor:
        """video_encoder(x, None, False, x_vis_return_idx=-2, x_vis_only=True)[:, 1:, :] (llava_next_video.py:532).
        px [n,3,T,H,W]."""
        px = px.to(self.device, torch.float32).contiguous()
        n, T = px.shape[0], px.shape[2]
        assert T == self.geo.frames_per_seg
        Lp = (self.geo.iv2_image // self.geo.iv2_patch) ** 2
        out = torch.empty((n, T * Lp, self.geo.iv2_dim), dtype=bf, device=self.device)
        self._chk(self.lib.gvl_iv2_encode(self.ctx, _ptr(px), n, _ptr(out), self.stream), "gvl_iv2_encode")
        return out

    @property
    def tokens_per_seg(self) -> int:
        return int(self.lib.gvl_tokens_per_seg(self.ctx))

    def build_visual(self, clip_feats: torch.Tensor, iv2_feats: torch.Tensor) -> torch.Tensor:
        n = clip_feats.shape[0]
        out = torch.empty((n * self.tokens_per_seg, self.geo.hidden), dtype=bf, device=self.device)
        self._chk(self.lib.gvl_build_visual(self.ctx, _ptr(clip_feats.contiguous()), _ptr(iv2_feats.contiguous()), n, _ptr(out), self.stream),
                  "gvl_build_visual")
        return out

    def encode_segments(self, spatial_px: torch.Tensor, temporal_px: torch.Tensor) -> torch.Tensor:
        """encode_images() for n segments: spatial [n,3,336,336], temporal [n,3,T,224,224] -> [n*L, hidden] bf16."""
        sp = spatial_px.to(self.device, torch.float32).contiguous()
        tp = temporal_px.to(self.device, torch.float32).contiguous()
        n = sp.shape[0]
        out = torch.empty((n * self.tokens_per_seg, self.geo.hidden), dtype=bf, device=self.device)
        self._chk(self.lib.gvl_encode_segments(self.ctx, _ptr(sp), _ptr(tp), n, _ptr(out), self.stream), "gvl_encode_segments")
        return out

    def preprocess_frames(self, frames_u8: torch.Tensor, size: int, mean, std) -> torch.Tensor:
        """frame_transform of the reference on the GPU (bit-exact to PIL bicubic + torchvision glue): uint8 frames [n,H,W,3]
        (decoder order) or [n,3,H,W] -> f32 [n,3,size,size].  mm_utils/utils.py:153-183."""
        assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4
        if frames_u8.shape[-1] == 3 and frames_u8.shape[1] != 3:
            layout, (n, H, W) = 0, (frames_u8.shape[0], frames_u8.shape[1], frames_u8.shape[2])
        else:
            layout, (n, H, W) = 1, (frames_u8.shape[0], frames_u8.shape[2], frames_u8.shape[3])
        fr = frames_u8.to(self.device).contiguous()
        out = torch.empty((n, 3, size, size), dtype=torch.float32, device=self.device)
        m = (C.c_float * 3)(*[float(v) for v in mean]); s = (C.c_float * 3)(*[float(v) for v in std])
        self._chk(self.lib.gvl_preprocess_frames(self.ctx, _ptr(fr), n, H, W, layout, int(size), m, s, _ptr(out), self.stream), "gvl_preprocess_frames")
        return out

    # ---- LLM -------------------------------------------------------------------------------------
    def splice(self, input_ids: Sequence[int], visual: torch.Tensor) -> torch.Tensor:
        ids = (C.c_int64 * len(input_ids))(*[int(i) for i in input_ids])
        nv = visual.shape[0]
        out = torch.empty((len(input_ids) - 1 + nv, self.geo.hidden), dtype=bf, device=self.device)
        S = C.c_int(0)
        self._chk(self.lib.gvl_splice(self.ctx, ids, len(input_ids), _ptr(visual.contiguous()), nv, _ptr(out), C.byref(S), self.stream), "gvl_splice")
        assert S.value == out.shape[0]
        return out

    def seq_alloc(self, max_tokens: int) -> int:
        s = C.c_int(-1)
        self._chk(self.lib.gvl_seq_alloc(self.ctx, int(max_tokens), C.byref(s)), "gvl_seq_alloc")
        return s.value

    def seq_free(self, seq: int):
        self._chk(self.lib.gvl_seq_free(self.ctx, int(seq)), "gvl_seq_free")

    def seq_fork(self, src: int, n_tokens: int, max_tokens: int) -> int:
        """A new sequence that SHARES the first n_tokens (multiple of 64) of `src` -- whole KV pages, referenced not copied (gvl_seq_fork)."""
        sid = C.c_int(-1)
        self._chk(self.lib.gvl_seq_fork(self.ctx, int(src), int(n_tokens), int(max_tokens), C.byref(sid)), "gvl_seq_fork")
        return sid.value

    def seq_clone(self, src: int, max_tokens: int) -> int:
        """A copy of `src` at its current length: whole pages shared by reference, the partial last page copied (gvl_seq_clone; beam search)."""
        sid = C.c_int(-1)
        self._chk(self.lib.gvl_seq_clone(self.ctx, int(src), int(max_tokens), C.byref(sid), self.stream), "gvl_seq_clone")
        return sid.value

    def prefill_extend(self, seq: int, embeds_new: torch.Tensor, want_logits: bool = False) -> Optional[torch.Tensor]:
        """Prefill of the rows that FOLLOW a forked prefix (gvl_prefill_extend): positions prefix .. prefix + n - 1, attention over prefix + new rows."""
        embeds_new = embeds_new.contiguous()
        assert embeds_new.dtype == bf and embeds_new.dim() == 2
        lg = torch.empty((self.geo.vocab,), dtype=torch.float32, device=self.device) if want_logits else None
        self._chk(self.lib.gvl_prefill_extend(self.ctx, int(seq), _ptr(embeds_new), embeds_new.shape[0], _ptr(lg), self.stream), "gvl_prefill_extend")
        return lg

    def prefill(self, seq: int, embeds: torch.Tensor, want_logits: bool = False) -> Optional[torch.Tensor]:
        embeds = embeds.contiguous()
        logits = torch.empty((self.geo.vocab,), dtype=torch.float32, device=self.device) if want_logits else None
        self._chk(self.lib.gvl_prefill(self.ctx, seq, _ptr(embeds), embeds.shape[0], _ptr(logits), self.stream), "gvl_prefill")
        return logits

    def decode_greedy(self, seq: int, max_new: int, eos_id: Optional[int]) -> List[int]:
        buf = (C.c_int32 * max_new)()
        n = C.c_int(0)
        self._chk(self.lib.gvl_decode_greedy(self.ctx, seq, int(max_new), -1 if eos_id is None else int(eos_id), buf, C.byref(n), self.stream),
                  "gvl_decode_greedy")
        return [int(buf[i]) for i in range(n.value)]

    def prefill_batch(self, seqs: Sequence[int], embeds: Sequence[torch.Tensor]) -> None:
        """Prefill several sequences together -- equal or ragged lengths (one pass of the decoder GEMMs over all their rows)."""
        es = [e.contiguous() for e in embeds]
        n = len(seqs)
        ids = (C.c_int * n)(*[int(s) for s in seqs])
        ptrs = (C.c_void_p * n)(*[e.data_ptr() for e in es])
        lens = (C.c_int * n)(*[int(e.shape[0]) for e in es])
        self._chk(self.lib.gvl_prefill_varlen(self.ctx, ids, n, ptrs, lens, self.stream), "gvl_prefill_varlen")

    def decode_greedy_batch(self, seqs: Sequence[int], max_new: int, eos_id: Optional[int]) -> List[List[int]]:
        """Greedy decode of several freshly prefilled sequences together (weights streamed once per step per group of up to 16 sequences)."""
        n = len(seqs)
        ids = (C.c_int * n)(*[int(s) for s in seqs])
        buf = (C.c_int32 * (n * max_new))()
        nout = (C.c_int * n)()
        self._chk(self.lib.gvl_decode_greedy_batch(self.ctx, ids, n, int(max_new), -1 if eos_id is None else int(eos_id), buf, nout, self.stream),
                  "gvl_decode_greedy_batch")
        return [[int(buf[i * max_new + j]) for j in range(nout[i])] for i in range(n)]

    def forward_loss(self, embeds: torch.Tensor, labels: Sequence[int]) -> Tuple[float, int]:
        """Causal-LM loss terms of one sample: (sum of token nll, number of labelled tokens); labels use -100 = ignore."""
        embeds = embeds.contiguous()
        S = embeds.shape[0]
        assert len(labels) == S, "labels must cover every row of inputs_embeds"
        lab = (C.c_int64 * S)(*[int(v) for v in labels])
        nll, n = C.c_double(0.0), C.c_int(0)
        seq = self.seq_alloc(S)
        try:
            self._chk(self.lib.gvl_forward_loss(self.ctx, seq, _ptr(embeds), S, lab, C.byref(nll), C.byref(n), self.stream), "gvl_forward_loss")
        finally:
            self.seq_free(seq)
        return nll.value, n.value

    def decode_steps(self, seqs: Sequence[int], n_steps: int) -> None:
        """Advance every listed sequence by n_steps greedy tokens (mixed generation steps allowed; asynchronous)."""
        n = len(seqs)
        ids = (C.c_int * n)(*[int(s) for s in seqs])
        self._chk(self.lib.gvl_decode_steps(self.ctx, ids, n, int(n_steps), self.stream), "gvl_decode_steps")

    def seq_read(self, seq: int, first: int = 0, cap: int = 4096) -> List[int]:
        """Ids generated so far by `seq`, from generation index `first` (synchronises the stream)."""
        buf = (C.c_int32 * max(cap, 1))()
        n = C.c_int(0)
        self._chk(self.lib.gvl_seq_read(self.ctx, int(seq), int(first), buf, int(cap), C.byref(n), self.stream), "gvl_seq_read")
        return [int(buf[i]) for i in range(max(0, min(n.value - first, cap)))]

    def decode_step_logits(self, seq: int, tok: int) -> torch.Tensor:
        logits = torch.empty((self.geo.vocab,), dtype=torch.float32, device=self.device)
        self._chk(self.lib.gvl_decode_step_logits(self.ctx, seq, int(tok), _ptr(logits), self.stream), "gvl_decode_step_logits")
        return logits

    def decode_step_logits_batch(self, seqs: Sequence[int], toks: Sequence[int]) -> torch.Tensor:
        """One teacher-forced decode step for several sequences together (one stream of the weights): [n, vocab] fp32 logits."""
        n = len(seqs)
        logits = torch.empty((n, self.geo.vocab), dtype=torch.float32, device=self.device)
        ids = (C.c_int * n)(*[int(s) for s in seqs]); tk = (C.c_int32 * n)(*[int(t) for t in toks])
        self._chk(self.lib.gvl_decode_step_logits_batch(self.ctx, ids, n, tk, _ptr(logits), self.stream), "gvl_decode_step_logits_batch")
        return logits

    def generate_ids(self, embeds: torch.Tensor, max_new_tokens: int, eos_id: Optional[int]) -> List[int]:
        """language_model.generate(inputs_embeds=..., greedy): returns only the NEW ids (eos included)."""
        S = embeds.shape[0]
        seq = self.seq_alloc(min(S + max_new_tokens, self.geo.max_seq))
        try:
            self.prefill(seq, embeds)
            return self.decode_greedy(seq, max_new_tokens, eos_id)
        finally:
            self.seq_free(seq)

    # ---- measurement -----------------------------------------------------------------------------
    def trace_marker(self, tag: int = 0, stream=None):
        """gvl_trace_marker on `stream` (a torch stream; default: the current one): a named empty dispatch that brackets a region of a kernel trace."""
        st = C.c_void_p(stream.cuda_stream) if stream is not None else self.stream
        self._chk(self.lib.gvl_trace_marker(int(tag), st), "gvl_trace_marker")

    def prof_enable(self, on: bool):
        self._chk(self.lib.gvl_prof_enable(self.ctx, 1 if on else 0), "gvl_prof_enable")

    def prof_read(self, cat: int):
        ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
        self._chk(self.lib.gvl_prof_read(self.ctx, cat, C.byref(ms), C.byref(n), C.byref(w)), "gvl_prof_read")
        return ms.value, n.value, w.value

    # ---- operator-level (parity tests) -------------------------------------------------------------
    def op_gemm(self, A, W, bias=None, gamma=None, resid=None, act=L.ACT_NONE, out_f32=False, tile_cfg=0):
        M, K = A.shape
        N = W.shape[0]
        n_out = N // 2 if act == L.ACT_SILU_MUL else N
        Cc = torch.empty((M, n_out), dtype=torch.float32 if out_f32 else bf, device=self.device)
        self._chk(self.lib.gvl_op_gemm(self.ctx, _ptr(A.contiguous()), _ptr(W.contiguous()), _ptr(Cc), M, N, K, _ptr(bias), _ptr(gamma),
                                       _ptr(resid), act, 1 if out_f32 else 0, tile_cfg, self.stream), "gvl_op_gemm")
        return Cc

    def op_gemm_rows(self, A, W, bias=None, gamma=None, resid=None, act=L.ACT_NONE, rowscale=None, want_rowsq=False, tile_cfg=0):
        """gvl_op_gemm_rows: the fused-RMSNorm epilogues.  rowscale [M] f32 multiplies the accumulator rows; want_rowsq: also returns the sums of squares of the
        bf16 outputs per aligned 64-column block, [M, N // 64] f32 (NaN-filled first: every slot must be written)."""
        M, K = A.shape
        N = W.shape[0]
        n_out = N // 2 if act == L.ACT_SILU_MUL else N
        Cc = torch.empty((M, n_out), dtype=bf, device=self.device)
        sq = torch.full((M, N // 64), float("nan"), dtype=torch.float32, device=self.device) if want_rowsq else None
        self._chk(self.lib.gvl_op_gemm_rows(self.ctx, _ptr(A.contiguous()), _ptr(W.contiguous()), _ptr(Cc), M, N, K, _ptr(bias), _ptr(gamma), _ptr(resid), act,
                                            _ptr(rowscale), _ptr(sq), N // 64 if want_rowsq else 0, tile_cfg, self.stream), "gvl_op_gemm_rows")
        return (Cc, sq) if want_rowsq else Cc

    def op_fold_gamma(self, W, gamma):
        W = W.contiguous()
        out = torch.empty_like(W)
        self._chk(self.lib.gvl_op_fold_gamma(self.ctx, _ptr(W), _ptr(gamma.contiguous()), _ptr(out), W.shape[0], W.shape[1], self.stream), "gvl_op_fold_gamma")
        return out

    def op_rowsq_finish(self, sq, cols, eps, b0=0, nblk=None):
        sq = sq.contiguous()
        nblk = sq.shape[1] - b0 if nblk is None else nblk
        rs = torch.empty((sq.shape[0],), dtype=torch.float32, device=self.device)
        self._chk(self.lib.gvl_op_rowsq_finish(self.ctx, _ptr(sq), sq.shape[1], b0, nblk, _ptr(rs), sq.shape[0], cols, float(eps), self.stream), "gvl_op_rowsq_finish")
        return rs

    def op_attention(self, qkv: torch.Tensor, B, S, H, KV, Dr, scale, causal):
        """qkv bf16 [B*S, (H+2KV)*Dr] fused rows -> out bf16 [B*S, H*Dr]."""
        qkv = qkv.contiguous()
        out = torch.empty((B * S, H * Dr), dtype=bf, device=self.device)
        base = qkv.data_ptr()
        self._chk(self.lib.gvl_op_attention(self.ctx, C.c_void_p(base), C.c_void_p(base + 2 * H * Dr), C.c_void_p(base + 2 * (H + KV) * Dr),
                                            _ptr(out), B, S, H, KV, Dr, float(scale), int(causal), self.stream), "gvl_op_attention")
        return out

    def op_layernorm(self, x, w, b, eps):
        y = torch.empty(x.shape, dtype=bf, device=self.device)
        self._chk(self.lib.gvl_op_layernorm(self.ctx, _ptr(x.contiguous()), _ptr(w), _ptr(b), _ptr(y), x.shape[0], x.shape[1], float(eps), self.stream),
                  "gvl_op_layernorm")
        return y

    def op_rmsnorm(self, x, w, eps):
        y = torch.empty(x.shape, dtype=bf, device=self.device)
        self._chk(self.lib.gvl_op_rmsnorm(self.ctx, _ptr(x.contiguous()), _ptr(w), _ptr(y), x.shape[0], x.shape[1], float(eps), self.stream),
                  "gvl_op_rmsnorm")
        return y

    def debug_set(self, key: str, value: int):
        """gvl_debug_set: result-neutral launch parameters ("decode_attn_cpb", "decode_attn_hpb", "decode_graph"); 0 = the launcher's choice."""
        self._chk(self.lib.gvl_debug_set(self.ctx, key.encode(), int(value)), "gvl_debug_set")

    def set_sampling(self, do_sample, temperature=1.0, top_k=50, top_p=None, seed=0):
        """Token selection of every later prefill / decode call: greedy argmax (do_sample False) or temperature / top-k / top-p sampling
        on the device (HF generate's do_sample=True; `top_k` 50 is HF's GenerationConfig default, `top_p` None / 1.0 = off)."""
        self._chk(self.lib.gvl_set_sampling(self.ctx, int(bool(do_sample)), float(temperature), int(top_k or 0),
                                            float(top_p) if top_p is not None else 0.0, int(seed) & (2 ** 64 - 1)), "gvl_set_sampling")

    def op_sample(self, logits, temperature, top_k, top_p, seed, streams, steps):
        """logits f32 [B, n] -> int32 [B] drawn tokens (row b: random stream streams[b], generation step steps[b])."""
        import ctypes as C
        B, n = logits.shape
        st = (C.c_uint32 * B)(*[int(x) for x in streams])
        steps_d = torch.tensor(list(steps), dtype=torch.int32, device=self.device)
        out = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._chk(self.lib.gvl_op_sample(self.ctx, _ptr(logits.contiguous()), n, B, float(temperature), int(top_k or 0), float(top_p or 0.0),
                                         int(seed) & (2 ** 64 - 1), st, _ptr(steps_d), _ptr(out), self.stream), "gvl_op_sample")
        return out

    def op_dgemm(self, W, x, bias=None):
        """x bf16 [B, K] (B <= 16) -> y f32 [B, N]: the skinny MFMA decode GEMM."""
        N, K = W.shape
        B = x.shape[0]
        y = torch.empty((B, N), dtype=torch.float32, device=self.device)
        self._chk(self.lib.gvl_op_dgemm(self.ctx, _ptr(W.contiguous()), _ptr(x.contiguous()), _ptr(bias), _ptr(y), N, K, B, self.stream), "gvl_op_dgemm")
        return y

    def op_gemv(self, W, x, bias=None):
        N, K = W.shape
        y = torch.empty((N,), dtype=torch.float32, device=self.device)
        self._chk(self.lib.gvl_op_gemv(self.ctx, _ptr(W.contiguous()), _ptr(x.contiguous()), _ptr(bias), _ptr(y), N, K, self.stream), "gvl_op_gemv")
        return y
