"""Host-side mirror of the reference's model surface:  LLAVA_NEXT_VIDEO(...).generate(samples, **kw).

Same constructor arguments and `generate()` contract as models/llava_next_video.py:73-89,616-666:
`samples` = {"prompts": [str], "temporal_pixel_values": [bs,F,3,224,224], "spatial_pixel_values":
[bs,S,3,336,336], "video_ids": [...]}; returns List[str].  Everything numeric runs in libgvl.so
(grounded_video_llm_amd.engine); this file only does the text plumbing and the call order.

Differences that are deliberate and documented: 23 CLIP layers instead of 24
(#2), last-row lm_head (#7), paged KV instead of DynamicCache (#8), glb_GN projected once (#5), LoRA
merged at load (#16), greedy decoding only on this tier (the reference CLI defaults to sampling).
"""
from __future__ import annotations

import dataclasses

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import dist as gdist
from . import prompts as P
from . import weights as Wt
from .engine import Engine, TowerGeometry

bf = torch.bfloat16


class SyntheticTokenizer:
    """Stand-in for the HF tokenizer (tokenizer files are not available offline): whitespace words -> ids by a
    stable hash; the 302 temporal tokens map to the last 302 rows of the vocabulary like `add_tokens` does."""

    def __init__(self, vocab: int, num_temporal_tokens: int = 300, bos_token_id: Optional[int] = 1, eos_token_id: int = 2, pad_token_id: int = 0):
        self.vocab, self.bos_token_id, self.eos_token_id, self.pad_token_id = vocab, bos_token_id, eos_token_id, pad_token_id
        self.added = P.temporal_token_strings(num_temporal_tokens)
        self.base = vocab - len(self.added)
        self.tok2id = {t: self.base + i for i, t in enumerate(self.added)}
        self.id2tok = {v: k for k, v in self.tok2id.items()}

    def __len__(self):
        return self.vocab

    def _word(self, w: str) -> int:
        if w in self.tok2id:
            return self.tok2id[w]
        h = 2166136261
        for ch in w.encode():
            h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
        return 3 + h % (self.base - 3)

    def __call__(self, text: str) -> List[int]:
        import re
        words = re.findall(r"<\d+>|<timestamp_grounding>|\S+", text)
        ids = [self._word(w) for w in words]
        return ([self.bos_token_id] if self.bos_token_id is not None else []) + ids

    def batch_decode(self, batch: Sequence[Sequence[int]], skip_special_tokens: bool = True) -> List[str]:
        out = []
        for ids in batch:
            toks = []
            for i in ids:
                if skip_special_tokens and i in (self.bos_token_id, self.eos_token_id, self.pad_token_id):
                    continue
                toks.append(self.id2tok.get(int(i), f"w{int(i)}"))
            out.append(" ".join(toks))
        return out


def fit_geometry(geometry: TowerGeometry, llm: str, num_frames: int, num_segs: int, max_txt_len: int) -> TowerGeometry:
    """A copy of `geometry` whose limits hold the reference's configuration (frames / segments / max_txt_len): the prefill workspace and
    the RoPE tables must cover the visual prefix plus the longest prompt the reference accepts (llava_next_video.py:622-647).  Shared by
    the model constructor and tools/pack_checkpoint.py, so a packed file's RoPE tables have the max_seq the loader will ask for.
    The paged KV pool is sized from the free HBM (kv_pages = 0) only when the caller left kv_pages at the dataclass default or passed 0;
    an explicit pool that cannot hold one full-length sequence is an error, never silently replaced by "all of the HBM"."""
    g = dataclasses.replace(geometry)
    g.frames_per_seg = num_frames // num_segs
    g.max_segs = max(g.max_segs, num_segs)
    tok_seg = (156 if llm == "phi3.5" else 64) + 16 * g.frames_per_seg + 1
    need = num_segs * tok_seg + max_txt_len
    g.max_prefill = max(g.max_prefill, need)
    g.max_seq = max(g.max_seq, min(need + 256, 131072))
    if g.kv_pages * 64 < need + 256:
        default_pages = TowerGeometry.__dataclass_fields__["kv_pages"].default
        if g.kv_pages in (0, default_pages):
            g.kv_pages = 0                            # 0 = size the paged KV pool from the free HBM (gvl_finalize_weights)
        else:
            raise ValueError(f"kv_pages = {g.kv_pages} ({g.kv_pages * 64} tokens) cannot hold one sequence of {need + 256} tokens "
                             f"({num_segs} segments x {tok_seg} visual tokens + max_txt_len {max_txt_len} + 256 new): pass a larger pool, "
                             "a smaller max_txt_len, or kv_pages = 0 to size the pool from the free HBM")
    return g


def packed_file_metadata(path: str) -> Dict[str, str]:
    """The __metadata__ dict of a `gvl-packed-1` safetensors file (8-byte little-endian header length + JSON header)."""
    import json
    import struct
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        hdr = json.loads(f.read(n))
    meta = hdr.get("__metadata__") or {}
    if meta.get("format") != "gvl-packed-1":
        raise ValueError(f"{path}: not a gvl packed weight file")
    return meta


class LLAVA_NEXT_VIDEO:
    def __init__(self, dtype=torch.bfloat16, stage="pretrain", max_txt_len=2048, num_frames=96, num_segs=12, lora=False,
                 num_temporal_tokens=300, llm="llama3", attn_implementation="flash_attention_2",
                 config_path="weight_path/Phi-3.5-vision-instruct", tokenizer_path="weight_path/Phi-3.5-mini-instruct",
                 pretrained_video_path="weight_path/internvideo/vision-encoder-InternVideo2-stage2_1b-224p-f4.pt",
                 pretrained_vision_proj_llm_path="weight_path/Phi-3.5-vision-instruct-seperated/",
                 *, geometry: Optional[TowerGeometry] = None, tokenizer=None, state_dicts: Optional[Dict[str, Dict[str, torch.Tensor]]] = None,
                 device: str = "cuda:0", group=None, packed_weights: Optional[str] = None, ckpt_path: Optional[str] = None,
                 exchange: str = "torch"):
        if dtype not in (torch.bfloat16,):
            raise ValueError("the MI355X path computes in bfloat16 (the reference's recommended dtype, README.md:57)")
        if num_frames % num_segs != 0:
            raise ValueError("num_frames must be a multiple of num_segs (einops rearrange at models/llava_next_video.py:530)")
        self.dtype, self.stage, self.max_txt_len = dtype, stage, max_txt_len
        self.num_frames, self.num_segs, self.lora, self.num_temporal_tokens, self.llm = num_frames, num_segs, lora, num_temporal_tokens, llm
        self.group = group
        if exchange not in ("torch", "gvl"):
            raise ValueError("exchange: 'torch' (torch.distributed all_gather_into_tensor) or 'gvl' (libgvl's own RCCL communicator through the C ABI)")
        self.exchange = exchange         # the all-gather of the visual tokens when the segments are sharded over a process group (encode_images)
        if geometry is None:
            geometry = geometry_from_checkpoint_dirs(llm, config_path, pretrained_vision_proj_llm_path, stage, num_temporal_tokens)
        geometry = fit_geometry(geometry, llm, num_frames, num_segs, max_txt_len)     # a COPY: the caller's object is never edited
        if llm == "phi3.5" and geometry.rope_short is None and geometry.rope_orig_max_pos > 0:
            raise ValueError("Phi-3.5 needs its LongRoPE short_factor / long_factor (config.json rope_scaling): Phi3LongRoPEScaledRotaryEmbedding "
                             "applies the short factors and the sqrt(1 + ln(s)/ln(4096)) scale even below 4096 tokens (modeling_phi3.py:380-409); "
                             "pass geometry=TowerGeometry().apply_hf_config(cfg) or a config_path that holds config.json")
        self.geo = geometry
        self.tokenizer = tokenizer
        if self.tokenizer is None:
            try:
                from transformers import AutoTokenizer
                self.tokenizer = AutoTokenizer.from_pretrained(tokenizer_path)
            except Exception as e:   # no network / no files: the caller must inject one
                raise RuntimeError(f"cannot load a tokenizer from {tokenizer_path!r} ({e}); pass tokenizer=...") from e
            if llm == "llama3":
                self.tokenizer.eos_token_id, self.tokenizer.pad_token_id = 128009, 128001   # llava_next_video.py:103-104
            elif llm == "phi3.5":
                self.tokenizer.pad_token = "<|end|>"                                       # :114
            if stage in ("grounded", "sft"):
                self.tokenizer.add_tokens(P.temporal_token_strings(num_temporal_tokens))   # :235-236
        self.engine = Engine(geometry, device)
        self._base_sd = None
        if packed_weights is not None:                    # file written by tools/pack_checkpoint.py: no per-start packing / LoRA merge;
            meta = packed_file_metadata(packed_weights)
            for key, have in (("max_seq", geometry.max_seq), ("frames_per_seg", geometry.frames_per_seg)):
                if key in meta and int(meta[key]) != int(have):
                    raise ValueError(f"{packed_weights}: packed for {key} = {meta[key]} but this model needs {key} = {have} (num_frames / num_segs / "
                                     "max_txt_len differ from the ones given to tools/pack_checkpoint.py: re-pack, or pass the same values)")
            self.engine.load_packed_file(packed_weights)  # read by libgvl itself (gvl_load_packed: mmap + one upload per tensor)
            self.engine.finalize()
            return
        if state_dicts is None:
            state_dicts = load_reference_checkpoints(llm, pretrained_video_path, pretrained_vision_proj_llm_path)
        self._base_sd = state_dicts
        if ckpt_path is not None:                         # inference.py:156-162 in one pass: base + fine-tuned overlay, packed ONCE
            ckpt = torch.load(ckpt_path, map_location="cpu")
            self.load_ckpt(ckpt.get("model", ckpt))
        else:
            self.load_state_dicts(state_dicts)

    # weights -------------------------------------------------------------------------------------------
    def load_state_dicts(self, sd: Dict[str, Dict[str, torch.Tensor]], ckpt_frames: Optional[int] = None):
        g = self.geo
        lm = sd["language_model"]
        if self.stage in ("grounded", "sft"):
            # the constructor of the reference grows the vocabulary BEFORE any fine-tuned weights arrive (reset_embeddings, :231-268)
            ek = next(k for k in lm if k.endswith("embed_tokens.weight"))
            n_new = g.vocab - lm[ek].shape[0]
            if n_new > 0:
                lm = Wt.reset_embeddings(lm, n_new, g.lm_head_bias)
        self.engine.load_packed(Wt.pack_clip(sd["vision_tower"], g.clip_layers - 1))
        self.engine.load_packed(Wt.pack_iv2(sd["video_encoder"], g.iv2_depth - 1, g.frames_per_seg, ckpt_frames,
                                            tokens_per_frame=(g.iv2_image // g.iv2_patch) ** 2))
        self.engine.load_packed(Wt.pack_projectors(sd["projectors"], self.llm))
        self.engine.load_packed(Wt.pack_llm(lm, g.kind, g.layers, g.heads, g.kv_heads, g.max_seq, g.rope_theta,
                                            g.rope_short, g.rope_long, g.rope_max_pos, g.rope_orig_max_pos))
        self.engine.finalize()

    def load_ckpt(self, ckpt: Dict[str, Dict[str, torch.Tensor]], base: Optional[Dict[str, Dict[str, torch.Tensor]]] = None):
        """inference.py:156-162: overlay the fine-tuned groups {multi_modal_projector, video_projecter, language_model} on the base
        state dicts this object was built from (kept in memory: nothing is read from disk twice)."""
        base = base if base is not None else self._base_sd
        if base is None:
            raise RuntimeError("load_ckpt: this model was started from a packed weight file, which holds no base state dicts to overlay; "
                               "pass base=..., or overlay the checkpoint when packing (tools/pack_checkpoint.py --ckpt_path)")
        proj = dict(base["projectors"])
        for grp in ("multi_modal_projector", "video_projecter"):
            for k, v in ckpt.get(grp, {}).items():
                proj[f"{grp}.{k}"] = v
        sd = dict(base)
        sd["projectors"] = proj
        if "language_model" in ckpt:
            sd["language_model"] = ckpt["language_model"]
        self.load_state_dicts(sd)

    # text plumbing ----------------------------------------------------------------------------------------
    def tokenizer_image_token(self, prompt: str) -> List[int]:
        tok = (lambda s: self.tokenizer(s).input_ids) if hasattr(self.tokenizer("x"), "input_ids") else self.tokenizer
        return P.tokenize_with_image(prompt, tok, getattr(self.tokenizer, "bos_token_id", None))

    # vision -------------------------------------------------------------------------------------------------
    def encode_images(self, samples) -> torch.Tensor:
        """[bs, num_segs*L, hidden] bf16.  With a process group, segments are sharded over the ranks and the
        token blocks all-gathered (dist.py)."""
        sp, tp = samples["spatial_pixel_values"], samples["temporal_pixel_values"]
        bs, S = sp.shape[:2]
        fps = tp.shape[1] // S
        sp = sp.reshape(bs * S, *sp.shape[2:])
        tp = tp.reshape(bs, S, fps, *tp.shape[2:]).permute(0, 1, 3, 2, 4, 5).reshape(bs * S, tp.shape[2], fps, *tp.shape[3:])
        n = bs * S
        L = self.engine.tokens_per_seg
        if self.group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1):
            world, rank = torch.distributed.get_world_size(self.group), torch.distributed.get_rank(self.group)
            lo, hi = gdist.my_shard(n, rank, world)
            ms = max(1, self.geo.max_segs)
            parts = [self.engine.encode_segments(sp[i:min(i + ms, hi)], tp[i:min(i + ms, hi)]) for i in range(lo, hi, ms)]   # bs > 1: the shard may exceed max_segs
            local = (torch.cat(parts, 0) if len(parts) > 1 else parts[0]) if parts else torch.empty((0, self.geo.hidden), dtype=bf, device=self.engine.device)
            # the ONE collective of the sharded plan: torch.distributed's all_gather_into_tensor (RCCL), or -- exchange="gvl" -- libgvl's own
            # communicator through the C ABI (gvl_comm_init / gvl_allgather_visual), what a non-Python host of the library calls
            if self.exchange == "gvl" and getattr(self.engine, "comm_world", 0) != world:
                gdist.init_gvl_comm(self.engine, self.group)
            vis = gdist.allgather_visual(local, n, L, self.group, gatherv=self.engine.allgatherv_visual if self.exchange == "gvl" else None)
        else:
            ms = self.geo.max_segs
            if bs > 1 and S <= ms:
                # several samples: the CLIP tower takes as many key frames per call as the workspace allows (its GEMMs are small:
                # +1.6 % clips/s in bench.py), InternVideo2 + projectors run per sample (batching them further is slower).
                # Every kernel is batch-invariant, so the tokens are bit-identical to per-sample encodes.
                cf = torch.cat([self.engine.clip_encode(sp[i:i + ms]) for i in range(0, n, ms)], 0)
                chunks = [self.engine.build_visual(cf[i:i + S], self.engine.iv2_encode(tp[i:i + S])) for i in range(0, n, S)]
            else:
                chunks = [self.engine.encode_segments(sp[i:i + ms], tp[i:i + ms]) for i in range(0, n, ms)]
            vis = torch.cat(chunks, 0) if len(chunks) > 1 else chunks[0]
        return vis.view(bs, S * L, self.geo.hidden)

    # generate -----------------------------------------------------------------------------------------------
    def _select_tokens(self, kw):
        """HF generate's token selection for the kwargs the reference forwards (inference.py:170-176 -> llava_next_video.py:655-661):
        greedy, or temperature -> top-k (HF default 50) -> top-p sampling on the device; `seed` (extra) makes a run reproducible,
        otherwise every call draws a fresh seed from torch's CPU generator (so torch.manual_seed governs it, as it does HF's)."""
        if not kw.get("do_sample", False) or kw.get("num_beams", 1) not in (1, None):
            # greedy -- and every kind of beam search: the steps return logits, the selection (top-2k, or the beam-sample draw) lives in beam.py
            self.engine.set_sampling(False)
            return
        t = kw.get("temperature", 1.0)
        t = 1.0 if t is None else float(t)
        if not t > 0:
            raise ValueError("`temperature` has to be a strictly positive float")      # HF's TemperatureLogitsWarper check
        top_p = kw.get("top_p")
        if top_p is not None and not (0 < float(top_p) <= 1.0):
            raise ValueError("`top_p` has to be a float > 0 and <= 1")
        top_k = kw.get("top_k", 50)
        seed = kw.get("seed")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.engine.set_sampling(True, t, 0 if top_k is None else int(top_k), top_p, seed)

    @torch.inference_mode()
    def generate(self, samples, **generate_kwargs) -> List[str]:
        self._select_tokens(generate_kwargs)
        max_new = int(generate_kwargs.get("max_new_tokens", 2048))
        if any(v == "text" for v in samples.get("video_ids", [])):
            # prepare_multimodal_inputs' `video_ids == 'text'` branch (llava_next_video.py:583-586) is a TRAINING device (dummy visual
            # rows appended with mask 0 so FSDP sees every parameter); the reference's inference never produces it.  forward() handles it.
            raise ValueError("generate(): 'text' samples are a training-only construct of the reference; use forward() for them")
        ids = [self.tokenizer_image_token(t) for t in samples["prompts"]]
        pad_id = getattr(self.tokenizer, "pad_token_id", 0) or 0
        ids_arr, mask = P.left_pad_truncate(ids, pad_id, self.max_txt_len)
        feats = self.encode_images(samples)
        k = generate_kwargs.get("num_beams", 1) or 1
        if k > 1:                                         # HF beam search (do_sample=False), one sample at a time
            sample = None
            if generate_kwargs.get("do_sample", False):      # beam-sample (HF _beam_sample): the warpers' arguments as generate() takes them; one generator per call
                t = generate_kwargs.get("temperature", 1.0)
                t = 1.0 if t is None else float(t)
                if not t > 0:
                    raise ValueError("`temperature` has to be a strictly positive float")
                top_p = generate_kwargs.get("top_p")
                if top_p is not None and not (0 < float(top_p) <= 1.0):
                    raise ValueError("`top_p` has to be a float > 0 and <= 1")
                seed = generate_kwargs.get("seed")
                gen = torch.Generator(device=self.engine.device)
                gen.manual_seed(int(seed) if seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item()))
                sample = dict(temperature=t, top_k=generate_kwargs.get("top_k", 50), top_p=top_p, generator=gen)
            out_ids = [self.beam_generate_ids([int(t) for t, m in zip(ids_arr[b], mask[b]) if m], feats[b], k, max_new,
                                              float(generate_kwargs.get("length_penalty", 1.0)), generate_kwargs.get("early_stopping", False), sample)
                       for b in range(ids_arr.shape[0])]
        else:
            out_ids = self.generate_ids(ids_arr, mask, feats, max_new)
        texts = self.tokenizer.batch_decode(out_ids, skip_special_tokens=True)
        return [t.strip() for t in texts]

    def beam_generate_ids(self, row: List[int], vis: torch.Tensor, num_beams: int, max_new: int, length_penalty: float = 1.0, early_stopping=False,
                          sample: Optional[dict] = None) -> List[int]:
        """generate(num_beams = k, do_sample = False): HF beam search (beam.py restates transformers 4.40.1's scorer) on the paged KV cache.  The k running
        beams are k sequences; HF's per-step cache reorder becomes gvl_seq_clone -- a beam that continues another one shares its whole KV pages by
        reference and copies only the partial last page; the first child of a parent simply keeps the parent's sequence.  All beams advance by ONE
        teacher-forced batched decode step per token (gvl_decode_step_logits_batch: one stream of the weights for the k beams); log-softmax / top-2k of the step run on the device (torch), the bookkeeping on
        the host."""
        from . import beam as B
        eng = self.engine
        gi = eng.decode_group_info()                     # which group sizes ONE batched step takes: asked from the library, not restated here
        eos = getattr(self.tokenizer, "eos_token_id", None)
        emb = eng.splice(row, vis)
        cap = min(emb.shape[0] + max_new + 1, self.geo.max_seq)
        beams: List[Optional[int]] = [eng.seq_alloc(cap)]
        fresh: List[int] = []                            # clones of the step in progress: owned here until they are installed in `beams`
        try:
            first = eng.prefill(beams[0], emb, want_logits=True)

            def step(parents: List[int], toks: List[int]) -> torch.Tensor:
                keep, new = {}, [None] * len(parents)
                for j, p_ in enumerate(parents):             # clones first: every parent is still at the length the children continue from
                    if p_ in keep:
                        new[j] = eng.seq_clone(beams[p_], cap)
                        fresh.append(new[j])                 # if a later clone raises (pool exhausted, kMaxSeqs), `finally` frees these
                    else:
                        keep[p_] = j
                for p_, j in keep.items():
                    new[j] = beams[p_]
                losers = [s_ for p_, s_ in enumerate(beams) if p_ not in keep]
                beams[:] = new                               # install before freeing: `finally` never sees an id twice or a freed id
                del fresh[:]
                for s_ in losers:
                    eng.seq_free(s_)
                if len(beams) <= gi["max_group"] and (gi["any_size"] or len(beams) in (1, 2, 4)):
                    return eng.decode_step_logits_batch(beams, toks)        # the k beams share ONE stream of the weights
                return torch.stack([eng.decode_step_logits(s_, t) for s_, t in zip(beams, toks)])

            return B.beam_search(step, first, num_beams, max_new, eos, length_penalty, early_stopping, sample)
        finally:
            for s_ in set(x for x in list(beams) + fresh if x is not None):
                try:
                    eng.seq_free(s_)
                except Exception:                            # never mask the original error with a cleanup error
                    pass

    @torch.inference_mode()
    def generate_shared(self, samples, prompts: Sequence[str], **generate_kwargs) -> List[str]:
        """Several prompts about ONE video.  The reference's inference.py calls generate() once per prompt (grounding / QA /
        referring, inference.py:178-182) and re-runs both vision towers every time; here the video is encoded once and the prompts
        are prefilled and decoded together.  Texts are identical to one generate() call per prompt (batch-invariant kernels)."""
        self._select_tokens(generate_kwargs)
        if samples["spatial_pixel_values"].shape[0] != 1:
            raise ValueError("generate_shared takes the pixel tensors of one video")
        max_new = int(generate_kwargs.get("max_new_tokens", 2048))
        ids = [self.tokenizer_image_token(t) for t in prompts]
        pad_id = getattr(self.tokenizer, "pad_token_id", 0) or 0
        ids_arr, mask = P.left_pad_truncate(ids, pad_id, self.max_txt_len)
        feats = self.encode_images(samples)
        rows = [[int(t) for t, m in zip(ids_arr[i], mask[i]) if m] for i in range(len(prompts))]
        out_ids = self._generate_shared_prefix(rows, feats[0], max_new)
        if out_ids is None:                              # nothing worth sharing (one prompt, or the prompts part ways before 128 tokens)
            out_ids = self.generate_ids(ids_arr, mask, feats.expand(len(prompts), -1, -1), max_new)
        return [t.strip() for t in self.tokenizer.batch_decode(out_ids, skip_special_tokens=True)]

    def _generate_shared_prefix(self, rows: List[List[int]], vis: torch.Tensor, max_new: int) -> Optional[List[List[int]]]:
        """Prompts about one video share the system prompt and the visual tokens: the common prefix (rounded down to 128 tokens = whole KV pages AND
        whole query blocks, so that every later row is computed exactly as in a full prefill) is prefilled ONCE; every prompt forks it
        (gvl_seq_fork: pages referenced, not copied) and prefills only its own tail (gvl_prefill_extend); the answers are decoded together.
        Ids are bit-identical to one full prefill per prompt."""
        eng = self.engine
        self.last_shared_prefix = 0                      # tokens prefilled once for all prompts in the last generate_shared call (0 = not shared)
        if len(rows) < 2:
            return None
        embs = [eng.splice(r, vis) for r in rows]
        # common prefix in EMBEDDING rows: ids before the image slot must agree, then the visual rows, then the ids after it
        k = rows[0].index(P.IMAGE_TOKEN_INDEX)
        if any(r[:k + 1] != rows[0][:k + 1] for r in rows):
            return None
        n_vis = vis.shape[0]
        tail = 0
        while all(k + 1 + tail < len(r) for r in rows) and len({r[k + 1 + tail] for r in rows}) == 1:
            tail += 1
        shared = min(k + n_vis + tail, min(e.shape[0] for e in embs) - 1)     # every prompt keeps at least one row of its own (its last row feeds the first token)
        prefix = shared // 128 * 128
        if prefix < 128:
            return None
        # LongRoPE (modeling_phi3.py:381-385) picks ONE factor set per forward from the TOTAL length: a full prefill of a prompt longer than
        # original_max_position_embeddings ropes every row -- the shared ones too -- with the long factors, while a base prefill of a prefix
        # that still fits the original context would cache its K with the short ones.  Share only when base and forks agree.
        omax = self.geo.rope_orig_max_pos if self.geo.rope_long is not None else 0
        if omax > 0 and prefix <= omax and any(e.shape[0] > omax for e in embs):
            return None
        self.last_shared_prefix = prefix
        eos = getattr(self.tokenizer, "eos_token_id", None)
        base, seqs = None, []
        try:
            base = eng.seq_alloc(prefix)
            eng.prefill(base, embs[0][:prefix])
            for e in embs:
                seqs.append(eng.seq_fork(base, prefix, min(e.shape[0] + max_new, self.geo.max_seq)))
                eng.prefill_extend(seqs[-1], e[prefix:])
            return eng.decode_greedy_batch(seqs, max_new, eos)
        finally:
            for s_ in seqs:
                eng.seq_free(s_)
            if base is not None:
                eng.seq_free(base)

    # training forward --------------------------------------------------------------------------
    @torch.inference_mode()
    def forward(self, samples) -> Dict[str, torch.Tensor]:
        """LLAVA_NEXT_VIDEO.forward(samples) -> {"loss"} (models/llava_next_video.py:598-614), forward only: prepare_batch
        (labels / right padding / truncation), encode_images, label + mask splice, then the causal-LM loss of the language model.
        The padded batch is never materialised: the masked rows of every sample sit at its END (right padding; the dummy visual
        rows of a 'text' sample), so dropping them leaves positions and causal attention of the kept rows unchanged, and
        CrossEntropyLoss(mean) over the flattened batch == sum of token losses / number of labelled tokens over the samples."""
        tk = self.tokenizer
        tok = (lambda s: tk(s).input_ids) if hasattr(tk("x"), "input_ids") else tk
        ids, labels, mask = P.prepare_batch(self.llm, samples["text_inputs"], tok, getattr(tk, "bos_token_id", None),
                                            tk.pad_token_id, tk.eos_token_id, self.max_txt_len)
        feats = self.encode_images(samples)
        total, count = 0.0, 0
        for b, vid in enumerate(samples["video_ids"]):
            is_text = vid == "text"
            ml, mm = P.splice_labels(ids[b], labels[b], mask[b], feats.shape[1], is_text)
            n = int(mm.sum())
            assert bool((mm[:n] == 1).all()), "attention mask is not a prefix of ones"
            if bool((ml[n:] != P.IGNORE_INDEX).any()) and not getattr(self, "_warned_pad_label", False):
                # reference quirk: after truncation `batch_labels[:, -1] = eos` also labels the PAD slot of shorter rows, whose
                # logits come from a masked pad row; that term is dropped here (it trains nothing but the pad embedding)
                print("WARNING: label on a masked (padding) position ignored")
                self._warned_pad_label = True
            row = [int(t) for t, m in zip(ids[b], mask[b]) if m]
            emb = self.engine.splice(row, feats[b][:0] if is_text else feats[b])
            assert emb.shape[0] == n
            s, c = self.engine.forward_loss(emb, ml[:n].tolist())
            total, count = total + s, count + c
        loss = total / count if count else float("nan")         # CrossEntropyLoss over zero targets is nan in torch too
        return {"loss": torch.tensor(loss, dtype=torch.float32, device=self.engine.device)}

    __call__ = forward

    def generate_ids(self, ids_arr, mask, feats, max_new: int) -> List[List[int]]:
        eos = getattr(self.tokenizer, "eos_token_id", None)
        eng = self.engine
        if ids_arr.shape[0] == 1:
            row = [int(t) for t, m in zip(ids_arr[0], mask[0]) if m]
            return [eng.generate_ids(eng.splice(row, feats[0]), max_new, eos)]
        # bs > 1 (the reference left-pads the batch, llava_next_video.py:622-647): every sample keeps its own paged KV and its
        # un-padded length -- identical maths to the masked left-padded batch.  Prefill runs over the packed rows of the batch
        # (gvl_prefill_varlen) and the greedy decode of the whole batch runs together (gvl_decode_greedy_batch: one weight stream
        # per token for groups of up to 16 sequences)
        # The KV pool bounds how many samples are resident at once: the batch is processed in as many groups as it takes.
        from .lib import GvlError, ERR_OOM
        out: List[List[int]] = []
        b, n = 0, ids_arr.shape[0]
        while b < n:
            seqs, embs = [], []
            try:
                while b + len(seqs) < n:
                    i = b + len(seqs)
                    row = [int(t) for t, m in zip(ids_arr[i], mask[i]) if m]
                    emb = eng.splice(row, feats[i])
                    try:
                        seqs.append(eng.seq_alloc(min(emb.shape[0] + max_new, self.geo.max_seq)))
                    except GvlError as e:
                        if e.status == ERR_OOM and seqs:
                            break                      # pool full: run what fits, the rest in the next group
                        raise
                    embs.append(emb)
                eng.prefill_batch(seqs, embs)
                out += eng.decode_greedy_batch(seqs, max_new, eos)
            finally:
                for seq in seqs:
                    eng.seq_free(seq)
            b += len(seqs)
        return out


BASE_VOCAB = {"phi3.5": 32064, "llama3": 128256, "vicuna": 32000}     # tokenizer sizes before reset_embeddings [ext]


def geometry_from_checkpoint_dirs(llm: str, config_path: Optional[str], pretrained_vision_proj_llm_path: Optional[str],
                                  stage: str = "sft", num_temporal_tokens: int = 300) -> TowerGeometry:
    """Default geometry of an LLM family, completed from the HF config.json next to the weights: the reference builds its language
    model from `<pretrained_vision_proj_llm_path>/language_model_seperated` (models/llava_next_video.py:149-151), whose config.json
    carries rope_scaling.{short,long}_factor, both context limits, rope_theta, rms_norm_eps and the base vocab_size.
    Vocabulary: stages 'grounded' / 'sft' run reset_embeddings (:153-154, :231-268): num_temporal_tokens + 2 new rows and an lm_head
    WITH bias; stage 'pretrain' keeps the base vocabulary and the bias-free lm_head."""
    import json
    geo = {"phi3.5": TowerGeometry, "llama3": TowerGeometry.llama3_8b, "vicuna": TowerGeometry.vicuna_7b}[llm]()
    geo.kv_pages = 0                   # production default: the paged KV pool takes the HBM left once the weights are resident
    base_vocab = BASE_VOCAB[llm]
    cands = []
    if pretrained_vision_proj_llm_path:
        cands.append(os.path.join(pretrained_vision_proj_llm_path, "language_model_seperated", "config.json"))
    if config_path:
        cands.append(os.path.join(config_path, "config.json"))
    for c in cands:
        if os.path.exists(c):
            with open(c) as f:
                cfg = json.load(f)
            geo.apply_hf_config(cfg)
            base_vocab = int(cfg.get("text_config", cfg).get("vocab_size", base_vocab))
            break
    grown = stage in ("grounded", "sft")
    geo.vocab = base_vocab + (num_temporal_tokens + 2 if grown else 0)
    geo.lm_head_bias = grown
    return geo


def load_reference_checkpoints(llm: str, pretrained_video_path: str, pretrained_vision_proj_llm_path: str):
    """The reference's on-disk layout (models/llava_next_video.py:117-151): vision_model.pth, image_newline(s).pth,
    multi_modal_projector.pth, language_model_seperated/ (HF safetensors), InternVideo2 .pt."""
    d = pretrained_vision_proj_llm_path
    need = [os.path.join(d, "vision_model.pth"), pretrained_video_path]
    missing = [p for p in need if not os.path.exists(p)]
    if missing:
        raise FileNotFoundError(f"reference checkpoints not found: {missing}; pass state_dicts=... (e.g. synthetic weights from grounded_video_llm_amd.synth)")
    sd = {"vision_tower": torch.load(os.path.join(d, "vision_model.pth"), map_location="cpu"),
          "video_encoder": torch.load(pretrained_video_path, map_location="cpu")}
    proj = {f"multi_modal_projector.{k}": v for k, v in torch.load(os.path.join(d, "multi_modal_projector.pth"), map_location="cpu").items()}
    if llm == "phi3.5":
        nl = torch.load(os.path.join(d, "image_newlines.pth"), map_location="cpu")
        proj["glb_GN"], proj["sub_GN"] = nl["glb_GN"], nl["sub_GN"]
    else:
        proj["image_newline"] = torch.load(os.path.join(d, "image_newline.pth"), map_location="cpu")["image_newline"]
    sd["projectors"] = proj
    from safetensors.torch import load_file
    lm = {}
    lmdir = os.path.join(d, "language_model_seperated")
    for f in sorted(os.listdir(lmdir)):
        if f.endswith(".safetensors"):
            lm.update(load_file(os.path.join(lmdir, f)))
    sd["language_model"] = lm
    if "video_projecter.up_proj.weight" not in proj:   # trained group: arrives with the fine-tuned ckpt (inference.py:159-160)
        emb = next(v for k, v in lm.items() if k.endswith("embed_tokens.weight"))
        hid, vd = emb.shape[1], sd["video_encoder"]["cls_token"].shape[-1]
        proj.update({"video_projecter.up_proj.weight": torch.zeros(hid, vd), "video_projecter.up_proj.bias": torch.zeros(hid),
                     "video_projecter.down_proj.weight": torch.zeros(hid, hid), "video_projecter.down_proj.bias": torch.zeros(hid)})
    return sd
