"""bench.py -- the reference's headline metric on MI355X: end-to-end clips/s (and grounding-answer
tokens/s) for the 96-frame Phi-3.5-3.8B configuration (BASELINE.json configs[1]).

A "step" = one pass of the hot path over one batch of synthetic input: for every rank `--clips-per-step` (default 8)
96-frame clips, each (12 x 336^2 spatial frames + 96 x 224^2 temporal frames, already resident in HBM) -> CLIP ViT-L/14-336
(23 layers) + InternVideo2-1B (39 blocks) -> merge/pool + projectors -> 3420 visual tokens spliced into a prompt of 60..140 tokens
-> Phi-3.5 prefill (S ~ 3480..3560, the clips of a step packed into ONE ragged pass: gvl_prefill_varlen) -> greedy decode of 12 new
tokens through the paged KV cache, the clips of a step decoded TOGETHER at their different lengths (gvl_decode_greedy_batch: every
weight matrix is streamed once per token for all of them -- the reference batches clips in generate() too,
llava_next_video.py:622-647), while the vision encode of the next step's clips runs on a second stream (the CLIP tower once over the
12 x cps key frames of the step, then InternVideo2 + projectors per clip).  EVERY clip of a step has its own pixels and its own
prompt (ids and length), and consecutive steps use different clips (a resident pool of 2 x cps seeded clips is cycled).
Random-init weights of the real architecture, synthetic pixels (no network for checkpoints / datasets).

N > 1 (one process per GPU, RCCL), two plans, both reported:
  * `value` (weak scaling): N x cps clips in flight per step; every clip's 12-segment frame batch is sharded over ALL N ranks
    (rotated per clip so the 12 % N remainder balances: each rank encodes exactly 12 segments per clip round), ONE all-gather of the
    visual tokens over xGMI per step, then rank c runs the LLM for clip c.
  * `single_clip_latency_ms_sharded` (the north-star plan, strong scaling): ONE clip, its 12 segments sharded over the N ranks,
    all-gather of the visual tokens, prefill + decode on rank 0 (llava_next_video.py:503-505,530-532,563 are the independent
    segments that make the shard).
A watchdog thread turns a hung collective into a JSON line that names the stage every rank was in (`hang`), instead of a silent
time-out; `n_ranks_seen_by_rccl` is ncclCommCount of a communicator libgvl itself builds over all ranks (gvl_comm_init).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (the bf16 MFMA GEMM: gemm_a4p / gemm_a4 / gemm_pp / gemm_bf16 kernels), measured
live with HIP event pairs around every GEMM launch of ONE extra profiled step of the timed kind (8 clips: the launches back to back on one stream;
`roofline.gemm_ms_per_clip`, `launches_in_profiled_pass`); the one-clip-alone figure of rounds 1-2 is kept under `roofline.one_clip_serial`.
`roofline.achieved` = 77.59 TFLOP algorithmic GEMM work per clip (78.1 un-padded 2 M N K, minus the 0.53 of the last decoder layer's MLP that a
prefill without a loss request does not execute) / that time.  `cpu_baseline` is the CPU oracle timed on a bounded sample on the host cores (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import _gvl_bootstrap  # noqa: E402,F401
from grounded_video_llm_amd import dist as gdist, engine as E, lib as L, synth, weights as Wt  # noqa: E402

bf = torch.bfloat16
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
PROMPT_LEN_RANGE = (60, 140)   # text tokens of a clip's prompt (the reference's grounding / QA templates land in this range)


# ---- progress + watchdog -------------------------------------------------------------------------------------------------------
class Progress:
    """What this rank is doing right now (stage name + when it started).  The watchdog thread reads it: if a stage lasts longer than
    `limit` seconds (a hung collective, a dead peer), rank 0 still prints the ONE JSON line -- with `hang` = the stage of this rank and
    whatever per-rank stage times were gathered so far -- and every rank says on stderr where it stood; then the process exits."""

    def __init__(self, rank: int, world: int, limit_s: float, args_echo: dict):
        self.rank, self.world, self.limit = rank, world, limit_s
        self.stage, self.t0, self.done = "init", time.monotonic(), False
        self.partial = {"metric": "clips/sec + grounding tokens/sec, 96-frame Phi3.5-3.8B @1/2/4/8 MI355X", "value": None, "unit": "clips/s",
                        "n_gpus": world, "higher_is_better": True, **args_echo}
        self.thread = None

    def enter(self, stage: str):
        self.stage, self.t0 = stage, time.monotonic()

    def start(self):
        if self.limit <= 0:
            return
        self.thread = threading.Thread(target=self._watch, daemon=True)
        self.thread.start()

    def _watch(self):
        while not self.done:
            time.sleep(min(5.0, max(0.05, self.limit / 4)))
            stuck = time.monotonic() - self.t0
            if not self.done and stuck > self.limit:
                msg = {"rank": self.rank, "stage": self.stage, "seconds_in_stage": round(stuck, 1)}
                print(f"bench: WATCHDOG rank {self.rank}/{self.world} stuck in stage '{self.stage}' for {stuck:.0f} s", file=sys.stderr, flush=True)
                if self.rank == 0:
                    print(json.dumps({**self.partial, "hang": msg}), flush=True)
                os._exit(3)

    def finish(self):
        self.done = True


# ---- device shim: the N > 1 plumbing test runs this file on CPU (gloo, stub engine) --------------------------------------------
class _NullStream:
    def wait_event(self, ev): pass
    def synchronize(self): pass


class _NullEvent:
    def __init__(self, enable_timing=False): self.t = 0.0
    def record(self, stream=None): self.t = time.perf_counter()
    def elapsed_time(self, other): return 1e3 * (other.t - self.t)


class Dev:
    def __init__(self, dev: torch.device):
        self.dev, self.gpu = dev, dev.type == "cuda"

    def sync(self):
        if self.gpu:
            torch.cuda.synchronize()

    def stream(self):
        return torch.cuda.Stream(self.dev) if self.gpu else _NullStream()

    def event(self, timing=False):
        return torch.cuda.Event(enable_timing=timing) if self.gpu else _NullEvent()

    def on(self, stream):
        return torch.cuda.stream(stream) if self.gpu else contextlib.nullcontext()

    def current_stream_sync(self):
        if self.gpu:
            torch.cuda.current_stream().synchronize()

    def generator(self, seed):
        g = torch.Generator(device=self.dev); g.manual_seed(seed)
        return g


# segments per clip: 12 = the 96-frame configurations (BASELINE configs[1], [2]); GVL_BENCH_SEGS=32 = the 256-frame dense-captioning clip of configs[4] (the CPU plumbing
# tests run its 8-rank split; every published line is a 12-segment line and says so in config.workload)
SEGS = int(os.environ.get("GVL_BENCH_SEGS", "12"))


def build_engine(dev, frames_per_seg=8, max_segs=12, new_tokens=12, clips_per_step=1, kv_pages=None):
    """kv_pages None: a pool that just holds the step's clips (tests share the GPU with other engines); 0: sized from the free HBM
    once the weights are resident (gvl_finalize_weights) -- what bench.py's main uses."""
    geo = E.TowerGeometry(llm="phi3.5", frames_per_seg=frames_per_seg, max_segs=max_segs, max_seq=4096, max_prefill=3712 * clips_per_step,
                          kv_pages=64 * clips_per_step if kv_pages is None else kv_pages)
    geo.rope_short, geo.rope_long = synth.longrope_factors(96)
    eng = E.Engine(geo, dev)
    d = str(dev)
    W = synth.clip_weights(seed="bench.clip", device=d)
    eng.load_packed(Wt.pack_clip(W, geo.clip_layers - 1)); del W
    W = synth.iv2_weights(frames=frames_per_seg, seed="bench.iv2", device=d)
    eng.load_packed(Wt.pack_iv2(W, geo.iv2_depth - 1, frames_per_seg)); del W
    W = synth.projector_weights("phi3.5", seed="bench.proj", device=d)
    eng.load_packed(Wt.pack_projectors(W, "phi3.5")); del W
    W = synth.llm_weights("phi3", seed="bench.llm", device=d)
    eng.load_packed(Wt.pack_llm(W, "phi3", geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta, geo.rope_short, geo.rope_long)); del W
    torch.cuda.empty_cache()
    eng.finalize()
    return eng, geo


def make_prompt(seed: int, n_text: int = 0):
    """Token ids of one clip's prompt: n_text tokens (drawn from PROMPT_LEN_RANGE when 0) with the <image> slot (-200) after the
    36-token system prompt, as the reference's templates place it (datasets/chat/base_template.py)."""
    gi = torch.Generator(); gi.manual_seed(seed)
    if n_text <= 0:
        n_text = int(torch.randint(PROMPT_LEN_RANGE[0], PROMPT_LEN_RANGE[1] + 1, (1,), generator=gi).item())
    ids = torch.randint(3, 32000, (n_text,), generator=gi).tolist()
    ids[36] = -200
    return ids


def make_pixels(D: Dev, seed: int, n_segs=12, fps=8, hw=(336, 224)):
    g = D.generator(seed)
    sp = torch.randn((n_segs, 3, hw[0], hw[0]), device=D.dev, generator=g)
    tp = torch.randn((n_segs, 3, fps, hw[1], hw[1]), device=D.dev, generator=g)
    return sp, tp


def make_inputs(dev, rank, n_segs=12, fps=8, n_text=100):
    """One clip's pixels and a prompt of n_text ids (tests/test_gpu_fullsize.py): (spatial [n,3,336,336], temporal [n,3,fps,224,224], ids)."""
    sp, tp = make_pixels(Dev(torch.device(dev)), 42 + rank, n_segs, fps)
    return sp, tp, make_prompt(42, n_text)


class Stepper:
    """The stages of the hot path on one rank, over a resident POOL of distinct inputs.
    world == 1: pool entry i = clip i (12 segments) and its prompt.  world > 1: pool entry i = the 12 segment blocks this rank encodes in
    clip round i (rotated plan) -- every rank generates the pixels of the segments it encodes itself -- and the prompt of clip == rank."""

    def __init__(self, eng, geo, rank, world, new_tokens, pool=2, hw=(336, 224)):
        self.eng, self.geo, self.rank, self.world, self.new_tokens = eng, geo, rank, world, new_tokens
        self.D = Dev(torch.device(eng.device))
        self.dev = self.D.dev
        self.L = eng.tokens_per_seg
        self.decode_s = 0.0
        self.h2d = False
        self.pool = max(1, pool)
        self.cursor = 0
        fps = geo.frames_per_seg
        px = [make_pixels(self.D, 42 + 1009 * i + 7919 * rank, SEGS, fps, hw) for i in range(self.pool)]
        self.sp_pool = torch.cat([p[0] for p in px], 0)          # [pool * 12, 3, 336, 336]: a window of cps clips is one contiguous slice
        self.tp_pool = torch.cat([p[1] for p in px], 0)
        self.prompts = [make_prompt(77 + 31 * i + 7919 * rank) for i in range(self.pool)]
        if world > 1:
            # clip c's segment block b goes to rank (b + c) % world (dist.rotated_encode_plan): every rank encodes exactly 12 segments
            self.mine = gdist.rotated_encode_plan(SEGS, rank, world)
            self.gather = gdist.rotated_gather_index(SEGS, rank, world)
            assert sum(h - l for _, l, h in self.mine) == SEGS

    # ---- pool access -------------------------------------------------------------------------------------------
    def px(self, i):
        i %= self.pool
        return self.sp_pool[i * SEGS:(i + 1) * SEGS], self.tp_pool[i * SEGS:(i + 1) * SEGS]

    def window(self, cps):
        """pool indices of the next cps clips; the pool size is a multiple of cps, so the window is one contiguous slice"""
        start = self.cursor % self.pool
        self.cursor += cps
        assert start + cps <= self.pool
        return list(range(start, start + cps))

    # ---- the stages of one clip ---------------------------------------------------------------------------------------
    def _exchange(self, vis, force=False):
        return self._exchange_multi([vis], force)[0]

    def _exchange_multi(self, vis_list, force=False, all_clips=False):
        """N > 1: ONE all-gather per STEP: this rank's 12 segment blocks of each of the step's `cps` clip rounds travel together
        (cps x 3.5 MB per rank; xGMI is point-to-point, every rank pushes to its 7 peers at once), then each round picks the blocks
        of this rank's clip in segment order.  force: run the collective at world == 1 too (RCCL smoke test on a 1-GPU box)."""
        if self.world == 1 and not force:
            return vis_list
        cps, rows = len(vis_list), vis_list[0].shape[0]
        send = torch.cat(vis_list, 0) if cps > 1 else vis_list[0]
        recv = torch.empty((self.world * send.shape[0], send.shape[1]), dtype=bf, device=self.dev)
        recv = self._allgather(send, recv).view(self.world, cps, rows, -1)
        if all_clips:                                                  # the rank-0-LLM plan: every clip of the round, in clip order
            idx = [gdist.rotated_gather_index(SEGS, c, self.world) for c in range(self.world)] if self.world > 1 else [[(0, 0, rows // self.L)]]
            return [[torch.cat([recv[src, c, off * self.L:(off + n) * self.L] for src, off, n in g], 0) for g in idx] for c in range(cps)]
        gather = self.gather if self.world > 1 else [(0, 0, rows // self.L)]
        return [torch.cat([recv[src, c, off * self.L:(off + n) * self.L] for src, off, n in gather], 0) for c in range(cps)]   # clip == rank, segment order

    def _allgather(self, send, recv):
        """The collective of the exchange.  exchange == "gvl": libgvl's own RCCL communicator behind the C ABI (gvl_comm_init once,
        gvl_allgather_visual per step) -- what a non-Python host calls; "torch": torch.distributed.all_gather_into_tensor (RCCL; gloo on CPU
        tensors in the plumbing test)."""
        if self.exchange == "gvl":
            return self.eng.allgather_visual(send)
        if torch.distributed.get_backend() == "gloo" and send.device.type != "cpu":   # debug only (GVL_BENCH_BACKEND=gloo on a GPU): stage through the host
            rc = recv.cpu(); torch.distributed.all_gather_into_tensor(rc, send.cpu()); recv.copy_(rc)
        else:
            torch.distributed.all_gather_into_tensor(recv, send)
        return recv

    def encode(self, i):
        """vision towers + projectors for pool entry i (+ the all-gather for N > 1) -> visual tokens of THIS rank's clip."""
        sp, tp = self.px(i)
        if self.h2d:                                                    # extra (untimed for `value`): pixels arrive over PCIe
            k = i % self.pool
            sp.copy_(self.sp_host[k * SEGS:(k + 1) * SEGS], non_blocking=True)
            tp.copy_(self.tp_host[k * SEGS:(k + 1) * SEGS], non_blocking=True)
        return self._exchange(self.eng.encode_segments(sp, tp))   # [12*L, hidden]

    def encode_multi(self, idx):
        """Vision encode of the clips `idx` (a contiguous pool window).  The CLIP tower runs ONCE over the 12 x cps key frames of the step
        (per clip its N = 1024 GEMMs have only 112 tiles: +1.6 % clips/s measured); InternVideo2 + projectors (+ the all-gather) stay per
        clip.  GVL_BENCH_CLIP_BATCH=0 restores one gvl_encode_segments call per clip."""
        if not self.clip_batch:
            if self.h2d:
                return [self.encode(i) for i in idx]
            return self._exchange_multi([self.eng.encode_segments(*self.px(i)) for i in idx])
        lo, hi = idx[0] * SEGS, (idx[-1] + 1) * SEGS
        if self.h2d:
            self.sp_pool[lo:hi].copy_(self.sp_host[lo:hi], non_blocking=True)
            self.tp_pool[lo:hi].copy_(self.tp_host[lo:hi], non_blocking=True)
        cf = self.eng.clip_encode(self.sp_pool[lo:hi])
        # InternVideo2 over `iv2_batch` clips per call (default: the whole step -- M = 197 k rows: the GEMM rounds fill better, -3.9 % tower time
        # against one call per clip, tools/iv2_batch_lab.py; the towers are bit-exactly batch-invariant, and ids_match_serial re-checks it)
        nb = max(1, min(self.iv2_batch, len(idx)))
        if nb >= len(idx):                                              # the whole step in ONE call: no torch.cat (of one tensor it is a 554 MB device copy)
            vf = self.eng.iv2_encode(self.tp_pool[lo:hi])
        else:
            vf = torch.cat([self.eng.iv2_encode(self.tp_pool[(idx[c]) * SEGS:(idx[min(c + nb, len(idx)) - 1] + 1) * SEGS]) for c in range(0, len(idx), nb)], 0) if nb > 1 else None
        return self._exchange_multi([self.eng.build_visual(cf[c * SEGS:(c + 1) * SEGS], vf[c * SEGS:(c + 1) * SEGS] if vf is not None else self.eng.iv2_encode(self.px(i)[1]))
                                     for c, i in enumerate(idx)])

    def stage_times(self, prog=None):
        """One clip round on this rank, stages back to back with events between them (diagnosis of the N > 1 runs):
        vision encode of 12 segments / exchange / splice + prefill / greedy decode, in ms."""
        ev = [self.D.event(True) for _ in range(5)]
        self.D.sync()
        names = ("vision_ms", "exchange_ms", "prefill_ms", "decode_ms")
        if prog: prog.enter("stage_times:vision")
        ev[0].record()
        vis = self.eng.encode_segments(*self.px(0))
        ev[1].record(); self.D.sync()
        if prog: prog.enter("stage_times:exchange (all_gather_into_tensor)")
        vis = self._exchange(vis)
        ev[2].record(); self.D.sync()
        if prog: prog.enter("stage_times:prefill")
        seq, _ = self.llm(vis, 0)
        ev[3].record(); self.D.sync()
        if prog: prog.enter("stage_times:decode")
        self.decode(seq)
        ev[4].record()
        self.D.sync()
        return {n: round(ev[i].elapsed_time(ev[i + 1]), 3) for i, n in enumerate(names)}

    def llm(self, vis, i):
        eng = self.eng
        emb = eng.splice(self.prompts[i % self.pool], vis)
        seq = eng.seq_alloc(emb.shape[0] + self.new_tokens)
        eng.prefill(seq, emb)
        return seq, emb.shape[0]

    def decode(self, seq):
        if self.time_decode:
            self.D.current_stream_sync()
        t0 = time.perf_counter()
        out = self.eng.decode_greedy(seq, self.new_tokens, None)       # synchronises its stream
        if self.time_decode:
            self.decode_s += time.perf_counter() - t0
        self.eng.seq_free(seq)
        return out

    def step(self, i=None):
        """One clip, stages back to back on one stream (single-clip latency)."""
        if i is None:
            i = self.window(1)[0]
        vis = self.encode(i)
        seq, S = self.llm(vis, i)
        return self.decode(seq), S

    def step_multi_serial(self, idx):
        """The launches of ONE timed step (CLIP over the step's key frames, InternVideo2 + projectors per clip, one ragged prefill, one
        batched decode of the clips `idx`) back to back on the CURRENT stream: what the profiled pass brackets with events."""
        embs = [self.eng.splice(self.prompts[i % self.pool], vis) for vis, i in zip(self.encode_multi(idx), idx)]
        seqs = [self.eng.seq_alloc(e.shape[0] + self.new_tokens) for e in embs]
        self.eng.prefill_batch(seqs, embs)
        outs = self.eng.decode_greedy_batch(seqs, self.new_tokens, None)
        for seq in seqs:
            self.eng.seq_free(seq)
        return outs

    def step_sharded(self, i=0):
        """The north-star plan for ONE clip on N ranks: every rank encodes its contiguous share of the clip's 12 segments (the same clip
        on every rank: pixels from a rank-independent seed), ONE all-gather of the token blocks, prefill + greedy decode on rank 0
        (the other ranks wait at the caller's barrier).  Returns rank 0's ids."""
        if not hasattr(self, "_shared_px"):
            self._shared_px = make_pixels(self.D, 4242, SEGS, self.geo.frames_per_seg, (self.sp_pool.shape[-1], self.tp_pool.shape[-1]))
            self._shared_prompt = make_prompt(4242)
        sp, tp = self._shared_px
        lo, hi = gdist.my_shard(SEGS, self.rank, self.world)
        local = self.eng.encode_segments(sp[lo:hi], tp[lo:hi]) if hi > lo else torch.empty((0, self.geo.hidden), dtype=bf, device=self.dev)
        if self.world > 1:
            if self.exchange == "gvl":
                vis = gdist.allgather_visual(local, SEGS, self.L, gather=self.eng.allgather_visual)
            elif torch.distributed.get_backend() == "gloo" and local.device.type != "cpu":
                vis = gdist.allgather_visual(local.cpu(), SEGS, self.L).to(self.dev)
            else:
                vis = gdist.allgather_visual(local, SEGS, self.L)
        else:
            vis = local
        if self.rank != 0:
            return None
        emb = self.eng.splice(self._shared_prompt, vis)
        seq = self.eng.seq_alloc(emb.shape[0] + self.new_tokens)
        self.eng.prefill(seq, emb)
        out = self.eng.decode_greedy(seq, self.new_tokens, None)
        self.eng.seq_free(seq)
        return out

    # ---- the north-star plan as a THROUGHPUT mode: every rank encodes, rank 0 alone runs the LLM, pipelined ------------------------
    def r0_prompt(self, rnd, clip):
        """prompt of clip `clip` of pool round `rnd` -- what rank `clip` uses for it in the weak-scaling plan"""
        return make_prompt(77 + 31 * (rnd % self.pool) + 7919 * clip)

    def r0_start(self):
        self.sV, self.sL = self.D.stream(), self.D.stream()
        self.evV = self.D.event()
        with self.D.on(self.sV):
            self.r0_round = self.window(1)[0]
            self.r0_next = self._exchange_multi([self.eng.encode_segments(*self.px(self.r0_round))], all_clips=True)[0]
            self.evV.record(self.sV)

    def r0_step(self):
        """One clip ROUND = `world` clips: every rank has encoded its 12 segment blocks of the round (rotated plan) and all-gathered them;
        rank 0 prefills the `world` clips as ONE ragged pass and decodes them together while ALL ranks (rank 0's vision stream too) encode
        and exchange the next round.  Ranks >= 1 never touch their LLM.  Returns rank 0's ids of the round (None elsewhere)."""
        rnd, clips = self.r0_round, self.r0_next
        seqs = None
        if self.rank == 0:
            with self.D.on(self.sL):
                self.sL.wait_event(self.evV)
                embs = []
                for c, vis in enumerate(clips):
                    if self.D.gpu:
                        vis.record_stream(self.sL)
                    embs.append(self.eng.splice(self.r0_prompt(rnd, c), vis))
                seqs = [self.eng.seq_alloc(e.shape[0] + self.new_tokens) for e in embs]
                self.eng.prefill_batch(seqs, embs)
        with self.D.on(self.sV):
            self.r0_round = self.window(1)[0]
            self.r0_next = self._exchange_multi([self.eng.encode_segments(*self.px(self.r0_round))], all_clips=True)[0]
            self.evV.record(self.sV)
        if self.rank != 0:
            return None
        with self.D.on(self.sL):
            outs = self.eng.decode_greedy_batch(seqs, self.new_tokens, None)
            for q in seqs:
                self.eng.seq_free(q)
        return outs

    # ---- two clips in flight per GPU: vision of clip k+1 overlaps the (HBM-bound) decode of clip k ---------------
    def pipe_start(self):
        self.sV, self.sL = self.D.stream(), self.D.stream()
        self.evV, self.evL = self.D.event(), self.D.event()
        with self.D.on(self.sV):
            self.idx_next = self.window(1)
            self.vis_next = self.encode(self.idx_next[0])
            self.evV.record(self.sV)

    def pipe_start_multi(self, cps):
        self.cps = cps
        self.clip_batch = os.environ.get("GVL_BENCH_CLIP_BATCH", "1") != "0" and cps > 1
        self.sV, self.sL = self.D.stream(), self.D.stream()
        self.evV, self.evL = self.D.event(), self.D.event()
        with self.D.on(self.sV):
            self.idx_next = self.window(cps)
            self.vis_next = self.encode_multi(self.idx_next)
            self.evV.record(self.sV)

    def pipe_step_multi(self):
        """Completes `cps` clips (ONE ragged prefill over all of them, then ONE batched greedy decode: the LLM weights are streamed once
        per token for all of them -- continuous batching) and launches the vision encode of the next `cps` clips beside them."""
        idx = self.idx_next
        with self.D.on(self.sL):
            self.sL.wait_event(self.evV)
            embs = []
            for vis, i in zip(self.vis_next, idx):
                if self.D.gpu:
                    vis.record_stream(self.sL)
                embs.append(self.eng.splice(self.prompts[i % self.pool], vis))
            S = max(e.shape[0] for e in embs)
            seqs = [self.eng.seq_alloc(e.shape[0] + self.new_tokens) for e in embs]
            if self.batch_prefill:
                self.eng.prefill_batch(seqs, embs)       # gvl_prefill_varlen: the decoder GEMMs run over the packed rows of all clips of the step
            else:
                for seq, emb in zip(seqs, embs):
                    self.eng.prefill(seq, emb)
        with self.D.on(self.sV):
            self.idx_next = self.window(self.cps)
            self.vis_next = self.encode_multi(self.idx_next)
            self.evV.record(self.sV)
        with self.D.on(self.sL):
            outs = self.eng.decode_greedy_batch(seqs, self.new_tokens, None)   # synchronises its stream
            for seq in seqs:
                self.eng.seq_free(seq)
        self.last_idx = idx[-1]
        return outs[-1], S

    def pipe_step(self):
        """Completes ONE clip (prefill + decode) and launches ONE clip's vision encode for the next step."""
        i = self.idx_next[0]
        with self.D.on(self.sL):
            self.sL.wait_event(self.evV)                   # this clip's visual tokens are ready
            vis = self.vis_next
            if self.D.gpu:
                vis.record_stream(self.sL)
            seq, S = self.llm(vis, i)                      # splice + prefill (uses the workspace arena)
            self.evL.record(self.sL)
        with self.D.on(self.sV):
            if self.overlap == "decode":
                self.sV.wait_event(self.evL)               # next clip's towers start after this clip's prefill ...
            self.idx_next = self.window(1)
            self.vis_next = self.encode(self.idx_next[0])  # ... and run concurrently with the decode below (MFMA-bound vs HBM-bound);
            self.evV.record(self.sV)                       # "full": also beside the prefill (separate workspace arenas), filling its tail/launch bubbles
        with self.D.on(self.sL):
            out = self.decode(seq)
        self.last_idx = i
        return out, S

    time_decode = False
    exchange = "torch"
    clip_batch = False
    iv2_batch = int(os.environ.get("GVL_BENCH_IV2_BATCH", "0")) or 64      # clips per InternVideo2 call (capped by the clips of a step)
    last_idx = 0
    batch_prefill = os.environ.get("GVL_BENCH_BATCH_PREFILL", "1") != "0"
    overlap = os.environ.get("GVL_BENCH_OVERLAP", "full")


def _cpu_layer_samples(O):
    """One full-width layer of each tower at the C1 shapes on the host cores (seconds) -- scaled to a 96-frame clip by the caller."""
    t = {}
    Wv = synth.iv2_weights(depth=1, frames=8, seed="cpu.iv2")
    x = synth.det_tensor("cpu.iv2.x", (1, 2049, 1408), 0.5)
    t0 = time.perf_counter(); O.iv2_block(x, Wv, 0, 16); t["iv2_block_1seg"] = time.perf_counter() - t0
    Wc = synth.clip_weights(layers=1, seed="cpu.clip")
    x = synth.det_tensor("cpu.clip.x", (1, 577, 1024), 0.5)
    t0 = time.perf_counter(); O.clip_layer(x, Wc, 0, 16); t["clip_layer_1img"] = time.perf_counter() - t0
    Wl = synth.llm_weights("phi3", layers=1, vocab=64, seed="cpu.llm")
    ocfg = O.LLMConfig("phi3", 3072, 8192, 1, 32, 32, 64, 1e-5, 10000.0, 131072, 4096, *synth.longrope_factors(96))
    S = 880
    x = synth.det_tensor("cpu.llm.x", (S, 3072), 0.5)
    t0 = time.perf_counter(); O.llm_forward(ocfg, Wl, x, last_only=True); t["phi_layer_S880"] = time.perf_counter() - t0
    cache = [None]
    O.llm_forward(ocfg, Wl, x, cache=cache, last_only=True)
    t0 = time.perf_counter()
    for i in range(4):
        O.llm_forward(ocfg, Wl, x[:1], cache=cache, pos0=S + i, last_only=True)
    t["phi_layer_decode_tok"] = (time.perf_counter() - t0) / 4
    return t


def cpu_c1_measured(dev, O, Wc, Wv, Wp, Wl, new_tokens):
    """ONE real pass of the HEADLINE configuration (BASELINE configs[1]: 96 frames = 12 segments, S = 3519 prefill, `new_tokens` greedy tokens) through the fp32
    oracle on the host cores (--cpu-c1; ~6 min): the CPU figure beside `value`, measured instead of scaled from per-layer samples.  Inputs and weights are the
    streams of tests/golden/c1_free.json, whose `free_ids` are the REFERENCE's own free-running greedy ids for this clip."""
    with open(os.path.join(ROOT, "tests", "golden", "c1_free.json")) as f:
        meta = json.load(f)
    sd = meta["seeds"]
    sp = synth.exact_tensor(sd["sp"], (1, 12, 3, 336, 336), device=str(dev)).cpu()
    tp = synth.exact_tensor(sd["tp"], (1, 96, 3, 224, 224), device=str(dev)).cpu()
    ocfg = O.LLMConfig("phi3", 3072, 8192, 32, 32, 32, 32366, 1e-5, 10000.0, 131072, 4096, *synth.longrope_factors(96))
    ids = torch.tensor(json.loads(meta["ids"]) if isinstance(meta["ids"], str) else meta["ids"])
    t0 = time.perf_counter()
    vis = O.encode_images(sp, tp, Wc, Wv, Wp, "phi3.5")
    t1 = time.perf_counter()
    emb = O.splice(ids, vis[0], Wl["model.embed_tokens.weight"])
    out = O.greedy_generate(ocfg, Wl, emb, new_tokens, None, use_cache=True)
    t2 = time.perf_counter()
    n = min(len(out), len(meta["free_ids"]))
    return {"clips_per_s": round(1.0 / (t2 - t0), 6), "seconds": {"vision": round(t1 - t0, 1), "llm_prefill_plus_decode": round(t2 - t1, 1), "total": round(t2 - t0, 1)},
            "prefill_len": int(emb.shape[0]), "new_tokens": new_tokens, "cores": torch.get_num_threads(),
            "oracle_ids": out, "reference_free_ids": meta["free_ids"], "oracle_ids_equal_reference_golden": out[:n] == meta["free_ids"][:n]}


def cpu_baseline(dev, new_tokens=12, c1=False):
    """The CPU oracle (fp32 torch restatement of the reference, eager attention, KV-cached greedy as HF generate does) timed END TO END
    on BASELINE configs[0] -- Phi-3.5, ONE 8-frame segment: CLIP 23 L + InternVideo2 39 blocks + projectors + splice + 32-layer
    prefill (S = 384) + 12 greedy tokens -- on this box's host cores (SURVEY §8d 'CPU baseline').  The weights are
    the synth.exact_tensor streams of tests/golden/c0_full.npz (generated on the GPU, copied to the host), so the run also checks
    the oracle's ids against the REFERENCE's own greedy ids stored in that golden.  Plus the round-1 per-layer sample scaled to
    the 96-frame clip, for context."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gvl_oracle as O
    import numpy as np
    torch.set_grad_enabled(False)
    d = str(dev)
    z = np.load(os.path.join(ROOT, "tests", "golden", "c0_full.npz"))
    meta = json.loads(str(z["meta"]))
    sd = meta["seeds"]
    cpu = lambda W: {k: v.cpu() for k, v in W.items()}
    Wc = cpu(synth.clip_weights(seed=sd["clip"], device=d, exact=True))
    Wv = cpu(synth.iv2_weights(seed=sd["iv2"], device=d, exact=True))
    Wp = cpu(synth.projector_weights("phi3.5", seed=sd["proj"], device=d, exact=True))
    Wl = cpu(synth.llm_weights("phi3", seed=sd["llm"], device=d, exact=True))
    torch.cuda.empty_cache()
    sp = synth.exact_tensor(sd["sp"], (1, 1, 3, 336, 336), device=d).cpu()
    tp = synth.exact_tensor(sd["tp"], (1, 8, 3, 224, 224), device=d).cpu()
    ocfg = O.LLMConfig("phi3", 3072, 8192, 32, 32, 32, 32366, 1e-5, 10000.0, 131072, 4096, *synth.longrope_factors(96))
    ids = torch.tensor(meta["ids"])
    t0 = time.perf_counter()
    vis = O.encode_images(sp, tp, Wc, Wv, Wp, "phi3.5")
    t1 = time.perf_counter()
    emb = O.splice(ids, vis[0], Wl["model.embed_tokens.weight"])
    out = O.greedy_generate(ocfg, Wl, emb, new_tokens, None, use_cache=True)
    t2 = time.perf_counter()
    ref_ids = meta["greedy_ids"][:new_tokens]
    c1_skip = None
    if c1 == "auto" and (t2 - t0) * 7.3 > 720.0:          # C1 / C0 measured 7.25x on the 128-core box (350.6 s / 48.4 s, profiles/r04_cpu_c1.json)
        c1_skip = f"skipped: the C0 pass took {t2 - t0:.0f} s on {torch.get_num_threads()} threads, which predicts ~{(t2 - t0) * 7.3:.0f} s for the 96-frame pass (limit 720 s)"
        c1 = False
    c1_measured = cpu_c1_measured(dev, O, Wc, Wv, Wp, Wl, new_tokens) if c1 else None
    t = _cpu_layer_samples(O)
    c1_s = t["iv2_block_1seg"] * 39 * 12 + t["clip_layer_1img"] * 23 * 12 + t["phi_layer_S880"] * 4 * 32 + t["phi_layer_decode_tok"] * 32 * 12
    return {"value": round(1.0 / (t2 - t0), 5), "unit": "clips/s (8-frame C0 clip)", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"BASELINE configs[0] end to end on the host: Phi-3.5, 1 segment x 8 frames, S = {emb.shape[0]} prefill, {new_tokens} greedy tokens, "
                      "fp32 torch oracle (oracle/gvl_oracle.py), KV-cached greedy",
            "c0_seconds": {"vision": round(t1 - t0, 3), "llm_prefill_plus_decode": round(t2 - t1, 3), "total": round(t2 - t0, 3)},
            "oracle_ids_equal_reference_golden": out == ref_ids, "oracle_ids": out, "reference_ids": ref_ids,
            "c1_measured": c1_measured if c1_measured is not None else (c1_skip or "not run (--no-cpu-c1; the default run times one real 96-frame oracle pass, ~6 min of host time)"),
            "c1_scaled_estimate_clips_per_s": round(1.0 / c1_s, 5),
            "c1_scaled_estimate_from": "one full-width layer of each tower at the 96-frame shapes x layer counts (attention growth of the S=3520 prefill ignored): "
                                       + json.dumps({k: round(v, 3) for k, v in t.items()})}


def rccl_ranks_seen(eng, rank, world, dev, backend):
    """libgvl's OWN communicator over all ranks (gvl_comm_unique_id on rank 0 -> broadcast -> gvl_comm_init), ncclCommCount of it, and a
    gvl_allgather_visual of one row per rank checked against the expected rank order.  Returns (n_ranks, all-gather ok) or (None, None)
    when the engine has no RCCL path (the CPU plumbing test's stub)."""
    if not hasattr(eng, "comm_unique_id") or os.environ.get("GVL_BENCH_SAME_DEVICE"):      # RCCL refuses two ranks on one device (debug runs)
        return None, None
    if getattr(eng, "comm_world", 0) != world:                # (already built when the exchange itself ran through it)
        gdist.init_gvl_comm(eng)
    n = eng.comm_count()
    local = torch.full((1, 64), float(rank + 1), dtype=bf, device=dev)
    got = eng.allgather_visual(local)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    ok = bool(torch.equal(got[:, 0].float().cpu(), torch.arange(1, world + 1, dtype=torch.float32)))
    return n, ok


def main(argv=None, engine_factory=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--new-tokens", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain", action="store_true", help="timed region only (warmup + steps clips in the process): the target of the rocprofv3 passes")
    ap.add_argument("--clips-per-step", type=int, default=int(os.environ.get("GVL_BENCH_CPS", "8")),
                    help="pipelined mode: clips per GPU per step; their prefill is one ragged pass and their greedy decode is batched (one weight stream per token for all of them)")
    ap.add_argument("--mode", choices=["pipelined", "serial", "serial_step"], default="pipelined",
                    help="pipelined: 2 x cps clips in flight per GPU (vision of the next clips overlaps prefill + decode of the current ones); serial: one clip at a "
                         "time; serial_step: the launches of a pipelined step (cps clips, batched CLIP / InternVideo2 / prefill / decode) back to back on ONE stream -- "
                         "the rocprofv3 target whose per-kernel times `roofline` must agree with")
    ap.add_argument("--exchange", choices=["torch", "gvl"], default=os.environ.get("GVL_BENCH_EXCHANGE", "torch"),
                    help="N > 1: the all-gather of the visual tokens -- torch.distributed.all_gather_into_tensor (default), or libgvl's OWN RCCL communicator "
                         "through the C ABI (gvl_comm_init + gvl_allgather_visual: the exchange a non-Python host of the library performs).  The other one is "
                         "timed as an extra either way (`clips_per_s_other_exchange`)")
    ap.add_argument("--cpu-c1", action="store_true", help="(default since round 5; kept for old command lines) cpu_baseline: also time ONE real 96-frame (C1) oracle pass on the host cores (~6 min)")
    ap.add_argument("--no-cpu-c1", action="store_true", help="cpu_baseline: only the 8-frame C0 pass + the scaled C1 estimate (skips the ~6 min measured 96-frame pass)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=INT",
                    help="gvl_debug_set(KEY, INT) before the run (result-neutral launch parameters, include/gvl.h): A/B measurements in one place")
    ap.add_argument("--watchdog-s", type=float, default=float(os.environ.get("GVL_BENCH_WATCHDOG_S", "900")),
                    help="seconds one stage may last before the run is declared hung (0 = off)")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GVL_BENCH_BACKEND", "nccl")        # "gloo" + GVL_BENCH_SAME_DEVICE=1: debug the N>1 path on one GPU; "gloo" + a CPU engine_factory: plumbing test
    prog = Progress(rank, world, args.watchdog_s, {"steps": args.steps, "warmup": args.warmup, "mode": args.mode})
    prog.start()
    if os.environ.get("GVL_BENCH_SAME_DEVICE"):
        local = 0
    cpu_run = engine_factory is not None and not torch.cuda.is_available()
    if world > 1:
        prog.enter("init_process_group")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not cpu_run:
            torch.cuda.set_device(local)
        if not torch.distributed.is_initialized():
            if backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            else:
                torch.distributed.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    dev = torch.device("cpu") if cpu_run else torch.device(f"cuda:{local}")
    if not cpu_run:
        torch.cuda.set_device(dev)
    D = Dev(dev)

    cps = args.clips_per_step if args.mode in ("pipelined", "serial_step") else 1
    clip_batch = os.environ.get("GVL_BENCH_CLIP_BATCH", "1") != "0" and cps > 1
    prog.enter("build_engine (weights)")
    if engine_factory is not None:
        eng, geo = engine_factory(dev)
    else:
        eng, geo = build_engine(dev, max_segs=SEGS * cps if clip_batch else SEGS, new_tokens=args.new_tokens, clips_per_step=cps,
                                kv_pages=int(os.environ.get("GVL_BENCH_KV_PAGES", "0")))
    for kv in args.debug_set:
        k, v = kv.split("=")
        eng.debug_set(k, int(v))
    hw = getattr(eng, "bench_hw", (336, 224))
    st = Stepper(eng, geo, rank, world, args.new_tokens, pool=2 * cps, hw=hw)
    can_gvl = world > 1 and hasattr(eng, "comm_unique_id") and not os.environ.get("GVL_BENCH_SAME_DEVICE")      # RCCL refuses two ranks on one device (debug runs)
    if world > 1 and args.exchange == "gvl":
        if not can_gvl:
            raise SystemExit("--exchange gvl: this engine has no RCCL path")
        prog.enter("gvl_comm_init (libgvl's own RCCL communicator: --exchange gvl)")
        gdist.init_gvl_comm(eng)
        st.exchange = "gvl"

    def barrier(stage):
        prog.enter(stage + ": barrier")
        if world > 1:
            torch.distributed.barrier()
        D.sync()

    # ---- N > 1: diagnostics FIRST, so that a run that later hangs has already said what every rank can do ------------------------
    per_rank, ranks_seen, gvl_gather_ok = None, None, None
    if world > 1:
        mine = dict(st.stage_times(prog), rank=rank)
        prog.enter("all_gather_object(per-rank stage times)")
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        per_rank = gathered
        prog.partial["per_rank_stage_ms"] = per_rank

    S = 0
    if args.mode == "pipelined" and cps > 1:
        prog.enter("pipe_start (first vision encode + exchange)")
        st.pipe_start_multi(cps)
        stepfn = st.pipe_step_multi
    elif args.mode == "pipelined":
        prog.enter("pipe_start (first vision encode + exchange)")
        st.pipe_start()
        stepfn = st.pipe_step
    elif args.mode == "serial_step":
        st.cps, st.clip_batch = cps, clip_batch

        def stepfn():
            idx = st.window(cps)
            outs = st.step_multi_serial(idx)
            st.last_idx = idx[-1]
            return outs[-1], 0
    else:
        stepfn = st.step
    for w in range(args.warmup):
        prog.enter(f"warmup step {w}")
        _, S = stepfn()
    barrier("before timed region")
    mark = args.plain and hasattr(eng, "trace_marker") and D.gpu     # profiling runs: bracket the timed steps (tools/rocpd_stats.py --between gvl_trace_marker_kernel)
    if mark:
        eng.trace_marker(1); D.sync()
    t0 = time.perf_counter()
    out_timed = None
    for k in range(args.steps):
        prog.enter(f"timed step {k} (vision + exchange + prefill + decode)")
        out_timed, S = stepfn()
    barrier("after timed region")   # synchronises every stream, incl. the vision encode launched by the last step
    dt = time.perf_counter() - t0
    if mark:
        eng.trace_marker(2); D.sync()
    if world > 1:
        prog.enter("all_reduce(MAX) of the step time")
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    clips_per_s = world * args.steps * cps / dt
    # from here on a hang (watchdog) or a failure of an EXTRA still reports the measured headline value
    prog.partial.update(value=round(clips_per_s, 4), ms_per_step=round(1e3 * dt / args.steps, 2), scaling="weak", dtype="bf16", vs_baseline=None,
                        note="the run stopped after the timed region: extras missing")
    last_idx = st.last_idx if args.mode in ("pipelined", "serial_step") else (st.cursor - 1)

    if args.plain:
        if rank == 0:
            print(json.dumps({"metric": "clips/sec (plain run for profiling)", "value": round(clips_per_s, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2), "mode": args.mode, "clips_per_step": cps, "clips_in_process": (args.steps + args.warmup) * cps + (cps if args.mode == "pipelined" else 0)}), flush=True)
        prog.finish()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    # ---- the north-star plan: ONE clip sharded over the N ranks, LLM on rank 0 (strong scaling; N == 1: the serial single-clip latency) --
    prog.enter("single clip, sharded over the ranks (warm)")
    st.step_sharded()
    barrier("sharded clip")
    ts = time.perf_counter()
    for k in range(2):
        prog.enter(f"single clip, sharded over the ranks ({k})")
        st.step_sharded()
        barrier("sharded clip")
    sharded_ms = 1e3 * (time.perf_counter() - ts) / 2
    prog.partial["single_clip_latency_ms_sharded"] = round(sharded_ms, 2)
    # ---- the north-star plan as a throughput mode (N > 1): all ranks encode, ONE all-gather per clip round, rank 0 alone runs the LLM for the
    # `world` clips of the round (one ragged prefill + one batched decode) while every rank encodes the next round -- the third N > 1 figure
    r0 = None
    if world > 1 and world <= cps and args.mode == "pipelined" and os.environ.get("GVL_BENCH_R0", "1") != "0":
        try:
            prog.enter("rank-0-LLM pipelined mode: start")
            st.r0_start()
            st.r0_step()
            barrier("rank-0-LLM mode warm")
            t0r = time.perf_counter()
            outs_r0 = None
            for k in range(args.steps):
                prog.enter(f"rank-0-LLM pipelined mode: round {k}")
                rnd_done = st.r0_round
                outs_r0 = st.r0_step()
            barrier("rank-0-LLM mode")
            dtr = time.perf_counter() - t0r
            if world > 1:
                tt = torch.tensor([dtr], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                dtr = float(tt.item())
            # the ids rank 0 produced for clip c of the last round == what rank c produces for its own clip of that round (weak-scaling path, serial)
            prog.enter("rank-0-LLM pipelined mode: id check")
            mine_ids, _ = st.step(rnd_done)
            gathered_ids = [None] * world
            torch.distributed.all_gather_object(gathered_ids, list(mine_ids))
            r0 = {"clips_per_s": round(world * args.steps / dtr, 4), "ms_per_round": round(1e3 * dtr / args.steps, 2), "clips_per_round": world,
                  "ids_match_the_per_rank_llm": None if rank != 0 else bool(outs_r0 is not None and [list(o) for o in outs_r0] == gathered_ids),
                  "plan": f"{world} ranks encode {SEGS} segments each per round (rotated shard of the round's {world} clips), one all-gather, rank 0 prefills the "
                          f"{world} clips as one ragged pass and decodes them together while every rank encodes the next round"}
            prog.partial["rank0_llm_pipelined"] = r0
            st.cursor = (st.cursor + cps - 1) // cps * cps            # the rounds advanced the pool cursor one clip at a time: back onto a window boundary
        except Exception as e:                                   # an extra must not cost the line: the headline value is measured by now
            r0 = {"error": f"{type(e).__name__}: {e}"[:300]}
            prog.partial["rank0_llm_pipelined"] = r0
            print(f"bench: rank {rank}: rank-0-LLM mode failed: {e}", file=sys.stderr, flush=True)
    # ---- untimed extras: PCIe-inclusive rate, single-clip latency, decode-only rate, per-kernel-family profile, CPU baseline -----
    # The boundary takes DEVICE pixel tensors (`value` above); here every clip's 74 MB of f32 pixels is first copied from pinned
    # host memory on the vision stream, as a caller holding CPU-preprocessed frames would (inference.py:119-120 of the reference).
    clips_per_s_h2d = None
    if world == 1 and D.gpu:
        prog.enter("PCIe-inclusive passes")
        st.sp_host, st.tp_host = st.sp_pool.cpu().pin_memory(), st.tp_pool.cpu().pin_memory()
        st.h2d = True
        _ = stepfn()
        barrier("h2d")
        th = time.perf_counter()
        for _ in range(args.steps):
            _ = stepfn()
        barrier("h2d")
        clips_per_s_h2d = args.steps * cps / (time.perf_counter() - th)
        st.h2d = False
        _ = stepfn()                      # the next vision encode in flight reads resident pixels again
        del st.sp_host, st.tp_host
    prog.enter("serial single-clip passes")
    D.sync()
    tl = time.perf_counter()
    out_serial = None
    for _ in range(2):
        out_serial, _ = st.step(last_idx)
    D.sync()
    latency_ms = 1e3 * (time.perf_counter() - tl) / 2
    same_ids = list(out_timed) == list(out_serial)       # overlapped streams / ragged batches must not change a clip's ids: the last timed clip again, alone
    if not same_ids:
        print(f"bench: WARNING timed-mode ids {list(out_timed)} differ from serial ids {list(out_serial)}", file=sys.stderr)
    st.time_decode = True
    st.decode_s = 0.0
    for _ in range(2):
        st.step(0)
    decode_tok_s = 2 * (args.new_tokens - 1) / st.decode_s if st.decode_s > 0 else None
    st.time_decode = False
    # batched decode alone (cps sequences of different lengths share one weight stream per token)
    prog.enter("decode-only passes")
    decode_tok_s_batched = None

    def decode_only(n):
        seqs = [st.llm(st.encode(i), i)[0] for i in range(n)]
        D.sync()
        tb = time.perf_counter()
        eng.decode_greedy_batch(seqs, args.new_tokens, None)
        r = n * (args.new_tokens - 1) / (time.perf_counter() - tb)
        for q in seqs:
            eng.seq_free(q)
        return r
    if cps > 1:
        decode_tok_s_batched = decode_only(cps)
    # the decode path at the width it is built for: 16 sequences share one weight stream per step (skinny MFMA GEMM, gvl_decode.hip)
    decode_tok_s_16 = decode_only(16) if os.environ.get("GVL_BENCH_DECODE16", "1") != "0" and hasattr(eng, "prof_enable") else None
    # ---- per-kernel-family profile (hipEvent pairs around every launch, on the launch stream): (1) the launches of ONE TIMED STEP -- cps clips:
    # CLIP batched over the step's key frames, ONE ragged prefill, ONE batched decode -- back to back on one stream (what `roofline` reports: the
    # dominant kernel family as the timed region runs it), (2) one clip alone (rounds 1-2 reported this; kept as `roofline_one_clip_serial`)
    prog.enter("profiled passes")
    prof, prof1 = {}, {}
    fams = (("gemm", L.PROF_GEMM), ("attention", L.PROF_ATTN), ("gemv", L.PROF_GEMV), ("decode_attention", L.PROF_DECODE_ATTN), ("other", L.PROF_OTHER))
    n_prof = 1
    if hasattr(eng, "prof_enable"):
        if args.mode in ("pipelined", "serial_step") and cps > 1:
            idx = list(range(cps))
            st.step_multi_serial(idx)                         # warm (workspace, allocator)
            D.sync()
            eng.prof_enable(True)
            st.step_multi_serial(idx)
            n_prof = cps
        else:
            eng.prof_enable(True)
            st.step(0)
        for name, cat in fams:
            ms, n, work = eng.prof_read(cat)
            prof[name] = {"ms": ms, "launches": n, "work": work}
        eng.prof_enable(False)
        eng.prof_enable(True)
        st.step(0)
        for name, cat in fams:
            ms, n, work = eng.prof_read(cat)
            prof1[name] = {"ms": ms, "launches": n, "work": work}
        eng.prof_enable(False)
    g = prof.get("gemm", {"ms": 0.0, "launches": 0, "work": 0.0})
    g1 = prof1.get("gemm", {"ms": 0.0, "launches": 0, "work": 0.0})
    gemm_tflops = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    gemm1_tflops = g1["work"] / (g1["ms"] * 1e-3) / 1e12 if g1["ms"] > 0 else 0.0
    # HBM-side bytes per GEMM launch: NOT measured by this run -- PMC counters need their own rocprofv3 passes (MI355X_MICROARCH.md);
    # the figure is read from the committed summary of those passes over this same command and labelled as such (traffic_source)
    traffic, traffic_src, traffic_note = None, None, None
    import glob
    from grounded_video_llm_amd.build import source_sha16
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True)
    try:
        with open(cands[0]) as fpm:
            pm = json.load(fpm)
        if pm.get("src_sha16") != source_sha16():
            traffic_note = (f"profiles/{os.path.basename(cands[0])} was collected on other sources (stamp {pm.get('src_sha16')} != this tree {source_sha16()}): "
                            "not this run's figure, so none is printed; re-run tools/run_profiles.sh <tag> pmc")
        else:
            traffic_src = "profiles/" + os.path.basename(cands[0])
            # per LOGICAL GEMM launch of the traced mode: serial_step traces hold the step's batched launches, older ones one clip at a time
            per_clip = g["launches"] / n_prof if "serial_step" in pm.get("source", "") else g1["launches"]
            traffic = int(pm["families"]["gemm"]["total_traffic_bytes"] / max(1.0, pm["clips_in_trace"] * per_clip))
    except Exception as e:
        traffic_note = f"no usable PMC summary under profiles/ ({type(e).__name__})"
    roofline = {"bound": "mfma", "kernel": "gemm_a4p_kernel / gemm_a4_kernel / gemm_pp_kernel / gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16)", "achieved": round(gemm_tflops, 1), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_note if traffic is None else traffic_src + " (separate rocprofv3 --pmc passes of the serial_step bench on THIS source tree -- stamp matches; not measured by this run)",
                "measured_on": (f"the launches of ONE timed step ({n_prof} clips: CLIP batched over {12 * n_prof} key frames, one ragged prefill, one batched decode) "
                                "back to back on one stream, hipEvent pairs around every launch" if n_prof > 1 else
                                "ONE clip, stages back to back on one stream, hipEvent pairs around every launch"),
                "clips_in_profiled_pass": n_prof, "avg_launch_us": round(1e3 * g["ms"] / max(g["launches"], 1), 2), "launches_in_profiled_pass": g["launches"],
                "algorithmic_tflop_per_clip": round(g["work"] / 1e12 / n_prof, 2), "gemm_ms_per_clip": round(g["ms"] / n_prof, 3),
                "one_clip_serial": {"achieved": round(gemm1_tflops, 1), "frac": round(gemm1_tflops / PEAK_BF16_TFLOPS, 4), "launches_per_clip_serial": g1["launches"],
                                    "gemm_ms_per_clip_serial": round(g1["ms"], 3), "note": "one clip alone: CLIP on 12 frames and a 3.5 k-row prefill leave the 256 CUs under-filled (rounds 1-2 reported this figure)"}}
    stages = {}
    a = prof.get("attention", {"ms": 0})
    if a["ms"] > 0:
        stages["attention_tflops"] = round(a["work"] / (a["ms"] * 1e-3) / 1e12, 1)
    v = prof.get("gemv", {"ms": 0})
    if v["ms"] > 0:
        stages["decode_gemv_gbs"] = round(v["work"] / (v["ms"] * 1e-3) / 1e9, 1)      # work = 2*N*K flops == N*K*2 bytes of bf16 weights
        stages["decode_gemv_frac_hbm"] = round(stages["decode_gemv_gbs"] / PEAK_HBM_GBS, 4)
    for k, p in prof.items():
        stages[k + "_ms_per_clip"] = round(p["ms"] / n_prof, 3)          # the timed step's launches, per clip
    for k, p in prof1.items():
        stages[k + "_ms_per_clip_serial"] = round(p["ms"], 3)           # one clip alone

    other_exchange = None
    diag_stuck = False
    if world > 1 and can_gvl and args.mode == "pipelined" and cps > 1:
        # the same pipelined steps through the OTHER collective (libgvl's own communicator when `value` used torch.distributed, and vice versa):
        # in a helper thread with its own bound -- a communicator bootstrap that hangs must not cost the line
        other = "torch" if st.exchange == "gvl" else "gvl"
        prog.enter(f"timed steps with --exchange {other} (bounded extra)")
        reso = {}

        def _other():
            keep = st.exchange
            try:
                if D.gpu:
                    torch.cuda.set_device(dev)
                if other == "gvl" and getattr(eng, "comm_world", 0) != world:
                    gdist.init_gvl_comm(eng)
                st.exchange = other
                stepfn(); stepfn()                                   # the encode in flight still used the first collective
                torch.distributed.barrier(); D.sync()
                tq = time.perf_counter()
                for _ in range(args.steps):
                    stepfn()
                torch.distributed.barrier(); D.sync()
                reso["dt"] = time.perf_counter() - tq
                st.exchange = keep
                stepfn()
            except Exception as e:
                reso["err"] = f"{type(e).__name__}: {e}"
            finally:
                st.exchange = keep
        tho = threading.Thread(target=_other, daemon=True)
        tho.start()
        tho.join(timeout=2.0 * float(os.environ.get("GVL_BENCH_DIAG_S", "120")))
        if tho.is_alive():
            diag_stuck, other_exchange = True, {"exchange": other, "status": "timed out"}
            ranks_seen = "timed out"                             # the communicator check below would park in the same bootstrap: skipped
            print(f"bench: rank {rank}: the timed steps through --exchange {other} are still running after their bound: skipped", file=sys.stderr, flush=True)
        elif "err" in reso:
            other_exchange = {"exchange": other, "status": "failed: " + reso["err"][:200]}
        else:
            tt = torch.tensor([reso["dt"]], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            other_exchange = {"exchange": other, "clips_per_s": round(world * args.steps * cps / float(tt.item()), 4)}
    if world > 1 and not diag_stuck:
        # libgvl's OWN communicator over all ranks (how many ranks RCCL itself reports, and its all-gather) -- the last collective work of
        # the run, in a helper thread with its own 120 s bound: a diagnostic that hangs or fails must not cost the line (everything else
        # is measured by now); on a time-out the line says so and the process leaves without tearing the process group down
        prog.enter("gvl_comm_init / ncclCommCount / gvl_allgather_visual (bounded diagnostic)")
        res = {}

        def _diag():
            try:
                res["v"] = rccl_ranks_seen(eng, rank, world, dev, backend)
            except Exception as e:
                res["err"] = str(e)
        th = threading.Thread(target=_diag, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("GVL_BENCH_DIAG_S", "120")))
        if th.is_alive():
            diag_stuck, ranks_seen = True, "timed out"
            print(f"bench: rank {rank}: libgvl communicator check still running after its bound: skipped", file=sys.stderr, flush=True)
        elif "err" in res:
            ranks_seen = "failed: " + res["err"][:200]
            print(f"bench: rank {rank}: libgvl communicator check failed: {res['err']}", file=sys.stderr, flush=True)
        else:
            ranks_seen, gvl_gather_ok = res["v"]

    if rank == 0:
        out = {"metric": "clips/sec + grounding tokens/sec, 96-frame Phi3.5-3.8B @1/2/4/8 MI355X", "value": round(clips_per_s, 4), "unit": "clips/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic (random-init weights at full shape; every clip of a step has its own N(0,1) pixels and its own prompt ids and length; consecutive steps use different clips)",
               "config": {"workload": "Phi-3.5-3.8B, 96 frames (12 segs x 8), 336^2 spatial + 224^2 temporal, "
                                      f"prompts of {PROMPT_LEN_RANGE[0]}..{PROMPT_LEN_RANGE[1]} tokens (ragged), "
                                      f"{args.new_tokens} greedy tokens, {cps} clip{'s' if cps > 1 else ''} per GPU per step" +
                                      ((f" ({2 * cps} clips in flight per GPU: the vision encode of the next {cps} overlaps the ragged prefill + " +
                                        ("batched greedy decode" if cps > 1 else "decode") + f" of the current {cps}" +
                                        (f"; CLIP tower batched over the {SEGS * cps} key frames of a step" if clip_batch else "") + ")") if args.mode == "pipelined" else ""),
                          "clips_per_step": cps, "ms_per_clip": round(1e3 * dt / (args.steps * cps), 2), "prefill_len_max": S, "visual_tokens": 12 * st.L,
                          "distinct_clips_resident": st.pool,
                          "parallelism": "1 GPU" if world == 1 else f"frame-batch sharded over {world} GPUs + all-gather of visual tokens, LLM replica per clip"},
               "decode_tokens_per_s": None if decode_tok_s is None else round(world * decode_tok_s, 1),
               "decode_tokens_per_s_batched": None if decode_tok_s_batched is None else round(world * decode_tok_s_batched, 1),
               "decode_tokens_per_s_16seq": None if decode_tok_s_16 is None else round(world * decode_tok_s_16, 1),
               "clips_per_s_incl_pixel_h2d": None if clips_per_s_h2d is None else round(clips_per_s_h2d, 4), "single_clip_latency_ms": round(latency_ms, 2),
               "single_clip_latency_ms_sharded": round(sharded_ms, 2),
               "single_clip_sharded_plan": f"one clip's 12 segments over {world} rank{'s' if world > 1 else ''}, one all-gather, prefill + decode on rank 0 (strong scaling; `value` is the weak-scaling plan)",
               "mode": args.mode, "ids_match_serial": same_ids,
               "roofline": roofline, "stages": stages, "kv_pool": eng.kv_info()}
        if world > 1:
            out["exchange"] = st.exchange + (" (torch.distributed.all_gather_into_tensor)" if st.exchange == "torch" else " (gvl_allgather_visual through the C ABI, libgvl's own communicator)")
            out["clips_per_s_other_exchange"] = other_exchange
            out["rank0_llm_pipelined"] = r0
            out["per_rank_stage_ms"] = per_rank
            out["n_ranks_seen_by_rccl"] = ranks_seen
            out["gvl_allgather_matches_rank_order"] = gvl_gather_ok
        if world == 1 and not args.no_cpu_baseline and D.gpu:
            prog.enter("cpu_baseline (host cores)")
            prog.limit = max(prog.limit, 1800.0) if prog.limit > 0 else 0
            eng.close()                                  # give the HBM back: the C0 weights are generated on the GPU, then copied to the host
            # the measured 96-frame pass runs by default (the driver's own line carries the whole CPU baseline); it is skipped when the
            # C0 pass just timed predicts more than ~12 min for it (a host with few cores), so that the default run still ends within minutes
            out["cpu_baseline"] = cpu_baseline(dev, args.new_tokens, c1="auto" if not args.no_cpu_c1 else False)
        print(json.dumps(out), flush=True)
    prog.finish()
    if diag_stuck:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)                                      # a thread is parked inside a collective: no orderly teardown possible, and none needed
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
