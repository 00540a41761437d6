"""bench.py -- the reference's headline metric on MI355X: end-to-end clips/s (and grounding-answer
tokens/s) for the 96-frame Phi-3.5-3.8B configuration (BASELINE.json configs[1]).

A "step" = one pass of the hot path over one batch of synthetic input: for every rank `--clips-per-step` (default 8)
96-frame clips, each (12 x 336^2 spatial frames + 96 x 224^2 temporal frames, already resident in HBM) -> CLIP ViT-L/14-336
(23 layers) + InternVideo2-1B (39 blocks) -> merge/pool + projectors -> 3420 visual tokens spliced into a
~100-token prompt -> Phi-3.5 prefill (S ~ 3520) -> greedy decode of 12 new tokens through the paged KV cache.
The clips of a step are prefilled one after the other and decoded TOGETHER (gvl_decode_greedy_batch: every weight matrix is
streamed once per token for all of them -- the reference batches clips in generate() too, llava_next_video.py:622-647), while
the vision encode of the next step's clips runs on a second stream (the CLIP tower once over the 12 x 4 key frames of the step --
gvl_clip_encode -- then InternVideo2 + projectors per clip: gvl_iv2_encode / gvl_build_visual).  `single_clip_latency_ms` (one clip, stages back to back)
is reported next to `value`; `--clips-per-step 1` gives the one-clip-per-step pipeline (8.6 clips/s); 4 / 8 / 12 clips per
step measured 10.0 / 10.3 / 10.15 clips/s on one box (the decode of a step's clips is ONE skinny-GEMM group of up to 16 sequences).
Random-init weights of the real architecture, synthetic pixels (no network for checkpoints/datasets).

N > 1 (one process per GPU, RCCL): N clips in flight per step; every clip's 12-segment frame batch is
sharded over ALL N ranks (rotated per clip so the 12 % N remainder balances: each rank encodes exactly 12
segments), ONE all-gather of the visual tokens over xGMI, then rank c runs the LLM for clip c.
Weak scaling: per-GPU work is fixed.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (the bf16 MFMA GEMM), measured
live with HIP events on the launch stream in one extra profiled step; `cpu_baseline` is the CPU oracle timed on
a bounded sample on the host cores (rank 0, N == 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import _gvl_bootstrap  # noqa: E402,F401
from grounded_video_llm_amd import dist as gdist, engine as E, lib as L, synth, weights as Wt  # noqa: E402

bf = torch.bfloat16
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


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


def make_inputs(dev, rank, n_segs=12, fps=8, n_text=100):
    g = torch.Generator(device=dev); g.manual_seed(42 + rank)
    sp = torch.randn((n_segs, 3, 336, 336), device=dev, generator=g)
    tp = torch.randn((n_segs, 3, fps, 224, 224), device=dev, generator=g)
    gi = torch.Generator(); gi.manual_seed(42)
    ids = torch.randint(3, 32000, (n_text,), generator=gi).tolist()
    ids[36] = -200                      # the template's image slot sits after the system prompt
    return sp, tp, ids


class Stepper:
    def __init__(self, eng, geo, rank, world, new_tokens):
        self.eng, self.geo, self.rank, self.world, self.new_tokens = eng, geo, rank, world, new_tokens
        self.dev = eng.device
        self.sp, self.tp, self.ids = make_inputs(self.dev, rank)
        self.L = eng.tokens_per_seg
        self.decode_s = 0.0
        self.h2d = False
        if world > 1:
            # clip c's segment block b goes to rank (b + c) % world (dist.rotated_encode_plan): every rank encodes exactly 12
            # segments.  Synthetic pixels: every rank generates the segments it encodes itself.
            self.mine = gdist.rotated_encode_plan(12, rank, world)
            self.gather = gdist.rotated_gather_index(12, rank, world)
            n_mine = sum(h - l for _, l, h in self.mine)
            assert n_mine == 12
            g = torch.Generator(device=self.dev); g.manual_seed(1000 + rank)
            self.sp = torch.randn((n_mine, 3, 336, 336), device=self.dev, generator=g)
            self.tp = torch.randn((n_mine, 3, 8, 224, 224), device=self.dev, generator=g)

    # ---- the three stages of one clip ---------------------------------------------------------------------
    def _exchange(self, vis, force=False):
        return self._exchange_multi([vis], force)[0]

    def _exchange_multi(self, vis_list, force=False):
        """N > 1: ONE all-gather per STEP: this rank's 12 segment blocks of each of the step's `cps` clip rounds travel together
        (cps x 3.5 MB per rank; xGMI is point-to-point, every rank pushes to its 7 peers at once), then each round picks the blocks
        of this rank's clip in segment order.  force: run the collective at world == 1 too (RCCL smoke test on a 1-GPU box)."""
        if self.world == 1 and not force:
            return vis_list
        cps, rows = len(vis_list), vis_list[0].shape[0]
        send = torch.cat(vis_list, 0) if cps > 1 else vis_list[0]
        recv = torch.empty((self.world * send.shape[0], send.shape[1]), dtype=bf, device=self.dev)
        if torch.distributed.get_backend() == "gloo":                 # debug only (GVL_BENCH_BACKEND=gloo): stage through the host
            rc = recv.cpu(); torch.distributed.all_gather_into_tensor(rc, send.cpu()); recv.copy_(rc)
        else:
            torch.distributed.all_gather_into_tensor(recv, send)      # RCCL over xGMI
        recv = recv.view(self.world, cps, rows, -1)
        gather = self.gather if self.world > 1 else [(0, 0, rows // self.L)]
        return [torch.cat([recv[src, c, off * self.L:(off + n) * self.L] for src, off, n in gather], 0) for c in range(cps)]   # clip == rank, segment order

    def encode(self):
        """vision towers + projectors for this rank's 12 segments (+ the all-gather for N > 1) -> visual tokens of THIS rank's clip."""
        if self.h2d:                                                    # extra (untimed for `value`): pixels arrive over PCIe
            self.sp.copy_(self.sp_host, non_blocking=True)
            self.tp.copy_(self.tp_host, non_blocking=True)
        return self._exchange(self.eng.encode_segments(self.sp, self.tp))   # [12*L, hidden]

    def encode_multi(self, cps):
        """Vision encode of the next `cps` clip rounds.  The CLIP tower runs ONCE over the 12 x cps key frames of the step (per clip
        its N = 1024 GEMMs have only 112 tiles: +1.6 % clips/s measured); InternVideo2 + projectors (+ the all-gather) stay per clip.
        GVL_BENCH_CLIP_BATCH=0 restores one gvl_encode_segments call per clip."""
        if not self.clip_batch:
            if self.h2d:
                return [self.encode() for _ in range(cps)]
            return self._exchange_multi([self.eng.encode_segments(self.sp, self.tp) for _ in range(cps)])
        if self.h2d:
            for c in range(cps):
                self.sp_multi[c * 12:(c + 1) * 12].copy_(self.sp_host, non_blocking=True)
            self.tp.copy_(self.tp_host, non_blocking=True)
        cf = self.eng.clip_encode(self.sp_multi)
        return self._exchange_multi([self.eng.build_visual(cf[c * 12:(c + 1) * 12], self.eng.iv2_encode(self.tp)) for c in range(cps)])

    def stage_times(self):
        """One clip round on this rank, stages back to back with HIP events between them (diagnosis of the N > 1 runs):
        vision encode of 12 segments / exchange / splice + prefill / greedy decode, in ms."""
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        torch.cuda.synchronize()
        ev[0].record()
        vis = self.eng.encode_segments(self.sp, self.tp)
        ev[1].record()
        vis = self._exchange(vis)
        ev[2].record()
        seq, _ = self.llm(vis)
        ev[3].record()
        self.decode(seq)
        ev[4].record()
        torch.cuda.synchronize()
        names = ("vision_ms", "exchange_ms", "prefill_ms", "decode_ms")
        return {n: round(ev[i].elapsed_time(ev[i + 1]), 3) for i, n in enumerate(names)}

    def llm(self, vis):
        eng = self.eng
        emb = eng.splice(self.ids, vis)
        seq = eng.seq_alloc(emb.shape[0] + self.new_tokens)
        eng.prefill(seq, emb)
        return seq, emb.shape[0]

    def decode(self, seq):
        if self.time_decode:
            torch.cuda.current_stream().synchronize()
        t0 = time.perf_counter()
        out = self.eng.decode_greedy(seq, self.new_tokens, None)       # synchronises its stream
        if self.time_decode:
            self.decode_s += time.perf_counter() - t0
        self.eng.seq_free(seq)
        return out

    def step(self):
        """One clip, stages back to back on one stream (single-clip latency)."""
        vis = self.encode()
        seq, S = self.llm(vis)
        return self.decode(seq), S

    # ---- two clips in flight per GPU: vision of clip k+1 overlaps the (HBM-bound) decode of clip k ---------------
    def pipe_start(self):
        self.sV, self.sL = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        self.evV, self.evL = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(self.sV):
            self.vis_next = self.encode()
            self.evV.record(self.sV)

    def pipe_start_multi(self, cps):
        self.cps = cps
        self.clip_batch = os.environ.get("GVL_BENCH_CLIP_BATCH", "1") != "0" and cps > 1
        self.sp_multi = self.sp.repeat(cps, 1, 1, 1) if self.clip_batch else None     # the key frames of the step's clips, resident
        self.sV, self.sL = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        self.evV, self.evL = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(self.sV):
            self.vis_next = self.encode_multi(cps)
            self.evV.record(self.sV)

    def pipe_step_multi(self):
        """Completes `cps` clips (prefill each, then ONE batched greedy decode: the LLM weights are streamed once per token for all
        of them -- continuous batching) and launches the vision encode of the next `cps` clips beside them."""
        with torch.cuda.stream(self.sL):
            self.sL.wait_event(self.evV)
            embs = []
            for vis in self.vis_next:
                vis.record_stream(self.sL)
                embs.append(self.eng.splice(self.ids, vis))
            S = embs[0].shape[0]
            seqs = [self.eng.seq_alloc(S + self.new_tokens) for _ in embs]
            if self.batch_prefill:
                self.eng.prefill_batch(seqs, embs)       # the decoder GEMMs run over the rows of all clips of the step at once
            else:
                for seq, emb in zip(seqs, embs):
                    self.eng.prefill(seq, emb)
        with torch.cuda.stream(self.sV):
            self.vis_next = self.encode_multi(self.cps)
            self.evV.record(self.sV)
        with torch.cuda.stream(self.sL):
            outs = self.eng.decode_greedy_batch(seqs, self.new_tokens, None)   # synchronises its stream
            for seq in seqs:
                self.eng.seq_free(seq)
        return outs[-1], S

    def pipe_step(self):
        """Completes ONE clip (prefill + decode) and launches ONE clip's vision encode for the next step."""
        with torch.cuda.stream(self.sL):
            self.sL.wait_event(self.evV)                   # this clip's visual tokens are ready
            vis = self.vis_next
            vis.record_stream(self.sL)
            seq, S = self.llm(vis)                         # splice + prefill (uses the workspace arena)
            self.evL.record(self.sL)
        with torch.cuda.stream(self.sV):
            if self.overlap == "decode":
                self.sV.wait_event(self.evL)               # next clip's towers start after this clip's prefill ...
            self.vis_next = self.encode()                  # ... and run concurrently with the decode below (MFMA-bound vs HBM-bound);
            self.evV.record(self.sV)                       # "full": also beside the prefill (separate workspace arenas), filling its tail/launch bubbles
        with torch.cuda.stream(self.sL):
            out = self.decode(seq)
        return out, S

    time_decode = False
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


def cpu_baseline(dev, new_tokens=12):
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
    t = _cpu_layer_samples(O)
    c1_s = t["iv2_block_1seg"] * 39 * 12 + t["clip_layer_1img"] * 23 * 12 + t["phi_layer_S880"] * 4 * 32 + t["phi_layer_decode_tok"] * 32 * 12
    return {"value": round(1.0 / (t2 - t0), 5), "unit": "clips/s (8-frame C0 clip)", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"BASELINE configs[0] end to end on the host: Phi-3.5, 1 segment x 8 frames, S = {emb.shape[0]} prefill, {new_tokens} greedy tokens, "
                      "fp32 torch oracle (oracle/gvl_oracle.py), KV-cached greedy",
            "c0_seconds": {"vision": round(t1 - t0, 3), "llm_prefill_plus_decode": round(t2 - t1, 3), "total": round(t2 - t0, 3)},
            "oracle_ids_equal_reference_golden": out == ref_ids, "oracle_ids": out, "reference_ids": ref_ids,
            "c1_scaled_estimate_clips_per_s": round(1.0 / c1_s, 5),
            "c1_scaled_estimate_from": "one full-width layer of each tower at the 96-frame shapes x layer counts (attention growth of the S=3520 prefill ignored): "
                                       + json.dumps({k: round(v, 3) for k, v in t.items()})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--new-tokens", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain", action="store_true", help="timed region only (warmup + steps clips in the process): the target of the rocprofv3 passes")
    ap.add_argument("--clips-per-step", type=int, default=int(os.environ.get("GVL_BENCH_CPS", "8")),
                    help="pipelined mode: clips per GPU per step; their greedy decode is batched (one weight stream per token for all of them)")
    ap.add_argument("--mode", choices=["pipelined", "serial"], default="pipelined",
                    help="pipelined: 2 clips in flight per GPU (vision of clip k+1 overlaps decode of clip k); serial: one clip at a time")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("GVL_BENCH_BACKEND", "nccl")        # "gloo" + GVL_BENCH_SAME_DEVICE=1: debug the N>1 path on one GPU
    if os.environ.get("GVL_BENCH_SAME_DEVICE"):
        local = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            torch.distributed.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    cps = args.clips_per_step if args.mode == "pipelined" else 1
    clip_batch = os.environ.get("GVL_BENCH_CLIP_BATCH", "1") != "0" and cps > 1
    eng, geo = build_engine(dev, max_segs=12 * cps if clip_batch else 12, new_tokens=args.new_tokens, clips_per_step=cps,
                            kv_pages=int(os.environ.get("GVL_BENCH_KV_PAGES", "0")))
    st = Stepper(eng, geo, rank, world, args.new_tokens)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    S = 0
    if args.mode == "pipelined" and cps > 1:
        st.pipe_start_multi(cps)
        stepfn = st.pipe_step_multi
    elif args.mode == "pipelined":
        st.pipe_start()
        stepfn = st.pipe_step
    else:
        stepfn = st.step
    for _ in range(args.warmup):
        _, S = stepfn()
    barrier()
    t0 = time.perf_counter()
    out_timed = None
    for _ in range(args.steps):
        out_timed, S = stepfn()
    barrier()                       # torch.cuda.synchronize(): every stream, incl. the vision encode launched by the last step
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    clips_per_s = world * args.steps * cps / dt

    if args.plain:
        if rank == 0:
            print(json.dumps({"metric": "clips/sec (plain run for profiling)", "value": round(clips_per_s, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2), "mode": args.mode, "clips_in_process": (args.steps + args.warmup) * cps + (cps if args.mode == "pipelined" else 0)}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    # ---- untimed extras: PCIe-inclusive rate, single-clip latency, decode-only rate, per-kernel-family profile, CPU baseline -----
    # The boundary takes DEVICE pixel tensors (`value` above); here every clip's 74 MB of f32 pixels is first copied from pinned
    # host memory on the vision stream, as a caller holding CPU-preprocessed frames would (inference.py:119-120 of the reference).
    clips_per_s_h2d = None
    if world == 1:
        st.sp_host, st.tp_host = st.sp.cpu().pin_memory(), st.tp.cpu().pin_memory()
        st.h2d = True
        _ = stepfn()
        barrier()
        th = time.perf_counter()
        for _ in range(args.steps):
            _ = stepfn()
        barrier()
        clips_per_s_h2d = args.steps * cps / (time.perf_counter() - th)
        st.h2d = False
        _ = stepfn()                      # the next vision encode in flight reads resident pixels again
    torch.cuda.synchronize()
    tl = time.perf_counter()
    for _ in range(2):
        out_serial, _ = st.step()
    torch.cuda.synchronize()
    latency_ms = 1e3 * (time.perf_counter() - tl) / 2
    same_ids = list(out_timed) == list(out_serial)       # overlapped streams must not change the generated ids (same inputs every step)
    if not same_ids:
        print(f"bench: WARNING timed-mode ids {list(out_timed)} differ from serial ids {list(out_serial)}", file=sys.stderr)
    st.time_decode = True
    st.decode_s = 0.0
    for _ in range(2):
        st.step()
    decode_tok_s = 2 * (args.new_tokens - 1) / st.decode_s if st.decode_s > 0 else None
    st.time_decode = False
    # batched decode alone (cps sequences share one weight stream per token)
    decode_tok_s_batched = None
    if cps > 1:
        vis = st.encode()
        seqs = [st.llm(vis)[0] for _ in range(cps)]
        torch.cuda.synchronize()
        tb = time.perf_counter()
        eng.decode_greedy_batch(seqs, args.new_tokens, None)
        decode_tok_s_batched = cps * (args.new_tokens - 1) / (time.perf_counter() - tb)
        for q in seqs:
            eng.seq_free(q)
    # the decode path at the width it is built for: 16 sequences share one weight stream per step (skinny MFMA GEMM, gvl_decode.hip)
    decode_tok_s_16 = None
    if os.environ.get("GVL_BENCH_DECODE16", "1") != "0":
        vis = st.encode()
        seqs = [st.llm(vis)[0] for _ in range(16)]
        torch.cuda.synchronize()
        tb = time.perf_counter()
        eng.decode_greedy_batch(seqs, args.new_tokens, None)
        decode_tok_s_16 = 16 * (args.new_tokens - 1) / (time.perf_counter() - tb)
        for q in seqs:
            eng.seq_free(q)
    eng.prof_enable(True)
    st.step()
    prof = {}
    for name, cat in (("gemm", L.PROF_GEMM), ("attention", L.PROF_ATTN), ("gemv", L.PROF_GEMV), ("decode_attention", L.PROF_DECODE_ATTN), ("other", L.PROF_OTHER)):
        ms, n, work = eng.prof_read(cat)
        prof[name] = {"ms": ms, "launches": n, "work": work}
    eng.prof_enable(False)
    g = prof["gemm"]
    gemm_tflops = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
    # HBM-side bytes per GEMM launch: NOT measured by this run -- PMC counters need their own rocprofv3 passes (MI355X_MICROARCH.md);
    # the figure is read from the committed summary of those passes over this same command and labelled as such (traffic_source)
    traffic, traffic_src = None, None
    for cand in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        if os.path.exists(os.path.join(ROOT, "profiles", cand)):
            traffic_src = "profiles/" + cand
            break
    try:
        with open(os.path.join(ROOT, traffic_src)) as fpm:
            pm = json.load(fpm)
            traffic = int(pm["families"]["gemm"]["total_traffic_bytes"] / (pm["clips_in_trace"] * max(1, g["launches"])))   # per LOGICAL GEMM launch
    except Exception:
        pass
    roofline = {"bound": "mfma", "kernel": "gemm_pp_kernel / gemm_bf16_kernel (v_mfma_f32_32x32x16_bf16)", "achieved": round(gemm_tflops, 1), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(gemm_tflops / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_source": None if traffic is None else traffic_src + " (separate rocprofv3 --pmc passes of this command; not measured by this run)",
                "avg_launch_us": round(1e3 * g["ms"] / max(g["launches"], 1), 2), "launches_per_step": g["launches"],
                "algorithmic_tflop_per_step": round(g["work"] / 1e12, 2)}
    stages = {}
    a = prof["attention"]
    if a["ms"] > 0:
        stages["attention_tflops"] = round(a["work"] / (a["ms"] * 1e-3) / 1e12, 1)
    v = prof["gemv"]
    if v["ms"] > 0:
        stages["decode_gemv_gbs"] = round(v["work"] / (v["ms"] * 1e-3) / 1e9, 1)      # work = 2*N*K flops == N*K*2 bytes of bf16 weights
        stages["decode_gemv_frac_hbm"] = round(stages["decode_gemv_gbs"] / PEAK_HBM_GBS, 4)
    for k, p in prof.items():
        stages[k + "_ms_per_step"] = round(p["ms"], 3)

    per_rank = None
    if world > 1:                                       # per-rank stage times of one un-overlapped clip round: makes a scaling run diagnosable
        mine = dict(st.stage_times(), rank=rank)
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        per_rank = gathered
    if rank == 0:
        out = {"metric": "clips/sec + grounding tokens/sec, 96-frame Phi3.5-3.8B @1/2/4/8 MI355X", "value": round(clips_per_s, 4), "unit": "clips/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights at full shape, N(0,1) pixels)",
               "config": {"workload": "Phi-3.5-3.8B, 96 frames (12 segs x 8), 336^2 spatial + 224^2 temporal, ~100-token prompt, "
                                      f"{args.new_tokens} greedy tokens, {cps} clip{'s' if cps > 1 else ''} per GPU per step" +
                                      ((f" ({2 * cps} clips in flight per GPU: the vision encode of the next {cps} overlaps the prefill + " +
                                        ("batched greedy decode" if cps > 1 else "decode") + f" of the current {cps}" +
                                        (f"; CLIP tower batched over the {12 * cps} key frames of a step" if clip_batch else "") + ")") if args.mode == "pipelined" else ""),
                          "clips_per_step": cps, "ms_per_clip": round(1e3 * dt / (args.steps * cps), 2), "prefill_len": S, "visual_tokens": 12 * st.L,
                          "parallelism": "1 GPU" if world == 1 else f"frame-batch sharded over {world} GPUs + all-gather of visual tokens, LLM replica per clip"},
               "decode_tokens_per_s": None if decode_tok_s is None else round(world * decode_tok_s, 1),
               "decode_tokens_per_s_batched": None if decode_tok_s_batched is None else round(world * decode_tok_s_batched, 1),
               "decode_tokens_per_s_16seq": None if decode_tok_s_16 is None else round(world * decode_tok_s_16, 1),
               "clips_per_s_incl_pixel_h2d": None if clips_per_s_h2d is None else round(clips_per_s_h2d, 4), "single_clip_latency_ms": round(latency_ms, 2), "mode": args.mode, "ids_match_serial": same_ids,
               "roofline": roofline, "stages": stages, "kv_pool": eng.kv_info()}
        if per_rank is not None:
            out["per_rank_stage_ms"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            eng.close()                                  # give the HBM back: the C0 weights are generated on the GPU, then copied to the host
            out["cpu_baseline"] = cpu_baseline(dev, args.new_tokens)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
