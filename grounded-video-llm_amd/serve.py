"""Continuous batching of clips on the LLM.

The reference answers a batch of clips with ONE `language_model.generate(inputs_embeds=...)` over a left-padded batch
(models/llava_next_video.py:622-661): all members start together and the call returns when the LAST one has produced its
eos.  `ClipScheduler` keeps the same per-request result (greedy ids up to and including eos, at most max_new_tokens) but
lets requests join and leave between decode chunks:

  admit    queued requests take a free slot: splice -> KV pages (gvl_seq_alloc) -> ONE ragged prefill for all newcomers
           (gvl_prefill_varlen: packed rows through the decoder GEMMs)
  decode   every active sequence advances `chunk` tokens (gvl_decode_steps: members at different generation steps share
           one weight stream per step in groups of up to 16)
  retire   ids are read back (gvl_seq_read), sequences that produced eos / reached max_new free their pages at once

All arithmetic is per sequence and batch-invariant, so the ids equal `Engine.generate_ids` of each request on its own
(tests/test_gpu_llm.py::test_scheduler_matches_one_at_a_time).  The scheduler itself is plain host logic and is tested
on CPU against a scripted engine (tests/test_host_logic.py).
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import Deque, Dict, List, Optional, Sequence

from . import lib as L


@dataclass
class _Request:
    rid: int
    embeds: object               # bf16 [S, hidden] device tensor (output of Engine.splice)
    max_new: int
    seq: int = -1
    ids: List[int] = field(default_factory=list)
    n_gen: int = 0               # tokens generated on the device so far (>= len(ids) once eos was seen)


class ClipScheduler:
    """engine: anything with seq_alloc / seq_free / prefill_batch / decode_steps / seq_read (grounded_video_llm_amd.engine.Engine)."""

    def __init__(self, engine, eos_id: Optional[int], max_active: int = 8, chunk: int = 8, max_prefill_rows: Optional[int] = None):
        if max_active < 1 or chunk < 1:
            raise ValueError("max_active and chunk must be >= 1")
        self.eng, self.eos, self.max_active, self.chunk = engine, eos_id, int(max_active), int(chunk)
        self.max_prefill_rows = max_prefill_rows
        self.queue: Deque[_Request] = deque()
        self.active: List[_Request] = []
        self.done: Dict[int, List[int]] = {}
        self._next = 0
        self.stats = {"prefill_calls": 0, "decode_chunks": 0, "decode_seq_steps": 0, "wasted_seq_steps": 0, "max_concurrent": 0}

    # ---- public ------------------------------------------------------------------------------------------
    def submit(self, embeds, max_new_tokens: int) -> int:
        if max_new_tokens < 1:
            raise ValueError("max_new_tokens must be >= 1")
        r = _Request(self._next, embeds, int(max_new_tokens))
        self._next += 1
        self.queue.append(r)
        return r.rid

    def pending(self) -> int:
        return len(self.queue) + len(self.active)

    def step(self) -> List[int]:
        """One scheduler iteration: admit -> decode chunk -> retire.  Returns the ids of the requests that finished."""
        self._admit()
        finished = self._retire()                # a request can finish on its prefill token (eos first, or max_new == 1)
        if self.active:
            k = min([self.chunk] + [r.max_new - r.n_gen for r in self.active])
            self.eng.decode_steps([r.seq for r in self.active], k)
            self.stats["decode_chunks"] += 1
            self.stats["decode_seq_steps"] += k * len(self.active)
            for r in self.active:
                r.n_gen += k
            finished += self._retire()
        return finished

    def run(self) -> Dict[int, List[int]]:
        while self.pending():
            before = (len(self.queue), len(self.active))
            self.step()
            if not self.active and self.queue and before == (len(self.queue), 0):
                raise L.GvlError("scheduler: the head request does not fit the KV pool / prefill workspace even on an idle engine")
        out, self.done = self.done, {}
        return out

    # ---- internals ---------------------------------------------------------------------------------------
    def _admit(self):
        new: List[_Request] = []
        rows = 0
        while self.queue and len(self.active) + len(new) < self.max_active:
            r = self.queue[0]
            S = int(r.embeds.shape[0])
            if self.max_prefill_rows is not None and new and rows + S > self.max_prefill_rows:
                break                                        # next iteration: keep the newcomers' prefill inside the workspace
            max_seq = getattr(getattr(self.eng, "geo", None), "max_seq", None)
            cap = S + r.max_new if max_seq is None else min(S + r.max_new, int(max_seq))
            if cap < S:
                raise ValueError(f"request {r.rid}: {S} prefill tokens exceed the engine's max_seq {max_seq}")
            try:
                r.seq = self.eng.seq_alloc(cap)
            except L.GvlError as e:
                if getattr(e, "status", 0) == L.ERR_OOM:     # KV pages exhausted: wait for a retirement (FIFO, no overtaking)
                    break
                raise
            r.max_new = min(r.max_new, cap - S + 1)          # generate() stops at the context limit (Engine.generate_ids does too)
            self.queue.popleft()
            new.append(r)
            rows += S
        if new:
            self.eng.prefill_batch([r.seq for r in new], [r.embeds for r in new])
            self.stats["prefill_calls"] += 1
            for r in new:
                r.n_gen = 1
                r.embeds = None                              # the KV cache holds it now
            self.active += new
            self.stats["max_concurrent"] = max(self.stats["max_concurrent"], len(self.active))

    def _retire(self) -> List[int]:
        finished, keep = [], []
        for r in self.active:
            fresh = self.eng.seq_read(r.seq, len(r.ids), r.n_gen - len(r.ids)) if r.n_gen > len(r.ids) else []
            stop = False
            for i, t in enumerate(fresh):
                r.ids.append(t)
                if self.eos is not None and t == self.eos:
                    self.stats["wasted_seq_steps"] += len(fresh) - 1 - i
                    stop = True
                    break
            if stop or len(r.ids) >= r.max_new:
                self.eng.seq_free(r.seq)                     # seq_read synchronised the stream: no step of r is in flight
                self.done[r.rid] = r.ids
                finished.append(r.rid)
            else:
                keep.append(r)
        self.active = keep
        return finished


def generate_many(engine, embeds_list: Sequence, max_new_tokens: int, eos_id: Optional[int], max_active: int = 8, chunk: int = 8,
                  max_prefill_rows: Optional[int] = None) -> List[List[int]]:
    """Greedy ids of every request, in submission order (== [engine.generate_ids(e, max_new_tokens, eos_id) for e in embeds_list])."""
    sch = ClipScheduler(engine, eos_id, max_active, chunk, max_prefill_rows)
    rids = [sch.submit(e, max_new_tokens) for e in embeds_list]
    out = sch.run()
    return [out[r] for r in rids]
