"""Beam search as the reference gets it from HF `generate(num_beams=k, do_sample=False)` (inference.py:46,170-176 ->
models/llava_next_video.py:655-661 -> transformers 4.40.1 GenerationMixin._beam_search + BeamSearchScorer / BeamHypotheses [ext],
the version requirements.txt pins): host-side bookkeeping only -- the log-softmax / top-2k of every step and the KV cache
live with the caller (`step`).  Restated, not imported: the installed transformers (5.x) no longer ships BeamSearchScorer; the
CPU test drives this function and the installed `generate()` with the same tiny HF model and compares the sequences.

Semantics kept from 4.40.1 (batch of one, one beam group):
  * beam_scores start as [0, -1e9, ...]; per step the log-softmax of every running beam plus its score is flattened, the best
    2k (token, beam) pairs are visited in order: an eos candidate of rank < k closes a hypothesis (score = sum of log-probs /
    generated_len ** length_penalty, generated_len counting the eos), of rank >= k is skipped; the first k non-eos candidates are the
    next running beams;
  * the search is done when k hypotheses exist and (early_stopping, or the worst of them is at least the best running
    sum-log-prob / cur_len ** length_penalty); at max_new_tokens the running beams are added as hypotheses;
  * the answer is the best hypothesis, with eos appended when it ended by eos.

Beam-SAMPLE (`generate(num_beams=k, do_sample=True)`; reachable from inference.py:45-49,170-176; 4.40.1 GenerationMixin._beam_sample): the same
bookkeeping, but the 2k candidates of a step are DRAWN: every running beam's log-softmax goes through the warpers (temperature -> top-k ->
top-p, min_tokens_to_keep = 2 as HF sets it for num_beams > 1), the beam scores are added, and 2k (token, beam) pairs are sampled without
replacement from softmax over the k x V grid, then visited in order of their score.  torch.multinomial's random stream cannot be reproduced
across implementations (the same caveat as plain sampling, include/gvl.h): parity = same candidate distribution; draws come from a seeded
torch.Generator, so a run is reproducible under `seed`.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch


class _Hyps:
    """BeamHypotheses of transformers 4.40.1 (generation/beam_search.py) for one batch item."""

    def __init__(self, num_beams: int, length_penalty: float, early_stopping):
        self.k, self.lp, self.early = num_beams, length_penalty, early_stopping
        self.beams: List[Tuple[float, List[int], bool]] = []
        self.worst = 1e9

    def add(self, toks: Sequence[int], sum_logprobs: float, generated_len: int, by_eos: bool):
        score = sum_logprobs / (generated_len ** self.lp)
        if len(self.beams) < self.k or score > self.worst:
            self.beams.append((score, list(toks), by_eos))
            if len(self.beams) > self.k:
                order = sorted((s, i) for i, (s, _, _) in enumerate(self.beams))
                del self.beams[order[0][1]]
                self.worst = order[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self.beams) < self.k:
            return False
        if self.early is True:
            return True
        if self.early is False:
            return self.worst >= best_sum_logprobs / (cur_len ** self.lp)
        # "never": a heuristic-free upper bound on what a running beam can still reach
        if self.lp > 0.0:
            raise ValueError("early_stopping='never' with length_penalty > 0 needs max_length; not used by the reference")
        return self.worst >= best_sum_logprobs / (cur_len ** self.lp)


def warp_scores(scores: torch.Tensor, temperature: float = 1.0, top_k: Optional[int] = 50, top_p: Optional[float] = None, min_keep: int = 2) -> torch.Tensor:
    """transformers 4.40.1 logits warpers in generate()'s order on `scores` [rows, V] (here: log-probabilities): TemperatureLogitsWarper (scores / T),
    TopKLogitsWarper (everything below the k-th largest -> -inf; k = max(top_k, min_keep)), TopPLogitsWarper (ascending sort, drop the tokens whose
    cumulative probability stays <= 1 - top_p, never the last min_keep)."""
    s = scores.float()
    if temperature is not None and temperature != 1.0:
        s = s / float(temperature)
    if top_k is not None and top_k > 0:
        kk = min(max(int(top_k), min_keep), s.shape[-1])
        s = s.masked_fill(s < torch.topk(s, kk).values[..., -1, None], float("-inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(s, descending=False)
        remove = srt.softmax(dim=-1).cumsum(dim=-1) <= (1.0 - float(top_p))
        remove[..., -min_keep:] = False
        s = s.masked_fill(remove.scatter(-1, idx, remove), float("-inf"))
    return s


def beam_search(step: Callable[[List[int], List[int]], torch.Tensor], first_logits: torch.Tensor, num_beams: int, max_new_tokens: int,
                eos_id: Optional[int], length_penalty: float = 1.0, early_stopping=False, sample: Optional[dict] = None) -> List[int]:
    """first_logits [vocab]: logits after the prompt.  step(parents, tokens) -> logits [k, vocab] of the k new running beams, where new beam j
    continues old beam parents[j] with tokens[j] (the caller reorders its KV cache accordingly; at the first call every parent is 0 = the
    prompt).  Returns the NEW ids of the best hypothesis (eos included when it ended by eos), as HF does for inputs_embeds prompts.
    sample: None = beam search; dict(temperature, top_k, top_p, generator) = beam-sample (module docstring)."""
    k = int(num_beams)
    if k < 2:
        raise ValueError("beam_search needs num_beams >= 2")
    V = first_logits.shape[-1]
    if V < 2 * k:
        raise ValueError("vocabulary smaller than 2 x num_beams")
    seqs: List[List[int]] = [[] for _ in range(k)]
    # transformers 4.40.1: _beam_search starts beams 1..k-1 at -1e9 (the first step expands beam 0 only); _beam_sample starts EVERY beam at 0 -- its
    # first draw is over k identical rows, so the same token may be drawn from two rows and the running beams may start as duplicates
    scores = torch.zeros((k,), dtype=torch.float32, device=first_logits.device)
    if sample is None:
        scores[1:] = -1e9
    logits = first_logits.float().unsqueeze(0).expand(k, V)
    hyps = _Hyps(k, length_penalty, early_stopping)
    done = False
    while True:
        lp = torch.log_softmax(logits.float(), dim=-1)
        if sample is None:
            lp = lp + scores[:, None]
            top = torch.topk(lp.reshape(-1), 2 * k, largest=True, sorted=True)
            vals, idxs = top.values.tolist(), top.indices.tolist()
        else:
            lp = warp_scores(lp, sample.get("temperature", 1.0), sample.get("top_k", 50), sample.get("top_p")) + scores[:, None]
            flat = lp.reshape(-1)
            n_fin = int(torch.isfinite(flat).sum())
            if n_fin < 2 * k:                                # torch.multinomial(replacement=False) would silently hand back zero-probability indices
                raise ValueError(f"beam-sample: only {n_fin} tokens survive the warpers but 2 x num_beams = {2 * k} draws are needed "
                                 "(raise top_p / top_k / temperature, or lower num_beams)")
            picks = torch.multinomial(torch.softmax(flat, dim=-1), 2 * k, replacement=False, generator=sample.get("generator"))
            pv, order = torch.sort(flat[picks], descending=True)
            vals, idxs = pv.tolist(), picks[order].tolist()
        cur_len = len(seqs[0]) + 1
        nxt: List[Tuple[float, int, int]] = []
        for rank, (v, ix) in enumerate(zip(vals, idxs)):
            b, tok = ix // V, ix % V
            if eos_id is not None and tok == eos_id:
                if rank >= k:
                    continue
                hyps.add(seqs[b], v, cur_len, True)
            else:
                nxt.append((v, tok, b))
            if len(nxt) == k:
                break
        if len(nxt) < k:
            raise ValueError("fewer than num_beams non-eos candidates among the top 2 x num_beams")
        done = done or hyps.is_done(max(vals), cur_len)
        # first step: every row is the prompt itself (beam-sample may have drawn from rows >= 1 of its k identical rows) -> parent 0
        parents, toks = [0 if cur_len == 1 else b for _, _, b in nxt], [t for _, t, _ in nxt]
        seqs = [seqs[b] + [t] for _, t, b in nxt]
        scores = torch.tensor([v for v, _, _ in nxt], dtype=torch.float32, device=first_logits.device)
        if done or len(seqs[0]) >= max_new_tokens:
            break
        logits = step(parents, toks)
    if not done:
        for j in range(k):                                   # finalize(): the open beams become hypotheses
            hyps.add(seqs[j], float(scores[j]), len(seqs[j]), False)
    best = max(hyps.beams, key=lambda x: x[0]) if hyps.beams else (0.0, seqs[0], False)
    # python's sort is stable and HF takes sorted(...)[-1]: among equal scores the LAST added wins
    top_score = best[0]
    for h in hyps.beams:
        if h[0] == top_score:
            best = h
    out = list(best[1])
    if best[2] and eos_id is not None and len(out) < max_new_tokens:
        out.append(eos_id)
    return out
