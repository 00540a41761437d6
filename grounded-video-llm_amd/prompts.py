"""Host-side text / integer plumbing of the hot path (no tensors): frame-index sampling, prompt
templates, temporal-token <-> seconds conversion, `<image>` token splitting, left-pad/truncate.

Counterpart of inference.py:65-134, mm_utils/video_utils.py:13-51, datasets/chat/base_template.py and
models/llava_next_video.py:409-426,626-647 in the reference; behaviour (including the quirks listed in
SURVEY.md Appendix C #12, #13, #15) is pinned by tests/golden/integer_paths.json.
"""
from __future__ import annotations

import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100
DEFAULT_IMAGE_TOKEN = "<image>"
GROUNDING_TOKEN = "<timestamp_grounding>"

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)
INTERNVIDEO_MEAN = (0.485, 0.456, 0.406)
INTERNVIDEO_STD = (0.229, 0.224, 0.225)


class Template:
    """system / user / assistant string assembly for one LLM family."""

    def __init__(self, system: str, user: str, assistant: str, eos: str):
        self.system, self.user, self.assistant, self.eos = system, user, assistant, eos

    @property
    def separator(self) -> Tuple[str, str]:
        return self.assistant, self.eos

    def encode(self, conv: Sequence[Dict[str, str]]) -> str:
        questions, answers = [], []
        skip_first = 0
        for i, m in enumerate(conv):
            if i == 0 and m["from"] != "human":
                skip_first = 1
                continue
            (questions if i % 2 == skip_first else answers).append(m["value"])
        if len(questions) != len(answers):
            raise AssertionError(f"qa is not match : length_q:{len(questions)} vs length_a:{len(answers)}")
        out = ""
        for i, (q, a) in enumerate(zip(questions, answers)):
            if i == 0:
                out += self.system
            if DEFAULT_IMAGE_TOKEN in q and GROUNDING_TOKEN not in q:
                q = (DEFAULT_IMAGE_TOKEN + "\n" + q.replace(DEFAULT_IMAGE_TOKEN, "").strip()).strip()
            out += self.user + q + self.assistant + a + self.eos
        return out


_HELPFUL = ("You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, "
            "and assist the user with a variety of tasks using natural language.")
TEMPLATES = {
    "phi3.5": Template("<|system|>\nYou are a helpful AI assistant that can generate responses based on visual inputs.",
                       "\n<|user|>\n", "\n<|assistant|>\n", "<|endoftext|>"),
    "llama3": Template("<|start_header_id|>system<|end_header_id|>" + _HELPFUL, "<|start_header_id|>user<|end_header_id|>",
                       "<|start_header_id|>assistant<|end_header_id|>", "<|eot_id|>"),
    "vicuna": Template(_HELPFUL, "\nUSER: ", "\nASSISTANT: ", "</s>"),
}


def sample_frame_indices(num_frames: int, vlen: int) -> List[int]:
    """'middle' sampling: the midpoint of each of `num_frames` equal intervals; short videos pad with the last frame."""
    n = min(num_frames, vlen)
    edges = np.linspace(start=0, stop=vlen, num=n + 1).astype(int)
    idx = [int((edges[i] + edges[i + 1] - 1) // 2) for i in range(n)]
    return idx + [idx[-1]] * (num_frames - n)


def spatial_indices(num_frames: int, num_segs: int) -> List[int]:
    per = int(num_frames // num_segs)
    return [i * per + int(per / 2) for i in range(num_segs)]


def seconds_to_tokens(query: str, duration: float, num_temporal_tokens: int = 300) -> str:
    return re.sub(r"(\d+) seconds", lambda m: f"<{int(float(m.group(1)) / duration * num_temporal_tokens)}>", query)


def quantize_time(t: float, duration: float, num_temporal_tokens: int = 300) -> int:
    return min(int(num_temporal_tokens * t / duration), num_temporal_tokens)


def parse_time_interval(text: str, duration: float, num_temporal_tokens: int = 300, llm: str = "phi3.5") -> str:
    fmt = {"phi3.5": " {:.2f} seconds", "llama3": "{:.2f} seconds"}
    if llm not in fmt:
        raise ValueError(f"parse_time_interval: unsupported llm {llm!r}")
    return re.sub(r"<(\d+)>", lambda m: fmt[llm].format(duration * int(m.group(1)) / num_temporal_tokens), text)


def build_prompt(llm: str, mode: str, text: str, duration: float = 0.0, num_temporal_tokens: int = 300) -> str:
    if mode == "grounding":
        value = DEFAULT_IMAGE_TOKEN + " " + GROUNDING_TOKEN + "\n" + text
    elif mode == "qa":
        value = DEFAULT_IMAGE_TOKEN + "\n" + text
    elif mode == "referring":
        value = DEFAULT_IMAGE_TOKEN + "\n" + seconds_to_tokens(text, duration, num_temporal_tokens)
    else:
        raise ValueError(f"mode must be one of qa/grounding/referring, got {mode!r}")
    t = TEMPLATES[llm]
    return t.encode([{"from": "human", "value": value}, {"from": "gpt", "value": ""}]).replace(t.eos, "")


def temporal_token_strings(num_temporal_tokens: int = 300) -> List[str]:
    """The strings added to the tokenizer, in order: <0>..<N> then <timestamp_grounding> (302 for N=300)."""
    return [f"<{i}>" for i in range(num_temporal_tokens + 1)] + [GROUNDING_TOKEN]


def tokenize_with_image(prompt: str, tokenize: Callable[[str], List[int]], bos_token_id: Optional[int]) -> List[int]:
    """Tokenise the text around every `<image>` and put IMAGE_TOKEN_INDEX in between (one BOS at most)."""
    chunks = [list(tokenize(c)) for c in prompt.split(DEFAULT_IMAGE_TOKEN)]
    has_bos = bool(chunks) and bool(chunks[0]) and chunks[0][0] == bos_token_id
    off = 1 if has_bos else 0
    ids: List[int] = [chunks[0][0]] if has_bos else []
    for j, ch in enumerate(chunks):
        if j > 0:
            ids.extend(([IMAGE_TOKEN_INDEX] * (off + 1))[off:])
        ids.extend(ch[off:])
    return ids


def left_pad_truncate(batch_ids: Sequence[Sequence[int]], pad_id: int, max_txt_len: int):
    """Left-pad to a rectangle and keep the LAST max_txt_len columns (text is cut before the visual splice)."""
    width = max(len(x) for x in batch_ids)
    ids = np.full((len(batch_ids), width), pad_id, dtype=np.int64)
    mask = np.zeros((len(batch_ids), width), dtype=np.int64)
    for i, x in enumerate(batch_ids):
        if len(x):
            ids[i, width - len(x):] = np.asarray(x, dtype=np.int64)
            mask[i, width - len(x):] = 1
    if width > max_txt_len:
        ids, mask = ids[:, width - max_txt_len:], mask[:, width - max_txt_len:]
    return ids, mask


# ---- training forward: label masking, right-padded batch, label / mask splice -------------------------
# Per-family corrections of the reference's mask arithmetic (llava_next_video.py:344-407): tokens to take off the instruction
# length, and the drift it applies to BOTH lengths from the second round on.
_LABEL_RULE = {"llama3": (1, 0), "vicuna": (2, -1), "phi3.5": (2, +1)}


def make_labels(llm: str, input_ids: Sequence[int], prompt: str, tokenize: Callable[[str], List[int]], bos_token_id: Optional[int]) -> np.ndarray:
    """Labels of one conversation: a copy of input_ids where everything except the assistant answers (+ their eos) is
    IGNORE_INDEX.  Round / instruction lengths are measured by re-tokenising the text pieces, like the reference does; numpy
    slice semantics equal torch's, so degenerate (negative) lengths behave identically."""
    if llm not in _LABEL_RULE:
        raise ValueError(f"unknown llm {llm!r}")
    minus, drift = _LABEL_RULE[llm]
    sep, eos_token = TEMPLATES[llm].separator
    n_tok = lambda text: len(tokenize_with_image(text, tokenize, bos_token_id))
    labels = np.asarray(list(input_ids), dtype=np.int64).copy()
    cur = 1                                                   # the bos slot
    labels[:cur] = IGNORE_INDEX
    for i, rnd in enumerate(prompt.split(eos_token)):
        pieces = rnd.split(sep) if rnd != "" else []
        if len(pieces) != 2:
            break
        d = drift if i >= 1 else 0
        round_len = n_tok(rnd) + d                            # + eos - bos cancel
        instr_len = n_tok(pieces[0] + sep) - minus + d
        labels[cur: cur + instr_len] = IGNORE_INDEX
        cur += round_len
    labels[cur:] = IGNORE_INDEX
    return labels


def prepare_batch(llm: str, texts: Sequence[str], tokenize, bos_token_id: Optional[int], pad_token_id: int, eos_token_id: int, max_txt_len: int):
    """ids / labels / attention mask of a training batch: RIGHT padded (pad id, IGNORE_INDEX, 0), cut to max_txt_len columns;
    when the cut happens the reference writes eos into the last label column of every row (llava_next_video.py:445-450)."""
    ids = [tokenize_with_image(t, tokenize, bos_token_id) for t in texts]
    width = max(len(x) for x in ids)
    bi = np.full((len(ids), width), pad_token_id, dtype=np.int64)
    bl = np.full((len(ids), width), IGNORE_INDEX, dtype=np.int64)
    bm = np.zeros((len(ids), width), dtype=np.int64)
    for r, (x, t) in enumerate(zip(ids, texts)):
        bi[r, :len(x)] = x
        bl[r, :len(x)] = make_labels(llm, x, t, tokenize, bos_token_id)
        bm[r, :len(x)] = 1
    if width > max_txt_len:
        bi, bl, bm = bi[:, :max_txt_len], bl[:, :max_txt_len].copy(), bm[:, :max_txt_len]
        bl[:, -1] = eos_token_id
    return bi, bl, bm


def splice_labels(input_ids: np.ndarray, labels: np.ndarray, mask: np.ndarray, n_visual: int, is_text: bool):
    """Labels / mask after the visual rows took the place of the `<image>` id: visual rows carry IGNORE_INDEX; for a text-only
    sample (video_ids == 'text') the dummy visual rows go to the END with mask 0 (llava_next_video.py:583-590)."""
    where = np.flatnonzero(np.asarray(input_ids) == IMAGE_TOKEN_INDEX)
    if where.size != 1:
        raise ValueError(f"expected exactly one <image> id, found {where.size}")
    k = int(where[0])
    ign = np.full(n_visual, IGNORE_INDEX, dtype=np.int64)
    if is_text:
        return (np.concatenate([labels[:k], labels[k + 1:], ign]), np.concatenate([mask[:k], mask[k + 1:], np.zeros(n_visual, dtype=np.int64)]))
    return (np.concatenate([labels[:k], ign, labels[k + 1:]]), np.concatenate([mask[:k], np.ones(n_visual, dtype=np.int64), mask[k + 1:]]))
