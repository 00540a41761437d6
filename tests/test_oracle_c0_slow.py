"""CPU, opt-in (GVL_SLOW=1; ~12 minutes on 8 cores: 5.2 G synthetic weights are regenerated): the ORACLE at BASELINE configs[0]'s real
width and depth against the golden produced by the reference's own modules (tests/golden/c0_full.npz).  The default CPU suite pins the
oracle on tiny full modules and full-width single layers; this pins the stacked depth (CLIP 23 L, InternVideo2 39 blocks, Phi-3.5
32 L, KV-cached greedy vs the reference's O(n^2) greedy).  bench.py's cpu_baseline re-checks the ids on the GPU box's host."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
import gvl_oracle as O
from grounded_video_llm_amd import synth

pytestmark = pytest.mark.skipif(os.environ.get("GVL_SLOW") != "1", reason="opt-in: GVL_SLOW=1 (regenerates 5.2 G weights on the CPU)")


def test_oracle_c0_full_depth_vs_reference_golden():
    torch.set_grad_enabled(False)
    meta, g = load_golden("c0_full")
    sd, st = meta["seeds"], meta["stride"]
    sp = synth.exact_tensor(sd["sp"], (1, 1, 3, 336, 336))
    tp = synth.exact_tensor(sd["tp"], (1, 8, 3, 224, 224))
    Wc = synth.clip_weights(seed=sd["clip"], exact=True)
    clip = O.clip_penultimate(sp[0], Wc, 24, 16)
    e = float(np.abs(clip[:, ::st["clip"][0], ::st["clip"][1]].numpy() - g["clip_penultimate"]).max() / np.abs(g["clip_penultimate"]).max())
    print(f"[oracle c0] CLIP 23 L: {e:.2e}"); assert e < 1e-4
    Wv = synth.iv2_weights(seed=sd["iv2"], exact=True)
    tseg = tp.reshape(1, 1, 8, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1)
    iv2 = O.iv2_encode(tseg, Wv, 40, 16)
    e = float(np.abs(iv2[:, ::st["iv2"][0], ::st["iv2"][1]].numpy() - g["iv2_out"]).max() / np.abs(g["iv2_out"]).max())
    print(f"[oracle c0] InternVideo2 39 blocks: {e:.2e}"); assert e < 1e-4
    Wp = synth.projector_weights("phi3.5", seed=sd["proj"], exact=True)
    vis = O.encode_images(sp, tp, Wc, Wv, Wp, "phi3.5")
    e = float(np.abs(vis[:, :, ::st["feats"][1]].numpy() - g["feats"]).max() / np.abs(g["feats"]).max())
    print(f"[oracle c0] encode_images: {e:.2e}"); assert e < 1e-4
    del Wc, Wv
    Wl = synth.llm_weights("phi3", seed=sd["llm"], exact=True)
    ocfg = O.LLMConfig("phi3", 3072, 8192, 32, 32, 32, 32366, 1e-5, 10000.0, 131072, 4096, *synth.longrope_factors(96))
    emb = O.splice(torch.tensor(meta["ids"]), vis[0], Wl["model.embed_tokens.weight"])
    assert emb.shape[0] == meta["S"]
    cache = [None] * 32
    lg = O.llm_forward(ocfg, Wl, emb, False, cache, 0, last_only=True)[0]
    scale = float(np.abs(g["logits_steps"]).max())
    e = float(np.abs(lg.numpy() - g["logits_step0"]).max()) / scale
    print(f"[oracle c0] Phi-3.5 32 L prefill logits: {e:.2e}"); assert e < 1e-4
    ids = O.greedy_generate(ocfg, Wl, emb, meta["new_tokens"], None, use_cache=True)
    print("[oracle c0] KV-cached greedy ids", ids, "reference O(n^2) greedy", meta["greedy_ids"])
    assert ids == meta["greedy_ids"]
