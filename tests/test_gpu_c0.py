"""-m gpu: BASELINE configs[0] (Phi-3.5, 8 frames / 1 segment) at REAL width and depth against a golden produced by the
reference's own modules on CPU in fp32 (oracle/make_golden.py c0): CLIP 23 of 24 layers, InternVideo2 39 of 40 blocks at
S = 2049, encode_images + splice, Phi-3.5 32 layers with the O(n^2) greedy of 12 tokens.  The weights are synth.exact_tensor
streams regenerated ON THE GPU (bit-identical to what the reference consumed on the CPU), so the fixture holds outputs only.

Tolerances (of the output scale; north_star: bf16 logits within 1e-2 rel).  The golden also stores the reference's own modules
evaluated in bf16 on the CPU (`*_bf16ref`): how far a bf16 evaluation of the REFERENCE is from its fp32 evaluation at this depth.
The HIP path must be within max(floor, 1.25 x that) on the max-abs error (round 2: 1.5 x; the largest ratio observed is 1.23) AND
within 1.15 x on the RMS error, and is also compared with the bf16 evaluation directly (gpu_util.noise_class).  Observed numbers are
printed; profiles/r03_parity_observed.txt keeps them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from gpu_util import DEV, BF16_CLASS_CAP, E2E_FP32PREFIX_CAP, bf, check, like_for_like, noise_class, rel_err  # noqa: E402
from grounded_video_llm_amd import engine as E, synth, weights as Wt  # noqa: E402


@pytest.fixture(scope="module")
def c0():
    meta, g = load_golden("c0_full")
    sd = meta["seeds"]
    geo = E.TowerGeometry(llm="phi3.5", frames_per_seg=8, max_segs=12, max_seq=4096, max_prefill=3712, kv_pages=64)   # also holds the C1 clip
    geo.rope_short, geo.rope_long = synth.longrope_factors(96)
    eng = E.Engine(geo, DEV)
    W = synth.clip_weights(seed=sd["clip"], device=DEV, exact=True)
    eng.load_packed(Wt.pack_clip(W, geo.clip_layers - 1)); del W
    W = synth.iv2_weights(seed=sd["iv2"], device=DEV, exact=True)
    eng.load_packed(Wt.pack_iv2(W, geo.iv2_depth - 1, 8)); del W
    W = synth.projector_weights("phi3.5", seed=sd["proj"], device=DEV, exact=True)
    eng.load_packed(Wt.pack_projectors(W, "phi3.5")); del W
    W = synth.llm_weights("phi3", seed=sd["llm"], device=DEV, exact=True)
    eng.load_packed(Wt.pack_llm(W, "phi3", geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta, geo.rope_short, geo.rope_long)); del W
    torch.cuda.empty_cache()
    eng.finalize()
    sp = synth.exact_tensor(sd["sp"], (1, 1, 3, 336, 336), device=DEV)[0]
    tp = synth.exact_tensor(sd["tp"], (1, 8, 3, 224, 224), device=DEV)
    tseg = tp.reshape(1, 1, 8, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1).contiguous()
    yield eng, geo, meta, g, sp, tseg
    eng.close()


def test_exact_tensor_is_bit_identical_on_the_gpu():
    for name, shape, std, mean in (("c0.llm/3.qkv", (9216, 3072), 3072 ** -0.5, 0.0), ("c0.clip/preln.w", (1024,), 0.1, 1.0), ("c0.tp", (1, 8, 3, 224, 224), 1.0, 0.0)):
        a = synth.exact_tensor(name, shape, std, mean, device=DEV).cpu()
        b = synth.exact_tensor(name, shape, std, mean)
        assert torch.equal(a, b), name


def test_c0_clip_23_layers(c0):
    eng, geo, meta, g, sp, tseg = c0
    st = meta["stride"]["clip"]
    got = eng.clip_encode(sp)
    check(got[:, ::st[0], ::st[1]], g["clip_penultimate"], 1e-2, "C0 CLIP ViT-L/14-336, 23 layers, vs reference (fp32)")


def test_c0_internvideo2_39_blocks(c0):
    eng, geo, meta, g, sp, tseg = c0
    st = meta["stride"]["iv2"]
    got = eng.iv2_encode(tseg)
    noise_class(got[:, ::st[0], ::st[1]], g["iv2_out"], g["iv2_out_bf16ref"], "C0 InternVideo2-1B, 39 blocks, S=2049")


def test_c0_encode_images_splice_prefill_greedy(c0):
    eng, geo, meta, g, sp, tseg = c0
    st = meta["stride"]
    vis = eng.encode_segments(sp, tseg)
    assert vis.shape == (285, 3072)
    check(vis[None][:, :, ::st["feats"][1]], g["feats"], 1e-2, "C0 encode_images (285 visual tokens) vs reference (fp32)")
    ids = meta["ids"]
    emb = eng.splice(ids, vis)
    S = meta["S"]
    assert emb.shape[0] == S
    check(emb[None][:, ::st["emb"][0], ::st["emb"][1]], g["emb"], 1e-2, "C0 spliced inputs_embeds vs reference (fp32)")
    # ---- prefill: last-row logits of the 32-layer Phi-3.5 on the HIP path's OWN visual prefix (end to end)
    scale = float(np.abs(g["logits_steps"]).max())
    ref_bf = float(np.abs(g["logits_steps_bf16ref"] - g["logits_steps"]).max()) / scale
    tol = max(1e-2, E2E_FP32PREFIX_CAP * ref_bf)
    print(f"[parity] the reference's own bf16 Phi-3.5 (32 L) is {ref_bf:.3e} from its fp32 logits (scale {scale:.3f}); tolerance {tol:.2e}")
    seq = eng.seq_alloc(S + 32)
    lg = eng.prefill(seq, emb, want_logits=True).clone()
    e0 = float((lg.cpu().double() - torch.as_tensor(g["logits_step0"]).double()).abs().max()) / scale
    print(f"[parity] C0 end-to-end prefill logits (step 0): {e0:.3e} of the logit scale")
    assert e0 <= tol
    # ---- teacher-forced decode on the reference's greedy ids: logits of every step through the paged KV cache
    gold_ids = meta["greedy_ids"]
    ls = meta["stride"]["logits"]
    errs = [e0]
    rows_c0 = [lg]
    for step in range(1, meta["new_tokens"]):
        ld = eng.decode_step_logits(seq, gold_ids[step - 1])
        rows_c0.append(ld.clone())
        errs.append(float((ld[::ls].cpu().double() - torch.as_tensor(g["logits_steps"][step]).double()).abs().max()) / scale)
    eng.seq_free(seq)
    print("[parity] C0 teacher-forced decode logits per step (of the logit scale):", " ".join(f"{e:.2e}" for e in errs))
    assert max(errs) <= tol
    # (fp32-PREFIX yardstick, kept from rounds 2-4: the reference's LLM in bf16 on its own fp32 visual prefix -- one error source fewer than ANY end-to-end bf16 path,
    #  the reference's included; not like for like, hence the wider RMS cap.  The binding statement is like_for_like() below.)
    noise_class(torch.stack([r[::ls].cpu() for r in rows_c0]), g["logits_steps"], g["logits_steps_bf16ref"], "C0 Phi-3.5 32 L logits, prefill row + 11 teacher-forced decode rows", cap=E2E_FP32PREFIX_CAP, rms_cap=1.20)
    # round 5, the like-for-like statement: the reference evaluated the way its GPU path runs (bf16 vision prefix feeding the bf16 LLM), caps 1.15 / 1.10
    like_for_like(torch.stack([r[::ls].cpu() for r in rows_c0]), "c0", "C0  Phi-3.5, 8 frames, S=384 (end to end)")
    # ---- free-running greedy: same ids as the reference wherever its top-1 margin exceeds the logit tolerance
    got_ids = eng.generate_ids(emb, meta["new_tokens"], None)
    margins = np.asarray(g["top1"]) - np.asarray(g["top2"])
    print("[parity] C0 greedy ids", got_ids, "reference", gold_ids, "margins/scale", np.round(margins / scale, 4).tolist())
    for i, (a, b) in enumerate(zip(got_ids, gold_ids)):
        if a != b:
            assert margins[i] < 2 * tol * scale, f"greedy id differs at step {i} although the reference's margin is {margins[i] / scale:.3e} of the scale"
            assert a == meta["second_ids"][i], "diverged to something other than the reference's runner-up"
            break
    else:
        assert len(got_ids) == meta["new_tokens"]


def test_c0_llm_on_the_reference_fp32_prefix(c0):
    """end to end, HIP's logits carry 7-13 % more RMS error than the reference's own bf16 evaluation.  The golden's bf16
    evaluation ran the LLM on the reference's FP32 visual prefix; HIP's end-to-end run feeds its LLM its own bf16 vision output.  Here the
    HIP LLM gets the same input the golden's bf16 LLM got: the fp32 prefix, recomputed on the host by the oracle (pinned to the reference at
    C0's full depth: 0 / 1e-6, tests/test_oracle_c0_slow.py; ~25 s of CPU) and checked against the golden's strided `emb` first.  If the
    excess is the prefix's, the RMS ratio falls to ~1 here; if the LLM kernels add noise of their own, it stays.  The observed ratios are printed
    (profiles/r04_parity_observed.txt) and bounded."""
    import gvl_oracle as O
    eng, geo, meta, g, sp, tseg = c0
    sd = meta["seeds"]
    cpu = lambda W: {k: v.cpu() for k, v in W.items()}
    Wc = cpu(synth.clip_weights(seed=sd["clip"], device=DEV, exact=True))
    Wv = cpu(synth.iv2_weights(seed=sd["iv2"], device=DEV, exact=True))
    Wp = cpu(synth.projector_weights("phi3.5", seed=sd["proj"], device=DEV, exact=True))
    Wl_embed = synth.exact_tensor(sd["llm"] + "/embed", (32366, 3072), 0.5, 0.0, DEV).cpu()      # synth.llm_weight_specs: the embedding table alone
    torch.cuda.empty_cache()
    tp = synth.exact_tensor(sd["tp"], (1, 8, 3, 224, 224), device=DEV).cpu()
    with torch.no_grad():
        vis = O.encode_images(sp[None].cpu(), tp, Wc, Wv, Wp, "phi3.5")[0]                  # fp32, the reference's arithmetic
        emb32 = O.splice(torch.tensor(meta["ids"]), vis, Wl_embed)
    st = meta["stride"]["emb"]
    check(emb32[None][:, ::st[0], ::st[1]], g["emb"], 1e-5, "oracle fp32 prefix vs the reference's inputs_embeds (golden)")
    S = meta["S"]
    emb = emb32.to(bf).to(DEV)
    seq = eng.seq_alloc(S + 32)
    ls = meta["stride"]["logits"]
    rows = [eng.prefill(seq, emb, want_logits=True).clone()]
    for step in range(1, meta["new_tokens"]):
        rows.append(eng.decode_step_logits(seq, meta["greedy_ids"][step - 1]).clone())
    eng.seq_free(seq)
    noise_class(torch.stack([r[::ls].cpu() for r in rows]), g["logits_steps"], g["logits_steps_bf16ref"],
                "C0 Phi-3.5 32 L logits on the REFERENCE's fp32 prefix (LLM kernels alone), prefill row + 11 teacher-forced decode rows", rms_cap=1.08)


def test_c1_headline_config_96_frames_vs_reference_golden(c0):
    """BASELINE configs[1] -- the configuration bench.py measures (Phi-3.5, 96 frames / 12 segments, S = 3519) -- end to end against the
    REFERENCE at real size (tests/golden/c1_full.npz, oracle/make_golden.py c1: the reference's own 12-segment encode_images,
    prepare_multimodal_inputs and ONE fp32 Phi3ForCausalLM forward over the prefix plus 11 teacher-forced tokens; same weights as the
    C0 golden).  HIP path: encode -> splice -> prefill (row S-1) -> 11 teacher-forced decode steps through the paged KV cache."""
    eng, geo, _, _, _, _ = c0
    meta, g = load_golden("c1_full")
    sd, st = meta["seeds"], meta["stride"]
    sp = synth.exact_tensor(sd["sp"], (1, 12, 3, 336, 336), device=DEV)[0]
    tp = synth.exact_tensor(sd["tp"], (1, 96, 3, 224, 224), device=DEV)
    tseg = tp.reshape(1, 12, 8, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1).contiguous()
    vis = eng.encode_segments(sp, tseg)
    assert vis.shape == (12 * 285, 3072)
    check(vis[None][:, ::st["feats"][0], ::st["feats"][1]], g["feats"], 1e-2, "C1 encode_images (12 segments, 3420 visual tokens) vs reference (fp32)")
    emb = eng.splice(meta["ids"], vis)
    S = meta["S"]
    assert emb.shape[0] == S == 3519
    scale = float(np.abs(g["logits_rows"]).max())
    ref_bf = float(np.abs(g["logits_rows_bf16ref"] - g["logits_rows"]).max()) / scale
    tol = max(1e-2, E2E_FP32PREFIX_CAP * ref_bf)
    ls = st["logits"]
    seq = eng.seq_alloc(S + 32)
    rows = [eng.prefill(seq, emb, want_logits=True).clone()]
    for tok in meta["forced"]:
        rows.append(eng.decode_step_logits(seq, tok).clone())
    eng.seq_free(seq)
    errs = [float((r[::ls].cpu().double() - torch.as_tensor(g["logits_rows"][i]).double()).abs().max()) / scale for i, r in enumerate(rows)]
    print(f"[parity] C1 Phi-3.5 32 L, S={S}: the reference's own bf16 evaluation is {ref_bf:.3e} from its fp32 logits (scale {scale:.3f}); bound {tol:.2e}")
    print("[parity] C1 logits, prefill row + 11 teacher-forced decode rows (of the logit scale):", " ".join(f"{e:.2e}" for e in errs))
    assert max(errs) <= tol
    noise_class(torch.stack([r[::ls].cpu() for r in rows]), g["logits_rows"], g["logits_rows_bf16ref"], f"C1 Phi-3.5 32 L logits, S={S}, prefill row + 11 decode rows", cap=E2E_FP32PREFIX_CAP, rms_cap=1.20)   # fp32-prefix yardstick (see C0)
    like_for_like(torch.stack([r[::ls].cpu() for r in rows]), "c1", "C1  Phi-3.5, 96 frames, S=3519 (end to end; the headline configuration)")
    margins = np.asarray(g["top1"]) - np.asarray(g["top2"])
    for i, r in enumerate(rows):
        if margins[i] > 2 * tol * scale:
            assert int(r.argmax()) == meta["argmax"][i], f"row {i}: argmax differs although the reference's margin is {margins[i] / scale:.3e} of the scale"


def test_c1_free_running_greedy_vs_oracle_continuation(c0):
    """Free-running greedy on the HEADLINE configuration (id equality beyond C0).  tests/golden/c1_free.json: the reference's own
    encode_images + prepare_multimodal_inputs give the fp32 prefix of the C1 clip, the continuation is the oracle's KV-cached fp32 greedy (pinned
    against the reference's O(n^2) greedy at this depth by tests/test_oracle_c0_slow.py).  The HIP path runs the whole clip itself (towers ->
    splice -> prefill -> paged-KV decode) and must produce the same 10 ids; a difference is tolerated only at a step whose top-1 / top-2 margin is
    below twice the bf16 noise of the logits (then the two greedy paths legitimately part ways)."""
    import json, os
    from conftest import GOLDEN
    eng, geo, _, _, _, _ = c0
    fr = json.load(open(os.path.join(GOLDEN, "c1_free.json")))
    sd = fr["seeds"]
    sp = synth.exact_tensor(sd["sp"], (1, 12, 3, 336, 336), device=DEV)[0]
    tp = synth.exact_tensor(sd["tp"], (1, 96, 3, 224, 224), device=DEV)
    tseg = tp.reshape(1, 12, 8, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1).contiguous()
    emb = eng.splice(fr["ids"], eng.encode_segments(sp, tseg))
    assert emb.shape[0] == fr["S"]
    got = eng.generate_ids(emb, len(fr["free_ids"]), None)
    rel = [m / fr["scale"] for m in fr["margins"]]
    n_same = next((i for i, (a, b) in enumerate(zip(got, fr["free_ids"])) if a != b), len(got))
    print(f"[parity] C1 free-running greedy: {n_same} of {len(got)} ids equal the fp32 continuation; ids {got} vs {fr['free_ids']}; margins/scale {[round(r, 4) for r in rel]}")
    if n_same < len(got):
        assert rel[n_same] < 2 * 2.1e-2, f"greedy id differs at step {n_same} although the margin is {rel[n_same]:.3e} of the logit scale"


def test_c1_free_running_greedy_equals_the_reference_continuations_without_a_near_tie_clause(c0):
    """VERDICT r5 #2.  tests/golden/c1_free2.json (oracle/make_golden.py free2_c1): the headline clip's fp32 prefix from the reference's own encode_images +
    prepare_multimodal_inputs, several prompt TAILS, and for each the leading greedy decisions whose top-1 / top-2 margin is >= 6e-2 of the logit scale -- three
    times the bf16 noise of a full-depth evaluation.  A random-init decoder settles into a short cycle within two or three tokens whatever the prompt, so the
    informative decisions are the first ones of each tail; every kept id must be reproduced exactly, no escape clause."""
    import json, os
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, "c1_free2.json")
    if not os.path.exists(path):
        pytest.skip("c1_free2.json not generated")
    eng, geo, _, _, _, _ = c0
    fr = json.load(open(path))
    sd = fr["seeds"]
    sp = synth.exact_tensor(sd["sp"], (1, 12, 3, 336, 336), device=DEV)[0]
    tp = synth.exact_tensor(sd["tp"], (1, 96, 3, 224, 224), device=DEV)
    tseg = tp.reshape(1, 12, 8, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1).contiguous()
    vis = eng.encode_segments(sp, tseg)
    n_dec, pairs = 0, set()
    for t in fr["tails"]:
        emb = eng.splice(t["ids"], vis)
        assert emb.shape[0] == fr["S"]
        got = eng.generate_ids(emb, len(t["free_ids"]), None)
        assert min(m / s for m, s in zip(t["margins"], t["scales"])) >= fr["criteria"]["min_margin_over_scale"]
        assert got == t["free_ids"], f"tail seed {t['tail_seed']}: {got} vs the reference's fp32 continuation {t['free_ids']}"
        assert t["free_ids_bf16emu"] == t["free_ids"]
        n_dec += len(got); pairs |= {(i, g) for i, g in enumerate(got)}
    print(f"[parity] C1 free-running greedy, {len(fr['tails'])} prompt tails: {n_dec} of {n_dec} ids equal the fp32 continuations ({len(pairs)} distinct (step, id) decisions; every margin >= 6e-2 of the scale; no near-tie clause)")


def test_kernel_form_of_the_big_gemm_changes_no_bit_at_full_size(c0):
    """gvl_debug_set("gemm_a4"): 0 = every 256 x 256 GEMM on the 8-wave ping-pong kernel (round 5), 1 = the per-epilogue choice (4-wave / pipelined 4-wave / 8-wave),
    2 / 3 = the plain / pipelined 4-wave kernel wherever it serves.  Every form accumulates an output element in the same k order and shares the epilogue arithmetic, so
    the full-size towers (InternVideo2: rowscale, erf-GELU, LayerScale + residual + row statistics; CLIP: bias) and the 32-layer prefill logits must be BIT-identical."""
    eng, geo, meta, g, sp, tseg = c0
    outs = {}
    try:
        for mode in (0, 1, 2, 3, 10, 12):              # 10 / 12: the default form with gemm_narrow = 0 / 2 (full-width half tiles / narrow tiles on the fixed tile walk)
            eng.debug_set("gemm_a4", 1 if mode >= 10 else mode)
            eng.debug_set("gemm_narrow", mode - 10 if mode >= 10 else 1)
            vis = eng.encode_segments(sp, tseg)
            emb = eng.splice(meta["ids"], vis)
            seq = eng.seq_alloc(emb.shape[0] + 8)
            lg = eng.prefill(seq, emb, want_logits=True).clone()
            eng.seq_free(seq)
            outs[mode] = (vis.clone(), lg)
    finally:
        eng.debug_set("gemm_a4", 1)
        eng.debug_set("gemm_narrow", 1)
    for mode in (1, 2, 3, 10, 12):
        assert torch.equal(outs[0][0], outs[mode][0]), f"gemm_a4 = {mode}: visual tokens differ from the 8-wave kernel's"
        assert torch.equal(outs[0][1], outs[mode][1]), f"gemm_a4 = {mode}: prefill logits differ from the 8-wave kernel's"
    print("[parity] full-size towers + 32-layer prefill: bit-identical under gemm_a4 = 0 / 1 / 2 / 3 (8-wave, chosen, plain 4-wave, pipelined 4-wave GEMM forms) and gemm_narrow = 0 / 1 / 2")
