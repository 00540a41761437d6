"""Pins oracle/gvl_oracle.py (the CPU restatement) against outputs of the reference's own modules
(tests/golden/*, produced by oracle/make_golden.py in the build container)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import gvl_oracle as O
from grounded_video_llm_amd import synth
from conftest import load_golden, GOLDEN


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max()
    ref = np.abs(b).max() + 1e-12
    assert err <= tol * ref, f"max err {err:.3e} vs scale {ref:.3e}"


def test_integer_paths():
    g = json.load(open(os.path.join(GOLDEN, "integer_paths.json")))
    for k, v in g["frame_indices"].items():
        n, vlen = map(int, k.split("_"))
        assert O.get_frame_indices(n, vlen, "middle") == v, k
    assert O.get_frame_indices(96, 2880)[:2] == [14, 44] and O.get_frame_indices(96, 2880)[-1] == 2864
    for k, v in g["prompts"].items():
        llm, mode = k.split("|")
        assert O.build_prompt(llm, mode, g["prompt_text"]) == v, k
    for k, v in g["parse_time_interval"].items():
        llm, txt, dur = k.split("|")
        assert O.parse_time_interval(txt, float(dur), 300, llm) == v, k
    for k, v in g["tokenizer_image_token"].items():
        name, pr = k.split("|", 1)
        def tok(s, nb=(name == "nobos")):
            ids = [3 + (sum(map(ord, w)) % 90) for w in s.split()]
            return ids if nb else [1] + ids
        assert O.tokenizer_image_token(pr, tok, 1) == v, k
    # SURVEY §8 a14 goldens
    assert O.parse_time_interval("From <36> to <64>.", 118.3, 300, "phi3.5") == "From  14.20 seconds to  25.24 seconds."
    assert O.seconds_to_temporal_tokens("What is happening from 70 seconds to 80 seconds?", 118.3) == "What is happening from <177> to <202>?"
    assert O.spatial_frame_indices(96, 12) == [4 + 8 * i for i in range(12)]
    rng = np.random.default_rng(0)
    for t, d in zip(rng.uniform(0, 500, 2000), rng.uniform(1, 500, 2000)):
        t = min(t, d)
        assert O.quantize_timestamp(float(t), float(d)) == min(int(300 * float(t) / float(d)), 300)


def test_left_pad_truncate():
    ids, mask = O.left_pad_truncate([[1, 2, 3], [4, 5, 6, 7, 8]], pad_id=0, max_txt_len=4)
    assert ids.tolist() == [[0, 1, 2, 3], [5, 6, 7, 8]] and mask.tolist() == [[0, 1, 1, 1], [1, 1, 1, 1]]


def test_clip_tiny():
    meta, g = load_golden("clip_tiny")
    c = meta["cfg"]
    W = synth.clip_weights(c["hidden"], c["inter"], c["layers"], c["image"], c["patch"], seed=meta["seed"])
    px = synth.det_tensor(meta["px"], meta["px_shape"])
    _close(O.clip_embeddings(px, W), g["embed"], 1e-5)
    _close(O.clip_penultimate(px, W, c["layers"], c["heads"]), g["penultimate"], 2e-5)


def test_clip_full_layer():
    meta, g = load_golden("clip_full_layer")
    c = meta["cfg"]
    W = synth.clip_weights(c["hidden"], c["inter"], c["layers"], c["image"], c["patch"], seed=meta["seed"])
    px = synth.det_tensor(meta["px"], meta["px_shape"])
    y = O.clip_penultimate(px, W, c["layers"], c["heads"])
    s = meta["stride"]
    _close(y[:, ::s[0], ::s[1]], g["penultimate"], 5e-5)
    ye = O.clip_penultimate(px, W, c["layers"], c["heads"], emu=True)
    _close(ye[:, ::s[0], ::s[1]], g["penultimate"], 3e-2)      # bf16-emulated path stays within bf16 noise of fp32


def test_iv2_tiny_and_pos_interp():
    meta, g = load_golden("iv2_tiny")
    c = meta["cfg"]
    W = synth.iv2_weights(c["dim"], c["inter"], c["depth"], c["frames"], c["image"], 14, seed=meta["seed"])
    px = synth.det_tensor(meta["px"], meta["px_shape"])
    _close(O.iv2_encode(px, W, c["depth"], c["heads"]), g["out"], 2e-5)
    _close(O.iv2_encode(px, W, c["depth"], c["heads"], emu=True), g["out_bf16"], 3e-2)   # vs the reference run in bf16 on CPU
    meta, g = load_golden("iv2_pos_interp")
    src = synth.det_tensor(meta["src"], meta["src_shape"])
    _close(O.interpolate_pos_embed_t(src, meta["orig_t"], meta["new_t"]), g["pos"], 1e-6)


def test_iv2_full_block():
    meta, g = load_golden("iv2_full_block")
    c = meta["cfg"]
    W = synth.iv2_weights(c["dim"], c["inter"], c["depth"], c["frames"], c["image"], 14, seed=meta["seed"])
    px = synth.det_tensor(meta["px"], meta["px_shape"])
    y = O.iv2_encode(px, W, c["depth"], c["heads"])
    s = meta["stride"]
    _close(y[:, ::s[0], ::s[1]], g["out"], 5e-5)


def _cfg(c, long=True):
    short, lng = synth.longrope_factors(c["hidden"] // c["heads"])
    if c["kind"] == "phi3":
        return O.LLMConfig("phi3", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], 1e-5, 10000.0,
                           131072, 4096, short, lng)
    return O.LLMConfig("llama", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], 1e-5,
                       c.get("rope_theta", 500000.0))


def test_phi3_tiny_logits_long_and_greedy():
    meta, g = load_golden("phi3_tiny")
    c = meta["cfg"]
    cfg = _cfg(c)
    W = synth.llm_weights("phi3", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], True, seed=meta["seed"])
    x = synth.det_tensor(meta["x"], meta["x_shape"], 0.5)[0]
    _close(O.llm_forward(cfg, W, x), g["logits"][0], 2e-5)
    xl = synth.det_tensor(meta["xl"], meta["xl_shape"], 0.5)[0]
    _close(O.llm_forward(cfg, W, xl)[-4:], g["logits_long"][0], 1e-4)      # S=4100 > 4096 -> long factors
    ids_full = O.greedy_generate(cfg, W, x, 16, None, use_cache=False)
    ids_kv, margins = O.greedy_generate(cfg, W, x, 16, None, use_cache=True, return_margins=True)
    assert ids_full == g["greedy_ids"].tolist()
    assert ids_kv == g["greedy_ids"].tolist()
    np.testing.assert_allclose(margins, g["greedy_margins"], rtol=0, atol=1e-4)


def test_phi3_full_layer():
    meta, g = load_golden("phi3_full_layer")
    c = meta["cfg"]
    W = synth.llm_weights("phi3", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], True, seed=meta["seed"])
    x = synth.det_tensor(meta["x"], meta["x_shape"], 0.5)[0]
    _close(O.llm_forward(_cfg(c), W, x), g["logits"][0], 5e-5)


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_lora_unmerged_peft_forward_and_merge_vs_reference_golden(tag):
    """a11' (SURVEY §8): tests/golden/lora_{tiny,full}.npz = the REFERENCE's Phi3ForCausalLM with its four target projections wrapped in
    peft 0.3.0's LoRA Linear (restated in oracle/make_golden.py with its citation; peft is not installable here) and loaded from a
    peft-KEYED state dict.  Pins (1) the key layout (`base_model.model.<path>.lora_{A,B}.default.weight`) the packer must accept,
    (2) the oracle's un-merged forward (_wlin) and (3) the merge W' = W + (alpha / r) B A that libgvl's loader performs."""
    meta, g = load_golden("lora_" + tag)
    c = meta["cfg"]
    W = synth.llm_weights("phi3", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], True, seed=meta["seed"])
    Wp = synth.lora_wrap(W, "phi3", r=meta["r"], seed=meta["ab_seed"], std=meta["ab_std"])
    assert sorted(k for k in Wp if "lora_" in k) == meta["lora_keys"] and len(Wp) == meta["n_keys"]
    assert all(k.startswith("base_model.model.") for k in Wp)
    x = synth.det_tensor(meta["x"], meta["x_shape"], 0.5)[0]
    Wo = {k[len("base_model.model."):]: v for k, v in Wp.items()}
    old = O.LORA_SCALE
    O.LORA_SCALE = meta["lora_alpha"] / meta["r"]
    try:
        _close(O.llm_forward(_cfg(c), Wo, x, last_only=True)[0], g["logits"], 5e-5)                    # un-merged, as peft runs it
    finally:
        O.LORA_SCALE = old
    from grounded_video_llm_amd import weights as Wt
    merged = Wt._strip_peft(Wp, meta["lora_alpha"], meta["r"])
    assert not any("lora_" in k or k.startswith("base_model.") for k in merged)
    _close(O.llm_forward(_cfg(c), merged, x, last_only=True)[0], g["logits"], 5e-5)                   # merged in fp32: the same function
    base = O.llm_forward(_cfg(c), W, x, last_only=True)[0]
    assert float((base - torch.as_tensor(g["logits"])).abs().max()) > 0.05 * float(np.abs(g["logits"]).max()), "the adapters of this fixture do not matter"


def test_llama_tiny():
    meta, g = load_golden("llama_tiny")
    c = meta["cfg"]
    W = synth.llm_weights("llama", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], True, seed=meta["seed"])
    x = synth.det_tensor(meta["x"], meta["x_shape"], 0.5)[0]
    _close(O.llm_forward(_cfg(c), W, x), g["logits"][0], 2e-5)


def _glue(name):
    meta, g = load_golden(name)
    llm, hid = meta["llm"], meta["hidden"]
    cc, vc = meta["clip"], meta["iv2"]
    Wc = synth.clip_weights(cc["hidden"], cc["inter"], cc["layers"], 336, 14, seed="g.glue.clip")
    Wv = synth.iv2_weights(vc["dim"], vc["inter"], vc["depth"], vc["frames"], 224, 14, seed="g.glue.iv2")
    Wp = synth.projector_weights(llm, hid, 1024, 1408, seed="g.glue.proj." + llm)
    sp = synth.det_tensor("g.glue.sp", (1, 2, 3, 336, 336))
    tp = synth.det_tensor("g.glue.tp", (1, 4, 3, 224, 224))
    feats = O.encode_images(sp, tp, Wc, Wv, Wp, llm, clip_layers=cc["layers"], clip_heads=cc["heads"], iv2_depth=vc["depth"], iv2_heads=vc["heads"])
    assert list(feats.shape) == meta["feats_shape"]
    s = meta["stride"]
    _close(feats[:, ::s[0], ::s[1]], g["feats"], 1e-4)
    emb_w = synth.det_tensor("g.glue.embed." + llm, (50, hid), 0.5)
    emb = O.splice(torch.tensor(meta["ids"]), feats[0], emb_w)[None]
    assert list(emb.shape) == meta["emb_shape"]
    _close(emb[:, ::s[0], ::s[1]], g["emb"], 1e-4)
    assert g["mask"].shape[1] == emb.shape[1] and g["mask"].min() == 1


def test_glue_phi35():
    _glue("glue_phi3_5")        # 2 segments x 285 tokens (SURVEY §3.2 [probe])


def test_glue_llama3():
    _glue("glue_llama3")        # 2 segments x 193 tokens


def test_preprocess_oracle_matches_pillow():
    """the numpy restatement of Pillow's 8-bit two-pass bicubic resampler (+ torchvision's size / crop rules) must be
    BIT-EXACT against the images Pillow itself produced (tests/golden/preprocess.npz, oracle/make_golden.py pre)."""
    meta, g = load_golden("preprocess")
    for name, c in meta["cases"].items():
        img = O.synthetic_frame(name, c["h"], c["w"])
        nh, nw = O.tv_resized_size(c["h"], c["w"], c["size"])
        out = O.pil_resize_bicubic(img, nw, nh) if (nh, nw) != (c["h"], c["w"]) else img
        top, left = O.tv_center_crop_offsets(nh, nw, c["size"])
        out = out[top:top + c["size"], left:left + c["size"]]
        assert out.shape == g[name].shape and np.array_equal(out, g[name]), f"{name}: {int((out != g[name]).sum())} bytes differ from Pillow"
        x = O.frame_transform(np.transpose(img, (2, 0, 1)), c["size"], O.INTERNVIDEO_MEAN, O.INTERNVIDEO_STD)
        ref = (np.transpose(g[name], (2, 0, 1)).astype(np.float32) / np.float32(255) - np.asarray(O.INTERNVIDEO_MEAN, np.float32).reshape(3, 1, 1)) / \
            np.asarray(O.INTERNVIDEO_STD, np.float32).reshape(3, 1, 1)
        assert x.dtype == np.float32 and np.array_equal(x, ref)
    assert O.tv_resized_size(360, 640, 224) == (224, 398) and O.tv_resized_size(400, 226, 224) == (396, 224)
    assert O.tv_center_crop_offsets(224, 398, 224) == (0, 87) and O.tv_center_crop_offsets(229, 224, 224) == (2, 0)   # round-half-even: 2.5 -> 2


def test_training_labels_oracle_and_host_match_reference():
    """f4 integer path: make_labels / prepare_batch / prepare_multimodal_inputs labels+mask of the reference (generated by
    oracle/make_golden.py g_train with a toy tokenizer) vs the oracle restatement AND the product's host mirror -- bit-exact."""
    from grounded_video_llm_amd import prompts as P
    with open(os.path.join(GOLDEN, "train_labels.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 12
    tok = lambda s: [1] + [3 + (sum(map(ord, w)) % 90) for w in s.split()]
    n_answer = 0
    for key, c in cases.items():
        llm, tname, mtl = key.split("|")
        pad = 0 if tname == "pad0" else 2
        for impl, as_list in ((O, lambda t: t.tolist()), (P, lambda a: np.asarray(a).tolist())):
            ids, labels, mask = impl.prepare_batch(llm, c["texts"], tok, 1, pad, 2, int(mtl))
            assert as_list(ids) == c["input_ids"] and as_list(labels) == c["labels"] and as_list(mask) == c["attention_mask"], f"{key} {impl.__name__}"
            for r, vid in enumerate(c["video_ids"]):
                ml, mm = impl.splice_labels(ids[r], labels[r], mask[r], c["n_visual"], vid == "text")
                assert as_list(ml) == c["mm_labels"][r] and as_list(mm) == c["mm_mask"][r], f"{key} row {r} {impl.__name__}"
        n_answer += sum(v != -100 for row in c["mm_labels"] for v in row)
    assert n_answer > 100                                      # the fixtures do contain supervised tokens


def test_training_loss_oracle_matches_reference():
    """f4 FP path: oracle forward (fp32) + causal_lm_loss_terms vs Phi3ForCausalLM(labels=...).loss of the reference, incl. the
    right-padded batch identity loss(batch) == sum(nll) / sum(count) over its members."""
    meta, g = load_golden("train_loss")
    c = meta["cfg"]
    short, long = synth.longrope_factors(c["hidden"] // c["heads"])
    cfg = O.LLMConfig("phi3", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], 1e-5, 10000.0, 131072, 4096, short, long)
    W = synth.llm_weights("phi3", c["hidden"], c["inter"], c["layers"], c["heads"], c["kv_heads"], c["vocab"], True, seed=meta["seed"])
    tot, cnt = 0.0, 0
    for name, m in meta["cases"].items():
        x = synth.det_tensor(m["x"], (1, m["S"], c["hidden"]), 0.5)[0]
        logits = O.llm_forward(cfg, W, x, False, None, 0, last_only=False)
        s, n = O.causal_lm_loss_terms(logits, torch.tensor(m["labels"]))
        assert n == m["n_valid"]
        assert abs(s / n - float(g[name + "_loss"])) < 2e-5 * abs(float(g[name + "_loss"])), name
        assert abs(s - float(g[name + "_nll_sum"])) < 2e-5 * abs(s)
        if name in ("phi_a", "phi_b"):
            tot, cnt = tot + s, cnt + n
    assert abs(tot / cnt - float(g["phi_batch_ab_loss"])) < 2e-5 * float(g["phi_batch_ab_loss"])
    # Llama (GQA, plain RoPE) through LlamaForCausalLM's labels branch
    lc = meta["llama_cfg"]
    lcfg = O.LLMConfig("llama", lc["hidden"], lc["inter"], lc["layers"], lc["heads"], lc["kv_heads"], lc["vocab"], 1e-5, lc["rope_theta"], 8192, 0, None, None)
    LW = synth.llm_weights("llama", lc["hidden"], lc["inter"], lc["layers"], lc["heads"], lc["kv_heads"], lc["vocab"], True, seed=meta["llama_seed"])
    for name, m in meta["llama_cases"].items():
        x = synth.det_tensor(m["x"], (1, m["S"], lc["hidden"]), 0.5)[0]
        s, n = O.causal_lm_loss_terms(O.llm_forward(lcfg, LW, x, False, None, 0, last_only=False), torch.tensor(m["labels"]))
        assert n == m["n_valid"] and abs(s / n - float(g[name + "_loss"])) < 2e-5 * float(g[name + "_loss"]), name


def test_pillow_resampler_restatement_fuzz():
    """Where Pillow is importable (this image: 12.2.0; the pinned reference version is 11.1.0, same Resample.c arithmetic) the
    restatement is checked LIVE on random geometries -- strong down-scaling (wide kernels), up-scaling, 1-pixel axes, odd aspect
    ratios -- beyond the 7 committed golden cases.  Bit-exact."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    rng = np.random.default_rng(20260927)
    geoms = [(1, 1, 7, 5), (2, 3, 1, 1), (17, 640, 224, 224), (1080, 1920, 126, 224), (37, 41, 224, 398), (224, 224, 224, 224), (5, 300, 300, 5)]
    geoms += [tuple(int(v) for v in (rng.integers(1, 400), rng.integers(1, 400), rng.integers(1, 300), rng.integers(1, 300))) for _ in range(25)]
    for h, w, oh, ow in geoms:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if h * w > 64:                                         # hard edges / saturation: exercises the clip8 of both passes
            img[: h // 2, : w // 2] = 255
            img[h // 2:, w // 2:] = 0
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        got = O.pil_resize_bicubic(img, ow, oh)
        assert got.shape == ref.shape and np.array_equal(got, ref), f"{(h, w)} -> {(oh, ow)}: {int((got != ref).sum())} bytes differ from Pillow {PIL.__version__}"


def test_sampler_restatement_matches_installed_transformers_warpers():
    """do_sample=True (the reference's CLI default, inference.py:45-49 -> llava_next_video.py:655-661 -> HF generate [ext]): the oracle's
    kept set must equal TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper of the transformers that is installed here
    (5.x; the reference pins 4.40.1, whose warpers have the same definitions), on 300 random rows."""
    lp = pytest.importorskip("transformers.generation.logits_process")
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.choice([16, 100, 1000, 5000]))
        l = (rng.standard_normal(n) * rng.choice([0.5, 2, 5])).astype(np.float32)
        T = float(rng.choice([0.2, 0.7, 1.0, 1.5])); k = int(rng.choice([0, 1, 5, 50]))
        p = [None, 0.1, 0.5, 0.9, 0.99][int(rng.integers(0, 5))]
        s = torch.tensor(l)[None].double()
        s = lp.TemperatureLogitsWarper(T)(None, s)
        if k:
            s = lp.TopKLogitsWarper(k)(None, s)
        if p is not None:
            s = lp.TopPLogitsWarper(p)(None, s)
        assert np.array_equal(torch.isfinite(s[0]).numpy(), O.sample_keep_mask(l, T, k, p)), (trial, n, T, k, p)


def test_sampler_counter_hash_known_answers_and_distribution():
    """The counter hash shared with the HIP sampler is pinned by value (a change would silently alter every sampled run), and the
    Gumbel-max draw over it samples softmax(l / T)."""
    u = O.sample_uniforms(5, 0x123456789ABCDEF0, 7, 3)
    assert [float(x).hex() for x in u] == ['0x1.34bd490000000p-1', '0x1.082d680000000p-4', '0x1.a8ff8d0000000p-1', '0x1.3829120000000p-2', '0x1.63d8370000000p-1']
    l = np.array([0.0, 1.0, 2.0, -1.0, 0.5], dtype=np.float32)
    tok, margin, keep = O.sample_token(l, 0.7, 0, None, 1234, 3, 17)
    assert tok == 2 and abs(margin - 3.989575346104739) < 1e-9 and keep.all()
    cnt = np.zeros(5)
    for st in range(6000):
        cnt[O.sample_token(l, 0.7, 0, None, 1234, 3, st)[0]] += 1
    p = np.exp(l / 0.7); p /= p.sum()
    assert np.all(np.abs(cnt / 6000 - p) <= 4.5 * np.sqrt(p * (1 - p) / 6000))


def test_fused_rmsnorm_linear_is_in_the_noise_class_of_the_reference_order():
    """Round 5: the HIP build fuses RMSNorm into the GEMMs around it (row statistics from the producing epilogue, norm weight folded into the consuming weight,
    row scale on the accumulator).  oracle.rmsnorm_linear_fused restates that arithmetic with its rounding points; against the fp32 truth it must be no worse than
    the reference-order bf16 emulation (`_lin(_rmsnorm(...))`: two activation roundings before the GEMM), and the two bf16 forms must sit within bf16 output
    rounding of each other -- at InternVideo2's and Phi-3.5's widths, with rows of very different scale."""
    g = torch.Generator(); g.manual_seed(11)
    for C, N in ((1408, 4224), (3072, 1024)):
        x = torch.randn((96, C), generator=g) * (0.2 + 3.0 * torch.rand((96, 1), generator=g))
        x = x.to(torch.bfloat16).float()
        gamma = 1.0 + 0.2 * torch.randn((C,), generator=g)
        w = torch.randn((N, C), generator=g) * C ** -0.5
        b = 0.5 * torch.randn((N,), generator=g)
        truth = F.linear(O._rmsnorm(x, gamma.to(torch.bfloat16).float(), 1e-6, False), w.to(torch.bfloat16).float(), b)
        ref_order = O._lin(O._rmsnorm(x, gamma, 1e-6, True), w, b, True)
        fused = O.rmsnorm_linear_fused(x, gamma, w, b, 1e-6)
        scale = float(truth.abs().max())
        e_ref, e_fused = float((ref_order - truth).abs().max()) / scale, float((fused - truth).abs().max()) / scale
        r_ref, r_fused = float((ref_order - truth).pow(2).mean().sqrt()) / scale, float((fused - truth).pow(2).mean().sqrt()) / scale
        d = float((fused - ref_order).abs().max()) / scale
        print(f"[parity] fused RMSNorm + linear C={C} N={N}: vs fp32 max {e_fused:.3e} rms {r_fused:.3e}; reference order max {e_ref:.3e} rms {r_ref:.3e}; fused vs reference order {d:.3e}")
        assert e_fused <= 1.25 * e_ref + 1e-4 and r_fused <= 1.10 * r_ref
        assert d <= 8e-3
