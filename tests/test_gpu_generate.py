"""-m gpu: the top-level surface LLAVA_NEXT_VIDEO(...).generate(samples, **kw) (the reference's models/llava_next_video.py:616-666
contract) end to end on a small model: prompt plumbing -> encode_images -> splice -> prefill -> paged greedy decode -> text,
against the CPU oracle run on the same seeded weights/pixels."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvl_oracle as O  # noqa: E402
from gpu_util import DEV, bf  # noqa: E402
from grounded_video_llm_amd import engine as E, prompts as P, synth  # noqa: E402
from grounded_video_llm_amd.model import LLAVA_NEXT_VIDEO, SyntheticTokenizer  # noqa: E402


@pytest.mark.parametrize("llm", ["phi3.5", "llama3"])
def test_generate_matches_oracle(llm):
    hid, vocab = 128, 640
    kind = "phi3" if llm == "phi3.5" else "llama"
    short, long = synth.longrope_factors(32)
    geo = E.TowerGeometry(llm=llm, clip_hidden=64, clip_inter=128, clip_layers=3, clip_heads=4, iv2_dim=64, iv2_inter=128, iv2_depth=3,
                          iv2_heads=4, hidden=hid, inter=256, layers=2, heads=4, kv_heads=4 if kind == "phi3" else 2, vocab=vocab,
                          rope_short=short if kind == "phi3" else None, rope_long=long if kind == "phi3" else None,
                          rope_theta=10000.0 if kind == "phi3" else 500000.0, max_seq=2048, max_segs=6, kv_pages=40, max_prefill=1024)
    sd = {"vision_tower": synth.clip_weights(64, 128, 3, seed="gen.clip"),
          "video_encoder": synth.iv2_weights(64, 128, 3, 2, seed="gen.iv2"),
          "projectors": synth.projector_weights(llm, hid, 64, 64, seed="gen.proj"),
          "language_model": synth.llm_weights(kind, hid, 256, 2, 4, geo.kv_heads, vocab, True, seed="gen.llm")}
    tok = SyntheticTokenizer(vocab, 300)
    model = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=64, num_frames=4, num_segs=2, num_temporal_tokens=300, lora=False, llm=llm,
                             geometry=geo, tokenizer=tok, state_dicts=sd, device=DEV)
    sp = synth.det_tensor("gen.sp", (1, 2, 3, 336, 336))
    tp = synth.det_tensor("gen.tp", (1, 4, 3, 224, 224))
    prompt = P.build_prompt(llm, "grounding", "When does the person open the door in the video?")
    samples = {"prompts": [prompt], "spatial_pixel_values": sp.to(DEV), "temporal_pixel_values": tp.to(DEV), "video_ids": ["synthetic"]}
    # --- oracle on the same inputs
    ids = O.tokenizer_image_token(prompt, tok, tok.bos_token_id)
    assert ids == model.tokenizer_image_token(prompt) and ids.count(-200) == 1
    ref_vis = O.encode_images(sp, tp, sd["vision_tower"], sd["video_encoder"], sd["projectors"], llm, clip_layers=3, clip_heads=4,
                              iv2_depth=3, iv2_heads=4, emu=True)[0]
    ocfg = O.LLMConfig(kind, hid, 256, 2, 4, geo.kv_heads, vocab, 1e-5, geo.rope_theta, 131072, 4096, geo.rope_short, geo.rope_long)
    ref_emb = O.splice(torch.tensor(ids), ref_vis, sd["language_model"]["model.embed_tokens.weight"], emu=True)
    ref_ids, margins, scales = O.greedy_generate(ocfg, sd["language_model"], ref_emb, 10, tok.eos_token_id, emu=True, return_margins=True, return_scales=True)
    # --- product
    feats = model.encode_images(samples)
    err = float((feats[0].float().cpu() - ref_vis).abs().max() / ref_vis.abs().max())
    print(f"[parity] generate({llm}): visual tokens rel err {err:.2e}")
    assert list(feats.shape) == [1, 2 * model.engine.tokens_per_seg, hid] and err < 2e-2
    ids_arr, mask = P.left_pad_truncate([ids], tok.pad_token_id, model.max_txt_len)
    got = model.generate_ids(ids_arr, mask, feats, 10)[0]
    for i, (a, b) in enumerate(zip(got, ref_ids)):
        if a != b:
            # ids may part ways only where the oracle's top-1 margin is inside twice the bf16-class logit tolerance (2e-2 of the logit scale,
            # the bound every logit test of these small models uses) -- relative to the scale, as tests/test_gpu_c0.py does at full size
            print(f"[parity] generate({llm}): ids part ways at token {i} ({a} vs {b}); oracle margin {margins[i] / scales[i]:.3e} of the logit scale")
            assert margins[i] < min(2 * 2e-2 * scales[i], 0.25), f"token {i}: {a} vs {b}, oracle margin {margins[i]} = {margins[i] / scales[i]:.3e} of the scale {scales[i]}"
            break
    else:
        assert len(got) == len(ref_ids)
    texts = model.generate(samples, do_sample=False, num_beams=1, max_new_tokens=10)
    assert isinstance(texts, list) and len(texts) == 1 and texts[0] == tok.batch_decode([got], skip_special_tokens=True)[0].strip()
    # the reference's default CLI configuration is SAMPLING (inference.py:45-49: do_sample True, temperature 0.2): reproducible under a
    # seed, top_k = 1 collapses to greedy, and a later do_sample=False call is greedy again
    smp = dict(do_sample=True, num_beams=1, max_new_tokens=10, temperature=1.5, top_p=None)
    s1 = model.generate(samples, seed=11, **smp)
    assert s1 == model.generate(samples, seed=11, **smp)
    draws = {model.generate(samples, seed=k, **smp)[0] for k in range(6)}
    assert len(draws) > 1, "six seeds at temperature 1.5 all gave the same text"
    assert model.generate(samples, seed=3, top_k=1, **smp) == texts
    assert model.generate(samples, do_sample=True, temperature=0.2, max_new_tokens=10)[0] is not None      # unseeded: seed from torch's generator
    with pytest.raises(ValueError):
        model.generate(samples, do_sample=True, temperature=0.0)
    bs1 = model.generate(samples, do_sample=True, num_beams=2, seed=4, max_new_tokens=8)          # beam-sample (HF _beam_sample; beam.py): reproducible under a seed,
    assert isinstance(bs1[0], str) and bs1 == model.generate(samples, do_sample=True, num_beams=2, seed=4, max_new_tokens=8)
    assert model.engine.kv_info()["free_pages"] == model.engine.kv_info()["total_pages"]             # ... and it gives its KV pages back
    # beam search (HF generate(num_beams = k, do_sample = False); grounded_video_llm_amd/beam.py restates the scorer, CPU-tested against the
    # installed transformers): the forked KV cache (gvl_seq_clone: whole pages shared, partial page copied) must give exactly what an
    # UNSHARED recomputation gives -- every beam rebuilt from the prompt by a fresh prefill + teacher-forced decode steps
    from grounded_video_llm_amd import beam as B
    eng = model.engine
    ids_arr, mask = __import__("grounded_video_llm_amd.prompts", fromlist=["x"]).left_pad_truncate([model.tokenizer_image_token(samples["prompts"][0])], 0, model.max_txt_len)
    row = [int(t) for t, m in zip(ids_arr[0], mask[0]) if m]
    vis = model.encode_images(samples)[0]
    emb = eng.splice(row, vis)
    for k, max_new in ((3, 7), (2, 12)):
        got = model.beam_generate_ids(row, vis, k, max_new)
        hist = [[] for _ in range(k)]

        def slow_step(parents, toks):
            hist[:] = [hist[p_] + [t] for p_, t in zip(parents, toks)]
            out = []
            for h in hist:
                s_ = eng.seq_alloc(emb.shape[0] + len(h) + 2)
                eng.prefill(s_, emb)
                for t in h:
                    lg = eng.decode_step_logits(s_, t)
                out.append(lg.clone())
                eng.seq_free(s_)
            return torch.stack(out)
        s0 = eng.seq_alloc(emb.shape[0] + 2)
        first = eng.prefill(s0, emb, want_logits=True).clone()
        eng.seq_free(s0)
        want = B.beam_search(slow_step, first, k, max_new, getattr(model.tokenizer, "eos_token_id", None))
        assert got == want and 1 <= len(got) <= max_new, f"beam search k={k}: forked KV {got} vs recomputed {want}"
    assert eng.kv_info()["free_pages"] == eng.kv_info()["total_pages"], "beam search leaked KV pages"
    assert isinstance(model.generate(samples, do_sample=False, num_beams=3, max_new_tokens=6)[0], str)
    assert model.generate(samples, do_sample=False, num_beams=1, max_new_tokens=10) == texts
    print(f"[parity] generate({llm}) ids {got} oracle {ref_ids}; text {texts[0]!r}")
    # batch of 3 samples with DIFFERENT prompts (the reference left-pads, llava_next_video.py:622-647): the batched decode must give,
    # sample by sample, exactly what the one-sample generate() gives
    prompts = [prompt, P.build_prompt(llm, "grounding", "When does the dog jump?"), P.build_prompt(llm, "qa", "Describe the video in detail please.")]
    sp3 = torch.cat([sp, synth.det_tensor("gen.sp2", (2, 2, 3, 336, 336))], 0).to(DEV)
    tp3 = torch.cat([tp, synth.det_tensor("gen.tp2", (2, 4, 3, 224, 224))], 0).to(DEV)
    batch = {"prompts": prompts, "spatial_pixel_values": sp3, "temporal_pixel_values": tp3, "video_ids": ["a", "b", "c"]}
    texts3 = model.generate(batch, do_sample=False, num_beams=1, max_new_tokens=10)
    singles = [model.generate({"prompts": [prompts[i]], "spatial_pixel_values": sp3[i:i + 1], "temporal_pixel_values": tp3[i:i + 1], "video_ids": ["x"]},
                              do_sample=False, num_beams=1, max_new_tokens=10)[0] for i in range(3)]
    assert texts3 == singles and texts3[0] == texts[0]
    # several prompts about ONE video: encoded once, prompts batched -- same texts as one generate() per prompt
    one = {"spatial_pixel_values": sp3[1:2], "temporal_pixel_values": tp3[1:2], "video_ids": ["x"]}
    shared = model.generate_shared(one, prompts, do_sample=False, num_beams=1, max_new_tokens=10)
    assert shared == [model.generate({**one, "prompts": [p]}, do_sample=False, num_beams=1, max_new_tokens=10)[0] for p in prompts]
    assert model.last_shared_prefix >= 128 and model.last_shared_prefix % 128 == 0      # the system prompt + visual tokens were prefilled ONCE (gvl_seq_fork / gvl_prefill_extend)
    model.engine.close()
    # start from ONE packed weight file (tools/pack_checkpoint.py's output format) instead of state dicts: same answers
    import os, tempfile
    from grounded_video_llm_amd import weights as Wt
    g2 = model.geo
    packed = {}
    packed.update(Wt.pack_clip(sd["vision_tower"], g2.clip_layers - 1))
    packed.update(Wt.pack_iv2(sd["video_encoder"], g2.iv2_depth - 1, g2.frames_per_seg))
    packed.update(Wt.pack_projectors(sd["projectors"], llm))
    packed.update(Wt.pack_llm(sd["language_model"], g2.kind, g2.layers, g2.heads, g2.kv_heads, g2.max_seq, g2.rope_theta, g2.rope_short, g2.rope_long,
                              g2.rope_max_pos, g2.rope_orig_max_pos))
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "w.gvl.safetensors")
        Wt.save_packed(path, packed)
        m2 = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=64, num_frames=4, num_segs=2, num_temporal_tokens=300, lora=False, llm=llm,
                              geometry=g2, tokenizer=tok, packed_weights=path, device=DEV)
        assert m2.generate(samples, do_sample=False, num_beams=1, max_new_tokens=10) == texts       # weights came through gvl_load_packed (C++)
        m2.engine.close()
        # the C++ reader agrees tensor by tensor with the Python one, and refuses a file that is not a packed weight file
        e3 = E.Engine(g2, DEV)
        assert e3.load_packed_file(path) == len(packed)
        from safetensors.torch import save_file
        other = os.path.join(td, "other.safetensors")
        save_file({"x": torch.zeros(4)}, other)
        with pytest.raises(E.L.GvlError, match="not a gvl packed"):
            e3.load_packed_file(other)
        with open(os.path.join(td, "trunc.safetensors"), "wb") as f:
            f.write(open(path, "rb").read()[:1000])
        with pytest.raises(E.L.GvlError):
            e3.load_packed_file(os.path.join(td, "trunc.safetensors"))
        e3.close()


@pytest.mark.parametrize("llm", ["phi3.5", "llama3"])
def test_training_forward_loss_matches_oracle(llm):
    """f4 end to end: LLAVA_NEXT_VIDEO.forward(samples)["loss"] (llava_next_video.py:598-614) on a batch of two video samples and
    one text-only sample with conversations of different lengths, against the oracle pipeline on the same seeded weights:
    prepare_batch -> encode_images -> splice (+labels/mask) -> decoder -> shifted cross entropy over the right-padded batch."""
    hid, vocab = 128, 640
    kind = "phi3" if llm == "phi3.5" else "llama"
    short, long = synth.longrope_factors(32)
    geo = E.TowerGeometry(llm=llm, clip_hidden=64, clip_inter=128, clip_layers=3, clip_heads=4, iv2_dim=64, iv2_inter=128, iv2_depth=3,
                          iv2_heads=4, hidden=hid, inter=256, layers=2, heads=4, kv_heads=4 if kind == "phi3" else 2, vocab=vocab,
                          rope_short=short if kind == "phi3" else None, rope_long=long if kind == "phi3" else None,
                          rope_theta=10000.0 if kind == "phi3" else 500000.0, max_seq=2048, max_segs=6, kv_pages=40, max_prefill=1024)
    sd = {"vision_tower": synth.clip_weights(64, 128, 3, seed="gen.clip"),
          "video_encoder": synth.iv2_weights(64, 128, 3, 2, seed="gen.iv2"),
          "projectors": synth.projector_weights(llm, hid, 64, 64, seed="gen.proj"),
          "language_model": synth.llm_weights(kind, hid, 256, 2, 4, geo.kv_heads, vocab, True, seed="gen.llm")}
    tok = SyntheticTokenizer(vocab, 300)
    model = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=256, num_frames=4, num_segs=2, num_temporal_tokens=300, lora=False, llm=llm,
                             geometry=geo, tokenizer=tok, state_dicts=sd, device=DEV)
    T = P.TEMPLATES[llm]
    convs = [[{"from": "human", "value": "<image> <timestamp_grounding>\nWhen does the person open the door ?"}, {"from": "gpt", "value": "From <36> to <64> ."}],
             [{"from": "human", "value": "<image>\nWhat happens ?"}, {"from": "gpt", "value": "A dog jumps over the fence ."},
              {"from": "human", "value": "And then ?"}, {"from": "gpt", "value": "It runs away quickly ."}],
             [{"from": "human", "value": "<image>\nTell me about doors ."}, {"from": "gpt", "value": "Doors open and close ."}]]
    texts = [T.encode(c) for c in convs]
    video_ids = ["v0", "v1", "text"]
    sp = synth.det_tensor("trn.sp", (3, 2, 3, 336, 336))
    tp = synth.det_tensor("trn.tp", (3, 4, 3, 224, 224))
    samples = {"text_inputs": texts, "video_ids": video_ids, "spatial_pixel_values": sp.to(DEV), "temporal_pixel_values": tp.to(DEV)}
    got = float(model.forward(samples)["loss"])
    # --- oracle
    ids, labels, mask = O.prepare_batch(llm, texts, tok, tok.bos_token_id, tok.pad_token_id, tok.eos_token_id, model.max_txt_len)
    vis = O.encode_images(sp, tp, sd["vision_tower"], sd["video_encoder"], sd["projectors"], llm, clip_layers=3, clip_heads=4, iv2_depth=3, iv2_heads=4, emu=True)
    ocfg = O.LLMConfig(kind, hid, 256, 2, 4, geo.kv_heads, vocab, 1e-5, geo.rope_theta, 131072, 4096, geo.rope_short, geo.rope_long)
    tot, cnt = 0.0, 0
    for b in range(3):
        is_text = video_ids[b] == "text"
        ml, mm = O.splice_labels(ids[b], labels[b], mask[b], vis.shape[1], is_text)
        n = int(mm.sum())
        row = ids[b][mask[b] == 1]
        emb = O.splice(row, vis[b][:0] if is_text else vis[b], sd["language_model"]["model.embed_tokens.weight"], emu=True)
        assert emb.shape[0] == n
        lg = O.llm_forward(ocfg, sd["language_model"], emb, True, None, 0, last_only=False).to(bf)
        s, c = O.causal_lm_loss_terms(lg, ml[:n])
        tot, cnt = tot + s, cnt + c
    assert cnt > 10
    print(f"[parity] forward({llm}) loss gpu {got:.5f} oracle {tot / cnt:.5f} over {cnt} labelled tokens")
    assert abs(got - tot / cnt) < 1e-2 * (tot / cnt)
    model.engine.close()


def test_generate_shared_respects_the_longrope_switch():
    """LongRoPE picks ONE factor set per forward from the total length (modeling_phi3.py:381-385).  When the prompts are longer than
    original_max_position_embeddings but their 128-aligned common prefix is not, a shared base prefill would cache the prefix K with the SHORT
    factors while a full prefill ropes every row with the LONG ones: generate_shared must fall back to full prefills there (same texts as one
    generate() per prompt, nothing shared); once the prefix itself is past the switch, sharing is on again and still identical."""
    llm, hid, vocab = "phi3.5", 128, 640
    short, long = synth.longrope_factors(32)
    tok = SyntheticTokenizer(vocab, 300)
    prompts = [P.build_prompt(llm, "grounding", "When does the person open the door in the video?"), P.build_prompt(llm, "grounding", "When does the dog jump?"),
               P.build_prompt(llm, "qa", "Describe the video in detail please.")]
    n_vis = 2 * (156 + 16 * 2 + 1)                  # 2 segments x (156 image tokens + 16 per frame x 2 frames + newline)
    totals = [len(P.tokenize_with_image(p, tok, tok.bos_token_id)) - 1 + n_vis for p in prompts]
    sd = {"vision_tower": synth.clip_weights(64, 128, 3, seed="gen.clip"), "video_encoder": synth.iv2_weights(64, 128, 3, 2, seed="gen.iv2"),
          "projectors": synth.projector_weights(llm, hid, 64, 64, seed="gen.proj"), "language_model": synth.llm_weights("phi3", hid, 256, 2, 4, 4, vocab, True, seed="gen.llm")}
    sp = synth.det_tensor("gen.sp", (1, 2, 3, 336, 336)).to(DEV)
    tp = synth.det_tensor("gen.tp", (1, 4, 3, 224, 224)).to(DEV)
    one = {"spatial_pixel_values": sp, "temporal_pixel_values": tp, "video_ids": ["x"]}
    texts = {}
    prefix = (17 + n_vis) // 128 * 128                # the common prefix ends inside the visual tokens' tail: 17 shared ids + 378 visual rows -> 384
    for omax, shared_expected in ((min(totals) - 8, False), (300, True)):       # prefix <= omax < every total: mixed factors -> fall back; omax < prefix: prefix and prompts both long
        assert (prefix <= omax < min(totals)) == (not shared_expected), (prefix, omax, totals)
        geo = E.TowerGeometry(llm=llm, clip_hidden=64, clip_inter=128, clip_layers=3, clip_heads=4, iv2_dim=64, iv2_inter=128, iv2_depth=3, iv2_heads=4, hidden=hid,
                              inter=256, layers=2, heads=4, kv_heads=4, vocab=vocab, rope_short=short, rope_long=long, rope_theta=10000.0, rope_orig_max_pos=omax,
                              max_seq=2048, max_segs=6, kv_pages=60, max_prefill=1024)
        model = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=64, num_frames=4, num_segs=2, num_temporal_tokens=300, lora=False, llm=llm, geometry=geo, tokenizer=tok,
                                 state_dicts=sd, device=DEV)
        per_prompt = [model.generate({**one, "prompts": [p]}, do_sample=False, num_beams=1, max_new_tokens=10)[0] for p in prompts]
        shared = model.generate_shared(one, prompts, do_sample=False, num_beams=1, max_new_tokens=10)
        assert shared == per_prompt, f"rope_orig_max_pos={omax}: generate_shared differs from one generate() per prompt"
        assert (model.last_shared_prefix >= 128) == shared_expected, (omax, model.last_shared_prefix)
        assert model.engine.kv_info()["free_pages"] == model.engine.kv_info()["total_pages"]
        texts[omax] = per_prompt
        model.engine.close()
