"""-m gpu: BASELINE-size (configs[1]: Phi-3.5-3.8B, 96 frames) properties that do not need a CPU reference at that size
(the oracle would take minutes per layer): batch independence of the towers, equality of the batched and the one-at-a-time LLM
paths, determinism.  All comparisons are BIT-EXACT: every kernel computes an output row from its own input row(s) in a fixed
order, whatever tile, launch split (wave-quantisation planner) or decode group it lands in."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV  # noqa: E402
import bench  # noqa: E402  (repo root is on sys.path via conftest)


@pytest.fixture(scope="module")
def full():
    eng, geo = bench.build_engine(torch.device(DEV), clips_per_step=2)
    yield eng, geo
    eng.close()


def test_towers_are_batch_independent_at_full_size(full):
    eng, geo = full
    sp, tp, ids = bench.make_inputs(torch.device(DEV), 0)
    all12 = eng.encode_segments(sp, tp)
    L = eng.tokens_per_seg
    assert all12.shape == (12 * L, geo.hidden) and torch.isfinite(all12.float()).all()
    part = torch.cat([eng.encode_segments(sp[:5], tp[:5]), eng.encode_segments(sp[5:], tp[5:])], 0)
    assert torch.equal(all12, part), "12 segments at once != 5 + 7 segments (GEMM launch planner / tiling must not change any row)"
    again = eng.encode_segments(sp, tp)
    assert torch.equal(all12, again), "encode_segments is not deterministic"
    perm = torch.tensor([3, 0, 7, 11, 1, 2, 4, 5, 6, 8, 9, 10], device=sp.device)
    shuffled = eng.encode_segments(sp[perm], tp[perm]).view(12, L, -1)
    assert torch.equal(shuffled, all12.view(12, L, -1)[perm]), "segments are not independent"


def test_llm_batched_paths_equal_single_at_full_size(full):
    eng, geo = full
    dev = torch.device(DEV)
    sp, tp, ids = bench.make_inputs(dev, 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    vis = [eng.encode_segments(sp, tp), eng.encode_segments(torch.randn(sp.shape, device=dev, generator=g), torch.randn(tp.shape, device=dev, generator=g))]
    embs = [eng.splice(ids, v) for v in vis]
    S = embs[0].shape[0]
    assert S == 3519
    new = 6
    single, logits = [], []
    for e in embs:
        s = eng.seq_alloc(S + new)
        logits.append(eng.prefill(s, e, want_logits=True).clone())
        single.append(eng.decode_greedy(s, new, None))
        eng.seq_free(s)
    assert torch.isfinite(logits[0]).all() and not torch.equal(logits[0], logits[1])
    seqs = [eng.seq_alloc(S + new) for _ in embs]
    eng.prefill_batch(seqs, embs)                       # M = 2 x 3519 rows through the decoder GEMMs
    got = eng.decode_greedy_batch(seqs, new, None)      # one weight stream per token for both
    for s in seqs:
        eng.seq_free(s)
    assert got == single, f"batched prefill + decode {got} != single {single}"


def test_decode_step_matches_prefill_at_full_context(full):
    """The paged-KV decode path (GEMV + split-KV decode attention + fused RoPE/append) against the prefill path (MFMA GEMM + flash
    attention) at the BASELINE context (3519 tokens): logits of `prefill(S) ; decode(tok)` vs `prefill(S + 1 rows)` -- two different
    kernel families computing the same function, bf16 class tolerance 1e-2 of the logit scale (north_star), same argmax."""
    eng, geo = full
    dev = torch.device(DEV)
    sp, tp, ids = bench.make_inputs(dev, 0)
    vis = eng.encode_segments(sp, tp)
    tok = 1234
    emb = eng.splice(ids, vis)
    emb1 = eng.splice(list(ids) + [tok], vis)
    assert emb1.shape[0] == emb.shape[0] + 1 and torch.equal(emb1[:-1], emb)
    s1 = eng.seq_alloc(emb1.shape[0] + 2)
    ref = eng.prefill(s1, emb1, want_logits=True).clone()
    eng.seq_free(s1)
    s0 = eng.seq_alloc(emb.shape[0] + 2)
    eng.prefill(s0, emb)
    got = eng.decode_step_logits(s0, tok)
    eng.seq_free(s0)
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max()) / scale
    # noise floor of THIS (random-init, 32-layer) network: the same prefill with every embedding element moved by at most one bf16 ulp
    g = torch.Generator(device=dev); g.manual_seed(9)
    flip = (torch.randint(0, 2, emb1.shape, device=dev, generator=g) * 2 - 1).to(torch.float32)
    emb_p = (emb1.float() * (1.0 + flip * 2.0 ** -7)).to(torch.bfloat16)     # ~1 ulp
    s2 = eng.seq_alloc(emb1.shape[0] + 2)
    pert = eng.prefill(s2, emb_p, want_logits=True).clone()
    eng.seq_free(s2)
    floor = float((pert - ref).abs().max()) / scale
    print(f"[parity] full-size decode-vs-prefill logits rel err {err:.2e}; 1-ulp input perturbation moves the logits by {floor:.2e} (scale {scale:.2f})")
    assert err < max(1e-2, 2.0 * floor), "decode path disagrees with the prefill path by more than the network's own bf16 noise"
    top2 = torch.topk(ref, 2).values
    if float(top2[0] - top2[1]) > 2e-2 * scale:
        assert int(got.argmax()) == int(ref.argmax())


def test_llama3_8b_batched_paths_equal_single_at_full_size():
    """BASELINE configs[3] (Llama-3-8B: GQA 32/8 heads of 128, separate q/k/v packed into one projection, plain RoPE) at its real size:
    batched prefill + batched decode must reproduce the one-at-a-time ids; LLM only (synthetic visual prefix of the C3 length)."""
    from grounded_video_llm_amd import engine as E, synth, weights as Wt
    geo = E.TowerGeometry.llama3_8b(max_seq=4096, max_prefill=2 * 2432, kv_pages=80, max_segs=1)
    eng = E.Engine(geo, DEV, towers=("llm",))
    W = synth.llm_weights("llama", geo.hidden, geo.inter, geo.layers, geo.heads, geo.kv_heads, geo.vocab, True, seed="full.llama", device=DEV)
    eng.load_packed(Wt.pack_llm(W, "llama", geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta, None, None)); del W
    torch.cuda.empty_cache()
    eng.finalize()
    S, new = 2416, 6                                        # 12 x 193 visual tokens + ~100 text tokens
    g = torch.Generator(device=DEV); g.manual_seed(5)
    embs = [(torch.randn((S, geo.hidden), device=DEV, generator=g) * 0.5).to(torch.bfloat16) for _ in range(2)]
    single = [eng.generate_ids(e, new, None) for e in embs]
    seqs = [eng.seq_alloc(S + new) for _ in embs]
    eng.prefill_batch(seqs, embs)
    got = eng.decode_greedy_batch(seqs, new, None)
    for s in seqs:
        eng.seq_free(s)
    eng.close()
    assert got == single and single[0] != single[1]
