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
