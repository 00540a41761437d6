"""-m gpu: every HIP kernel family through the C ABI against a plain fp32 torch reference of the same op
(floating-point kernels; tolerance = bf16 output rounding, stated per test)."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

import gvl_oracle as O  # noqa: E402
from gpu_util import DEV, bf, check, tiny_geo  # noqa: E402
from grounded_video_llm_amd import engine as E, lib as L, synth  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(tiny_geo(), DEV, towers=())
    yield e
    e.close()


def _rand(name, shape, std=1.0):
    return synth.det_tensor(name, shape, std).to(DEV)


@pytest.mark.parametrize("cfg", [1, 0, 21, 22, 82, 85])
@pytest.mark.parametrize("M,N,K", [(100, 64, 64), (300, 256, 128), (577, 1024, 640), (1000, 1408, 1408), (1, 128, 192), (513, 4224, 256), (259, 72, 64)])
def test_gemm_plain(eng, M, N, K, cfg):
    A = _rand(f"gA{M}{N}{K}", (M, K)).to(bf)
    W = _rand(f"gW{M}{N}{K}", (N, K), K ** -0.5).to(bf)
    ref = A.float() @ W.float().T
    got = eng.op_gemm(A, W, tile_cfg=cfg)
    check(got, ref, 6e-3, f"gemm {M}x{N}x{K} cfg{cfg} bf16 out")
    got32 = eng.op_gemm(A, W, out_f32=True, tile_cfg=cfg)
    check(got32, ref, 2e-4, f"gemm {M}x{N}x{K} cfg{cfg} f32 out")


def test_gemm_random_shapes_auto_planner(eng):
    """Seeded random (M, N, K) through the AUTOMATIC configuration (kernel choice + wave-quantisation planner: whole / tile-row
    split / tail columns) with a random fused epilogue, against fp32 torch; and bit-identical to the plain 128x128 kernel (cfg 21):
    every kernel accumulates an output element in the same k order."""
    import random
    rnd = random.Random(99)
    shapes = [(6924, 1024, 1024), (4103, 1408, 1024), (2049, 4224, 1408), (8200, 1028, 64), (1, 4096, 2048), (5000, 260, 1024)]
    shapes += [(rnd.randint(1, 9000), 4 * rnd.randint(1, 1200), 64 * rnd.randint(1, 32)) for _ in range(8)]
    rb = lambda t: t.to(bf).float()
    for i, (M, N, K) in enumerate(shapes):
        g = torch.Generator(device=DEV); g.manual_seed(1000 + i)
        A = torch.randn((M, K), device=DEV, generator=g).to(bf)
        W = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).to(bf)
        bias = torch.randn((N,), device=DEV, generator=g) * 0.5
        res = torch.randn((M, N), device=DEV, generator=g).to(bf)
        acc = A.float() @ W.float().T
        kind = i % 3
        if kind == 0:
            kw, ref, tol = {}, acc, 6e-3
        elif kind == 1:
            kw, ref, tol = dict(bias=bias, act=L.ACT_GELU), torch.nn.functional.gelu(rb(acc + bias)), 8e-3
        else:
            kw, ref, tol = dict(bias=bias, resid=res), res.float() + rb(acc + bias), 8e-3
        got = eng.op_gemm(A, W, tile_cfg=0, **kw)
        check(got, ref, tol, f"gemm auto {M}x{N}x{K} epilogue {kind}")
        assert torch.equal(got, eng.op_gemm(A, W, tile_cfg=21, **kw)), f"gemm auto {M}x{N}x{K}: planner result differs bitwise from cfg 21"


def test_gemm_transpose_detect(eng):
    # A = identity-like with ASYMMETRIC W catches swapped row/col in the C write (guide §3)
    M = N = K = 128
    A = torch.eye(M, K, device=DEV).to(bf)
    W = (torch.arange(N, device=DEV)[:, None] * 0.01 + torch.arange(K, device=DEV)[None, :] * 1.0).to(bf)
    got = eng.op_gemm(A, W, out_f32=True, tile_cfg=1)
    check(got, A.float() @ W.float().T, 1e-6, "gemm identity x asymmetric")


@pytest.mark.parametrize("cfg", [1, 21, 82, 85])
def test_gemm_epilogues(eng, cfg):
    M, N, K = 333, 512, 256
    A = _rand("eA", (M, K)).to(bf)
    W = _rand("eW", (N, K), K ** -0.5).to(bf)
    bias = _rand("eb", (N,), 0.5)
    acc = A.float() @ W.float().T + bias
    rb = lambda t: t.to(bf).float()
    # quick_gelu (CLIP fc1)
    x = rb(acc)
    ref = x * rb(torch.sigmoid(rb(1.702 * x)))
    check(eng.op_gemm(A, W, bias=bias, act=L.ACT_QUICK_GELU, tile_cfg=cfg), ref, 8e-3, "quick_gelu epilogue")
    # erf gelu (IV2 fc1, projectors)
    check(eng.op_gemm(A, W, bias=bias, act=L.ACT_GELU, tile_cfg=cfg), torch.nn.functional.gelu(rb(acc)), 8e-3, "gelu epilogue")
    # f32 residual (CLIP): x + bf16(acc+bias)
    res = _rand("er", (M, N), 2.0)
    check(eng.op_gemm(A, W, bias=bias, resid=res, out_f32=True, tile_cfg=cfg), res + rb(acc), 1e-5 + 2e-3, "f32 residual epilogue")
    # bf16 residual + LayerScale (IV2): bf16(x + bf16(bf16(acc+bias)*gamma))
    gam = _rand("eg", (N,), 0.05, ) + 0.1
    resb = res.to(bf)
    ref = resb.float() + rb(rb(acc) * gam)
    check(eng.op_gemm(A, W, bias=bias, gamma=gam, resid=resb, tile_cfg=cfg), ref, 8e-3, "bf16 residual + gamma epilogue")
    # SwiGLU on interleaved (gate, up) columns (Phi-3 gate_up_proj)
    accn = rb(A.float() @ W.float().T)
    g, u = accn[:, 0::2], accn[:, 1::2]
    ref = u * rb(torch.nn.functional.silu(g))
    check(eng.op_gemm(A, W, act=L.ACT_SILU_MUL, tile_cfg=cfg), ref, 8e-3, "silu-mul epilogue")


@pytest.mark.parametrize("M,N,K", [(1000, 1408, 1408), (5700, 1408, 1408), (333, 128, 64), (3000, 3072, 1024), (257, 4224, 128)])
def test_gemm_row_sums_of_squares(eng, M, N, K):
    """GemmArgs.rowsq (fused RMSNorm, producer side; internvideo2.py:443-448 needs mean(x^2) of the NEW residual stream): the GEMM that writes the stream leaves,
    per row and aligned 64-column block, the sum of squares of its bf16-ROUNDED outputs.  (1) equals torch on the stored outputs; (2) every slot is written
    exactly once (NaN-filled buffer), also through the planner's row / column splits; (3) BIT-identical whichever kernel of the launch plan computed the row
    (256x256 ping-pong incl. its dead-wave tail tiles, 128x128, 64x128) -- the consumer's row scale must not depend on the batch composition."""
    A = _rand(f"sqA{M}{N}{K}", (M, K)).to(bf)
    W = _rand(f"sqW{M}{N}{K}", (N, K), K ** -0.5).to(bf)
    bias = _rand(f"sqb{N}", (N,), 0.5)
    gam = _rand(f"sqg{N}", (N,), 0.05) + 0.1
    res = _rand(f"sqr{M}{N}", (M, N), 2.0).to(bf)
    outs = {}
    for cfg in (0, 21, 22, 82):
        for kw in (dict(bias=bias, gamma=gam, resid=res), dict(resid=res), dict()):
            c, sq = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=cfg, **kw)
            assert torch.equal(c, eng.op_gemm(A, W, tile_cfg=cfg, **kw)), "the statistics must not change the output"
            assert torch.isfinite(sq).all(), f"cfg {cfg}: a (row, block) slot was not written"
            ref = c.float().pow(2).view(M, N // 64, 64).sum(-1)
            check(sq, ref, 2e-6, f"rowsq {M}x{N}x{K} cfg{cfg} {sorted(kw)}")
            key = tuple(sorted(kw))
            if key in outs:
                assert torch.equal(sq, outs[key]), f"rowsq differs bitwise between tile configurations (cfg {cfg})"
            outs[key] = sq


@pytest.mark.parametrize("M,N,K,act", [(1000, 4224, 1408, "none"), (5700, 6144, 1408, "gelu"), (3000, 1024, 1024, "silu"), (130, 192, 64, "none"), (24588, 4224, 1408, "none")])
def test_gemm_row_scale_is_rmsnorm_folded(eng, M, N, K, act):
    """GemmArgs.rowscale + gvl_op_fold_gamma + gvl_op_rowsq_finish (fused RMSNorm, consumer side): rs[m] * (x . (W diag(gamma))^T) against (1) fp32 torch of
    RMSNorm(x) W^T and (2) the UNFUSED pair this build also ships (rmsnorm_bf16 pass, then the plain GEMM): the two differ only in where roundings sit -- the
    bound is the bf16 output rounding of either.  Bit-identical across tile configurations."""
    x = (_rand(f"rsx{M}{K}", (M, K)) * (1.0 + _rand(f"rsm{M}", (M, 1)).abs())).to(bf)           # rows of different scale: rs really varies
    W = _rand(f"rsW{N}{K}", (N, K), K ** -0.5).to(bf)
    g = (_rand(f"rsg{K}", (K,), 0.2) + 1.0).to(bf)
    bias = _rand(f"rsb{N}", (N,), 0.5) if act == "gelu" else None        # the fused-epilogue set the model uses: qkv (no bias), fc1 (bias + GELU), gate_up (SwiGLU)
    eps = 1e-6
    a = dict(none=L.ACT_NONE, gelu=L.ACT_GELU, silu=L.ACT_SILU_MUL)[act]
    rb = lambda t: t.to(bf).float()
    # the row statistics as the model gets them: from the epilogue of a GEMM that wrote x (identity-free here: any producer gives sum x^2 over 64-blocks)
    sq = x.float().pow(2).view(M, K // 64, 64).sum(-1).contiguous()
    rs = eng.op_rowsq_finish(sq, K, eps)
    check(rs, torch.rsqrt(x.float().pow(2).mean(-1) + eps), 2e-6, "rowsq_finish")
    Wf = eng.op_fold_gamma(W, g)
    assert torch.equal(Wf, (W.float() * g.float()).to(bf))
    got = eng.op_gemm_rows(x, Wf, bias=bias, act=a, rowscale=rs)
    h32 = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps) * g.float()
    acc = h32 @ W.float().T + (bias if bias is not None else 0.0)
    if act == "gelu":
        ref = torch.nn.functional.gelu(rb(acc))
    elif act == "silu":
        accn = rb(acc)
        ref = accn[:, 1::2] * rb(torch.nn.functional.silu(accn[:, 0::2]))
    else:
        ref = acc
    tol = 1.5e-2 if act == "silu" else 1e-2           # SwiGLU multiplies two rounded factors: the UNFUSED pair measures 1.17e-2 here, the fused one 8.7e-3
    e_f = check(got, ref, tol, f"fused rmsnorm gemm {M}x{N}x{K} {act} vs fp32")
    unf = eng.op_gemm(eng.op_rmsnorm(x, g, eps), W, bias=bias, act=a)
    e_u = check(unf, ref, tol, f"unfused rmsnorm + gemm {M}x{N}x{K} {act} vs fp32")
    assert e_f <= 1.5 * e_u + 1e-3, f"fused path is further from fp32 ({e_f:.3e}) than the unfused one ({e_u:.3e})"
    if M <= 6000:
        for cfg in (21, 22, 82):
            assert torch.equal(got, eng.op_gemm_rows(x, Wf, bias=bias, act=a, rowscale=rs, tile_cfg=cfg)), f"row-scaled gemm differs bitwise on cfg {cfg}"


def test_layernorm_rmsnorm(eng):
    for rows, cols in ((37, 64), (1154, 1024), (9, 4096)):
        x = _rand(f"ln{rows}", (rows, cols), 2.0) + 0.3
        w = _rand(f"lnw{cols}", (cols,), 0.2) + 1.0
        b = _rand(f"lnb{cols}", (cols,), 0.2)
        ref = torch.nn.functional.layer_norm(x, (cols,), w, b, 1e-5)
        check(eng.op_layernorm(x, w, b, 1e-5), ref, 5e-3, f"layernorm {rows}x{cols}")
    for rows, cols in ((37, 64), (2049, 1408), (9, 3072), (3, 8192)):
        x = _rand(f"rn{rows}", (rows, cols), 2.0).to(bf)
        w = (_rand(f"rnw{cols}", (cols,), 0.2) + 1.0).to(bf)
        xf = x.float()
        ref = w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(bf).float()
        check(eng.op_rmsnorm(x, w, 1e-6), ref, 5e-3, f"rmsnorm {rows}x{cols}")


@pytest.mark.parametrize("B,S,H,KV,Dr,causal", [(2, 77, 4, 4, 16, 0), (1, 577, 16, 16, 64, 0), (1, 513, 16, 16, 88, 0), (2, 64, 4, 4, 64, 0),
                                                (1, 200, 8, 8, 96, 1), (1, 333, 8, 2, 128, 1), (1, 1000, 4, 4, 96, 1), (1, 65, 4, 4, 32, 1)])
def test_attention(eng, B, S, H, KV, Dr, causal):
    qkv = _rand(f"att{B}{S}{H}{Dr}", (B * S, (H + 2 * KV) * Dr), 1.0).to(bf)
    scale = Dr ** -0.5
    t = qkv.float().view(B, S, H + 2 * KV, Dr)
    q, k, v = t[:, :, :H], t[:, :, H:H + KV], t[:, :, H + KV:]
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    k = k.repeat_interleave(H // KV, dim=1)
    v = v.repeat_interleave(H // KV, dim=1)
    s = (q @ k.transpose(-1, -2)) * scale
    if causal:
        m = torch.ones(S, S, device=DEV, dtype=torch.bool).tril()
        s = s.masked_fill(~m, float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * S, H * Dr)
    got = eng.op_attention(qkv, B, S, H, KV, Dr, scale, causal)
    check(got, ref, 1.5e-2, f"attention B{B} S{S} H{H}/{KV} D{Dr} causal{causal}")


def test_attention_random_shapes(eng):
    """Seeded random (batch, length, heads, KV heads, head dim, causal) -- lengths around the 64-key page and 128-query block
    boundaries, single-token sequences, grouped-query ratios 1 / 2 / 4 -- against fp32 softmax attention."""
    import random
    rnd = random.Random(20260927)
    cases = [(1, 1, 2, 2, 64, 1), (3, 63, 2, 1, 32, 0), (2, 129, 4, 2, 88, 0), (1, 641, 8, 2, 96, 1), (2, 128, 4, 4, 128, 1), (1, 257, 2, 2, 16, 0)]
    for _ in range(12):
        KV = rnd.choice([1, 2, 4]); H = KV * rnd.choice([1, 2, 4])
        cases.append((rnd.randint(1, 3), rnd.choice([rnd.randint(1, 700), 64 * rnd.randint(1, 6) + rnd.choice([-1, 0, 1])]), H, KV,
                      rnd.choice([16, 32, 64, 88, 96, 128]), rnd.randint(0, 1)))
    for B, S, H, KV, Dr, causal in cases:
        test_attention(eng, B, S, H, KV, Dr, causal)


def test_attention_in_place_operands_are_bit_identical_to_the_paged_path(eng):
    """Non-causal attention reads V (and Q, K at head dim 64) straight from the fused-qkv matrix -- row-major V tiles gathered by
    ds_read_b64_tr_b16 -- instead of going through Q / K pages and a V^T transpose pass (gvl_debug_set vision_in_place = 0): same MFMAs on
    the same values in the same order, so not one output bit may differ; lengths around the 64-key tile boundary, padded head dims, GQA."""
    for B, S, H, KV, Dr in ((1, 577, 16, 16, 64), (2, 2049, 4, 4, 88), (3, 63, 2, 1, 32), (2, 129, 4, 2, 88), (1, 257, 2, 2, 16), (2, 64, 4, 4, 64), (1, 65, 8, 2, 64), (1, 1, 2, 2, 64),
                            (2, 130, 4, 4, 96), (1, 700, 2, 1, 80)):
        qkv = _rand(f"inpl{B}{S}{H}{Dr}", (B * S, (H + 2 * KV) * Dr), 1.0).to(bf)
        got = eng.op_attention(qkv, B, S, H, KV, Dr, Dr ** -0.5, 0)
        eng.debug_set("vision_in_place", 0)
        try:
            ref = eng.op_attention(qkv, B, S, H, KV, Dr, Dr ** -0.5, 0)
        finally:
            eng.debug_set("vision_in_place", 1)
        assert torch.equal(got, ref), (B, S, H, KV, Dr, float((got.float() - ref.float()).abs().max()))


def test_attention_spike_forces_rescale(eng):
    # one key dominates late in the sequence: the online-softmax rescale branch must fire (guide §5.4 rule 26)
    B, S, H, Dr = 1, 300, 2, 64
    qkv = _rand("spike", (S, 3 * H * Dr), 0.3)
    qkv[250, H * Dr: 2 * H * Dr] *= 40.0
    qkv = qkv.to(bf)
    t = qkv.float().view(1, S, 3 * H, Dr)
    q, k, v = (t[:, :, i * H:(i + 1) * H].transpose(1, 2) for i in range(3))
    ref = (torch.softmax((q @ k.transpose(-1, -2)) * Dr ** -0.5, -1) @ v).transpose(1, 2).reshape(S, H * Dr)
    check(eng.op_attention(qkv, B, S, H, H, Dr, Dr ** -0.5, 0), ref, 1.5e-2, "attention with a spiking key")


@pytest.mark.parametrize("N,K", [(1000, 512), (32366, 3072), (9216, 3072), (3072, 8192), (7, 64)])
def test_gemv(eng, N, K):
    W = _rand(f"gv{N}", (N, K), K ** -0.5).to(bf)
    x = _rand(f"gx{K}", (K,), 1.0).to(bf)
    b = _rand(f"gb{N}", (N,), 0.3)
    check(eng.op_gemv(W, x, b), W.float() @ x.float() + b, 2e-4, f"gemv {N}x{K}")


@pytest.mark.parametrize("N,K,B", [(3072, 3072, 1), (3072, 3072, 16), (3072, 8192, 5), (9216, 3072, 16), (16384, 3072, 3), (32366, 3072, 16), (4096, 14336, 7),
                                   (48, 256, 2), (4100, 11008, 16)])
def test_decode_skinny_mfma_gemm(eng, N, K, B):
    """gvl_decode.hip dgemm_kernel (v_mfma_f32_16x16x32_bf16, A = weights, B = activations of up to 16 sequences) vs fp32 torch:
    asymmetric operands (a row / column swap in the D mapping would fail), ragged N (32366 = lm_head + 302 rows), K remainders
    (11008 / 256 = 43 steps: the non-multiple-of-4 tail loop), both row-block widths (N >= 8192: 32 rows per block)."""
    W = _rand(f"dw{N}.{K}", (N, K), K ** -0.5).to(bf)
    x = _rand(f"dx{K}.{B}", (B, K), 1.0).to(bf)
    b = _rand(f"db{N}", (N,), 0.3)
    got = eng.op_dgemm(W, x, b)
    check(got, x.float() @ W.float().T + b, 2e-4, f"skinny decode gemm N={N} K={K} B={B}")
    # a sequence's outputs do not depend on what the other columns of the MFMA hold: bit-identical to its own single-sequence run
    for j in {0, B - 1}:
        assert torch.equal(eng.op_dgemm(W, x[j:j + 1], b)[0], got[j]), f"column {j} of a batch of {B} differs from the same sequence alone"


@pytest.mark.parametrize("M,N,K,epi", [(24588, 1408, 1408, "bias_gamma_resid"), (24588, 4224, 1408, "plain"), (24588, 1408, 6144, "bias_gamma_resid"),
                                       (3519, 3072, 3072, "resid")])
def test_gemm_planner_full_shapes(eng, M, N, K, epi):
    """The launcher's wave-quantisation planner at the real InternVideo2 / Phi shapes: tile-row split (M), N % 256 tail columns on the
    128x128 kernel (N), persistent 256x256 kernel for the rest -- every output element must come from exactly one of the launches."""
    g = torch.Generator(device=DEV); g.manual_seed(M + N + K)
    A = torch.randn((M, K), device=DEV, generator=g).to(bf)
    W = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).to(bf)
    acc = A.float() @ W.float().T
    rb = lambda t: t.to(bf).float()
    if epi == "plain":
        got, ref = eng.op_gemm(A, W), acc
    elif epi == "resid":
        res = torch.randn((M, N), device=DEV, generator=g).to(bf)
        got, ref = eng.op_gemm(A, W, resid=res), res.float() + rb(acc)
    else:
        bias = torch.randn((N,), device=DEV, generator=g) * 0.5
        gam = torch.randn((N,), device=DEV, generator=g) * 0.05 + 0.1
        res = torch.randn((M, N), device=DEV, generator=g).to(bf)
        got, ref = eng.op_gemm(A, W, bias=bias, gamma=gam, resid=res), res.float() + rb(rb(acc + bias) * gam)
    check(got, ref, 8e-3, f"gemm planner {M}x{N}x{K} {epi}")


@pytest.mark.parametrize("M,N,K,epi", [(24588, 1408, 1408, "bias_gamma_resid"), (24588, 4224, 1408, "plain"), (24588, 6144, 1408, "bias_gelu"),
                                       (24588, 1408, 6144, "bias_gamma_resid"), (3519, 16384, 3072, "silu"), (27696, 1024, 4096, "bias_resid32"),
                                       (6924, 3072, 1024, "bias"), (7000, 2560, 1024, "plain")])
def test_gemm_auto_plan_is_bit_identical_and_stream_safe(eng, M, N, K, epi):
    """The automatic launch plan at the real tower shapes (persistent 256x256 kernel + remainder rows / tail columns on the small kernels)
    accumulates every output element in the same k order as the plain 128x128 kernel (cfg 21): bit-identical on every repetition, also
    while a second stream runs its own planned GEMM beside it and the chip is loaded unevenly.  (Round 3 ran the same test against an
    in-kernel in-order split of the last partial round -- green, but slower than the planner: profiles/r03_gemm_split_ab.txt.)"""
    g = torch.Generator(device=DEV); g.manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K), device=DEV, generator=g).to(bf)
    W = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).to(bf)
    kw = {}
    if "bias" in epi:
        kw["bias"] = torch.randn((N,), device=DEV, generator=g) * 0.5
    if "gamma" in epi:
        kw["gamma"] = torch.randn((N,), device=DEV, generator=g) * 0.05 + 0.1
    if "resid32" in epi:
        kw["resid"] = torch.randn((M, N), device=DEV, generator=g); kw["out_f32"] = True
    elif "resid" in epi:
        kw["resid"] = torch.randn((M, N), device=DEV, generator=g).to(bf)
    if "gelu" in epi:
        kw["act"] = L.ACT_GELU
    if "silu" in epi:
        kw["act"] = L.ACT_SILU_MUL
    want = eng.op_gemm(A, W, tile_cfg=21, **kw)
    for rep in range(4):
        assert torch.equal(eng.op_gemm(A, W, **kw), want), f"auto gemm {M}x{N}x{K} {epi}: repetition {rep} differs from cfg 21"
    # a second stream with its own GEMM (other shape) running beside it, and a burst of small kernels as uneven load
    A2 = torch.randn((5000, 2048), device=DEV, generator=g).to(bf)
    W2 = (torch.randn((4096, 2048), device=DEV, generator=g) * 2048 ** -0.5).to(bf)
    want2 = eng.op_gemm(A2, W2, tile_cfg=21)
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    outs, outs2 = [], []
    for rep in range(3):
        with torch.cuda.stream(s2):
            outs2.append(eng.op_gemm(A2, W2))
            junk = [torch.randn((257, 1031), device=DEV) for _ in range(4)]
        outs.append(eng.op_gemm(A, W, **kw))
    torch.cuda.synchronize()
    assert all(torch.equal(o, want) for o in outs) and all(torch.equal(o, want2) for o in outs2), "auto gemm under a concurrent stream differs"


# ---- do_sample=True token selection (gvl_op_sample == the kernel inside every prefill / decode step after gvl_set_sampling) ----------------
@pytest.mark.parametrize("n,temperature,top_k,top_p", [
    (100, 1.0, 0, None), (100, 0.7, 5, None), (1000, 0.2, 50, None), (1000, 1.0, 0, 0.9), (5000, 1.3, 50, 0.5),
    (32366, 0.2, 50, None),           # Phi-3.5 vocabulary, the reference's default sampling configuration
    (32366, 1.0, 0, 0.95), (128558, 0.2, 50, 0.9), (128558, 1.0, 1000, None), (64, 1.0, 1, None), (64, 0.5, 0, 0.01)])
def test_sampler_kept_set_and_draw_match_the_restated_hf_semantics(eng, n, temperature, top_k, top_p):
    """16 rows per launch, each with its own logits, random stream and generation step.  The drawn token must be the oracle's
    (HF warpers restated + the same counter hash); where the oracle's winner leads the runner-up by less than 1e-3 (f32 vs f64 log /
    exp, or a top-p boundary within rounding) the runner-up is accepted too.  A drawn token outside the HF-kept set is always an error."""
    rng = np.random.default_rng(n * 7 + top_k)
    B = 16
    logits = (rng.standard_normal((B, n)) * rng.choice([0.5, 2.0, 4.0], size=(B, 1))).astype(np.float32)
    streams = [int(x) for x in rng.integers(0, 2 ** 31, B)]
    steps = [int(x) for x in rng.integers(0, 4000, B)]
    seed = 0x1234_5678_9ABC_DEF0
    got = eng.op_sample(torch.from_numpy(logits).to(DEV), temperature, top_k, top_p, seed, streams, steps).cpu().tolist()
    soft = 0
    for b in range(B):
        tok, margin, keep = O.sample_token(logits[b], temperature, top_k, top_p, seed, streams[b], steps[b])
        if got[b] == tok:
            continue
        # boundary tokens of the top-p cut may legitimately flip under f32 summation: identify them by their distance to the cut
        s = logits[b].astype(np.float64) / temperature
        p = np.exp(s - s.max()); p = np.where(O.sample_keep_mask(logits[b], temperature, top_k, None), p, 0); p /= p.sum()
        greater = np.array([p[p > p[got[b]]].sum()])
        near_cut = top_p is not None and abs(float(greater[0]) - top_p) < 1e-4
        assert keep[got[b]] or near_cut, f"row {b}: token {got[b]} is outside the kept set"
        assert margin < 1e-3 or near_cut, f"row {b}: drew {got[b]}, oracle {tok} with margin {margin:.3e}"
        soft += 1
    assert soft <= 2, f"{soft} of {B} rows needed the near-tie allowance"


def test_sampler_is_a_pure_function_of_seed_stream_step_and_row(eng):
    """Same (seed, stream, step, logits) -> same token whatever else is in the launch (rows shuffled, batch 1 vs 16); another seed, stream
    or step changes the draws."""
    rng = np.random.default_rng(5)
    n, B = 2000, 16
    logits = torch.from_numpy(rng.standard_normal((B, n)).astype(np.float32)).to(DEV)
    streams, steps = list(range(B)), [3 * b for b in range(B)]
    a = eng.op_sample(logits, 1.0, 0, None, 99, streams, steps).cpu().tolist()
    perm = list(rng.permutation(B))
    b_ = eng.op_sample(logits[perm], 1.0, 0, None, 99, [streams[i] for i in perm], [steps[i] for i in perm]).cpu().tolist()
    assert [b_[perm.index(i)] for i in range(B)] == a
    one = [eng.op_sample(logits[i:i + 1], 1.0, 0, None, 99, [streams[i]], [steps[i]]).cpu().tolist()[0] for i in range(B)]
    assert one == a
    assert eng.op_sample(logits, 1.0, 0, None, 100, streams, steps).cpu().tolist() != a
    assert eng.op_sample(logits, 1.0, 0, None, 99, [s + 1 for s in streams], steps).cpu().tolist() != a
    assert eng.op_sample(logits, 1.0, 0, None, 99, streams, [s + 1 for s in steps]).cpu().tolist() != a


def test_sampler_distribution(eng):
    """8000 draws (16 rows x 500 steps) from a 12-token row at T = 0.7 with top-k 8: empirical frequencies within 4.5 sigma of
    softmax(l / T) renormalised over the 8 kept tokens; the 4 removed tokens are never drawn."""
    l = np.array([0.3, -1.0, 2.0, 1.1, 0.0, -3.0, 1.9, 0.7, -0.2, 2.4, -2.0, 0.9], dtype=np.float32)
    T, K = 0.7, 8
    keep = O.sample_keep_mask(l, T, K, None)
    p = np.where(keep, np.exp(l.astype(np.float64) / T), 0.0); p /= p.sum()
    rows = torch.from_numpy(np.tile(l, (16, 1))).to(DEV)
    cnt = np.zeros(12)
    for it in range(500):
        toks = eng.op_sample(rows, T, K, None, 7, list(range(16)), [it] * 16).cpu().numpy()
        cnt += np.bincount(toks, minlength=12)
    N = cnt.sum()
    assert N == 8000 and cnt[~keep].sum() == 0
    sigma = np.sqrt(p * (1 - p) / N)
    assert np.all(np.abs(cnt / N - p) <= 4.5 * sigma + 1e-9), (cnt / N, p)


@pytest.mark.parametrize("cfg", [0, 21])
def test_gelu_epilogue_is_correctly_rounded_on_every_bf16_input(eng, cfg):
    """The erf-GELU epilogue looks Phi(x) up in a table indexed by the bf16 bit pattern of its (already bf16) input.  Ablation of the
    round-2 suspicion that the table costs accuracy: ALL normal bf16 values (and zero) go through the fused epilogue of both GEMM kernels
    (A carries x in column 0, W row n = e_0, so the accumulator IS x) and must equal the erf GELU of the bf16 value (what nn.GELU
    computes in the reference, internvideo2.py:631-634) evaluated in DOUBLE precision and rounded to bf16 -- on all but a handful of inputs
    that sit within 1e-7 of a rounding boundary (one ulp there)."""
    bits = torch.arange(65536, dtype=torch.int32)
    x = bits.to(torch.int16).view(bf)                                  # every bf16 bit pattern
    # all normal bf16 values up to 2^60 and zero: the MFMA pipe flushes subnormal inputs, and torch's own vectorised erf overflows (gelu = inf)
    # beyond |x| ~ 1e19, where the kernel returns x resp. -0
    xf = x.float()
    finite = torch.isfinite(xf) & (((xf.abs() >= 2.0 ** -125) & (xf.abs() <= 2.0 ** 60)) | (xf == 0))
    K, N = 1024, 256                                                    # cfg 0 -> the persistent 256x256 kernel (LDS-resident table), 21 -> 128x128 (global table)
    A = torch.zeros((65536, K), dtype=bf)
    A[:, 0] = torch.where(finite, x, torch.zeros_like(x))
    W = torch.zeros((N, K), dtype=bf); W[:, 0] = 1.0
    got = eng.op_gemm(A.to(DEV), W.to(DEV), act=L.ACT_GELU, tile_cfg=cfg).cpu()
    # reference in float64 (torch's vectorised CPU gelu is itself only ~1e-3 accurate in the negative tail on some hosts): x * Phi(x) with
    # Phi = erfc(-x / sqrt 2) / 2, rounded to bf16.  The kernel multiplies the bf16 x by an f32 table entry of Phi (6e-8 relative), so it can
    # differ from the correctly rounded value only where x * Phi sits within ~1e-7 of a bf16 rounding boundary: a handful of inputs, by one ulp.
    xd = A[:, 0].double()
    want = (xd * 0.5 * torch.special.erfc(-xd * 2.0 ** -0.5)).float().to(bf)
    # value comparison (-0.0 == +0.0); below x ~ -5.4 the reference's own fp32 arithmetic returns 0 or one quantum of (1 + erf) ~ 6e-8 x |x| / 2: anything
    # within 2^-21 of the true (tiny) value counts as equal there
    diff = (got.float() != want.float()[:, None]) & finite[:, None] & ((got.float() - want.float()[:, None]).abs() > 2.0 ** -21)
    rows = diff.any(dim=1)
    assert (diff == diff[:, :1]).all(), "columns of one row disagree"
    n_bad = int(rows.sum())
    ulp = (got[rows, 0].view(torch.int16).int() - want[rows].view(torch.int16).int()).abs()
    print(f"[parity] erf-GELU epilogue cfg {cfg}: {n_bad} of {int(finite.sum())} bf16 inputs differ from the correctly rounded double-precision value" +
          (f" (by at most {int(ulp.max())} bf16 ulp: x = {A[rows, 0].float().tolist()[:8]})" if n_bad else ""))
    assert n_bad <= 8 and (n_bad == 0 or int(ulp.max()) <= 1), f"cfg {cfg}: {n_bad} inputs differ, max {int(ulp.max()) if n_bad else 0} ulp"


@pytest.mark.parametrize("act", ["quick_gelu", "silu_mul"])
def test_sigmoid_epilogues_on_every_bf16_input(eng, act):
    """CLIP's QuickGELU (x * sigmoid(1.702 x), transformers ACT2FN, modeling_clip.py:340-342) and the SwiGLU product of the decoders
    (up * silu(gate), modeling_phi3.py:246-252 / modeling_llama.py:238) are epilogues of the GEMM kernels built on rcp(1 + exp(-x)) with the
    hardware's fast exp / rcp.  ALL normal bf16 values go through them (accumulator = x exactly, as in the erf-GELU test) and must equal the
    reference's bf16 op sequence -- every op evaluated in DOUBLE and rounded to bf16 where torch rounds -- up to one bf16 ulp on the few
    inputs whose intermediate sits within the fast functions' ~1e-6 of a rounding boundary."""
    bits = torch.arange(65536, dtype=torch.int32)
    x = bits.to(torch.int16).view(bf)
    xf = x.float()
    ok = torch.isfinite(xf) & (((xf.abs() >= 2.0 ** -125) & (xf.abs() <= 2.0 ** 60)) | (xf == 0))
    K = 1024
    A = torch.zeros((65536, K), dtype=bf)
    A[:, 0] = torch.where(ok, x, torch.zeros_like(x))
    xd = A[:, 0].double()
    rb = lambda t: t.float().to(bf).double()                             # round a double through fp32 to bf16 (the fp32 step is exact for these magnitudes' purposes)
    if act == "quick_gelu":
        N = 256
        W = torch.zeros((N, K), dtype=bf); W[:, 0] = 1.0
        got = eng.op_gemm(A.to(DEV), W.to(DEV), act=L.ACT_QUICK_GELU).cpu()
        t1 = rb(torch.tensor(1.702, dtype=torch.float32).double() * xd)   # 1.702f * x in fp32, rounded to bf16
        want = rb(xd * rb(torch.sigmoid(t1))).float().to(bf)
    else:
        N = 512                                                          # interleaved (gate_j, up_j) rows -> 256 outputs; gate = x, up = 1
        A[:, 1] = 1.0
        W = torch.zeros((N, K), dtype=bf); W[0::2, 0] = 1.0; W[1::2, 1] = 1.0
        got = eng.op_gemm(A.to(DEV), W.to(DEV), act=L.ACT_SILU_MUL).cpu()
        want = rb(xd * torch.sigmoid(xd)).float().to(bf)                 # F.silu on a bf16 tensor: one rounding; times up = 1
    assert (got == got[:, :1]).all() or torch.equal(got.view(torch.int16), got[:, :1].view(torch.int16).expand_as(got)), "columns of one row disagree"
    g0 = got[:, 0]
    near = (g0.float() - want.float()).abs() <= 2.0 ** -100              # |x| huge negative: exp overflows to inf -> the kernel returns -0 / 0 like the reference
    bad = ok & (g0.float() != want.float()) & ~near
    n_bad = int(bad.sum())
    ulp = (g0[bad].view(torch.int16).int() - want[bad].view(torch.int16).int()).abs()
    print(f"[parity] {act} epilogue: {n_bad} of {int(ok.sum())} bf16 inputs differ from the double-precision op sequence" +
          (f" (by at most {int(ulp.max())} bf16 ulp; first x = {A[bad, 0].float().tolist()[:6]})" if n_bad else ""))
    assert n_bad <= 8 and (n_bad == 0 or int(ulp.max()) <= 1), f"{act}: {n_bad} inputs differ, max {int(ulp.max()) if n_bad else 0} ulp"
