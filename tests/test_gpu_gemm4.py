"""-m gpu: the 4-wave / AGPR-accumulator GEMM (gvl_gemm4.hip, tile_cfg 84 / 86 / 87 = loop schedule variants) against the 8-wave ping-pong kernel (82) and the
128 x 128 kernel (21) -- BIT-identical (every kernel accumulates an output element in the same k order and shares the epilogue arithmetic) -- and against fp32 torch.
Covers: k-tile counts from the minimum (3) up, odd / even (ring-slot parity carried across the tiles of a persistent workgroup), ragged M and N (rows beyond the
matrix come back as zeros from the buffer bounds check), more tiles than CUs (persistent walk + next-tile prefetch), every fused epilogue the kernel serves."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, bf, check  # noqa: E402
from grounded_video_llm_amd import engine as E, lib as L  # noqa: E402
from gpu_util import tiny_geo  # noqa: E402

A4 = [84, 86, 87]


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(tiny_geo(), DEV, towers=())
    yield e
    e.close()


def _ops(M, N, K, seed):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    A = torch.randn((M, K), device=DEV, generator=g).to(bf)
    W = (torch.randn((N, K), device=DEV, generator=g) * K ** -0.5).to(bf)
    return A, W, g


@pytest.mark.parametrize("M,N,K", [(256, 256, 192), (300, 256, 256), (1000, 1408, 1408), (513, 4224, 320), (77, 72 * 4, 448), (2049, 512, 1024),
                                   (24588, 1408, 384), (70000, 1024, 192), (3519, 3072, 3072)])
def test_plain_bit_identical_to_the_other_kernels(eng, M, N, K):
    A, W, _ = _ops(M, N, K, 7)
    want = eng.op_gemm(A, W, tile_cfg=21)
    assert torch.equal(want, eng.op_gemm(A, W, tile_cfg=82))
    for cfg in A4:
        got = eng.op_gemm(A, W, tile_cfg=cfg)
        bad = (got != want).nonzero()
        assert bad.numel() == 0, f"cfg {cfg} {M}x{N}x{K}: {bad.shape[0]} elements differ, first at {bad[0].tolist()}"
    check(want, A.float() @ W.float().T, 6e-3, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(1000, 1408, 1408), (5000, 512, 192), (3000, 1024, 832)])
def test_every_fused_epilogue_bit_identical(eng, M, N, K):
    A, W, g = _ops(M, N, K, 11)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.5
    gam = torch.randn((N,), device=DEV, generator=g) * 0.1
    resb = torch.randn((M, N), device=DEV, generator=g).to(bf)
    rs = torch.rand((M,), device=DEV, generator=g) + 0.5
    cases = {
        "bias": dict(bias=bias), "bias_qgelu": dict(bias=bias, act=L.ACT_QUICK_GELU), "bias_gelu": dict(bias=bias, act=L.ACT_GELU), "silu": dict(act=L.ACT_SILU_MUL),
        "bias_gamma_resid": dict(bias=bias, gamma=gam, resid=resb), "resid": dict(resid=resb),
    }
    for name, kw in cases.items():
        want = eng.op_gemm(A, W, tile_cfg=82, **kw)
        for cfg in A4:
            assert torch.equal(want, eng.op_gemm(A, W, tile_cfg=cfg, **kw)), f"{name} cfg {cfg}"
    rows = {
        "rowscale": dict(rowscale=rs), "rowscale_silu": dict(rowscale=rs, act=L.ACT_SILU_MUL), "rowscale_bias_gelu": dict(rowscale=rs, bias=bias, act=L.ACT_GELU),
    }
    for name, kw in rows.items():
        want = eng.op_gemm_rows(A, W, tile_cfg=82, **kw)
        for cfg in A4:
            assert torch.equal(want, eng.op_gemm_rows(A, W, tile_cfg=cfg, **kw)), f"{name} cfg {cfg}"
    if N % 64 == 0:
        sq = {"rowsq": dict(), "rowsq_resid": dict(resid=resb), "rowsq_bias_gamma_resid": dict(bias=bias, gamma=gam, resid=resb)}
        for name, kw in sq.items():
            wc, wq = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=82, **kw)
            for cfg in A4:
                c, q = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=cfg, **kw)
                assert torch.equal(wc, c) and torch.equal(wq, q), f"{name} cfg {cfg}"


def test_repeated_launches_are_stable(eng):
    """race screen: the same launch 20 times must give the same bits (a DMA that lands after its reader shows up as a rare differing tile)"""
    A, W, _ = _ops(6000, 2048, 1408, 3)
    want = eng.op_gemm(A, W, tile_cfg=82)
    for cfg in A4:
        for _ in range(20):
            assert torch.equal(want, eng.op_gemm(A, W, tile_cfg=cfg))


# ---- the pipelined-epilogue form (gvl_gemm4p.hip, tile_cfg 88): the drain + deferred program must reproduce the staged epilogue bit for bit ------------------------
P_SHAPES = [(256, 256, 320), (300, 512, 384), (1000, 1408, 1408), (513, 4224, 320), (77, 288, 448), (24588, 1408, 384), (70000, 1024, 320), (3519, 3072, 3072), (40000, 2048, 576)]


@pytest.mark.parametrize("M,N,K", P_SHAPES)
def test_pipelined_plain_bias_rowscale_bit_identical(eng, M, N, K):
    A, W, g = _ops(M, N, K, 5)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.5
    rs = torch.rand((M,), device=DEV, generator=g) + 0.5
    want = eng.op_gemm(A, W, tile_cfg=82)
    got = eng.op_gemm(A, W, tile_cfg=88)
    bad = (got != want).nonzero()
    assert bad.numel() == 0, f"plain {M}x{N}x{K}: {bad.shape[0]} elements differ, first at {bad[0].tolist()}, last at {bad[-1].tolist()}"
    assert torch.equal(eng.op_gemm(A, W, bias=bias, tile_cfg=82), eng.op_gemm(A, W, bias=bias, tile_cfg=88)), "bias"
    assert torch.equal(eng.op_gemm_rows(A, W, rowscale=rs, tile_cfg=82), eng.op_gemm_rows(A, W, rowscale=rs, tile_cfg=88)), "rowscale"


def test_pipelined_repeated_launches_are_stable(eng):
    A, W, _ = _ops(20000, 2048, 1408, 3)
    want = eng.op_gemm(A, W, tile_cfg=82)
    for _ in range(20):
        assert torch.equal(want, eng.op_gemm(A, W, tile_cfg=88))


@pytest.mark.parametrize("M,N,K", [(1000, 1408, 1408), (5000, 512, 1024), (3000, 1024, 1216), (24588, 1408, 1088), (70000, 1024, 1024), (3519, 3072, 3072)])
def test_pipelined_residual_gamma_row_statistics_bit_identical(eng, M, N, K):
    A, W, g = _ops(M, N, K, 13)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.5
    gam = torch.randn((N,), device=DEV, generator=g) * 0.1
    resb = torch.randn((M, N), device=DEV, generator=g).to(bf)
    for name, kw in {"resid": dict(resid=resb), "bias_gamma_resid": dict(bias=bias, gamma=gam, resid=resb)}.items():
        want = eng.op_gemm(A, W, tile_cfg=82, **kw)
        got = eng.op_gemm(A, W, tile_cfg=88, **kw)
        bad = (got != want).nonzero()
        assert bad.numel() == 0, f"{name} {M}x{N}x{K}: {bad.shape[0]} elements differ, first at {bad[0].tolist()}, last at {bad[-1].tolist()}"
    for name, kw in {"rowsq": dict(), "rowsq_resid": dict(resid=resb), "rowsq_bias_gamma_resid": dict(bias=bias, gamma=gam, resid=resb)}.items():
        wc, wq = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=82, **kw)
        c, q = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=88, **kw)
        assert torch.equal(wc, c), f"{name} {M}x{N}x{K}: output"
        bad = (wq != q).nonzero()
        assert bad.numel() == 0 and not torch.isnan(q).any(), f"{name} {M}x{N}x{K}: {bad.shape[0]} row statistics differ, first at {bad[0].tolist() if bad.numel() else None}"


def test_pipelined_in_place_residual(eng):
    """the residual stream may alias C (x += f(x) W): the deferred program reads a tile's residual rows before it stores them"""
    A, W, g = _ops(9000, 1408, 1408, 21)
    x = torch.randn((9000, 1408), device=DEV, generator=g).to(bf)
    want = eng.op_gemm(A, W, resid=x, tile_cfg=82)
    for _ in range(3):
        assert torch.equal(want, eng.op_gemm(A, W, resid=x, tile_cfg=88))


@pytest.mark.parametrize("M,N,K", [(1000, 1408, 1408), (5000, 512, 1024), (3000, 1024, 1216), (24588, 6144, 1088), (3519, 16384, 3072)])
def test_pipelined_activations_bit_identical(eng, M, N, K):
    A, W, g = _ops(M, N, K, 17)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.5
    rs = torch.rand((M,), device=DEV, generator=g) + 0.5
    cases = {"bias_gelu": (eng.op_gemm, dict(bias=bias, act=L.ACT_GELU)), "silu": (eng.op_gemm, dict(act=L.ACT_SILU_MUL)),
             "rowscale_bias_gelu": (eng.op_gemm_rows, dict(rowscale=rs, bias=bias, act=L.ACT_GELU)), "rowscale_silu": (eng.op_gemm_rows, dict(rowscale=rs, act=L.ACT_SILU_MUL))}
    for name, (fn, kw) in cases.items():
        want = fn(A, W, tile_cfg=82, **kw)
        got = fn(A, W, tile_cfg=88, **kw)
        bad = (got != want).nonzero()
        assert bad.numel() == 0, f"{name} {M}x{N}x{K}: {bad.shape[0]} elements differ, first at {bad[0].tolist()}, last at {bad[-1].tolist()}"


def test_pipelined_heavy_epilogues_repeated_launches_are_stable(eng):
    """race screen for the deferred program (LDS staging shared between the drain's bias slice and the transposition passes, residual loads a k-tile ahead of their
    use, stores at the statement's end): the same launch 15 times per epilogue, several tiles per workgroup, must give the same bits every time"""
    M, N, K = 40000, 2816, 1408
    A, W, g = _ops(M, N, K, 23)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.5
    gam = torch.randn((N,), device=DEV, generator=g) * 0.1
    resb = torch.randn((M, N), device=DEV, generator=g).to(bf)
    rs = torch.rand((M,), device=DEV, generator=g) + 0.5
    cases = {"rowsq_bias_gamma_resid": dict(bias=bias, gamma=gam, resid=resb, want_rowsq=True), "rowsq_resid": dict(resid=resb, want_rowsq=True),
             "rowscale_bias_gelu": dict(rowscale=rs, bias=bias, act=L.ACT_GELU), "rowscale_silu": dict(rowscale=rs, act=L.ACT_SILU_MUL)}
    for name, kw in cases.items():
        want = eng.op_gemm_rows(A, W, tile_cfg=82, **kw)
        for it in range(15):
            got = eng.op_gemm_rows(A, W, tile_cfg=88, **kw)
            if isinstance(want, tuple):
                assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1]), f"{name}, launch {it}"
            else:
                assert torch.equal(want, got), f"{name}, launch {it}"


@pytest.mark.parametrize("M,N,K", [(1000, 1408, 1408), (24588, 1408, 1024), (9000, 1344, 1408), (5000, 1472, 1088), (3000, 128, 1408), (70001, 384, 1024), (20000, 1408, 6144), (8000, 1408, 2048),
                                   (6000, 1408, 960)])
def test_narrow_column_tiles_bit_identical(eng, M, N, K):
    """The pipelined kernel runs a column tile with <= 128 real columns as a NARROW tile (4 waves x 128 rows x 64 columns: tools/gen_gemm4p.py body(nb = 2); N = 1408 =
    5.5 tile columns is InternVideo2's proj / fc2).  gvl_debug_set("gemm_narrow", 0 | 1) must not change a bit of the output or of the row statistics, against the 8-wave
    kernel, for tails of 128 / 64 columns (and 192: stays a wide tile), N <= 128 (every tile narrow), K at the narrow statement's minimum (16 k-tiles) and below it (K = 960: wide code), at its maximum (K = 2048) and above it (K = 6144: wide code -- operands from HBM arrive slower than a narrow k-tile runs)."""
    A, W, g = _ops(M, N, K, 29)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.5
    gam = torch.randn((N,), device=DEV, generator=g) * 0.1
    resb = torch.randn((M, N), device=DEV, generator=g).to(bf)
    kw = dict(bias=bias, gamma=gam, resid=resb)
    wc, wq = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=82, **kw)
    try:
        for nar in (1, 2, 0, 1):      # 1: narrow tiles + the balanced tile walk (GemmArgs.rot), 2: narrow tiles on the fixed walk, 0: neither
            eng.debug_set("gemm_narrow", nar)
            for rep in range(2):
                c, q = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=88, **kw)
                bad = (c != wc).nonzero()
                assert bad.numel() == 0, f"gemm_narrow={nar} {M}x{N}x{K}: {bad.shape[0]} outputs differ, first at {bad[0].tolist()}, last at {bad[-1].tolist()}"
                badq = (q != wq).nonzero()
                assert badq.numel() == 0 and not torch.isnan(q).any(), f"gemm_narrow={nar} {M}x{N}x{K}: {badq.shape[0]} row statistics differ, first at {badq[0].tolist() if badq.numel() else None}"
    finally:
        eng.debug_set("gemm_narrow", 1)


def test_narrow_column_tiles_in_place_residual_repeated(eng):
    """race screen: x += LayerScale(att W^T + b) in place with the statistics, N = 1408, many tiles per workgroup, narrow and wide statements alternating inside a workgroup"""
    M, N, K = 60000, 1408, 1408
    A, W, g = _ops(M, N, K, 31)
    bias = torch.randn((N,), device=DEV, generator=g) * 0.5
    gam = torch.randn((N,), device=DEV, generator=g) * 0.1
    x = torch.randn((M, N), device=DEV, generator=g).to(bf)
    wc, wq = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=82, bias=bias, gamma=gam, resid=x)
    for _ in range(12):
        c, q = eng.op_gemm_rows(A, W, want_rowsq=True, tile_cfg=88, bias=bias, gamma=gam, resid=x)
        assert torch.equal(c, wc) and torch.equal(q, wq)
