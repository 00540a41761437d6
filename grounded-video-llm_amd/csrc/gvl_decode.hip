// gvl_decode.hip -- the decode step's projections as ONE skinny MFMA GEMM for up to 16 sequences (gfx950).
//
// Replaces, at q_len = 1, the nn.Linear calls of Phi3DecoderLayer / LlamaDecoderLayer / lm_head (models/modeling_phi3.py:458-464,
// 659-663,770,1512-1526; models/modeling_llama.py:218-238,432-434,497) that transformers' generate() issues once per token and
// sequence.  The path is HBM-bound: 7.45 GB of bf16 weights per Phi-3.5 step.  The round-1 kernel
// (gvl_elem.hip gemv_kernel) multiplies every 16-byte weight chunk into B activation vectors on the VALU -- 8 FMAs per chunk and
// sequence -- and is VALU-bound from B = 4 on (916 tok/s at 4 sequences against 444 at one).  Here the same chunk is the
// A operand of v_mfma_f32_16x16x32_bf16 and the B operand holds the activations of up to 16 sequences: one MFMA per 1 KiB of
// weights whatever the batch, so the weight stream is paid ONCE per step for the whole active set.
//
//   D[i][j] = sum_k W[n0 + i][k] * x[j][k]      i: 16 weight rows (MFMA A rows), j: 16 sequences (MFMA B columns)
//   lane l: A = W[n0 + (l & 15)][k0 + 8 (l >> 4) .. +8], B = x[l & 15][same k]  -- both straight from global memory with one
//   16-byte load (weights: non-temporal, every byte is used once; x: 96 KiB at most, L2 resident), no LDS in the loop.
//   D: lane l holds sequence j = l & 15 and the 4 CONSECUTIVE rows 4 (l >> 4) .. +3 -> both rotate_half partners / both
//   (gate, up) members of a pair sit in one lane (rows are visited in partner order, as the round-1 kernel did).
//
// A block is 8 waves over the same 16 (or 32) rows, each wave one eighth of K; partial sums meet in LDS in a FIXED order.
// A column of D depends only on its own sequence's activations, so a sequence's logits are bit-identical whatever the other
// columns hold -- batched decode == single decode by construction (one kernel for every batch size 1..16).
//
// RMSNorm (in front of qkv_proj / gate_up_proj / lm_head).  Measured alternatives (tools/decode_bench.py, profiles/r02_decode_microbench.txt):
// letting the LAST block of the producing o_proj / down_proj launch normalise the new residual rows (ticket counter + write-through
// stores) costs +13 us (1 sequence) ... +34 us (16) on a 6 us launch -- the store-ack / atomic / re-read round trips do not hide.
// Shipped: groups of up to 4 sequences normalise INSIDE the consumer (wave b of every block normalises row b into LDS while the
// block's first weight loads are in flight; the B operand then comes from LDS); larger groups run a one-wave-per-row norm kernel
// once per projection (2 extra launches per layer on a step that streams 7.45 GB + 16 KV caches).  Both use wave_rmsnorm_row, so a
// sequence's normalised row -- and everything after it -- is bit-identical in either mode.
//
// Memory layout (round 2, measured: with row-major operands the kernel reached 2.7 TB/s -- a 16x16x32 MFMA operand puts 16
// DIFFERENT rows on 16 consecutive lanes, so every quad of lanes touches 4 cache lines and the address path, not HBM, is the limit):
// the decode path reads its own copy of every projection in MFMA-A TILE ORDER -- [row block of 16][k step of 32][64 lanes][8 bf16],
// lane = 16 (k chunk) + row -- built once at gvl_finalize_weights (gvl_retile_decode_weight; the rotate_half row permutation of
// qkv_proj is baked in).  One wave load = 1 KiB of consecutive addresses.  288 GB of HBM pay for the second copy (7.4 GB for
// Phi-3.5, 15 GB for Llama-3-8B); the prefill GEMM keeps the row-major copy its DMA needs.  Activations travel between the decode
// kernels in the matching B-operand tile order ([k step][64 lanes][8], lane = 16 (k chunk) + sequence): gvl_xt_index.
#include "gvl_internal.h"
#include <cstdlib>
#include <cstring>

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)

typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

namespace {

// RMSNorm of a bf16 row, xn = bf16(w * bf16(x * rsqrt(mean(x^2) + eps)))  (Phi3RMSNorm modeling_phi3.py:319-324 / LlamaRMSNorm: fp32
// statistics, result cast to bf16, then the bf16 weight product), in a FIXED arithmetic order shared by the two places that compute
// it -- the consumer block of the skinny GEMM (8 waves, wave w owns k slice w) and the one-wave-per-row norm kernel:
//   the row is cut into DG_SLICES = 8 slices of cols / 8 elements; lane l of a slice owns its 16-byte chunk l (cols <= 4096: at most
//   one chunk per lane); slice sum = wave_sum of the lane sums; row sum = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)).
constexpr int DG_SLICES = 8;
__device__ __forceinline__ float chunk_sumsq(const u32x4_t& v) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) { const float p = lo_bf(v[e]), q = hi_bf(v[e]); s += p * p + q * q; }
  return s;
}
__device__ __forceinline__ float row_rstd(const float (&p)[DG_SLICES], int cols, float eps) {
  const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  return rsqrtf(s / (float)cols + eps);
}
__device__ __forceinline__ u32x4_t chunk_normalise(const u32x4_t& v, const u32x4_t& w, float rs) {
  u32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(w[e]) * rbf(lo_bf(v[e]) * rs), hi_bf(w[e]) * rbf(hi_bf(v[e]) * rs));
  return o;
}

}  // namespace

// One wave per sequence: source row = table[*tok[b]] (embedding gather; the raw row also goes to the row-major residual stream x)
// or, without a table, x[b] itself; xn = rmsnorm(row) * w lands in slot b of the tiled activation buffer.
__global__ __launch_bounds__(64) void norm_rows_kernel(const bf16_t* __restrict__ table, const TokPtrs toks, bf16_t* __restrict__ x, bf16_t* __restrict__ xn,
                                                       const bf16_t* __restrict__ w, int cols, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  bf16_t* xrow = x + (size_t)b * cols;
  const bf16_t* src = table ? table + (size_t)(*toks.p[b]) * cols : xrow;
  const int ks = cols / DG_SLICES, nch = ks >> 3;          // slice length, 16-byte chunks per slice (<= 64)
  const bool on = lane < nch;
  u32x4_t v[DG_SLICES];
  float p[DG_SLICES];
#pragma unroll
  for (int sl = 0; sl < DG_SLICES; ++sl) v[sl] = on ? *(const u32x4_t*)(src + sl * ks + lane * 8) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
  for (int sl = 0; sl < DG_SLICES; ++sl) p[sl] = wave_sum(on ? chunk_sumsq(v[sl]) : 0.f);
  const float rs = row_rstd(p, cols, eps);
  if (!on) return;
#pragma unroll
  for (int sl = 0; sl < DG_SLICES; ++sl) {
    const int k = sl * ks + lane * 8;
    *(u32x4_t*)(xn + gvl_xt_index(b, k)) = chunk_normalise(v[sl], *(const u32x4_t*)(w + k), rs);   // 8 consecutive k = one 16-byte slot of the tile
    if (table) *(u32x4_t*)(xrow + k) = v[sl];
  }
}
int gvl_launch_embed_norm(const bf16_t* table, const TokPtrs& toks, bf16_t* x, bf16_t* xn, const bf16_t* w, int cols, float eps, hipStream_t st) {
  if (cols % 256 || cols > 4096 || toks.n < 1 || toks.n > GVL_MAX_DECODE_BATCH || !table) return -1;
  hipLaunchKernelGGL(norm_rows_kernel, dim3(toks.n), dim3(64), 0, st, table, toks, x, xn, w, cols, eps);
  return CHECK_LAUNCH();
}
int gvl_launch_norm_tiled(bf16_t* x, bf16_t* xn, const bf16_t* w, int batch, int cols, float eps, hipStream_t st) {
  if (cols % 256 || cols > 4096 || batch < 1 || batch > GVL_MAX_DECODE_BATCH) return -1;
  TokPtrs none; memset(&none, 0, sizeof(none)); none.n = batch;
  hipLaunchKernelGGL(norm_rows_kernel, dim3(batch), dim3(64), 0, st, (const bf16_t*)nullptr, none, x, xn, w, cols, eps);
  return CHECK_LAUNCH();
}

// RB: 16-row blocks per workgroup; NW: waves per workgroup = k slices (8 or 4); U: MFMA steps (32 k each) per load group;
// NT: weight loads carry the non-temporal hint; XN: the B operand is RMS-normalised by this block into LDS (batch <= 4, a.x row-major);
// W8: the weights are the FP8 (OCP e4m3, per-row power-of-two scale) tile copy -- 16 bytes per lane feed TWO k steps; the values are
// widened to bf16 in registers (exact) and the row scale multiplies the fp32 sum (exact): same arithmetic as bf16 weights holding
// the de-quantised values, at half the HBM bytes.  W8 = 2: the MXFP4 tile copy (OCP Microscaling v1.0: E2M1 elements, one
// E8M0 scale per 32 consecutive k = one k step of one row) -- 16 bytes per lane feed FOUR k steps plus one 4-byte word of scales;
// v_cvt_scalef32_pk_bf16_fp4 widens two elements per instruction (exact: <= 2 significant bits times a power of two); a quarter of
// the HBM bytes.
template <int RB, int NW, int U, int NT, int XN, int W8>
__global__ __launch_bounds__(NW * 64, U >= 8 ? 2 : 4) void dgemm_kernel(const GemvArgs a) {   // U = 4: 4 waves / SIMD (<= 128 VGPRs); U = 8 (few-block launches): 2
  __shared__ __attribute__((aligned(16))) float red[NW][RB][64][4];
  extern __shared__ __attribute__((aligned(16))) char xs_raw[];                  // XN: [k step][k chunk][batch][8] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (16 * RB);
  const int kw = a.K / NW;                       // k range of one wave
  const int halfd = a.Dr >> 1, npair_qk = (a.H + a.KV) * halfd;

  // operands in tile order: step s of this wave is the 1 KiB tile (row block, wave * steps + s); lane l reads bytes [16 l, 16 l + 16)
  const int steps = kw >> 5;
  const int nkt = a.K >> 5;                      // k steps per row block
  const bf16_t* wp[RB];
  const unsigned* sp[RB];                        // MXFP4: this lane's scale words (row i of the row block; one word = 4 consecutive k steps)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    sp[rb] = nullptr;
    int rbk = blockIdx.x * RB + rb;
    const int rbk_max = (a.N + 15) / 16 - 1;
    rbk = rbk < rbk_max ? rbk : rbk_max;         // a block past the end re-reads the last row block; its rows are never stored
    wp[rb] = W8 == 2 ? (const bf16_t*)((const char*)a.W + ((size_t)rbk * nkt + (size_t)wave * steps) * 256 + lane * 16)   // 256 bytes per k step
           : W8 == 1 ? (const bf16_t*)((const char*)a.W + ((size_t)rbk * nkt + (size_t)wave * steps) * 512 + lane * 16)   // 512 bytes per k step
                     : a.W + ((size_t)rbk * nkt + (size_t)wave * steps) * 512 + lane * 8;
    if constexpr (W8 == 2) sp[rb] = (const unsigned*)a.wscale + ((size_t)rbk * nkt + (size_t)wave * steps) * 4 + i;   // [row block][4 steps][16 rows] words
  }
  // B columns >= batch re-read the LAST sequence's chunk: a D column depends on its own B column only and those columns
  // are never stored, so no masking is needed in the loop (the duplicate addresses coalesce / broadcast)
  const int jj = i < a.batch ? i : a.batch - 1;
  const bf16_t* xp = XN ? (const bf16_t*)xs_raw + ((size_t)wave * steps * 4 + g) * a.batch * 8 + jj * 8
                        : a.x + (size_t)wave * steps * 512 + (g * 16 + jj) * 8;
  const int xstep = XN ? 4 * a.batch * 8 : 512;  // elements between two k steps of the B operand

  f32x4v_t acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
  // groups of U steps, double buffered: while the U x RB MFMAs of group t run, the U x (RB + 1) 16-byte loads of group t + 1 are
  // in flight.  steps = K / 256 is a multiple of U = 4 for every shipped geometry (3072 -> 12, 4096 -> 16, 8192 -> 32,
  // 14336 -> 56); the remainder loop covers anything else.
  bf16x8_t wv[2][U][RB], xv[2][U];
  auto widen = [](unsigned lo, unsigned hi) {     // 8 fp8 (e4m3) -> 8 bf16, exact
    typedef float f32x2v_t __attribute__((ext_vector_type(2)));
    const f32x2v_t a0 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2v_t b0 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const u32x4_t p = {pack2bf(a0[0], a0[1]), pack2bf(a1[0], a1[1]), pack2bf(b0[0], b0[1]), pack2bf(b1[0], b1[1])};
    return __builtin_bit_cast(bf16x8_t, p);
  };
  u32x4_t w8[2][U / 2][RB];                       // W8: raw 16-byte pieces (two k steps each), widened at consumption
  u32x4_t w4[2][RB]; unsigned s4[2][RB];          // MXFP4: one 16-byte piece (four k steps) + their four E8M0 scales per load group
  static_assert(W8 != 2 || U == 4, "MXFP4: a load group is exactly one 16-byte piece");
  auto widen4 = [](unsigned word, unsigned e8) {  // 8 E2M1 nibbles x 2^(e8 - 127) -> 8 bf16, exact
    typedef __bf16 bf16x2v_t __attribute__((ext_vector_type(2)));
    const float sc = __uint_as_float(e8 << 23);
    const bf16x2v_t p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(word, sc, 0), p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(word, sc, 1);
    const bf16x2v_t p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(word, sc, 2), p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(word, sc, 3);
    const u32x4_t p = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
    return __builtin_bit_cast(bf16x8_t, p);
  };
  auto request_w = [&](int buf, int s) {
    if constexpr (W8 == 2) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        w4[buf][rb] = __builtin_nontemporal_load((const u32x4_t*)((const char*)wp[rb] + (size_t)(s / 4) * 1024));
        s4[buf][rb] = sp[rb][(size_t)(s / 4) * 16];
      }
    } else if constexpr (W8 == 1) {
#pragma unroll
      for (int u = 0; u < U / 2; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) w8[buf][u][rb] = __builtin_nontemporal_load((const u32x4_t*)((const char*)wp[rb] + (size_t)(s / 2 + u) * 1024));
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) wv[buf][u][rb] = NT ? __builtin_nontemporal_load((const bf16x8_t*)(wp[rb] + (s + u) * 512)) : *(const bf16x8_t*)(wp[rb] + (s + u) * 512);
    }
  };
  auto request_x = [&](int buf, int s) {
#pragma unroll
    for (int u = 0; u < U; ++u) xv[buf][u] = *(const bf16x8_t*)(xp + (s + u) * xstep);
  };
  auto request = [&](int buf, int s) { request_w(buf, s); request_x(buf, s); };
  auto consume = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        bf16x8_t wa;
        if constexpr (W8 == 2) wa = widen4(w4[buf][rb][u], (s4[buf][rb] >> (8 * u)) & 255u);
        else if constexpr (W8 == 1) wa = (u & 1) ? widen(w8[buf][u / 2][rb][2], w8[buf][u / 2][rb][3]) : widen(w8[buf][u / 2][rb][0], w8[buf][u / 2][rb][1]);
        else wa = wv[buf][u][rb];
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xv[buf][u], acc[rb], 0, 0, 0);
      }
  };
  const int groups = steps / U;
  if (groups > 0) request_w(0, 0);               // the first weight tiles do not depend on the activations: in flight during the norm
                                                 // (requesting the second buffer's tiles here as well: measured +-0)
  // fused RMSNorm, consumer side: the epilogue waves add up their sequence's partial sums of squares (written by the projection that produced the
  // residual rows) behind the first weight requests -- lane (g, i) takes a quarter of sequence i's sq_n partials in index order, the four quarters meet
  // as (q0 + q1) + (q2 + q3): a fixed order that involves sequence i alone
  // EVERY wave takes a slice: lane (g, i) of wave w adds the sq_n / (4 NW) partials [(4 w + g) per, (4 w + g + 1) per) of sequence i in index order -- a
  // handful of L2-resident loads that are consumed only at the block's closing reduction, where the epilogue wave adds the 4 NW slice sums in a fixed order
  // (doing all of it in the epilogue wave up front cost -5 % tok/s at one sequence: 48 loads ahead of its main loop)
  float sq_mine = 0.f;
  if (a.sq_in) {
    const int b_ = i < a.batch ? i : a.batch - 1, per = a.sq_n / (4 * NW);
    const float* sp_ = a.sq_in + (size_t)b_ * a.sq_n + (4 * wave + g) * per;
    for (int j = 0; j < per; ++j) sq_mine += sp_[j];
  }
  if constexpr (XN) {
    // every wave normalises ITS k slice of the (<= 4) residual rows into LDS, B-operand order [step][chunk][batch][8]: element (b, k)
    // at (((k >> 5) * 4 + ((k >> 3) & 3)) * batch + b) * 8 + (k & 7).  Only the 8 slice sums cross waves (one barrier); a wave
    // reads back only what it wrote itself.
    static_assert(!XN || NW == DG_SLICES, "fused RMSNorm: one wave per k slice");
    __shared__ float psum[GVL_MAX_VALU_BATCH][DG_SLICES];
    bf16_t* xs = (bf16_t*)xs_raw;
    const int nb = a.batch, nch = kw >> 3;
    const bool on = lane < nch;
    u32x4_t xr[GVL_MAX_VALU_BATCH];
#pragma unroll
    for (int b = 0; b < GVL_MAX_VALU_BATCH; ++b) {
      if (b < nb) {
        xr[b] = on ? *(const u32x4_t*)(a.x + (size_t)b * a.x_stride + wave * kw + lane * 8) : u32x4_t{0u, 0u, 0u, 0u};
        const float ps = wave_sum(on ? chunk_sumsq(xr[b]) : 0.f);
        if (lane == 0) psum[b][wave] = ps;
      }
    }
    __syncthreads();
    if (on) {
      const int k = wave * kw + lane * 8;
      const u32x4_t wn = *(const u32x4_t*)(a.norm_w + k);
#pragma unroll
      for (int b = 0; b < GVL_MAX_VALU_BATCH; ++b) {
        if (b < nb) {
          float p[DG_SLICES];
#pragma unroll
          for (int sl = 0; sl < DG_SLICES; ++sl) p[sl] = psum[b][sl];
          *(u32x4_t*)(xs + ((size_t)((k >> 5) * 4 + ((k >> 3) & 3)) * nb + b) * 8) = chunk_normalise(xr[b], wn, row_rstd(p, a.K, a.eps));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();              // the B operand of this wave is its own LDS writes (program order within the wave)
  }
  if (groups > 0) request_x(0, 0);
  int gi = 0;
  for (; gi + 2 <= groups; gi += 2) {            // two groups per trip: the buffer index stays a compile-time constant
    request(1, (gi + 1) * U);
    consume(0);
    if (gi + 2 < groups) request(0, (gi + 2) * U);
    consume(1);
  }
  if (gi < groups) consume(0);
  for (int s0 = groups * U; s0 < steps; ++s0) {
    const bf16x8_t x1 = *(const bf16x8_t*)(xp + s0 * xstep);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      bf16x8_t wa;
      if constexpr (W8 == 2) {                     // (never taken: K % 1024 == 0 makes steps a multiple of U = 4)
        wa = widen4(*(const unsigned*)((const char*)wp[rb] + (size_t)(s0 / 4) * 1024 + (s0 & 3) * 4), (sp[rb][(size_t)(s0 / 4) * 16] >> (8 * (s0 & 3))) & 255u);
      } else if constexpr (W8 == 1) {              // (steps is even in W8 mode: K % 512 == 0)
        const u32x2_t piece = *(const u32x2_t*)((const char*)wp[rb] + (size_t)(s0 / 2) * 1024 + (s0 & 1) * 8);
        wa = widen(piece[0], piece[1]);
      } else wa = __builtin_nontemporal_load((const bf16x8_t*)(wp[rb] + s0 * 512));
      acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, x1, acc[rb], 0, 0, 0);
    }
  }
  // partial sums of the 8 k-slices meet in LDS; wave rb (< RB) adds them in a fixed order and runs that row block's epilogue
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) *(f32x4v_t*)red[wave][rb][lane] = acc[rb];
  __shared__ float sqred[NW][64];
  if (a.sq_in) sqred[wave][lane] = sq_mine;
  __syncthreads();
  if (wave < RB) {
    float rs_col = 1.f;
    if (a.sq_in) {                                  // slice sums of this lane's sequence i: waves in order, the four g of a wave as (g0 + g1) + (g2 + g3)
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += (sqred[w][i] + sqred[w][16 + i]) + (sqred[w][32 + i] + sqred[w][48 + i]);
      rs_col = rsqrtf(tot / (float)a.K + a.eps);
    }
    const int rb = wave;
    float v[4];
    {
      auto ld = [&](int w) { return *(const f32x4v_t*)red[w][rb][lane]; };
      f32x4v_t q = (ld(0) + ld(1)) + (ld(2) + ld(3));             // FIXED association: the result does not depend on anything
      if constexpr (NW == 8) q = q + ((ld(4) + ld(5)) + (ld(6) + ld(7)));   // but the partial sums themselves
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = q[r];
    }
    if constexpr (W8 == 1) {                      // per-row power-of-two scale of the FP8 copy (exact in fp32)
      const int n4 = n0 + rb * 16 + g * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= a.wscale[n4 + r < a.N ? n4 + r : a.N - 1];
    }
    if (a.sq_in) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rs_col;   // the RMS factor of this lane's sequence: the weight carries the norm weight, x came in raw
    }
    const int b = i;                             // this lane's sequence
    const int nr = n0 + rb * 16 + g * 4;         // its 4 consecutive logical rows nr .. nr + 3
    if (b < a.batch && nr < a.N) {
      if (a.rope_on) {
        // fused decode epilogue of the qkv projection: RoPE at the sequence's position (apply_rotary_pos_emb modeling_phi3.py:421-445,
        // short / long factors by kv length :382-385), q -> Q[b][H][D], k / v appended to the sequence's pages
        const int pos = *a.pos_ptrs[b];
        const float* cosp = a.cos_s; const float* sinp = a.sin_s;
        if (a.rope_switch > 0 && pos + 1 > a.rope_switch) { cosp = a.cos_l; sinp = a.sin_l; }
        const int page = a.tables[b][pos >> 6], slot = pos & 63;
        bf16_t* Qb = a.Q + (size_t)b * a.q_stride;
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const int n = nr + r;
          if (n + 1 < a.N) {
            const int j = n >> 1;
            if (j < npair_qk) {
              const int hd = j / halfd, d = j - hd * halfd;
              const float x1 = rbf(v[r]), x2 = rbf(v[r + 1]);
              const float c = cosp[(size_t)pos * halfd + d], sn = sinp[(size_t)pos * halfd + d];
              const bf16_t o1 = f2bf(rbf(x1 * c) + rbf(-x2 * sn)), o2 = f2bf(rbf(x2 * c) + rbf(x1 * sn));
              bf16_t* dst = hd < a.H ? Qb + (size_t)hd * a.D + d
                                     : a.Kt + (((size_t)page * a.KV + (hd - a.H)) * 64 + slot) * a.D + d;
              dst[0] = o1; dst[halfd] = o2;
            } else {
              const int vi = n - 2 * npair_qk, hv = vi / a.Dr, d = vi - hv * a.Dr;
              bf16_t* dst = a.Vt + (((size_t)page * a.KV + hv) * a.D + d) * 64 + slot;
              dst[0] = f2bf(v[r]); dst[64] = f2bf(v[r + 1]);
            }
          }
        }
      } else if (a.act == GVL_ACT_SILU_MUL) {    // interleaved (gate, up) rows: up * silu(gate), each op rounded to bf16 (Phi3MLP :458-464)
        float o[2];
#pragma unroll
        for (int r = 0; r < 4; r += 2) { const float gt = rbf(v[r]), u = rbf(v[r + 1]); o[r >> 1] = u * rbf(gt * fast_sigmoid(gt)); }
        if (nr + 3 < a.N) {
          if (a.out_bf16) *(unsigned*)(a.out_bf16 + (a.out_tiled ? gvl_xt_index(b, nr >> 1) : (size_t)b * a.out_stride + (nr >> 1))) = pack2bf(o[0], o[1]);
          if (a.out_f32) { a.out_f32[(size_t)b * a.out_stride + (nr >> 1)] = o[0]; a.out_f32[(size_t)b * a.out_stride + (nr >> 1) + 1] = o[1]; }
        } else if (nr + 1 < a.N) {
          if (a.out_bf16) a.out_bf16[a.out_tiled ? gvl_xt_index(b, nr >> 1) : (size_t)b * a.out_stride + (nr >> 1)] = f2bf(o[0]);
          if (a.out_f32) a.out_f32[(size_t)b * a.out_stride + (nr >> 1)] = o[0];
        }
      } else {
        if (a.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (nr + r < a.N) v[r] += a.bias[nr + r];
        }
        if (nr + 3 < a.N) {
          if (a.resid) {
            const u32x2_t rv = *(const u32x2_t*)(a.resid + (size_t)b * a.out_stride + nr);
            v[0] = lo_bf(rv[0]) + rbf(v[0]); v[1] = hi_bf(rv[0]) + rbf(v[1]); v[2] = lo_bf(rv[1]) + rbf(v[2]); v[3] = hi_bf(rv[1]) + rbf(v[3]);
          }
          if (a.out_bf16) {
            const unsigned p01 = pack2bf(v[0], v[1]), p23 = pack2bf(v[2], v[3]);
            const unsigned long long o = (unsigned long long)p01 | ((unsigned long long)p23 << 32);
            unsigned long long* dst = (unsigned long long*)(a.out_bf16 + (size_t)b * a.out_stride + nr);
            *dst = o;
            if (a.out_tiled2) *(unsigned long long*)(a.out_tiled2 + gvl_xt_index(b, nr)) = o;   // 4 consecutive k of one sequence: 8 contiguous bytes of the tile
            if (a.sq_out) {                      // fused RMSNorm, producer side: sum of squares of the ROUNDED outputs of this 16-row block, per sequence
              float ss = lo_bf(p01) * lo_bf(p01);
              ss += hi_bf(p01) * hi_bf(p01); ss += lo_bf(p23) * lo_bf(p23); ss += hi_bf(p23) * hi_bf(p23);
              const float q0 = __shfl(ss, i, 64), q1 = __shfl(ss, 16 + i, 64), q2 = __shfl(ss, 32 + i, 64), q3 = __shfl(ss, 48 + i, 64);   // the 4 lanes of sequence i
              if (g == 0) a.sq_out[(size_t)b * (a.N >> 4) + (blockIdx.x * RB + rb)] = (q0 + q1) + (q2 + q3);
            }
          }
          if (a.out_f32) *(f32x4v_t*)(a.out_f32 + (size_t)b * a.out_stride + nr) = f32x4v_t{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = nr + r;
            if (n < a.N) {
              float y = v[r];
              if (a.resid) y = bf2f(a.resid[(size_t)b * a.out_stride + n]) + rbf(y);
              if (a.out_bf16) a.out_bf16[(size_t)b * a.out_stride + n] = f2bf(y);
              if (a.out_f32) a.out_f32[(size_t)b * a.out_stride + n] = y;
            }
          }
        }
      }
    }
  }
}

// Skinny-GEMM decode projection.  a.W = the TILED copy of the weight (gvl_retile_decode_weight).  a.norm_w == null: a.x = activations
// in B-operand tile order (gvl_xt_index); a.norm_w != null (batch <= 4): a.x = the row-major residual rows, normalised by every block
// into LDS.  a.out_tiled: out_bf16 of the SwiGLU epilogue in tile order (it feeds down_proj).  Preconditions (else -1): K % 256 == 0.
int gvl_launch_dgemm(const GemvArgs& a_in, hipStream_t st) {
  GemvArgs a = a_in;
  if (a.batch <= 0) a.batch = 1;
  if (a.K % 256 || a.K <= 0 || a.N <= 0 || a.batch > GVL_MAX_DECODE_BATCH) return -1;
  if (((uintptr_t)a.x & 15) || ((uintptr_t)a.W & 15)) return -1;
  if (a.out_tiled && a.act != GVL_ACT_SILU_MUL) return -1;
  if (a.norm_w && (a.batch > GVL_MAX_VALU_BATCH || a.K > 4096 || (a.batch > 1 && a.x_stride % 4))) return -1;
  if (a.rope_on && ((a.Dr & 1) || (a.N & 3))) return -1;
  if (a.sq_in && (a.norm_w || a.w_fp8 || a.sq_n <= 0 || (a.sq_n & 31))) return -1;                // fused RMSNorm: raw tiled x, folded bf16 weights, 4 x NW (<= 32) equal slices
  if ((a.sq_out || a.out_tiled2) && ((a.N & 15) || !a.out_bf16 || a.act == GVL_ACT_SILU_MUL || a.rope_on)) return -1;
  if (a.act == GVL_ACT_SILU_MUL && (a.N & 3)) return -1;
  // variant (experiments: GVL_DGEMM_VARIANT = rb*1000 + nw*100 + u*10 + nt; 0 = the measured default)
  static const int env_variant = [] { const char* e = gvl_lab_env("GVL_DGEMM_VARIANT"); return e ? atoi(e) : 0; }();
  int variant = a.variant ? a.variant : env_variant;
  if (variant == 0) {                              // measured (tools/decode_bench.py, profiles/r02_decode_microbench.txt)
    variant = (a.N >= 16384 ? 2000 : 1000) + 800 + 40 + 1;
    // few blocks x long k (Phi down_proj: 192 blocks of 32 k steps per wave): 8-step load groups at 2 waves / SIMD keep twice the bytes
    // in flight per wave, 12.3 -> 11.3 us; every other shape is slower with them (more blocks than slots already).  Same MFMA order.
    if (!a.w_fp8 && !a.norm_w && a.N <= 3072 && a.K >= 8192 && a.K % 2048 == 0) variant = 1881;
  }
  const int RB = variant / 1000, NW = (variant / 100) % 10;
  if (a.K % (NW * 32)) return -1;
  const int blocks = (a.N + 16 * RB - 1) / (16 * RB);
  const size_t lds = a.norm_w ? (size_t)a.batch * a.K * 2 : 0;
  if (a.w_fp8 < 0 || a.w_fp8 > 2 || (a.w_fp8 && (a.K % (a.w_fp8 == 2 ? 1024 : 512) || !a.wscale))) return -1;
  const int key = variant * 100 + (a.norm_w ? 10 : 0) + a.w_fp8;
  switch (key) {
#define DG_CASE(rb, nw, u, nt, xn, w8) case (rb * 1000 + nw * 100 + u * 10 + nt) * 100 + xn * 10 + w8: \
      hipLaunchKernelGGL((dgemm_kernel<rb, nw, u, nt, xn, w8>), dim3(blocks), dim3(nw * 64), lds, st, a); break;
    DG_CASE(1, 8, 4, 1, 0, 0) DG_CASE(2, 8, 4, 1, 0, 0) DG_CASE(1, 4, 4, 1, 0, 0) DG_CASE(2, 4, 4, 1, 0, 0) DG_CASE(1, 8, 2, 1, 0, 0) DG_CASE(2, 8, 2, 1, 0, 0)
    DG_CASE(1, 8, 4, 1, 1, 0) DG_CASE(2, 8, 4, 1, 1, 0) DG_CASE(1, 8, 8, 1, 0, 0)
    DG_CASE(1, 8, 4, 1, 0, 1) DG_CASE(2, 8, 4, 1, 0, 1) DG_CASE(1, 8, 4, 1, 1, 1) DG_CASE(2, 8, 4, 1, 1, 1)
    DG_CASE(1, 8, 4, 1, 0, 2) DG_CASE(2, 8, 4, 1, 0, 2) DG_CASE(1, 8, 4, 1, 1, 2) DG_CASE(2, 8, 4, 1, 1, 2)
#undef DG_CASE
    default: return -1;
  }
  return CHECK_LAUNCH();
}

// ---- tile-order copies --------------------------------------------------------------------------------------------------------
// W [N][K] row-major -> Wt [ceil(N/16)][K/32][64][8]: lane l of tile (rbk, s) = row rbk*16 + (l & 15) (through the rotate_half pair
// permutation when Dr > 0: logical rows (2j, 2j+1) = weight rows (hd*Dr + d, hd*Dr + d + Dr/2) for the q and k heads), k chunk l >> 4.
// Rows >= N are zero.
__global__ void retile_weight_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ Wt, int N, int K, int Dr, int n_qk_heads) {
  const long total = (long)((N + 15) / 16) * (K / 32) * 64;
  const int halfd = Dr >> 1, npair_qk = n_qk_heads * halfd;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int l = (int)(idx & 63);
    const long tile = idx >> 6;
    const int s = (int)(tile % (K / 32));
    const int rbk = (int)(tile / (K / 32));
    int n = rbk * 16 + (l & 15);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (n < N) {
      if (Dr > 0) { const int j = n >> 1; if (j < npair_qk) { const int hd = j / halfd, d = j - hd * halfd; n = hd * Dr + d + (n & 1) * halfd; } }
      v = *(const u32x4_t*)(W + (size_t)n * K + s * 32 + (l >> 4) * 8);
    }
    *(u32x4_t*)(Wt + (size_t)idx * 8) = v;
  }
}
int gvl_retile_decode_weight(const bf16_t* W, bf16_t* Wt, int N, int K, int Dr, int n_qk_heads, hipStream_t st) {
  if (K % 32 || N <= 0) return -1;
  const long total = (long)((N + 15) / 16) * (K / 32) * 64;
  long blocks = (total + 255) / 256; if (blocks > 65535 * 8) blocks = 65535 * 8;
  hipLaunchKernelGGL(retile_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, st, W, Wt, N, K, Dr, n_qk_heads);
  return CHECK_LAUNCH();
}
// x [batch][cols] row-major -> tile order (operator-level tests)
__global__ void rows_to_tiled_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ xt, int batch, int cols, int stride) {
  const long total = (long)batch * cols;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / cols), k = (int)(idx - (long)b * cols);
    xt[gvl_xt_index(b, k)] = x[(size_t)b * stride + k];
  }
}
int gvl_launch_rows_to_tiled(const bf16_t* x, bf16_t* xt, int batch, int cols, int stride, hipStream_t st) {
  if (cols % 32 || batch < 1 || batch > GVL_MAX_DECODE_BATCH) return -1;
  hipLaunchKernelGGL(rows_to_tiled_kernel, dim3((unsigned)(((long)batch * cols + 255) / 256)), dim3(256), 0, st, x, xt, batch, cols, stride);
  return CHECK_LAUNCH();
}

// ---- FP8 (OCP e4m3) weight variant of the decode copies ---------------------------------------------------------------
// Per LOGICAL row n (rotate_half pair order when Dr > 0): scale = 2^ceil(log2(max|w| / 448)) -- a power of two, so w_q = fp8(w / scale)
// * scale is exactly representable in bf16 and the row-major bf16 weight is REPLACED by it: prefill (bf16 GEMM) and decode (FP8 stream)
// then evaluate the same model, and every invariant of the bf16 path (decode == prefill, batch invariance) carries over.
__device__ __forceinline__ int decode_row_perm(int n, int N, int Dr, int n_qk_heads) {
  if (Dr > 0) { const int halfd = Dr >> 1, j = n >> 1; if (j < n_qk_heads * halfd) { const int hd = j / halfd, d = j - hd * halfd; return hd * Dr + d + (n & 1) * halfd; } }
  return n;
}
__global__ __launch_bounds__(64) void fp8_row_scale_kernel(const bf16_t* __restrict__ W, float* __restrict__ scale, int N, int K, int Dr, int n_qk_heads) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const bf16_t* row = W + (size_t)decode_row_perm(n, N, Dr, n_qk_heads) * K;
  float m = 0.f;
  for (int c = lane; c < (K >> 3); c += 64) {
    const u32x4_t v = *(const u32x4_t*)(row + c * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(lo_bf(v[e])), fabsf(hi_bf(v[e]))));
  }
  m = wave_max(m);
  if (lane == 0) {
    float s = 1.f;
    if (m > 0.f) { int ex; const float fr = frexpf(m / 448.f, &ex); s = ldexpf(1.f, fr == 0.5f ? ex - 1 : ex); }   // smallest power of two >= m / 448
    scale[n] = s;
  }
}
// one thread per 16-byte piece of the FP8 tile copy [row block][k step pair][64 lanes][16]: bytes 0..7 = k chunk (l >> 4) of step 2p,
// bytes 8..15 = the same chunk of step 2p + 1; the source elements are overwritten with their de-quantised values
__global__ void fp8_requant_retile_kernel(bf16_t* __restrict__ W, unsigned char* __restrict__ Wt8, const float* __restrict__ scale, int N, int K, int Dr, int n_qk_heads) {
  const long total = (long)((N + 15) / 16) * (K / 64) * 64;
  typedef float f32x2v_t __attribute__((ext_vector_type(2)));
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int l = (int)(idx & 63);
    const long tile = idx >> 6;
    const int p = (int)(tile % (K / 64)), rbk = (int)(tile / (K / 64));
    const int n = rbk * 16 + (l & 15);
    u32x4_t out = {0u, 0u, 0u, 0u};
    if (n < N) {
      const float sc = scale[n], inv = 1.f / sc;                       // powers of two: both exact
      bf16_t* row = W + (size_t)decode_row_perm(n, N, Dr, n_qk_heads) * K;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        bf16_t* src = row + (2 * p + h) * 32 + (l >> 4) * 8;
        const u32x4_t v = *(const u32x4_t*)src;
        unsigned q[2]; u32x4_t back;
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {                                // dword d2 of the 8-byte piece = elements 4 d2 .. 4 d2 + 3
          int pk = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * d2]) * inv, hi_bf(v[2 * d2]) * inv, 0, false);
          pk = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2 * d2 + 1]) * inv, hi_bf(v[2 * d2 + 1]) * inv, pk, true);
          q[d2] = (unsigned)pk;
          const f32x2v_t a0 = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(pk, true);
          back[2 * d2] = pack2bf(a0[0] * sc, a0[1] * sc);               // exact: 4 significant bits times a power of two
          back[2 * d2 + 1] = pack2bf(a1[0] * sc, a1[1] * sc);
        }
        out[2 * h] = q[0]; out[2 * h + 1] = q[1];
        *(u32x4_t*)src = back;
      }
    }
    *(u32x4_t*)(Wt8 + (size_t)idx * 16) = out;
  }
}
// ---- MXFP4 (OCP Microscaling Formats v1.0 [ext]: E2M1 elements {0, .5, 1, 1.5, 2, 3, 4, 6} x sign, one E8M0 scale 2^(e - 127) per block of
// 32 consecutive k of one row).  Scale: e = floor(log2(max|w|)) - 2 + 127 (emax of E2M1 is 2), clamped to [0, 254], 0 for an all-zero
// block; elements: round to nearest, ties to the even code, saturating at 6.  Written out in plain comparisons (the CPU restatement in
// the tests follows it line by line) rather than through v_cvt_scalef32_pk_fp4_f32.
// Scale words: [row block][k step / 4][16 rows] u32, byte j = the scale of k step 4 (s / 4) + j.
__device__ __forceinline__ unsigned mx_e2m1_code(float v, float inv) {
  const float t = fabsf(v) * inv;                 // exact: inv is a power of two
  const unsigned c = t <= 0.25f ? 0u : t < 0.75f ? 1u : t <= 1.25f ? 2u : t < 1.75f ? 3u : t <= 2.5f ? 4u : t < 3.5f ? 5u : t <= 5.0f ? 6u : 7u;
  return c | ((__float_as_uint(v) >> 31) << 3);
}
__device__ __forceinline__ float mx_e2m1_value(unsigned code) {
  const unsigned c = code & 7u;
  const float m = c < 2u ? 0.5f * (float)c : ldexpf(1.f + 0.5f * (float)(c & 1u), (int)(c >> 1) - 1);   // 0, .5 | 1, 1.5, 2, 3, 4, 6
  return (code & 8u) ? -m : m;
}
__global__ void mxfp4_scale_kernel(const bf16_t* __restrict__ W, unsigned char* __restrict__ scale, int N, int K, int Dr, int n_qk_heads) {
  const int nkt = K / 32;
  const long total = (long)((N + 15) / 16) * 16 * nkt;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int s = (int)(idx % nkt); const int n = (int)(idx / nkt);
    unsigned e = 0;
    if (n < N) {
      const bf16_t* src = W + (size_t)decode_row_perm(n, N, Dr, n_qk_heads) * K + s * 32;
      float m = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const u32x4_t v = *(const u32x4_t*)(src + c * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) m = fmaxf(m, fmaxf(fabsf(lo_bf(v[q])), fabsf(hi_bf(v[q]))));
      }
      const int ef = (int)((__float_as_uint(m) >> 23) & 255u) - 2;          // exponent field of max|w| minus emax(E2M1)
      e = m > 0.f ? (unsigned)(ef < 0 ? 0 : (ef > 254 ? 254 : ef)) : 0u;
    }
    scale[(((size_t)(n >> 4) * (nkt / 4) + (s >> 2)) * 16 + (n & 15)) * 4 + (s & 3)] = (unsigned char)e;
  }
}
// one thread per 16-byte piece of the tile copy [row block][k step / 4][64 lanes][16]: word j = the lane's 8 elements (k chunk l >> 4) of
// k step 4 sg + j, element t in nibble t; the source elements are overwritten with their de-quantised values
__global__ void mxfp4_requant_retile_kernel(bf16_t* __restrict__ W, unsigned char* __restrict__ Wt4, const unsigned* __restrict__ scale, int N, int K, int Dr, int n_qk_heads) {
  const int nsg = K / 128;
  const long total = (long)((N + 15) / 16) * nsg * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int l = (int)(idx & 63);
    const long tile = idx >> 6;
    const int sg = (int)(tile % nsg), rbk = (int)(tile / nsg);
    const int n = rbk * 16 + (l & 15);
    u32x4_t out = {0u, 0u, 0u, 0u};
    if (n < N) {
      const unsigned sw = scale[((size_t)rbk * nsg + sg) * 16 + (l & 15)];
      bf16_t* row = W + (size_t)decode_row_perm(n, N, Dr, n_qk_heads) * K;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned e = (sw >> (8 * j)) & 255u;
        const float sc = __uint_as_float(e << 23), inv = __uint_as_float((254u - e) << 23);     // 2^(e-127) and its reciprocal (e = 0: block of zeros)
        bf16_t* src = row + (4 * sg + j) * 32 + (l >> 4) * 8;
        const u32x4_t v = *(const u32x4_t*)src;
        unsigned word = 0; u32x4_t back;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned c0 = e ? mx_e2m1_code(lo_bf(v[q]), inv) : 0u, c1 = e ? mx_e2m1_code(hi_bf(v[q]), inv) : 0u;
          word |= (c0 | (c1 << 4)) << (8 * q);
          back[q] = pack2bf(mx_e2m1_value(c0) * sc, mx_e2m1_value(c1) * sc);
        }
        out[j] = word;
        *(u32x4_t*)src = back;
      }
    }
    *(u32x4_t*)(Wt4 + (size_t)idx * 16) = out;
  }
}
int gvl_mxfp4_quantise_decode_weight(bf16_t* W, unsigned char* Wt4, unsigned* scale, int N, int K, int Dr, int n_qk_heads, hipStream_t st) {
  if (K % 1024 || N <= 0) return -1;
  { const long total = (long)((N + 15) / 16) * 16 * (K / 32);
    long blocks = (total + 255) / 256; if (blocks > 65535 * 8) blocks = 65535 * 8;
    hipLaunchKernelGGL(mxfp4_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, W, (unsigned char*)scale, N, K, Dr, n_qk_heads); }
  { const long total = (long)((N + 15) / 16) * (K / 128) * 64;
    long blocks = (total + 255) / 256; if (blocks > 65535 * 8) blocks = 65535 * 8;
    hipLaunchKernelGGL(mxfp4_requant_retile_kernel, dim3((unsigned)blocks), dim3(256), 0, st, W, Wt4, scale, N, K, Dr, n_qk_heads); }
  return CHECK_LAUNCH();
}
int gvl_fp8_quantise_decode_weight(bf16_t* W, unsigned char* Wt8, float* scale, int N, int K, int Dr, int n_qk_heads, hipStream_t st) {
  if (K % 512 || N <= 0) return -1;
  hipLaunchKernelGGL(fp8_row_scale_kernel, dim3(N), dim3(64), 0, st, W, scale, N, K, Dr, n_qk_heads);
  const long total = (long)((N + 15) / 16) * (K / 64) * 64;
  long blocks = (total + 255) / 256; if (blocks > 65535 * 8) blocks = 65535 * 8;
  hipLaunchKernelGGL(fp8_requant_retile_kernel, dim3((unsigned)blocks), dim3(256), 0, st, W, Wt8, scale, N, K, Dr, n_qk_heads);
  return CHECK_LAUNCH();
}
