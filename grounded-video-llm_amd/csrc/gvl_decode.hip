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
// RMSNorm: the reference normalises the residual stream in front of qkv_proj / gate_up_proj / lm_head.  Instead of letting every
// block of the CONSUMER re-normalise all sequences (round 1: 1024 blocks x B rows), the PRODUCER of the residual stream does it
// once: the last block to finish an o_proj / down_proj launch (ticket counter, write-through stores, the pattern of
// decode_attn_kernel) reads the new rows and writes bf16(w * bf16(x * rstd)) for the next consumer, one wave per sequence.
#include "gvl_internal.h"

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)

typedef __attribute__((ext_vector_type(4))) float f32x4v_t;

namespace {

// RMSNorm of one bf16 row by ONE wave: xn = bf16(w * bf16(x * rsqrt(mean(x^2) + eps)))  (Phi3RMSNorm modeling_phi3.py:319-324 /
// LlamaRMSNorm: fp32 statistics, result cast to bf16, then the bf16 weight product).  `ld64(p)` loads 8 bytes (4 bf16).
// cols % 256 == 0 is not required: chunks of 4 elements, cols % 4 == 0.
template <typename Load64>
__device__ __forceinline__ void wave_rmsnorm_row(const bf16_t* x, const bf16_t* w, bf16_t* xn, bf16_t* xcopy, int cols, float eps, int lane, Load64 ld64) {
  constexpr int MAXC = 16;                       // 64 lanes x 16 chunks x 4 elements = 4096 columns held in registers
  const int nchunk = cols >> 2;
  unsigned long long v[MAXC];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = c * 64 + lane;
    v[c] = ch < nchunk ? ld64(x + (size_t)ch * 4) : 0ull;
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const unsigned lo = (unsigned)v[c], hi = (unsigned)(v[c] >> 32);
    const float p0 = lo_bf(lo), p1 = hi_bf(lo), p2 = lo_bf(hi), p3 = hi_bf(hi);
    s += p0 * p0 + p1 * p1;
    s += p2 * p2 + p3 * p3;
  }
  s = wave_sum(s);
  const float rs = rsqrtf(s / (float)cols + eps);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = c * 64 + lane;
    if (ch < nchunk) {
      const unsigned lo = (unsigned)v[c], hi = (unsigned)(v[c] >> 32);
      const u32x2_t wv = *(const u32x2_t*)(w + (size_t)ch * 4);
      u32x2_t o;
      o[0] = pack2bf(lo_bf(wv[0]) * rbf(lo_bf(lo) * rs), hi_bf(wv[0]) * rbf(hi_bf(lo) * rs));
      o[1] = pack2bf(lo_bf(wv[1]) * rbf(lo_bf(hi) * rs), hi_bf(wv[1]) * rbf(hi_bf(hi) * rs));
      *(u32x2_t*)(xn + (size_t)ch * 4) = o;
      if (xcopy) *(unsigned long long*)(xcopy + (size_t)ch * 4) = v[c];
    }
  }
}

}  // namespace

// x[b] = table[*tok[b]] (the embedding row of the sequence's latest token) and xn[b] = rmsnorm(x[b]) * w: one wave per sequence.
__global__ __launch_bounds__(64) void embed_norm_kernel(const bf16_t* __restrict__ table, const TokPtrs toks, bf16_t* __restrict__ x, bf16_t* __restrict__ xn,
                                                        const bf16_t* __restrict__ w, int cols, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int tok = *toks.p[b];
  wave_rmsnorm_row(table + (size_t)tok * cols, w, xn + (size_t)b * cols, x + (size_t)b * cols, cols, eps, lane,
                   [](const bf16_t* p) { return *(const unsigned long long*)p; });
}
int gvl_launch_embed_norm(const bf16_t* table, const TokPtrs& toks, bf16_t* x, bf16_t* xn, const bf16_t* w, int cols, float eps, hipStream_t st) {
  if (cols % 4 || cols > 4096 || toks.n < 1 || toks.n > GVL_MAX_DECODE_BATCH) return -1;
  hipLaunchKernelGGL(embed_norm_kernel, dim3(toks.n), dim3(64), 0, st, table, toks, x, xn, w, cols, eps);
  return CHECK_LAUNCH();
}

template <int RB>
__global__ __launch_bounds__(512, 4) void dgemm_kernel(const GemvArgs a) {   // 4 waves / SIMD = 2 blocks per CU (<= 128 VGPRs)
  constexpr int U = 4;                           // MFMA steps (32 k each) whose operand loads are in flight together
  __shared__ __attribute__((aligned(16))) float red[8][RB][64][4];
  __shared__ int last_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * (16 * RB);
  const int kw = a.K >> 3;                       // k range of one wave
  const int kbase = wave * kw + g * 8;
  const int halfd = a.Dr >> 1, npair_qk = (a.H + a.KV) * halfd;

  const bf16_t* wp[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int n = n0 + rb * 16 + i; if (n > a.N - 1) n = a.N - 1;
    if (a.rope_on) {                             // logical row -> weight row: pairs (2j, 2j+1) = rotate_half partners (d, d + Dr/2)
      const int j = n >> 1;
      if (j < npair_qk) { const int hd = j / halfd, d = j - hd * halfd; n = hd * a.Dr + d + (n & 1) * halfd; }
    }
    wp[rb] = a.W + (size_t)n * a.K + kbase;
  }
  // B columns >= batch re-read the LAST sequence's activations: a D column depends on its own B column only and those columns
  // are never stored, so no masking is needed in the loop (the duplicate addresses coalesce)
  const bf16_t* xp = a.x + (size_t)(i < a.batch ? i : a.batch - 1) * a.x_stride + kbase;

  f32x4v_t acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb] = f32x4v_t{0.f, 0.f, 0.f, 0.f};
  const int steps = kw >> 5;
  // groups of U steps, double buffered: while the U x RB MFMAs of group t run, the U x (RB + 1) 16-byte loads of group t + 1 are
  // in flight.  steps = K / 256 is a multiple of U = 4 for every shipped geometry (3072 -> 12, 4096 -> 16, 8192 -> 32,
  // 14336 -> 56); the remainder loop covers anything else.
  bf16x8_t wv[2][U][RB], xv[2][U];
  auto request = [&](int buf, int s) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) wv[buf][u][rb] = __builtin_nontemporal_load((const bf16x8_t*)(wp[rb] + (s + u) * 32));
      xv[buf][u] = *(const bf16x8_t*)(xp + (s + u) * 32);
    }
  };
  auto consume = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[buf][u][rb], xv[buf][u], acc[rb], 0, 0, 0);
  };
  const int groups = steps / U;
  if (groups > 0) request(0, 0);
  int gi = 0;
  for (; gi + 2 <= groups; gi += 2) {            // two groups per trip: the buffer index stays a compile-time constant
    request(1, (gi + 1) * U);
    consume(0);
    if (gi + 2 < groups) request(0, (gi + 2) * U);
    consume(1);
  }
  if (gi < groups) consume(0);
  for (int s0 = groups * U; s0 < steps; ++s0) {
    const bf16x8_t x1 = *(const bf16x8_t*)(xp + s0 * 32);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_nontemporal_load((const bf16x8_t*)(wp[rb] + s0 * 32)), x1, acc[rb], 0, 0, 0);
  }
  // partial sums of the 8 k-slices meet in LDS; wave rb (< RB) adds them in a fixed order and runs that row block's epilogue
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) *(f32x4v_t*)red[wave][rb][lane] = acc[rb];
  __syncthreads();
  if (wave < RB) {
    const int rb = wave;
    float v[4];
    {
      auto ld = [&](int w) { return *(const f32x4v_t*)red[w][rb][lane]; };
      const f32x4v_t q0 = (ld(0) + ld(1)) + (ld(2) + ld(3));      // FIXED association: the result does not depend on anything
      const f32x4v_t q1 = (ld(4) + ld(5)) + (ld(6) + ld(7));      // but the 8 partial sums themselves
      const f32x4v_t q = q0 + q1;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = q[r];
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
          if (a.out_bf16) *(unsigned*)(a.out_bf16 + (size_t)b * a.out_stride + (nr >> 1)) = pack2bf(o[0], o[1]);
          if (a.out_f32) { a.out_f32[(size_t)b * a.out_stride + (nr >> 1)] = o[0]; a.out_f32[(size_t)b * a.out_stride + (nr >> 1) + 1] = o[1]; }
        } else if (nr + 1 < a.N) {
          if (a.out_bf16) a.out_bf16[(size_t)b * a.out_stride + (nr >> 1)] = f2bf(o[0]);
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
            const unsigned long long o = (unsigned long long)pack2bf(v[0], v[1]) | ((unsigned long long)pack2bf(v[2], v[3]) << 32);
            unsigned long long* dst = (unsigned long long*)(a.out_bf16 + (size_t)b * a.out_stride + nr);
            if (a.tail_xn) __hip_atomic_store(dst, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through: the merging block reads it
            else *dst = o;
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
  if (!a.tail_xn) return;
  // ---- the LAST block normalises the new residual rows for the next consumer (uniform branch: tail_xn is a kernel argument) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(a.tail_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = (t == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!last_s) return;
  for (int b = wave; b < a.batch; b += 8)
    wave_rmsnorm_row(a.out_bf16 + (size_t)b * a.out_stride, a.tail_norm_w, a.tail_xn + (size_t)b * a.tail_stride, (bf16_t*)nullptr, a.N, a.tail_eps, lane,
                     [](const bf16_t* p) { return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); });
  if (tid == 0) __hip_atomic_store(a.tail_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
}

// Skinny-GEMM decode projection.  Preconditions (else -1: the caller falls back to the VALU kernel): K % 256 == 0, batch <= 16,
// no fused RMSNorm prologue (a.norm_w == null: the activations arrive normalised), rows 16-byte aligned.
int gvl_launch_dgemm(const GemvArgs& a_in, hipStream_t st) {
  GemvArgs a = a_in;
  if (a.batch <= 0) a.batch = 1;
  if (a.K % 256 || a.K <= 0 || a.N <= 0 || a.batch > GVL_MAX_DECODE_BATCH || a.norm_w) return -1;
  if (a.batch > 1 && (a.x_stride % 8)) return -1;
  if (((uintptr_t)a.x & 15) || ((uintptr_t)a.W & 15)) return -1;
  if (a.rope_on && ((a.Dr & 1) || (a.N & 3))) return -1;
  if (a.act == GVL_ACT_SILU_MUL && (a.N & 3)) return -1;
  if (a.tail_xn && (!a.out_bf16 || !a.tail_norm_w || !a.tail_counter || (a.N & 3) || a.N > 4096 || a.act != GVL_ACT_NONE || a.rope_on)) return -1;
  if (a.batch == 1) { a.x_stride = 0; }
  const int RB = a.N >= 8192 ? 2 : 1;
  const int blocks = (a.N + 16 * RB - 1) / (16 * RB);
  if (RB == 2) hipLaunchKernelGGL(dgemm_kernel<2>, dim3(blocks), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(dgemm_kernel<1>, dim3(blocks), dim3(512), 0, st, a);
  return CHECK_LAUNCH();
}
