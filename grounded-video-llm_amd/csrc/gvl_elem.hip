// gvl_elem.hip -- HBM-bound kernels of the hot path: norms, patch im2col, embeddings, qkv re-tiling
// (+ qk-RMSNorm / RoPE), HD-merge / pooling glue, decode GEMV, argmax.   gfx950 only.
// Every kernel moves 8/16 bytes per lane (guide G13) and keeps statistics in fp32.
#include "gvl_internal.h"
#include <cstring>

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)

// =====================================================================================================
// LayerNorm (CLIP: models/modeling_clip.py:351,353,851; fp32 in, bf16 out -- the next op is an autocast
// linear) and RMSNorm (models/internvideo2.py:443-448, models/modeling_phi3.py:319-324).
// one wave per row, 4 rows per block.
// =====================================================================================================
template <int MAXV>  // MAXV float4 per lane: cols <= MAXV*256
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, bf16_t* __restrict__ y, int rows, int cols, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * cols;
  f32x4_t v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < cols) { v[i] = *(const f32x4_t*)(xr + c); s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
  }
  const float mean = wave_sum(s) / cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < cols) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / cols + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < cols) {
      const f32x4_t wv = *(const f32x4_t*)(w + c), bv = *(const f32x4_t*)(b + c);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
      u32x2_t pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
      *(u32x2_t*)(y + (size_t)row * cols + c) = pk;
    }
  }
}

int gvl_launch_layernorm_f32(const float* x, const float* w, const float* b, bf16_t* y, int rows, int cols, float eps, hipStream_t st) {
  if (cols % 4 || cols > 4096) return -1;
  dim3 g((rows + 3) / 4), t(256);
  if (cols <= 1024) hipLaunchKernelGGL(layernorm_f32_kernel<4>, g, t, 0, st, x, w, b, y, rows, cols, eps);
  else hipLaunchKernelGGL(layernorm_f32_kernel<16>, g, t, 0, st, x, w, b, y, rows, cols, eps);
  return CHECK_LAUNCH();
}

template <int MAXV>  // MAXV 16-byte vectors per lane: cols <= MAXV*512
__global__ __launch_bounds__(256) void rmsnorm_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ y, int rows, int cols, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * cols;
  u32x4_t v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < cols) {
      v[i] = *(const u32x4_t*)(xr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float a = lo_bf(v[i][e]), bb = hi_bf(v[i][e]); s += a * a + bb * bb; }
    }
  }
  const float rs = rsqrtf(wave_sum(s) / cols + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < cols) {
      const u32x4_t wv = *(const u32x4_t*)(w + c);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e)   // weight * bf16(x * rsqrt) , both factors bf16, product rounded to bf16
        o[e] = pack2bf(lo_bf(wv[e]) * rbf(lo_bf(v[i][e]) * rs), hi_bf(wv[e]) * rbf(hi_bf(v[i][e]) * rs));
      *(u32x4_t*)(y + (size_t)row * cols + c) = o;
    }
  }
}

int gvl_launch_rmsnorm_bf16(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int cols, float eps, hipStream_t st) {
  if (cols % 8 || cols > 8192) return -1;
  dim3 g((rows + 3) / 4), t(256);
  if (cols <= 2048) hipLaunchKernelGGL(rmsnorm_bf16_kernel<4>, g, t, 0, st, x, w, y, rows, cols, eps);
  else hipLaunchKernelGGL(rmsnorm_bf16_kernel<16>, g, t, 0, st, x, w, y, rows, cols, eps);
  return CHECK_LAUNCH();
}

// =====================================================================================================
// device -> device row copies of the step (packed prefill rows, the visual block of a splice, last-row gathers) as a kernel of this library on the
// caller's stream: the step's trace then holds no runtime blit kernels, and the copy is an ordinary node under stream capture
// =====================================================================================================
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
int gvl_launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t st) {
  if (bytes == 0) return 0;
  if ((bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : -3;   // odd sizes: the runtime's copy
  const long n16 = (long)(bytes >> 4);
  int blocks = (int)((n16 + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(256), 0, st, (const u32x4_t*)src, (u32x4_t*)dst, n16);
  return CHECK_LAUNCH();
}

// =====================================================================================================
// Fused RMSNorm (round 5), the two small kernels around the GEMM epilogues (GemmArgs.rowscale / rowsq):
//   fold_gamma:    W'[n][k] = bf16(W[n][k] * gamma[k])   once, at gvl_finalize_weights -- the norm weight rides in the projection that consumes the norm
//   rowsq_finish:  rs[m] = rsqrt((sum_b rowsq[m][b]) / cols + eps), blocks added in index order (fixed): M x nblk floats in, M floats out
// RMSNorm(x) . W^T == rs[m] * (x . W'^T) up to where the roundings sit: the reference rounds x * rs and the gamma product to bf16 before the GEMM
// (internvideo2.py:443-448, modeling_phi3.py:319-324); here those two roundings are gone and gamma * W is rounded once per weight instead -- a different,
// not larger, set of rounding points (tests/test_gpu_ops.py bounds the difference against the unfused pair; the tower / end-to-end goldens hold both).
// =====================================================================================================
__global__ __launch_bounds__(256) void fold_gamma_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ g, bf16_t* __restrict__ Wo, long rows, int cols) {
  const long total = rows * (cols / 8);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % (cols / 8)) * 8;
    const long r = idx / (cols / 8);
    const u32x4_t w = *(const u32x4_t*)(W + r * cols + c), gv = *(const u32x4_t*)(g + c);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(w[e]) * lo_bf(gv[e]), hi_bf(w[e]) * hi_bf(gv[e]));
    *(u32x4_t*)(Wo + r * cols + c) = o;
  }
}
int gvl_launch_fold_gamma(const bf16_t* W, const bf16_t* gamma, bf16_t* Wo, long rows, int cols, hipStream_t st) {
  if (cols % 8 || rows <= 0) return -1;
  const long total = rows * (cols / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(fold_gamma_kernel, dim3(blocks), dim3(256), 0, st, W, gamma, Wo, rows, cols);
  return CHECK_LAUNCH();
}
__global__ __launch_bounds__(256) void rowsq_finish_kernel(const float* __restrict__ sq, int ld, int b0, int nblk, float* __restrict__ rs, int rows, float inv_cols, float eps) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= rows) return;
  const float* p = sq + (size_t)m * ld + b0;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += p[b];
  rs[m] = rsqrtf(s * inv_cols + eps);
}
// blocks [b0, b0 + nblk) of every row (b0 > 0: a column range of a fused output, e.g. the k part of a qkv row)
int gvl_launch_rowsq_finish(const float* sq, int ld, int b0, int nblk, float* rs, int rows, int cols, float eps, hipStream_t st) {
  if (rows <= 0 || nblk <= 0 || b0 < 0 || b0 + nblk > ld || cols <= 0) return -1;
  hipLaunchKernelGGL(rowsq_finish_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, sq, ld, b0, nblk, rs, rows, 1.0f / (float)cols, eps);
  return CHECK_LAUNCH();
}

// =====================================================================================================
// patch im2col for conv with stride == kernel (models/modeling_clip.py:185, models/internvideo2.py:714-722)
// px f32 [n][3][T][HW][HW] -> A bf16 [n*T*g*g][Kp], k = c*p*p + py*p + px, zero padded to Kp
// =====================================================================================================
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ px, bf16_t* __restrict__ A, int n_img, int T, int image,
                                                       int patch, int Kp) {
  const int g = image / patch, K = 3 * patch * patch, kv = Kp / 8;
  const long total = (long)n_img * T * g * g * kv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int kc = (int)(idx % kv);
    const long row = idx / kv;
    const int gx = (int)(row % g), gy = (int)((row / g) % g), t = (int)((row / ((long)g * g)) % T), img = (int)(row / ((long)g * g * T));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kc * 8 + e;
      if (k < K) {
        const int c = k / (patch * patch), rem = k - c * patch * patch, py = rem / patch, pxx = rem - py * patch;
        v[e] = px[((((size_t)img * 3 + c) * T + t) * image + (gy * patch + py)) * image + gx * patch + pxx];
      } else v[e] = 0.f;
    }
    u32x4_t o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    *(u32x4_t*)(A + row * Kp + kc * 8) = o;
  }
}
int gvl_launch_patchify(const float* px, bf16_t* A, int n_img, int T, int image, int patch, int Kp, hipStream_t st) {
  if (Kp % 8 || Kp < 3 * patch * patch || image % patch) return -1;
  const int g = image / patch;
  const long total = (long)n_img * T * g * g * (Kp / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, st, px, A, n_img, T, image, patch, Kp);
  return CHECK_LAUNCH();
}

// =====================================================================================================
// CLIP embeddings + pre_layrnorm (models/modeling_clip.py:182-191, :851): one wave per token row
// =====================================================================================================
template <int MAXV>
__global__ __launch_bounds__(256) void clip_embed_ln_kernel(const bf16_t* __restrict__ patch, const float* __restrict__ cls,
                                                            const float* __restrict__ pos, const float* __restrict__ lnw,
                                                            const float* __restrict__ lnb, float* __restrict__ x, int n_img, int P, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int S = P + 1;
  if (row >= n_img * S) return;
  const int img = row / S, s = row - img * S;
  f32x4_t v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
      const f32x4_t pv = *(const f32x4_t*)(pos + (size_t)s * C + c);
      if (s == 0) {
        const f32x4_t cv = *(const f32x4_t*)(cls + c);
        v[i] = cv + pv;
      } else {
        const u32x2_t pk = *(const u32x2_t*)(patch + ((size_t)img * P + (s - 1)) * C + c);
        f32x4_t e = {lo_bf(pk[0]), hi_bf(pk[0]), lo_bf(pk[1]), hi_bf(pk[1])};
        v[i] = e + pv;
      }
      sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(sum) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < C) {
      const f32x4_t wv = *(const f32x4_t*)(lnw + c), bv = *(const f32x4_t*)(lnb + c);
      f32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * wv[e] + bv[e];
      *(f32x4_t*)(x + (size_t)row * C + c) = o;
    }
  }
}
int gvl_launch_clip_embed_ln(const bf16_t* patch, const float* cls, const float* pos, const float* lnw, const float* lnb, float* x,
                             int n_img, int P, int C, float eps, hipStream_t st) {
  if (C % 4 || C > 4096) return -1;
  const int rows = n_img * (P + 1);
  dim3 g((rows + 3) / 4), t(256);
  if (C <= 1024) hipLaunchKernelGGL(clip_embed_ln_kernel<4>, g, t, 0, st, patch, cls, pos, lnw, lnb, x, n_img, P, C, eps);
  else hipLaunchKernelGGL(clip_embed_ln_kernel<16>, g, t, 0, st, patch, cls, pos, lnw, lnb, x, n_img, P, C, eps);
  return CHECK_LAUNCH();
}

// =====================================================================================================
// InternVideo2 embeddings (models/internvideo2.py:972-1011): cat(cls, patches) + pos_embed, all bf16
// =====================================================================================================
__global__ __launch_bounds__(256) void iv2_embed_kernel(const bf16_t* __restrict__ patch, const bf16_t* __restrict__ cls,
                                                        const bf16_t* __restrict__ pos, bf16_t* __restrict__ x, int B, int TL, int C) {
  const int cv = C / 8;
  const long total = (long)B * (TL + 1) * cv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * 8;
    const long row = idx / cv;
    const int s = (int)(row % (TL + 1)), b = (int)(row / (TL + 1));
    const u32x4_t pv = *(const u32x4_t*)(pos + (size_t)s * C + c);
    const u32x4_t sv = s == 0 ? *(const u32x4_t*)(cls + c) : *(const u32x4_t*)(patch + ((size_t)b * TL + (s - 1)) * C + c);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(sv[e]) + lo_bf(pv[e]), hi_bf(sv[e]) + hi_bf(pv[e]));
    *(u32x4_t*)(x + row * C + c) = o;
  }
}
int gvl_launch_iv2_embed(const bf16_t* patch, const bf16_t* cls, const bf16_t* pos, bf16_t* x, int B, int TL, int C, hipStream_t st) {
  if (C % 8) return -1;
  const long total = (long)B * (TL + 1) * (C / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(iv2_embed_kernel, dim3(blocks), dim3(256), 0, st, patch, cls, pos, x, B, TL, C);
  return CHECK_LAUNCH();
}

// =====================================================================================================
// qkv_post: fused-qkv row -> Q [B][H][S][D], K pages [page][KV][64][D] (+ V^T column for the decode row)
//   mode 1: IV2 q/k RMSNorm over the full width (models/internvideo2.py:560-561,590-598)
//   mode 2: RoPE (models/modeling_phi3.py:413-445; cos/sin tables hold the bf16-rounded values of :397-409)
// one block per token row.
// =====================================================================================================
template <int WPR>   // waves per token row: 1 -> 4 rows per block (bulk), 4 -> one row per block (decode)
__global__ __launch_bounds__(256) void qkv_post_kernel(const QkvPostArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = WPR == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
  if (row >= (a.vl_n ? a.vl_rows[a.vl_n] : a.B * a.S)) return;                     // no block-level barrier below
  constexpr int TEAM = 64 * WPR;
  const int lt = WPR == 1 ? lane : tid;
  // ragged group (vl_n > 0): the row's sequence u by <= 8 prefix sums; after that it is that sequence's own single-sequence launch (b = 0, its table, its length)
  int S_ = a.S, b = 0, s_in = 0, row0 = 0;
  const int* table = a.block_table;
  if (a.vl_n) {
    int u = 0;
#pragma unroll
    for (int i = 1; i < GVL_MAX_PREFILL_BATCH; ++i) u += (i < a.vl_n && row >= a.vl_rows[i]) ? 1 : 0;
    row0 = a.vl_rows[u]; S_ = a.vl_rows[u + 1] - row0; s_in = row - row0; table = a.vl_tables[u];
  } else { b = row / a.S; s_in = row - b * a.S; }
  const bf16_t* qr = a.qkv + (size_t)row * a.ld;
  const bf16_t* kr = qr + a.H * a.Dr;
  const bf16_t* vr = kr + a.KV * a.Dr;
  const int half = a.Dr >> 1, cpr = a.Dr >> 3, cpd = a.D >> 3, hc = half >> 3;
  const int nq = a.H * cpr, nk = a.KV * cpr;

  float q_rs = 1.f, k_rs = 1.f;
  // mode 1 (WPR == 1: one wave owns the row): the row's q and k chunks (<= 3 per lane each, i.e. rows up to 1536 wide) are
  // requested up front and stay in registers between the full-width RMS statistics and the transform -- one pass over the row
  constexpr int RC = 3;
  u32x4_t qreg[RC], kreg[RC];
  const bool in_regs = WPR == 1 && a.mode == 1 && nq <= RC * 64 && nk <= RC * 64;
  if (a.mode == 1) {
    float sq = 0.f, sk = 0.f;
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < RC; ++i) { const int c = lane + 64 * i; qreg[i] = c < nq ? *(const u32x4_t*)(qr + c * 8) : u32x4_t{0u, 0u, 0u, 0u}; }
#pragma unroll
      for (int i = 0; i < RC; ++i) { const int c = lane + 64 * i; kreg[i] = c < nk ? *(const u32x4_t*)(kr + c * 8) : u32x4_t{0u, 0u, 0u, 0u}; }
#pragma unroll
      for (int i = 0; i < RC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = lo_bf(qreg[i][e]), y = hi_bf(qreg[i][e]); sq += x * x + y * y;
          const float u = lo_bf(kreg[i][e]), w = hi_bf(kreg[i][e]); sk += u * u + w * w;
        }
    } else {
      for (int c = lane; c < nq; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(qr + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float x = lo_bf(v[e]), y = hi_bf(v[e]); sq += x * x + y * y; }
      }
      for (int c = lane; c < nk; c += 64) {
        const u32x4_t v = *(const u32x4_t*)(kr + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float x = lo_bf(v[e]), y = hi_bf(v[e]); sk += x * x + y * y; }
      }
    }
    q_rs = rsqrtf(wave_sum(sq) / (a.H * a.Dr) + a.eps);
    k_rs = rsqrtf(wave_sum(sk) / (a.KV * a.Dr) + a.eps);
    if (a.q_rs && lt == 0) a.q_rs[row] = q_rs;
  }
  int pos = a.pos0 + s_in;                          // token position (RoPE, cache slot)
  if (a.pos_ptr) pos = *a.pos_ptr;
  const float* cosp = a.cos; const float* sinp = a.sin;
  if (a.mode == 2 && a.rope_switch > 0 && pos + 1 > a.rope_switch) { cosp = a.cos_l; sinp = a.sin_l; }
  const int n_tiles = (S_ + 63) >> 6;
  const int tile = pos >> 6, slot = pos & 63;
  const int page = table ? table[b * a.max_pages + tile] : b * n_tiles + tile;

  // one 16-byte chunk (8 elements) of q or k -> transformed chunk
  auto xform = [&](const bf16_t* src, int c, float rs, const bf16_t* nw) -> u32x4_t {
    const u32x4_t v = *(const u32x4_t*)(src + c * 8);
    if (a.mode == 0) return v;
    u32x4_t o;
    if (a.mode == 1) {
      const u32x4_t w = *(const u32x4_t*)(nw + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(w[e]) * rbf(lo_bf(v[e]) * rs), hi_bf(w[e]) * rbf(hi_bf(v[e]) * rs));
      return o;
    }
    const int dc = c % cpr;                          // chunk inside the head
    const bool first = dc < hc;
    const u32x4_t p = *(const u32x4_t*)(src + (first ? c + hc : c - hc) * 8);   // rotate_half partner
    const int j0 = (first ? dc : dc - hc) * 8;
    const float* cp = cosp + (size_t)pos * half + j0;
    const float* sp = sinp + (size_t)pos * half + j0;
    const f32x4_t c0 = *(const f32x4_t*)cp, c1 = *(const f32x4_t*)(cp + 4), s0 = *(const f32x4_t*)sp, s1 = *(const f32x4_t*)(sp + 4);
    const float sg = first ? -1.f : 1.f;
    const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    const float ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x0 = lo_bf(v[e]), x1 = hi_bf(v[e]), p0 = sg * lo_bf(p[e]), p1 = sg * hi_bf(p[e]);
      o[e] = pack2bf(rbf(x0 * cc[2 * e]) + rbf(p0 * ss[2 * e]), rbf(x1 * cc[2 * e + 1]) + rbf(p1 * ss[2 * e + 1]));
    }
    return o;
  };
  const u32x4_t zero = {0u, 0u, 0u, 0u};
  auto norm_chunk = [&](const u32x4_t& v, int c, float rs, const bf16_t* nw) -> u32x4_t {   // mode 1 on an already loaded chunk
    const u32x4_t w = *(const u32x4_t*)(nw + c * 8);
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(w[e]) * rbf(lo_bf(v[e]) * rs), hi_bf(w[e]) * rbf(hi_bf(v[e]) * rs));
    return o;
  };
  // Q [B][H][S][D]
  bf16_t* Qb = a.Q + ((size_t)b * a.H * S_ + s_in) * a.D + (size_t)row0 * a.H * a.D;
  // K page row [page][KV][64][D]
  bf16_t* Kb = a.Kt + ((size_t)page * a.KV) * (64 * a.D) + (size_t)slot * a.D;
  if (in_regs) {
#pragma unroll
    for (int i = 0; i < RC; ++i) {
      const int c = lane + 64 * i;
      if (c < nq && !a.q_rs) { const int hd = c / cpr, dc = c - hd * cpr; *(u32x4_t*)(Qb + (size_t)hd * S_ * a.D + dc * 8) = norm_chunk(qreg[i], c, q_rs, a.qn); }
      if (c < nk) { const int hd = c / cpr, dc = c - hd * cpr; *(u32x4_t*)(Kb + (size_t)hd * (64 * a.D) + dc * 8) = norm_chunk(kreg[i], c, k_rs, a.kn); }
    }
  } else {
    if (!a.q_rs) for (int c = lt; c < nq; c += TEAM) {
      const int hd = c / cpr, dc = c - hd * cpr;
      *(u32x4_t*)(Qb + (size_t)hd * S_ * a.D + dc * 8) = xform(qr, c, q_rs, a.qn);
    }
    for (int c = lt; c < nk; c += TEAM) {
      const int hd = c / cpr, dc = c - hd * cpr;
      *(u32x4_t*)(Kb + (size_t)hd * (64 * a.D) + dc * 8) = xform(kr, c, k_rs, a.kn);
    }
  }
  if (cpd > cpr) {                                   // zero the head-dim padding (88 -> 96)
    const int np = cpd - cpr;
    if (!a.q_rs) for (int c = lt; c < a.H * np; c += TEAM) { const int hd = c / np; *(u32x4_t*)(Qb + (size_t)hd * S_ * a.D + (cpr + c % np) * 8) = zero; }
    for (int c = lt; c < a.KV * np; c += TEAM) {
      const int hd = c / np;
      u32x4_t v = zero; if (a.k_ones && c % np == 0) v[0] = 0x3F80u;      // bf16 1.0 in column Dr
      *(u32x4_t*)(Kb + (size_t)hd * (64 * a.D) + (cpr + c % np) * 8) = v;
    }
  }
  // V^T column of this single row (decode).  Bulk rows go through v_transpose_kernel.
  if (a.pos_ptr) {
    bf16_t* Vbase = a.Vt + ((size_t)page * a.KV) * (64 * a.D) + slot;
    for (int i = lt; i < a.KV * a.Dr; i += TEAM) {
      const int hd = i / a.Dr, d = i - hd * a.Dr;
      Vbase[(size_t)hd * (64 * a.D) + d * 64] = vr[i];
    }
  }
}

// V rows of one 64-token tile -> V^T page [KV][D][64] through LDS (tokens >= S and d >= Dr are zero filled).
// 16-byte global accesses both ways: 8 head-dim elements of a token in, 8 consecutive tokens of one head-dim row out.
__global__ __launch_bounds__(256) void v_transpose_kernel(const QkvPostArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][136];          // 272-byte rows (16-byte aligned, bank-staggered)
  int t = blockIdx.x, S_ = a.S, row0 = 0;
  const int hkv = blockIdx.y, b = blockIdx.z;
  const int* table = a.block_table;
  if (a.vl_n) {                                         // ragged group: blockIdx.x runs over the 64-token tiles of all sequences
    int u = 0, t0 = 0;
    for (; u + 1 < a.vl_n; ++u) { const int nt = (a.vl_rows[u + 1] - a.vl_rows[u] + 63) >> 6; if (t < t0 + nt) break; t0 += nt; }
    t -= t0; row0 = a.vl_rows[u]; S_ = a.vl_rows[u + 1] - row0; table = a.vl_tables[u];
  }
  const int tid = threadIdx.x;
  const int s0 = a.pos0 + t * 64;                       // first token (sequence space) of this tile; pos0 is a multiple of 64
  const int n_tiles = (S_ + 63) >> 6;
  const int tile_idx = s0 >> 6;
  const int page = table ? table[b * a.max_pages + tile_idx] : b * n_tiles + tile_idx;
  const int voff = (a.H + a.KV) * a.Dr + hkv * a.Dr;
  const int cpr = a.Dr >> 3;
  for (int i = tid; i < 64 * cpr; i += 256) {
    const int r = i / cpr, c = i - r * cpr;
    const int s_in = t * 64 + r;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (s_in < S_) v = *(const u32x4_t*)(a.qkv + ((size_t)b * S_ + row0 + s_in) * a.ld + voff + c * 8);
    *(u32x4_t*)&tile[r][c * 8] = v;
  }
  __syncthreads();
  bf16_t* dst = a.Vt + ((size_t)page * a.KV + hkv) * (64 * a.D);
  for (int i = tid; i < a.D * 8; i += 256) {
    const int d = i >> 3, k8 = (i & 7) * 8;
    u32x4_t o = {0u, 0u, 0u, 0u};
    if (d < a.Dr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (unsigned)tile[k8 + 2 * e][d] | ((unsigned)tile[k8 + 2 * e + 1][d] << 16);
    } else if (a.ones_row && d == a.Dr) {
      o = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};     // bf16 1.0: O^T[Dr] = sum_k P[k] (masked keys carry P = 0)
    }
    *(u32x4_t*)(dst + d * 64 + k8) = o;
  }
}

int gvl_launch_qkv_post(const QkvPostArgs& a, hipStream_t st) {
  if ((a.Dr & 7) || (a.mode == 2 && (a.Dr & 15)) || (a.D & 7) || a.D < a.Dr || a.Dr > 128 || (a.ld & 7)) return -1;   // 16-byte chunks; rotate_half partner chunk-aligned
  if (a.vl_n && (a.mode != 2 || a.B != 1 || a.pos0 != 0 || a.pos_ptr || a.vl_n > GVL_MAX_PREFILL_BATCH)) return -1;
  const int rows = a.vl_n ? a.vl_rows[a.vl_n] : a.B * a.S;
  if (a.q_rs && (a.mode != 1 || a.pos_ptr)) return -1;
  if (a.pos_ptr) {
    if (rows != 1 || a.mode != 2) return -1;
    hipLaunchKernelGGL(qkv_post_kernel<4>, dim3(1), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(qkv_post_kernel<1>, dim3((rows + 3) / 4), dim3(256), 0, st, a);
  }
  if (!a.pos_ptr && a.Vt) {                          // Vt == null: the attention kernel reads V in place (AttnArgs.Vrows)
    int n_tiles = (a.S + 63) >> 6;
    if (a.vl_n) { n_tiles = 0; for (int u = 0; u < a.vl_n; ++u) n_tiles += (a.vl_rows[u + 1] - a.vl_rows[u] + 63) >> 6; }
    hipLaunchKernelGGL(v_transpose_kernel, dim3(n_tiles, a.KV, a.B), dim3(256), 0, st, a);
  }
  return CHECK_LAUNCH();
}

// =====================================================================================================
// glue: HD 2x2 merge + sub_GN (models/llava_next_video.py:454-489), Llama 3x3 pool (:509-517),
// temporal 4x4 pool (:543-549)
// =====================================================================================================
__global__ __launch_bounds__(256) void hd_merge_kernel(const float* __restrict__ f, const float* __restrict__ sub_gn, bf16_t* __restrict__ out, int n, int C) {
  const int c4 = 4 * C, cv = c4 / 4;
  const long total = (long)n * 156 * cv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * 4;
    const long row = idx / cv;
    const int r = (int)(row % 156), img = (int)(row / 156);
    const int hh = r / 13, ww = r - hh * 13;
    f32x4_t v;
    if (ww == 12) v = *(const f32x4_t*)(sub_gn + c);
    else {
      const int chunk = c / C, cc = c - chunk * C, dy = chunk >> 1, dx = chunk & 1;
      v = *(const f32x4_t*)(f + ((size_t)img * 576 + (2 * hh + dy) * 24 + 2 * ww + dx) * C + cc);
    }
    u32x2_t o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(u32x2_t*)(out + row * c4 + c) = o;
  }
}
int gvl_launch_hd_merge(const float* f, const float* sub_gn, bf16_t* out, int n, int C, hipStream_t st) {
  if (C % 4) return -1;
  const long total = (long)n * 156 * C;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(hd_merge_kernel, dim3(blocks), dim3(256), 0, st, f, sub_gn, out, n, C);
  return CHECK_LAUNCH();
}

__global__ __launch_bounds__(256) void pool_spatial_kernel(const float* __restrict__ f, bf16_t* __restrict__ out, int n, int C) {
  const int cv = C / 4;
  const long total = (long)n * 64 * cv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * 4;
    const long row = idx / cv;
    const int r = (int)(row % 64), img = (int)(row / 64);
    const int ph = r >> 3, pw = r & 7;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx) acc += *(const f32x4_t*)(f + ((size_t)img * 576 + (3 * ph + dy) * 24 + 3 * pw + dx) * C + c);
    u32x2_t o = {pack2bf(acc[0] / 9.f, acc[1] / 9.f), pack2bf(acc[2] / 9.f, acc[3] / 9.f)};
    *(u32x2_t*)(out + row * C + c) = o;
  }
}
int gvl_launch_pool_spatial(const float* f, bf16_t* out, int n, int C, hipStream_t st) {
  if (C % 4) return -1;
  const long total = (long)n * 64 * (C / 4);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pool_spatial_kernel, dim3(blocks), dim3(256), 0, st, f, out, n, C);
  return CHECK_LAUNCH();
}

__global__ __launch_bounds__(256) void pool_temporal_kernel(const bf16_t* __restrict__ f, bf16_t* __restrict__ out, int n, int T, int C) {
  const int cv = C / 8;
  const long total = (long)n * T * 16 * cv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * 8;
    const long row = idx / cv;
    const int r = (int)(row % 16), t = (int)((row / 16) % T), seg = (int)(row / (16L * T));
    const int ph = r >> 2, pw = r & 3;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int dy = 0; dy < 4; ++dy)
      for (int dx = 0; dx < 4; ++dx) {
        const u32x4_t v = *(const u32x4_t*)(f + ((size_t)seg * T * 256 + t * 256 + (4 * ph + dy) * 16 + 4 * pw + dx) * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += lo_bf(v[e]); acc[2 * e + 1] += hi_bf(v[e]); }
      }
    u32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e] * 0.0625f, acc[2 * e + 1] * 0.0625f);
    *(u32x4_t*)(out + row * C + c) = o;
  }
}
int gvl_launch_pool_temporal(const bf16_t* f, bf16_t* out, int n, int T, int C, hipStream_t st) {
  if (C % 8) return -1;
  const long total = (long)n * T * 16 * (C / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pool_temporal_kernel, dim3(blocks), dim3(256), 0, st, f, out, n, T, C);
  return CHECK_LAUNCH();
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
int gvl_launch_f32_to_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t st) {
  int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, st, x, y, (long)n);
  return CHECK_LAUNCH();
}

__global__ void bcast_row_kernel(const bf16_t* __restrict__ row, bf16_t* __restrict__ dst, int n, int stride_rows, int row_off, int cols) {
  const long total = (long)n * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols), s = (int)(i / cols);
    dst[((size_t)s * stride_rows + row_off) * cols + c] = row[c];
  }
}
int gvl_launch_bcast_row(const bf16_t* row, bf16_t* dst, int n, int stride_rows, int row_off, int cols, hipStream_t st) {
  const long total = (long)n * cols;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bcast_row_kernel, dim3(blocks), dim3(256), 0, st, row, dst, n, stride_rows, row_off, cols);
  return CHECK_LAUNCH();
}

__global__ void gather_rows_kernel(const bf16_t* __restrict__ table, const int* __restrict__ ids, bf16_t* __restrict__ dst, int n, int cols) {
  const int cv = cols / 8;
  const long total = (long)n * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 8, r = (int)(i / cv);
    *(u32x4_t*)(dst + (size_t)r * cols + c) = *(const u32x4_t*)(table + (size_t)ids[r] * cols + c);
  }
}
int gvl_launch_gather_rows(const bf16_t* table, const int* ids, bf16_t* dst, int n, int cols, hipStream_t st) {
  if (cols % 8) return -1;
  if (n <= 0) return 0;
  const long total = (long)n * (cols / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, st, table, ids, dst, n, cols);
  return CHECK_LAUNCH();
}

// y[n][s][:] = x[n][s+1][:]   (drop the CLS row: hidden_states[-2][:, 1:], x_vis[:, 1:, :])
__global__ void strip_cls_kernel(const uint32_t* __restrict__ x, uint32_t* __restrict__ y, int n, int S, int Cw) {
  const long total = (long)n * (S - 1) * Cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cw);
    const long row = i / Cw;
    const int s = (int)(row % (S - 1)), b = (int)(row / (S - 1));
    y[i] = x[((size_t)b * S + s + 1) * Cw + c];
  }
}
int gvl_launch_strip_cls(const void* x, void* y, int n, int S, int C, int elem_bytes, hipStream_t st) {
  if ((C * elem_bytes) % 4) return -1;
  const int Cw = C * elem_bytes / 4;
  const long total = (long)n * (S - 1) * Cw;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(strip_cls_kernel, dim3(blocks), dim3(256), 0, st, (const uint32_t*)x, (uint32_t*)y, n, S, Cw);
  return CHECK_LAUNCH();
}

// =====================================================================================================
// decode GEMV: y[N] = W[N,K] x[K]; HBM-bound weight stream (SURVEY.md K16/K20/K21 at q_len = 1).
// Weights go straight to VGPRs with 16-byte loads (no LDS round trip: each byte is used once), R rows per
// wave in flight; x (optionally RMS-normalised on the fly) lives in LDS as bf16.
// =====================================================================================================
// B = sequences decoded together: every weight chunk is loaded ONCE and multiplied into B activation
// vectors, so the per-sequence HBM cost of a decode step falls as 1/B.  Per (row, sequence) the accumulation order is the
// one of the B = 1 kernel: a batched decode produces bit-identical logits to B separate decodes.
template <int R, int B>
__global__ __launch_bounds__(256) void gemv_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;                 // [B][K]
  __shared__ float red[B][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K8 = a.K >> 3;
  const int n0 = (blockIdx.x * 4 + wave) * R;
  const bf16_t* wp[R];
  const int halfd = a.Dr >> 1, npair_qk = (a.H + a.KV) * halfd;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int n = n0 + r; if (n > a.N - 1) n = a.N - 1;
    if (a.rope_on) {                        // logical row -> weight row: pairs (2j, 2j+1) = rotate_half partners
      const int j = n >> 1;
      if (j < npair_qk) { const int hd = j / halfd, d = j - hd * halfd; n = hd * a.Dr + d + (n & 1) * halfd; }
    }
    wp[r] = a.W + (size_t)n * a.K;
  }
  // the first weight chunks do not depend on x: request them before the (latency-bound) x / RMSNorm prologue
  u32x4_t w0[R];
#pragma unroll
  for (int r = 0; r < R; ++r) w0[r] = __builtin_nontemporal_load((const u32x4_t*)(wp[r] + (lane < K8 ? lane : 0) * 8));

  if (a.norm_w) {
    float s[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      s[b] = 0.f;
      for (int c = tid; c < K8; c += 256) {
        const u32x4_t v = *(const u32x4_t*)(a.x + (size_t)b * a.x_stride + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float p = lo_bf(v[e]), q = hi_bf(v[e]); s[b] += p * p + q * q; }
      }
      s[b] = wave_sum(s[b]);
      if (lane == 0) red[b][wave] = s[b];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const float rs = rsqrtf((red[b][0] + red[b][1] + red[b][2] + red[b][3]) / a.K + a.eps);
      for (int c = tid; c < K8; c += 256) {
        const u32x4_t v = *(const u32x4_t*)(a.x + (size_t)b * a.x_stride + c * 8);
        const u32x4_t w = *(const u32x4_t*)(a.norm_w + c * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(w[e]) * rbf(lo_bf(v[e]) * rs), hi_bf(w[e]) * rbf(hi_bf(v[e]) * rs));
        *(u32x4_t*)(xs + (size_t)b * a.K + c * 8) = o;
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < B; ++b)
      for (int c = tid; c < K8; c += 256) *(u32x4_t*)(xs + (size_t)b * a.K + c * 8) = *(const u32x4_t*)(a.x + (size_t)b * a.x_stride + c * 8);
  }
  __syncthreads();
  if (n0 >= a.N) return;

  float acc[R][B];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < B; ++b) acc[r][b] = 0.f;
  if (lane < K8) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const u32x4_t xv = *(const u32x4_t*)(xs + (size_t)b * a.K + lane * 8);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][b] += lo_bf(w0[r][e]) * lo_bf(xv[e]) + hi_bf(w0[r][e]) * hi_bf(xv[e]);
    }
  }
#pragma unroll (R * B >= 6 ? 2 : 4)
  for (int c = lane + 64; c < K8; c += 64) {
    u32x4_t wv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wv[r] = __builtin_nontemporal_load((const u32x4_t*)(wp[r] + c * 8));
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const u32x4_t xv = *(const u32x4_t*)(xs + (size_t)b * a.K + c * 8);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][b] += lo_bf(wv[r][e]) * lo_bf(xv[e]) + hi_bf(wv[r][e]) * hi_bf(xv[e]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < B; ++b) acc[r][b] = wave_sum(acc[r][b]);
  if (lane != 0) return;
#pragma unroll
  for (int b = 0; b < B; ++b) {
    if (a.rope_on) {
      if constexpr ((R & 1) == 0) {
        const int pos = *a.pos_ptrs[b];
        const float* cosp = a.cos_s; const float* sinp = a.sin_s;
        if (a.rope_switch > 0 && pos + 1 > a.rope_switch) { cosp = a.cos_l; sinp = a.sin_l; }
        const int page = a.tables[b][pos >> 6], slot = pos & 63;
        bf16_t* Qb = a.Q + (size_t)b * a.q_stride;
#pragma unroll
        for (int r = 0; r < R; r += 2) {
          const int i = n0 + r;
          if (i + 1 < a.N) {
            const int j = i >> 1;
            if (j < npair_qk) {
              const int hd = j / halfd, d = j - hd * halfd;
              const float x1 = rbf(acc[r][b]), x2 = rbf(acc[r + 1][b]);
              const float c = cosp[(size_t)pos * halfd + d], sn = sinp[(size_t)pos * halfd + d];
              const bf16_t o1 = f2bf(rbf(x1 * c) + rbf(-x2 * sn)), o2 = f2bf(rbf(x2 * c) + rbf(x1 * sn));
              bf16_t* dst = hd < a.H ? Qb + (size_t)hd * a.D + d
                                     : a.Kt + (((size_t)page * a.KV + (hd - a.H)) * 64 + slot) * a.D + d;
              dst[0] = o1; dst[halfd] = o2;
            } else {
              const int vi = i - 2 * npair_qk, hv = vi / a.Dr, d = vi - hv * a.Dr;
              bf16_t* dst = a.Vt + (((size_t)page * a.KV + hv) * a.D + d) * 64 + slot;
              dst[0] = f2bf(acc[r][b]); dst[64] = f2bf(acc[r + 1][b]);
            }
          }
        }
      }
    } else if (a.act == GVL_ACT_SILU_MUL) {
      if constexpr ((R & 1) == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
          const int n = n0 + r;
          if (n + 1 < a.N) {
            const float g = rbf(acc[r][b]), u = rbf(acc[r + 1][b]);
            const float o = u * rbf(g * fast_sigmoid(g));
            if (a.out_bf16) a.out_bf16[(size_t)b * a.out_stride + (n >> 1)] = f2bf(o);
            if (a.out_f32) a.out_f32[(size_t)b * a.out_stride + (n >> 1)] = o;
          }
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = n0 + r;
        if (n < a.N) {
          float v = acc[r][b];
          if (a.bias) v += a.bias[n];
          if (a.resid) v = bf2f(a.resid[(size_t)b * a.out_stride + n]) + rbf(v);
          if (a.out_bf16) a.out_bf16[(size_t)b * a.out_stride + n] = f2bf(v);
          if (a.out_f32) a.out_f32[(size_t)b * a.out_stride + n] = v;
        }
      }
    }
  }
}

template <int B>
static int launch_gemv_b(const GemvArgs& a, int R, int blocks, size_t lds, hipStream_t st) {
  switch (R) {
#define GEMV_CASE(RR) case RR: { auto k = gemv_kernel<RR, B>; \
      if (lds > 48 * 1024) { static GvlDevOnce once_; if (gvl_set_max_lds(once_, (const void*)k, 160 * 1024 - 1024)) return -3; } \
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, a); break; }
    GEMV_CASE(1) GEMV_CASE(2) GEMV_CASE(3) GEMV_CASE(4) GEMV_CASE(6)
    default: { auto k = gemv_kernel<8, B>;
      if (lds > 48 * 1024) { static GvlDevOnce once_; if (gvl_set_max_lds(once_, (const void*)k, 160 * 1024 - 1024)) return -3; }
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, a); break; }
#undef GEMV_CASE
  }
  return CHECK_LAUNCH();
}

int gvl_launch_gemv(const GemvArgs& a_in, hipStream_t st) {
  GemvArgs a = a_in;
  if (a.batch <= 0) a.batch = 1;
  if (a.K % 8 || a.K > 32768 || a.batch > GVL_MAX_VALU_BATCH || a.batch == 3) return -1;
  if (a.batch == 1) { a.x_stride = 0; a.out_stride = 0; a.q_stride = 0; }
  const size_t lds = (size_t)a.K * 2 * a.batch;
  if (lds > 159 * 1024) return -1;
  // rows per wave: aim at ~768-1024 blocks (3-4 per CU, all co-resident) so the weight stream has no tail
  int R = (a.N + 4 * 1024 - 1) / (4 * 1024);
  if (R < 1) R = 1;
  if (R == 5) R = 6;
  if (R == 7 || R > 8) R = 8;
  if (a.batch > 1 && R > 4) R = 4;            // R x B accumulators and R x 4 weight registers per lane
  if (a.rope_on) { if ((a.Dr & 1) || (a.N & 1)) return -1; R = 2; }
  if (a.act == GVL_ACT_SILU_MUL && (R & 1)) R += 1;
  const int blocks = (a.N + 4 * R - 1) / (4 * R);
  switch (a.batch) {
    case 1: return launch_gemv_b<1>(a, R, blocks, lds, st);
    case 2: return launch_gemv_b<2>(a, R, blocks, lds, st);
    default: return launch_gemv_b<4>(a, R, blocks, lds, st);
  }
}

// greedy sampling: first index of the maximum (torch.argmax tie rule); one block per logit row
__global__ __launch_bounds__(1024) void argmax_kernel(const ArgmaxArgs a) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int b = blockIdx.x;
  const float* logits = a.logits + (size_t)b * a.n;
  float best = -3.4e38f; int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < a.n; i += 1024) {
    const float v = logits[i];
    if (v > best) { best = v; idx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    *a.tok_ptrs[b] = idx;
    if (a.ngen_ptrs[b]) {
      const int g = *a.ngen_ptrs[b]; if (a.out_lists[b]) a.out_lists[b][g] = idx; *a.ngen_ptrs[b] = g + 1;
      if (a.eos_flags[b] && idx == a.eos_id && *a.eos_flags[b] == 0) __hip_atomic_store(a.eos_flags[b], g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (a.pos_ptrs[b]) (*a.pos_ptrs[b])++;
  }
}
int gvl_launch_argmax(const ArgmaxArgs& a, hipStream_t st) {
  if (a.batch < 1 || a.batch > GVL_MAX_DECODE_BATCH) return -1;
  hipLaunchKernelGGL(argmax_kernel, dim3(a.batch), dim3(1024), 0, st, a);
  return CHECK_LAUNCH();
}
// ---- sampling (do_sample=True): the reference forwards do_sample / temperature / top_p to HF generate (models/llava_next_video.py:655-661;
// inference.py:45-49 defaults do_sample=True, T=0.2, top_p=None; HF's GenerationConfig adds top_k=50).  HF order [ext: transformers
// generation/logits_process.py]: scores / T -> top-k (keep scores >= the k-th largest, ties kept) -> top-p (sorted ascending, drop while
// the cumulative probability <= 1 - top_p, i.e. keep a token iff the mass of strictly larger scores is < top_p) -> softmax -> one draw.
// The draw is Gumbel-max, token = argmax_i (l_i - max) / T - log(-log u_i), u_i = counter hash of (seed, stream, step, i): a sample of
// exactly softmax(l / T) restricted to the kept set, with no sort and no prefix sum.  torch.multinomial's Philox stream cannot be
// reproduced, so parity is: same kept set and same token as the CPU restatement `sample_token` used by the tests (same hash), and the right distribution.
// One block per row; every pass re-reads the row from L2 (32 k - 128 k floats).  All reductions run in a fixed order and every
// thread sees the same totals, so the thresholds are wave-uniform and the result does not depend on the batch a row travels in.
__device__ __forceinline__ unsigned smp_fmix32(unsigned h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
__device__ __forceinline__ unsigned smp_key(float v) { const unsigned b = __float_as_uint(v); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }   // order-preserving
__device__ __forceinline__ float smp_block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);          // butterfly: bitwise the same total in every lane
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += sh[w];
  return t;
}
__global__ __launch_bounds__(1024) void sample_kernel(const ArgmaxArgs a) {
  __shared__ float shf[16];
  __shared__ int shi[16];
  __shared__ int hist[256];
  __shared__ unsigned s_sel[2];
  const int b = blockIdx.x, tid = threadIdx.x, n = a.n;
  const float* l = a.logits + (size_t)b * n;
  // 1. row maximum
  float m = -3.4e38f;
  for (int i = tid; i < n; i += 1024) m = fmaxf(m, l[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) shf[tid >> 6] = m;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w) m = fmaxf(m, shf[w]);
  // 2. top-k: key of the k-th largest score by an 8-bit radix select over the order-preserving keys (integer counts: exact)
  unsigned kth = 0;
  if (a.top_k > 0 && a.top_k < n) {
    unsigned prefix = 0; int remaining = a.top_k;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += 1024) {
        const unsigned k = smp_key(l[i]);
        if (shift == 24 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(k >> shift) & 255], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int c = 0, bsel = 0;
        for (int q = 255; q >= 0; --q) { if (c + hist[q] >= remaining) { bsel = q; break; } c += hist[q]; }
        s_sel[0] = prefix | ((unsigned)bsel << shift); s_sel[1] = (unsigned)(remaining - c);
      }
      __syncthreads();
      prefix = s_sel[0]; remaining = (int)s_sel[1];
      __syncthreads();
    }
    kth = prefix;
  }
  // 3. normaliser of the kept scores
  float z = 0.f;
  for (int i = tid; i < n; i += 1024) { const float v = l[i]; if (smp_key(v) >= kth) z += expf((v - m) * a.inv_temp); }
  const float Z = smp_block_sum(z, shf);
  // 4. top-p: smallest key t such that the mass of keys > t is < top_p * Z (the maximum itself always qualifies: min_tokens_to_keep = 1)
  unsigned thr = kth;
  if (a.top_p > 0.f && a.top_p < 1.f) {
    const float target = a.top_p * Z;
    unsigned lo = kth, hi = smp_key(m);
    while (lo < hi) {
      const unsigned mid = lo + ((hi - lo) >> 1);
      float s = 0.f;
      for (int i = tid; i < n; i += 1024) { const float v = l[i]; if (smp_key(v) > mid) s += expf((v - m) * a.inv_temp); }
      const float S = smp_block_sum(s, shf);
      if (S < target) hi = mid; else lo = mid + 1;
    }
    thr = lo;
  }
  // 5. Gumbel-max draw over the kept set
  const int step = a.ngen_ptrs[b] ? *a.ngen_ptrs[b] : (a.step_override ? a.step_override[b] : 0);
  const unsigned k0 = smp_fmix32(a.seed_lo ^ 0x9e3779b9u), k1 = smp_fmix32(a.seed_hi ^ k0 ^ 0x85ebca77u);
  const unsigned kk = smp_fmix32(k1 ^ smp_fmix32(a.stream[b] * 0x9e3779b1u + 0x7f4a7c15u) ^ smp_fmix32((unsigned)step * 0x85ebca77u + 0x165667b1u));
  const unsigned kk2 = smp_fmix32(kk + 0x632be5abu);
  float best = -3.4e38f; int idx = 0x7fffffff;
  for (int i = tid; i < n; i += 1024) {
    const float v = l[i];
    if (smp_key(v) < thr) continue;
    const unsigned h = smp_fmix32(smp_fmix32((unsigned)i + kk) ^ kk2);
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float sc = (v - m) * a.inv_temp - logf(-logf(u));
    if (sc > best) { best = sc; idx = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { shf[tid >> 6] = best; shi[tid >> 6] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) if (shf[w] > best || (shf[w] == best && shi[w] < idx)) { best = shf[w]; idx = shi[w]; }
    *a.tok_ptrs[b] = idx;
    if (a.ngen_ptrs[b]) {
      const int g = *a.ngen_ptrs[b]; if (a.out_lists[b]) a.out_lists[b][g] = idx; *a.ngen_ptrs[b] = g + 1;
      if (a.eos_flags[b] && idx == a.eos_id && *a.eos_flags[b] == 0) __hip_atomic_store(a.eos_flags[b], g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (a.pos_ptrs[b]) (*a.pos_ptrs[b])++;
  }
}
int gvl_launch_sample(const ArgmaxArgs& a, hipStream_t st) {
  if (a.batch < 1 || a.batch > GVL_MAX_DECODE_BATCH || !(a.inv_temp > 0.f) || a.top_k < 0 || a.top_p < 0.f) return -1;
  hipLaunchKernelGGL(sample_kernel, dim3(a.batch), dim3(1024), 0, st, a);
  return CHECK_LAUNCH();
}
__global__ void gather_tok_rows_kernel(const bf16_t* __restrict__ table, const TokPtrs toks, bf16_t* __restrict__ dst, int cols) {
  const int r = blockIdx.y;
  const int tok = *toks.p[r];
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8; c < cols; c += gridDim.x * blockDim.x * 8)
    *(u32x4_t*)(dst + (size_t)r * cols + c) = *(const u32x4_t*)(table + (size_t)tok * cols + c);
}
int gvl_launch_gather_tok_rows(const bf16_t* table, const TokPtrs& toks, bf16_t* dst, int cols, hipStream_t st) {
  if (cols % 8 || toks.n < 1 || toks.n > GVL_MAX_DECODE_BATCH) return -1;
  hipLaunchKernelGGL(gather_tok_rows_kernel, dim3((cols / 8 + 255) / 256, toks.n), dim3(256), 0, st, table, toks, dst, cols);
  return CHECK_LAUNCH();
}
// dst[r] = table[ids[r]] with the ids passed BY VALUE (256 per launch): no device staging buffer, so back-to-back calls on a busy
// stream cannot overwrite each other's ids (a blocking hipMemcpy into one shared buffer could).
__global__ void gather_rows_byval_kernel(const bf16_t* __restrict__ table, const IntList ids, bf16_t* __restrict__ dst, int cols) {
  const int r = blockIdx.y;
  const size_t src = (size_t)ids.v[r] * cols;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8; c < cols; c += gridDim.x * blockDim.x * 8)
    *(u32x4_t*)(dst + (size_t)r * cols + c) = *(const u32x4_t*)(table + src + c);
}
int gvl_launch_gather_rows_host_ids(const bf16_t* table, const int* host_ids, int n, bf16_t* dst, int cols, hipStream_t st) {
  if (cols % 8 || n < 0) return -1;
  constexpr int CH = (int)(sizeof(IntList::v) / sizeof(int));
  for (int r0 = 0; r0 < n; r0 += CH) {
    IntList l; l.n = n - r0 < CH ? n - r0 : CH;
    memcpy(l.v, host_ids + r0, (size_t)l.n * sizeof(int));
    hipLaunchKernelGGL(gather_rows_byval_kernel, dim3((cols / 8 + 255) / 256, l.n), dim3(256), 0, st, table, l, dst + (size_t)r0 * cols, cols);
  }
  return CHECK_LAUNCH();
}
// ---- training-forward loss tail: rows with a label -> final norm -> lm_head -> cross entropy -------------
// nll[r] = logsumexp(logits[r,:]) - logits[r, target[r]] in f32 on the bf16 logits (`logits.float()` then CrossEntropyLoss,
// modeling_phi3.py:1527-1539).  One block per row; max pass, then sum-of-exponentials pass, fixed reduction order.
__global__ void __launch_bounds__(256) ce_rows_kernel(const bf16_t* __restrict__ logits, int ld, const int* __restrict__ targets,
                                                      float* __restrict__ nll, int V) {
  __shared__ float red[4];
  const bf16_t* row = logits + (size_t)blockIdx.x * ld;
  const int tid = threadIdx.x, wid = tid >> 6, ln = tid & 63;
  float m = -3.0e38f;
  for (int c = tid; c < V; c += 256) m = fmaxf(m, bf2f(row[c]));
  m = wave_max(m);
  if (ln == 0) red[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < V; c += 256) s += expf(bf2f(row[c]) - m);
  s = wave_sum(s);
  if (ln == 0) red[wid] = s;
  __syncthreads();
  if (tid == 0) nll[blockIdx.x] = (m + logf((red[0] + red[1]) + (red[2] + red[3]))) - bf2f(row[targets[blockIdx.x]]);
}
int gvl_launch_ce_rows(const bf16_t* logits, int ld, const int* targets, float* nll, int n, int V, hipStream_t st) {
  if (n < 1 || V < 1) return -1;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(n), dim3(256), 0, st, logits, ld, targets, nll, V);
  return CHECK_LAUNCH();
}
// one KV page (all layers, K and V^T pools) copied to another page: grid (layers, 2), 16-byte pieces.  The partial last page of a cloned sequence.
__global__ __launch_bounds__(256) void kv_page_copy_kernel(bf16_t* kpool, bf16_t* vpool, size_t layer_stride, size_t page_elems, int src_page, int dst_page) {
  bf16_t* pool = blockIdx.y == 0 ? kpool : vpool;
  const u32x4_t* s = (const u32x4_t*)(pool + (size_t)blockIdx.x * layer_stride + (size_t)src_page * page_elems);
  u32x4_t* d = (u32x4_t*)(pool + (size_t)blockIdx.x * layer_stride + (size_t)dst_page * page_elems);
  for (size_t i = threadIdx.x; i < page_elems / 8; i += 256) d[i] = s[i];
}
int gvl_launch_kv_page_copy(bf16_t* kpool, bf16_t* vpool, size_t layer_stride, size_t page_elems, int layers, int src_page, int dst_page, hipStream_t st) {
  if (layers < 1 || (page_elems & 7) || src_page == dst_page) return -1;
  hipLaunchKernelGGL(kv_page_copy_kernel, dim3(layers, 2), dim3(256), 0, st, kpool, vpool, layer_stride, page_elems, src_page, dst_page);
  return CHECK_LAUNCH();
}
__global__ void inc_many_kernel(const IntPtrs ptrs) { if ((int)threadIdx.x < ptrs.n) (*ptrs.p[threadIdx.x])++; }
int gvl_launch_inc_many(const IntPtrs& ptrs, hipStream_t st) {
  if (ptrs.n < 1 || ptrs.n > GVL_MAX_DECODE_BATCH) return -1;
  hipLaunchKernelGGL(inc_many_kernel, dim3(1), dim3(64), 0, st, ptrs);
  return CHECK_LAUNCH();
}
__global__ void inc_kernel(int* p) { if (threadIdx.x == 0) (*p)++; }
int gvl_launch_inc(int* p, hipStream_t st) {
  hipLaunchKernelGGL(inc_kernel, dim3(1), dim3(64), 0, st, p);
  return CHECK_LAUNCH();
}
__global__ void set_int_kernel(int* p, int v) { if (threadIdx.x == 0) *p = v; }
int gvl_launch_set_int(int* p, int v, hipStream_t st) {
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(64), 0, st, p, v);
  return CHECK_LAUNCH();
}
