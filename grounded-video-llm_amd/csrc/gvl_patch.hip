// gvl_patch.hip -- patch embedding of both vision towers as ONE kernel (round 4), gfx950.
//
// Replaces, per tower, three passes -- patchify_kernel (f32 pixels -> bf16 im2col matrix, 258 us per 96-segment call: 5 x the GEMM it feeds),
// the patch GEMM, and clip_embed_ln_kernel / iv2_embed_kernel (CLS row, position embedding, CLIP's pre_layrnorm) -- i.e.
//   CLIP        : embeddings = cat(class_embedding, conv2d(px, stride = kernel = 14)) + position_embedding; pre_layrnorm
//                 (models/modeling_clip.py:182-191, :851)            -> x f32 [n][1 + P][C]
//   InternVideo2: x = cat(cls_token, conv3d(px, kernel (1,14,14)) + bias) + pos_embed, all bf16
//                 (models/internvideo2.py:714-725, :972-1011)        -> x bf16 [n][1 + T L][C]
//
// Structure.  The convolution is a GEMM over K = 3 * 14 * 14 = 588.  Its contraction is re-ordered to k' = (c * 14 + py) * 16 + px with
// two zero weight columns per pixel row (K' = 672 = 21 MFMA k steps of 32): one k step is then TWO WHOLE PIXEL ROWS of a patch, so
//   * the im2col gather happens in the loader of the patch operand: 8-byte coalesced reads of the f32 pixels (a patch row is 56 contiguous,
//     8-byte aligned bytes), converted to bf16 on the way into LDS -- the pixels are read once, nothing is written back;
//   * a block owns 48 patch rows x ALL C output columns (8 waves x C / 8 columns): the CLS row, the position rows and CLIP's LayerNorm over
//     the full row are an epilogue on the accumulators.
// MFMA: v_mfma_f32_16x16x32_bf16, A = 16 weight rows (output columns), B = 16 patch rows, so a lane ends up with 4 CONSECUTIVE output
// columns of one patch row (16-byte f32 / 8-byte bf16 stores).  The weights are read by every block, so they never touch LDS: a tile-order
// copy [column block of 16][k step][64 lanes][8 bf16] is built once at gvl_finalize_weights (retile_patch_weight_kernel; 1.4 / 1.9 MB, L2
// resident) and one wave load is 1 KiB of consecutive addresses, straight into the A operand registers.  The patch operand of a channel
// (48 rows x 14 pixel rows x 16 = 21 KB as bf16) is double buffered in LDS: channel c + 1 is in flight (global -> registers) while the 7 k
// steps of channel c run.  Accumulators: 48 x C / 8 per wave = 96 (C = 1024) / 132 (C = 1408) registers; 8 waves, one block per CU.
//
// Numerics: the reference's rounding points are kept -- conv output (+ bias) rounded to bf16, position add in f32 (CLIP, then LayerNorm in f32)
// or in bf16 (InternVideo2).  The fp32 accumulation ORDER over k differs from the three-pass path (c, py, px instead of the GEMM's k-tile
// order): results agree with it to fp32 rounding before the bf16 round, not bit for bit (tests: both paths against the reference goldens,
// and against each other).
#include "gvl_internal.h"

#define CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -3)

namespace {
constexpr int PE_BM = 48;              // patch rows per block
constexpr int PE_PW = 16;              // a pixel row of a patch, padded 14 -> 16
constexpr int PE_NW = 8;               // waves per block: wave w owns output columns [w * C / 8, (w + 1) * C / 8)
typedef __attribute__((ext_vector_type(4))) float pf32x4_t;
}  // namespace

// W [C][Kp] row-major, k = c * p * p + py * p + px  ->  Wt [C / 16][3 * p / 2][64 lanes][8]:
// lane l of tile (cb, s): output column cb * 16 + (l & 15), k chunk g = l >> 4 = pixel row 2 s + (g >> 1) of the (c, py) sequence, px 8 (g & 1) .. + 7 (0 beyond p)
__global__ void retile_patch_weight_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ Wt, int C, int Kp, int p) {
  const int KS = 3 * p / 2;
  const long total = (long)(C / 16) * KS * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int l = (int)(idx & 63);
    const long tile = idx >> 6;
    const int s = (int)(tile % KS), cb = (int)(tile / KS);
    const int col = cb * 16 + (l & 15), g = l >> 4;
    const int prow = 2 * s + (g >> 1), c = prow / p, py = prow - c * p, px0 = 8 * (g & 1);
    bf16_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = px0 + e < p ? W[(size_t)col * Kp + c * p * p + py * p + px0 + e] : (bf16_t)0;
    u32x4_t o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
    *(u32x4_t*)(Wt + idx * 8) = o;
  }
}
int gvl_retile_patch_weight(const bf16_t* W, bf16_t* Wt, int C, int Kp, int p, hipStream_t st) {
  if (C % 16 || p > PE_PW || (p & 1)) return -1;
  const long total = (long)(C / 16) * (3 * p / 2) * 64;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(retile_patch_weight_kernel, dim3(blocks), dim3(256), 0, st, W, Wt, C, Kp, p);
  return CHECK_LAUNCH();
}

// NCT: 16-column tiles per wave (C / 128).  MODE 0: CLIP (f32 rows, LayerNorm), MODE 1: InternVideo2 (bf16 rows, conv bias).
template <int NCT, int MODE>
__global__ __launch_bounds__(PE_NW * 64, 2) void patch_embed_kernel(const PatchEmbedArgs a) {
  constexpr int NT = PE_NW * 64, RT = PE_BM / 16;          // threads; 16-row tiles per block
  const int p = a.patch, g_ = a.image / p, L = g_ * g_;
  const int ASTR = p * 32 + 16;                             // bytes per patch row of one channel's LDS image: p pixel rows of 16 bf16, + 16 (bank spread)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = (a.M + PE_BM - 1) / PE_BM;
  const int C = a.C;

  if ((int)blockIdx.x >= n_tiles) {
    // ---- the CLS rows: cls + pos[0] (the same for every image), one wave per image -------------------------------------------------
    const int img = ((int)blockIdx.x - n_tiles) * PE_NW + wave;
    if (img >= a.n_img) return;
    const size_t row = (size_t)img * a.S;
    if constexpr (MODE == 0) {
      float sum = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const pf32x4_t v = *(const pf32x4_t*)(a.cls_f32 + c) + *(const pf32x4_t*)(a.pos_f32 + c);
        sum += v[0] + v[1] + v[2] + v[3];
      }
      const float mean = wave_sum(sum) / C;
      float q = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const pf32x4_t v = *(const pf32x4_t*)(a.cls_f32 + c) + *(const pf32x4_t*)(a.pos_f32 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
      }
      const float rstd = rsqrtf(wave_sum(q) / C + a.eps);
      for (int c = lane * 4; c < C; c += 256) {
        const pf32x4_t v = *(const pf32x4_t*)(a.cls_f32 + c) + *(const pf32x4_t*)(a.pos_f32 + c);
        const pf32x4_t wv = *(const pf32x4_t*)(a.lnw + c), bv = *(const pf32x4_t*)(a.lnb + c);
        pf32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * wv[e] + bv[e];
        *(pf32x4_t*)(a.x_f32 + row * C + c) = o;
      }
    } else {
      for (int c = lane * 8; c < C; c += 512) {
        const u32x4_t sv = *(const u32x4_t*)(a.cls_bf + c), pv = *(const u32x4_t*)(a.pos_bf + c);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(sv[e]) + lo_bf(pv[e]), hi_bf(sv[e]) + hi_bf(pv[e]));
        *(u32x4_t*)(a.x_bf + row * C + c) = o;
      }
    }
    return;
  }

  const int m0 = blockIdx.x * PE_BM;
  // ---- loader: channel c of the block's 48 patch rows, items (row r, pixel row py, float2 j): p * p / 2 per row ------------------------
  const int ipr = p * (p / 2);                              // items per patch row and channel
  const int n_items = PE_BM * ipr;
  constexpr int MAXI = 10;                                  // items per thread (48 * 98 / 512 = 9.2)
  unsigned goff[MAXI]; unsigned short loff[MAXI];           // pixel offset (floats, from the channel-0 plane of the item's image / frame), LDS byte offset
  const size_t plane = (size_t)a.T * a.image * a.image;     // floats between channels
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int q = it * NT + tid;
    goff[it] = 0xffffffffu; loff[it] = 0;
    if (q < n_items) {
      const int r = q / ipr, rem = q - r * ipr, py = rem / (p / 2), j = rem - py * (p / 2);
      int m = m0 + r; if (m > a.M - 1) m = a.M - 1;         // tail tile: duplicates of the last row (never stored)
      const int gx = m % g_, gy = (m / g_) % g_, t = (m / L) % a.T, img = m / (L * a.T);
      const size_t off = (((size_t)img * 3) * a.T + t) * a.image * a.image + (size_t)(gy * p + py) * a.image + gx * p + 2 * j;
      goff[it] = (unsigned)off;                             // < 2^32 floats: checked by the launcher
      loff[it] = (unsigned short)(r * ASTR + py * 32 + 4 * j);
    }
  }
  typedef __attribute__((ext_vector_type(2))) float pf32x2_t;
  pf32x2_t stage[MAXI];
  auto load_channel = [&](int c) {
#pragma unroll
    for (int it = 0; it < MAXI; ++it)
      if (goff[it] != 0xffffffffu) stage[it] = *(const pf32x2_t*)(a.px + (size_t)c * plane + goff[it]);
  };
  auto store_channel = [&](int buf) {
    char* dst = smem + buf * (PE_BM * ASTR);
#pragma unroll
    for (int it = 0; it < MAXI; ++it)
      if (goff[it] != 0xffffffffu) *(unsigned*)(dst + loff[it]) = pack2bf(stage[it][0], stage[it][1]);
  };
  // zero both LDS images once: the two pad columns of every pixel row (and the row padding) stay zero, the loader never writes them
  for (int i = tid * 16; i < 2 * PE_BM * ASTR; i += NT * 16) *(u32x4_t*)(smem + i) = u32x4_t{0u, 0u, 0u, 0u};
  load_channel(0);
  __syncthreads();
  store_channel(0);

  // ---- MFMA operands ----------------------------------------------------------------------------------------------------------------------
  const int j16 = lane & 15, g4 = lane >> 4;
  const int KS = 3 * p / 2, spc = p / 2;                    // k steps in all, per channel
  const bf16_t* wt = a.Wt + ((size_t)(wave * NCT) * KS * 64 + lane) * 8;      // tile (wave * NCT + ct, s): + (ct * KS + s) * 512 elements
  const unsigned b_lds = (unsigned)(j16 * ASTR + (g4 >> 1) * 32 + (g4 & 1) * 16);   // + rt * 16 * ASTR + (s within the channel) * 64
  pf32x4_t acc[RT][NCT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[rt][ct] = pf32x4_t{0.f, 0.f, 0.f, 0.f};

  // The weight fragments come straight from L2 (1 KiB per wave load): fragment ct of k step s + 1 is requested right behind the MFMAs that
  // consumed fragment ct of step s -- the same registers, a whole k step of MFMAs (NCT - 1 other fragments x RT) to land in.
  bf16x8_t wfr[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) wfr[ct] = *(const bf16x8_t*)(wt + (size_t)ct * KS * 512);
  constexpr int SPC = 7;                                    // k steps per channel (patch 14: checked by the launcher)
  for (int c = 0; c < 3; ++c) {
    __syncthreads();                                        // channel c's image is complete; every wave is done reading the other buffer
    if (c + 1 < 3) load_channel(c + 1);                     // in flight under the 7 k steps below
    const char* img_ = smem + (c & 1) * (PE_BM * ASTR);
#pragma unroll
    for (int sl = 0; sl < SPC; ++sl) {
      const int s = c * SPC + sl;
      const int sn = s + 1 < KS ? s + 1 : s;                // the last step re-requests itself (never consumed)
      bf16x8_t bfr[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) bfr[rt] = *(const bf16x8_t*)(img_ + b_lds + rt * 16 * ASTR + sl * 64);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[ct], bfr[rt], acc[rt][ct], 0, 0, 0);
        wfr[ct] = *(const bf16x8_t*)(wt + ((size_t)ct * KS + sn) * 512);
      }
    }
    if (c + 1 < 3) store_channel((c + 1) & 1);              // the other buffer: its last readers passed the barrier at the top of this iteration
  }

  // ---- epilogue: lane (j16, g4) holds, per (row tile rt, column tile ct), patch row m0 + rt * 16 + j16, columns col0 + ct * 16 + 4 g4 .. + 3 --------
  const int col0 = wave * NCT * 16 + 4 * g4;
  if constexpr (MODE == 1) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int m = m0 + rt * 16 + j16;
      if (m < a.M) {
        const int b = m / (L * a.T), pidx = m - b * (L * a.T);
        const bf16_t* prow = a.pos_bf + (size_t)(1 + pidx) * C;
        bf16_t* xrow = a.x_bf + ((size_t)b * a.S + 1 + pidx) * C;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int col = col0 + ct * 16;
          const pf32x4_t bv = *(const pf32x4_t*)(a.bias + col);
          const u32x2_t pv = *(const u32x2_t*)(prow + col);
          const pf32x4_t v = acc[rt][ct];
          // conv output (+ bias) rounded to bf16 (the GEMM's store), then the bf16 position add (iv2_embed_kernel)
          u32x2_t o = {pack2bf(rbf(v[0] + bv[0]) + lo_bf(pv[0]), rbf(v[1] + bv[1]) + hi_bf(pv[0])), pack2bf(rbf(v[2] + bv[2]) + lo_bf(pv[1]), rbf(v[3] + bv[3]) + hi_bf(pv[1]))};
          *(u32x2_t*)(xrow + col) = o;
        }
      }
    }
  } else {
    // CLIP: v = bf16(conv) + pos (f32), LayerNorm over the C columns of the row: 8 waves x 4 lanes hold a row -> statistics through LDS
    __syncthreads();                                        // the channel images are dead: reuse the LDS for the row statistics
    float* st_sum = (float*)smem;                           // [PE_NW][PE_BM]
    float* st_sq = st_sum + PE_NW * PE_BM;
    float mean[RT], rstd[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int m = m0 + rt * 16 + j16; if (m > a.M - 1) m = a.M - 1;
      const int pidx = m % L;
      const float* prow = a.pos_f32 + (size_t)(1 + pidx) * C;
      float s = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const pf32x4_t pv = *(const pf32x4_t*)(prow + col0 + ct * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[rt][ct][e] = rbf(acc[rt][ct][e]) + pv[e]; s += acc[rt][ct][e]; }
      }
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      if (g4 == 0) st_sum[wave * PE_BM + rt * 16 + j16] = s;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < PE_NW; ++w) s += st_sum[w * PE_BM + rt * 16 + j16];
      mean[rt] = s / C;
      float q = 0.f;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = acc[rt][ct][e] - mean[rt]; q += d * d; }
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (g4 == 0) st_sq[wave * PE_BM + rt * 16 + j16] = q;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float q = 0.f;
#pragma unroll
      for (int w = 0; w < PE_NW; ++w) q += st_sq[w * PE_BM + rt * 16 + j16];
      rstd[rt] = rsqrtf(q / C + a.eps);
      const int m = m0 + rt * 16 + j16;
      if (m < a.M) {
        const int img = m / L, pidx = m - img * L;
        float* xrow = a.x_f32 + ((size_t)img * a.S + 1 + pidx) * C;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int col = col0 + ct * 16;
          const pf32x4_t wv = *(const pf32x4_t*)(a.lnw + col), bv = *(const pf32x4_t*)(a.lnb + col);
          pf32x4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (acc[rt][ct][e] - mean[rt]) * rstd[rt] * wv[e] + bv[e];
          *(pf32x4_t*)(xrow + col) = o;
        }
      }
    }
  }
}

// 0 launched; -1 geometry outside this kernel (the caller takes the three-pass path); -3 launch error
int gvl_launch_patch_embed(const PatchEmbedArgs& a, hipStream_t st) {
  const int p = a.patch;
  if (p != 14 || a.image % p || a.C % 128 || (a.C != 1024 && a.C != 1408) || a.M <= 0 || a.T <= 0) return -1;
  if ((size_t)a.n_img * 3 * a.T * a.image * a.image >= 0xffffffffull) return -1;      // 32-bit pixel offsets
  const int ASTR = p * 32 + 16;
  const int lds = 2 * PE_BM * ASTR;
  const int tiles = (a.M + PE_BM - 1) / PE_BM, grid = tiles + (a.n_img + PE_NW - 1) / PE_NW;
  if (a.mode == 0) {
    if (a.C != 1024 || !a.x_f32 || !a.cls_f32 || !a.pos_f32 || !a.lnw || !a.lnb || a.T != 1) return -1;
    hipLaunchKernelGGL((patch_embed_kernel<8, 0>), dim3(grid), dim3(PE_NW * 64), lds, st, a);
  } else {
    if (!a.x_bf || !a.cls_bf || !a.pos_bf || !a.bias) return -1;
    if (a.C == 1408) hipLaunchKernelGGL((patch_embed_kernel<11, 1>), dim3(grid), dim3(PE_NW * 64), lds, st, a);
    else hipLaunchKernelGGL((patch_embed_kernel<8, 1>), dim3(grid), dim3(PE_NW * 64), lds, st, a);
  }
  return CHECK_LAUNCH();
}
