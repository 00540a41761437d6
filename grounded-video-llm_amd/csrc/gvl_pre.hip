// gvl_pre.hip -- frame pre-processing on the GPU: uint8 video frames -> normalised f32 pixel tensors.
//
// Replaces, per frame, the CPU chain of the reference's frame_transform (mm_utils/utils.py:153-183, called 96 + 12 times per clip
// from inference.py:69-88): ToPILImage -> torchvision Resize(S, BICUBIC) [= PIL.Image.resize, Pillow src/libImaging/Resample.c] ->
// CenterCrop(S) -> ToTensor (/255) -> Normalize.  It removes 108 PIL resizes and the 74 MB host->device copy of f32 pixels from a
// clip's critical path: the decoder's uint8 frames are uploaded once (or decoded on the device) and stay in HBM.
//
// BIT-EXACT to Pillow: its 8-bit resampler is integer arithmetic -- two separable passes (horizontal, then vertical on the 8-bit
// result), per output pixel a window of ceil(2*max(scale,1))*2+1 taps with fixed-point weights round(w * 2^22), an int32
// accumulator that starts at 2^21, an arithmetic shift and a clamp to [0,255].  The weights are computed on the host in double
// precision with exactly the operation order of precompute_coeffs()/normalize_coeffs_8bpc() (no FMA contraction), the passes run
// here.  Only the centre-crop window is computed: S columns of the horizontal pass, and of those only the rows the vertical pass
// of the S kept rows reads.  HBM-bound byte work: each source byte inside the window is read by ~ksize neighbouring threads of a
// wave (served by L1/L2), each output written once.  The CPU checker of this path (test infrastructure, not part of the product) is pinned
// against Pillow itself (tests/golden/preprocess.npz).
#include "gvl_internal.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;     // Resample.c

#pragma clang fp contract(off)
double bicubic_filter(double x) {               // Resample.c bicubic_filter, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// precompute_coeffs + normalize_coeffs_8bpc for output positions [o0, o0 + n_out) of an in_size -> out_size resize
void resample_coeffs(int in_size, int out_size, int o0, int n_out, std::vector<int>& bounds, std::vector<int>& kk, int& ksize) {
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  ksize = (int)ceil(support) * 2 + 1;
  bounds.assign((size_t)n_out * 2, 0);
  kk.assign((size_t)n_out * ksize, 0);
  std::vector<double> k(ksize);
  const double ss = 1.0 / filterscale;
  for (int i = 0; i < n_out; ++i) {
    const int xx = o0 + i;
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      const double v = k[x];
      kk[(size_t)i * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[2 * i] = xmin; bounds[2 * i + 1] = xmax;
  }
}

__device__ __forceinline__ int clip8(int v) { v >>= PRECISION_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

struct PreArgs {
  const unsigned char* src; int n, H, W, layout;          // layout 0: [n][H][W][3], 1: [n][3][H][W]
  int size, left, top, r0, R;                              // crop window; tmp holds source rows [r0, r0 + R) of the horizontally resized image
  const int *hb, *hk; int hks; int h_identity;             // horizontal taps for columns [left, left + size)
  const int *vb, *vk; int vks; int v_identity;             // vertical taps for rows [top, top + size)
  unsigned char* tmp;                                      // [n][3][R][size]
  float mean[3], inv_dummy, stdv[3];
  float* out;                                              // [n][3][size][size]
};

// horizontal pass: one block per (frame, source row): the span of the row that the S output columns read is staged in LDS with
// coalesced 4-byte loads (once), then one thread per output column runs its taps out of LDS for the three channels
__global__ __launch_bounds__(256) void pre_h_kernel(const PreArgs a, int x_lo, int x_hi) {
  extern __shared__ unsigned char srow[];           // [3][span] (planar) -- span = x_hi - x_lo source pixels
  const int r = blockIdx.x % a.R, f = blockIdx.x / a.R;
  const int y = a.r0 + r, span = x_hi - x_lo;
  if (a.layout == 0) {
    const unsigned char* p = a.src + (((size_t)f * a.H + y) * a.W + x_lo) * 3;      // interleaved RGB: de-interleave into planes
    for (int i = threadIdx.x; i < span * 3; i += 256) { const int x = i / 3, c = i - 3 * x; srow[c * span + x] = p[i]; }
  } else {
    for (int c = 0; c < 3; ++c) {
      const unsigned char* p = a.src + (((size_t)f * 3 + c) * a.H + y) * a.W + x_lo;
      for (int i = threadIdx.x; i < span; i += 256) srow[c * span + i] = p[i];
    }
  }
  __syncthreads();
  for (int ox = threadIdx.x; ox < a.size; ox += 256) {
    int px[3];
    if (a.h_identity) {
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = srow[c * span + a.left + ox - x_lo];
    } else {
      const int xmin = a.hb[2 * ox] - x_lo, cnt = a.hb[2 * ox + 1];
      const int* k = a.hk + (size_t)ox * a.hks;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int x = 0; x < cnt; ++x) { const int kv = k[x]; s0 += srow[xmin + x] * kv; s1 += srow[span + xmin + x] * kv; s2 += srow[2 * span + xmin + x] * kv; }
      px[0] = clip8(s0); px[1] = clip8(s1); px[2] = clip8(s2);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) a.tmp[(((size_t)f * 3 + c) * a.R + r) * a.size + ox] = (unsigned char)px[c];
  }
}

// vertical pass + ToTensor + Normalize: one thread per 4 adjacent output columns (4-byte loads of the 8-bit intermediate)
__global__ __launch_bounds__(256) void pre_v_kernel(const PreArgs a) {
  const int q = a.size >> 2;                         // size % 4 == 0 on this path
  const long total = (long)a.n * 3 * a.size * q;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % q) * 4;
    long t = i / q;
    const int oy = (int)(t % a.size); t /= a.size;
    const int c = (int)(t % 3), f = (int)(t / 3);
    const unsigned char* col = a.tmp + (((size_t)f * 3 + c) * a.R) * a.size + ox;
    int u[4];
    if (a.v_identity) {
      const unsigned w = *(const unsigned*)(col + (size_t)(a.top + oy - a.r0) * a.size);
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = (w >> (8 * e)) & 255;
    } else {
      const int ymin = a.vb[2 * oy], cnt = a.vb[2 * oy + 1];
      const int* k = a.vk + (size_t)oy * a.vks;
      int s[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
      for (int y = 0; y < cnt; ++y) {
        const unsigned w = *(const unsigned*)(col + (size_t)(ymin + y - a.r0) * a.size);
        const int kv = k[y];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += (int)((w >> (8 * e)) & 255) * kv;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = clip8(s[e]);
    }
    // ToTensor: uint8 -> f32, / 255 ; Normalize: (x - mean) / std   (three separately rounded IEEE f32 operations, as torch does)
    f32x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u[e], 255.0f), a.mean[c]), a.stdv[c]);
    *(f32x4_t*)(a.out + ((((size_t)f * 3 + c) * a.size + oy) * a.size + ox)) = o;
  }
}

// generic vertical pass (size % 4 != 0): one thread per output element
__global__ __launch_bounds__(256) void pre_v1_kernel(const PreArgs a) {
  const long total = (long)a.n * 3 * a.size * a.size;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % a.size);
    long t = i / a.size;
    const int oy = (int)(t % a.size); t /= a.size;
    const int c = (int)(t % 3), f = (int)(t / 3);
    const unsigned char* col = a.tmp + (((size_t)f * 3 + c) * a.R) * a.size + ox;
    int u;
    if (a.v_identity) {
      u = col[(size_t)(a.top + oy - a.r0) * a.size];
    } else {
      const int ymin = a.vb[2 * oy], cnt = a.vb[2 * oy + 1];
      const int* k = a.vk + (size_t)oy * a.vks;
      int s = 1 << (PRECISION_BITS - 1);
      for (int y = 0; y < cnt; ++y) s += col[(size_t)(ymin + y - a.r0) * a.size] * k[y];
      u = clip8(s);
    }
    a.out[i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)u, 255.0f), a.mean[c]), a.stdv[c]);
  }
}

}  // namespace

// torchvision 0.16.2: _compute_resized_output_size (shortest edge -> size) and center_crop offsets (Python round(): half to even)
static void tv_geometry(int H, int W, int size, int& nh, int& nw, int& top, int& left) {
  const int shrt = W <= H ? W : H, lng = W <= H ? H : W;
  const int new_long = (int)((double)size * lng / shrt);          // int(size * long / short): true division, truncation
  if (W <= H) { nw = size; nh = new_long; } else { nh = size; nw = new_long; }
  auto round_half_even = [](double v) { return (int)nearbyint(v); };   // default rounding mode = to nearest even, like Python round()
  top = round_half_even((nh - size) / 2.0);
  left = round_half_even((nw - size) / 2.0);
}

int gvl_launch_preprocess(const unsigned char* frames, int n, int H, int W, int layout, int size, const float* mean, const float* stdv, float* out,
                          void** scratch, size_t* scratch_bytes, hipStream_t st) {
  if (!frames || !out || n <= 0 || H <= 0 || W <= 0 || size <= 0 || (layout != 0 && layout != 1)) return -1;
  int nh, nw, top, left;
  tv_geometry(H, W, size, nh, nw, top, left);
  if (nh < size || nw < size) return -1;                              // cannot happen for a shortest-edge resize
  PreArgs a; memset(&a, 0, sizeof(a));
  a.src = frames; a.n = n; a.H = H; a.W = W; a.layout = layout; a.size = size; a.left = left; a.top = top; a.out = out;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.stdv[c] = stdv[c]; }
  std::vector<int> hb, hk, vb, vk;
  a.h_identity = nw == W; a.v_identity = nh == H;
  if (!a.h_identity) resample_coeffs(W, nw, left, size, hb, hk, a.hks);
  if (!a.v_identity) resample_coeffs(H, nh, top, size, vb, vk, a.vks);
  int r0 = top, r1 = top + size;
  if (!a.v_identity) {
    r0 = vb[0]; r1 = 0;
    for (int i = 0; i < size; ++i) { if (vb[2 * i] < r0) r0 = vb[2 * i]; if (vb[2 * i] + vb[2 * i + 1] > r1) r1 = vb[2 * i] + vb[2 * i + 1]; }
  }
  a.r0 = r0; a.R = r1 - r0;
  // scratch: [tmp bytes | horizontal bounds, taps | vertical bounds, taps] -- grown on demand, owned by the caller (ctx)
  const size_t tmp_bytes = ((size_t)n * 3 * a.R * size + 255) & ~(size_t)255;
  const size_t tab_ints = hb.size() + hk.size() + vb.size() + vk.size();
  const size_t need = tmp_bytes + tab_ints * 4 + 256;
  if (*scratch_bytes < need) {
    if (*scratch) { hipStreamSynchronize(st); hipFree(*scratch); *scratch = nullptr; *scratch_bytes = 0; }
    if (hipMalloc(scratch, need) != hipSuccess) return -2;
    *scratch_bytes = need;
  }
  a.tmp = (unsigned char*)*scratch;
  int* tab = (int*)((char*)*scratch + tmp_bytes);
  std::vector<int> host(tab_ints);
  size_t off = 0;
  auto put = [&](const std::vector<int>& v, const int*& dptr) { dptr = tab + off; std::copy(v.begin(), v.end(), host.begin() + off); off += v.size(); };
  put(hb, a.hb); put(hk, a.hk); put(vb, a.vb); put(vk, a.vk);
  if (tab_ints && hipMemcpyAsync(tab, host.data(), tab_ints * 4, hipMemcpyHostToDevice, st) != hipSuccess) return -3;
  if (tab_ints && hipStreamSynchronize(st) != hipSuccess) return -3;   // `host` dies with this frame; a few KB, once per call
  {
    // source span [x_lo, x_hi) read by the S output columns of a row
    int x_lo = left, x_hi = left + size;
    if (!a.h_identity) {
      x_lo = hb[0]; x_hi = 0;
      for (int i = 0; i < size; ++i) { if (hb[2 * i] < x_lo) x_lo = hb[2 * i]; if (hb[2 * i] + hb[2 * i + 1] > x_hi) x_hi = hb[2 * i] + hb[2 * i + 1]; }
    }
    const size_t lds = (size_t)(x_hi - x_lo) * 3;
    if (lds > 64 * 1024) return -1;                                    // > 21 k source pixels per row: not a video frame
    hipLaunchKernelGGL(pre_h_kernel, dim3((unsigned)(n * a.R)), dim3(256), lds, st, a, x_lo, x_hi);
  }
  {
    const bool vec = (size & 3) == 0 && (((uintptr_t)out) & 15) == 0;
    const long total = vec ? (long)n * 3 * size * (size >> 2) : (long)n * 3 * size * size;
    int blocks = (int)((total + 255) / 256); if (blocks > 65535) blocks = 65535;
    if (vec) hipLaunchKernelGGL(pre_v_kernel, dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pre_v1_kernel, dim3(blocks), dim3(256), 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
