// gvl_gemm4.hip -- the 4-wave form of the big bf16 MFMA GEMM (round 6):  C[M,N] = epilogue(A[M,K] . W[N,K]^T), gfx950 only.
//
// Same contract, same operand layout, same fused epilogues and -- element by element -- the same k order (hence the same bits) as gemm_pp_kernel of
// gvl_gemm.hip; what changes is who holds the tile and who schedules the loop:
//   * ONE wave per SIMD (256 threads, 512 registers per lane): CU tile 256 x 256 x 64, wave tile 128 x 128, the 256 accumulators in AGPRs a[0:255], operands
//     and addresses in VGPRs.  Per 16 MFMAs a wave reads 8 fragments (0.5 ds_read_b128 per MFMA; the 8-wave form: 0.75) -- a third less LDS -> register traffic
//     for the same flops, on a kernel that runs AT the board's power cap.
//   * the k loop is ONE hand-placed inline-asm statement per output tile (generated: tools/gen_gemm4_loop.py -> gvl_gemm4_loop.inc): fragment reads a k step
//     ahead into a second register set, global -> LDS DMA (buffer_load ... lds) and its M0 writes in the gaps between MFMAs, ONE barrier per k-tile (the
//     8-wave ping-pong needs eight), no priorities.  The DMA stream does not stop at a tile boundary: the last two k-tiles of a tile fetch k-tile 0 of the
//     workgroup's NEXT tile, so the epilogue runs with the next operands already in LDS.
//   * rows beyond the matrix are not clamped: the buffer resource's bounds check returns zeros for them (a wave whose sub-tile lies outside the matrix
//     multiplies zeros -- the cheapest thing a matrix pipe can do -- and skips its epilogue).
// The epilogue is the shared LDS-staged whole-row epilogue (gvl_gemm_epi.h); it reads the accumulators block row by block row out of the AGPRs.
// Reference shapes served: models/internvideo2.py:587,603,631-634; models/modeling_phi3.py:459-464,659-663; models/modeling_clip.py:264-266,340-342.
#include "gvl_gemm_epi.h"
#include "gvl_gemm4_loop.inc"
#include <cstdio>
#include <vector>

namespace {

// 16 accumulator registers a[BASE .. BASE + 15] -> one f32x16 (two statements: 30-operand limit of an asm statement)
template <int BASE>
__device__ __forceinline__ f32x16_t a4_read_block() {
  float v[16];
  asm volatile("v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c8+1]\n\tv_accvgpr_read_b32 %2, a[%c8+2]\n\tv_accvgpr_read_b32 %3, a[%c8+3]\n\t"
               "v_accvgpr_read_b32 %4, a[%c8+4]\n\tv_accvgpr_read_b32 %5, a[%c8+5]\n\tv_accvgpr_read_b32 %6, a[%c8+6]\n\tv_accvgpr_read_b32 %7, a[%c8+7]"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7]) : "i"(BASE));
  asm volatile("v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c8+1]\n\tv_accvgpr_read_b32 %2, a[%c8+2]\n\tv_accvgpr_read_b32 %3, a[%c8+3]\n\t"
               "v_accvgpr_read_b32 %4, a[%c8+4]\n\tv_accvgpr_read_b32 %5, a[%c8+5]\n\tv_accvgpr_read_b32 %6, a[%c8+6]\n\tv_accvgpr_read_b32 %7, a[%c8+7]"
               : "=v"(v[8]), "=v"(v[9]), "=v"(v[10]), "=v"(v[11]), "=v"(v[12]), "=v"(v[13]), "=v"(v[14]), "=v"(v[15]) : "i"(BASE + 8));
  f32x16_t r;
#pragma unroll
  for (int e = 0; e < 16; ++e) r[e] = v[e];
  return r;
}
template <int J>
__device__ __forceinline__ void a4_read_row(f32x16_t (&r)[4]) {
  r[0] = a4_read_block<(4 * J + 0) * 16>(); r[1] = a4_read_block<(4 * J + 1) * 16>(); r[2] = a4_read_block<(4 * J + 2) * 16>(); r[3] = a4_read_block<(4 * J + 3) * 16>();
}
struct A4AccRow {                                   // AccRow of gemm_epilogue_staged_rows: block row j of the wave tile, from the AGPRs the loop left
  __device__ __forceinline__ void operator()(int j, f32x16_t (&r)[4]) const {
    switch (j) { case 0: a4_read_row<0>(r); break; case 1: a4_read_row<1>(r); break; case 2: a4_read_row<2>(r); break; default: a4_read_row<3>(r); break; }
  }
};

template <int VAR>
__device__ __forceinline__ void a4_tile_loop(unsigned rdW0, unsigned rdA0, unsigned voW, unsigned voA, unsigned voWn, unsigned voAn, __amdgpu_buffer_rsrc_t rsW,
                                             __amdgpu_buffer_rsrc_t rsA, unsigned stepW, unsigned stepA, unsigned dma_other, unsigned nk) {
#define GVL_A4_OPERANDS : : "v"(rdW0), "v"(rdA0), "v"(voW), "v"(voA), "v"(voWn), "v"(voAn), "s"(rsW), "s"(rsA), "s"(stepW), "s"(stepA), "s"(dma_other), "s"(nk) \
                        : "memory", "scc", GVL_A4_CLOBBER_SGPRS, GVL_A4_CLOBBER_VGPRS, GVL_A4_CLOBBER_AGPRS
  if constexpr (VAR == 0) asm volatile(GVL_A4_TILE_ASM_V0 GVL_A4_OPERANDS);
  else if constexpr (VAR == 1) asm volatile(GVL_A4_TILE_ASM_V1 GVL_A4_OPERANDS);
  else asm volatile(GVL_A4_TILE_ASM_V2 GVL_A4_OPERANDS);
#undef GVL_A4_OPERANDS
}

}  // namespace

template <int EPI, int VAR>
__global__ __launch_bounds__(256) void gemm_a4_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 256, NWAVES = 4, TM = 128, TN = 128, MB = 4, NB = 4, SLOT = 65536;
  static_assert(EPI >= 0 && !(EPI & 4), "4-wave kernel: compile-time epilogues with bf16 output");
  using G = StgGeom<NB, EPI>;
  constexpr bool SILU = G::silu;
  constexpr int SWZ = SILU ? 0 : 1;
  constexpr int STG_BYTES = SILU ? 32 * (TN + 16) : 32 * TN * 2;
  constexpr int BG_OFF = 2 * SLOT;                                  // bias / gamma scratch: TN floats each per wave
  constexpr bool GELU_TAB = (EPI & 3) == GVL_ACT_GELU;
  constexpr int TAB_OFF = BG_OFF + NWAVES * TN * 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // persistent XCD-aware tile walk: identical to gemm_pp_kernel (gvl_gemm.hip)
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x, Gd = gridDim.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int xbase = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int xcnt = q + (xcd < r ? 1 : 0);
  const int wpx = (Gd + 7 - xcd) >> 3;
  auto tile_of = [&](int vid, int& om0, int& on0) {
    const int GM = a.band, band = GM * tiles_n;
    const int g = vid / band, first_m = g * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_band = vid - g * band;
    om0 = a.m_begin + (first_m + in_band % gm) * BM; on0 = (in_band / gm) * BN;
  };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int l31 = lane & 31, h = lane >> 5;
  const unsigned rd_lane = (unsigned)(l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4));   // this lane's k-step-0 chunk of fragment row l31 (chunk = (2 ph + h) ^ swizzle)
  const unsigned rdW = smem_base + (unsigned)(wn * TN * 128) + rd_lane, rdA = smem_base + (unsigned)(BN * 128 + wm * TM * 128) + rd_lane;
  const int rowl = wave * 8 + (lane >> 3);                                           // DMA piece i covers tile rows 32 i + rowl, 16-byte chunk lane & 7 of the LDS row
  const unsigned dma_lane = (unsigned)(((lane & 7) ^ ((rowl >> 1) & 7)) << 4);
  const unsigned pitchW = (unsigned)a.ldw * 2u, pitchA = (unsigned)a.lda * 2u;
  const unsigned stepW = __builtin_amdgcn_readfirstlane(pitchW * 32u), stepA = __builtin_amdgcn_readfirstlane(pitchA * 32u);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.W, 0, (unsigned)a.N * pitchW, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (unsigned)a.M * pitchA, 0x00020000);
  const unsigned nk = (unsigned)(a.K / BK);

  int it = bid >> 3;
  if (it >= xcnt) return;
  if (smem_base & 0x1ffffu) __builtin_trap();        // the ring slots are toggled by XOR 0x10000: the dynamic LDS block must start at a multiple of 128 KiB (it starts at 0)
  if constexpr (GELU_TAB) {                          // once per (persistent) workgroup; its first read is many barriers away
    for (int o = tid * 16; o < GELU_TAB_BYTES; o += 256 * 16) *(u32x4_t*)(smem + TAB_OFF + o) = *(const u32x4_t*)((const char*)a.act_table + o);
  }
  const char* tabp = GELU_TAB ? smem + TAB_OFF : nullptr;
  char* bgw = smem + BG_OFF + wave * TN * 8;

  int m0, n0;
  tile_of(xbase + it, m0, n0);
  unsigned par = 0;                                  // ring slot of k-tile 0 of the current tile
  {
    const unsigned voW = (unsigned)(n0 + rowl) * pitchW + dma_lane, voA = (unsigned)(m0 + rowl) * pitchA + dma_lane;
    const unsigned d0 = smem_base + (unsigned)wave * 1024u;
    asm volatile(GVL_A4_DMA_TILE_ASM : : "v"(voW), "v"(voA), "s"(rsW), "s"(rsA), "s"(stepW), "s"(stepA), "s"(d0) : "memory", "scc", GVL_A4_CLOBBER_SGPRS, "v236", "v237");
  }
  bool first = true;
  unsigned long long t_begin = 0, t_loop = 0, t_epi = 0, n_tiles = 0;   // anatomy probe (a.dbg; LAB builds only: GVL_GEMM_TIMING)
  if (a.dbg) t_begin = __builtin_readcyclecounter();
  for (; it < xcnt; it += wpx) {
    const bool has_next = it + wpx < xcnt;
    int m0n = 0, n0n = 0;
    if (has_next) tile_of(xbase + it + wpx, m0n, n0n);
    const unsigned voW = (unsigned)(n0 + rowl) * pitchW + dma_lane, voA = (unsigned)(m0 + rowl) * pitchA + dma_lane;
    // no next tile: every lane's offset is out of range of the resource -- the pieces are still issued (the loop is one straight stream) and move nothing
    const unsigned voWn = has_next ? (unsigned)(n0n + rowl) * pitchW + dma_lane : 0x80000000u, voAn = has_next ? (unsigned)(m0n + rowl) * pitchA + dma_lane : 0x80000000u;
    const int mw = m0 + wm * TM, nw = n0 + wn * TN;
    const bool dead = nw >= a.N || mw >= a.M;        // the wave still takes its share of the DMA and every barrier
    // operands of the epilogue are requested BEFORE the loop and consumed after it: one memory round trip hidden behind the whole k loop
    constexpr bool PRE_RES = G::has_resid && G::KI <= 8;
    constexpr int PRE = 1 | (PRE_RES ? 2 : 0) | 4;
    u32x4_t rv[G::KI];
    u32x2_t ebv, egv;
    float rsc[MB];
    if (!dead) {
      if constexpr (G::has_bias || G::has_gamma) stg_request_bias<NB, EPI>(a, nw, lane, ebv, egv);
      if constexpr (G::has_rowscale) stg_request_rowscale<MB>(a, mw, lane, rsc);
      if constexpr (PRE_RES) stg_request_resid<MB, NB, EPI>(a, mw, nw, lane, 0, rv);
    }
    if (first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); first = false; }   // the first tile's k-tile 0 was requested just above; later tiles' landed inside the previous loop
    unsigned long long ts0 = 0, ts1 = 0;
    if (a.dbg) ts0 = __builtin_readcyclecounter();
    a4_tile_loop<VAR>(rdW + par * SLOT, rdA + par * SLOT, voW, voA, voWn, voAn, rsW, rsA, stepW, stepA, smem_base + (par ^ 1u) * SLOT + (unsigned)wave * 1024u, nk);
    if (a.dbg) ts1 = __builtin_readcyclecounter();
    const unsigned last_slot = (par + nk - 1u) & 1u;    // the slot of the last k-tile: read out by everybody (barrier of its phase 2) -> staging area
    par = (par + nk) & 1u;
#ifdef GVL_A4_LAB_NOEPI                              // LAB (wrong results, timing only): what a fully hidden epilogue would be worth
    if (false) {
#else
    if (!dead) {
#endif
      if constexpr (G::has_bias || G::has_gamma) stg_store_bias<NB, EPI>(bgw, lane, ebv, egv);
      gemm_epilogue_staged_rows<MB, NB, EPI, SWZ, PRE, GELU_TAB ? 1 : 0>(a, A4AccRow{}, smem + last_slot * SLOT + wave * STG_BYTES, bgw, mw, nw, lane, rv, rsc, NoHook(), tabp);
    }
    if (a.dbg) { t_loop += ts1 - ts0; t_epi += __builtin_readcyclecounter() - ts1; ++n_tiles; }
    m0 = m0n; n0 = n0n;
  }
  if (a.dbg && lane == 0) {
    unsigned long long* d = a.dbg + ((size_t)bid * 4 + wave) * 4;
    d[0] = t_loop; d[1] = t_epi; d[2] = n_tiles; d[3] = __builtin_readcyclecounter() - t_begin;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pieces requested for a tile that does not exist
}

template <int EPI, int VAR>
static int launch_a4(const GemmArgs& a_in, hipStream_t st) {
  constexpr bool TAB = (EPI & 3) == GVL_ACT_GELU;
  constexpr int LDS = 2 * 65536 + 4 * 128 * 8 + (TAB ? GELU_TAB_BYTES : 0);
  static GvlDevOnce once;
  static const int n_cu = [] {
    hipDeviceProp_t p; int d = 0;
    return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? (p.multiProcessorCount & ~7) : 256;
  }();
  auto kern = gemm_a4_kernel<EPI, VAR>;
  if (gvl_set_max_lds(once, (const void*)kern, LDS)) return -3;
  GemmArgs a = a_in;
  const int tiles_m = (a.M - a.m_begin + 255) / 256, tiles_n = (a.N + 255) / 256;
  const int tiles = tiles_m * tiles_n;
  if (a.band <= 0) a.band = GVL_GEMM_BAND;
  const int grid = tiles <= n_cu ? tiles : n_cu;
  static const bool timing = gvl_lab_env("GVL_GEMM_TIMING") != nullptr;                        // anatomy probe (LAB builds; tools/gemm4_lab.py)
  if (timing) {
    GemmArgs b = a;
    const size_t n = (size_t)grid * 4 * 4;
    if (hipMalloc((void**)&b.dbg, n * 8) != hipSuccess) return -3;
    hipMemsetAsync(b.dbg, 0, n * 8, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, b, tiles_m, tiles_n);
    std::vector<unsigned long long> hbuf(n);
    hipStreamSynchronize(st);
    hipMemcpy(hbuf.data(), b.dbg, n * 8, hipMemcpyDeviceToHost);
    hipFree(b.dbg);
    double s[4] = {0, 0, 0, 0};
    for (size_t w = 0; w < n / 4; ++w) for (int k = 0; k < 4; ++k) s[k] += (double)hbuf[w * 4 + k];
    const double nt = s[2] > 0 ? s[2] : 1, nw = (double)(n / 4);
    fprintf(stderr, "[gemm4 timing] EPI %d VAR %d M %d N %d K %d grid %d tiles %d: per tile and wave (s_memtime cycles): loop %.0f (%.0f per k-tile) epilogue %.0f other %.0f ; tiles per wave %.2f\n",
            EPI, VAR, a.M - a.m_begin, a.N, a.K, grid, tiles, s[0] / nt, s[0] / nt / (a.K / 64), s[1] / nt, (s[3] - s[0] - s[1]) / nt, nt / nw);
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, a, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// -2: this (epilogue, geometry) is not served by the 4-wave kernel -- the caller takes the 8-wave one
int gvl_launch_gemm_a4(const GemmArgs& a, int epi, int var, hipStream_t st) {
  if (a.K % BK != 0 || a.K / BK < 3) return -2;
  if (((size_t)a.N + 256) * (size_t)a.ldw * 2 >= (1ull << 32) || ((size_t)a.M + 256) * (size_t)a.lda * 2 >= (1ull << 32)) return -2;   // 32-bit buffer offsets
  switch (epi) {
#define A4_CASE(E) case E: return var == 1 ? launch_a4<E, 1>(a, st) : (var == 2 ? launch_a4<E, 2>(a, st) : launch_a4<E, 0>(a, st));
    A4_CASE(0) A4_CASE(32) A4_CASE(33) A4_CASE(34) A4_CASE(3) A4_CASE(56) A4_CASE(8) A4_CASE(64) A4_CASE(67) A4_CASE(98) A4_CASE(128) A4_CASE(136) A4_CASE(184)
#undef A4_CASE
    default: return -2;
  }
}
