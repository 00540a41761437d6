// gvl_gemm.hip -- bf16 MFMA GEMM for gfx950 (MI355X):  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Replaces every nn.Linear / patch-conv on the hot path (SURVEY.md §2.3 K1,K2,K4,K6,K7,K10,K13,K16,K20):
// models/modeling_clip.py:185,264-266,326,340-342; models/internvideo2.py:587,603,631-634,722;
// models/llava_next_video.py:36-38,51-53; models/modeling_phi3.py:459-464,659-663,770.
//
// Design (CDNA4-first, not a port of any CUDA tiling):
//   * v_mfma_f32_32x32x16_bf16, operands swapped: the MFMA "A" rows are WEIGHT rows (n) and the
//     "B" columns are activation rows (m), so each lane ends up owning 4 CONSECUTIVE output
//     columns of one output row -> 8/16-byte epilogue stores into row-major C.
//   * both operand tiles are K-contiguous 128-byte rows, staged global->LDS with
//     global_load_lds (16 B/lane, no VGPR round trip), double buffered, ONE barrier per K tile.
//   * LDS image is lane-linear (DMA constraint); the bank-conflict-free layout is obtained by
//     permuting the 16-byte chunks on the SOURCE address and applying the same XOR on the
//     ds_read_b128 side: phys_chunk = chunk ^ ((row >> 1) & 7)  (conflict-free for the
//     ds_read_b128 lane groups of MI355X_MICROARCH.md §LDS).
//   * XCD-aware block order: each of the 8 XCDs gets a contiguous range of tiles so that the
//     A row-panel of a tile row stays in that XCD's private L2.
#include <atomic>
#include "gvl_internal.h"
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>

#include "gvl_gemm_epi.h"

// NS = LDS ring slots.  2: k-tile t+1 is requested while t is consumed (compiler-tracked DMA).  3: TWO k-tiles ahead, DMA issued from
// inline asm with the waits counted by hand (`s_waitcnt vmcnt(NI)` leaves the newest tile in flight): for the planner's remainder /
// tail launches (a few hundred small tiles, <= 2 blocks per CU) the k loop is bound by the DMA round trip, not by MFMA issue --
// 96 k-tiles x ~0.95 us on InternVideo2's fc2 tail; the per-element k order, hence every output bit, is unchanged.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAG, int EPI, int STG = 0, int NS = 2>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64) void gemm_bf16_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int NWAVES = WAVES_M * WAVES_N;
  constexpr int NT = NWAVES * 64;
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;  // wave tile
  constexpr int MB = TM / 32, NB = TN / 32;            // 32x32 blocks per wave
  constexpr int ROWS = BM + BN;
  constexpr int NI = ROWS * 8 / NT;                    // global_load_lds instructions / thread / stage
  constexpr int STAGE_BYTES = ROWS * 128;
  static_assert(ROWS * 8 % NT == 0, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- XCD-aware tile id (bijective for any grid size) ------------------------------------------
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  // grouped rasterisation: consecutive tiles walk DOWN a band of GM tile-rows before moving to the next tile
  // column, so the ~32 blocks resident on one XCD form an (8 x 4)-tile patch that re-reads 8 A-panels and
  // 4 W-panels from that XCD's L2 instead of 1 + 32 panels.
  constexpr int GM = 8;
  const int band = GM * tiles_n;
  const int g = vid / band, first_m = g * GM;
  const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int in_band = vid - g * band;
  const int tm = first_m + in_band % gm, tn = in_band / gm;
  const int m0 = a.m_begin + tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- per-thread DMA source pointers (k0 excluded) -------------------------------------------
  const bf16_t* src[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = i * NWAVES + wave;           // 1-KiB chunk = 8 tile rows
    const int row = c * 8 + (lane >> 3);       // row in the combined [W ; A] tile
    const int pc = lane & 7;
    if (c * 8 < BN) {
      const int lc = pc ^ ((row >> 1) & 7);
      int gr = n0 + row; gr = gr < a.N ? gr : a.N - 1;
      src[i] = a.W + (size_t)gr * a.ldw + lc * 8;
    } else {
      const int ra = row - BN;
      const int lc = pc ^ ((ra >> 1) & 7);
      int gr = m0 + ra; gr = gr < a.M ? gr : a.M - 1;
      src[i] = a.A + (size_t)gr * a.lda + lc * 8;
    }
  }

  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      char* dst = smem + buf * STAGE_BYTES + (i * NWAVES + wave) * 1024;  // wave-uniform; the DMA adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + k0),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x16_t acc[NB][MB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets: row (lane&31) of a 32-row block, logical chunk kk*2 + (lane>>5)
  const int l31 = lane & 31, h = lane >> 5;
  const int swz = (l31 >> 1) & 7;
  const int w_row_off = (wn * TN + l31) * 128;
  const int a_row_off = BN * 128 + (wm * TM + l31) * 128;

  const int nk = a.K / BK;
  auto compute = [&](const char* sb, int kk) {
    const int coff = ((kk * 2 + h) ^ swz) << 4;
    bf16x8_t wf[NB], af[MB];
#pragma unroll
    for (int i = 0; i < NB; ++i) wf[i] = *(const bf16x8_t*)(sb + w_row_off + i * 32 * 128 + coff);
#pragma unroll
    for (int j = 0; j < MB; ++j) af[j] = *(const bf16x8_t*)(sb + a_row_off + j * 32 * 128 + coff);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
  };
  if constexpr (NS == 3) {
    const unsigned smem_u = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    auto stage3 = [&](int slot, int k0) {
#pragma unroll
      for (int i = 0; i < NI; ++i) glds16(src[i] + k0, smem_u + slot * STAGE_BYTES + (i * NWAVES + wave) * 1024);
    };
    stage3(0, 0);
    if (nk > 1) stage3(1, BK);
    int slot = 0;
    for (int t = 0; t < nk; ++t) {
      // outstanding DMA groups: tile t and (if it exists) tile t+1 -- vmcnt retires in order, so NI leaves exactly the newer one
      if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                        // tile t is in LDS for everybody; everybody is done reading tile t-1, whose slot is refilled next
      int nslot = slot + 2; nslot = nslot >= 3 ? nslot - 3 : nslot;
      if (t + 2 < nk) stage3(nslot, (t + 2) * BK);
      const char* sb = smem + slot * STAGE_BYTES;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) compute(sb, kk);
      slot = slot == 2 ? 0 : slot + 1;
    }
  } else {
  // 2-slot LDS ring: k-tile t+1 is requested while k-tile t is consumed.  STAG = 1: the waves request it at DIFFERENT k-steps
  // (slot = f(wave)), so that on every SIMD some wave always has MFMAs to issue while another pays the ~8 x 60-cycle
  // global_load_lds issue cost; STAG = 0: everybody at the top of the iteration (baseline, cfg 1).
  const int dma_slot = (((wave >> 2) << 1) + (wave & 1)) & 3;
  stage(0, 0);
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's DMA pieces of tile t have landed
    __syncthreads();  // ... and everybody's; all waves are also done reading the slot that is refilled next
    const bool more = t + 1 < nk;
    if constexpr (STAG == 0) { if (more) stage((t + 1) & 1, (t + 1) * BK); }
    const char* sb = smem + (t & 1) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (STAG != 0) { if (kk == dma_slot && more) stage((t + 1) & 1, (t + 1) * BK); }
      compute(sb, kk);
    }
  }
  }

  if constexpr (STG != 0 && EPI >= 0) {
    constexpr int STG_BYTES = 32 * ((EPI & 4) ? TN * 4 + 16 : TN * 2 + 16);
    static_assert(NWAVES * STG_BYTES <= NS * STAGE_BYTES, "staging does not fit the ring");
    __syncthreads();                               // the other waves may still be reading the last k-tile
    constexpr int BG_OFF = NS * STAGE_BYTES - NWAVES * TN * 8;   // bias/gamma scratch: TN floats each per wave, top of the ring
    static_assert(NWAVES * STG_BYTES <= BG_OFF, "staging overlaps the bias scratch");
    u32x4_t rv[StgGeom<NB, EPI>::KI];
    float rsc[MB];
    gemm_epilogue_staged<MB, NB, EPI>(a, acc, smem + wave * STG_BYTES, smem + BG_OFF + wave * TN * 8, m0 + wm * TM, n0 + wn * TN, lane, rv, rsc);
  } else {
    gemm_epilogue<TM, TN, MB, NB, EPI>(a, acc, m0, n0, wm, wn, l31, h);
  }
}

// =====================================================================================================
// Ping-pong variant (256x256x64, 8 waves): the two waves that share a SIMD run HALF A PHASE APART.  Every k-step
// is split into a LOAD phase (6 ds_read_b128 + a share of the next tile's DMA) and an MFMA phase (8 x 32x32x16),
// separated by workgroup barriers; waves 4-7 start one barrier later than waves 0-3, so at any time one wave per
// SIMD is issuing MFMAs (at raised priority) while its partner fetches operands.  The matrix pipe therefore never
// waits for LDS latency, DMA issue or the tile-boundary restart that the lock-step structure pays every 64 K.
// Barrier schedule (global barrier index b): group 0: loads(s) | b=2s | MFMA(s) | b=2s+1 ; group 1 is shifted by one.
// Hazards: a wave's `s_waitcnt vmcnt(0)` for tile t+1 sits in the LOAD phase of its k-step 3, which is ahead of
// barrier 8t+7 for BOTH groups; the first read of tile t+1 (group 0, step 0) comes after that barrier.  The DMA for
// tile t+1 overwrites the slot of tile t-1, last read before barrier 8t-1; it is issued after that barrier.
// =====================================================================================================
template <int BM, int BN, int EPI, int STG = 0>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int WAVES_M = 4, WAVES_N = 2, NWAVES = 8, NT = 512;
  constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N, MB = TM / 32, NB = TN / 32;
  constexpr int ROWS = BM + BN, NI = ROWS * 8 / NT, STAGE_BYTES = ROWS * 128;
  static_assert(NI % 2 == 0, "DMA pieces are issued in two halves");
  // PERSISTENT: the grid is min(tiles, CUs) and every workgroup walks its XCD's share of the tile list.  With the staged
  // bf16 epilogue (8 KiB / wave = one ring slot) the NEXT tile's first k-tile is requested into slot 0 BEFORE the epilogue,
  // so its HBM/L2 latency is covered by the epilogue's VALU work, LDS transpose and row stores.
  constexpr bool STAGED = STG != 0 && EPI >= 0;
  constexpr bool F32OUT = EPI >= 0 && (EPI & 4);
  constexpr bool SILU = EPI >= 0 && (EPI & 3) == GVL_ACT_SILU_MUL;
  constexpr bool OVERLAP = STAGED && !F32OUT;                     // staging fits ring slot 1
  constexpr int SWZ = (OVERLAP && !SILU) ? 1 : 0;
  constexpr int STG_BYTES = F32OUT ? 32 * (TN * 4 + 16) : (SILU ? 32 * (TN + 16) : (SWZ ? 32 * TN * 2 : 32 * (TN * 2 + 16)));
  static_assert(!OVERLAP || NWAVES * STG_BYTES <= STAGE_BYTES, "staging must fit one ring slot");
  constexpr int PP_BG_OFF = (2 * STAGE_BYTES > NWAVES * STG_BYTES ? 2 * STAGE_BYTES : NWAVES * STG_BYTES);
  constexpr bool GELU_TAB = STAGED && (EPI & 3) == GVL_ACT_GELU;       // Phi table behind the bias scratch
  constexpr int PP_TAB_OFF = PP_BG_OFF + NWAVES * TN * 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x, G = gridDim.x;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int xbase = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;   // this XCD's contiguous run of tile ids
  const int xcnt = q + (xcd < r ? 1 : 0);
  const int wpx = (G + 7 - xcd) >> 3;                                          // workgroups resident on this XCD

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // waves w and w+4 share a SIMD
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int l31 = lane & 31, h = lane >> 5;
  const int swz = (l31 >> 1) & 7;
  const int w_row_off = (wn * TN + l31) * 128;
  const int a_row_off = BN * 128 + (wm * TM + l31) * 128;
  const int nk = a.K / BK;

  // DMA sources: wave-uniform bases (a.W / a.A advanced by k0, in SGPRs) + per-lane 32-bit byte offsets (row, swizzled chunk);
  // pieces 0..NI/2-1 are W rows, the rest A rows.  (The launcher routes operands >= 4 GiB to the lock-step kernel.)
  static_assert(NI / 2 == 4 && BN * 8 == (NI / 2) * NWAVES * 64, "piece halves = operand halves");
  unsigned voff[NI];
  int m0 = 0, n0 = 0;
  auto setup = [&](int vid, int& om0, int& on0) {
    // grouped rasterisation inside the XCD's run: walk DOWN a band of GM tile-rows, then the next tile column
    const int GM = a.band;                        // tile rows per band (launch_pp: 8, or 32 / tiles_n when a whole tile row fits the XCD's 32 workgroups)
    const int band = GM * tiles_n;
    const int g = vid / band, first_m = g * GM;
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_band = vid - g * band;
    const int tm = first_m + in_band % gm, tn = in_band / gm;
    om0 = a.m_begin + tm * BM; on0 = tn * BN;
    int ln = lane;
    asm volatile("" : "+v"(ln));                   // re-derive the lane terms per call: hoisted out of the tile loop they cost
                                                   // VGPRs that spill around the epilogue (a reload = vmcnt(0) in front of the DMA)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = i * NWAVES + wave;
      const int row = c * 8 + (ln >> 3);
      const int pc = ln & 7;
      if (i < NI / 2) {
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = on0 + row; gr = gr < a.N ? gr : a.N - 1;
        voff[i] = ((unsigned)gr * (unsigned)a.ldw + lc * 8) * 2;
      } else {
        const int ra = row - BN;
        const int lc = pc ^ ((ra >> 1) & 7);
        int gr = om0 + ra; gr = gr < a.M ? gr : a.M - 1;
        voff[i] = ((unsigned)gr * (unsigned)a.lda + lc * 8) * 2;
      }
    }
  };
  auto stage_half = [&](int buf, int k0, int half) {
    const bf16_t* sb = (half == 0 ? a.W : a.A) + k0;
    const unsigned d0 = smem_base + buf * STAGE_BYTES + (half * (NI / 2) * NWAVES + wave) * 1024;
    if (half == 0) glds16x4(sb, voff[0], voff[1], voff[2], voff[3], d0, NWAVES * 1024);
    else glds16x4(sb, voff[4], voff[5], voff[6], voff[7], d0, NWAVES * 1024);
  };
  // (3 + 3 + 2 pieces over the LOAD phases 0/1/2 instead of 4 + 4 was measured 6 % slower at 8192^3: the later pieces land late.)
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

  // The tile PROTOCOL every wave of the block keeps, live or dead (one definition for both paths -- a change to the barrier count of the live loop
  // cannot leave the dead-wave path behind):  tile_open | per k-tile t, phase ph: {phase_dma, barrier, [8 MFMAs], barrier} | tile_close
  auto tile_open = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_BARRIER();                                   // k-tile 0 is in LDS for everybody; everybody is done with the staging slot
    if (grp == 1) PP_BARRIER();                     // half-phase offset of the second wave group
  };
  // this wave's share of the next k-tile's DMA (phases 0 / 1) and the wait for it (phase 3; `wait_last`: also behind the last k-tile)
  auto phase_dma = [&](int t, int ph, bool more, bool wait_last) {
    if (ph < 2 && more) stage_half((t + 1) & 1, (t + 1) * BK, ph);
    if (ph == 3 && (more || wait_last)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto tile_close = [&]() { if (grp == 0) PP_BARRIER(); };   // pairs with the extra barrier group 1 took in tile_open

  int it = bid >> 3;
  if (it >= xcnt) return;
  if constexpr (GELU_TAB) {                          // once per (persistent) workgroup; first read is many barriers away
    for (int o = tid * 16; o < GELU_TAB_BYTES; o += NT * 16) *(u32x4_t*)(smem + PP_TAB_OFF + o) = *(const u32x4_t*)((const char*)a.act_table + o);
  }
  const char* tabp = GELU_TAB ? smem + PP_TAB_OFF : nullptr;
  setup(xbase + it, m0, n0);
  stage_half(0, 0, 0); stage_half(0, 0, 1);
  for (; it < xcnt; it += wpx) {
    // DEAD wave: every output column (N = 1408 = 5.5 tile columns: the wn = 1 waves of the last column) or row of its 64 x 128 sub-tile lies outside the
    // matrix.  It used to run the full MFMA + fragment-read stream on clamped (repeated) operand rows -- 8.3 % of InternVideo2's proj / fc2 MFMA work, results
    // dropped by the bounds check.  The board runs this kernel AT its power cap, so work that produces nothing is clock taken from the
    // CUs that do: a dead wave now only keeps its place in the protocol -- its share of the DMA, the waits and all 8 barriers of every k-tile -- and
    // issues no MFMA, no LDS read, no epilogue.  (wn is the same for the two waves of a SIMD: SIMDs 1 and 3 idle through such a tile.)  Kept as ONE early
    // block of the tile loop so that the live path's register allocation is untouched (the bias + gamma + residual epilogue sits 11 VGPRs under the limit).
    if ((n0 + wn * TN >= a.N) || (m0 + wm * TM >= a.M)) {
      tile_open();
      for (int t = 0; t < nk; ++t) {
        const bool more = t + 1 < nk;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
          phase_dma(t, ph, more, false);
          PP_BARRIER();
          PP_BARRIER();
        }
      }
      tile_close();
      if (it + wpx < xcnt) {                        // the next tile's first k-tile still needs this wave's DMA pieces
        if constexpr (STAGED && !OVERLAP) __syncthreads();   // the live waves' staging slices overlap ring slot 0
        setup(xbase + it + wpx, m0, n0); stage_half(0, 0, 0); stage_half(0, 0, 1);
      }
      continue;
    }
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (a.dbg) ts0 = __builtin_readcyclecounter();
    f32x16_t acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    tile_open();
    if (a.dbg) ts1 = __builtin_readcyclecounter();

    // 4 phases per k-tile (one 16-wide k-step each): LOAD = 6 ds_read_b128 (+ a half of the next k-tile's DMA in phases
    // 0/1, vmcnt(0) in phase 3), MFMA = 8 x 32x32x16 at raised priority.
    // (Waiting for the DMA at the END of the last MFMA phase instead is a RACE: group 1 would retire its pieces after
    //  barrier 8t+8 while group 0 already reads tile t+1 after barrier 8t+7 -- measured wrong results.)
    // operands of the staged epilogue are requested INSIDE the main loop: bias/gamma with the first k-tile (parked in the wave's
    // LDS scratch after that tile's wait), residual block 0 with the last k-tile (it stays in flight across the tail)
    constexpr int EPI_G = STAGED ? EPI : 0;
    using G = StgGeom<NB, EPI_G>;
    constexpr bool PRE_RES = G::has_resid && G::KI <= 8;           // 32 VGPRs across the last k-tile; the f32 residual (64) would spill
    constexpr int PRE = 1 | (PRE_RES ? 2 : 0) | 4;
    u32x4_t rv[G::KI];
    u32x2_t ebv, egv;
    float rsc[MB];                                  // fused RMSNorm, consumer side: the row scale of this lane's MB rows, requested with the first k-tile
    char* bgw = smem + PP_BG_OFF + wave * TN * 8;   // bias / gamma scratch of this wave (above the ring and the staging slices)
    for (int t = 0; t < nk; ++t) {
      const char* sb = smem + (t & 1) * STAGE_BYTES;
      const bool more = t + 1 < nk;
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        if constexpr (STAGED) {
          if (ph == 0 && t == 0 && (G::has_bias || G::has_gamma)) stg_request_bias<NB, EPI_G>(a, n0 + wn * TN, lane, ebv, egv);
          if (ph == 0 && t == 0 && G::has_rowscale) stg_request_rowscale<MB>(a, m0 + wm * TM, lane, rsc);
          if (ph == 0 && !more && PRE_RES) stg_request_resid<MB, NB, EPI_G>(a, m0 + wm * TM, n0 + wn * TN, lane, 0, rv);
        }
        bf16x8_t wf[NB], af[MB];
        const int coff = ((ph * 2 + h) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) wf[i] = *(const bf16x8_t*)(sb + w_row_off + i * 32 * 128 + coff);
#pragma unroll
        for (int j = 0; j < MB; ++j) af[j] = *(const bf16x8_t*)(sb + a_row_off + j * 32 * 128 + coff);
        phase_dma(t, ph, more, !STAGED);            // last k-tile: no DMA pending, the (staged epilogue's) residual request stays in flight
        if constexpr (STAGED) {
          if (ph == 3 && t == 0 && (G::has_bias || G::has_gamma)) stg_store_bias<NB, EPI_G>(bgw, lane, ebv, egv);
        }
        PP_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int j = 0; j < MB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_BARRIER();
      }
    }
    tile_close();
    // every LDS read of the ring is complete (both groups are past their last MFMA phase)
    if (a.dbg) ts2 = __builtin_readcyclecounter();
    const int em0 = m0, en0 = n0;
    const bool has_next = it + wpx < xcnt;
    if constexpr (OVERLAP) {
      gemm_epilogue_staged<MB, NB, EPI, SWZ, PRE, GELU_TAB ? 1 : 0>(a, acc, smem + STAGE_BYTES + wave * STG_BYTES, bgw, em0 + wm * TM, en0 + wn * TN, lane, rv, rsc, [&]() {
        if (has_next) { setup(xbase + it + wpx, m0, n0); stage_half(0, 0, 0); stage_half(0, 0, 1); }
      }, tabp);
      if (has_next) {
        // the 16 DMA source pointers are RE-derived here instead of staying live across the epilogue (they would spill)
        int vnext = xbase + it + wpx;
        asm volatile("" : "+s"(vnext));
        setup(vnext, m0, n0);
      }
    } else {
      if constexpr (STAGED) gemm_epilogue_staged<MB, NB, EPI, 0, PRE>(a, acc, smem + wave * STG_BYTES, bgw, em0 + wm * TM, en0 + wn * TN, lane, rv, rsc);
      else gemm_epilogue<TM, TN, MB, NB, EPI>(a, acc, em0, en0, wm, wn, l31, h);
      if (has_next) {
        if constexpr (STAGED && !OVERLAP) __syncthreads();      // staging slices overlap ring slot 0
        setup(xbase + it + wpx, m0, n0); stage_half(0, 0, 0); stage_half(0, 0, 1);
      }
    }
    if (a.dbg && lane == 0) {
      unsigned long long* d = a.dbg + ((size_t)bid * 8 + wave) * 4;
      d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_readcyclecounter();
    }
  }
#undef PP_BARRIER
}

// A/B only (gvl_debug_set("gemm_band" / "gemm_a4")): PROCESS-wide by design (documented in gvl.h), result-neutral; atomics because contexts of several host threads may
// launch while one of them flips a knob
static std::atomic<int> g_band_override{0};
void gvl_gemm_set_band(int v) { g_band_override.store(v, std::memory_order_relaxed); }
// Which launches of the 256 x 256 kernel take a 4-wave / AGPR form (gvl_gemm4.hip, gvl_gemm4p.hip) instead of the 8-wave ping-pong: see big_form_preferred and the
// dispatch in gvl_launch_gemm.  Bit-identical whatever the value (gvl_debug_set("gemm_a4"); process-wide).
static std::atomic<int> g_a4_mode{1};
void gvl_gemm_set_a4(int v) { g_a4_mode.store(v, std::memory_order_relaxed); }
// Which form of the 256 x 256 kernel an (epilogue) takes when the library chooses (cfg 80), from same-box interleaved runs on the model's shapes
// (profiles/r06_gemm4_lab_model.txt; tools/gemm4_lab.py 82,86,88 model):
//   pipelined 4-wave (88): the epilogues with real VALU / LDS work behind the bf16 rounding -- erf-GELU (98: +5 % over the 8-wave kernel, +10 % over the plain 4-wave
//                          one), SwiGLU (67: +4.5 %), residual + row statistics (136: +3 ... +7 %; 184: +1 ... +5 %) -- that work rides in the next tile's MFMA gaps;
//   plain 4-wave (86):     the store-only epilogues (64, 0, ...: +2 ... +4 %; pipelining them buys nothing: what remains exposed either way is the accumulator drain);
//                          bias alone (32: +2 %, its slice read once per tile into dead fragment registers);
//   8-wave ping-pong (82): CLIP's quick-GELU (33: the 4-wave forms lose 1 ... 2 % there), the f32-output epilogues and everything the 4-wave kernels do not serve.
static int big_form_preferred(int epi) {
  switch (epi) {
    case 98: case 67: case 136: case 184: case 32: return 88;
    case 64: case 0: case 3: case 8: case 128: return 86;
    default: return 82;
  }
}
// Phi(x) table of the current device (built once per device; blocking upload on first use, outside any timed region after warmup)
static const float* gelu_table_device() {
  static const float* tabs[64] = {nullptr};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  if (!tabs[d]) {
    std::vector<float> h(GELU_TAB_BYTES / 4, 0.f);
    for (int sgn = 0; sgn < 2; ++sgn)
      for (int k = 0; k < GELU_NE; ++k) {
        const unsigned bits = ((unsigned)(GELU_LO + k) | (sgn ? 0x8000u : 0u)) << 16;
        float x; memcpy(&x, &bits, 4);
        h[sgn * (GELU_NEG_OFF / 4) + k] = (float)(0.5 * std::erfc(-(double)x * 0.70710678118654752440));
      }
    // every x <= -5.5 clamps onto the last negative entry: Phi = 0 there, as in the reference's fp32 arithmetic (1 + erf(x / sqrt 2) is exactly 0
    // below x ~ -5.4) -- with Phi(-5.5) = 1.9e-8 instead the product x * Phi would GROW with |x| for huge negative inputs
    h[GELU_NEG_OFF / 4 + GELU_NE - 1] = 0.f;
    float* dptr = nullptr;
    if (hipMalloc((void**)&dptr, GELU_TAB_BYTES) != hipSuccess) return nullptr;
    if (hipMemcpy(dptr, h.data(), GELU_TAB_BYTES, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    tabs[d] = dptr;
  }
  return tabs[d];
}

template <int EPI, int STG = 0>
static int launch_pp(const GemmArgs& a_in, hipStream_t st) {
  constexpr int BM = 256, BN = 256, RING = 2 * (BM + BN) * 128;
  constexpr int STGB = (STG && EPI >= 0 && (EPI & 4)) ? 8 * 32 * (128 * 4 + 16) : 0;    // f32 staging: 132 KiB
  constexpr bool TAB = STG && EPI >= 0 && (EPI & 3) == GVL_ACT_GELU;
  constexpr int LDS = (RING > STGB ? RING : STGB) + (STG ? 8 * 128 * 8 : 0) + (TAB ? GELU_TAB_BYTES : 0);   // + bias/gamma scratch + Phi table
  static GvlDevOnce once;
  static const int n_cu = [] {                      // the devices of a node are alike: asked once
    hipDeviceProp_t p; int d = 0;
    return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? (p.multiProcessorCount & ~7) : 256;
  }();
  auto kern = gemm_pp_kernel<BM, BN, EPI, STG>;
  if (gvl_set_max_lds(once, (const void*)kern, LDS)) return -3;
  GemmArgs a = a_in;
  const int tiles_m = (a.M - a.m_begin + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int tiles = tiles_m * tiles_n;
  // Rasterisation band: the 32 workgroups of an XCD walk a band of GM tile rows column by column, k-tile by k-tile in lock-step (an 8 x 4 patch of tiles).
  // MEASURED AND NOT ADOPTED (round 5, profiles/r05_ab_gemm_band.json + the PMC passes): GM = 32 / tile columns for narrow matrices (N = 1408: 5 rows x 6
  // columns, so that an A slice enters the XCD's L2 once) -- +0.4 % clips/s, inside the run-to-run spread, while the family's fabric-side traffic ROSE from
  // 193 to 231 GB per clip (the W panel is re-read per band, and there are 154 bands instead of 96).  8 rows stay; gvl_debug_set("gemm_band") varies it.
  if (a.band <= 0) a.band = GVL_GEMM_BAND;
  if (const int bo = g_band_override.load(std::memory_order_relaxed); bo > 0) a.band = bo;
  static const bool no_persist = gvl_lab_env("GVL_GEMM_NO_PERSIST") != nullptr;                // A/B only
  const int grid = (tiles <= n_cu || no_persist) ? tiles : n_cu;
  static const bool timing = gvl_lab_env("GVL_GEMM_TIMING") != nullptr;                        // anatomy probe (tools/gemm_one.py)
  if (timing) {
    GemmArgs b = a;
    const size_t n = (size_t)grid * 8 * 4;
    if (hipMalloc((void**)&b.dbg, n * 8) != hipSuccess) return -3;
    hipMemsetAsync(b.dbg, 0, n * 8, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, b, tiles_m, tiles_n);
    std::vector<unsigned long long> hbuf(n);
    hipStreamSynchronize(st);
    hipMemcpy(hbuf.data(), b.dbg, n * 8, hipMemcpyDeviceToHost);
    hipFree(b.dbg);
    double s[3] = {0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < 8; ++w) {
      const unsigned long long* d = &hbuf[((size_t)g * 8 + w) * 4];
      for (int k = 0; k < 3; ++k) s[k] += (double)(d[k + 1] - d[k]);
      if (d[0] < tmin) tmin = d[0];
      if (d[3] > tmax) tmax = d[3];
    }
    const double nw = (double)grid * 8;
    fprintf(stderr, "[gemm timing] EPI %d M %d N %d K %d grid %d tiles %d: last tile per wave (cycles of s_memtime): prologue %.0f mainloop %.0f epilogue %.0f ; span of last tiles %llu\n",
            EPI, a.M - a.m_begin, a.N, a.K, grid, tiles, s[0] / nw, s[1] / nw, s[2] / nw, tmax - tmin);
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, a, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// (A second ping-pong variant with a 4-slot ring of HALF k-tiles -- 64-byte LDS rows, 2 DMA pieces per phase, counted
//  vmcnt(8) -- was built and measured 10-15 % SLOWER than gemm_pp_kernel on every hot-path shape (64-byte DMA rows fetch
//  each 128-byte line twice); it was removed.  See DESIGN.md §3.1 and profiles/r01_gemm_microbench_pp.txt.)

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAG, int EPI = -1, int STG = 0, int NS = 2>
static int launch_cfg(const GemmArgs& a, hipStream_t st) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int LDS = NS * (BM + BN) * 128;
  static GvlDevOnce once;
  auto kern = gemm_bf16_kernel<BM, BN, WAVES_M, WAVES_N, STAG, EPI, STG, NS>;
  if (gvl_set_max_lds(once, (const void*)kern, LDS)) return -3;
  const int tiles_m = (a.M - a.m_begin + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(NT), LDS, st, a, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

double gvl_gemm_flops(const GemmArgs& a) { return 2.0 * a.M * (double)a.N * a.K; }

int gvl_launch_gemm(const GemmArgs& a_in, hipStream_t st) {
  GemmArgs a = a_in;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return -1;
  if (a.ldw == 0) a.ldw = a.K;
  if (a.act == GVL_ACT_GELU && !a.act_table) { a.act_table = gelu_table_device(); if (!a.act_table) return -3; }   // every erf-GELU epilogue reads Phi from the table
  if (a.K % BK != 0 || a.N % 4 != 0 || a.lda % 8 != 0 || a.ldw % 8 != 0 || a.ldw < a.K) return -1;   // K padded to 64 by the packer; 16-byte rows
  if (a.act == GVL_ACT_SILU_MUL && (a.out_f32 || a.resid || a.gamma)) return -1;
  int cfg = a.tile_cfg;
  static const int env_cfg = [] { const char* e = gvl_lab_env("GVL_GEMM_CFG"); return e ? atoi(e) : 0; }();   // experiments only
  if (cfg == 0 && env_cfg) cfg = env_cfg;
  if (cfg == 0) {
    // measured on MI355X (tools/gemm_bench.py, profiles/r01_gemm_microbench*.txt):
    //  * cfg 82 = 256x256 ping-pong kernel: best whenever K is long enough to amortise its un-overlapped
    //    prologue/epilogue (one block / CU) and there are enough tiles;
    //  * cfg 21 = 128x128, 2 blocks / CU: short K or few tiles (CLIP out / fc2: 112 tiles).
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    cfg = (a.K >= 1024 && t256 >= 128) ? 80 : 21;   // CLIP qkv / fc1 (K = 1024, 336 / 448 tiles): 82 measured +8...17 % over 21
    // short K but thousands of tiles (the three-pass patch GEMM at the bench's M: K = 640, 864 / 4 608 tiles): 82 measured 76.6 vs 83.2 us (CLIP) and 454.6 vs
    // 477.7 us (InternVideo2, bias) -- profiles/r05_patch_gemm_floor.txt
    if (cfg == 21 && a.K >= 512 && t256 >= 512) cfg = 80;
  }
  if ((cfg == 80 || cfg == 82 || (cfg >= 84 && cfg <= 87)) && ((size_t)a.N * a.ldw * 2 >= (1ull << 32) || (size_t)a.M * a.lda * 2 >= (1ull << 32))) cfg = 21;   // 32-bit DMA offsets
  const int epi = (a.act & 3) | ((a.out_f32 ? 1 : 0) << 2) | ((a.resid ? 1 : 0) << 3) | ((a.gamma ? 1 : 0) << 4) | ((a.bias ? 1 : 0) << 5) |
                  ((a.rowscale ? 1 : 0) << 6) | ((a.rowsq ? 1 : 0) << 7);
  if (a.rowsq && (a.N % 64 != 0 || a.out_f32 || a.act == GVL_ACT_SILU_MUL || a.rowsq_ld < a.N / 64 || a.grp_rows)) return -1;
  if (cfg == 80 && a.tile_cfg == 0 && env_cfg == 0 && a.m_begin == 0) {
    // Wave-quantisation planner.  The persistent 256x256 kernel runs one block per CU, so a launch costs ceil(tiles / CUs)
    // tile times, and a partial last tile column (N = 1408 = 5.5 x 256) wastes half of its MFMA work.  Candidate plans, costed
    // in units of one 256x256 tile time (the small kernel: see small_unit below):
    //   W  whole GEMM on the big kernel;
    //   M  whole rounds of tile ROWS on the big kernel, the remaining rows on the small kernel;
    //   N  the full 256-wide tile columns through W or M, the N % 256 tail columns on the small kernel.
    // e.g. InternVideo2 proj/fc2 (M = 24588, N = 1408): W = 3, M = 3.0, N = 2.5 (485 big tiles in 2 rounds + 193 small).
    static const int n_cu = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
    // cost of up to one CU-count of small tiles, in big-tile times.  Half the FLOPs at ~0.76x the rate would be 0.33, but an
    // under-filled small launch runs its lone blocks far below that rate: same-box A/B of the whole bench, 0.33 / 0.45-0.75 / 0.90
    // -> GEMM time 73.4 / 73.0 / 75.7 ms per clip (GVL_GEMM_SMALLCOST = percent, experiments only)
    static const double small_unit = [] { const char* e = gvl_lab_env("GVL_GEMM_SMALLCOST"); return e ? atoi(e) / 100.0 : 0.5; }();
    auto small_cost = [&](long t) { const long halves = (t + n_cu - 1) / n_cu; return t > 0 ? small_unit * (double)halves : 0.0; };
    struct Plan { double cost; int big_rows; };   // big_rows = tile rows given to the big kernel (all of them: no M split)
    auto plan_mw = [&](int M, int N) {            // best of W and M for an [M, N] problem
      const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
      const long tiles = (long)tiles_m * tiles_n, rounds = tiles / n_cu;
      Plan best{(double)((tiles + n_cu - 1) / n_cu), tiles_m};
      if (rounds >= 1 && tiles % n_cu != 0) {
        const int br = (int)((rounds * n_cu) / tiles_n);
        if (br >= 1 && br < tiles_m) {
          const double c = (double)rounds + small_cost((long)((M - br * 256 + 127) / 128) * ((N + 127) / 128));
          if (c < best.cost - 0.1) best = Plan{c, br};
        }
      }
      return best;
    };
    const Plan whole = plan_mw(a.M, a.N);
    int n_big = a.N;
    Plan chosen = whole;
    if (a.N % 256 != 0 && a.N > 256 && a.act != GVL_ACT_SILU_MUL) {
      const int nb = (a.N / 256) * 256;
      const Plan p = plan_mw(a.M, nb);
      const double c = p.cost + small_cost((long)((a.M + 127) / 128) * ((a.N - nb + 127) / 128));
      if (c < whole.cost - 0.1) { chosen = Plan{c, p.big_rows}; n_big = nb; }
    }
    const int tiles_m = (a.M + 255) / 256;
    if (n_big != a.N || chosen.big_rows < tiles_m) {
      const int es_ = a.out_f32 ? 4 : 2;
      auto cols = [&](const GemmArgs& g, int n0, int n1) {       // sub-problem on output columns [n0, n1): pointer offsets only
        GemmArgs r = g;
        r.W = g.W + (size_t)n0 * g.ldw; r.N = n1 - n0;
        r.C = (char*)g.C + (size_t)n0 * es_;
        if (g.resid) r.resid = (const char*)g.resid + (size_t)n0 * es_;
        if (g.bias) r.bias = g.bias + n0;
        if (g.gamma) r.gamma = g.gamma + n0;
        if (g.rowsq) r.rowsq = g.rowsq + n0 / 64;    // n0 is a multiple of 256: whole 64-column blocks
        return r;
      };
      GemmArgs left = cols(a, 0, n_big);
      int rc = 0;
      if (chosen.big_rows < tiles_m) {
        GemmArgs big = left; big.M = chosen.big_rows * 256; big.tile_cfg = 80;
        GemmArgs rest = left; rest.m_begin = chosen.big_rows * 256; rest.tile_cfg = 21;
        rc = gvl_launch_gemm(big, st);
        if (!rc) rc = gvl_launch_gemm(rest, st);
      } else {
        left.tile_cfg = 80;
        rc = gvl_launch_gemm(left, st);
      }
      if (!rc && n_big != a.N) { GemmArgs tail = cols(a, n_big, a.N); tail.tile_cfg = 21; rc = gvl_launch_gemm(tail, st); }
      return rc;
    }
  }
  const int es = a.out_f32 ? 4 : 2;
  const bool stg_ok = a.N % 16 == 0 && a.grp_rows == 0 && ((size_t)a.ldc * es) % 16 == 0 && ((uintptr_t)a.C & 15) == 0 &&
                      (!a.resid || (((size_t)a.ldr * es) % 16 == 0 && ((uintptr_t)a.resid & 15) == 0));
  if ((a.rowscale || a.rowsq) && (!stg_ok || cfg == 1 || cfg == 85)) return -1;   // the fused-RMSNorm epilogues exist in the staged (whole-row) form only
  if (cfg == 21 && a.tile_cfg == 21 && env_cfg == 0) {   // planner remainder / tail launches only
    static const int small64 = [] { const char* e = gvl_lab_env("GVL_GEMM_SMALL64"); return e ? atoi(e) : 3; }();   // 0 = off (A/B); measured -0.6 ms of GEMM time per clip
    static const int n_cu2 = [] { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
    const long t128 = (long)((a.M - a.m_begin + 127) / 128) * ((a.N + 127) / 128);
    if (small64 && t128 * 2 <= (long)small64 * n_cu2) cfg = 22;
  }
  // cfg 80 = "the 256 x 256 kernel, the library's choice of form" (what the automatic selection and the planner ask for).  Forms, all bit-identical:
  // 82 the 8-wave ping-pong kernel; 84 / 86 / 87 the 4-wave kernel (gvl_gemm4.hip, loop schedule 0 / 1 / 2); 88 the 4-wave kernel with the epilogue
  // pipelined into the next tile's main loop (gvl_gemm4p.hip).  An explicit 84 ... 88 that does not serve the (epilogue, geometry) falls back towards 82.
  if (cfg == 80 || (cfg >= 84 && cfg <= 88)) {
    // gemm_a4 (gvl_debug_set): 0 = always the 8-wave kernel, 1 (default) = big_form_preferred, 2 = the plain 4-wave kernel wherever it serves, 3 = the pipelined one
    // wherever it serves (then the plain one)
    int form = cfg;
    const int a4m = g_a4_mode.load(std::memory_order_relaxed);
    if (cfg == 80) form = a4m == 0 ? 82 : (a4m == 1 ? big_form_preferred(epi) : (a4m == 2 ? 86 : 88));
    GemmArgs b = a;
    if (b.band <= 0) b.band = GVL_GEMM_BAND;
    if (const int bo = g_band_override.load(std::memory_order_relaxed); bo > 0) b.band = bo;
    if (stg_ok && form == 88) {
      const int rc4 = gvl_launch_gemm_a4p(b, epi, st);
      if (rc4 != -2) return rc4;
      form = 86;
    }
    if (stg_ok && form >= 84 && form <= 87) {
      const int rc4 = gvl_launch_gemm_a4(b, epi, form == 84 ? 0 : (form == 87 ? 2 : 1), st);
      if (rc4 != -2) return rc4;
    }
    cfg = 82;
  }
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 2, 2, 0>(a, st);            // plain lock-step baseline (tests / A-B)
    // cfg 21 / 82 run the LDS-staged whole-row epilogue (compile-time specialised per fused-epilogue code) whenever the
    // output rows are 16-byte aligned; anything else takes the generic per-lane epilogue (EPI = -1).
    case 21: {
      if (stg_ok) switch (epi) {
#define S_CASE(E) case E: return launch_cfg<128, 128, 2, 2, 1, E, 1>(a, st);
        S_CASE(0) S_CASE(32) S_CASE(33) S_CASE(34) S_CASE(3) S_CASE(44) S_CASE(56) S_CASE(8) S_CASE(4) S_CASE(36)
        S_CASE(64) S_CASE(67) S_CASE(98) S_CASE(128) S_CASE(136) S_CASE(184)     // fused RMSNorm: +64 row scale (consumer), +128 row sums of squares (producer)
#undef S_CASE
        default: break;
      }
      if (a.rowscale || a.rowsq) return -1;
      return launch_cfg<128, 128, 2, 2, 1, -1>(a, st);
    }
    // cfg 22 = 64x128 tiles, same kernel (wave tile 32x64): twice the blocks of cfg 21 for launches that would leave most CUs with
    // ONE 128x128 block (remainder rows / tail columns of the planner: 193-264 tiles on 512 slots).  Same k-order per output
    // element as every other cfg, so the results are bit-identical.
    case 22: {
      // 3-slot ring (two k-tiles of DMA in flight, 72 KB: still 2 blocks per CU); the 2-slot form measured 3-15 % slower on these launches (round 2)
      if (stg_ok) switch (epi) {
#define S_CASE(E) case E: return launch_cfg<64, 128, 2, 2, 1, E, 1, 3>(a, st);
        S_CASE(0) S_CASE(32) S_CASE(33) S_CASE(34) S_CASE(3) S_CASE(44) S_CASE(56) S_CASE(8) S_CASE(4) S_CASE(36)
        S_CASE(64) S_CASE(67) S_CASE(98) S_CASE(128) S_CASE(136) S_CASE(184)
#undef S_CASE
        default: break;
      }
      if (a.rowscale || a.rowsq) return -1;
      return launch_cfg<64, 128, 2, 2, 1, -1>(a, st);
    }
    case 82: {
      if (stg_ok) switch (epi) {
#define PP_CASE(E) case E: return launch_pp<E, 1>(a, st);
        PP_CASE(0) PP_CASE(32) PP_CASE(33) PP_CASE(34) PP_CASE(3) PP_CASE(44) PP_CASE(56) PP_CASE(8) PP_CASE(4) PP_CASE(36)
        PP_CASE(64) PP_CASE(67) PP_CASE(98) PP_CASE(128) PP_CASE(136) PP_CASE(184)
#undef PP_CASE
        default: break;
      }
      if (a.rowscale || a.rowsq) return -1;
      return launch_pp<-1>(a, st);
    }
    case 85: return launch_pp<-1>(a, st);                   // ping-pong with the per-lane epilogue (A/B only)
    default: return -1;
  }
}
