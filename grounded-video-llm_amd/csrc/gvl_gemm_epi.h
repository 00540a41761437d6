// gvl_gemm_epi.h -- the fused GEMM epilogues shared by the MFMA GEMM kernels of gvl_gemm.hip (8-wave ping-pong, lock-step) and gvl_gemm4.hip (4-wave,
// accumulators in AGPRs): per-lane generic form and the LDS-staged whole-row form (bias, quick-GELU, erf-GELU by table, SwiGLU, LayerScale, residual,
// fused-RMSNorm row scale / row sums of squares).  Arithmetic and rounding points are identical whichever kernel calls them.
#pragma once
#include "gvl_internal.h"

#define BK 64  // K tile (bf16 elements) == one 128-byte LDS row

// Epilogue.  EPI >= 0 encodes the flag set at compile time (act | out_f32<<2 | resid<<3 | gamma<<4 | bias<<5) so the
// hot instantiations carry no per-element branching; EPI = -1 reads the flags at run time (tests, rare shapes).
// A lane owns row m = ..+(lane&31) and, per 32x32 block, four groups of 4 consecutive columns n = ..+8*b+4*h.
// ---- erf-GELU by table --------------------------------------------------------------------------------------------------
// The GELU input is ALREADY rounded to bf16 (reference: nn.GELU on a bf16 tensor), so gelu(x) = x * Phi(x) needs Phi only at
// bf16 points.  Phi(x) in f32 is tabulated for 2^-12 <= |x| <= 5.5 (1841 bf16 values per sign, 2 x 8 KiB, built on the host
// in double precision); below 2^-12 Phi is 0.5 to 2e-4 relative, above 5.5 it is 1 (resp. 0: the reference's f32 1 + erf is exactly 0 there) -- both ends clamp.
// 7 full-rate VALU + one ds_read_b32 per element instead of ~16 issue slots with v_rcp + v_exp: the epilogue of
// InternVideo2's fc1 tile drops from ~16 k to ~7 k cycles (tools/gemm_one.py, GVL_GEMM_TIMING=1).
constexpr int GELU_LO = 0x3980, GELU_HI = 0x40B0, GELU_NE = GELU_HI - GELU_LO + 1;
constexpr int GELU_NEG_OFF = 0x2000;               // byte offset of the negative half = sign << 13: no select needed
constexpr int GELU_TAB_BYTES = 2 * GELU_NEG_OFF;
static_assert(GELU_NE * 4 <= GELU_NEG_OFF, "positive half overlaps the negative half");
// The same table read from GLOBAL memory (L1/L2 resident, 16 KiB): used by the 128x128 kernel and the generic epilogue, so that a
// row gets the same value whichever kernel the launch planner hands it to (batch-invariance is asserted at full size).
__device__ __forceinline__ float gelu_tab_global(const float* __restrict__ tab, float v) {
  const unsigned bits = __float_as_uint(rbf(v)) >> 16;
  int key = (int)(bits & 0x7fffu);
  key = key < GELU_LO ? GELU_LO : (key > GELU_HI ? GELU_HI : key);
  const int idx = (key - GELU_LO) + ((bits >> 15) ? GELU_NEG_OFF / 4 : 0);
  return __uint_as_float(bits << 16) * tab[idx];
}

template <int TM, int TN, int MB, int NB, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16_t (&acc)[NB][MB], int m0, int n0, int wm, int wn, int l31, int h) {
  const int act = EPI >= 0 ? (EPI & 3) : a.act;
  const bool out_f32 = EPI >= 0 ? ((EPI >> 2) & 1) : (a.out_f32 != 0);
  const bool has_resid = EPI >= 0 ? ((EPI >> 3) & 1) : (a.resid != nullptr);
  const bool has_gamma = EPI >= 0 ? ((EPI >> 4) & 1) : (a.gamma != nullptr);
  const bool has_bias = EPI >= 0 ? ((EPI >> 5) & 1) : (a.bias != nullptr);
  const int nbase = n0 + wn * TN + 4 * h;
#pragma unroll
  for (int j = 0; j < MB; ++j) {
    const int m = m0 + wm * TM + j * 32 + l31;
    if (m >= a.M) continue;
    const size_t orow = a.grp_rows ? (size_t)(m / a.grp_rows) * a.grp_stride + (m % a.grp_rows) + a.row_off : (size_t)m + a.row_off;
    char* crow = (char*)a.C + orow * a.ldc * (out_f32 ? 4 : 2);
    const char* rrow = has_resid ? (const char*)a.resid + orow * a.ldr * (out_f32 ? 4 : 2) : nullptr;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int n = nbase + i * 32 + 8 * b;
        if (n >= a.N) continue;
        float v[4] = {acc[i][j][4 * b + 0], acc[i][j][4 * b + 1], acc[i][j][4 * b + 2], acc[i][j][4 * b + 3]};
        if (has_bias) {
          const f32x4_t bv = *(const f32x4_t*)(a.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
        }
        if (act == GVL_ACT_SILU_MUL) {
          // interleaved (gate, up) pairs -> 2 outputs at column n/2.  reference: up * silu(gate), each op in bf16
          float o2[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float g = rbf(v[2 * e]), u = rbf(v[2 * e + 1]);
            o2[e] = u * rbf(g * fast_sigmoid(g));
          }
          *(unsigned*)(crow + (n >> 1) * 2) = pack2bf(o2[0], o2[1]);
          continue;
        }
        if (act == GVL_ACT_QUICK_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float x = rbf(v[e]); v[e] = x * rbf(fast_sigmoid(rbf(1.702f * x))); }
        } else if (act == GVL_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_tab_global(a.act_table, v[e]);
        }
        if (has_gamma) {
          const f32x4_t gv = *(const f32x4_t*)(a.gamma + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]) * gv[e];
        }
        if (out_f32) {
          if (has_resid) {
            const f32x4_t rv = *(const f32x4_t*)(rrow + n * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rv[e] + (a.round_pre_resid ? rbf(v[e]) : v[e]);
          }
          const f32x4_t o = {v[0], v[1], v[2], v[3]};
          *(f32x4_t*)(crow + n * 4) = o;
        } else {
          if (has_resid) {
            const u32x2_t rv = *(const u32x2_t*)(rrow + n * 2);
            v[0] = lo_bf(rv[0]) + rbf(v[0]); v[1] = hi_bf(rv[0]) + rbf(v[1]); v[2] = lo_bf(rv[1]) + rbf(v[2]); v[3] = hi_bf(rv[1]) + rbf(v[3]);
          }
          const u32x2_t o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *(u32x2_t*)(crow + n * 2) = o;
        }
      }
    }
  }
}

// Staged epilogue (ping-pong kernel): the natural MFMA store is one 8-byte piece per lane at a ROW stride -- a wave
// instruction touches 32 different cache lines with 16 bytes each and the store tail is issue-bound (MI355X guide, T21).
// Here every wave transposes its 32 x TN sub-tile through its private slice of the (now idle) LDS ring and writes whole
// rows: 16-byte pieces, 4 (bf16) / 2 (f32) full rows per wave instruction; the residual is read the same way.
// Requires 16-byte aligned rows (checked by the launcher); arithmetic and rounding points are those of gemm_epilogue.
// SWZ = 1 (bf16 rows of 256 bytes only): no row padding, 16-byte chunk c of row r is stored at chunk c ^ (r & 15) -- the
// slice is then exactly 8 KiB per wave, which lets the persistent kernel keep one ring slot free for the next tile.
// Latency: bias / gamma go global -> per-wave LDS scratch `bg` (ONE load per lane) -> broadcast ds_read_b128, and the
// residual pieces are requested up front into registers, so a tile pays one memory round trip instead of one per piece
// (hipcc serialises `load; s_waitcnt; use` chains inside the unrolled loops: measured 20-39 k cycles per tile before).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// HI = 0: bf16 pattern in bits 0..15 of p; HI = 1: in bits 16..31.  Returns the LDS byte address of Phi(x); tab_adj is the
// table's LDS address minus GELU_LO * 4 (uniform).  Five VALU ops, spelled out because hipcc's own selection needs seven.
template <int HI>
__device__ __forceinline__ unsigned gelu_tab_addr(unsigned p, unsigned tab_adj, unsigned lo, unsigned hi) {
  unsigned key, sg, off;
  if (HI) asm("v_bfe_u32 %0, %1, 16, 15" : "=v"(key) : "v"(p)); else asm("v_and_b32 %0, 0x7fff, %1" : "=v"(key) : "v"(p));
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(key) : "v"(key), "s"(lo), "v"(hi));   // one SGPR per VOP3 on gfx9
  asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(key), "s"(tab_adj));
  if (HI) asm("v_lshrrev_b32 %0, 31, %1" : "=v"(sg) : "v"(p)); else asm("v_bfe_u32 %0, %1, 15, 1" : "=v"(sg) : "v"(p));
  asm("v_lshl_add_u32 %0, %1, 13, %2" : "=v"(off) : "v"(sg), "v"(off));
  return off;
}
template <int NB, int EPI>
struct StgGeom {                                   // compile-time geometry of one wave's staged read-back
  static constexpr int act = EPI & 3;
  static constexpr bool out_f32 = (EPI >> 2) & 1, has_resid = (EPI >> 3) & 1, has_gamma = (EPI >> 4) & 1, has_bias = (EPI >> 5) & 1;
  static constexpr bool has_rowscale = (EPI >> 6) & 1, has_rowsq = (EPI >> 7) & 1;   // fused RMSNorm: consumer / producer side (GemmArgs)
  static constexpr bool silu = act == GVL_ACT_SILU_MUL;
  static constexpr int TN = NB * 32, OUTC = silu ? TN / 2 : TN, ES = out_f32 ? 4 : 2;
  static constexpr int LPR = OUTC * ES / 16, RPI = 64 / LPR, KI = 32 / RPI;   // lanes / row, rows / instruction, instructions / 32-row block
  static constexpr int CPL = TN / 64;              // bias columns per lane (2 for TN = 128, 1 for TN = 64)
};
// buffer descriptor over the EXISTING rows [mw, min(mw + rows, M)) of a wave tile of `base` (row pitch row_bytes): the
// hardware bounds check then drops rows >= M, and lanes whose column is out of range use the offset 2^31 (always dropped)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stg_rsrc(const GemmArgs& a, const void* base, size_t row_bytes, int mw, int rows) {
  const int rows_ok = a.M - mw < rows ? (a.M - mw > 0 ? a.M - mw : 0) : rows;
  const unsigned long long p = (unsigned long long)base + (size_t)(mw + a.row_off) * row_bytes;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  const unsigned nrec = __builtin_amdgcn_readfirstlane((unsigned)(rows_ok * row_bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, nrec, 0x00020000);
}
// residual pieces of 32-row block j of the wave tile at (mw, nw) -> rv  (KI x 16 bytes per lane, whole rows per instruction)
template <int MB, int NB, int EPI>
__device__ __forceinline__ void stg_request_resid(const GemmArgs& a, int mw, int nw, int lane, int j, u32x4_t (&rv)[StgGeom<NB, EPI>::KI]) {
  using G = StgGeom<NB, EPI>;
  const int nc = nw + (lane % G::LPR) * (16 / G::ES);
  const size_t ldrb = (size_t)a.ldr * G::ES;
  const __amdgpu_buffer_rsrc_t rrs = stg_rsrc(a, a.resid, ldrb, mw, MB * 32);
  const unsigned roff = nc < a.N ? (unsigned)((lane / G::LPR) * ldrb) + (unsigned)nc * G::ES : 0x80000000u;
#pragma unroll
  for (int k = 0; k < G::KI; ++k)
    rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rrs, (int)(roff + (unsigned)((j * 32 + k * G::RPI) * ldrb)), 0, 0);
}
// bias / gamma slice of the wave (TN floats each): one global load per lane ...
template <int NB, int EPI>
__device__ __forceinline__ void stg_request_bias(const GemmArgs& a, int nw, int lane, u32x2_t& bv, u32x2_t& gv) {
  using G = StgGeom<NB, EPI>;
  int c0 = nw + lane * G::CPL;
  c0 = c0 + G::CPL <= a.N ? c0 : a.N - G::CPL;     // overhanging columns read a valid address and are never stored
  if constexpr (G::CPL == 2) {
    if (G::has_bias) bv = *(const u32x2_t*)(a.bias + c0);
    if (G::has_gamma) gv = *(const u32x2_t*)(a.gamma + c0);
  } else {
    if (G::has_bias) bv[0] = *(const unsigned*)(a.bias + c0);
    if (G::has_gamma) gv[0] = *(const unsigned*)(a.gamma + c0);
  }
}
// ... and into the wave's LDS scratch `bg` (read back as broadcast ds_read_b128 in the epilogue)
template <int NB, int EPI>
__device__ __forceinline__ void stg_store_bias(char* bg, int lane, const u32x2_t& bv, const u32x2_t& gv) {
  using G = StgGeom<NB, EPI>;
  if constexpr (G::CPL == 2) {
    if (G::has_bias) *(u32x2_t*)(bg + lane * 8) = bv;
    if (G::has_gamma) *(u32x2_t*)(bg + G::TN * 4 + lane * 8) = gv;
  } else {
    if (G::has_bias) *(unsigned*)(bg + lane * 4) = bv[0];
    if (G::has_gamma) *(unsigned*)(bg + G::TN * 4 + lane * 4) = gv[0];
  }
}

// rowscale[m] of the MB rows this lane owns (row mw + 32 j + (lane & 31), clamped to the last row: overhanging rows are never stored)
template <int MB>
__device__ __forceinline__ void stg_request_rowscale(const GemmArgs& a, int mw, int lane, float (&rsc)[MB]) {
#pragma unroll
  for (int j = 0; j < MB; ++j) { int r = mw + j * 32 + (lane & 31); r = r < a.M ? r : a.M - 1; rsc[j] = a.rowscale[r]; }
}
// sum of squares of the 8 bf16 values of a 16-byte piece, then over the 8 lanes that hold one aligned 64-column block of a row.  The order is FIXED
// (v_dot2c per dword in order; lane pairs, quads, the two quads) and depends only on the column -> lane map of the staged read-back, which every kernel
// shares: a row's partial sums are the same whichever kernel of the launch plan stored it.
__device__ __forceinline__ float stg_sumsq8(const u32x4_t& v) {
  // v_dot2c_f32_bf16: d += a.lo * b.lo + a.hi * b.hi.  Spelled in asm: hipcc's own lowering of __builtin_amdgcn_fdot2_f32_bf16 on the elements of a
  // 4-dword vector reads element 0 four times (ROCm 7.2 clang; found by tests/test_gpu_ops.py::test_gemm_row_sums_of_squares).  s_nop 1: the DPP
  // reads below need two wait states behind a VALU write of the same register, and the hazard recogniser does not look inside asm.
  float s = 0.f;
  const unsigned a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
  asm("v_dot2c_f32_bf16 %0, %1, %1\n\tv_dot2c_f32_bf16 %0, %2, %2\n\tv_dot2c_f32_bf16 %0, %3, %3\n\tv_dot2c_f32_bf16 %0, %4, %4\n\ts_nop 1" : "+v"(s) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]: lane ^ 1
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]: lane ^ 2
  s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, false));   // row_half_mirror: lane i <-> 7 - i of its 8
  return s;
}

// PRE bit 0: the caller has already put bias/gamma into `bg`; bit 1: it has requested residual block 0 into rv (ping-pong
// kernel: both are issued inside the main loop, so the epilogue starts with its operands on chip); bit 2: rowscale is in rsc.
// AccRow: acc_row(j, r) fills r[0..NB) with the NB accumulator blocks of 32-row block j of the wave tile (j is a constant after unrolling).  The kernels
// whose accumulators are compiler values pass AccArrayRow (a copy the optimiser removes); the 4-wave kernel reads its AGPRs there (gvl_gemm4.hip).
template <int MB, int NB>
struct AccArrayRow {
  f32x16_t (&acc)[NB][MB];
  __device__ __forceinline__ void operator()(int j, f32x16_t (&r)[NB]) const {
#pragma unroll
    for (int i = 0; i < NB; ++i) r[i] = acc[i][j];
  }
};
template <int MB, int NB, int EPI, int SWZ = 0, int PRE = 0, int TABLE = 0, typename AccRow, typename Hook = NoHook>
__device__ __forceinline__ void gemm_epilogue_staged_rows(const GemmArgs& a, AccRow&& acc_row, char* stg, char* bg, int mw, int nw, int lane,
                                                          u32x4_t (&rv)[StgGeom<NB, EPI>::KI], float (&rsc)[MB], Hook&& after_requests = NoHook(), const char* tab = nullptr) {
  static_assert(EPI >= 0, "staged epilogue is compile-time specialised");
  using G = StgGeom<NB, EPI>;
  constexpr int act = G::act;
  constexpr bool out_f32 = G::out_f32, has_resid = G::has_resid, has_gamma = G::has_gamma, has_bias = G::has_bias, silu = G::silu;
  constexpr bool has_rowscale = G::has_rowscale, has_rowsq = G::has_rowsq;
  static_assert(!has_rowsq || (!out_f32 && !silu), "row sums of squares: bf16 outputs of the full tile width");
  constexpr int TN = G::TN, OUTC = G::OUTC, ES = G::ES, LPR = G::LPR, RPI = G::RPI, KI = G::KI;
  static_assert(SWZ == 0 || (OUTC * ES == 256 && !out_f32 && !silu), "swizzled staging: 256-byte bf16 rows");
  constexpr int ROWB = OUTC * ES + (SWZ ? 0 : 16); // +16: the column-of-rows writes spread over the banks
  const int l31 = lane & 31, h = lane >> 5;
  const int n_out0 = silu ? (nw >> 1) : nw, n_out_end = silu ? (a.N >> 1) : a.N;
  const int rrow_l = lane / LPR, chunk = lane % LPR;
  const int nc = n_out0 + chunk * (16 / ES);
  const bool col_ok = nc < n_out_end;

  // ---- requests first (PRE = 0): bias / gamma slice of this wave, residual block 0 ------------------------------------
  if constexpr ((PRE & 1) == 0 && (has_bias || has_gamma)) {
    u32x2_t bv, gv;
    stg_request_bias<NB, EPI>(a, nw, lane, bv, gv);
    stg_store_bias<NB, EPI>(bg, lane, bv, gv);
  }
  if constexpr ((PRE & 2) == 0 && has_resid) stg_request_resid<MB, NB, EPI>(a, mw, nw, lane, 0, rv);
  if constexpr ((PRE & 4) == 0 && has_rowscale) stg_request_rowscale<MB>(a, mw, lane, rsc);
  // row statistics: one f32 per (row, aligned 64-column block), stored by the first of the 8 lanes that hold the block; rows >= M and blocks >= N fall
  // outside the descriptor / take the dropped offset
  __amdgpu_buffer_rsrc_t qrs;
  unsigned qoff = 0x80000000u;
  if constexpr (has_rowsq) {
    const unsigned long long qp = (unsigned long long)a.rowsq;
    const unsigned qlo = __builtin_amdgcn_readfirstlane((unsigned)qp), qhi = __builtin_amdgcn_readfirstlane((unsigned)(qp >> 32));
    const long long qbytes = (long long)a.M * a.rowsq_ld * 4;
    qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)qhi << 32) | qlo), 0, __builtin_amdgcn_readfirstlane((unsigned)(qbytes > 0xffffffffll ? 0xffffffffll : qbytes)), 0x00020000);
    const int blk = (nw >> 6) + (chunk >> 3);                    // aligned 64-column block of this lane's piece
    if ((lane & 7) == 0 && blk * 64 < a.N) qoff = (unsigned)(((mw + rrow_l) * a.rowsq_ld + blk) * 4);
  }
  // output addressing: straight-line buffer stores (no exec-masked branches into which hipcc would sink the residual adds)
  const size_t ldcb = (size_t)a.ldc * ES;
  const __amdgpu_buffer_rsrc_t crs = stg_rsrc(a, a.C, ldcb, mw, MB * 32);
  const unsigned coff = col_ok ? (unsigned)(rrow_l * ldcb) + (unsigned)nc * ES : 0x80000000u;
  // the caller's next-tile DMA goes BEHIND the requests: vmcnt retires in order, so a wait for an operand would otherwise
  // also wait for the whole prefetch
  after_requests();

#pragma unroll
  for (int j = 0; j < MB; ++j) {
    f32x16_t accj[NB];
    acc_row(j, accj);
    char* wrow = stg + l31 * ROWB;
    const int wx = SWZ ? ((l31 & 15) << 4) : 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      // bias / gamma of this 32-column block: 2 + 2 broadcast reads at a time, fenced so that hipcc does not hoist all 32
      // reads of the tile to the top (128 VGPRs -> spills, whose reloads force vmcnt(0) in front of every store)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
      f32x4_t bv4[4], gv4[4];
#pragma unroll
      for (int b = 2 * hb; b < 2 * hb + 2; ++b) {
        if (has_bias) bv4[b] = *(const f32x4_t*)(bg + (i * 32 + 8 * b + 4 * h) * 4);
        if (has_gamma) gv4[b] = *(const f32x4_t*)(bg + TN * 4 + (i * 32 + 8 * b + 4 * h) * 4);
      }
      float vv[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int b = 2 * hb + q;
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[q][e] = (has_rowscale ? accj[i][4 * b + e] * rsc[j] : accj[i][4 * b + e]) + (has_bias ? bv4[b][e] : 0.f);
      }
      if constexpr (act == GVL_ACT_GELU && TABLE != 0) {
        // Phi table resident in LDS: 8 offsets, 8 reads in flight, 8 products -- one LDS latency per 8 elements
        unsigned pk[4]; float phi[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) { pk[2 * q] = pack2bf(vv[q][0], vv[q][1]); pk[2 * q + 1] = pack2bf(vv[q][2], vv[q][3]); }
        const unsigned tab_adj = __builtin_amdgcn_readfirstlane(lds_addr(tab)) - GELU_LO * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          phi[2 * e] = *(const __attribute__((address_space(3))) float*)(size_t)gelu_tab_addr<0>(pk[e], tab_adj, GELU_LO, GELU_HI);
          phi[2 * e + 1] = *(const __attribute__((address_space(3))) float*)(size_t)gelu_tab_addr<1>(pk[e], tab_adj, GELU_LO, GELU_HI);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vv[e >> 1][2 * (e & 1)] = lo_bf(pk[e]) * phi[2 * e];
          vv[e >> 1][2 * (e & 1) + 1] = hi_bf(pk[e]) * phi[2 * e + 1];
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int b = 2 * hb + q;
        const int nl = i * 32 + 8 * b + 4 * h;
        float (&v)[4] = vv[q];
        if (act == GVL_ACT_QUICK_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float x = rbf(v[e]); v[e] = x * rbf(fast_sigmoid(rbf(1.702f * x))); }
        } else if (act == GVL_ACT_GELU && TABLE == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_tab_global(a.act_table, v[e]);
        }
        if (has_gamma) {
          const f32x4_t gv = gv4[b];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]) * gv[e];
        }
        if (silu) {
          float o2[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) { const float g = rbf(v[2 * e]), u = rbf(v[2 * e + 1]); o2[e] = u * rbf(g * fast_sigmoid(g)); }
          *(unsigned*)(wrow + (nl >> 1) * 2) = pack2bf(o2[0], o2[1]);
        } else if (out_f32) {
          if (has_resid && a.round_pre_resid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
          }
          const f32x4_t o = {v[0], v[1], v[2], v[3]};
          *(f32x4_t*)(wrow + nl * 4) = o;
        } else {
          const u32x2_t o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *(u32x2_t*)(wrow + ((nl * 2) ^ wx)) = o;
        }
      }
      if constexpr (has_bias || has_gamma) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // read-back in two passes: hipcc must use vmcnt(0) whenever loads AND stores are pending (they retire out of order
    // with respect to each other), so a store issued between two residual uses would serialise a full round trip per piece
    u32x4_t ov[KI];
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int row = k * RPI + rrow_l;
      u32x4_t sv = *(const u32x4_t*)(stg + row * ROWB + ((chunk ^ (SWZ ? (row & 15) : 0)) << 4));
      if (has_resid) {
        const u32x4_t r4 = rv[k];
        if (out_f32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) sv[e] = __float_as_uint(__uint_as_float(r4[e]) + __uint_as_float(sv[e]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) sv[e] = pack2bf(lo_bf(r4[e]) + lo_bf(sv[e]), hi_bf(r4[e]) + hi_bf(sv[e]));
        }
      }
      ov[k] = sv;
      if constexpr (has_rowsq) {
        const float ssq = stg_sumsq8(sv);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ssq), qrs, (int)(qoff + (unsigned)((j * 32 + k * RPI) * a.rowsq_ld * 4)), 0, 0);
      }
    }
    if constexpr (has_resid) {
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < MB) stg_request_resid<MB, NB, EPI>(a, mw, nw, lane, j + 1, rv);   // ahead of this block's stores and the next block's math
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      __builtin_amdgcn_raw_buffer_store_b128(ov[k], crs, (int)(coff + (unsigned)((j * 32 + k * RPI) * ldcb)), 0, 0);
    }
  }
}

template <int MB, int NB, int EPI, int SWZ = 0, int PRE = 0, int TABLE = 0, typename Hook = NoHook>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs& a, f32x16_t (&acc)[NB][MB], char* stg, char* bg, int mw, int nw, int lane,
                                                     u32x4_t (&rv)[StgGeom<NB, EPI>::KI], float (&rsc)[MB], Hook&& after_requests = NoHook(), const char* tab = nullptr) {
  gemm_epilogue_staged_rows<MB, NB, EPI, SWZ, PRE, TABLE>(a, AccArrayRow<MB, NB>{acc}, stg, bg, mw, nw, lane, rv, rsc, static_cast<Hook&&>(after_requests), tab);
}
