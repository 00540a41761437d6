// gvl_gemm4p.hip -- the 4-wave bf16 MFMA GEMM with its epilogue SOFTWARE-PIPELINED ACROSS OUTPUT TILES (round 6), gfx950 only.
//
// Same contract, operand layout, k order and epilogue arithmetic (hence the same bits) as gemm_pp_kernel (gvl_gemm.hip) and gemm_a4_kernel (gvl_gemm4.hip).
// What is new: with one wave per SIMD nothing hides a tile's epilogue -- 6 k ... 21 k cycles of VALU / LDS / store work per tile with the matrix pipe idle, 11 ... 31 %
// of a K = 1408 tile (profiles/r06_gemm4_anatomy.txt).  Here a tile's loop statement ends with a short DRAIN (accumulators -> x row scale, + bias -> bf16 pairs in 128
// VGPRs: the value every fused epilogue of the library is defined on), and everything behind it -- transposition through a private LDS staging area, residual add,
// row sums of squares, whole-row stores -- runs as fillers in the MFMA gaps of the workgroup's NEXT tile (tools/gen_gemm4p.py -> gvl_gemm4p_loop.inc; a flush
// statement finishes the last tile).  The statements take fixed-register operand blocks (v[0:15], s[36:51], s[52:67]) and the P registers v[32:159] as in / out
// operands, so the compiler keeps them alive between tiles by construction.
// Reference shapes: models/internvideo2.py:587,603,631-634; models/modeling_phi3.py:459-464,659-663.
#include "gvl_gemm_epi.h"
#include "gvl_gemm4p_loop.inc"
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

typedef __attribute__((ext_vector_type(16))) unsigned int u32x16_t;

namespace {
struct PRegs { u32x16_t p[8]; };     // P: the drained tile, v[32:159]
struct SBlock {                      // 16 wave-uniform dwords, s[N : N + 15], as four tuples (+ the bias resource s[68:71] behind the second block)
  u32x4_t q[4];
  u32x4_t bias;
  // readfirstlane: every value IS wave-uniform, but hipcc's divergence analysis gives up on vectors carried around the tile loop and would build the tuple in VGPRs
  __device__ __forceinline__ void set(int e, unsigned v) { q[e >> 2][e & 3] = __builtin_amdgcn_readfirstlane(v); }
};

#define GVL_A4P_P_OUT(P) "={v[32:47]}"(P.p[0]), "={v[48:63]}"(P.p[1]), "={v[64:79]}"(P.p[2]), "={v[80:95]}"(P.p[3]), "={v[96:111]}"(P.p[4]), "={v[112:127]}"(P.p[5]), "={v[128:143]}"(P.p[6]), "={v[144:159]}"(P.p[7])
#define GVL_A4P_P_INOUT(P) "+{v[32:47]}"(P.p[0]), "+{v[48:63]}"(P.p[1]), "+{v[64:79]}"(P.p[2]), "+{v[80:95]}"(P.p[3]), "+{v[96:111]}"(P.p[4]), "+{v[112:127]}"(P.p[5]), "+{v[128:143]}"(P.p[6]), "+{v[144:159]}"(P.p[7])
#define GVL_A4P_P_IN(P) "{v[32:47]}"(P.p[0]), "{v[48:63]}"(P.p[1]), "{v[64:79]}"(P.p[2]), "{v[80:95]}"(P.p[3]), "{v[96:111]}"(P.p[4]), "{v[112:127]}"(P.p[5]), "{v[128:143]}"(P.p[6]), "{v[144:159]}"(P.p[7])
// the scalar blocks travel as 4-dword tuples (the width of a buffer resource: the one SGPR tuple class hipcc copies around without detours through VGPRs)
#define GVL_A4P_IN(vp, sa, sb) "{v[0:15]}"(vp), "{s[36:39]}"(sa.q[0]), "{s[40:43]}"(sa.q[1]), "{s[44:47]}"(sa.q[2]), "{s[48:51]}"(sa.q[3]), \
                               "{s[52:55]}"(sb.q[0]), "{s[56:59]}"(sb.q[1]), "{s[60:63]}"(sb.q[2]), "{s[64:67]}"(sb.q[3]), "{s[68:71]}"(sb.bias)
#define GVL_A4P_CLOB "memory", "scc", GVL_A4P_CLOBBER_SGPRS, "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", \
  "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", \
  "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", \
  "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", \
  "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

template <int EPI> struct A4pAsm;
#define GVL_A4P_DEF(E)                                                                                                                                   \
  template <> struct A4pAsm<E> {                                                                                                                         \
    static constexpr int MIN_NK = GVL_A4P_MIN_NK_E##E;                                                                                                   \
    static __device__ __forceinline__ void tile0(PRegs& P, const u32x16_t& vp, const SBlock& sa, const SBlock& sb) {                                 \
      asm volatile(GVL_A4P_TILE0_E##E : GVL_A4P_P_OUT(P) : GVL_A4P_IN(vp, sa, sb) : GVL_A4P_CLOB, GVL_A4P_CLOBBER_AGPRS);                               \
    }                                                                                                                                                    \
    static __device__ __forceinline__ void tile(PRegs& P, const u32x16_t& vp, const SBlock& sa, const SBlock& sb) {                                  \
      asm volatile(GVL_A4P_TILE_E##E : GVL_A4P_P_INOUT(P) : GVL_A4P_IN(vp, sa, sb) : GVL_A4P_CLOB, GVL_A4P_CLOBBER_AGPRS);                              \
    }                                                                                                                                                    \
    static __device__ __forceinline__ void flush(const PRegs& P, const u32x16_t& vp, const SBlock& sa, const SBlock& sb) {                           \
      asm volatile(GVL_A4P_FLUSH_E##E : : GVL_A4P_P_IN(P), GVL_A4P_IN(vp, sa, sb) : GVL_A4P_CLOB);                                                     \
    }                                                                                                                                                    \
  };
GVL_A4P_EPI_LIST(GVL_A4P_DEF)
#undef GVL_A4P_DEF
// NARROW statements (tools/gen_gemm4p.py, body(nb = 2)): a tile with <= 128 valid columns -- N = 1408 = 5.5 tile columns: every sixth tile of InternVideo2's proj /
// fc2 -- runs as 4 waves x (128 rows x 64 columns): 32 instead of 64 MFMAs per wave and k-tile, none of them on the zero half.  Same deferred program, same k order
// per output element: same bits.
// The tile statements of these epilogues hold BOTH codes and branch on a flag the host puts into vp[12] (the row-scale slot: these epilogues have none): two asm statements under a C++ `if` made hipcc spill P at the join.
constexpr int A4P_NARROW_MAX_NK = 32;
template <int EPI> struct A4pNarrow { static constexpr bool has = false; static constexpr int MIN_NK = 1 << 30; };
#define GVL_A4P_DEFN(E) template <> struct A4pNarrow<E> { static constexpr bool has = true; static constexpr int MIN_NK = GVL_A4P_MIN_NK_N_E##E; };
GVL_A4P_NARROW_LIST(GVL_A4P_DEFN)
#undef GVL_A4P_DEFN
}  // namespace

// LDS: ring 2 x 64 KiB | staging 4 waves x 4 KiB (the bias slice of a tile is DMA'd there for its drain) | LayerScale scratch 2 buffers x 4 x 512 B  OR  the Phi
// table of the erf-GELU (16 KiB: with it the workgroup owns all 160 KiB of the CU; no epilogue has both)
constexpr int A4P_STG_OFF = 2 * 65536, A4P_GAMMA_OFF = A4P_STG_OFF + 4 * 4096, A4P_TAB_OFF = A4P_GAMMA_OFF;
template <int EPI> constexpr int a4p_lds() { return A4P_GAMMA_OFF + ((EPI & 3) == GVL_ACT_GELU ? GELU_TAB_BYTES : ((EPI & 16) ? 2 * 4 * 512 : 0)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_a4p_kernel(const GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int BM = 256, BN = 256, TM = 128, TN = 128, MB = 4, NB = 4, SLOT = 65536;
  static_assert(EPI >= 0 && !(EPI & 4), "bf16 output, compile-time epilogue");
  using G = StgGeom<NB, EPI>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

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

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  // Every per-lane term is RE-derived from the lane id where it is used (fresh_lane launders it): the loop statements leave the compiler 16 VGPRs (v16 - v31) for
  // what it keeps alive across them, and rd / dma / output offsets kept live would spill.
  int ln = tid & 63;
  auto fresh_lane = [&]() { asm volatile("" : "+v"(ln)); return ln; };
  const unsigned pitchW = (unsigned)a.ldw * 2u, pitchA = (unsigned)a.lda * 2u;
  const unsigned ldcb = (unsigned)a.ldc * 2u, ldrb = (unsigned)a.ldr * 2u, ldqb = (unsigned)a.rowsq_ld * 4u;
  const unsigned nk = (unsigned)(a.K / BK);
  // narrow tiles (A4pNarrow): the column tile that holds <= 128 real columns; wave (wm, wn) then owns columns 64 wn ... 64 wn + 63 of it
  // Only for K <= A4P_NARROW_MAX_NK k-tiles: a narrow k-tile is 1 024 MFMA cycles, less than the latency of an operand piece that comes from HBM -- at K = 6144
  // (InternVideo2 fc2: A streams from HBM) a narrow tile measured 1.16 x a full one; at K = 1408 (proj: A is the attention output, cache-resident) ~0.7 x
  const bool can_narrow = A4pNarrow<EPI>::has && a.narrow && (int)nk >= A4pNarrow<EPI>::MIN_NK && (int)nk <= A4P_NARROW_MAX_NK;
  auto is_narrow = [&](int n0_) { return can_narrow && a.N - n0_ <= 128; };
  // LayerScale gamma -> buffer `gb` of the wave's gamma scratch (read by the deferred program one statement LATER: two buffers, alternating per tile).  The bias is
  // not staged by this code at all: the loop statement DMAs the tile's slice into the idle staging area (tools/gen_gemm4p.py, LAST body)
  auto store_gamma = [&](const u32x2_t& gv, unsigned gb) {
    if constexpr (G::has_gamma) *(u32x2_t*)(smem + A4P_GAMMA_OFF + gb * 2048 + wave * 512 + fresh_lane() * 8) = gv;
  };

  // Tile walk: round j of this XCD = the wpx consecutive tiles [j wpx, (j + 1) wpx) of its chunk (what its CUs run at the same time: the L2 sharing set); workgroup c
  // takes offset (c + rot[j]) mod wpx of the round.  rot = 0 is the fixed assignment of rounds <= 6a; with NARROW tiles (0.7 of a tile's time, every sixth tile at
  // N = 1408) the fixed assignment hands all of them to a quarter of the CUs and nobody finishes earlier -- the launcher's rot table (balance_rounds) spreads them evenly.
  const int cw = bid >> 3;
  if (cw >= xcnt) return;
  const int rounds = (xcnt + wpx - 1) / wpx;
  auto vid_of = [&](int j) -> int {                  // tile of round j, or -1 (only the last round can leave a workgroup without one)
    int o = cw + (int)a.rot[j & (GVL_GEMM_ROT_LEN - 1)];
    o = o >= wpx ? o - wpx : o;
    const int v = j * wpx + o;
    return (j < rounds && v < xcnt) ? xbase + v : -1;
  };
  if (smem_base & 0x1ffffu) __builtin_trap();        // the ring slots are toggled by XOR 0x10000: the dynamic LDS block must start at a multiple of 128 KiB (it starts at 0)

  // scalar block A: what does not change from tile to tile (dma_other, the scratch addresses are patched per tile)
  SBlock sa, sb;
  u32x16_t vp;
  {
    const unsigned long long pw = (unsigned long long)a.W, pa = (unsigned long long)a.A;
    sa.set(0, (unsigned)pw); sa.set(1, (unsigned)(pw >> 32) & 0xffffu); sa.set(2, (unsigned)a.N * pitchW); sa.set(3, 0x00020000u);
    sa.set(4, (unsigned)pa); sa.set(5, (unsigned)(pa >> 32) & 0xffffu); sa.set(6, (unsigned)a.M * pitchA); sa.set(7, 0x00020000u);
    sa.set(8, pitchW * 32u); sa.set(9, pitchA * 32u); sa.set(10, 0); sa.set(11, nk);
    sa.set(12, 0); sa.set(13, 0); sa.set(14, smem_base + A4P_TAB_OFF - GELU_LO * 4); sa.set(15, smem_base + A4P_STG_OFF + (unsigned)wave * 4096u);
    const unsigned long long pb = (unsigned long long)a.bias;
    sb.bias[0] = __builtin_amdgcn_readfirstlane((unsigned)pb); sb.bias[1] = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32) & 0xffffu);
    sb.bias[2] = __builtin_amdgcn_readfirstlane(G::has_bias ? (unsigned)a.N * 4u : 0u); sb.bias[3] = 0x00020000u;
  }
  if constexpr ((EPI & 3) == GVL_ACT_GELU) {          // Phi table -> LDS once per (persistent) workgroup; its first read is behind the first tile's barriers
    for (int o = tid * 16; o < GELU_TAB_BYTES; o += 256 * 16) *(u32x4_t*)(smem + A4P_TAB_OFF + o) = *(const u32x4_t*)((const char*)a.act_table + o);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) { sb.set(e, 0); vp[e] = 0; }
  sb.set(3, 0x00020000u); sb.set(7, 0x00020000u); sb.set(11, 0x00020000u);
  sb.set(12, ldcb * 8u); sb.set(13, ldrb * 8u); sb.set(14, ldqb * 8u);

  // per-tile pieces -------------------------------------------------------------------------------------------------------------------------------------
  auto loop_params = [&](int m0, int n0, int m0n, int n0n, bool has_next, unsigned par, bool narrow) {
    const int wcol = wn * (narrow ? 64 : TN);
    const int lane = fresh_lane();
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned rd_lane = (unsigned)(l31 * 128 + ((h ^ ((l31 >> 1) & 7)) << 4));    // this lane's k-step-0 chunk of fragment row l31
    const int rowl = wave * 8 + (lane >> 3);                                            // DMA piece i covers tile rows 32 i + rowl, chunk lane & 7 of the LDS row
    const unsigned dma_lane = (unsigned)(((lane & 7) ^ ((rowl >> 1) & 7)) << 4);
    vp[0] = smem_base + par * SLOT + (unsigned)(wcol * 128) + rd_lane; vp[1] = smem_base + par * SLOT + (unsigned)(BN * 128 + wm * TM * 128) + rd_lane;
    vp[2] = (unsigned)(n0 + rowl) * pitchW + dma_lane; vp[3] = (unsigned)(m0 + rowl) * pitchA + dma_lane;
    vp[4] = has_next ? (unsigned)(n0n + rowl) * pitchW + dma_lane : 0x80000000u; vp[5] = has_next ? (unsigned)(m0n + rowl) * pitchA + dma_lane : 0x80000000u;
    vp[11] = (unsigned)(n0 + wcol + lane) * 4u;                                       // bias of this tile's columns (beyond N: out of the resource's range -> 0)
    sa.set(10, smem_base + (par ^ 1u) * SLOT + (unsigned)wave * 1024u);
  };
  // the deferred program's view of a finished tile (wave tile at mw, nw): resources over its existing rows, per-lane offsets (row lane >> 3, 16-byte piece lane & 7)
  auto defer_params = [&](int m0, int n0, bool narrow) {
    const int lane = fresh_lane();
    const int mw = m0 + wm * TM, nw = n0 + wn * (narrow ? 64 : TN);
    const int nlim = narrow && nw + 64 < a.N ? nw + 64 : a.N;      // narrow: the wave tile IS its first 64 columns (a scalar limit: a per-lane `&& !(narrow && half)` made hipcc spill vp)
    const int rows_ok = a.M - mw < TM ? (a.M - mw > 0 ? a.M - mw : 0) : TM;
    const unsigned long long pc = (unsigned long long)a.C + (unsigned long long)(mw + a.row_off) * ldcb;
    sb.set(0, __builtin_amdgcn_readfirstlane((unsigned)pc)); sb.set(1, __builtin_amdgcn_readfirstlane((unsigned)(pc >> 32)) & 0xffffu); sb.set(2, (unsigned)rows_ok * ldcb);
    if constexpr (G::has_resid) {
      const unsigned long long pr = (unsigned long long)a.resid + (unsigned long long)(mw + a.row_off) * ldrb;
      sb.set(4, __builtin_amdgcn_readfirstlane((unsigned)pr)); sb.set(5, __builtin_amdgcn_readfirstlane((unsigned)(pr >> 32)) & 0xffffu); sb.set(6, (unsigned)rows_ok * ldrb);
    }
    if constexpr (G::has_rowsq) {
      const unsigned long long pq = (unsigned long long)a.rowsq + (unsigned long long)mw * ldqb;
      sb.set(8, __builtin_amdgcn_readfirstlane((unsigned)pq)); sb.set(9, __builtin_amdgcn_readfirstlane((unsigned)(pq >> 32)) & 0xffffu); sb.set(10, (unsigned)rows_ok * ldqb);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // SwiGLU: the wave tile has 64 OUTPUT columns (one 128-byte piece row), at column nw / 2 of the [M, N / 2] output
      const int col = G::silu ? (nw >> 1) + (lane & 7) * 8 : nw + half * 64 + (lane & 7) * 8;
      const bool ok = G::silu ? (half == 0 && col < (a.N >> 1)) : col < nlim;
      vp[6 + half] = ok ? (unsigned)(lane >> 3) * ldcb + (unsigned)col * 2u : 0x80000000u;
      if constexpr (G::has_resid) vp[8 + half] = ok ? (unsigned)(lane >> 3) * ldrb + (unsigned)col * 2u : 0x80000000u;
    }
    if constexpr (G::has_rowsq) {
      // row statistics: lane L keeps, per piece k, the sum of pass L & 7 = (block row j, column half) for row 32 j + 8 k + (L >> 3): one store per k writes them all
      const int pj = (lane & 7) >> 1, phalf = lane & 1;
      const int blk = (nw >> 6) + phalf;
      vp[10] = blk * 64 < nlim ? (unsigned)(32 * pj + (lane >> 3)) * ldqb + (unsigned)blk * 4u : 0x80000000u;
    }
  };
  auto load_rs = [&](int m0, float (&rs)[MB]) {
    if constexpr (G::has_rowscale) stg_request_rowscale<MB>(a, m0 + wm * TM, fresh_lane(), rs);
  };
  auto load_gamma = [&](int n0, u32x2_t& gv) {        // the two gamma values of columns nw + 2 lane, + 1 (overhanging columns read a valid address and are never stored)
    if constexpr (G::has_gamma) { int c0 = n0 + wn * (is_narrow(n0) ? 64 : TN) + fresh_lane() * 2; c0 = c0 + 2 <= a.N ? c0 : a.N - 2; gv = *(const u32x2_t*)(a.gamma + c0); }
  };

  PRegs P;
  int m0, n0, m0n = 0, n0n = 0;
  int jr = 0;
  tile_of(vid_of(0), m0, n0);
  unsigned par = 0;
  float rs[MB], rsn[MB];
  u32x2_t gv, gvn;
  load_rs(m0, rs);
  load_gamma(n0, gv);
  int vnext = vid_of(1);
  bool has_next = vnext >= 0;
  if (has_next) tile_of(vnext, m0n, n0n);
  bool nar = is_narrow(n0);
  loop_params(m0, n0, m0n, n0n, has_next, par, nar);
  asm volatile(GVL_A4P_DMA_TILE_ASM : : GVL_A4P_IN(vp, sa, sb) : "memory", "scc", GVL_A4P_CLOBBER_SGPRS, "v232", "v233");
  unsigned gb = 0;                                   // gamma buffer of the tile about to run
  store_gamma(gv, gb);
  if (has_next) { load_rs(m0n, rsn); load_gamma(n0n, gvn); }
#pragma unroll
  for (int j = 0; j < MB; ++j) vp[12 + j] = G::has_rowscale ? __builtin_bit_cast(unsigned, rs[j]) : ((A4pNarrow<EPI>::has && j == 0 && nar) ? 1u : 0u);   // (no row scale: slot 12 = the narrow flag)
  A4pAsm<EPI>::tile0(P, vp, sa, sb);
  par = (par + nk) & 1u;
  int m0p = m0, n0p = n0;
  bool narp = nar;
  for (jr = 1; has_next; ++jr) {
    m0 = m0n; n0 = n0n;
    vnext = vid_of(jr + 1);
    has_next = vnext >= 0;
    if (has_next) tile_of(vnext, m0n, n0n);
    // the operands requested one tile ahead have landed (the loop statement waited for every vector-memory operation of this wave)
    sa.set(13, smem_base + A4P_GAMMA_OFF + gb * 2048u + (unsigned)wave * 512u);      // the deferred program of this statement belongs to the PREVIOUS tile
    gb ^= 1u;
    store_gamma(gvn, gb);
    nar = is_narrow(n0);
#pragma unroll
    for (int j = 0; j < MB; ++j) vp[12 + j] = G::has_rowscale ? __builtin_bit_cast(unsigned, rsn[j]) : ((A4pNarrow<EPI>::has && j == 0 && nar) ? 1u : 0u);
    if (has_next) { load_rs(m0n, rsn); load_gamma(n0n, gvn); }
    loop_params(m0, n0, m0n, n0n, has_next, par, nar);
    defer_params(m0p, n0p, narp);
    A4pAsm<EPI>::tile(P, vp, sa, sb);
    par = (par + nk) & 1u;
    m0p = m0; n0p = n0; narp = nar;
  }
  defer_params(m0p, n0p, narp);
  sa.set(13, smem_base + A4P_GAMMA_OFF + gb * 2048u + (unsigned)wave * 512u);
  A4pAsm<EPI>::flush(P, vp, sa, sb);
}

// Which offset of its XCD's round a workgroup takes (GemmArgs.rot), chosen so that the NARROW tiles (cost ~0.7 of a tile) are spread evenly over the workgroups: a greedy
// pass over the rounds, one shift per round for all XCDs, minimising the largest accumulated load (then the sum of squares).  Mirrors the kernel's walk exactly
// (chunk per XCD, band rasterisation).  Cached per geometry: a few hundred thousand cheap steps once per shape.
namespace {
struct RotKey { int tm, tn, band, grid, n_tail; bool operator<(const RotKey& o) const { return std::tie(tm, tn, band, grid, n_tail) < std::tie(o.tm, o.tn, o.band, o.grid, o.n_tail); } };
struct RotTab { unsigned char r[GVL_GEMM_ROT_LEN]; };
std::mutex g_rot_mu;
std::map<RotKey, RotTab> g_rot_cache;
}  // namespace
static void balance_rounds(int tiles_m, int tiles_n, int band, int grid, bool last_col_narrow, unsigned char (&rot)[GVL_GEMM_ROT_LEN]) {
  memset(rot, 0, sizeof(rot));
  if (!last_col_narrow || grid < 16) return;
  const RotKey key{tiles_m, tiles_n, band, grid, 1};
  { std::lock_guard<std::mutex> lk(g_rot_mu); auto it = g_rot_cache.find(key); if (it != g_rot_cache.end()) { memcpy(rot, it->second.r, sizeof(rot)); return; } }
  const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7;
  int xbase[8], xcnt[8], wpx[8], max_rounds = 0;
  for (int x = 0; x < 8; ++x) {
    xbase[x] = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q; xcnt[x] = q + (x < r ? 1 : 0); wpx[x] = (grid + 7 - x) >> 3;
    if (wpx[x] > 0) max_rounds = std::max(max_rounds, (xcnt[x] + wpx[x] - 1) / wpx[x]);
  }
  auto cost = [&](int vid) {                    // narrow: the last tile column
    const int bw = band * tiles_n, g = vid / bw, first_m = g * band, gm = std::min(band, tiles_m - first_m);
    return (vid - g * bw) / gm == tiles_n - 1 ? 0.7f : 1.0f;
  };
  std::vector<float> load(8 * 64, 0.f);
  RotTab tab; memset(tab.r, 0, sizeof(tab.r));
  const int w0 = wpx[0];
  bool uniform = w0 <= 64; for (int x = 0; x < 8; ++x) uniform = uniform && wpx[x] == w0;     // one shift must mean the same thing on every XCD
  if (uniform && max_rounds <= GVL_GEMM_ROT_LEN) {
    for (int j = 0; j < max_rounds; ++j) {
      int best = 0; float best_mx = 1e30f, best_sq = 1e30f;
      for (int sft = 0; sft < w0; ++sft) {
        float mx = 0.f, sq = 0.f;
        for (int x = 0; x < 8; ++x)
          for (int c = 0; c < w0; ++c) {
            int o = c + sft; o = o >= w0 ? o - w0 : o;
            const int v = j * w0 + o;
            const float l = load[x * 64 + c] + (v < xcnt[x] ? cost(xbase[x] + v) : 0.f);
            mx = std::max(mx, l); sq += l * l;
          }
        if (mx < best_mx - 1e-4f || (mx < best_mx + 1e-4f && sq < best_sq - 1e-3f)) { best = sft; best_mx = mx; best_sq = sq; }
      }
      tab.r[j] = (unsigned char)best;
      for (int x = 0; x < 8; ++x)
        for (int c = 0; c < w0; ++c) {
          int o = c + best; o = o >= w0 ? o - w0 : o;
          const int v = j * w0 + o;
          if (v < xcnt[x]) load[x * 64 + c] += cost(xbase[x] + v);
        }
    }
  }
  memcpy(rot, tab.r, sizeof(rot));
  std::lock_guard<std::mutex> lk(g_rot_mu); g_rot_cache[key] = tab;
}

static std::atomic<int> g_narrow{1};          // A/B only, process-wide like gemm_a4 / gemm_band
void gvl_gemm_set_narrow(int v) { g_narrow.store(v, std::memory_order_relaxed); }

template <int EPI>
static int launch_a4p(const GemmArgs& a_in, hipStream_t st) {
  static GvlDevOnce once;
  static const int n_cu = [] {
    hipDeviceProp_t p; int d = 0;
    return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? (p.multiProcessorCount & ~7) : 256;
  }();
  auto kern = gemm_a4p_kernel<EPI>;
  if (gvl_set_max_lds(once, (const void*)kern, a4p_lds<EPI>())) return -3;
  GemmArgs a = a_in;
  if (a.K / BK < A4pAsm<EPI>::MIN_NK) return -2;
  const int tiles_m = (a.M - a.m_begin + 255) / 256, tiles_n = (a.N + 255) / 256;
  const int tiles = tiles_m * tiles_n;
  if (a.band <= 0) a.band = GVL_GEMM_BAND;
  a.narrow = g_narrow.load(std::memory_order_relaxed);
  const int grid = tiles <= n_cu ? tiles : n_cu;
  // gemm_narrow: 1 = narrow tiles + balanced walk, 2 = narrow tiles on the fixed walk (A/B), 0 = neither
  const bool has_narrow_col = A4pNarrow<EPI>::has && a.narrow && a.K / BK >= A4pNarrow<EPI>::MIN_NK && a.K / BK <= A4P_NARROW_MAX_NK && tiles_n * 256 - a.N >= 128;
  balance_rounds(tiles_m, tiles_n, a.band, grid, has_narrow_col && a.narrow == 1, a.rot);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), a4p_lds<EPI>(), st, a, tiles_m, tiles_n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

// -2: this (epilogue, geometry) is not served by the pipelined kernel
int gvl_launch_gemm_a4p(const GemmArgs& a, int epi, hipStream_t st) {
  if (a.K % BK != 0) return -2;
  if (((size_t)a.N + 256) * (size_t)a.ldw * 2 >= (1ull << 32) || ((size_t)a.M + 256) * (size_t)a.lda * 2 >= (1ull << 32)) return -2;   // 32-bit buffer offsets
  if ((size_t)a.ldc * 2 * 136 >= (1ull << 31) || (a.resid && (size_t)a.ldr * 2 * 136 >= (1ull << 31))) return -2;
  switch (epi) {
#define A4P_CASE(E) case E: return launch_a4p<E>(a, st);
    GVL_A4P_EPI_LIST(A4P_CASE)
#undef A4P_CASE
    default: return -2;
  }
}
