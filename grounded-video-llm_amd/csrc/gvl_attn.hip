// gvl_attn.hip -- LDS-staged flash attention (vision + LLM prefill) and paged-KV decode attention, gfx950.
//
// Replaces: eager bmm/softmax/bmm in CLIPAttention (models/modeling_clip.py:274-314),
// flash_attn_varlen_qkvpacked_func in InternVideo2 (models/internvideo2.py:514-517, naive :575-580),
// flash_attn_func(causal=True) / eager attention in Phi-3 / Llama (models/modeling_phi3.py:575-594,857-864;
// models/modeling_llama.py:316-399) and the per-token cached attention of generate().
//
// Operand layouts (written by qkv_post in gvl_elem.hip):
//   Q  [B][H][S][D]                      D = head dim padded to a multiple of 32 (88 -> 96, pad = 0)
//   K  pages [page][KV][64][D]           one page = 64 consecutive tokens = one key tile
//   V^T pages [page][KV][D][64]          transposed inside the page: the P.V contraction (over keys)
//                                        then reads K-contiguous rows exactly like a GEMM operand.
// The same page layout is the KV cache of the LLM (block_table != null) and the scratch K/V of the
// vision towers (block_table == null, page(b,t) = b*n_tiles + t).
//
// Math per 64-key tile and 32-query wave (v_mfma_f32_32x32x16_bf16):
//   S^T[key,q] = K . Q^T     (A = K rows from LDS, B = Q from registers)
//   online softmax on the lane's 32 scores (the lane owns ONE query: max/sum are lane-local plus one
//   exchange with lane^32)
//   O^T[d,q] += V^T[d,keys] . P^T[keys,q]   (A = V^T rows from LDS, B = P straight from the score
//   registers: K rows are read in a permuted order so that the 8 scores a lane holds per 16-key step
//   are exactly the 8 k-slots the MFMA B operand wants -- no cross-lane movement of P at all).
#include "gvl_internal.h"
#include <cstdlib>

template <int D> struct KSwz;
template <> struct KSwz<64> {   // 128-byte rows: phys = chunk ^ ((row>>1)&7)
  static __device__ __forceinline__ int phys(int row, int lc) { return lc ^ ((row >> 1) & 7); }
  static __device__ __forceinline__ int logical(int row, int pc) { return pc ^ ((row >> 1) & 7); }
};
template <> struct KSwz<96> {   // 192-byte rows (12 chunks): rotate by 3*((row>>2)&3) -- conflict-free, see DESIGN.md
  static __device__ __forceinline__ int phys(int row, int lc) { int p = lc + 3 * ((row >> 2) & 3); return p >= 12 ? p - 12 : p; }
  static __device__ __forceinline__ int logical(int row, int pc) { int l = pc - 3 * ((row >> 2) & 3); return l < 0 ? l + 12 : l; }
};
template <> struct KSwz<128> {  // 256-byte rows: phys = chunk ^ (row&15)
  static __device__ __forceinline__ int phys(int row, int lc) { return lc ^ (row & 15); }
  static __device__ __forceinline__ int logical(int row, int pc) { return pc ^ (row & 15); }
};

// Row-major V image [64 keys][D] (VROW: the vision towers read V straight out of the fused-qkv GEMM output, no V^T pass): the P.V A operand
// (lane (d, h) holds 8 consecutive KEYS of column d) is gathered by two ds_read_b64_tr_b16 -- each 16-lane group reads a [4 keys][16 d] block,
// 4 contiguous d per lane, and receives it transposed (cdna_hip_programming.md T10).  One LDS cycle serves a 32-lane half: 4 key rows x 64
// bytes, which must cover all 64 banks -> per-D chunk swizzle (involutions; invariant under row += 4, so a lane's base address is reused
// with immediates for every 16-key step):  D = 96 (192-byte rows) needs none: rows 4j..4j+3 start at 0, 192, 128, 64 (mod 256).
template <int D> struct VSwz;
template <> struct VSwz<64> { static __device__ __forceinline__ int phys(int row, int c) { return c ^ (((row >> 1) & 1) << 2); } };
template <> struct VSwz<96> { static __device__ __forceinline__ int phys(int, int c) { return c; } };
template <> struct VSwz<128> { static __device__ __forceinline__ int phys(int row, int c) { return c ^ ((row & 3) << 2); } };
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
__device__ __forceinline__ bf16x8_t lds_tr8(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4_t* lp;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p0), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)p1);
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// one DMA piece from a wave-uniform base + per-lane 32-bit byte offset (callers may predicate it per lane: inactive lanes leave their
// 16 bytes of the lane-linear LDS image untouched)
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// row i of the S^T MFMA reads key kperm(i) of the 32-key block, so that accumulator register r of
// lane (q, h) holds key (r>>3)*16 + 8*h + (r&7).
__device__ __forceinline__ int kperm(int i) {
  const int c = i & 3, hh = (i >> 2) & 1, b = i >> 3;
  return (b >> 1) * 16 + 8 * hh + 4 * (b & 1) + c;
}

// ONES: the V^T pad row Dout is all ones (D = 96, Dout = 88: InternVideo2), so O^T[Dout] accumulates the softmax row sum inside the
// P.V MFMAs the kernel issues anyway -- the 32 VALU adds per key tile of the VALU-bound loop are dropped (the sum then runs over the
// bf16-rounded probabilities the P.V product uses, in fp32).
// VROW = 3: V rows + q rows normalised in the prologue (a.q_rs / a.q_nw: InternVideo2), K pages.
// VROW = 2: Q and K are token rows too (a.Qrows / a.Krows; needs D == Dout and no per-token transform of q / k: CLIP) -- no qkv_post pass at all.
// VROW: V comes as token rows of a row-major matrix (a.Vrows, row stride a.v_ld, head h at column h * Dout) instead of V^T pages; the pad
// columns Dout..D-1 of the LDS image are written once by the kernel (1.0 in column Dout when ONES) and skipped by the DMA.
// VL = 1 (paged causal prefill only): RAGGED batch in ONE grid -- the launch covers a.vl_n sequences whose rows are packed back to back
// (sequence u: rows [vl_rows[u], vl_rows[u + 1]) of Q / O, its own block table vl_tables[u]); a workgroup id decodes to (sequence, query
// block) through the <= 8 prefix sums, everything after that is the single-sequence kernel on that sequence's (S, Q, O, table) -- the
// same instructions on the same values, so every row is bit-identical to a launch of its sequence alone.
template <int D, int NWAVES, int NS, int ONES = 0, int VROW = 0, int VL = 0>
__global__ __launch_bounds__(NWAVES * 64) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int NT = NWAVES * 64;
  constexpr int DK = D / 16;          // k-steps of the QK^T contraction
  constexpr int DB = D / 32;          // 32-row d blocks of O^T
  constexpr int CPR = D / 8;          // 16-byte chunks per K row
  constexpr int KCH = 64 * CPR;       // chunks in a K tile (== chunks in a V^T tile)
  constexpr int NIK = KCH / NT;       // DMA instructions per thread for K (and for V^T)
  constexpr int TILE_BYTES = 64 * D * 2;
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  static_assert(KCH % NT == 0, "tile/threads mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // XCD-aware block -> (query block, head, batch): consecutive workgroup ids go round-robin over the 8 XCDs, so with the
  // natural (q-block fastest) order the 17 query blocks of one (batch, head) would pull the same K/V pages through 8 different
  // L2s (measured 2.5x fetch amplification).  Here every (batch, head) lives on ONE XCD: id = 8*j + xcd, j = local*nq + qb.
  // (With grouped-query attention the unit is the (batch, KV head) GROUP: its H/KV query heads share the pages too.)
  int S_ = a.S;                                       // queries of THIS block's sequence (VL: decoded below)
  const bf16_t* Qp_ = a.Q; bf16_t* Op_ = a.O; const int* tbl_ = a.block_table;
  int nq = (a.S + NWAVES * 32 - 1) / (NWAVES * 32);
  const int rep = a.H / a.KV;
  if constexpr (VL) { nq = 0; for (int u = 0; u < a.vl_n; ++u) nq += (a.vl_rows[u + 1] - a.vl_rows[u] + NWAVES * 32 - 1) / (NWAVES * 32); }   // query blocks of all sequences
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int qb_ = j % nq, t_ = j / nq, member = t_ % rep, grp = (t_ / rep) * 8 + xcd;
  int qb = a.causal ? nq - 1 - qb_ : qb_;           // causal: the late query blocks see the most keys -- dispatch them FIRST, the short ones fill the tail
  if (grp >= a.KV * a.B) return;                     // grid is padded to a multiple of 8 groups; uniform per block
  const int b = grp / a.KV, head = (grp - b * a.KV) * rep + member;
  if constexpr (VL) {                                // flattened (sequence-major) query block -> (sequence u, block of u); wave-uniform scalar loop
    int u = 0, nqu = (a.vl_rows[1] - a.vl_rows[0] + NWAVES * 32 - 1) / (NWAVES * 32);
    while (qb >= nqu && u + 1 < a.vl_n) { qb -= nqu; ++u; nqu = (a.vl_rows[u + 1] - a.vl_rows[u] + NWAVES * 32 - 1) / (NWAVES * 32); }
    const int r0 = a.vl_rows[u];
    S_ = a.vl_rows[u + 1] - r0;
    Qp_ = a.Q + (size_t)r0 * a.H * D;                // sequence u's [H][S_u][D] block starts at its first packed row
    Op_ = a.O + (size_t)r0 * (size_t)(a.H * a.Dout);
    tbl_ = a.vl_tables[u];
  }
  const int hkv = head / (a.H / a.KV);
  const int q0 = qb * (NWAVES * 32);
  const int qw = q0 + wave * 32;
  const int Sk = a.Sk > 0 ? a.Sk : S_, qpos0 = a.qpos0;    // keys of the context; absolute position of query 0 (extend-prefill: the cached prefix comes first)
  const int n_tiles_all = (Sk + 63) >> 6;
  int last_q = q0 + NWAVES * 32 - 1; if (last_q > S_ - 1) last_q = S_ - 1;
  const int n_tiles = a.causal ? ((qpos0 + last_q) >> 6) + 1 : n_tiles_all;

  // ---- DMA source offsets inside a page (elements), loop invariant --------------------------------
  unsigned koff[NIK], voff[NIK];              // BYTE offsets (32-bit): the page base stays in SGPRs, no 64-bit VALU adds per piece
#pragma unroll
  for (int i = 0; i < NIK; ++i) {
    const int pos = i * NT + tid;
    const int kr = pos / CPR, kpc = pos - kr * CPR;
    koff[i] = VROW == 2 ? (unsigned)(kr * a.k_ld + KSwz<D>::logical(kr, kpc) * 8) * 2 : (unsigned)(kr * D + KSwz<D>::logical(kr, kpc) * 8) * 2;
    if constexpr (VROW) {
      const int vr = pos / CPR, lc = VSwz<D>::phys(vr, pos - vr * CPR);
      voff[i] = lc * 8 < a.Dout ? (unsigned)(vr * a.v_ld + lc * 8) * 2 : 0xffffffffu;      // pad chunk: no DMA
    } else {
      const int vr = pos >> 3, vpc = pos & 7;
      voff[i] = (unsigned)(vr * 64 + (vpc ^ ((vr >> 1) & 7)) * 8) * 2;
    }
  }
  // ---- Q fragments (MFMA B operand): lane (q = qw + l31, h) holds d = kk*16 + 8h + 0..7 ----------
  // The loads are only ISSUED here; they are consumed after the first K / V tile has been requested below, so the block pays one memory
  // latency for both instead of two in a row.
  const float sc = a.scale * 1.4426950408889634f;  // scores in log2 units
  // FOLD (VROW = 3, Dout = D - 8: InternVideo2): the softmax shift and scale ride inside the S^T MFMAs.  q is pre-multiplied by `sc` before its
  // one rounding to bf16; K's pad column Dout holds 1.0 (qkv_post, k_ones) and q's pad element Dout holds MINUS the row's reference point,
  // so the accumulators come out as (score - reference) in log2 units and the probabilities are a bare v_exp_f32: the 32 FMAs per key tile
  // of the VALU-bound loop are gone.  The reference only has to be SOME value near the running max (the division by the row sum cancels
  // it), so keeping it bf16-representable costs nothing; it moves (rarely: `lazy`) by rewriting that one q element.
  constexpr bool FOLD = VROW == 3;
  u32x4_t qf[DK];                                   // kept as dwords: FOLD rewrites one of them inside the loop, and hipcc re-packs a loop-carried bf16x8 every iteration
  u32x4_t qraw[DK], qw_[VROW == 3 ? DK : 1];
  float q_rs = 1.f;
  {
    int qi = qw + l31; if (qi > S_ - 1) qi = S_ - 1;
    if constexpr (VROW == 3) {                        // q in place + InternVideo2's full-width RMSNorm (qkv_post_kernel::norm_chunk, same rounding points)
      const bf16_t* qp = a.Qrows + ((size_t)b * S_ + qi) * a.q_ld + head * a.Dout + 8 * h;
      const bf16_t* wp = a.q_nw + head * a.Dout + 8 * h;
      q_rs = a.q_rs[(size_t)b * S_ + qi];
#pragma unroll
      for (int kk = 0; kk < DK; ++kk) {
        const bool real = kk * 16 + 8 * h < a.Dout;   // Dout is a multiple of 8: a chunk is all real or all padding
        qraw[kk] = real ? *(const u32x4_t*)(qp + kk * 16) : u32x4_t{0u, 0u, 0u, 0u};
        qw_[kk] = real ? *(const u32x4_t*)(wp + kk * 16) : u32x4_t{0u, 0u, 0u, 0u};
      }
    } else {
      const bf16_t* qp = VROW == 2 ? a.Qrows + ((size_t)b * S_ + qi) * a.q_ld + head * D + 8 * h : Qp_ + (((size_t)b * a.H + head) * S_ + qi) * D + 8 * h;
#pragma unroll
      for (int kk = 0; kk < DK; ++kk) qraw[kk] = *(const u32x4_t*)(qp + kk * 16);
    }
  }

  // page ids of this (b) row of the block table, staged in LDS once: a per-iteration global load of the table would
  // make hipcc wait vmcnt(0) (draining the DMA ring) every tile
  int* pages_s = (int*)(smem + NS * STAGE_BYTES);
  for (int i = tid; i < n_tiles; i += NT) pages_s[i] = tbl_ ? tbl_[b * a.max_pages + i] : b * n_tiles_all + i;
  if constexpr (VROW) {                              // pad columns of the V image, both ring slots, once
    const int npc = CPR - (a.Dout >> 3);
    for (int i = tid; i < NS * 64 * npc; i += NT) {
      const int slot = i / (64 * npc), r = (i / npc) & 63, lc = (a.Dout >> 3) + i % npc;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (ONES && lc * 8 == a.Dout) v[0] = 0x3F80u;  // bf16 1.0 in column Dout: O^T[Dout] = sum_k P[k]
      *(u32x4_t*)(smem + slot * STAGE_BYTES + TILE_BYTES + (r * CPR + VSwz<D>::phys(r, lc)) * 16) = v;
    }
  }
  __syncthreads();
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  auto stage = [&](int buf, int t) {
    const int pg = __builtin_amdgcn_readfirstlane(pages_s[t]);
    const size_t pb = ((size_t)pg * a.KV + hkv) * (size_t)(64 * D);
    const bf16_t* kp = a.Kt + pb;
    const bf16_t* vp = a.Vt + pb;
    const unsigned base = smem_base + buf * STAGE_BYTES + wave * 1024;
    if constexpr (VROW) {
      const int rows_left = S_ - t * 64;            // wave-uniform; < 64 only in the last tile: rows past the end re-read the last real key row (P = 0 there)
      if constexpr (VROW == 2) {
        const bf16_t* kr_ = a.Krows + ((size_t)b * S_ + (size_t)t * 64) * a.k_ld + hkv * D;
        if (rows_left >= 64) {
          if constexpr (NIK == 2 || NIK == 3) glds16xn<NIK>(kr_, koff, base, NT * 16);
          else {
#pragma unroll
            for (int i = 0; i < NIK; ++i) glds16s(kr_, koff[i], base + i * NT * 16);
          }
        } else {
#pragma unroll
          for (int i = 0; i < NIK; ++i) {
            const int r = (i * NT + tid) / CPR;
            glds16s(kr_, r >= rows_left ? koff[i] - (unsigned)((r - rows_left + 1) * a.k_ld * 2) : koff[i], base + i * NT * 16);
          }
        }
      } else {
        glds16xn<NIK>(kp, koff, base, NT * 16);
      }
      const bf16_t* vr_ = a.Vrows + ((size_t)b * S_ + (size_t)t * 64) * a.v_ld + hkv * a.Dout;
      if (rows_left >= 64) {                         // wave-uniform
#pragma unroll
        for (int i = 0; i < NIK; ++i)
          if (voff[i] != 0xffffffffu) glds16s(vr_, voff[i], base + TILE_BYTES + i * NT * 16);
      } else {
#pragma unroll
        for (int i = 0; i < NIK; ++i) {
          const int r = (i * NT + tid) / CPR;
          const unsigned off = r >= rows_left ? voff[i] - (unsigned)((r - rows_left + 1) * a.v_ld * 2) : voff[i];
          if (voff[i] != 0xffffffffu) glds16s(vr_, off, base + TILE_BYTES + i * NT * 16);
        }
      }
    } else if constexpr (NIK == 2 || NIK == 3) {
      glds16xn<NIK>(kp, koff, base, NT * 16);
      glds16xn<NIK>(vp, voff, base + TILE_BYTES, NT * 16);
    } else {
#pragma unroll
      for (int i = 0; i < NIK; ++i) glds16((const char*)kp + koff[i], base + i * NT * 16);
#pragma unroll
      for (int i = 0; i < NIK; ++i) glds16((const char*)vp + voff[i], base + TILE_BYTES + i * NT * 16);
    }
  };

  f32x16_t o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // K fragment rows (permuted) and their swizzled chunk offsets; V^T fragment rows
  const int krow0 = kperm(l31);
  const int vswz = (l31 >> 1) & 7;
  const int my_q = qw + l31;
  unsigned vtr[DB];                                    // VROW: byte offset of this lane's 4-element run in the V image, per 32-column block
  if constexpr (VROW) {
    const int i16 = lane & 15, g4 = (lane >> 4) & 1, r0 = 8 * h + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < DB; ++db) vtr[db] = (unsigned)(r0 * (D * 2) + VSwz<D>::phys(r0, db * 4 + 2 * g4 + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8);
  }

  // NS-deep LDS ring: tile t+NS-1 is requested while tile t is consumed, so a DMA has NS-1 iterations to land.
  // Counted vmcnt (never 0 in steady state) + raw s_barrier: __syncthreads() would drain the DMA queue (guide §5).
  stage(0, 0);
  if (NS == 3 && n_tiles > 1) stage(1, 1);
#pragma unroll
  for (int kk = 0; kk < DK; ++kk) {
    if constexpr (VROW == 3) {
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(qw_[kk][e]) * rbf(lo_bf(qraw[kk][e]) * q_rs) * sc, hi_bf(qw_[kk][e]) * rbf(hi_bf(qraw[kk][e]) * q_rs) * sc);   // FOLD: scores come out in log2 units
      qf[kk] = o;
    } else {
      qf[kk] = qraw[kk];
    }
  }
  // make hipcc retire the q loads HERE: otherwise its scoreboard keeps them pending around the loop back-edge and emits vmcnt(5..0)
  // waits inside every iteration, which (in hardware) also drain our un-counted DMA ring
#pragma unroll
  for (int kk = 0; kk < DK; ++kk) asm volatile("" ::"v"(qf[kk]));
  int cur = 0;
  int first_tile = 1;                                      // FOLD: the first tile sets the reference to its own max (whatever its sign)
  if constexpr (FOLD) m_run = 0.f;
  for (int t = 0; t < n_tiles; ++t) {
    if (NS == 3 && t + 1 < n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIK) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef GVL_ATTN_LAB                  // LAB (wrong results, timing only): bit 0 no K/V DMA after the first tile, 1 exp2 -> one FMA, 2 no S^T MFMAs / reads, 3 no P.V MFMAs / reads
    if (!(GVL_ATTN_LAB & 1))
#endif
    if (t + NS - 1 < n_tiles) { int nb = cur + NS - 1; if (nb >= NS) nb -= NS; stage(nb, t + NS - 1); }
    const int tb = cur;
    cur = cur + 1 == NS ? 0 : cur + 1;
    if (a.causal && t * 64 > qpos0 + qw + 31) continue;   // wave-uniform: every key of this tile is in the future
    // (Letting the waves of the last query block that hold no real query -- 3 of 4 at S = 2049 -- skip the tile body was measured: the extra
    //  wave-uniform branch costs 16 VGPRs (172: 2 blocks / CU; also when folded into the causal skip's own compare) or, capped at 168 by __launch_bounds__, a worse schedule: +4 % either way.)
    const char* kb_ = smem + tb * STAGE_BYTES;
    const char* vb_ = kb_ + TILE_BYTES;

    // ---- S^T = K . Q^T ---------------------------------------------------------------------------
    f32x16_t s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kb][e] = 0.f;
#ifdef GVL_ATTN_LAB
    if (GVL_ATTN_LAB & 4) { s[0][0] = (float)t; s[1][3] = qf[0][0]; } else
#endif
#ifdef GVL_ATTN_PRIO
    if (GVL_ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) {            // kk outer: the two accumulators alternate, no back-to-back dependent MFMAs
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + krow0;
        const bf16x8_t kf = *(const bf16x8_t*)(kb_ + row * (D * 2) + (KSwz<D>::phys(row, kk * 2 + h) << 4));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8_t, qf[kk]), s[kb], 0, 0, 0);
      }
    }
#ifdef GVL_ATTN_PRIO
    if (GVL_ATTN_PRIO & 1) __builtin_amdgcn_s_setprio(0);
#endif
    // ---- online softmax (lane-local; raw-score max, scale folded into the exp2 argument) ---------------
    const bool need_mask = (t == n_tiles_all - 1 && (Sk & 63)) || (a.causal && t * 64 + 63 > qpos0 + qw);
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kb * 32 + (r >> 3) * 16 + 8 * h + (r & 7);
          const bool dead = key >= Sk || (a.causal && key > qpos0 + my_q);
          s[kb][r] = dead ? -1e30f : s[kb][r];
        }
    }
    // row max of the lane's 32 scores: a depth-4 tree of 16 v_max3_f32 (fmaxf() makes hipcc quiet every MFMA result with a v_max_f32 x, x first --
    // 8 extra instructions per tile -- and chains the 15 max3 it then emits)
    float mx;
    {
      // vmax3() is an `asm` statement: hipcc pads the XDL-write -> VALU-read distance (8-pass MFMA: 11 wait states) for its own instructions only, and the
      // hardware does not interlock it (round 4: the same tree straight behind the S^T MFMAs of attn_iv2_pipe_kernel's first tile read stale accumulators on
      // some builds).  Here two taken branches used to sit in between by luck; the wait is now explicit -- tied to the accumulators so that it cannot move.
      asm volatile("s_nop 11" : "+v"(s[0]), "+v"(s[1]));
      float l1[11];
#pragma unroll
      for (int i = 0; i < 5; ++i) { l1[i] = vmax3(s[0][3 * i], s[0][3 * i + 1], s[0][3 * i + 2]); l1[5 + i] = vmax3(s[1][3 * i], s[1][3 * i + 1], s[1][3 * i + 2]); }
      l1[10] = vmax3(s[0][15], s[1][15], l1[0]);
      const float a2 = vmax3(l1[1], l1[2], l1[3]), b2 = vmax3(l1[4], l1[5], l1[6]), c2 = vmax3(l1[7], l1[8], l1[9]);
      mx = vmax3(vmax3(a2, b2, c2), l1[10], l1[10]);
    }
    {   // exchange with the partner half-wave (lane ^ 32) in the VALU: v_permlane32_swap, no LDS round trip
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    // Lazy reference point: m_run is only a reference for the exponentials (any value >= true max - 126 gives the same softmax after the
    // final division), so it -- and with it the 16 * DB accumulator multiplies, a quarter of the VALU work of a tile -- is moved only
    // when some row of the wave sees a tile max more than a.lazy (8 = a factor 256) above it: P stays <= 256, exact in bf16's range,
    // and after the first tile of a row almost never moves again.  (Moving it whenever any row's max grew at all skipped 12 % of the
    // rescales at S = 2049: with 32 rows per wave some row nearly always grows a little.)
    float alpha = 1.f;
    if constexpr (FOLD) {
      // m_run = the reference (bf16-representable, log2 units, starts at 0); mx = tile max RELATIVE to it
      if (first_tile || !__all(mx <= a.lazy)) {
        const float m_new = rbf(m_run + (first_tile ? mx : fmaxf(mx, 0.f)));
        const float de = m_new - m_run;
        alpha = first_tile ? 1.f : __builtin_amdgcn_exp2f(-de);      // first tile: O is still 0 -- and 0 x exp2(+huge) = NaN for a row whose first tile is far below 0
        m_run = m_new;
        first_tile = 0;
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] -= de;      // this tile's scores were taken against the old reference
        qf[DK - 1][0] = h ? (unsigned)f2bf(-m_run) : qf[DK - 1][0];   // element Dout of q: the h = 1 lane's last chunk, low half of dword 0 (Dout + 1 is padding: 0)
      }
    } else
    if (!__all((mx - m_run) * sc <= a.lazy)) {   // wave-uniform: some row's tile max is far above its reference -> move it, rescale O
      const float m_new = fmaxf(m_run, mx);
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
    }
    const float nm = -m_run * sc;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
#ifdef GVL_ATTN_LAB
        const float p = (GVL_ATTN_LAB & 2) ? fmaf(s[kb][r], sc, nm) : ((GVL_ATTN_LAB & 16) ? __builtin_amdgcn_exp2f(s[kb][r]) : __builtin_amdgcn_exp2f(fmaf(s[kb][r], sc, nm)));   // bit 4: exp2 without the scale / shift FMA
#else
        const float p = FOLD ? __builtin_amdgcn_exp2f(s[kb][r]) : __builtin_amdgcn_exp2f(fmaf(s[kb][r], sc, nm));
#endif
        s[kb][r] = p;
        if constexpr (!ONES) psum += p;
      }
    if constexpr (!ONES) l_run = l_run * alpha + psum;
    // ---- O^T += V^T . P^T ---------------------------------------------------------------------------
#ifdef GVL_ATTN_LAB
    if (GVL_ATTN_LAB & 8) { o[0][0] += s[0][1] + s[1][2]; } else
#endif
#ifdef GVL_ATTN_PRIO
    if (GVL_ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int kb = st >> 1, r0 = (st & 1) * 8;
      union { bf16x8_t v; unsigned u[4]; } pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) pf.u[e] = pack2bf(s[kb][r0 + 2 * e], s[kb][r0 + 2 * e + 1]);
      const int coff = ((st * 2 + h) ^ vswz) << 4;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        bf16x8_t vf;
        if constexpr (VROW) { const char* p = vb_ + vtr[db] + st * (16 * D * 2); vf = lds_tr8(p, p + 4 * D * 2); }
        else vf = *(const bf16x8_t*)(vb_ + (db * 32 + l31) * 128 + coff);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf.v, o[db], 0, 0, 0);
      }
    }
#ifdef GVL_ATTN_PRIO
    if (GVL_ATTN_PRIO & 2) __builtin_amdgcn_s_setprio(0);
#endif
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  float l_tot;
  if constexpr (ONES) {
    // row Dout of O^T: block DB-1, local row lr = Dout - 32 (DB-1); MFMA layout row = (reg & 3) + 8 (reg >> 2) + 4 h -> held by the
    // h = 0 lane of the query in register (lr / 8) * 4 + lr % 4 (the launcher only selects ONES when lr % 8 < 4)
    const int lr = a.Dout - 32 * (DB - 1);
    float mine = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mine = (e == (lr >> 3) * 4 + (lr & 3)) ? o[DB - 1][e] : mine;
    const float other = __shfl_xor(mine, 32, 64);
    l_tot = h ? other : mine;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.f / l_tot;
  // Row-per-lane store, widened (MI355X guide T21): lanes l and l+32 hold columns 8g..8g+3 / 8g+4..8g+7 of the SAME row, so
  // one v_permlane32_swap per dword pairs the column groups (g, g+1): afterwards the lower half-wave owns all 8 columns of
  // group g and the upper half-wave those of group g+1 -> one 16-byte store per lane per pair instead of two 8-byte ones.
  {
    const int qs = my_q < S_ ? my_q : S_ - 1;
    char* op = (char*)(Op_ + ((size_t)b * S_ + qs) * (size_t)(a.H * a.Dout) + head * a.Dout);
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int g0 = 2 * gp;
        unsigned ax = pack2bf(o[db][4 * g0] * inv, o[db][4 * g0 + 1] * inv), ay = pack2bf(o[db][4 * g0 + 2] * inv, o[db][4 * g0 + 3] * inv);
        unsigned bx = pack2bf(o[db][4 * g0 + 4] * inv, o[db][4 * g0 + 5] * inv), by = pack2bf(o[db][4 * g0 + 6] * inv, o[db][4 * g0 + 7] * inv);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false); ax = rx[0]; bx = rx[1];
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false); ay = ry[0]; by = ry[1];
        const int grp = db * 4 + g0 + h;           // column group this lane now owns entirely
        if (my_q < S_ && grp * 8 < a.Dout) {
          const u32x4_t w = {ax, ay, bx, by};
          *(u32x4_t*)(op + grp * 16) = w;
        }
      }
  }
}
// =====================================================================================================
// attn_iv2_pipe_kernel -- the InternVideo2 shape (head dim 88 -> 96, q in place + folded softmax, ones-row sum, non-causal, no block table)
// with a HAND-PLACED, software-pipelined key-tile loop (round 4).
//
// attn_fwd_kernel runs a tile as three serial stretches -- 12 S^T MFMAs (matrix pipe only), the row max (VALU only), 32 exp2 + 16 packs
// feeding 12 P.V MFMAs -- and relies on three resident waves per SIMD to overlap them: PMC showed the matrix pipe 0.64 busy, the two
// pipes adding up instead of overlapping (profiles/r03_attention_pmc.txt).  Here ONE wave's own instruction stream keeps both pipes fed:
//   phase 1:  S^T(t+1) = K(t+1).Q^T   12 MFMAs   ||   P(t) = exp2(S(t)), packed to bf16   (32 v_exp_f32 + 16 v_cvt_pk) + the 12 K fragment reads
//   phase 2:  O^T += V(t).P(t)        12 MFMAs   ||   row max of S(t+1)                    (16 v_max3 + the half-wave exchange) + the 24 V tr-reads
// Every MFMA heads a GROUP {MFMA, <= 5 single-issue fillers} closed by __builtin_amdgcn_sched_barrier(0): hipcc may not move anything
// across a group boundary, so the emitted stream IS this listing (its sched_group_barrier requests were honoured only partially in
// round 2), while hipcc still owns register allocation, s_waitcnt counts and the hazard nops.  A 32x32x16 MFMA occupies the matrix pipe
// for 32 cycles = 8 issue slots; <= 5 fillers ride in its shadow (MI355X_MICROARCH.md, cycle constants).  The LDS fragment of MFMA i is
// requested two groups ahead.  The two score tiles swap roles by a 2x unroll (no register copies); the unroll parity also fixes the ring
// slot, so every LDS address is one per-lane register plus an immediate.  Two waves per SIMD (<= 256 VGPRs), 2 blocks of 4 waves per CU.
//
// Ring (2 slots of {K tile, V tile}, 48 KB per block as before), now phase-shifted: iteration t reads K(t+1) and V(t), so after the top
// barrier of iteration t the K half of slot t&1 (K(t), consumed by iteration t-1) takes K(t+2) and the V half of slot (t+1)&1 (V(t-1))
// takes V(t+1); both DMA sets are issued inside phase 1 (under MFMAs) and awaited (vmcnt(0) + barrier) at the top of iteration t+1.
//
// Arithmetic is EXACTLY attn_fwd_kernel<96, 4, 2, 1, 3>'s: same MFMAs on the same values in the same order, same lazy-reference rule at
// the same point of the O accumulation (after tile t's P.V, before tile t+1's) -- the two kernels are bit-identical (asserted:
// gvl_debug_set("attn_pipe", 0 | 1), tests/test_gpu_towers.py).
#define GVL_SB() __builtin_amdgcn_sched_barrier(0)
template <int I> struct IC { static constexpr int v = I; };
// three DMA pieces from one SGPR base, each under its own lane mask (the V image's pad-chunk lanes stay off), in ONE branch-free statement:
// hipcc's own predication is an s_cbranch_execz per piece, i.e. three basic-block splits in the middle of a hand-placed phase
__device__ __forceinline__ void glds16x3_masked(const void* sbase, const unsigned (&v)[3], const unsigned long long (&mask)[3], unsigned lds_dst0, unsigned lds_step) {
  unsigned keep, d;
  unsigned long long ex;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b64 %2, exec\n\t"
      "s_mov_b32 m0, %7\n\ts_and_b64 exec, %2, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
      "s_add_u32 %1, %7, %8\n\ts_mov_b32 m0, %1\n\ts_and_b64 exec, %2, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
      "s_add_u32 %1, %1, %8\n\ts_mov_b32 m0, %1\n\ts_and_b64 exec, %2, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %6\n\t"
      "s_mov_b64 exec, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(d), "=&s"(ex)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "s"(sbase), "s"(lds_dst0), "s"(lds_step), "s"(mask[0]), "s"(mask[1]), "s"(mask[2])
      : "memory", "scc");
}

__device__ __forceinline__ void glds16s_masked(const void* sbase, unsigned v, unsigned long long mask, unsigned lds_dst) {
  unsigned keep;
  unsigned long long ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_and_b64 exec, %1, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(v), "s"(sbase), "s"(lds_dst), "s"(mask) : "memory", "scc");
}

__device__ __forceinline__ void glds16x2_masked(const void* sbase, const unsigned (&v)[2], const unsigned long long (&mask)[2], unsigned lds_dst0, unsigned lds_step) {
  unsigned keep, d;
  unsigned long long ex;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b64 %2, exec\n\t"
      "s_mov_b32 m0, %6\n\ts_and_b64 exec, %2, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
      "s_add_u32 %1, %6, %7\n\ts_mov_b32 m0, %1\n\ts_and_b64 exec, %2, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
      "s_mov_b64 exec, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(d), "=&s"(ex)
      : "v"(v[0]), "v"(v[1]), "s"(sbase), "s"(lds_dst0), "s"(lds_step), "s"(mask[0]), "s"(mask[1])
      : "memory", "scc");
}

// NW = 4 (shipped): 128 query rows per block, two blocks per CU.  NW = 8 (opt-in, measured 12 % slower: see the launcher): 256 query rows per block, ONE
// block of eight waves per CU -- every fetched K / V tile serves twice the MFMAs: 3 instead of 6 DMA pieces per wave and tile.
// A launch covers the query rows [a.q_begin, a.q_begin + a.q_rows) of every (batch, head).
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_iv2_pipe_kernel(const AttnArgs a) {
  constexpr int D = 96, NT = NW * 64, DK = 6, DB = 3, CPR = 12, NIK = (64 * CPR + NT - 1) / NT, TILE_BYTES = 64 * D * 2, STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int QB = NW * 32;                          // query rows per block
  constexpr int VLAST = 2 * STAGE_BYTES;               // a third V slot that only ever holds the LAST key tile when it is partial
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // XCD-aware block -> (query block, head, batch) exactly as attn_fwd_kernel (H == KV here)
  const int q_end = a.q_begin + a.q_rows;              // this launch's query window
  const int nq = (a.q_rows + QB - 1) / QB;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int qb = j % nq, grp = (j / nq) * 8 + xcd;
  if (grp >= a.KV * a.B) return;
  const int b = grp / a.KV, head = grp - b * a.KV;
  const int qw = a.q_begin + qb * QB + wave * 32;
  const int S = a.S, n_tiles = (S + 63) >> 6, v_ld = a.v_ld, Dout = a.Dout;
  const bool partial = (S & 63) != 0;
  const int tail_keys = partial ? (S & 63) : 64;                 // real keys of the last tile
  const int last_full = partial ? n_tiles - 2 : n_tiles - 1;     // -1: the only tile is partial

  unsigned koff[NIK], voff[NIK];
  unsigned long long kmask[NIK], vmask[NIK];           // lanes of piece i that carry a chunk of the tile (NW = 8: the second piece belongs to waves 0-3 only) / a non-pad chunk
#pragma unroll
  for (int i = 0; i < NIK; ++i) {
    const int pos0 = i * NT + tid, pos = pos0 < 64 * CPR ? pos0 : 64 * CPR - 1, r = pos / CPR, c = pos - r * CPR;
    koff[i] = (unsigned)(r * D + KSwz<D>::logical(r, c) * 8) * 2;
    voff[i] = (unsigned)(r * v_ld + c * 8) * 2;
    kmask[i] = __ballot(pos0 < 64 * CPR);
    vmask[i] = __ballot(pos0 < 64 * CPR && c * 8 < Dout);     // pad chunk: no DMA (VSwz<96> is the identity)
  }
  // q fragments: issued here, consumed after the first tiles are requested (one memory latency for both)
  const float sc = a.scale * 1.4426950408889634f;
  u32x4_t qf[DK], qraw[DK], qwt[DK];
  float q_rs;
  {
    int qi = qw + l31; if (qi > S - 1) qi = S - 1;
    const bf16_t* qp = a.Qrows + ((size_t)b * S + qi) * a.q_ld + head * Dout + 8 * h;
    const bf16_t* wp = a.q_nw + head * Dout + 8 * h;
    q_rs = a.q_rs[(size_t)b * S + qi];
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) {
      const bool real = kk * 16 + 8 * h < Dout;
      qraw[kk] = real ? *(const u32x4_t*)(qp + kk * 16) : u32x4_t{0u, 0u, 0u, 0u};
      qwt[kk] = real ? *(const u32x4_t*)(wp + kk * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  {                                                    // pad columns of the V image (1.0 in column Dout: the row sum), all three V slots, once
    const int npc = CPR - (Dout >> 3);
    for (int i = tid; i < 3 * 64 * npc; i += NT) {
      const int slot = i / (64 * npc), r = (i / npc) & 63, lc = (Dout >> 3) + i % npc;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (lc * 8 == Dout) v[0] = 0x3F80u;
      *(u32x4_t*)(smem + (slot < 2 ? slot * STAGE_BYTES + TILE_BYTES : VLAST) + (r * CPR + lc) * 16) = v;
    }
  }
  __syncthreads();
  const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const bf16_t* kbase = a.Kt + ((size_t)b * n_tiles * a.KV + head) * (size_t)(64 * D);       // page (b, t) = b * n_tiles + t
  const size_t kstep = (size_t)a.KV * (64 * D);
  const bf16_t* vbase = a.Vrows + (size_t)b * S * v_ld + head * Dout;
  auto stage_k = [&](int slot, int t) {
    if constexpr (NIK == 3) glds16xn<NIK>(kbase + (size_t)t * kstep, koff, smem_base + slot * STAGE_BYTES + wave * 1024, NT * 16);
    else glds16x2_masked(kbase + (size_t)t * kstep, koff, kmask, smem_base + slot * STAGE_BYTES + wave * 1024, NT * 16);
  };
  auto stage_v = [&](int slot, int t) {                // FULL tiles only
    if constexpr (NIK == 3) glds16x3_masked(vbase + (size_t)t * 64 * v_ld, voff, vmask, smem_base + slot * STAGE_BYTES + TILE_BYTES + wave * 1024, NT * 16);
    else glds16x2_masked(vbase + (size_t)t * 64 * v_ld, voff, vmask, smem_base + slot * STAGE_BYTES + TILE_BYTES + wave * 1024, NT * 16);
  };

  f32x16_t o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
  float m_run = 0.f;                                   // the reference point (bf16-representable, log2 units)

  // per-lane LDS offsets: K fragment rows (permuted, swizzled) per k step; V image runs per 32-column block
  const int krow0 = kperm(l31);
  unsigned kfo[DK], vtr[DB];
#pragma unroll
  for (int kk = 0; kk < DK; ++kk) kfo[kk] = smem_base + (unsigned)(krow0 * (D * 2) + (KSwz<D>::phys(krow0, kk * 2 + h) << 4));
  {
    const int i16 = lane & 15, g4 = (lane >> 4) & 1, r0 = 8 * h + (i16 >> 2);
#pragma unroll
    for (int db = 0; db < DB; ++db) vtr[db] = smem_base + (unsigned)(r0 * (D * 2) + (db * 4 + 2 * g4 + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8);
  }
  const int my_q = qw + l31;

  auto stage_first = [&]() {
    stage_k(0, 0);
    if (last_full >= 0) stage_v(0, 0);
    if (n_tiles > 1) stage_k(1, 1);
    if (partial) {                                       // the partial last tile goes to its own slot now: rows past the end re-read the last real key row (their P is 0)
      const int tl = n_tiles - 1, rows_left = S - tl * 64;
      const bf16_t* vr_ = vbase + (size_t)tl * 64 * v_ld;
#pragma unroll
      for (int i = 0; i < NIK; ++i) {
        const int pos = i * NT + tid, r = pos / CPR, c = pos - r * CPR;
        const unsigned off = r >= rows_left ? voff[i] - (unsigned)((r - rows_left + 1) * v_ld * 2) : voff[i];
        if (pos < 64 * CPR && c * 8 < Dout) glds16s(vr_, off, smem_base + VLAST + wave * 1024 + i * NT * 16);
      }
    }
  };
  stage_first();
#ifndef GVL_PIPE_LAB                 // (LAB builds drop barriers / the second pass from the main path: keep every wave on it)
  if (qw >= q_end) {
    // A wave without a single real query (3 of the 4 waves of the last query block at S = 2049: 4.4 % of all wave-tiles): it owes the block its
    // share of every DMA and every barrier, nothing else -- no fragment reads, no MFMAs, no softmax.  (attn_fwd_kernel could not afford the
    // branch inside its tile body; here the idle waves run their own loop.)  The barrier sequence mirrors run() exactly.
    for (int pass = a.pipe == 2 ? 1 : 0;; ++pass) {     // (pipe == 2, tests: the safe pass only)
      if (pass) stage_first();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // tile 0
      for (int t = 0; t + 1 < n_tiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_k(t & 1, t + 2 < n_tiles ? t + 2 : n_tiles - 1);
        stage_v(1 - (t & 1), t + 1 <= last_full ? t + 1 : last_full);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // last tile
      if (pass || !__syncthreads_or(0)) break;          // the block repeats the pass in safe mode: once more
    }
    return;
  }
#endif
#pragma unroll
  for (int kk = 0; kk < DK; ++kk) {
    u32x4_t t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = pack2bf(lo_bf(qwt[kk][e]) * rbf(lo_bf(qraw[kk][e]) * q_rs) * sc, hi_bf(qwt[kk][e]) * rbf(hi_bf(qraw[kk][e]) * q_rs) * sc);
    qf[kk] = t;
  }
#pragma unroll
  for (int kk = 0; kk < DK; ++kk) asm volatile("" ::"v"(qf[kk]));      // retire the q loads here (see attn_fwd_kernel)

  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // (kfo / vtr hold ABSOLUTE LDS byte addresses: with the dynamic-LDS base inside the per-lane register every read is register + immediate;
  //  a `smem + constant + offset` expression costs one v_add per read once the kernel also owns static LDS)
  typedef __attribute__((address_space(3))) const bf16x8_t* lds_b128_p;
  typedef __attribute__((address_space(3))) s16x4_t* lds_tr_p;
  auto kfrag = [&](int slot, int i) -> bf16x8_t {       // A operand of S^T MFMA i: k step i >> 1, key block i & 1
    return *(lds_b128_p)(size_t)(kfo[i >> 1] + (unsigned)(slot * STAGE_BYTES + (i & 1) * (32 * D * 2)));
  };
  auto vfrag = [&](int vofs, int i) -> bf16x8_t {       // A operand of P.V MFMA i: 16-key step i / 3, d block i % 3; vofs = byte offset of the V slot
    const unsigned p = vtr[i % 3] + (unsigned)(vofs + (i / 3) * (16 * D * 2));
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(size_t)p), hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(size_t)(p + 4 * D * 2));
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto mask_tail = [&](f32x16_t (&s)[2], int t) {       // keys past the end of the last tile
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * 64 + kb * 32 + (r >> 3) * 16 + 8 * h + (r & 7);
        s[kb][r] = key >= S ? -1e30f : s[kb][r];
      }
  };
  // the reference point moves: rescale O, shift the pending score tile, rewrite q's pad element (element Dout of q = -reference)
  auto move_ref = [&](f32x16_t (&s)[2], float mx, bool first) {
    const float m_new = rbf(m_run + (first ? mx : fmaxf(mx, 0.f)));
    const float de = m_new - m_run;
    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-de);      // first tile: O is still 0 (and 0 x exp2(+huge) would be NaN)
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][e] *= alpha;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] -= de;
    qf[DK - 1][0] = h ? (unsigned)f2bf(-m_run) : qf[DK - 1][0];
  };
  // tile 0's row max, written with operations hipcc SEES: the reader of an MFMA result has to keep its distance from the MFMA (XDL write -> VALU read),
  // hipcc pads that for its own instructions but not for the operands of an `asm` statement (cdna_hip_programming.md §5.7) -- vmax3() straight behind
  // the twelve S^T MFMAs read the accumulators too early on some builds (a reference point made of stale registers: results correct to rounding,
  // but not reproducible).  Inside the loop the asm form is safe by construction: two MFMA groups lie between a score tile's last MFMA and its first reader.
  auto row_max = [&](const f32x16_t (&s)[2]) -> float {
    float m0 = fmaxf(s[0][0], s[1][0]), m1 = fmaxf(s[0][1], s[1][1]), m2 = fmaxf(s[0][2], s[1][2]), m3 = fmaxf(s[0][3], s[1][3]);
#pragma unroll
    for (int e = 4; e < 16; e += 4) {
      m0 = fmaxf(m0, fmaxf(s[0][e], s[1][e])); m1 = fmaxf(m1, fmaxf(s[0][e + 1], s[1][e + 1]));
      m2 = fmaxf(m2, fmaxf(s[0][e + 2], s[1][e + 2])); m3 = fmaxf(m3, fmaxf(s[0][e + 3], s[1][e + 3]));
    }
    const float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };
  // One pass over the key tiles.  SAFE = 0 (the normal pass): the reference point is the FIRST tile's row maximum and never moves -- no
  // per-tile row max at all, the loop's VALU work is the 32 exp2 + 16 packs, spread over all 24 MFMA groups.  Exact as long as no later
  // score exceeds the first tile's maximum by ~2^100 (fp32 headroom of P, of the row sum and of O; the final division cancels the
  // reference, and floating point keeps the relative precision of P whatever its scale); attn_fwd_kernel's own rule (move when a tile
  // exceeds the reference by 2^8) never fires after tile 0 on such data, so the two kernels agree bit for bit there.  A row sum that is
  // not a sane positive number afterwards (inf / NaN / 0: a score ran > 100 log2 units past the reference) makes the BLOCK repeat the
  // pass with SAFE = 1: the per-tile row max and the lazy reference rule of attn_fwd_kernel, bit-identical to it.
  f32x16_t sA[2], sB[2];
  auto run = [&](auto safe_) {
    constexpr bool SAFE = decltype(safe_)::v != 0;
    if constexpr (SAFE) stage_first();                   // (the normal pass: requested before the q prologue, one memory latency for both)
    // ---- tile 0: S^T and its row max the plain way; its maximum becomes the reference ------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sA[0] = zero16; sA[1] = zero16;
#pragma unroll
    for (int i = 0; i < 12; ++i)
      sA[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfrag(0, i), __builtin_bit_cast(bf16x8_t, qf[i >> 1]), sA[i & 1], 0, 0, 0);
    if (n_tiles == 1 && partial) mask_tail(sA, 0);
    move_ref(sA, row_max(sA), true);

    // ---- one pipelined iteration: `sc_` = S(t) (relative to the reference), `sn` receives S(t+1); PAR = t & 1; LAST: t + 1 is the last tile ----
    auto body = [&](auto par, auto last_, int t, f32x16_t (&sc_)[2], f32x16_t (&sn)[2]) {
      constexpr int PAR = decltype(par)::v;
      constexpr bool LAST = decltype(last_)::v != 0;
      constexpr int KS = 1 - PAR, VS = PAR * STAGE_BYTES + TILE_BYTES;      // K(t+1) sits in slot (t+1)&1, V(t) (a full tile) in slot t&1
      // DMA targets of this iteration, clamped so that the statements are unconditional (a redundant re-fetch lands in a slot nobody reads again):
      // K(t+2) -> K half of slot t&1, V(t+1) -> V half of slot (t+1)&1 (a partial last tile already sits in its own slot)
      const int tk = t + 2 < n_tiles ? t + 2 : n_tiles - 1, tv = t + 1 <= last_full ? t + 1 : last_full;
#ifdef GVL_PIPE_LAB                 // LAB (wrong results, timing only): bit 0 no K / V DMA in the loop, 1 no top-of-iteration wait + barrier, 2 no exp2, 3 only two K and two V fragment reads per tile
      if (!(GVL_PIPE_LAB & 2)) {
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K(t+1), V(t) (requested one iteration ago) have landed -- for every wave after the barrier,
      __builtin_amdgcn_s_barrier();                      // which also says: every wave is done reading K(t) and V(t-1)
      asm volatile("" ::: "memory");
#ifdef GVL_PIPE_LAB
      }
#endif
      bf16x8_t kf[12], vf[12];
      unsigned pw[16];
      // the VALU fillers of MFMA group g (0..11 = phase 1, 12..23 = phase 2): chunk c = the 8 scores of P.V k-step c (needed by group 12 + 3c).
      // fast: 12 ops per chunk over 5 groups from group 5c (<= 3 per gap); safe: over 3 groups from group 3c (phase 2 carries the row max)
      auto fill = [&](int g) {
        constexpr int SPAN = SAFE ? 3 : 5;
        const int c = g / SPAN, sub = g - SPAN * c;
        if (c >= 4) return;
        const int kb = c >> 1, r0 = (c & 1) * 8;
        f32x16_t& s = sc_[kb];
#ifdef GVL_PIPE_LAB
        auto ex = [&](int e) { if (!(GVL_PIPE_LAB & 4)) s[r0 + e] = __builtin_amdgcn_exp2f(s[r0 + e]); };
#else
        auto ex = [&](int e) { s[r0 + e] = __builtin_amdgcn_exp2f(s[r0 + e]); };
#endif
        auto pk = [&](int e) { pw[4 * c + e] = pack2bf(s[r0 + 2 * e], s[r0 + 2 * e + 1]); };
        if constexpr (SAFE) {
          if (sub == 0) { ex(0); ex(1); ex(2); }
          else if (sub == 1) { ex(3); ex(4); ex(5); pk(0); }
          else { ex(6); ex(7); pk(1); pk(2); pk(3); }
        } else {
          if (sub == 0) { ex(0); ex(1); ex(2); }
          else if (sub == 1) { ex(3); ex(4); pk(0); }
          else if (sub == 2) { ex(5); ex(6); pk(1); }
          else if (sub == 3) { ex(7); pk(2); }
          else pk(3);
        }
      };
      kf[0] = kfrag(KS, 0); kf[1] = kfrag(KS, 1);
      GVL_SB();
      // phase 1: S^T(t+1) MFMAs
#pragma unroll
      for (int i = 0; i < 12; ++i) {
#ifdef GVL_PIPE_LAB
        if (GVL_PIPE_LAB & 8) { if (i + 2 < 12) kf[i + 2] = kf[i]; } else
#endif
        if (i + 2 < 12) kf[i + 2] = kfrag(KS, i + 2);
        // (the peeled last iteration: a partial last tile of <= 32 keys -- S = 2049: one key -- has nothing but masked scores in its second key block)
        if (!(LAST && (i & 1) && tail_keys <= 32))
          sn[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i], __builtin_bit_cast(bf16x8_t, qf[i >> 1]), i < 2 ? zero16 : sn[i & 1], 0, 0, 0);
        fill(i);
#ifdef GVL_PIPE_LAB
        if ((GVL_PIPE_LAB & 64) && NIK == 3) {                         // one piece per group (groups 1..3 K, 5..7 V) instead of two statements of three
          if (i >= 1 && i <= 3) glds16s(kbase + (size_t)tk * kstep, koff[(i - 1) % NIK], smem_base + PAR * STAGE_BYTES + wave * 1024 + (i - 1) * NT * 16);
          if (i >= 5 && i <= 7) glds16s_masked(vbase + (size_t)tv * 64 * v_ld, voff[(i - 5) % NIK], vmask[(i - 5) % NIK], smem_base + (1 - PAR) * STAGE_BYTES + TILE_BYTES + wave * 1024 + (i - 5) * NT * 16);
        } else
        if (!(GVL_PIPE_LAB & 1)) {
        if (i == 1 && !(GVL_PIPE_LAB & 32)) stage_k(PAR, tk);
        if (i == 4 && !(GVL_PIPE_LAB & 16)) stage_v(1 - PAR, tv);
        }
#else
        if (i == 1) stage_k(PAR, tk);                    // the DMA issue rides under the MFMAs
        if (i == 4) stage_v(1 - PAR, tv);                // (a body only runs when n_tiles >= 2, i.e. last_full >= 0)
#endif
        if (i == 10) vf[0] = vfrag(VS, 0);
        if (i == 11) vf[1] = vfrag(VS, 1);
        GVL_SB();
      }
      if constexpr (LAST) { if (tail_keys <= 32) sn[1] = zero16; if (partial) mask_tail(sn, t + 1); GVL_SB(); }
      // phase 2: P.V(t) MFMAs (|| row max of S(t+1) in the safe pass)
      float l1[11], a2 = 0.f, b2 = 0.f, c2 = 0.f, mx = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
#ifdef GVL_PIPE_LAB
        if (GVL_PIPE_LAB & 8) { if (i + 2 < 12) vf[i + 2] = vf[i]; } else
#endif
        if (i + 2 < 12) vf[i + 2] = vfrag(VS, i + 2);
        fill(12 + i);
        const int st = i / 3, db = i - 3 * st;
        const u32x4_t pu = {pw[4 * st], pw[4 * st + 1], pw[4 * st + 2], pw[4 * st + 3]};
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i], __builtin_bit_cast(bf16x8_t, pu), o[db], 0, 0, 0);
        if constexpr (SAFE) {
          // 16 v_max3 over groups 1..8 (S(t+1)'s first key block finished one MFMA earlier than the second: start there), exchange in group 10
          if (i == 1) { l1[0] = vmax3(sn[0][0], sn[0][1], sn[0][2]); l1[1] = vmax3(sn[0][3], sn[0][4], sn[0][5]); }
          if (i == 2) { l1[2] = vmax3(sn[0][6], sn[0][7], sn[0][8]); l1[3] = vmax3(sn[0][9], sn[0][10], sn[0][11]); }
          if (i == 3) { l1[4] = vmax3(sn[0][12], sn[0][13], sn[0][14]); l1[5] = vmax3(sn[1][0], sn[1][1], sn[1][2]); }
          if (i == 4) { l1[6] = vmax3(sn[1][3], sn[1][4], sn[1][5]); l1[7] = vmax3(sn[1][6], sn[1][7], sn[1][8]); }
          if (i == 5) { l1[8] = vmax3(sn[1][9], sn[1][10], sn[1][11]); l1[9] = vmax3(sn[1][12], sn[1][13], sn[1][14]); }
          if (i == 6) { l1[10] = vmax3(sn[0][15], sn[1][15], l1[0]); a2 = vmax3(l1[1], l1[2], l1[3]); }
          if (i == 7) { b2 = vmax3(l1[4], l1[5], l1[6]); c2 = vmax3(l1[7], l1[8], l1[9]); }
          if (i == 8) { mx = vmax3(a2, b2, c2); mx = vmax3(mx, l1[10], l1[10]); }
          if (i == 10) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
          }
        }
        GVL_SB();
      }
      // safe pass: the lazy reference (attn_fwd_kernel's rule, at the same point of the accumulation: after tile t's P.V, before tile t+1's)
      if constexpr (SAFE) { if (!__all(mx <= a.lazy)) move_ref(sn, mx, false); }
    };

    // iterations t = 0 .. n_tiles - 2; the last one (t + 1 = the last tile) is peeled: it alone carries the tail mask
    int t = 0;
    for (; t + 2 < n_tiles - 1; t += 2) {
      body(IC<0>{}, IC<0>{}, t, sA, sB);
      body(IC<1>{}, IC<0>{}, t + 1, sB, sA);
    }
    if (t + 2 == n_tiles - 1) {                          // two left
      body(IC<0>{}, IC<0>{}, t, sA, sB);
      body(IC<1>{}, IC<1>{}, t + 1, sB, sA);
    } else if (t + 1 == n_tiles - 1) {                   // one left: S(n_tiles - 1) lands in sB
      body(IC<0>{}, IC<1>{}, t, sA, sB);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sA[kb] = sB[kb];
    }
    // ---- last tile: exp2 + P.V, nothing left to overlap with ---------------------------------------------------------------------
    {
      const int tl = n_tiles - 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int vofs = partial ? VLAST : (tl & 1) * STAGE_BYTES + TILE_BYTES;
      const int nst = (tail_keys + 15) >> 4;             // 16-key steps that hold real keys: the others carry P = exp2(-1e30) = 0, i.e. add exact zeros
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        if (st < nst) {
          const int kb = st >> 1, r0 = (st & 1) * 8;
          union { bf16x8_t v; unsigned u[4]; } pf;
#pragma unroll
          for (int e = 0; e < 4; ++e) pf.u[e] = pack2bf(__builtin_amdgcn_exp2f(sA[kb][r0 + 2 * e]), __builtin_amdgcn_exp2f(sA[kb][r0 + 2 * e + 1]));
#pragma unroll
          for (int db = 0; db < DB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(vofs, st * 3 + db), pf.v, o[db], 0, 0, 0);
        }
      }
    }
  };
  // row Dout of O^T = the softmax row sum (ones column of the V image): held by the h = 0 lane of the query
  auto row_sum = [&]() -> float {
    const int lr = Dout - 32 * (DB - 1);
    float mine = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) mine = (e == (lr >> 3) * 4 + (lr & 3)) ? o[DB - 1][e] : mine;
    const float other = __shfl_xor(mine, 32, 64);
    return h ? other : mine;
  };
  float l_tot = 0.f;
  bool redo = a.pipe == 2;                               // gvl_debug_set("attn_pipe", 2) (tests): the safe pass only
  if (!redo) {
    run(IC<0>{});
    l_tot = row_sum();
#ifndef GVL_PIPE_LAB                                     // (LAB builds compute garbage on purpose: no second pass)
    redo = __syncthreads_or(!(l_tot > 0.f && l_tot < 1.2676506e30f)) != 0;     // 2^100; also catches NaN.  Block-uniform: every wave repeats the pass (the barriers are shared)
#endif
  }
  if (redo) {
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[i][e] = 0.f;
    m_run = 0.f;
    qf[DK - 1][0] = h ? 0u : qf[DK - 1][0];            // q's pad element: reference 0 again
    run(IC<1>{});
    l_tot = row_sum();
  }
  // ---- epilogue (attn_fwd_kernel's ONES branch) ---------------------------------------------------------------------------------
  const float inv = 1.f / l_tot;
  {
    const int qs = my_q < q_end ? my_q : q_end - 1;
    char* op = (char*)(a.O + ((size_t)b * S + qs) * (size_t)(a.H * Dout) + head * Dout);
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int g0 = 2 * gp;
        unsigned ax = pack2bf(o[db][4 * g0] * inv, o[db][4 * g0 + 1] * inv), ay = pack2bf(o[db][4 * g0 + 2] * inv, o[db][4 * g0 + 3] * inv);
        unsigned bx = pack2bf(o[db][4 * g0 + 4] * inv, o[db][4 * g0 + 5] * inv), by = pack2bf(o[db][4 * g0 + 6] * inv, o[db][4 * g0 + 7] * inv);
        const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false); ax = rx[0]; bx = rx[1];
        const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false); ay = ry[0]; by = ry[1];
        const int cg = db * 4 + g0 + h;
        if (my_q < q_end && cg * 8 < Dout) {
          const u32x4_t w = {ax, ay, bx, by};
          *(u32x4_t*)(op + cg * 16) = w;
        }
      }
  }
}
static int launch_attn_iv2_pipe(const AttnArgs& a_in, hipStream_t st) {
  constexpr int LDS = 2 * 2 * 64 * 96 * 2 + 64 * 96 * 2;      // the ring + the partial-last-tile V slot: 60 KB
  static GvlDevOnce once4, once8;
  if (gvl_set_max_lds(once4, (const void*)attn_iv2_pipe_kernel<4>, LDS) || gvl_set_max_lds(once8, (const void*)attn_iv2_pipe_kernel<8>, LDS)) return -3;
  // a.pipe_rows == 256 (gvl_debug_set "attn_pipe_rows", tests / A-B): whole 256-row query blocks go to the 8-wave form (half the DMA pieces per MFMA), the
  // remaining rows (S = 2049: one) to the 4-wave form in a second launch.  MEASURED SLOWER and therefore not the default: 106.5-106.8 against 95.3-95.9 ms
  // per 39 launches, same box (profiles/r04_attention_pipe_lab.txt) -- eight waves in lock-step behind one barrier and one block per CU lose more than the
  // halved DMA issue gives back (round 2 saw the same with 6-wave blocks).  A row's arithmetic does not depend on which form computes it (asserted).
  const int rows8 = a_in.pipe_rows == 256 ? (a_in.S / 256) * 256 : 0;
  const int groups8 = ((a_in.KV * a_in.B + 7) / 8) * 8;
  if (rows8 > 0) {
    AttnArgs a = a_in; a.q_begin = 0; a.q_rows = rows8;
    hipLaunchKernelGGL(attn_iv2_pipe_kernel<8>, dim3((unsigned)(groups8 * (rows8 / 256))), dim3(512), LDS, st, a);
  }
  if (rows8 < a_in.S) {
    AttnArgs a = a_in; a.q_begin = rows8; a.q_rows = a_in.S - rows8;
    hipLaunchKernelGGL(attn_iv2_pipe_kernel<4>, dim3((unsigned)(groups8 * ((a.q_rows + 127) / 128))), dim3(256), LDS, st, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int D, int NWAVES, int NS, int ONES = 0, int VROW = 0, int VL = 0>
static int launch_attn(const AttnArgs& a, hipStream_t st) {
  constexpr int LDS = NS * 2 * 64 * D * 2 + 1024;   // ring + page-id table (256 pages)
  static GvlDevOnce once;
  auto kern = attn_fwd_kernel<D, NWAVES, NS, ONES, VROW, VL>;
  if (gvl_set_max_lds(once, (const void*)kern, LDS)) return -3;
  int nq = (a.S + NWAVES * 32 - 1) / (NWAVES * 32);
  if (VL) { nq = 0; for (int u = 0; u < a.vl_n; ++u) nq += (a.vl_rows[u + 1] - a.vl_rows[u] + NWAVES * 32 - 1) / (NWAVES * 32); }
  dim3 grid((unsigned)(((a.KV * a.B + 7) / 8) * 8 * (a.H / a.KV) * nq));
  hipLaunchKernelGGL(kern, grid, dim3(NWAVES * 64), LDS, st, a);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

double gvl_attn_flops(const AttnArgs& a) {
  // algorithmic: 4*S^2*d*H non-causal, half that causal, d = real head dim
  const double Sk = a.Sk > 0 ? a.Sk : a.S;
  const double f = 4.0 * (double)a.S * Sk * a.Dout * a.H * a.B;
  return a.causal ? f * (a.qpos0 + 0.5 * a.S) / Sk : f;      // causal: query i sees qpos0 + i + 1 keys
}

int gvl_launch_attention(const AttnArgs& a_in, hipStream_t st) {
  AttnArgs a = a_in;
  static const float lazy = [] { const char* e = gvl_lab_env("GVL_ATTN_LAZY"); return e ? (float)atof(e) : 8.f; }();      // A/B: 0 = move the reference whenever a max grows
  a.lazy = lazy >= 0.f && lazy <= 64.f ? lazy : 8.f;
  if (a.Sk < 0 || a.qpos0 < 0 || (a.Sk > 0 && (a.Sk < a.S + a.qpos0 || !a.block_table)) || (a.Sk == 0 && a.qpos0 != 0)) return -1;   // a context longer than the queries lives in pages of a block table
  if (a.Vrows && (a.block_table || a.Sk || a.v_ld < a.KV * a.Dout || (a.v_ld & 7) || ((uintptr_t)a.Vrows & 15) || (size_t)64 * a.v_ld * 2 >= 0xffffffffull)) return -1;
  if ((a.q_rs != nullptr) != (a.q_nw != nullptr) || (a.q_rs && (!a.Qrows || a.Krows || !a.Vrows || a.D != 96 || a.q_ld < a.H * a.Dout || (a.q_ld & 7) || (((uintptr_t)a.Qrows | (uintptr_t)a.q_nw) & 15)))) return -1;
  if ((!a.q_rs && (a.Krows != nullptr) != (a.Qrows != nullptr)) || (a.Krows && (!a.Vrows || a.D != 64 || a.Dout != a.D || a.k_ld < a.KV * a.D || a.q_ld < a.H * a.D || ((a.k_ld | a.q_ld) & 7) ||
                                                              (((uintptr_t)a.Krows | (uintptr_t)a.Qrows) & 15) || (size_t)64 * a.k_ld * 2 >= 0xffffffffull))) return -1;
  if (a.B <= 0 || a.S <= 0 || a.S > 256 * 64 || a.Sk > 256 * 64 || a.H % a.KV != 0 || a.Dout > a.D || (a.Dout & 7) || ((uintptr_t)a.O & 15)) return -1;   // 16-byte O stores
  const int ring = a.ring == 3 && !a.Vrows ? 3 : 2;
  if (a.vl_n) {                                                  // ragged causal prefill: one grid for all sequences of the group
    if (a.vl_n < 1 || a.vl_n > GVL_MAX_PREFILL_BATCH || a.B != 1 || !a.causal || a.Sk || a.qpos0 || a.Vrows || a.Qrows || a.Krows || a.q_rs || a.block_table) return -1;
    for (int u = 0; u < a.vl_n; ++u) if (!a.vl_tables[u] || a.vl_rows[u + 1] <= a.vl_rows[u] || a.vl_rows[u + 1] - a.vl_rows[u] > a.S) return -1;
    switch (a.D) {
      case 64: return launch_attn<64, 4, 2, 0, 0, 1>(a, st);
      case 96: return launch_attn<96, 4, 2, 0, 0, 1>(a, st);
      case 128: return launch_attn<128, 4, 2, 0, 0, 1>(a, st);
      default: return -1;
    }
  }
  switch (a.D) {
    // ring depth 2: 48 KB (D=96) -> 3 blocks / CU at 151 VGPRs (measured 427 us vs 461 us for the 73 KB depth-3 ring, which
    // caps residency at 2 blocks / CU; DMA latency is not the limiter -- PMC shows the kernel is VALU-issue-bound)
    case 64: return a.Krows ? launch_attn<64, 4, 2, 0, 2>(a, st) : a.Vrows ? launch_attn<64, 4, 2, 0, 1>(a, st) : launch_attn<64, 4, 2>(a, st);
    case 96: {
      static const bool no_ones = gvl_lab_env("GVL_ATTN_NO_ONES") != nullptr;                       // A/B
      // (192-query blocks of 6 waves -- 3 % instead of 5.9 % tail waste at S = 2049, K/V tiles shared by more waves -- measured 24.6 ms
      //  of attention per clip against 18.0: two 98 KB blocks per CU hide less latency than three 49 KB ones.  Round 2, dropped.)
      const int lr = a.Dout - 64;
      const bool ones = a.ones_row && !no_ones && a.Dout < 96 && lr >= 0 && (lr & 7) < 4 && !a.causal;
      if (a.Vrows && a.q_rs) {
        if (a.Dout != 88 || !a.k_ones) return -1;
        if (ones && a.pipe && a.H == a.KV) return launch_attn_iv2_pipe(a, st);      // hand-placed pipelined loop (round 4); bit-identical to the kernel below
        return ones ? launch_attn<96, 4, 2, 1, 3>(a, st) : launch_attn<96, 4, 2, 0, 3>(a, st);
      }
      if (a.Vrows) return ones ? launch_attn<96, 4, 2, 1, 1>(a, st) : launch_attn<96, 4, 2, 0, 1>(a, st);
      if (!ones && ring == 3) return launch_attn<96, 4, 3>(a, st);
      return ones ? launch_attn<96, 4, 2, 1>(a, st) : launch_attn<96, 4, 2>(a, st);
    }
    case 128: return a.Vrows ? -1 : (ring == 3 ? launch_attn<128, 4, 3>(a, st) : launch_attn<128, 4, 2>(a, st));     // row-major V is the vision towers' mode (head dims 64 and 88)
    default: return -1;
  }
}

// =====================================================================================================
// decode attention: one new query per head against the paged cache, split over the context
// =====================================================================================================
// One block per (query head, context split, sequence).  Grouped-query attention (Llama-3: 32 query heads on 8 KV heads): the G = H / KV
// heads of a group read the SAME pages.  Round 1 launched grid (H, nsplit, batch): consecutive query heads are consecutive workgroup
// ids, which the dispatcher deals round-robin over the 8 XCDs -- the 4 heads of a group landed on 4 different L2s and every page crossed
// the fabric 4 times (Llama-3-8B, 16 sequences at 3.5 k context: 0.33 of the HBM roofline, profiles/r02_other_configs.txt).
// PH = 1: XCD-aware grid (8, ceil(units / 8) * G): unit (KV head, split, sequence) = (y / G) * 8 + x, head = KV head * G + y % G -- the G
// blocks of a unit share x, i.e. ONE XCD's L2, and are dispatched 8 ids apart (their misses merge); the pages are then read with plain
// (L2-allocating) loads instead of non-temporal ones.  (A variant with ONE block per KV head looping over its G query heads keeps the
// page in registers but needs 256+ VGPRs at D = 128 -- hipcc spilled 401 of them; dropped.)  G stays 1 in both modes.
template <int D, int G, int PH = 0>
__global__ __launch_bounds__(256, 2) void decode_attn_kernel(const DecodeAttnArgs a) {
  constexpr int CPR = D / 8;       // 16-byte chunks per key row
  constexpr int NIT = D / 8;       // 64*CPR chunks per page / 64 lanes
  constexpr int QP = (64 % CPR == 0) ? 1 : 3;   // distinct q chunks a lane meets: (it*64+lane) % CPR has period 3 for CPR = 12
  static_assert(D == 64 || D == 96 || D == 128, "head dim");
  __shared__ __attribute__((aligned(16))) float part_s[4][64 * (CPR + 1)];   // +1: conflict-free per-key reads
  __shared__ __attribute__((aligned(16))) float p_s[4][64];
  __shared__ float red_s[G][4][D + 2];
  __shared__ int last_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int split = blockIdx.y, bz = blockIdx.z;                            // bz: sequence of the decode batch
  int hkv = blockIdx.x, head0 = hkv * G;
  if constexpr (PH == 2) { head0 = blockIdx.x; hkv = head0 / (a.H / a.KV); }
  if constexpr (PH == 1) {
    const int Gq = a.H / a.KV, units = a.KV * a.gsplit * a.batch;
    const int unit = ((int)blockIdx.y / Gq) * 8 + (int)blockIdx.x;
    if (unit >= units) return;
    hkv = unit % a.KV; split = (unit / a.KV) % a.gsplit; bz = unit / (a.KV * a.gsplit);
    head0 = hkv * Gq + (int)blockIdx.y % Gq;
  }
  const int* __restrict__ block_table = a.tables[bz];
  const bf16_t* qb = a.q + (size_t)bz * a.q_stride;
  float* part_b = a.part + (size_t)bz * a.H * a.nsplit * (D + 2);
  int* counters_b = a.counters + bz * a.H;
  const int pos = *a.pos_ptrs[bz] + 1;              // tokens in the cache, new one included
  const int npages = (pos + 63) >> 6;
  // splits actually used by THIS sequence: one per 4 pages (a block's 4 waves take a page each), at most a.nsplit.  A function of the
  // sequence's own length only, so its arithmetic does not depend on who else is in the batch.  The surplus blocks of the fixed grid
  // leave at once; a short context (<= 256 tokens) is ONE block per head, which writes the output itself -- no partials, no ticket,
  // no merge round trip (measured before: 12 us per layer at 64 tokens, 44 us for 16 sequences -- almost all of it that chain).
  int ns = (npages + 3) >> 2;
  ns = ns > a.nsplit ? a.nsplit : ns;
  // A block takes a.cpb CONSECUTIVE splits, one after the other, and publishes one partial per split -- the partials (and therefore the
  // merged result) are the same whatever cpb is, so the host is free to pick it per launch: 1 when the launch is small (one sequence:
  // as many blocks as possible), up to 4 for long contexts x many sequences, where a block's publish -> ticket (-> merge) tail costs
  // more than its page loads and is then paid once per 16 pages instead of once per 4.
  const int cpb = a.cpb;
  const int c_begin = split * cpb;                   // `split` = block index along the context here
  if (c_begin >= ns) return;
  const int c_end = c_begin + cpb < ns ? c_begin + cpb : ns;
  const int nblk = (ns + cpb - 1) / cpb;             // blocks working on this (head, sequence): the ticket's target
  const int pps = (npages + ns - 1) / ns;

  // the lane's q chunks (chunk index (it*64+lane) % CPR): 16-byte L2 hits, no LDS staging / block barrier
  u32x4_t qv[G][QP];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int it = 0; it < QP; ++it) qv[g][it] = *(const u32x4_t*)(qb + (head0 + g) * D + ((it * 64 + lane) % CPR) * 8);

  const float sc = a.scale * 1.4426950408889634f;
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
  const int p_begin = chunk * pps;
  int p_end = p_begin + pps; if (p_end > npages) p_end = npages;
  float m_run[G], l_run[G];
  float oacc[G][NIT];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m_run[g] = -1e30f; l_run[g] = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) oacc[g][i] = 0.f;
  }

  for (int pg = p_begin + wave; pg < p_end; pg += 4) {
    const size_t pb = ((size_t)block_table[pg] * a.KV + hkv) * (size_t)(64 * D);
    const bf16_t* kp = a.Kt + pb;
    const bf16_t* vp = a.Vt + pb;
    // the whole page (K and V^T) is requested up front: 2*NIT fully coalesced 16-byte loads in flight per lane
    u32x4_t kv[NIT], vv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) kv[it] = PH == 1 ? *(const u32x4_t*)(kp + (it * 64 + lane) * 8) : __builtin_nontemporal_load((const u32x4_t*)(kp + (it * 64 + lane) * 8));
#pragma unroll
    for (int it = 0; it < NIT; ++it) vv[it] = PH == 1 ? *(const u32x4_t*)(vp + (it * 64 + lane) * 8) : __builtin_nontemporal_load((const u32x4_t*)(vp + (it * 64 + lane) * 8));
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += lo_bf(kv[it][e]) * lo_bf(qv[g][it % QP][e]) + hi_bf(kv[it][e]) * hi_bf(qv[g][it % QP][e]);
        part_s[wave][c + c / CPR] = acc;              // key*(CPR+1) + chunk
      }
      __builtin_amdgcn_wave_barrier();
      float sv = 0.f;
#pragma unroll
      for (int i = 0; i < CPR; ++i) sv += part_s[wave][lane * (CPR + 1) + i];
      sv *= sc;
      if (pg * 64 + lane >= pos) sv = -1e30f;
      const float mx = wave_max(sv);
      const float m_new = fmaxf(m_run[g], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
      m_run[g] = m_new;
      const float p = __builtin_amdgcn_exp2f(sv - m_new);
      l_run[g] = l_run[g] * alpha + wave_sum(p);
      p_s[wave][lane] = rbf(p);                       // the reference multiplies bf16 probabilities into V
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = it * 64 + lane;                 // chunk of V^T: d = c/8, keys 8*(c%8)..+7
        const float* pp = p_s[wave] + (c & 7) * 8;
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += lo_bf(vv[it][e]) * pp[2 * e] + hi_bf(vv[it][e]) * pp[2 * e + 1];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        oacc[g][it] = oacc[g][it] * alpha + acc;      // d = it*8 + lane/8 (same value in the 8 lanes)
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // combine the 4 waves of this block
  if (chunk > c_begin) __syncthreads();              // the previous split's red_s has been read by everybody
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if ((lane & 7) == 0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) red_s[g][wave][it * 8 + (lane >> 3)] = oacc[g][it];
    }
    if (lane == 0) { red_s[g][wave][D] = m_run[g]; red_s[g][wave][D + 1] = l_run[g]; }
  }
  __syncthreads();
  if (ns == 1) {                                      // the only split of its head: finish here (same arithmetic as a merge of one partial)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int head = head0 + g;
      const float mm = fmaxf(fmaxf(red_s[g][0][D], red_s[g][1][D]), fmaxf(red_s[g][2][D], red_s[g][3][D]));
      float l = 0.f, acc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float e = __builtin_amdgcn_exp2f(red_s[g][w][D] - mm);
        l += red_s[g][w][D + 1] * e;
        if (tid < D) acc += red_s[g][w][tid] * e;
      }
      if (tid < a.Dout) a.out[a.out_tiled ? gvl_xt_index(bz, head * a.Dout + tid) : (size_t)bz * a.out_stride + head * a.Dout + tid] = f2bf(acc / l);
    }
    return;
  }
  // ---- publish this block's partials with write-through (sc1) stores, take a ticket; the last block of the KV group
  //      merges the partials reading them with sc1 loads (bypass the stale L1): no release/acquire fences needed
  //      (guide G16 recipe R1; placement independent) ---------------------------------------------------------
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float* outp = part_b + ((size_t)(head0 + g) * a.nsplit + chunk) * (D + 2);
    const float mm = fmaxf(fmaxf(red_s[g][0][D], red_s[g][1][D]), fmaxf(red_s[g][2][D], red_s[g][3][D]));
    if (tid < D) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += red_s[g][w][tid] * __builtin_amdgcn_exp2f(red_s[g][w][D] - mm);
      __hip_atomic_store(outp + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
      float l = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) l += red_s[g][w][D + 1] * __builtin_amdgcn_exp2f(red_s[g][w][D] - mm);
      __hip_atomic_store(outp + D, mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(outp + D + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  }                                                  // next split of this block
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(counters_b + head0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = (t == nblk - 1);
  }
  __syncthreads();
  if (!last_s) return;
  // one round trip per head: every thread fetches its share of the nsplit*(D+2) partial words (sc1 loads, all independent)
  // into LDS (the score scratch is free by now), then the merge runs out of LDS
  float* mg = &part_s[0][0];
  const int nword = ns * (D + 2);
  for (int g = 0; g < G; ++g) {
    const int head = head0 + g;
    const float* pp = part_b + (size_t)head * a.nsplit * (D + 2);
    for (int i = tid; i < nword; i += 256) mg[i] = __hip_atomic_load(pp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    float gm = -1e30f;
    for (int s2 = 0; s2 < ns; ++s2) gm = fmaxf(gm, mg[s2 * (D + 2) + D]);
    float l = 0.f, acc = 0.f;
    for (int s2 = 0; s2 < ns; ++s2) {
      const float w = __builtin_amdgcn_exp2f(mg[s2 * (D + 2) + D] - gm);
      l += mg[s2 * (D + 2) + D + 1] * w;
      if (tid < D) acc += mg[s2 * (D + 2) + tid] * w;
    }
    if (tid < a.Dout) a.out[a.out_tiled ? gvl_xt_index(bz, head * a.Dout + tid) : (size_t)bz * a.out_stride + head * a.Dout + tid] = f2bf(acc / l);
    __syncthreads();                                  // mg is refilled for the next head of the group
  }
  if (tid == 0) __hip_atomic_store(counters_b + head0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
}

// =====================================================================================================
// decode attention for GROUPED-query models on the matrix pipe (Llama-3: 32 query heads on 8 KV heads, G = 4)
// =====================================================================================================
// The per-head kernel above runs the G query heads of a KV head as G blocks: every page is fetched G times (from L2) and the
// dot products are G x VALU work; with 231 VGPRs at D = 128 only two blocks fit a CU, so 16 sequences at 3.5 k context (7168 blocks)
// ran 14 rounds of a latency-bound block: 109 us per layer for 231 MB of KV (2.1 TB/s).  Here ONE block per (KV head, split, sequence)
// serves the whole group, and the page layouts are already MFMA operands:
//   S = Q K^T : A = Q  [16 heads (rows >= G zero) x 32 d]  lane (l & 15) = head, (l >> 4) = 8-wide d chunk   (16 B of Q)
//               B = K^T [32 d x 16 keys]                  lane (l & 15) = key,  (l >> 4) = d chunk            (16 B of the K page row)
//   O += P V  : A = P  [16 heads x 32 keys]  (bf16, through LDS: the C layout has keys on lanes, A wants 8 consecutive keys per lane)
//               B = V  [32 keys x 16 d]                   lane (l & 15) = d,    (l >> 4) = 8-wide key chunk   (16 B of the V^T page row)
// C layout of v_mfma_f32_16x16x32_bf16: lane holds column (l & 15) and rows (l >> 4) * 4 + r: a lane owns heads (l >> 4) * 4 + r of its
// key / d column, so the per-head running max / sum / rescale are per-register scalars replicated over the 16 lanes of a row group.
// Wave = page (as above); partials, ticket and merge are the per-head ones (same record format, one record per head of the group).
// all-reduce over the 16 lanes of a DPP row in four VALU instructions (no LDS round trip): xor 1, xor 2 (quad_perm), mirror within 8,
// mirror within 16.  Partners always hold a + b and b + a, so every lane ends with bitwise the same value (sums included).
template <int CTRL>
__device__ __forceinline__ float dpp_row_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_row_f<0xB1>(v)); v = fmaxf(v, dpp_row_f<0x4E>(v)); v = fmaxf(v, dpp_row_f<0x141>(v)); v = fmaxf(v, dpp_row_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_row_f<0xB1>(v); v += dpp_row_f<0x4E>(v); v += dpp_row_f<0x141>(v); v += dpp_row_f<0x140>(v);
  return v;
}
// STG = 1 (default): the page travels global -> registers with the natural, fully coalesced mapping (lane l = bytes [16 l, 16 l + 16) of
// each KiB) and reaches the operand layout through a per-wave 16 KB LDS tile (K first, then V^T in the same tile; 16-byte chunk c of row
// r is stored at chunk c ^ (r & 15) for K rows of 256 B, c ^ ((r >> 1) & 7) for V^T rows of 128 B: writes and operand reads are both
// bank-conflict free).  STG = 0: operands straight from global memory -- 16 consecutive lanes then read 16 different rows (four cache
// lines per quad of lanes), measured 22.6 us per layer for ONE sequence at 3.5 k context against 16.1 us for the per-head VALU kernel.
template <int D, int GM, int STG>
__global__ __launch_bounds__(256, 2) void decode_attn_gqa_kernel(const DecodeAttnArgs a) {
  constexpr int NS = D / 32;       // d steps of the score MFMAs
  constexpr int NDB = D / 16;      // 16-wide d blocks of the output
  static_assert(D % 32 == 0 && (GM == 4 || GM == 16), "geometry");
  __shared__ __attribute__((aligned(16))) bf16_t p_s[4][16][64 + 8];        // per wave: P [head][key] (+8: the A-operand reads of 16 heads hit distinct banks)
  constexpr int MGH = GM == 4 ? 4 : 1;              // heads merged per round trip (LDS: 4 x 16 records of D + 2 words = 33 KB at D = 128)
  constexpr int RED_BYTES = GM * 4 * (D + 2) * 4, MG_BYTES = MGH * 16 * (D + 2) * 4, STG_BYTES = STG ? 4 * 64 * D * 2 : 0;
  constexpr int POOL = (RED_BYTES + MG_BYTES) > STG_BYTES ? (RED_BYTES + MG_BYTES) : STG_BYTES;
  // one pool: the staging tiles (page loop) are dead when the wave partials (red_s) and the merge records (mg_s) are written
  __shared__ __attribute__((aligned(16))) char pool_s[POOL];
  float (*red_s)[4][D + 2] = (float (*)[4][D + 2])pool_s;
  float* mg_s = (float*)(pool_s + RED_BYTES);
  __shared__ int last_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, cg = lane >> 4;
  // a block serves hpb CONSECUTIVE query heads of one KV head (hpb divides H / KV).  The MFMA rows are independent, so a head's result
  // does not depend on hpb: the host picks it per launch -- the whole group when there are enough (KV head, split, sequence) units to
  // fill the chip, fewer heads per block (more blocks, the page re-read from L2) for a single sequence.
  const int G = a.hpb, nhb = a.H / G, units = nhb * a.gsplit * a.batch;
  const int unit = (int)blockIdx.x;
  if (unit >= units) return;
  const int hb = unit % nhb, split = (unit / nhb) % a.gsplit, bz = unit / (nhb * a.gsplit);
  const int head0 = hb * G, hkv = head0 / (a.H / a.KV);
  const int* __restrict__ block_table = a.tables[bz];
  const bf16_t* qb = a.q + (size_t)bz * a.q_stride;
  float* part_b = a.part + (size_t)bz * a.H * a.nsplit * (D + 2);
  int* counters_b = a.counters + bz * a.H;
  const int pos = *a.pos_ptrs[bz] + 1;
  const int npages = (pos + 63) >> 6;
  int ns = (npages + 3) >> 2;                       // same split rule as the per-head kernel: a function of this sequence's length only
  ns = ns > a.nsplit ? a.nsplit : ns;
  const int cpb = a.cpb;
  const int c_begin = split * cpb;
  if (c_begin >= ns) return;
  const int c_end = c_begin + cpb < ns ? c_begin + cpb : ns;
  const int nblk = (ns + cpb - 1) / cpb;
  const int pps = (npages + ns - 1) / ns;

  bf16x8_t qa[NS];                                  // A operand of the score MFMAs: head `col` (zero rows beyond the group)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (col < G) v = *(const u32x4_t*)(qb + (size_t)(head0 + col) * D + s * 32 + cg * 8);
    qa[s] = __builtin_bit_cast(bf16x8_t, v);
  }
  const float sc = a.scale * 1.4426950408889634f;

  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int p_begin = chunk * pps;
    int p_end = p_begin + pps; if (p_end > npages) p_end = npages;
    float m_run[4], l_run[4];
    f32x4_t oacc[NDB];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -1e30f; l_run[r] = 0.f; }
#pragma unroll
    for (int db = 0; db < NDB; ++db) oacc[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int pg = p_begin + wave; pg < p_end; pg += 4) {
      const size_t pb = ((size_t)block_table[pg] * a.KV + hkv) * (size_t)(64 * D);
      const bf16_t* kp = a.Kt + pb;
      const bf16_t* vp = a.Vt + pb;
      u32x4_t kr[4][NS], vr[NDB][2];                // the whole page, requested up front (read ONCE for the G heads)
      if constexpr (STG) {
#pragma unroll
        for (int q = 0; q < 4 * NS; ++q) kr[q / NS][q % NS] = __builtin_nontemporal_load((const u32x4_t*)(kp + q * 512 + lane * 8));
#pragma unroll
        for (int q = 0; q < 2 * NDB; ++q) vr[q / 2][q % 2] = __builtin_nontemporal_load((const u32x4_t*)(vp + q * 512 + lane * 8));
        // K: KiB q holds rows (keys) q * RPK .. with RPK = 1024 / (2 D) rows of CK = D / 8 chunks
        constexpr int CK = D / 8, RPK = 64 / CK;
        char* tile = pool_s + wave * (64 * D * 2);
#pragma unroll
        for (int q = 0; q < 4 * NS; ++q) {
          const int row = q * RPK + lane / CK, c = lane % CK;
          *(u32x4_t*)(tile + row * (D * 2) + ((c ^ (row & (CK - 1) & 15)) << 4)) = kr[q / NS][q % NS];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int s2 = 0; s2 < NS; ++s2) {
            const int row = kb * 16 + col, c = s2 * 4 + cg;
            kr[kb][s2] = *(const u32x4_t*)(tile + row * (D * 2) + ((c ^ (row & (CK - 1) & 15)) << 4));
          }
      } else {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int s = 0; s < NS; ++s) kr[kb][s] = __builtin_nontemporal_load((const u32x4_t*)(kp + (size_t)(kb * 16 + col) * D + s * 32 + cg * 8));
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int t = 0; t < 2; ++t) vr[db][t] = __builtin_nontemporal_load((const u32x4_t*)(vp + (size_t)(db * 16 + col) * 64 + t * 32 + cg * 8));
      }
      // scores: sacc[kb][r] = q(head cg*4+r) . k(key kb*16+col)
      f32x4_t sacc[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        sacc[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[s], __builtin_bit_cast(bf16x8_t, kr[kb][s]), sacc[kb], 0, 0, 0);
      }
      float alpha[4], mx[4], ls[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        mx[r] = -1e30f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          float v = sacc[kb][r] * sc;
          if (pg * 64 + kb * 16 + col >= pos) v = -1e30f;
          sacc[kb][r] = v;
          mx[r] = fmaxf(mx[r], v);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) mx[r] = row16_max(mx[r]);                          // the 16 lanes of a row group hold the 64 keys
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float m_new = fmaxf(m_run[r], mx[r]);
        alpha[r] = __builtin_amdgcn_exp2f(m_run[r] - m_new);
        m_run[r] = m_new;
        ls[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const float pv = __builtin_amdgcn_exp2f(sacc[kb][r] - m_new);
          ls[r] += pv;
          p_s[wave][cg * 4 + r][kb * 16 + col] = f2bf(pv);                           // the reference multiplies bf16 probabilities into V
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) l_run[r] = l_run[r] * alpha[r] + row16_sum(ls[r]);
      __builtin_amdgcn_wave_barrier();
      if constexpr (STG) {                          // V^T rows: 64 keys = 128 B = 8 chunks; KiB q holds d rows 8 q .. 8 q + 7
        char* tile = pool_s + wave * (64 * D * 2);
#pragma unroll
        for (int q = 0; q < 2 * NDB; ++q) {
          const int row = q * 8 + (lane >> 3), c = lane & 7;
          *(u32x4_t*)(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = vr[q / 2][q % 2];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int row = db * 16 + col, c = t * 4 + cg;
            vr[db][t] = *(const u32x4_t*)(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
          }
      }
      bf16x8_t pa[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) pa[t] = *(const bf16x8_t*)(&p_s[wave][col][t * 32 + cg * 8]);
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[db][r] *= alpha[r];
        oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[0], __builtin_bit_cast(bf16x8_t, vr[db][0]), oacc[db], 0, 0, 0);
        oacc[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[1], __builtin_bit_cast(bf16x8_t, vr[db][1]), oacc[db], 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // combine the 4 waves: red_s[head][wave][d], m, l  (red_s aliases the staging tiles: every wave must be out of its page loop)
    if (STG || chunk > c_begin) __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = cg * 4 + r;
      if (h < G) {
#pragma unroll
        for (int db = 0; db < NDB; ++db) red_s[h][wave][db * 16 + col] = oacc[db][r];
        if (col == 0) { red_s[h][wave][D] = m_run[r]; red_s[h][wave][D + 1] = l_run[r]; }
      }
    }
    __syncthreads();
    if (ns == 1) {                                  // the only split of this sequence: finish here, all heads of the group at once
      for (int idx = tid; idx < G * D; idx += 256) {
        const int g = idx / D, d = idx - g * D, head = head0 + g;
        const float mm = fmaxf(fmaxf(red_s[g][0][D], red_s[g][1][D]), fmaxf(red_s[g][2][D], red_s[g][3][D]));
        float l = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float e = __builtin_amdgcn_exp2f(red_s[g][w][D] - mm);
          l += red_s[g][w][D + 1] * e;
          acc += red_s[g][w][d] * e;
        }
        if (d < a.Dout) a.out[a.out_tiled ? gvl_xt_index(bz, head * a.Dout + d) : (size_t)bz * a.out_stride + head * a.Dout + d] = f2bf(acc / l);
      }
      return;
    }
    for (int g = 0; g < G; ++g) {                    // one partial record per head (same format as the per-head kernel)
      float* outp = part_b + ((size_t)(head0 + g) * a.nsplit + chunk) * (D + 2);
      const float mm = fmaxf(fmaxf(red_s[g][0][D], red_s[g][1][D]), fmaxf(red_s[g][2][D], red_s[g][3][D]));
      if (tid < D) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) acc += red_s[g][w][tid] * __builtin_amdgcn_exp2f(red_s[g][w][D] - mm);
        __hip_atomic_store(outp + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid == 0) {
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) l += red_s[g][w][D + 1] * __builtin_amdgcn_exp2f(red_s[g][w][D] - mm);
        __hip_atomic_store(outp + D, mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(outp + D + 1, l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (STG && chunk + 1 < c_end) __syncthreads();   // the next split's staging tiles overwrite red_s
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(counters_b + head0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = (t == nblk - 1);
  }
  __syncthreads();
  if (!last_s) return;
  // merge: the partial records of up to MGH heads are fetched in ONE round trip (sc1 loads, all independent) and merged by (head, d)
  // work items -- a group of 4 heads costs one round trip instead of four
  const int nword = ns * (D + 2);
  for (int g0 = 0; g0 < G; g0 += MGH) {
    const int gn = G - g0 < MGH ? G - g0 : MGH;
    for (int i = tid; i < gn * nword; i += 256) {
      const int g = i / nword, wd = i - g * nword;
      mg_s[g * 16 * (D + 2) + wd] = __hip_atomic_load(part_b + (size_t)(head0 + g0 + g) * a.nsplit * (D + 2) + wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int idx = tid; idx < gn * D; idx += 256) {
      const int g = idx / D, d = idx - g * D, head = head0 + g0 + g;
      const float* mg = mg_s + g * 16 * (D + 2);
      float gm = -1e30f;
      for (int s2 = 0; s2 < ns; ++s2) gm = fmaxf(gm, mg[s2 * (D + 2) + D]);
      float l = 0.f, acc = 0.f;
      for (int s2 = 0; s2 < ns; ++s2) {
        const float w = __builtin_amdgcn_exp2f(mg[s2 * (D + 2) + D] - gm);
        l += mg[s2 * (D + 2) + D + 1] * w;
        acc += mg[s2 * (D + 2) + d] * w;
      }
      if (d < a.Dout) a.out[a.out_tiled ? gvl_xt_index(bz, head * a.Dout + d) : (size_t)bz * a.out_stride + head * a.Dout + d] = f2bf(acc / l);
    }
    __syncthreads();
  }
  if (tid == 0) __hip_atomic_store(counters_b + head0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int D>
static int launch_decode_g(const DecodeAttnArgs& a, hipStream_t st) {
  static const bool no_gqa = gvl_lab_env("GVL_DECODE_ATTN_NOGQA") != nullptr;      // A/B: the round-1 grid for GQA models
  const int G = a.H / a.KV;
  static const bool gqa_valu = gvl_lab_env("GVL_DECODE_ATTN_GQA_VALU") != nullptr;  // A/B: the per-head VALU kernel on the XCD-aware grid
  if (G > 1 && G <= 16 && !no_gqa && !gqa_valu && D % 32 == 0) {
    DecodeAttnArgs b = a;
    if (b.hpb < 1 || b.hpb > G || G % b.hpb) b.hpb = G;
    const int units = (b.H / b.hpb) * b.gsplit * b.batch;
    static const bool direct_env = gvl_lab_env("GVL_DECODE_ATTN_GQA_DIRECT") != nullptr;   // A/B: operands straight from global memory
    const bool direct = direct_env || (D & (D - 1)) != 0;                            // the staged tiles need power-of-two rows (64 / 128)
    if (direct) {
      if (b.hpb <= 4) hipLaunchKernelGGL((decode_attn_gqa_kernel<D, 4, 0>), dim3(units), dim3(256), 0, st, b);
      else hipLaunchKernelGGL((decode_attn_gqa_kernel<D, 16, 0>), dim3(units), dim3(256), 0, st, b);
    } else {
      if (b.hpb <= 4) hipLaunchKernelGGL((decode_attn_gqa_kernel<D, 4, 1>), dim3(units), dim3(256), 0, st, b);
      else hipLaunchKernelGGL((decode_attn_gqa_kernel<D, 16, 1>), dim3(units), dim3(256), 0, st, b);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  if (G == 1 || no_gqa) {
    if (G == 1) hipLaunchKernelGGL((decode_attn_kernel<D, 1, 0>), dim3(a.H, a.gsplit, a.batch), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((decode_attn_kernel<D, 1, 2>), dim3(a.H, a.gsplit, a.batch), dim3(256), 0, st, a);
  } else {
    const int units = a.KV * a.gsplit * a.batch;
    hipLaunchKernelGGL((decode_attn_kernel<D, 1, 1>), dim3(8, (units + 7) / 8 * G), dim3(256), 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int gvl_launch_decode_attention(const DecodeAttnArgs& a_in, hipStream_t st) {
  DecodeAttnArgs a = a_in;
  if (a.batch <= 0) a.batch = 1;
  if (a.batch > GVL_MAX_DECODE_BATCH) return -1;
  if (a.H % a.KV) return -1;
  if (a.cpb < 1) a.cpb = 1;
  if (a.nsplit < 1 || a.nsplit > 16) return -1;      // the merge buffers hold 16 partial records
  if (a.gsplit <= 0 || a.gsplit > a.nsplit) a.gsplit = (a.nsplit + a.cpb - 1) / a.cpb;
  switch (a.D) {
    case 64: return launch_decode_g<64>(a, st);
    case 96: return launch_decode_g<96>(a, st);
    case 128: return launch_decode_g<128>(a, st);
    default: return -1;
  }
}
