// gvl_internal.h -- shared declarations for the libgvl.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits

// Kernel-selection switches of the A/B history exist only in LAB builds (GVL_BUILD_DEFS=-DGVL_LAB, tools/): the shipped library
// reads no such environment variable -- neither per launch nor per token.  Deployment knobs (GVL_RCCL_LIB, GVL_KV_FRACTION) are documented in
// include/gvl.h; result-neutral launch parameters that tests vary go through gvl_debug_set().
#include <cstdlib>
#ifdef GVL_LAB
inline const char* gvl_lab_env(const char* name) { return getenv(name); }
#else
inline const char* gvl_lab_env(const char*) { return nullptr; }
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define GVL_KV_PAGE 64          // tokens per KV page == key tile of the attention kernels
constexpr int GVL_MAX_DECODE_BATCH = 16;  // sequences decoded together: the weight stream is read ONCE for all of them;
                                          // = the 16 columns of the MFMA B operand of the skinny decode GEMM (gvl_decode.hip)
constexpr int GVL_MAX_VALU_BATCH = 4;     // the round-1 VALU GEMV (fallback for K % 256 != 0 geometries) holds B vectors in LDS: 1, 2 or 4
constexpr int GVL_GEMM_ROT_LEN = 64;     // GemmArgs.rot entries (a power of two; longer walks repeat the table)
constexpr int GVL_MAX_PREFILL_BATCH = 8;  // most sequences whose rows share one pass of the prefill GEMMs (gvl_debug_set prefill_group picks 1 .. 8; default 4)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a host may hold one gvl_ctx per device in ONE process
// (include/gvl.h allows it), so "set once" means once per device ordinal -- a process-global flag would leave the second device at the
// 64 KB default and its first big-LDS launch would fail.  One bit per ordinal (mod 64); racing first launches set it twice, harmlessly.
#include <atomic>
struct GvlDevOnce { std::atomic<unsigned long long> mask{0}; };
inline int gvl_set_max_lds(GvlDevOnce& once, const void* kern, int bytes) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return -3;
  const unsigned long long bit = 1ull << (d & 63);
  if (!(once.mask.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -3;
    once.mask.fetch_or(bit, std::memory_order_release);
  }
  return 0;
}

// ---- device helpers ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// float -> bf16 round-to-nearest-even: the (__bf16) cast lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) { const __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float rbf(float f) { return (float)(__bf16)f; }  // round through bf16
// fast transcendental helpers for epilogues (errors << bf16 resolution)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float lo_bf(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_bf(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// global -> LDS DMA, 16 bytes per lane, issued from inline asm so that hipcc does NOT count it in its vmcnt
// bookkeeping (a builtin glds makes the compiler drain vmcnt(0) before the next LDS read: it cannot prove the
// DMA destination and the ds_read source are different ring slots).  The caller owns the waits:
// `s_waitcnt vmcnt(N)` + a barrier before any wave reads the slot.  LDS destination = lds_dst (wave-uniform
// byte address, via M0) + lane*16; gsrc is per lane.  Recipe: cdna_hip_programming.md §5.7.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// Four DMA pieces from ONE wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offsets: no per-piece 64-bit VALU
// address arithmetic in the issuing phase.  lds_dst0 + q*lds_step is piece q's wave-uniform LDS byte address.
__device__ __forceinline__ void glds16x4(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst0, unsigned lds_step) {
  unsigned keep, d;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
      "s_add_u32 %1, %7, %8\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
      "s_add_u32 %1, %1, %8\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
      "s_add_u32 %1, %1, %8\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %6\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(d)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst0), "s"(lds_step)
      : "memory", "scc");
}
// N pieces (N = 2 or 3) from one SGPR base, LDS destinations lds_dst0 + q*lds_step  (attention K / V^T tiles)
template <int N>
__device__ __forceinline__ void glds16xn(const void* sbase, const unsigned (&v)[N], unsigned lds_dst0, unsigned lds_step) {
  static_assert(N == 2 || N == 3, "2 or 3 pieces");
  unsigned keep, d;
  if constexpr (N == 3) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
        "s_add_u32 %1, %6, %7\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
        "s_add_u32 %1, %1, %7\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(d)
        : "v"(v[0]), "v"(v[1]), "v"(v[N - 1]), "s"(sbase), "s"(lds_dst0), "s"(lds_step)
        : "memory", "scc");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
        "s_add_u32 %1, %5, %6\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(d)
        : "v"(v[0]), "v"(v[1]), "s"(sbase), "s"(lds_dst0), "s"(lds_step)
        : "memory", "scc");
  }
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {   // 32-bit LDS byte address of a __shared__ pointer
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}

// ---- GEMM ---------------------------------------------------------------------------------------
enum { GVL_ACT_NONE = 0, GVL_ACT_QUICK_GELU = 1, GVL_ACT_GELU = 2, GVL_ACT_SILU_MUL = 3 };

struct GemmArgs {
  const bf16_t* A;  int lda;     // [M,K] bf16
  const bf16_t* W;  int ldw;     // [N,K] bf16 (nn.Linear layout); ldw = row pitch in elements (0: K)
  void* C;          int ldc;     // [M,N'] f32 or bf16
  int M, N, K;
  const float* bias;             // [N] or null
  const float* gamma;            // [N] LayerScale (after bf16 rounding of acc+bias) or null
  const void* resid; int ldr;    // residual stream, same dtype as C, or null (may alias C)
  int act;                       // GVL_ACT_*
  int out_f32;                   // 1: C/resid f32, 0: bf16
  int round_pre_resid;           // 1: round (acc+bias) [and the gamma product] to bf16 before adding resid
  // output row remap: row m -> (m / grp_rows) * grp_stride + (m % grp_rows) + row_off   (grp_rows==0: identity)
  int grp_rows, grp_stride, row_off;
  int tile_cfg;                  // 0 auto; see gvl_launch_gemm
  int band;                      // ping-pong kernel: tile ROWS per rasterisation band (0 = the launcher's choice); any value gives the same result
  unsigned char rot[64];         // pipelined 4-wave kernel: workgroup c of an XCD takes offset (c + rot[round & 63]) mod (workgroups of the XCD) of its XCD's round (0: fixed)
  int narrow;                    // set by the pipelined 4-wave launcher: tiles with <= 128 real columns run its narrow statements (gvl_debug_set "gemm_narrow"); result-neutral
  int m_begin;                   // launch covers rows [m_begin, M) -- set internally by the wave-quantisation split
  // Fused RMSNorm (round 5).  CONSUMER side: rowscale[m] (f32, [M]) multiplies row m of the accumulator BEFORE bias / activation -- RMSNorm(x) . W^T =
  // rs[m] * (x . (W diag(gamma))^T): A is the RAW residual stream, W carries the norm weight (gvl_fold_gamma), rs = rsqrt(mean x^2 + eps).
  // PRODUCER side: rowsq[m * rowsq_ld + n / 64] (f32) receives the sum of squares of the bf16-ROUNDED outputs of row m over each aligned block of 64
  // columns -- the statistics the NEXT norm needs, taken while the row piece is in registers (N % 64 == 0; bf16 output; staged epilogue only).
  const float* rowscale;
  float* rowsq; int rowsq_ld;
  const float* act_table;        // set by the launcher: Phi(x) table for the erf-GELU epilogue of the ping-pong kernel (see gvl_gemm.hip)
  unsigned long long* dbg;       // null, or [grid][8 waves][4] s_memtime stamps of the LAST tile (GVL_GEMM_TIMING=1, ping-pong kernel)
};
// Rasterisation band of the 256 x 256 kernels: an XCD's 32 workgroups walk a band of this many tile rows column by column.  4 since round 6: the A panel of a
// band (1 024 rows x K) then fits the XCD's 4 MiB L2 for K <= 1 408 and halves the A re-reads of the wide matrices; same-box A/B of the bench step with the 4-wave
// kernels: 8 -> 4 rows = 701 -> 686 ms per step (-2.1 %), 2 / 3 / 6 / 16 rows: 694 / 688 / 692 / 732 (profiles/r06_ab_gemm_band.json; per shape: iv2.fc2 +3.7 %,
// phi.down +4.9 %, the others within +-1.5 %: profiles/r06_gemm_band_shapes.txt).  (Round 5 had tried 5-row bands for N = 1408 only, on the 8-wave kernel.)
constexpr int GVL_GEMM_BAND = 4;
int gvl_launch_gemm(const GemmArgs& a, hipStream_t st);
// gvl_gemm4.hip: the 4-wave / AGPR-accumulator form of the 256 x 256 kernel for the staged bf16 epilogue code `epi` (loop schedule variant `var`);
// -2 = this epilogue or geometry is not served there (the caller launches the 8-wave kernel instead)
int gvl_launch_gemm_a4(const GemmArgs& a, int epi, int var, hipStream_t st);
// gvl_gemm4p.hip: the same kernel with the epilogue software-pipelined into the next tile's main loop; -2 as above
int gvl_launch_gemm_a4p(const GemmArgs& a, int epi, hipStream_t st);
void gvl_gemm_set_a4(int v);     // A/B: 0 = the 256 x 256 launches stay on the 8-wave ping-pong kernel, 1 (default) = the 4-wave kernel where it is faster, 2 = wherever it serves
void gvl_gemm_set_narrow(int v); // A/B: 0 = the pipelined 4-wave kernel runs a half-empty column tile (N = 1408: every sixth) as a full one (rounds <= 6a), 1 (default) = as a narrow tile
void gvl_gemm_set_band(int v);   // A/B: tile rows per rasterisation band of the ping-pong kernel for every later launch of the process (0 = automatic)
double gvl_gemm_flops(const GemmArgs& a);

// ---- attention (prefill / vision) -----------------------------------------------------------------
struct AttnArgs {
  const bf16_t* Q;      // [B][H][S][D] bf16 (D = padded head dim: 64, 96 or 128)
  const bf16_t* Kt;     // key pages  [page][KV][64][D]
  const bf16_t* Vt;     // value pages [page][KV][D][64]  (transposed inside the page)
  const bf16_t* Vrows;  // null, or V as token rows: V[b][s][kv head][0..Dout) = Vrows[(b*S + s)*v_ld + head*Dout + d] (the fused-qkv GEMM output of a
  int v_ld;             //   vision tower, read in place: no V^T pass); then Vt is unused and block_table must be null
  const bf16_t* Qrows; const bf16_t* Krows;   // both null, or (with Vrows; D == Dout == 64) q and k as token rows too: Qrows[(b*S + s)*q_ld + head*D + d], Krows likewise
  int q_ld, k_ld;       //   -- a tower whose q / k need no per-token transform (CLIP) runs no qkv_post pass; then Q / Kt are unused
  int k_ones;           // the K pages' pad column Dout holds 1.0 (QkvPostArgs.k_ones): required by the q_rs mode, which folds the softmax shift into the S^T MFMAs
  const float* q_rs; const bf16_t* q_nw;      // both null, or (Qrows set, Krows null, Vrows set): q needs the full-width RMSNorm of InternVideo2 --
                        //   q'[d] = q_nw[head*Dout + d] * bf16(q[d] * q_rs[token]) applied to the fragments as they are loaded (K stays in pages)
  bf16_t* O;            // [B][S][H*Dout]
  const int* block_table;  // [B][max_pages] page ids, or null: page(b,t) = b*n_tiles + t
  int max_pages;
  int B, H, KV, S, D, Dout;   // S = number of queries; keys: Sk (0 = S, plain self attention)
  int Sk, qpos0;              // extend-prefill (gvl_prefill_extend): query i sits at position qpos0 + i of a context of Sk = qpos0 + S keys (causal: key <= qpos0 + i)
  float scale;
  int causal;
  int ones_row;         // D > Dout only: V^T pad row Dout holds 1.0 for every key (QkvPostArgs.ones_row), so the P.V MFMAs deliver the
                        // softmax row sum in O^T[Dout] for free and the kernel drops its 32 VALU adds per key tile
  int ring;             // K/V LDS ring depth: 0 = launcher's choice, 2 or 3 (paged-KV path only)
  int pipe;             // 1: the q_rs (InternVideo2) mode takes the software-pipelined kernel attn_iv2_pipe_kernel; 0: attn_fwd_kernel (bit-identical); 2: its safe pass only
  int pipe_rows;        // 256: whole 256-row query blocks on the 8-wave form of that kernel, the rest on the 4-wave form; anything else (default): the 4-wave form for every row
  int q_begin, q_rows;  // set by the launcher: the query rows one attn_iv2_pipe_kernel launch covers
  float lazy;           // set by the launcher: the running max (and O) of a wave's rows is only moved when some row's tile max exceeds it by
                        // more than `lazy` (log2 units; 0 = every time it grows)
  // RAGGED causal prefill in one grid (paged K/V, B == 1, Sk == qpos0 == 0): vl_n > 0 sequences packed back to back -- sequence u owns rows
  // [vl_rows[u], vl_rows[u + 1]) of Q ([H][S_u][D] per sequence, at row offset vl_rows[u]) and O, and the block table vl_tables[u]; S = longest sequence
  int vl_n;
  int vl_rows[GVL_MAX_PREFILL_BATCH + 1];
  const int* vl_tables[GVL_MAX_PREFILL_BATCH];
};
int gvl_launch_attention(const AttnArgs& a, hipStream_t st);
double gvl_attn_flops(const AttnArgs& a);

// ---- decode attention -----------------------------------------------------------------------------
struct DecodeAttnArgs {
  const bf16_t* q;        // [batch][H][D], q_stride elements apart
  const bf16_t* Kt; const bf16_t* Vt;   // page pools (one layer)
  const int* tables[GVL_MAX_DECODE_BATCH];   // per sequence: [max_pages]
  const int* pos_ptrs[GVL_MAX_DECODE_BATCH]; // per sequence, device: index of the new token (the cache holds *pos + 1 tokens, new one included)
  float* part;            // workspace [batch][H][nsplit][D+2]
  int* counters;          // [batch][H] arrival tickets, zero between launches (the merging block re-arms them)
  bf16_t* out;            // [batch][H*Dout], out_stride elements apart; out_tiled: B-operand tile order (gvl_xt_index) for the skinny decode GEMM
  int H, KV, D, Dout, nsplit, batch, q_stride, out_stride, out_tiled;
  float scale;
  int hpb;                // grouped-query kernel: query heads of a KV head served by one block (0 = all H / KV; must divide it); any value gives the same result
  int cpb;                // consecutive splits per block (0 = 1); any value gives the same result (one partial per split either way)
  int gsplit;             // block slots along the context actually launched (0 = ceil(nsplit / cpb)): the host may pass min(nsplit, ceil(longest context in pages / 4)) -- a
                          // sequence uses ceil(its pages / 4) splits whatever the grid offers, so this only trims blocks that would leave at once
};
int gvl_launch_decode_attention(const DecodeAttnArgs& a, hipStream_t st);

// ---- elementwise / norm / glue kernels (gvl_elem.hip) ----------------------------------------------
int gvl_launch_layernorm_f32(const float* x, const float* w, const float* b, bf16_t* y, int rows, int cols, float eps, hipStream_t st);
int gvl_launch_rmsnorm_bf16(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int cols, float eps, hipStream_t st);
int gvl_launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t st);   // device -> device, 16-byte vectors (falls back to hipMemcpyAsync for odd sizes)
int gvl_launch_fold_gamma(const bf16_t* W, const bf16_t* gamma, bf16_t* Wo, long rows, int cols, hipStream_t st);   // W' = bf16(W diag(gamma)): fused RMSNorm, consumer weights
int gvl_launch_rowsq_finish(const float* sq, int ld, int b0, int nblk, float* rs, int rows, int cols, float eps, hipStream_t st);   // partial sums of squares -> rsqrt(mean + eps)
// im2col for a stride==kernel patch conv.  px f32 [n_img][3][T][HW][HW] (T==1 for CLIP) -> A bf16 [n_img*T*g*g][Kp]
int gvl_launch_patchify(const float* px, bf16_t* A, int n_img, int T, int image, int patch, int Kp, hipStream_t st);
// CLIP embeddings + pre-LN: x[n,0]=cls+pos0, x[n,1+p]=bf16r(patch)+pos -> LN -> f32 [n,1+P,C]
int gvl_launch_clip_embed_ln(const bf16_t* patch, const float* cls, const float* pos, const float* lnw, const float* lnb,
                             float* x, int n_img, int P, int C, float eps, hipStream_t st);
// IV2 embeddings: x[b,0]=bf16(cls+pos0), x[b,1+j]=bf16(patch+pos) (all bf16) -> bf16 [B,1+TL,C]
int gvl_launch_iv2_embed(const bf16_t* patch, const bf16_t* cls, const bf16_t* pos, bf16_t* x, int B, int TL, int C, hipStream_t st);
// ---- fused patch embedding (gvl_patch.hip): im2col in the operand loader + patch GEMM + CLS / position rows (+ CLIP's pre-LayerNorm) as ONE kernel ----
struct PatchEmbedArgs {
  const float* px;        // f32 [n_img][3][T][image][image]
  const bf16_t* Wt;       // tile-order conv weight (gvl_retile_patch_weight): [C / 16][3 * patch / 2][64][8]
  int n_img, T, image, patch, C;
  int M;                  // patch rows = n_img * T * (image / patch)^2
  int S;                  // output rows per image (1 + T * L)
  int mode;               // 0: CLIP (x f32, LayerNorm), 1: InternVideo2 (x bf16, conv bias)
  const float* bias;      // mode 1: conv bias [C]
  const float *cls_f32, *pos_f32, *lnw, *lnb; float eps; float* x_f32;      // mode 0
  const bf16_t *cls_bf, *pos_bf; bf16_t* x_bf;                               // mode 1
};
int gvl_retile_patch_weight(const bf16_t* W, bf16_t* Wt, int C, int Kp, int p, hipStream_t st);
int gvl_launch_patch_embed(const PatchEmbedArgs& a, hipStream_t st);      // -1: geometry outside the fused kernel (caller takes the three-pass path)
// split a fused qkv row into attention operands.
//  mode 0 (CLIP): plain.  mode 1 (IV2): RMS-normalise q and k over the full width with weights qn/kn.
//  mode 2 (LLM): RoPE with cos/sin tables at positions pos0+s.
struct QkvPostArgs {
  const bf16_t* qkv; int ld;       // [B*S][(H+2KV)*Dr]
  bf16_t* Q; bf16_t* Kt; bf16_t* Vt;
  const int* block_table; int max_pages;   // null: page(b,t) = b*n_tiles + t
  int B, S, H, KV, Dr, D;          // Dr = real head dim, D = padded
  int mode;
  const bf16_t* qn; const bf16_t* kn; float eps;     // mode 1
  const float* cos; const float* sin; int pos0;      // mode 2: tables [max_seq][Dr/2] (already bf16-rounded values)
  const int* pos_ptr;              // mode 2 decode: device position of the (single) row, overrides pos0 when non-null
  const float* cos_l; const float* sin_l; int rope_switch;   // decode: long-factor tables used when pos+1 > rope_switch (>0)
  int ones_row;                    // V^T pad row Dr (needs D > Dr) is filled with 1.0 instead of 0: see AttnArgs.ones_row
  int k_ones;                      // K pad column Dr (needs D > Dr) is 1.0 instead of 0 (harmless while q's pad is 0; see AttnArgs.k_ones)
  float* q_rs;                     // mode 1, or null: [B*S] -- the q rows are NOT written; their RMS factor rsqrt(mean q^2 + eps) is, and the attention
                                   //   kernel normalises the q fragments it loads from the qkv matrix itself (AttnArgs.q_rs / q_nw)
  // RAGGED prefill group in one launch (mode 2, B == 1, pos0 == 0; AttnArgs.vl_* on the attention side): vl_n > 0 sequences packed back to back -- sequence u owns
  // the rows [vl_rows[u], vl_rows[u + 1]) of qkv and of Q ([H][S_u][D] per sequence, at row offset vl_rows[u]) and the block table vl_tables[u]; S is ignored
  int vl_n;
  int vl_rows[GVL_MAX_PREFILL_BATCH + 1];
  const int* vl_tables[GVL_MAX_PREFILL_BATCH];
};
int gvl_launch_qkv_post(const QkvPostArgs& a, hipStream_t st);
// HD 2x2 merge + sub_GN newline (Phi): f32 [n,576,C] -> bf16 [n,156,4C]
int gvl_launch_hd_merge(const float* f, const float* sub_gn, bf16_t* out, int n, int C, hipStream_t st);
// 3x3 block mean (Llama): f32 [n,576,C] -> bf16 [n,64,C]
int gvl_launch_pool_spatial(const float* f, bf16_t* out, int n, int C, hipStream_t st);
// temporal 4x4 block mean: bf16 [n,T*256,C] -> bf16 [n,T*16,C]
int gvl_launch_pool_temporal(const bf16_t* f, bf16_t* out, int n, int T, int C, hipStream_t st);
int gvl_launch_f32_to_bf16(const float* x, bf16_t* y, int64_t n, hipStream_t st);
// broadcast one row to rows row_off + s*stride, s in [0,n)
int gvl_launch_bcast_row(const bf16_t* row, bf16_t* dst, int n, int stride_rows, int row_off, int cols, hipStream_t st);
// embedding gather: ids (device int32) -> rows of dst
int gvl_launch_gather_rows(const bf16_t* table, const int* ids, bf16_t* dst, int n, int cols, hipStream_t st);
int gvl_launch_strip_cls(const void* x, void* y, int n, int S, int C, int elem_bytes, hipStream_t st);
// decode-side
struct GemvArgs {
  const bf16_t* W; int N, K;      // [N][K]
  const bf16_t* x;                // [batch][K] bf16, rows x_stride elements apart
  const bf16_t* norm_w; float eps;  // non-null: x <- rmsnorm(x)*norm_w (bf16 roundings as the reference)
  const float* bias;              // [N] or null
  const bf16_t* resid;            // [batch][N'] or null: out = bf16(resid + bf16(y)); rows out_stride apart
  int act;                        // GVL_ACT_NONE or GVL_ACT_SILU_MUL (interleaved gate/up rows, N' = N/2)
  bf16_t* out_bf16;               // [batch][N'] or null, rows out_stride apart
  float* out_f32;                 // [batch][N'] or null, rows out_stride apart
  int batch, x_stride, out_stride;   // batch 0/1 = one vector (strides ignored)
  // fused decode epilogue of the qkv projection (rope_on): RoPE on q/k at the sequence's position, q -> Q[b][H][D], k/v appended
  // to that sequence's pages.  Rows are visited in (d, d+Dr/2) partner pairs so one lane owns both halves of a rotation.
  int rope_on; const float *cos_s, *sin_s, *cos_l, *sin_l; int rope_switch;
  const int* pos_ptrs[GVL_MAX_DECODE_BATCH]; const int* tables[GVL_MAX_DECODE_BATCH];   // per sequence of the batch
  bf16_t *Q, *Kt, *Vt; int H, KV, Dr, D; int q_stride;
  int variant;                    // skinny-GEMM path: kernel variant (0 = default; tools/decode_bench.py)
  int w_fp8; const float* wscale; // skinny-GEMM path: 1 = W is the FP8 tile copy, wscale[N] its per-row power-of-two scales; 2 = the MXFP4 tile
                                  // copy, wscale = its E8M0 scale words (gvl_mxfp4_quantise_decode_weight)
  int out_tiled;                  // skinny-GEMM path: the SwiGLU epilogue writes out_bf16 in B-operand tile order (it feeds down_proj)
  // Fused RMSNorm on the decode path (round 5; skinny-GEMM path, bf16 weights): rmsnorm(x) W^T = rs[b] * (x (W diag gamma)^T) per sequence b.
  //   PRODUCER (o_proj / down_proj, the residual epilogue; N % 16 == 0): sq_out[b * (N / 16) + block] = sum of squares of the 16 bf16 outputs of that row
  //   block for sequence b, and out_tiled2 = a second copy of the new residual rows in B-operand tile order (the next projection's x);
  //   CONSUMER (qkv_proj / gate_up_proj / lm_head; W = the tile copy of the norm-FOLDED weight, x = that raw tiled stream): sq_in[b * sq_n + j], j < sq_n,
  //   are summed in a fixed order by the epilogue waves while the first weight tiles are in flight; the accumulators are scaled by rsqrt(sum / K + eps).
  // A sequence's sums and scale depend on its own column only: batched decode == single decode still holds bit for bit.
  const float* sq_in; int sq_n;
  float* sq_out; bf16_t* out_tiled2;
};
// B-operand tile order of the decode activations: element (sequence j < 16, column k) of a [16][cols] matrix lives at
// [k / 32][lane = 16 * ((k / 8) % 4) + j][k % 8] -- one k step of the MFMA is 1 KiB of consecutive addresses
__host__ __device__ __forceinline__ size_t gvl_xt_index(int j, int k) { return ((size_t)(k >> 5) * 64 + (size_t)(((k >> 3) & 3) * 16 + j)) * 8 + (k & 7); }
int gvl_retile_decode_weight(const bf16_t* W, bf16_t* Wt, int N, int K, int Dr, int n_qk_heads, hipStream_t st);
// FP8 variant: scale[n] (per logical row, power of two), Wt8 = e4m3 tile copy, and W (row-major bf16) REPLACED by its de-quantised values
int gvl_fp8_quantise_decode_weight(bf16_t* W, unsigned char* Wt8, float* scale, int N, int K, int Dr, int n_qk_heads, hipStream_t st);
// MXFP4 twin: Wt4 [ceil(N/16)][K/128][64][16] bytes, scale [ceil(N/16)][K/128][16] words of four E8M0 bytes; W is replaced by the de-quantised values
int gvl_mxfp4_quantise_decode_weight(bf16_t* W, unsigned char* Wt4, unsigned* scale, int N, int K, int Dr, int n_qk_heads, hipStream_t st);
int gvl_launch_rows_to_tiled(const bf16_t* x, bf16_t* xt, int batch, int cols, int stride, hipStream_t st);
int gvl_launch_gemv(const GemvArgs& a, hipStream_t st);
// the same projection as ONE MFMA skinny GEMM for 1..16 sequences (gvl_decode.hip); -1 when the geometry needs the VALU kernel
int gvl_launch_dgemm(const GemvArgs& a, hipStream_t st);
// greedy sampling for `batch` logit rows (stride n): token -> *tok_ptrs[b] and out_lists[b][steps[b]]
// small host int list passed to kernels by value (stream ordered, no host / staging buffer lifetime)
struct IntList { int v[256]; int n; };
// dst[r] = table[host_ids[r]], ids travel in the kernel arguments (256 per launch)
int gvl_launch_gather_rows_host_ids(const bf16_t* table, const int* host_ids, int n, bf16_t* dst, int cols, hipStream_t st);
// loss tail of the training forward: nll[r] = logsumexp(logits[r]) - logits[r][targets[r]] (f32 on bf16 logits)
int gvl_launch_ce_rows(const bf16_t* logits, int ld, const int* targets, float* nll, int n, int V, hipStream_t st);
// token -> *tok_ptrs[b] and out_lists[b][*ngen_ptrs[b]]; then (*ngen_ptrs[b])++ and, when non-null, (*pos_ptrs[b])++ (the step's
// bookkeeping lives on the device: a decode step's launches carry no host-side counters)
struct ArgmaxArgs { const float* logits; int n, batch; int* tok_ptrs[GVL_MAX_DECODE_BATCH]; int* out_lists[GVL_MAX_DECODE_BATCH];
                    int* ngen_ptrs[GVL_MAX_DECODE_BATCH]; int* pos_ptrs[GVL_MAX_DECODE_BATCH];
                    // sampling (gvl_launch_sample only): scores / temperature -> top-k -> top-p -> one draw per row; the draw of row b is a
                    // pure function of (seed, stream[b], generation step, logits row), i.e. independent of how sequences are grouped
                    // eos watch (gvl_decode_greedy*): when row b's token == eos_id, its generation count is stored (once) into the host-mapped word
                    // eos_flags[b] -- the host reads it two steps behind the GPU instead of draining the stream every 16 steps
                    int eos_id; int* eos_flags[GVL_MAX_DECODE_BATCH];
                    float inv_temp, top_p; int top_k; unsigned seed_lo, seed_hi; unsigned stream[GVL_MAX_DECODE_BATCH];
                    const int* step_override; };   // operator tests: generation step of row b when the row has no ngen counter
int gvl_launch_argmax(const ArgmaxArgs& a, hipStream_t st);
int gvl_launch_sample(const ArgmaxArgs& a, hipStream_t st);
// x[b][:] = table[*tok_ptrs[b]][:]  and  (*pos_ptrs[b])++ helpers of the batched decode loop
struct TokPtrs { const int* p[GVL_MAX_DECODE_BATCH]; int n; };
int gvl_launch_gather_tok_rows(const bf16_t* table, const TokPtrs& toks, bf16_t* dst, int cols, hipStream_t st);
// x[b] = table[*toks.p[b]] and xn[b] = bf16(w * bf16(x[b] * rstd)): embedding gather + the first layer's input RMSNorm (gvl_decode.hip)
int gvl_launch_embed_norm(const bf16_t* table, const TokPtrs& toks, bf16_t* x, bf16_t* xn, const bf16_t* w, int cols, float eps, hipStream_t st);
// xn (B-operand tile order) = rmsnorm(x[b]) * w for the row-major residual rows x [batch][cols]: one wave per sequence (gvl_decode.hip)
int gvl_launch_norm_tiled(bf16_t* x, bf16_t* xn, const bf16_t* w, int batch, int cols, float eps, hipStream_t st);
struct IntPtrs { int* p[GVL_MAX_DECODE_BATCH]; int n; };
int gvl_launch_inc_many(const IntPtrs& ptrs, hipStream_t st);
int gvl_launch_inc(int* p, hipStream_t st);
int gvl_launch_kv_page_copy(bf16_t* kpool, bf16_t* vpool, size_t layer_stride, size_t page_elems, int layers, int src_page, int dst_page, hipStream_t st);
int gvl_launch_set_int(int* p, int v, hipStream_t st);

// ---- frame pre-processing (gvl_pre.hip) -------------------------------------------------------------------------
// frames uint8, layout 0 = [n][H][W][3] / 1 = [n][3][H][W] -> out f32 [n][3][size][size]; *scratch grows on demand (caller-owned).
int gvl_launch_preprocess(const unsigned char* frames, int n, int H, int W, int layout, int size, const float* mean, const float* stdv, float* out,
                          void** scratch, size_t* scratch_bytes, hipStream_t st);

