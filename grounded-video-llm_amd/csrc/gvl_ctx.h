// gvl_ctx.h -- the context object behind the C ABI (include/gvl.h) and the helpers every host-side translation unit of libgvl.so
// shares: packed-weight table, resolved weight pointers, workspace arenas, paged KV pool, per-sequence state, error reporting.
#pragma once
#include "gvl_internal.h"
#include "../../include/gvl.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

// text of the last gvl_create() failure (no ctx exists to carry it)
std::string& gvl_create_error();


struct Tensor { void* p = nullptr; int dtype = 0; int64_t numel = 0; std::vector<int64_t> shape; };

struct ClipLayerW { const float *ln1w, *ln1b, *ln2w, *ln2b, *qkvb, *outb, *fc1b, *fc2b; const bf16_t *qkvw, *outw, *fc1w, *fc2w; };
struct Iv2BlockW { const bf16_t *n1, *n2, *qkvw, *qn, *kn, *projw, *fc1w, *fc2w; const float *projb, *ls1, *ls2, *fc1b, *fc2b;
                   const bf16_t *qkvw_f = nullptr, *fc1w_f = nullptr; };   // fused RMSNorm: qkv.w diag(n1), fc1.w diag(n2) (gvl_launch_fold_gamma at finalize)
struct LlmLayerW { const bf16_t *ln1, *ln2, *qkvw, *ow, *guw, *downw;
                   const bf16_t *qkvw_f = nullptr, *guw_f = nullptr;   // fused RMSNorm (prefill): qkv.w diag(ln1), gate_up.w diag(ln2)
                   const bf16_t *qkvd_f = nullptr, *gud_f = nullptr;   // ... and their decode tile copies (bf16 decode weights only)
                   const bf16_t *qkvd, *od, *gud, *downd;      // decode copies in MFMA tile order (gvl_decode.hip; bf16, or FP8 e4m3 when cfg.decode_fp8); null on the VALU fallback
                   const float *qkvs, *os, *gus, *downs; };    // FP8 variant: per-row power-of-two scales

struct Seq {
  bool used = false; int max_tokens = 0, n_pages = 0; std::vector<int> pages;
  int* d_block_table = nullptr; int* d_pos = nullptr; int pos = 0; int n_gen = 0;
  bool table_dirty = false;   // `pages` changed on the host (alloc / fork): the device table is rewritten, stream ordered, by the first prefill / decode that uses the sequence
  int* d_tok = nullptr;   // the sequence's latest greedy token (input of its next decode step)
  int* d_out = nullptr;   // [outlist_cap] generated ids, index = generation step (device view of host-mapped memory)
  const int* h_out = nullptr;   // the same list as the host sees it (valid for entries whose step's stream work has completed)
  int* d_ngen = nullptr;  // device copy of n_gen: where the next generated id goes (a decode step carries no host counters)
  unsigned rng_stream = 0;  // sampling: which random stream this sequence draws from (assigned at its prefill)
  int* d_eos = nullptr; volatile int* h_eos = nullptr;   // host-mapped word: generation count at which this sequence produced eos (0 = not yet)
};

struct ProfRec { int cat; hipEvent_t e0, e1; double work; };

struct gvl_ctx {
  gvl_config cfg;
  std::string err;
  std::unordered_map<std::string, Tensor> w;
  bool finalized = false;
  // derived geometry
  int c_P = 0, c_S = 0, c_Kp = 0, c_Dr = 0, c_D = 0;
  int v_L = 0, v_TL = 0, v_S = 0, v_Kp = 0, v_Dr = 0, v_D = 0;
  int l_Dr = 0, l_D = 0, tok_per_seg = 0, img_tok = 0, seg_tok = 0;
  bool has_clip = false, has_iv2 = false, has_llm = false, has_proj = false;
  // resolved weights
  const bf16_t* c_patchw = nullptr; const float *c_cls = nullptr, *c_pos = nullptr, *c_prelnw = nullptr, *c_prelnb = nullptr;
  std::vector<ClipLayerW> cl;
  const bf16_t *v_patchw = nullptr, *v_cls = nullptr, *v_pos = nullptr; const float* v_patchb = nullptr;
  std::vector<Iv2BlockW> vb;
  const bf16_t *mm0w = nullptr, *mm1w = nullptr, *vp0w = nullptr, *vp1w = nullptr, *glb_gn = nullptr, *newline = nullptr;
  const float *mm0b = nullptr, *mm1b = nullptr, *vp0b = nullptr, *vp1b = nullptr, *sub_gn = nullptr;
  const bf16_t *l_embed = nullptr, *l_norm = nullptr, *l_headw = nullptr; const float* l_headb = nullptr;
  const float *cos_s = nullptr, *sin_s = nullptr, *cos_l = nullptr, *sin_l = nullptr;
  std::vector<LlmLayerW> ll;
  // arena
  char* arena = nullptr; size_t arena_bytes = 0, arena_off = 0;          // vision towers, glue, op-level entries
  char* arena_l = nullptr; size_t arena_l_bytes = 0, arena_l_off = 0;    // LLM prefill (own arena: may overlap vision on another stream)
  // KV pool
  bf16_t *kpool = nullptr, *vpool = nullptr; size_t layer_stride = 0; std::vector<int> free_pages;
  std::vector<int> page_ref;             // sequences holding each page: full pages of a shared prefix are referenced, never copied (gvl_seq_fork)
  std::vector<Seq> seqs;
  static constexpr int kMaxSeqs = 256;   // live sequences (slots of the device-side tables); the KV pool is the real limit
  int* d_seq_tables = nullptr; int* d_seq_pos = nullptr; int seq_table_cap = 0;   // [kMaxSeqs][seq_table_cap], [kMaxSeqs]
  // decode buffers
  bf16_t *d_x = nullptr, *d_qkv = nullptr, *d_q = nullptr, *d_attn = nullptr, *d_act = nullptr;
  bf16_t* d_xn = nullptr;            // [NB][hidden] RMS-normalised residual rows for the next projection (skinny-GEMM decode path)
  int* d_seq_ngen = nullptr;         // [kMaxSeqs]
  bool decode_mfma = false;          // geometry allows the skinny MFMA GEMM decode path (K % 256 == 0 for every projection)
  const bf16_t* l_headd = nullptr;   // lm_head in tile order
  const bf16_t* l_headd_f = nullptr; // lm_head diag(final norm weight) in tile order: fused RMSNorm on the decode path
  bf16_t* d_xt = nullptr;            // the decode step's residual rows a second time, raw, in B-operand tile order (written by o_proj / down_proj)
  float* d_sqpart = nullptr;         // [GVL_MAX_DECODE_BATCH][hidden / 16] partial sums of squares of those rows
  const float* l_heads = nullptr;    // its FP8 row scales
  int fp8 = 0;                       // format of the decode copies: 0 bf16, 1 FP8 e4m3 + row scales, 2 MXFP4 (cfg.decode_fp8 and the geometry allows it)
  std::vector<void*> dw_allocs;      // tile-order weight copies owned by the ctx
  std::vector<void*> pw_allocs;      // tile-order patch-conv weights (gvl_patch.hip)
  std::vector<void*> nf_allocs;      // norm-folded projection weights (fused RMSNorm)
  const bf16_t *c_patchwt = nullptr, *v_patchwt = nullptr;   // null: the tower's geometry takes the three-pass patch embedding
  // (all decode work buffers hold GVL_MAX_DECODE_BATCH rows: one per sequence of a batched decode step)
  float *d_logits = nullptr, *d_part = nullptr; int* d_counters = nullptr; int *d_seq_tok = nullptr, *d_seq_out = nullptr;
  int* h_seq_out = nullptr;          // d_seq_out is HOST-MAPPED memory (round 5): the selection kernels write the generated ids where the host reads them -- no read-back copy per call
  int nsplit = 16, outlist_cap = 8192, ids_cap = 16384;
  // frame pre-processing scratch (tmp image + tap tables), grown on demand
  void* pre_scratch = nullptr; size_t pre_scratch_bytes = 0;
  int kv_total_pages = 0;
  // eos watch of gvl_decode_greedy*: one host-mapped word per sequence slot, written by the token-selection kernel
  int* h_eos_flags = nullptr; int* d_eos_flags = nullptr; int watch_eos = -1;
  hipEvent_t step_ev[3] = {nullptr, nullptr, nullptr};
  // token selection (gvl_set_sampling): greedy argmax unless `on`
  struct { bool on = false; float inv_temp = 1.f, top_p = 0.f; int top_k = 0; unsigned long long seed = 0; unsigned next_stream = 0; } sample;
  // result-neutral launch parameters (gvl_debug_set): 0 = the launcher's own choice.  decode_graph: a decode group's step is captured once and
  // replayed (hipGraph) for the following tokens -- the host pays one graph launch per token instead of ~165 kernel launches
  struct { int decode_attn_cpb = 0, decode_attn_hpb = 0; bool decode_graph = true; int vision_in_place = 1, prefill_group = 4, attn_ring = 0, attn_pipe = 1, attn_pipe_rows = 128, patch_fused = 1, varlen_attn = 1, norm_fused = 1, last_layer_tail = 1; } dbg;
  // RCCL communicator owned by the ctx (gvl_comm_init); the library is dlopen'ed on first use
  void* comm = nullptr; int comm_rank = 0, comm_world = 1;
  // profiling
  bool prof = false; std::vector<ProfRec> recs;
  double prof_ms[GVL_PROF_NCAT] = {0}, prof_work[GVL_PROF_NCAT] = {0}; int64_t prof_n[GVL_PROF_NCAT] = {0};
};

inline int gvl_fail(gvl_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else gvl_create_error() = msg;
  return code;
}
inline int gvl_hipfail(gvl_ctx* c, hipError_t e, const char* what) { return gvl_fail(c, GVL_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
#define HIPCHK(c, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return gvl_hipfail(c, _e, #expr); } while (0)
