// gvl_model.h -- what the three host translation units of libgvl.so share: gvl_model.hip (context, weights, paged KV pool, the C ABI of include/gvl.h),
// gvl_vision.hip (the towers' launch sequences) and gvl_llm.hip (prefill, the decode step, the decode loop).  Internal: nothing here is part of the ABI.
#pragma once
#include "gvl_ctx.h"

namespace gvlm {

inline int fail(gvl_ctx* c, int code, const std::string& msg) { return gvl_fail(c, code, msg); }

inline int pad_head(int dr) { return dr <= 64 ? 64 : (dr <= 96 ? 96 : (dr <= 128 ? 128 : -1)); }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct ProfScope {
  gvl_ctx* c; hipStream_t st; int idx = -1;
  ProfScope(gvl_ctx* c_, int cat, double work, hipStream_t st_) : c(c_), st(st_) {
    if (!c->prof) return;
    ProfRec r; r.cat = cat; r.work = work;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1);
    hipEventRecord(r.e0, st);
    c->recs.push_back(r); idx = (int)c->recs.size() - 1;
  }
  ~ProfScope() { if (idx >= 0) hipEventRecord(c->recs[idx].e1, st); }
};
#define RUN(cat, work, expr) do { ProfScope _ps(ctx, cat, work, st); int _rc = (expr); if (_rc) return fail(ctx, _rc == -1 ? GVL_ERR_ARG : GVL_ERR_HIP, std::string("launch failed: ") + #expr); } while (0)

inline void* arena_alloc(gvl_ctx* c, size_t bytes) {
  const size_t off = al256(c->arena_off);
  if (off + bytes > c->arena_bytes) return nullptr;
  c->arena_off = off + bytes;
  return c->arena + off;
}
#define AALLOC(var, type, count) type* var = (type*)arena_alloc(ctx, (size_t)(count) * sizeof(type)); if (!var) return fail(ctx, GVL_ERR_OOM, "workspace arena too small for " #var)
inline void* arena_l_alloc(gvl_ctx* c, size_t bytes) {
  const size_t off = al256(c->arena_l_off);
  if (off + bytes > c->arena_l_bytes) return nullptr;
  c->arena_l_off = off + bytes;
  return c->arena_l + off;
}
#define LALLOC(var, type, count) type* var = (type*)arena_l_alloc(ctx, (size_t)(count) * sizeof(type)); if (!var) return fail(ctx, GVL_ERR_OOM, "LLM workspace arena too small for " #var)
// workspace arenas are bump allocators: a call takes a mark and every exit path -- errors included -- must give the space back
struct ArenaScope { size_t& off; const size_t mark; explicit ArenaScope(size_t& o) : off(o), mark(o) {} ~ArenaScope() { off = mark; } };

inline GemmArgs gemm(const bf16_t* A, int lda, const bf16_t* W, void* C, int ldc, int M, int N, int K) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.W = W; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  return g;
}

constexpr int kLossChunk = 128;    // rows of logits materialised at a time by the training-forward loss tail
// labelled rows of a training forward: device lists (row index into the sequence, target id) and the per-row nll output
struct LossReq { int n; const int* h_rows; const int* h_targets; float* h_nll; };

// workspace sizes of one call (gvl_create sizes the arenas from them)
size_t clip_bytes(const gvl_ctx* c, int n);
size_t iv2_bytes(const gvl_ctx* c, int n);
size_t visual_bytes(const gvl_ctx* c, int n);
size_t feats_bytes(const gvl_ctx* c, int n);
size_t prefill_bytes(const gvl_ctx* c, int S);

// gvl_vision.hip
int clip_encode(gvl_ctx* ctx, const float* px, int n, float* out, hipStream_t st);
int iv2_encode(gvl_ctx* ctx, const float* px, int n, bf16_t* out, hipStream_t st);
int build_visual(gvl_ctx* ctx, const float* clip_feats, const bf16_t* iv2_feats, int n, bf16_t* visual, hipStream_t st);

// gvl_llm.hip
int upload_table(gvl_ctx* ctx, Seq& s, hipStream_t st);
int pick_tokens(gvl_ctx* ctx, ArgmaxArgs& am, Seq* const* sqs, hipStream_t st);
int llm_prefill(gvl_ctx* ctx, Seq* const* sqs, int nb, const bf16_t* const* embeds, const int* lens, hipStream_t st, const LossReq* loss = nullptr, int pos0 = 0);
int decode_step(gvl_ctx* ctx, Seq* const* sqs, int B, hipStream_t st);
int decode_group_size(const gvl_ctx* ctx, int left);
int decode_group(gvl_ctx* ctx, Seq* const* sqs, int B, int max_new, int eos_id, int32_t* const* out_ids, int* const* n_out, hipStream_t st);

}  // namespace gvlm
