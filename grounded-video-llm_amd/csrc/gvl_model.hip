// gvl_model.hip -- host side of libgvl.so: context, packed weights, workspace arena, paged KV pool,
// the three towers' launch sequences, prefill / greedy decode loop, and the C ABI of include/gvl.h.
//
// Launch sequences restate (file:line in the reference):
//   CLIP tower      models/modeling_clip.py:182-191,355-393,626-651,851   (23 of 24 layers, hidden_states[-2])
//   InternVideo2    models/internvideo2.py:680-684,721-725,970-1040      (39 of 40 blocks)
//   glue/projectors models/llava_next_video.py:454-489,507-564
//   splice          models/llava_next_video.py:568-596
//   LLM             models/modeling_phi3.py:1034-1095,1249-1383,1512-1526 / models/modeling_llama.py:699-760
//   generate()      models/llava_next_video.py:655-661 (greedy; transformers GenerationMixin [ext])
#include "gvl_ctx.h"

std::string& gvl_create_error() { static std::string e; return e; }

namespace {

int fail(gvl_ctx* c, int code, const std::string& msg) { return gvl_fail(c, code, msg); }

int pad_head(int dr) { return dr <= 64 ? 64 : (dr <= 96 ? 96 : (dr <= 128 ? 128 : -1)); }
int round_up(int x, int m) { return (x + m - 1) / m * m; }
size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

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

void* arena_alloc(gvl_ctx* c, size_t bytes) {
  const size_t off = al256(c->arena_off);
  if (off + bytes > c->arena_bytes) return nullptr;
  c->arena_off = off + bytes;
  return c->arena + off;
}
#define AALLOC(var, type, count) type* var = (type*)arena_alloc(ctx, (size_t)(count) * sizeof(type)); if (!var) return fail(ctx, GVL_ERR_OOM, "workspace arena too small for " #var)
void* arena_l_alloc(gvl_ctx* c, size_t bytes) {
  const size_t off = al256(c->arena_l_off);
  if (off + bytes > c->arena_l_bytes) return nullptr;
  c->arena_l_off = off + bytes;
  return c->arena_l + off;
}
#define LALLOC(var, type, count) type* var = (type*)arena_l_alloc(ctx, (size_t)(count) * sizeof(type)); if (!var) return fail(ctx, GVL_ERR_OOM, "LLM workspace arena too small for " #var)
// workspace arenas are bump allocators: a call takes a mark and every exit path -- errors included -- must give the space back
struct ArenaScope { size_t& off; const size_t mark; explicit ArenaScope(size_t& o) : off(o), mark(o) {} ~ArenaScope() { off = mark; } };

const Tensor* find(gvl_ctx* c, const std::string& n) { auto it = c->w.find(n); return it == c->w.end() ? nullptr : &it->second; }

template <typename T>
int need(gvl_ctx* c, const std::string& name, int dtype, int64_t numel, const T** out, bool optional = false) {
  const Tensor* t = find(c, name);
  if (!t) { if (optional) { *out = nullptr; return 0; } return fail(c, GVL_ERR_STATE, "missing weight: " + name); }
  if (t->dtype != dtype) return fail(c, GVL_ERR_ARG, "wrong dtype for weight: " + name);
  if (t->numel != numel) return fail(c, GVL_ERR_ARG, "wrong size for weight: " + name + " (have " + std::to_string(t->numel) + ", want " + std::to_string(numel) + ")");
  *out = (const T*)t->p;
  return 0;
}
#define NEED(name, dt, n, outp) do { int _r = need(ctx, name, dt, (int64_t)(n), outp); if (_r) return _r; } while (0)

GemmArgs gemm(const bf16_t* A, int lda, const bf16_t* W, void* C, int ldc, int M, int N, int K) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = A; g.lda = lda; g.W = W; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  return g;
}

// paged KV pool: [layer][page][KV][64][D] for K and for V^T, zero-initialised (padded keys / values must be finite)
int alloc_kv_pool(gvl_ctx* ctx, int pages) {
  const gvl_config& f = ctx->cfg;
  ctx->layer_stride = (size_t)pages * f.kv_heads * 64 * ctx->l_D;
  const size_t pool = ctx->layer_stride * f.layers * 2;
  if (hipMalloc((void**)&ctx->kpool, pool) != hipSuccess || hipMalloc((void**)&ctx->vpool, pool) != hipSuccess) {
    (void)hipGetLastError();
    return fail(ctx, GVL_ERR_OOM, "hipMalloc(kv pool) failed: " + std::to_string(pages) + " pages = " + std::to_string(2 * pool >> 20) + " MiB");
  }
  if (hipMemset(ctx->kpool, 0, pool) != hipSuccess || hipMemset(ctx->vpool, 0, pool) != hipSuccess) return fail(ctx, GVL_ERR_HIP, "hipMemset(kv pool) failed");
  ctx->free_pages.clear();
  for (int p = pages - 1; p >= 0; --p) ctx->free_pages.push_back(p);
  ctx->page_ref.assign(pages, 0);
  ctx->kv_total_pages = pages;
  return 0;
}
// pages that fit in the HBM still free now (weights and workspaces are resident): `frac` of it, minus a fixed reserve for the
// caller's own tensors (pixels, embeddings, logits) and the runtime
int auto_kv_pages(const gvl_ctx* ctx, double frac, size_t reserve) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return -1;
  const gvl_config& f = ctx->cfg;
  const size_t page_bytes = (size_t)f.kv_heads * 64 * ctx->l_D * 2 /*bf16*/ * 2 /*K and V^T*/ * f.layers;
  double usable = (double)free_b * frac - (double)reserve;
  if (usable < (double)page_bytes) return 1;
  double pages = usable / (double)page_bytes;
  const double cap = (double)gvl_ctx::kMaxSeqs * ((f.max_seq + 63) / 64);   // more than every slot at full context is never reachable
  if (pages > cap) pages = cap;
  return (int)pages;
}

// ---------------------------------------------------------------------------------------------------
size_t clip_bytes(const gvl_ctx* c, int n) {
  const gvl_config& f = c->cfg; const size_t M = (size_t)n * c->c_S, C = f.clip_hidden;
  const size_t tiles = (c->c_S + 63) / 64;
  size_t b = 0;
  b += al256(M * C * 4) + al256(M * C * 2) + al256(M * 3 * C * 2) + al256(M * C * 2) + al256(M * f.clip_inter * 2);
  b += al256((size_t)n * c->c_P * c->c_Kp * 2) + al256((size_t)n * c->c_P * C * 2);
  b += al256((size_t)n * f.clip_heads * c->c_S * c->c_D * 2) + 2 * al256((size_t)n * tiles * f.clip_heads * 64 * c->c_D * 2);
  return b + 4096;
}
size_t iv2_bytes(const gvl_ctx* c, int n) {
  const gvl_config& f = c->cfg; const size_t M = (size_t)n * c->v_S, C = f.iv2_dim;
  const size_t tiles = (c->v_S + 63) / 64;
  size_t b = 0;
  b += 2 * al256(M * C * 2) + al256(M * 3 * C * 2) + al256(M * C * 2) + al256(M * f.iv2_inter * 2);
  b += al256((size_t)n * c->v_TL * c->v_Kp * 2) + al256((size_t)n * c->v_TL * C * 2);
  b += al256((size_t)n * f.iv2_heads * c->v_S * c->v_D * 2) + 2 * al256((size_t)n * tiles * f.iv2_heads * 64 * c->v_D * 2);
  b += al256(M * 4);                                   // per-token RMS factor of q (iv2_encode: qrs)
  b += al256(M * ((C + 63) / 64) * 4) + al256(M * 4);  // fused RMSNorm: row sums of squares per 64-column block + the row scale
  return b + 4096;
}
size_t visual_bytes(const gvl_ctx* c, int n) {
  const gvl_config& f = c->cfg;
  const size_t cin = f.llm_kind == GVL_LLM_PHI3 ? 4 * (size_t)f.clip_hidden : (size_t)f.clip_hidden;
  size_t b = 0;
  b += al256((size_t)n * c->img_tok * cin * 2) + al256((size_t)n * c->img_tok * f.hidden * 2);
  b += al256((size_t)n * c->seg_tok * f.iv2_dim * 2) + al256((size_t)n * c->seg_tok * f.hidden * 2);
  b += 4 * al256((size_t)f.hidden * 2 + cin * 2);
  return b + 4096;
}
constexpr int kLossChunk = 128;    // rows of logits materialised at a time by the training-forward loss tail
// labelled rows of a training forward: device lists (row index into the sequence, target id) and the per-row nll output
struct LossReq { int n; const int* h_rows; const int* h_targets; float* h_nll; };
size_t prefill_bytes(const gvl_ctx* c, int S) {
  const gvl_config& f = c->cfg;
  const size_t qkvw = (size_t)(f.heads + 2 * f.kv_heads) * c->l_Dr;
  size_t b = 0;
  b += 2 * al256((size_t)S * f.hidden * 2) + al256((size_t)S * qkvw * 2) + al256((size_t)S * f.heads * c->l_Dr * 2);
  b += al256((size_t)S * f.inter * 2) + al256((size_t)f.heads * S * c->l_D * 2);
  b += al256((size_t)kLossChunk * f.vocab * 2) + al256((size_t)kLossChunk * f.hidden * 2) + 3 * al256((size_t)S * 4);   // loss tail (gvl_forward_loss)
  b += al256((size_t)S * ((f.hidden + 63) / 64) * 4) + al256((size_t)S * 4);                                          // fused RMSNorm: row statistics + row scale
  b += (size_t)GVL_MAX_PREFILL_BATCH * (al256((size_t)f.hidden * 2) * 2 + al256((size_t)f.inter * 2) + al256(((f.hidden + 63) / 64) * 4) + 256) + 1024;   // last-layer tail rows
  return b + 4096;
}
size_t feats_bytes(const gvl_ctx* c, int n) {
  return al256((size_t)n * c->c_P * c->cfg.clip_hidden * 4) + al256((size_t)n * c->v_TL * c->cfg.iv2_dim * 2) + 1024;
}

// ---------------------------------------------------------------------------------------------------
int clip_encode(gvl_ctx* ctx, const float* px, int n, float* out, hipStream_t st) {
  const gvl_config& f = ctx->cfg;
  const int C = f.clip_hidden, H = f.clip_heads, S = ctx->c_S, P = ctx->c_P, M = n * S, I = f.clip_inter, D = ctx->c_D, Dr = ctx->c_Dr;
  const int tiles = (S + 63) / 64;
  ArenaScope arena_scope(ctx->arena_off);
  AALLOC(x, float, (size_t)M * C); AALLOC(h, bf16_t, (size_t)M * C); AALLOC(qkv, bf16_t, (size_t)M * 3 * C);
  AALLOC(att, bf16_t, (size_t)M * C); AALLOC(mlp, bf16_t, (size_t)M * I);
  AALLOC(pA, bf16_t, (size_t)n * P * ctx->c_Kp); AALLOC(pO, bf16_t, (size_t)n * P * C);
  AALLOC(Q, bf16_t, (size_t)n * H * S * D); AALLOC(Kt, bf16_t, (size_t)n * tiles * H * 64 * D); AALLOC(Vt, bf16_t, (size_t)n * tiles * H * 64 * D);

  bool fused_done = false;
  if (ctx->c_patchwt && ctx->dbg.patch_fused) {   // ONE kernel: im2col in the operand loader, patch GEMM, CLS + position rows, pre_layrnorm (gvl_patch.hip)
    PatchEmbedArgs e; memset(&e, 0, sizeof(e)); e.px = px; e.Wt = ctx->c_patchwt; e.n_img = n; e.T = 1; e.image = f.clip_image; e.patch = f.clip_patch; e.C = C;
    e.M = n * P; e.S = S; e.mode = 0; e.cls_f32 = ctx->c_cls; e.pos_f32 = ctx->c_pos; e.lnw = ctx->c_prelnw; e.lnb = ctx->c_prelnb; e.eps = 1e-5f; e.x_f32 = x;
    int prc = 0;
    { ProfScope ps_(ctx, GVL_PROF_GEMM, 2.0 * n * P * (double)C * 3 * f.clip_patch * f.clip_patch, st); prc = gvl_launch_patch_embed(e, st); }
    if (prc == -3) return fail(ctx, GVL_ERR_HIP, "launch failed: gvl_launch_patch_embed (clip)");
    fused_done = prc == 0;                         // -1: geometry outside the fused kernel (e.g. > 4 G pixel elements per call) -> the three passes below
    if (!fused_done && ctx->prof && !ctx->recs.empty()) ctx->recs.back().work = 0;   // nothing was launched: the fallback's GEMM carries the flops
  }
  if (!fused_done) {
    RUN(GVL_PROF_OTHER, 0, gvl_launch_patchify(px, pA, n, 1, f.clip_image, f.clip_patch, ctx->c_Kp, st));
    { GemmArgs g = gemm(pA, ctx->c_Kp, ctx->c_patchw, pO, C, n * P, C, ctx->c_Kp);
      // algorithmic flops use the real K = 3*p*p, not the padded one
      RUN(GVL_PROF_GEMM, 2.0 * n * P * (double)C * 3 * f.clip_patch * f.clip_patch, gvl_launch_gemm(g, st)); }
    RUN(GVL_PROF_OTHER, 0, gvl_launch_clip_embed_ln(pO, ctx->c_cls, ctx->c_pos, ctx->c_prelnw, ctx->c_prelnb, x, n, P, C, 1e-5f, st));
  }
  const bool vt_pages = !ctx->dbg.vision_in_place || D == 128;     // gvl_debug_set: the round-2 path (V^T pages written by a transpose pass), bit-identical; head dims 97..128 always take it
  for (int l = 0; l < f.clip_layers_run; ++l) {
    const ClipLayerW& w = ctx->cl[l];
    RUN(GVL_PROF_OTHER, 0, gvl_launch_layernorm_f32(x, w.ln1w, w.ln1b, h, M, C, 1e-5f, st));
    { GemmArgs g = gemm(h, C, w.qkvw, qkv, 3 * C, M, 3 * C, C); g.bias = w.qkvb; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
    const bool in_place = ctx->dbg.vision_in_place == 1 && D == Dr && D == 64;     // q, k, v read by the attention kernel straight from the fused-qkv matrix
    if (!in_place) { QkvPostArgs q; memset(&q, 0, sizeof(q)); q.qkv = qkv; q.ld = 3 * C; q.Q = Q; q.Kt = Kt; q.Vt = vt_pages ? Vt : nullptr; q.B = n; q.S = S; q.H = H; q.KV = H; q.Dr = Dr; q.D = D; q.mode = 0;
      RUN(GVL_PROF_OTHER, 0, gvl_launch_qkv_post(q, st)); }
    { AttnArgs a; memset(&a, 0, sizeof(a)); a.Q = Q; a.Kt = Kt; a.Vt = Vt; if (!vt_pages) { a.Vrows = qkv + 2 * C; a.v_ld = 3 * C; }
      if (in_place) { a.Qrows = qkv; a.Krows = qkv + C; a.q_ld = a.k_ld = 3 * C; }
      a.O = att; a.B = n; a.H = H; a.KV = H; a.S = S; a.D = D; a.Dout = Dr;
      a.scale = 1.0f / sqrtf((float)Dr); a.causal = 0; RUN(GVL_PROF_ATTN, gvl_attn_flops(a), gvl_launch_attention(a, st)); }
    { GemmArgs g = gemm(att, C, w.outw, x, C, M, C, C); g.bias = w.outb; g.resid = x; g.ldr = C; g.out_f32 = 1; g.round_pre_resid = 1;
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
    RUN(GVL_PROF_OTHER, 0, gvl_launch_layernorm_f32(x, w.ln2w, w.ln2b, h, M, C, 1e-5f, st));
    { GemmArgs g = gemm(h, C, w.fc1w, mlp, I, M, I, C); g.bias = w.fc1b; g.act = GVL_ACT_QUICK_GELU; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
    { GemmArgs g = gemm(mlp, I, w.fc2w, x, C, M, C, I); g.bias = w.fc2b; g.resid = x; g.ldr = C; g.out_f32 = 1; g.round_pre_resid = 1;
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
  }
  RUN(GVL_PROF_OTHER, 0, gvl_launch_strip_cls(x, out, n, S, C, 4, st));
  return 0;
}

int iv2_encode(gvl_ctx* ctx, const float* px, int n, bf16_t* out, hipStream_t st) {
  const gvl_config& f = ctx->cfg;
  const int C = f.iv2_dim, H = f.iv2_heads, S = ctx->v_S, TL = ctx->v_TL, M = n * S, I = f.iv2_inter, D = ctx->v_D, Dr = ctx->v_Dr;
  const int tiles = (S + 63) / 64;
  ArenaScope arena_scope(ctx->arena_off);
  AALLOC(x, bf16_t, (size_t)M * C); AALLOC(h, bf16_t, (size_t)M * C); AALLOC(qkv, bf16_t, (size_t)M * 3 * C);
  AALLOC(att, bf16_t, (size_t)M * C); AALLOC(mlp, bf16_t, (size_t)M * I);
  AALLOC(pA, bf16_t, (size_t)n * TL * ctx->v_Kp); AALLOC(pO, bf16_t, (size_t)n * TL * C);
  AALLOC(Q, bf16_t, (size_t)n * H * S * D); AALLOC(Kt, bf16_t, (size_t)n * tiles * H * 64 * D); AALLOC(Vt, bf16_t, (size_t)n * tiles * H * 64 * D);

  bool fused_done = false;
  if (ctx->v_patchwt && ctx->dbg.patch_fused) {   // ONE kernel: im2col in the operand loader, patch GEMM + bias, CLS + position rows (gvl_patch.hip)
    PatchEmbedArgs e; memset(&e, 0, sizeof(e)); e.px = px; e.Wt = ctx->v_patchwt; e.n_img = n; e.T = f.iv2_frames_per_seg; e.image = f.iv2_image; e.patch = f.iv2_patch; e.C = C;
    e.M = n * TL; e.S = S; e.mode = 1; e.bias = ctx->v_patchb; e.cls_bf = ctx->v_cls; e.pos_bf = ctx->v_pos; e.x_bf = x;
    int prc = 0;
    { ProfScope ps_(ctx, GVL_PROF_GEMM, 2.0 * n * TL * (double)C * 3 * f.iv2_patch * f.iv2_patch, st); prc = gvl_launch_patch_embed(e, st); }
    if (prc == -3) return fail(ctx, GVL_ERR_HIP, "launch failed: gvl_launch_patch_embed (iv2)");
    fused_done = prc == 0;
    if (!fused_done && ctx->prof && !ctx->recs.empty()) ctx->recs.back().work = 0;
  }
  if (!fused_done) {
    RUN(GVL_PROF_OTHER, 0, gvl_launch_patchify(px, pA, n, f.iv2_frames_per_seg, f.iv2_image, f.iv2_patch, ctx->v_Kp, st));
    { GemmArgs g = gemm(pA, ctx->v_Kp, ctx->v_patchw, pO, C, n * TL, C, ctx->v_Kp); g.bias = ctx->v_patchb;
      RUN(GVL_PROF_GEMM, 2.0 * n * TL * (double)C * 3 * f.iv2_patch * f.iv2_patch, gvl_launch_gemm(g, st)); }
    RUN(GVL_PROF_OTHER, 0, gvl_launch_iv2_embed(pO, ctx->v_cls, ctx->v_pos, x, n, TL, C, st));
  }
  const bool vt_pages = !ctx->dbg.vision_in_place || D == 128, q_in_place = ctx->dbg.vision_in_place == 1 && D == 96 && Dr == 88;
  AALLOC(qrs, float, (size_t)M);
  // Fused RMSNorm (gvl_debug_set("norm_fused"), default on): the two norm passes of a block (read x, write h: 1.1 GB each at 96 segments) are gone.  The
  // GEMM that writes the residual stream (proj / fc2) leaves the row sums of squares of its bf16 outputs per 64-column block (GemmArgs.rowsq), a tiny
  // kernel turns them into rs[m] = rsqrt(mean + eps), and the consuming GEMM (qkv / fc1) reads the RAW stream x with the norm weight folded into its
  // weight and multiplies its accumulator rows by rs (GemmArgs.rowscale).  Block 0's first norm has no producer GEMM and keeps the pass.
  const int NBLK = C / 64;
  const bool nf = ctx->dbg.norm_fused && C % 64 == 0 && (3 * C) % 16 == 0 && f.iv2_inter % 16 == 0 && !ctx->vb.empty() && ctx->vb[0].qkvw_f;   // widths the staged (whole-row) epilogue takes
  AALLOC(sq, float, (size_t)M * (nf ? NBLK : 1)); AALLOC(nrs, float, (size_t)M);
  // (Taking InternVideo2's q / k RMSNorm statistics the same way -- row sums of squares of the qkv GEMM's 3 C outputs, qkv_post reading k only -- was built
  //  and measured a net loss: +0.8 ms of GEMM per clip for the 66 blocks per row, two more small launches per block, and a K pass that is bound by its
  //  scattered page writes, not by the q read it lost.  profiles/r05_ab_norm_fused_with_qk_stats.json; removed.)
  for (int l = 0; l < f.iv2_blocks_run; ++l) {
    const Iv2BlockW& w = ctx->vb[l];
    if (nf && l > 0) {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rowsq_finish(sq, NBLK, 0, NBLK, nrs, M, C, 1e-6f, st));
      GemmArgs g = gemm(x, C, w.qkvw_f, qkv, 3 * C, M, 3 * C, C); g.rowscale = nrs;
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    } else {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rmsnorm_bf16(x, w.n1, h, M, C, 1e-6f, st));
      GemmArgs g = gemm(h, C, w.qkvw, qkv, 3 * C, M, 3 * C, C);
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    }
    { QkvPostArgs q; memset(&q, 0, sizeof(q)); q.qkv = qkv; q.ld = 3 * C; q.Q = Q; q.Kt = Kt; q.Vt = vt_pages ? Vt : nullptr; q.B = n; q.S = S; q.H = H; q.KV = H; q.Dr = Dr; q.D = D;
      q.mode = 1; q.qn = w.qn; q.kn = w.kn; q.eps = 1e-6f; q.ones_row = D > Dr ? 1 : 0; q.q_rs = q_in_place ? qrs : nullptr; q.k_ones = D > Dr ? 1 : 0;
      RUN(GVL_PROF_OTHER, 0, gvl_launch_qkv_post(q, st)); }
    { AttnArgs a; memset(&a, 0, sizeof(a)); a.Q = Q; a.Kt = Kt; a.Vt = Vt; if (!vt_pages) { a.Vrows = qkv + 2 * C; a.v_ld = 3 * C; } a.O = att; a.B = n; a.H = H; a.KV = H; a.S = S; a.D = D; a.Dout = Dr;
      if (q_in_place) { a.Qrows = qkv; a.q_ld = 3 * C; a.q_rs = qrs; a.q_nw = w.qn; a.k_ones = 1; a.pipe = ctx->dbg.attn_pipe; a.pipe_rows = ctx->dbg.attn_pipe_rows; }      // q read in place, normalised by the attention prologue: no Q write pass
      a.scale = 1.0f / sqrtf((float)Dr); a.causal = 0; a.ones_row = D > Dr ? 1 : 0;   // head dim 88 padded to 96: the pad row of V^T carries the softmax row sum
      RUN(GVL_PROF_ATTN, gvl_attn_flops(a), gvl_launch_attention(a, st)); }
    { GemmArgs g = gemm(att, C, w.projw, x, C, M, C, C); g.bias = w.projb; g.gamma = w.ls1; g.resid = x; g.ldr = C;
      if (nf) { g.rowsq = sq; g.rowsq_ld = NBLK; }
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
    if (nf) {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rowsq_finish(sq, NBLK, 0, NBLK, nrs, M, C, 1e-6f, st));
      GemmArgs g = gemm(x, C, w.fc1w_f, mlp, I, M, I, C); g.bias = w.fc1b; g.act = GVL_ACT_GELU; g.rowscale = nrs; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    } else {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rmsnorm_bf16(x, w.n2, h, M, C, 1e-6f, st));
      GemmArgs g = gemm(h, C, w.fc1w, mlp, I, M, I, C); g.bias = w.fc1b; g.act = GVL_ACT_GELU; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    }
    { GemmArgs g = gemm(mlp, I, w.fc2w, x, C, M, C, I); g.bias = w.fc2b; g.gamma = w.ls2; g.resid = x; g.ldr = C;
      if (nf && l + 1 < f.iv2_blocks_run) { g.rowsq = sq; g.rowsq_ld = NBLK; }
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
  }
  RUN(GVL_PROF_OTHER, 0, gvl_launch_strip_cls(x, out, n, S, C, 2, st));
  return 0;
}

int build_visual(gvl_ctx* ctx, const float* clip_feats, const bf16_t* iv2_feats, int n, bf16_t* visual, hipStream_t st) {
  const gvl_config& f = ctx->cfg;
  const int Hd = f.hidden, L = ctx->tok_per_seg, IT = ctx->img_tok, ST = ctx->seg_tok, T = f.iv2_frames_per_seg;
  const bool phi = f.llm_kind == GVL_LLM_PHI3;
  const int cin = phi ? 4 * f.clip_hidden : f.clip_hidden;
  ArenaScope arena_scope(ctx->arena_off);
  AALLOC(A1, bf16_t, (size_t)n * IT * cin); AALLOC(T1, bf16_t, (size_t)n * IT * Hd);
  AALLOC(A2, bf16_t, (size_t)n * ST * f.iv2_dim); AALLOC(T2, bf16_t, (size_t)n * ST * Hd);
  AALLOC(nl1, bf16_t, Hd); AALLOC(nl2, bf16_t, Hd);
  if (phi) RUN(GVL_PROF_OTHER, 0, gvl_launch_hd_merge(clip_feats, ctx->sub_gn, A1, n, f.clip_hidden, st));
  else RUN(GVL_PROF_OTHER, 0, gvl_launch_pool_spatial(clip_feats, A1, n, f.clip_hidden, st));
  { GemmArgs g = gemm(A1, cin, ctx->mm0w, T1, Hd, n * IT, Hd, cin); g.bias = ctx->mm0b; g.act = GVL_ACT_GELU; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
  { GemmArgs g = gemm(T1, Hd, ctx->mm1w, visual, Hd, n * IT, Hd, Hd); g.bias = ctx->mm1b; g.grp_rows = IT; g.grp_stride = L; g.row_off = 0;
    RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
  RUN(GVL_PROF_OTHER, 0, gvl_launch_pool_temporal(iv2_feats, A2, n, T, f.iv2_dim, st));
  { GemmArgs g = gemm(A2, f.iv2_dim, ctx->vp0w, T2, Hd, n * ST, Hd, f.iv2_dim); g.bias = ctx->vp0b; g.act = GVL_ACT_GELU; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
  { GemmArgs g = gemm(T2, Hd, ctx->vp1w, visual, Hd, n * ST, Hd, Hd); g.bias = ctx->vp1b; g.grp_rows = ST; g.grp_stride = L; g.row_off = IT;
    RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
  if (phi) {   // glb_GN through the image projector (llava_next_video.py:560-561); one row, broadcast (App. C #5)
    { GemmArgs g = gemm(ctx->glb_gn, cin, ctx->mm0w, nl1, Hd, 1, Hd, cin); g.bias = ctx->mm0b; g.act = GVL_ACT_GELU; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
    { GemmArgs g = gemm(nl1, Hd, ctx->mm1w, nl2, Hd, 1, Hd, Hd); g.bias = ctx->mm1b; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
    RUN(GVL_PROF_OTHER, 0, gvl_launch_bcast_row(nl2, visual, n, L, IT + ST, Hd, st));
  } else {
    RUN(GVL_PROF_OTHER, 0, gvl_launch_bcast_row(ctx->newline, visual, n, L, IT + ST, Hd, st));
  }
  return 0;
}

// page ids of a batch of equal-length sequences, passed by value to a stream-ordered fill (no host buffer lifetime)
__global__ void fill_ints_kernel(int* dst, const IntList l) { for (int i = threadIdx.x; i < l.n; i += blockDim.x) dst[i] = l.v[i]; }

// A sequence's page ids -> its device block table, by value in the kernel arguments (256 per launch) on the stream that is about to use the table:
// no host -> device copy (the runtime implements small ones as a blit kernel plus a staging buffer) and no host buffer lifetime to respect.
int upload_table(gvl_ctx* ctx, Seq& s, hipStream_t st) {
  if (!s.table_dirty) return 0;
  const int np = (int)s.pages.size();
  for (int p0 = 0; p0 < np; p0 += 256) {
    IntList l; l.n = np - p0 < 256 ? np - p0 : 256;
    for (int i = 0; i < l.n; ++i) l.v[i] = s.pages[p0 + i];
    hipLaunchKernelGGL(fill_ints_kernel, dim3(1), dim3(256), 0, st, s.d_block_table + p0, l);
  }
  if (hipGetLastError() != hipSuccess) return fail(ctx, GVL_ERR_HIP, "block table upload failed");
  s.table_dirty = false;
  return 0;
}

// The next token of every row of `am`: argmax (greedy), or one draw per row when gvl_set_sampling switched sampling on
int pick_tokens(gvl_ctx* ctx, ArgmaxArgs& am, Seq* const* sqs, hipStream_t st) {
  if (!ctx->sample.on) return gvl_launch_argmax(am, st);
  am.inv_temp = ctx->sample.inv_temp; am.top_p = ctx->sample.top_p; am.top_k = ctx->sample.top_k;
  am.seed_lo = (unsigned)ctx->sample.seed; am.seed_hi = (unsigned)(ctx->sample.seed >> 32);
  for (int b = 0; b < am.batch; ++b) am.stream[b] = sqs[b]->rng_stream;
  return gvl_launch_sample(am, st);
}

// Prefill of nb = 1 .. 8 sequences together (lens[b] tokens each).  The decoder GEMMs run over the rows of all of them
// (packed back to back, no padding); RoPE / KV append / causal attention run per sequence on its own pages -- as ONE launch with a
// batch dimension when the lengths are equal, as nb launches otherwise.  Every kernel is batch-invariant, so each sequence's
// result is bit-identical to a prefill on its own.
// pos0 > 0 (one sequence only): EXTEND -- the sequence already holds pos0 tokens (a multiple of 64: whole pages, possibly shared with other
// sequences); the new rows take positions pos0 .. pos0 + len - 1 and attend to the cached prefix plus themselves.
int llm_prefill(gvl_ctx* ctx, Seq* const* sqs, int nb, const bf16_t* const* embeds, const int* lens, hipStream_t st, const LossReq* loss = nullptr, int pos0 = 0) {
  const gvl_config& f = ctx->cfg;
  const int Hd = f.hidden, H = f.heads, KV = f.kv_heads, Dr = ctx->l_Dr, D = ctx->l_D, I = f.inter;
  const int qkvw = (H + 2 * KV) * Dr;
  if (nb < 1 || nb > GVL_MAX_PREFILL_BATCH) return fail(ctx, GVL_ERR_ARG, "llm_prefill: batch must be 1 .. 8 sequences");
  if (pos0 != 0 && (nb != 1 || (pos0 & 63) || loss)) return fail(ctx, GVL_ERR_ARG, "llm_prefill: extend takes one sequence whose cached prefix is whole pages");
  for (int b = 0; b < nb; ++b) { const int rc = upload_table(ctx, *sqs[b], st); if (rc) return rc; }
  int off[GVL_MAX_PREFILL_BATCH + 1]; off[0] = 0;
  bool uniform = true;
  for (int b = 0; b < nb; ++b) { off[b + 1] = off[b] + lens[b]; uniform = uniform && lens[b] == lens[0]; }
  const int M = off[nb], S0 = lens[0], P0 = (S0 + 63) / 64;
  if (nb > 1 && uniform && nb * P0 > (int)(sizeof(IntList::v) / sizeof(int))) uniform = false;   // page-id list below holds 256 entries
  ArenaScope arena_scope(ctx->arena_l_off);
  LALLOC(x, bf16_t, (size_t)M * Hd); LALLOC(h, bf16_t, (size_t)M * Hd); LALLOC(qkv, bf16_t, (size_t)M * qkvw);
  LALLOC(att, bf16_t, (size_t)M * H * Dr); LALLOC(act, bf16_t, (size_t)M * I); LALLOC(Q, bf16_t, (size_t)M * H * D);
  const int* table = sqs[0]->d_block_table;
  int table_stride = sqs[0]->n_pages;
  if (nb > 1 && uniform) {   // [nb][P] page ids of the batch, written by a stream-ordered kernel (ids passed by value: no host buffer lifetime)
    LALLOC(tb, int, (size_t)nb * P0);
    IntList l; l.n = nb * P0;
    for (int b = 0; b < nb; ++b) for (int p = 0; p < P0; ++p) l.v[b * P0 + p] = sqs[b]->pages[p];
    hipLaunchKernelGGL(fill_ints_kernel, dim3(1), dim3(256), 0, st, tb, l);
    table = tb; table_stride = P0;
  }
  for (int b = 0; b < nb; ++b) RUN(GVL_PROF_OTHER, 0, gvl_launch_copy_bytes(embeds[b], x + (size_t)off[b] * Hd, (size_t)lens[b] * Hd * 2, st));
  // fused RMSNorm (see iv2_encode): o_proj / down_proj leave the row statistics of the new residual stream, qkv_proj / gate_up_proj consume the raw
  // stream with the norm weight folded in and scale their accumulator rows; layer 0's input norm keeps the pass
  const int NBLK = Hd / 64;
  const bool nf = ctx->dbg.norm_fused && Hd % 64 == 0 && ((f.heads + 2 * f.kv_heads) * ctx->l_Dr) % 16 == 0 && (2 * f.inter) % 16 == 0 && !ctx->ll.empty() && ctx->ll[0].qkvw_f;   // widths the staged epilogue takes
  LALLOC(sq, float, (size_t)M * (nf ? NBLK : 1)); LALLOC(nrs, float, (size_t)M);
  const bf16_t* tail_rows = nullptr;                  // [nb][Hd]: the sequences' last rows after the last layer, when only they went through its MLP
  // one (RoPE + KV append, attention) launch for the whole batch when the lengths agree, one per sequence otherwise
  const int n_att = (nb == 1 || uniform) ? 1 : nb;
  for (int l = 0; l < f.layers; ++l) {
    const LlmLayerW& w = ctx->ll[l];
    bf16_t* Kt = ctx->kpool + (size_t)l * ctx->layer_stride; bf16_t* Vt = ctx->vpool + (size_t)l * ctx->layer_stride;
    if (nf && l > 0) {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rowsq_finish(sq, NBLK, 0, NBLK, nrs, M, Hd, f.rms_eps, st));
      GemmArgs g = gemm(x, Hd, w.qkvw_f, qkv, qkvw, M, qkvw, Hd); g.rowscale = nrs; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    } else {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rmsnorm_bf16(x, w.ln1, h, M, Hd, f.rms_eps, st));
      GemmArgs g = gemm(h, Hd, w.qkvw, qkv, qkvw, M, qkvw, Hd); RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    }
    // ragged group: RoPE / KV append per sequence (HBM-bound passes), then ONE causal-attention grid over the query blocks of all sequences
    // (AttnArgs.vl_*; 8 launches of ~900 blocks on 768 block slots each -> one of ~7 000: the causal tail is paid once) -- bit-identical per row
    const bool vl_attn = n_att > 1 && ctx->dbg.varlen_attn && pos0 == 0;
    // round 6: RoPE / KV append / V^T pages of the group in ONE launch each as well (QkvPostArgs.vl_*: 2 launches per layer instead of 2 per sequence -- 25 + 14 us
    // launches of 3.5 k rows each, bit-identical per row); varlen_attn = 2 keeps them per sequence (round 5).  One table choice (LongRoPE short / long) per launch.
    bool vl_post = vl_attn && ctx->dbg.varlen_attn == 1;
    if (vl_post && f.rope_orig_max_pos > 0 && ctx->cos_l) {
      const bool l0 = lens[0] > f.rope_orig_max_pos;
      for (int u = 1; u < nb; ++u) vl_post = vl_post && (lens[u] > f.rope_orig_max_pos) == l0;
    }
    if (vl_post) {
      const bool use_long = f.rope_orig_max_pos > 0 && lens[0] > f.rope_orig_max_pos && ctx->cos_l;
      QkvPostArgs q; memset(&q, 0, sizeof(q)); q.qkv = qkv; q.ld = qkvw; q.Q = Q; q.Kt = Kt; q.Vt = Vt; q.B = 1; q.H = H; q.KV = KV; q.Dr = Dr; q.D = D; q.mode = 2;
      q.cos = use_long ? ctx->cos_l : ctx->cos_s; q.sin = use_long ? ctx->sin_l : ctx->sin_s; q.vl_n = nb;
      for (int u = 0; u < nb; ++u) { q.vl_rows[u] = off[u]; q.vl_tables[u] = sqs[u]->d_block_table; q.S = lens[u] > q.S ? lens[u] : q.S; }
      q.vl_rows[nb] = off[nb];
      RUN(GVL_PROF_OTHER, 0, gvl_launch_qkv_post(q, st));
    }
    for (int u = 0; u < n_att && !vl_post; ++u) {
      const int S = lens[u], B = n_att == 1 ? nb : 1;
      const int* tbl = n_att == 1 ? table : sqs[u]->d_block_table;
      const int tstride = n_att == 1 ? table_stride : sqs[u]->n_pages;
      // LongRoPE: short factors up to the original context, long factors past it (modeling_phi3.py:381-385), per sequence
      const bool use_long = f.rope_orig_max_pos > 0 && pos0 + S > f.rope_orig_max_pos && ctx->cos_l;
      bf16_t* Qu = Q + (size_t)off[u] * H * D;
      { QkvPostArgs q; memset(&q, 0, sizeof(q)); q.qkv = qkv + (size_t)off[u] * qkvw; q.ld = qkvw; q.Q = Qu; q.Kt = Kt; q.Vt = Vt; q.block_table = tbl; q.max_pages = tstride;
        q.B = B; q.S = S; q.H = H; q.KV = KV; q.Dr = Dr; q.D = D; q.mode = 2; q.cos = use_long ? ctx->cos_l : ctx->cos_s; q.sin = use_long ? ctx->sin_l : ctx->sin_s; q.pos0 = pos0;
        RUN(GVL_PROF_OTHER, 0, gvl_launch_qkv_post(q, st)); }
      if (vl_attn) continue;
      { AttnArgs a; memset(&a, 0, sizeof(a)); a.Q = Qu; a.Kt = Kt; a.Vt = Vt; a.O = att + (size_t)off[u] * H * Dr; a.block_table = tbl; a.max_pages = tstride;
        a.B = B; a.H = H; a.KV = KV; a.S = S; a.D = D; a.Dout = Dr; a.scale = 1.0f / sqrtf((float)Dr); a.causal = 1;
        if (pos0) { a.Sk = pos0 + S; a.qpos0 = pos0; }
        a.ring = ctx->dbg.attn_ring;
        RUN(GVL_PROF_ATTN, gvl_attn_flops(a), gvl_launch_attention(a, st)); }
    }
    if (vl_attn) {
      AttnArgs a; memset(&a, 0, sizeof(a)); a.Q = Q; a.Kt = Kt; a.Vt = Vt; a.O = att; a.B = 1; a.H = H; a.KV = KV; a.D = D; a.Dout = Dr;
      a.scale = 1.0f / sqrtf((float)Dr); a.causal = 1; a.vl_n = nb;
      double fl = 0;
      for (int u = 0; u < nb; ++u) { a.vl_rows[u] = off[u]; a.vl_tables[u] = sqs[u]->d_block_table; a.S = lens[u] > a.S ? lens[u] : a.S;
        AttnArgs one = a; one.S = lens[u]; one.vl_n = 0; fl += gvl_attn_flops(one); }
      a.vl_rows[nb] = off[nb]; a.max_pages = 0;
      RUN(GVL_PROF_ATTN, fl, gvl_launch_attention(a, st));
    }
    { GemmArgs g = gemm(att, H * Dr, w.ow, x, Hd, M, Hd, H * Dr); g.resid = x; g.ldr = Hd; if (nf) { g.rowsq = sq; g.rowsq_ld = NBLK; }
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
    // LAST layer, no loss request: nothing downstream reads the MLP output of any row but a sequence's last (the KV cache is complete after qkv_post, the
    // lm_head takes last rows only) -- gate_up / down run on the nb last rows alone: -2 x M x hidden x 3 inter flops (2.1 % of the prefill's GEMM work at
    // S = 3.5 k).  The rows are gathered (x and, fused norm, their row statistics); GEMM rows do not depend on their neighbours, so the logits are
    // bit-identical to the full pass (asserted; gvl_debug_set("last_layer_tail", 0) = the full pass).
    if (l == f.layers - 1 && !loss && ctx->dbg.last_layer_tail && (!nf || NBLK % 4 == 0)) {
      int ids[GVL_MAX_PREFILL_BATCH];
      for (int b = 0; b < nb; ++b) ids[b] = off[b + 1] - 1;
      LALLOC(xl, bf16_t, (size_t)nb * Hd); LALLOC(actl, bf16_t, (size_t)nb * I);
      RUN(GVL_PROF_OTHER, 0, gvl_launch_gather_rows_host_ids(x, ids, nb, xl, Hd, st));
      if (nf) {
        LALLOC(sql, float, (size_t)nb * NBLK); LALLOC(rsl, float, nb);
        RUN(GVL_PROF_OTHER, 0, gvl_launch_gather_rows_host_ids((const bf16_t*)sq, ids, nb, (bf16_t*)sql, NBLK * 2, st));      // a row of partial sums = NBLK floats
        RUN(GVL_PROF_OTHER, 0, gvl_launch_rowsq_finish(sql, NBLK, 0, NBLK, rsl, nb, Hd, f.rms_eps, st));
        GemmArgs g = gemm(xl, Hd, w.guw_f, actl, I, nb, 2 * I, Hd); g.act = GVL_ACT_SILU_MUL; g.rowscale = rsl; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
      } else {
        LALLOC(hl, bf16_t, (size_t)nb * Hd);
        RUN(GVL_PROF_OTHER, 0, gvl_launch_rmsnorm_bf16(xl, w.ln2, hl, nb, Hd, f.rms_eps, st));
        GemmArgs g = gemm(hl, Hd, w.guw, actl, I, nb, 2 * I, Hd); g.act = GVL_ACT_SILU_MUL; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
      }
      { GemmArgs g = gemm(actl, I, w.downw, xl, Hd, nb, Hd, I); g.resid = xl; g.ldr = Hd; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
      tail_rows = xl;
      continue;
    }
    if (nf) {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rowsq_finish(sq, NBLK, 0, NBLK, nrs, M, Hd, f.rms_eps, st));
      GemmArgs g = gemm(x, Hd, w.guw_f, act, I, M, 2 * I, Hd); g.act = GVL_ACT_SILU_MUL; g.rowscale = nrs; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    } else {
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rmsnorm_bf16(x, w.ln2, h, M, Hd, f.rms_eps, st));
      GemmArgs g = gemm(h, Hd, w.guw, act, I, M, 2 * I, Hd); g.act = GVL_ACT_SILU_MUL; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
    }
    { GemmArgs g = gemm(act, I, w.downw, x, Hd, M, Hd, I); g.resid = x; g.ldr = Hd; if (nf && l + 1 < f.layers) { g.rowsq = sq; g.rowsq_ld = NBLK; }
      RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
  }
  if (loss && loss->n > 0) {
    // training forward (llava_next_video.py:598-614 -> Phi3ForCausalLM.forward labels branch, modeling_phi3.py:1512-1539): only rows
    // whose NEXT token carries a label need logits.  gather -> final RMSNorm -> lm_head GEMM (+bias, bf16 logits as under
    // autocast) -> f32 cross entropy per row; the host adds the rows up in order (deterministic).
    if (nb != 1) return fail(ctx, GVL_ERR_ARG, "llm_prefill: loss tail takes one sequence");
    LALLOC(d_rows, int, loss->n); LALLOC(d_tgt, int, loss->n); LALLOC(d_nll, float, loss->n);
    LALLOC(hs, bf16_t, (size_t)kLossChunk * Hd); LALLOC(lg, bf16_t, (size_t)kLossChunk * f.vocab);
    HIPCHK(ctx, hipMemcpyAsync(d_rows, loss->h_rows, (size_t)loss->n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_tgt, loss->h_targets, (size_t)loss->n * 4, hipMemcpyHostToDevice, st));
    for (int r0 = 0; r0 < loss->n; r0 += kLossChunk) {
      const int n = loss->n - r0 < kLossChunk ? loss->n - r0 : kLossChunk;
      RUN(GVL_PROF_OTHER, 0, gvl_launch_gather_rows(x, d_rows + r0, hs, n, Hd, st));
      RUN(GVL_PROF_OTHER, 0, gvl_launch_rmsnorm_bf16(hs, ctx->l_norm, h, n, Hd, f.rms_eps, st));      // h is free by now; n <= S - 1 rows
      { GemmArgs g = gemm(h, Hd, ctx->l_headw, lg, f.vocab, n, f.vocab, Hd); g.bias = ctx->l_headb; RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st)); }
      RUN(GVL_PROF_OTHER, 0, gvl_launch_ce_rows(lg, f.vocab, d_tgt + r0, d_nll + r0, n, f.vocab, st));
    }
    HIPCHK(ctx, hipMemcpyAsync(loss->h_nll, d_nll, (size_t)loss->n * 4, hipMemcpyDeviceToHost, st));
  }
  // last-row-only lm_head: final RMSNorm fused into the GEMV; one weight stream for the nb last rows
  const bf16_t* last = x + (size_t)(S0 - 1) * Hd;
  int last_stride = S0 * Hd;
  if (tail_rows) { last = tail_rows; last_stride = Hd; }
  else if (n_att > 1) {                    // ragged: gather the nb last rows (h is free by now)
    for (int b = 0; b < nb; ++b) RUN(GVL_PROF_OTHER, 0, gvl_launch_copy_bytes(x + (size_t)(off[b + 1] - 1) * Hd, h + (size_t)b * Hd, (size_t)Hd * 2, st));
    last = h; last_stride = Hd;
  }
  for (int b0 = 0; b0 < nb;) {             // the GEMV holds 1, 2 or 4 vectors in LDS: chunks of 4 / 2 / 1 last rows (row results do not depend on the chunking)
    const int nbc = nb - b0 >= 4 ? 4 : (nb - b0 >= 2 ? 2 : 1);
    GemvArgs g; memset(&g, 0, sizeof(g)); g.W = ctx->l_headw; g.N = f.vocab; g.K = Hd; g.x = last + (size_t)b0 * last_stride; g.norm_w = ctx->l_norm; g.eps = f.rms_eps;
    g.batch = nbc; g.x_stride = last_stride; g.out_stride = f.vocab;
    g.bias = ctx->l_headb; g.out_f32 = ctx->d_logits + (size_t)b0 * f.vocab; RUN(GVL_PROF_GEMV, 2.0 * f.vocab * Hd, gvl_launch_gemv(g, st));
    b0 += nbc;
  }
  for (int b = 0; b < nb; ++b) RUN(GVL_PROF_OTHER, 0, gvl_launch_set_int(sqs[b]->d_ngen, 0, st));
  { ArgmaxArgs am; memset(&am, 0, sizeof(am)); am.logits = ctx->d_logits; am.n = f.vocab; am.batch = nb;
    for (int b = 0; b < nb; ++b) { am.tok_ptrs[b] = sqs[b]->d_tok; am.out_lists[b] = sqs[b]->d_out; am.ngen_ptrs[b] = sqs[b]->d_ngen; }   // first generated token
    for (int b = 0; b < nb; ++b) sqs[b]->rng_stream = ctx->sample.next_stream++;     // a fresh random stream per prefilled sequence
    RUN(GVL_PROF_OTHER, 0, pick_tokens(ctx, am, sqs, st)); }
  for (int b = 0; b < nb; ++b) {
    RUN(GVL_PROF_OTHER, 0, gvl_launch_set_int(sqs[b]->d_pos, pos0 + lens[b], st));
    sqs[b]->pos = pos0 + lens[b]; sqs[b]->n_gen = 1;
  }
  return 0;
}

// One greedy decode step for B sequences together (1..16 on the skinny-GEMM path, 1 / 2 / 4 on the VALU fallback): each weight
// matrix is streamed ONCE for the whole batch, attention / RoPE / KV append run per sequence on its own pages.  Every launch
// argument is a device pointer or a constant of the group: the step can be replayed (hipGraph) without host-side counters.
int decode_step(gvl_ctx* ctx, Seq* const* sqs, int B, hipStream_t st) {
  const gvl_config& f = ctx->cfg;
  const int Hd = f.hidden, H = f.heads, KV = f.kv_heads, Dr = ctx->l_Dr, D = ctx->l_D, I = f.inter;
  const int qkvw = (H + 2 * KV) * Dr;
  const bool mfma = ctx->decode_mfma;
  if (mfma ? (B < 1 || B > GVL_MAX_DECODE_BATCH) : (B != 1 && B != 2 && B != 4)) return fail(ctx, GVL_ERR_ARG, "decode_step: unsupported batch");
  for (int b = 0; b < B; ++b) { const int rc = upload_table(ctx, *sqs[b], st); if (rc) return rc; }   // (a no-op after the sequence's prefill; under capture it would be part of the graph -- never dirty there)
  TokPtrs tp; memset(&tp, 0, sizeof(tp)); tp.n = B; for (int b = 0; b < B; ++b) tp.p[b] = sqs[b]->d_tok;
  // RMSNorm in front of qkv / gate_up / lm_head: groups of <= 4 normalise inside the consumer (LDS, like the VALU kernel), larger
  // groups run one norm launch per projection whose output every block of the consumer shares (gvl_decode.hip header)
  const bool fused_norm = !mfma || B <= GVL_MAX_VALU_BATCH;
  if (fused_norm) RUN(GVL_PROF_OTHER, 0, gvl_launch_gather_tok_rows(ctx->l_embed, tp, ctx->d_x, Hd, st));
  else RUN(GVL_PROF_OTHER, 0, gvl_launch_embed_norm(ctx->l_embed, tp, ctx->d_x, ctx->d_xn, ctx->ll[0].ln1, Hd, f.rms_eps, st));
  double ctx_tokens = 0; for (int b = 0; b < B; ++b) ctx_tokens += sqs[b]->pos + 1;
  // Decode-attention launch shape.  A sequence always uses one context split per 4 pages of ITS OWN length and one partial per split
  // (its arithmetic never depends on the batch); how many block slots the grid offers (gsplit) and how many consecutive splits one
  // block works through (cpb) are free.  cpb stays 1: letting a block amortise its publish -> ticket tail over 8 / 16 pages was
  // measured neutral to slower (Phi-3.5, 3.5 k context, 16 sequences: 2631 tok/s at cpb 1, 2613 at 2, 2574 at 4; one sequence:
  // 455 / 445 / 408) -- at 5.5 TB/s over pages scattered through a 244 GB pool the page reads, not the tail, are the limit.
  // gvl_debug_set("decode_attn_cpb") overrides (tests).  Under stream capture the shape must stay valid for later steps: every slot.
  int gsplit = ctx->nsplit, cpb = 1, hpb = 0;
  { hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st == nullptr || hipStreamIsCapturing(st, &cs) != hipSuccess || cs == hipStreamCaptureStatusNone) {
      int nsb[GVL_MAX_DECODE_BATCH];
      for (int b = 0; b < B; ++b) { const int np = (sqs[b]->pos + 1 + 63) >> 6; const int n = (np + 3) >> 2; nsb[b] = n < 1 ? 1 : (n > ctx->nsplit ? ctx->nsplit : n); }
      const int force_cpb = ctx->dbg.decode_attn_cpb;          // gvl_debug_set: tests vary this result-neutral launch parameter
      if (force_cpb >= 1 && force_cpb <= 16) cpb = force_cpb;
      gsplit = 1; for (int b = 0; b < B; ++b) { const int g = (nsb[b] + cpb - 1) / cpb; gsplit = g > gsplit ? g : gsplit; }
      // grouped-query models: the whole group per block when that still gives >= ~1.5 blocks per CU, else fewer heads per block
      // (measured, Llama-3-8B at 3.5 k context: one sequence 268 / 278 / 273 tok/s at 4 / 2 / 1 heads per block, two sequences 520 / 525)
      const int G = H / KV;
      if (G > 1) {
        long splits = 0; for (int b = 0; b < B; ++b) splits += (nsb[b] + cpb - 1) / cpb;
        hpb = G;
        while (hpb > 2 && hpb % 2 == 0 && (long)(H / hpb) * splits < 400) hpb >>= 1;
        if (hpb == 2 && (long)(H / 2) * splits < 200) hpb = 1;
        const int fh = ctx->dbg.decode_attn_hpb;                 // gvl_debug_set
        if (fh >= 1 && G % fh == 0) hpb = fh;
      }
    } }
  auto proj = [&](GemvArgs& g, const float* wscale) {
    if (!mfma) return gvl_launch_gemv(g, st);
    if (ctx->fp8) { g.w_fp8 = ctx->fp8; g.wscale = wscale; }
    return gvl_launch_dgemm(g, st);
  };
  auto normed_input = [&](GemvArgs& g, const bf16_t* w) {       // the projection reads rmsnorm(d_x) * w
    if (fused_norm) { g.x = ctx->d_x; g.norm_w = w; g.eps = f.rms_eps; } else g.x = ctx->d_xn;
  };
  // Fused RMSNorm on the decode path (round 5; GemvArgs.sq_*): o_proj / down_proj leave per-sequence partial sums of squares of the new residual rows and a
  // raw tile-order copy of them; qkv_proj (layers >= 1), gate_up_proj and lm_head run on that raw copy with the norm weight folded into their (tile-order)
  // weights and scale their accumulators per sequence.  No norm launch (groups > 4: two per layer) and no in-block normalisation (groups <= 4) any more;
  // layer 0's input norm (no producer projection) keeps the old path.  bf16 decode weights only.
  const int nblk = Hd >> 4;
  const bool rs = mfma && !ctx->fp8 && ctx->dbg.norm_fused && ctx->l_headd_f && Hd % 64 == 0 && (nblk & 31) == 0 && nblk <= 256;
  auto rs_input = [&](GemvArgs& g) { g.x = ctx->d_xt; g.sq_in = ctx->d_sqpart; g.sq_n = nblk; g.eps = f.rms_eps; };
  auto rs_output = [&](GemvArgs& g) { if (rs) { g.sq_out = ctx->d_sqpart; g.out_tiled2 = ctx->d_xt; } };
  for (int l = 0; l < f.layers; ++l) {
    const LlmLayerW& w = ctx->ll[l];
    bf16_t* Kt = ctx->kpool + (size_t)l * ctx->layer_stride; bf16_t* Vt = ctx->vpool + (size_t)l * ctx->layer_stride;
    { GemvArgs g; memset(&g, 0, sizeof(g)); g.W = mfma ? w.qkvd : w.qkvw; g.N = qkvw; g.K = Hd; g.batch = B; g.x_stride = Hd;
      if (rs && l > 0) { g.W = w.qkvd_f; rs_input(g); } else normed_input(g, w.ln1);
      // fused epilogue: RoPE + Q write + paged-KV append (replaces a separate qkv_post launch per layer per token)
      g.rope_on = 1; g.cos_s = ctx->cos_s; g.sin_s = ctx->sin_s; g.cos_l = ctx->cos_l; g.sin_l = ctx->sin_l;
      g.rope_switch = ctx->cos_l ? f.rope_orig_max_pos : 0;
      for (int b = 0; b < B; ++b) { g.pos_ptrs[b] = sqs[b]->d_pos; g.tables[b] = sqs[b]->d_block_table; }
      g.Q = ctx->d_q; g.q_stride = H * D; g.Kt = Kt; g.Vt = Vt; g.H = H; g.KV = KV; g.Dr = Dr; g.D = D;
      RUN(GVL_PROF_GEMV, 2.0 * qkvw * Hd, proj(g, w.qkvs)); }
    { DecodeAttnArgs a; memset(&a, 0, sizeof(a)); a.q = ctx->d_q; a.q_stride = H * D; a.Kt = Kt; a.Vt = Vt;
      for (int b = 0; b < B; ++b) { a.tables[b] = sqs[b]->d_block_table; a.pos_ptrs[b] = sqs[b]->d_pos; }
      a.part = ctx->d_part; a.counters = ctx->d_counters; a.batch = B; a.gsplit = gsplit; a.cpb = cpb; a.hpb = hpb;
      a.out = ctx->d_attn; a.out_stride = H * Dr; a.out_tiled = mfma ? 1 : 0; a.H = H; a.KV = KV; a.D = D; a.Dout = Dr; a.nsplit = ctx->nsplit; a.scale = 1.0f / sqrtf((float)Dr);
      RUN(GVL_PROF_DECODE_ATTN, 4.0 * ctx_tokens * (double)KV * D, gvl_launch_decode_attention(a, st)); }
    { GemvArgs g; memset(&g, 0, sizeof(g)); g.W = mfma ? w.od : w.ow; g.N = Hd; g.K = H * Dr; g.x = ctx->d_attn; g.resid = ctx->d_x; g.out_bf16 = ctx->d_x;
      g.batch = B; g.x_stride = H * Dr; g.out_stride = Hd; rs_output(g);
      RUN(GVL_PROF_GEMV, 2.0 * Hd * H * Dr, proj(g, w.os)); }
    if (!fused_norm && !rs) RUN(GVL_PROF_OTHER, 0, gvl_launch_norm_tiled(ctx->d_x, ctx->d_xn, w.ln2, B, Hd, f.rms_eps, st));     // post_attention_layernorm
    { GemvArgs g; memset(&g, 0, sizeof(g)); g.W = mfma ? w.gud : w.guw; g.N = 2 * I; g.K = Hd; g.act = GVL_ACT_SILU_MUL; g.out_bf16 = ctx->d_act;
      g.batch = B; g.x_stride = Hd; g.out_stride = I; g.out_tiled = mfma ? 1 : 0;
      if (rs) { g.W = w.gud_f; rs_input(g); } else normed_input(g, w.ln2);
      RUN(GVL_PROF_GEMV, 4.0 * I * Hd, proj(g, w.gus)); }
    { GemvArgs g; memset(&g, 0, sizeof(g)); g.W = mfma ? w.downd : w.downw; g.N = Hd; g.K = I; g.x = ctx->d_act; g.resid = ctx->d_x; g.out_bf16 = ctx->d_x;
      g.batch = B; g.x_stride = I; g.out_stride = Hd; rs_output(g);
      RUN(GVL_PROF_GEMV, 2.0 * Hd * I, proj(g, w.downs)); }
    if (!fused_norm && !rs)   // the next layer's input_layernorm, or the final norm in front of lm_head
      RUN(GVL_PROF_OTHER, 0, gvl_launch_norm_tiled(ctx->d_x, ctx->d_xn, l + 1 < f.layers ? ctx->ll[l + 1].ln1 : ctx->l_norm, B, Hd, f.rms_eps, st));
  }
  { GemvArgs g; memset(&g, 0, sizeof(g)); g.W = mfma ? ctx->l_headd : ctx->l_headw; g.N = f.vocab; g.K = Hd; g.bias = ctx->l_headb;
    g.batch = B; g.x_stride = Hd; g.out_stride = f.vocab;
    if (rs) { g.W = ctx->l_headd_f; rs_input(g); } else normed_input(g, ctx->l_norm);
    g.out_f32 = ctx->d_logits; RUN(GVL_PROF_GEMV, 2.0 * f.vocab * Hd, proj(g, ctx->l_heads)); }
  { ArgmaxArgs am; memset(&am, 0, sizeof(am)); am.logits = ctx->d_logits; am.n = f.vocab; am.batch = B;   // token, output list, n_gen++ and pos++ on the device
    for (int b = 0; b < B; ++b) { am.tok_ptrs[b] = sqs[b]->d_tok; am.out_lists[b] = sqs[b]->d_out; am.ngen_ptrs[b] = sqs[b]->d_ngen; am.pos_ptrs[b] = sqs[b]->d_pos; }
    if (ctx->watch_eos >= 0) { am.eos_id = ctx->watch_eos; for (int b = 0; b < B; ++b) am.eos_flags[b] = sqs[b]->d_eos; }
    RUN(GVL_PROF_OTHER, 0, pick_tokens(ctx, am, sqs, st)); }
  for (int b = 0; b < B; ++b) { sqs[b]->pos += 1; sqs[b]->n_gen += 1; }
  return 0;
}
// largest group the decode path takes at once, and the group size for `left` waiting sequences
int decode_group_size(const gvl_ctx* ctx, int left) {
  if (ctx->decode_mfma) return left < GVL_MAX_DECODE_BATCH ? left : GVL_MAX_DECODE_BATCH;
  return left >= 4 ? 4 : (left >= 2 ? 2 : 1);
}

// A decode step's launches carry device pointers and group constants only, so ONE captured step can be replayed for the following
// tokens of the same group (hipGraph): the host pays one graph launch instead of ~165 kernel launches per token.
struct StepGraph {
  hipGraph_t g = nullptr; hipGraphExec_t e = nullptr; bool failed = false;
  ~StepGraph() { if (e) hipGraphExecDestroy(e); if (g) hipGraphDestroy(g); }
};
int decode_step_replay(gvl_ctx* ctx, Seq* const* sqs, int B, hipStream_t st, StepGraph& sg) {
  if (!ctx->dbg.decode_graph || ctx->prof || st == nullptr || sg.failed) return decode_step(ctx, sqs, B, st);
  if (!sg.e) {
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); sg.failed = true; return decode_step(ctx, sqs, B, st); }
    bool was_dirty[GVL_MAX_DECODE_BATCH];                      // upload_table clears table_dirty when it ENQUEUES the upload -- inside a capture that is a recording only
    for (int b = 0; b < B; ++b) was_dirty[b] = sqs[b]->table_dirty;
    const int rc = decode_step(ctx, sqs, B, st);               // recorded, not executed; the host counters advance once
    const hipError_t ce = hipStreamEndCapture(st, &sg.g);
    if (rc) return rc;
    if (ce != hipSuccess || hipGraphInstantiate(&sg.e, sg.g, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError(); sg.failed = true; sg.e = nullptr;
      // the capture never ran: the block tables it would have written are still unwritten -- the eager step below must upload them
      for (int b = 0; b < B; ++b) { sqs[b]->pos -= 1; sqs[b]->n_gen -= 1; if (was_dirty[b]) sqs[b]->table_dirty = true; }
      return decode_step(ctx, sqs, B, st);
    }
    HIPCHK(ctx, hipGraphLaunch(sg.e, st));
    return 0;
  }
  HIPCHK(ctx, hipGraphLaunch(sg.e, st));
  for (int b = 0; b < B; ++b) { sqs[b]->pos += 1; sqs[b]->n_gen += 1; }
  return 0;
}

// decode of one group (prefilled sequences at the same generation step) until every member hit eos / max_new / its capacity.
// eos: the token-selection kernel stores a sequence's generation count into its host-mapped flag word the moment it picks eos; the host
// keeps at most two steps ahead of the GPU (an event per step) and reads the flags after each event -- at most two steps are decoded in
// vain and the stream is never drained mid-answer (round 1 drained it every 16 steps and could run 15 steps past eos).
int decode_group(gvl_ctx* ctx, Seq* const* sqs, int B, int max_new, int eos_id, int32_t* const* out_ids, int* const* n_out, hipStream_t st) {
  const int start_gen = sqs[0]->n_gen;   // members of a group must be in the same generation step
  for (int b = 1; b < B; ++b) if (sqs[b]->n_gen != start_gen) return fail(ctx, GVL_ERR_STATE, "decode batch: sequences are at different generation steps");
  struct Watch { gvl_ctx* c; ~Watch() { c->watch_eos = -1; } } watch{ctx};
  bool done[GVL_MAX_DECODE_BATCH] = {false};
  if (eos_id >= 0) {
    // the tokens produced so far (the prefill's first token) were selected without a watch: look at them once, then arm the flags
    HIPCHK(ctx, hipStreamSynchronize(st));
    for (int b = 0; b < B; ++b) {
      memcpy(out_ids[b], sqs[b]->h_out, (size_t)start_gen * 4);          // host-mapped list; the stream was synchronised above
      for (int i = 0; i < start_gen && !done[b]; ++i) if (out_ids[b][i] == eos_id) done[b] = true;
      *sqs[b]->h_eos = 0;
    }
    ctx->watch_eos = eos_id;
  }
  // Only LIVE members are stepped: a sequence that produced eos (seen two steps late) or reached its own capacity / max_new leaves the
  // group, so it can neither truncate the answers of longer-running members nor append past its pages.  A sequence's ids do not depend
  // on the group it is decoded in (gvl_decode.hip), so shrinking the group changes no result; the captured step is re-recorded.
  // The live set is stepped as PARTS of sizes the decode path takes (decode_group_size: any size up to 16 on the skinny-MFMA path;
  // 4 / 2 / 1 on the VALU fallback, so three survivors of a group of four run as 2 + 1), one captured step graph per part.
  // Retired graphs are destroyed after the final synchronise only: up to two replays may still be in flight when a member leaves.
  struct Part { Seq* m[GVL_MAX_DECODE_BATCH]; int n; std::unique_ptr<StepGraph> sg; };
  std::vector<Part> parts;
  std::vector<std::unique_ptr<StepGraph>> retired;
  Seq* live[GVL_MAX_DECODE_BATCH]; int n_live = -1;
  int enq = 0;
  for (;;) {
    if (eos_id >= 0 && enq >= 2) {
      HIPCHK(ctx, hipEventSynchronize(ctx->step_ev[(enq - 2) % 3]));
      for (int b = 0; b < B; ++b) if (!done[b] && *sqs[b]->h_eos != 0) done[b] = true;
    }
    Seq* now[GVL_MAX_DECODE_BATCH]; int n_now = 0;
    for (int b = 0; b < B; ++b) if (!done[b] && sqs[b]->n_gen < max_new && sqs[b]->pos < sqs[b]->max_tokens) now[n_now++] = sqs[b];
    if (n_now == 0) break;
    if (n_now != n_live || memcmp(now, live, sizeof(Seq*) * n_now) != 0) {
      for (auto& p : parts) retired.push_back(std::move(p.sg));
      parts.clear();
      for (int o = 0; o < n_now;) {
        Part p; p.n = decode_group_size(ctx, n_now - o);
        memcpy(p.m, now + o, sizeof(Seq*) * p.n); p.sg.reset(new StepGraph());
        o += p.n; parts.push_back(std::move(p));
      }
      memcpy(live, now, sizeof(Seq*) * n_now); n_live = n_now;
    }
    for (auto& p : parts) {
      const int rc = decode_step_replay(ctx, p.m, p.n, st, *p.sg);
      if (rc) { (void)hipStreamSynchronize(st); return rc; }
    }
    if (eos_id >= 0) HIPCHK(ctx, hipEventRecord(ctx->step_ev[enq % 3], st));
    ++enq;
  }
  HIPCHK(ctx, hipStreamSynchronize(st));
  for (int b = 0; b < B; ++b) {
    const int n = sqs[b]->n_gen < max_new ? sqs[b]->n_gen : max_new;
    memcpy(out_ids[b], sqs[b]->h_out, (size_t)n * 4);
    int cut = n;
    if (eos_id >= 0) for (int i = 0; i < n; ++i) if (out_ids[b][i] == eos_id) { cut = i + 1; break; }
    *n_out[b] = cut;
  }
  return 0;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char* gvl_last_error(const gvl_ctx* ctx) { return ctx ? ctx->err.c_str() : gvl_create_error().c_str(); }

int gvl_device_info(char* arch_out, int arch_len, int* num_cus) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return GVL_ERR_NOGPU;
  hipDeviceProp_t p; int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return GVL_ERR_HIP;
  if (arch_out && arch_len > 0) { strncpy(arch_out, p.gcnArchName, arch_len - 1); arch_out[arch_len - 1] = 0; }
  if (num_cus) *num_cus = p.multiProcessorCount;
  return 0;
}

int gvl_create(const gvl_config* cfg, gvl_ctx** out) {
  if (!cfg || !out) return fail(nullptr, GVL_ERR_ARG, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, GVL_ERR_NOGPU, "no HIP device: libgvl has no CPU fallback");
  gvl_ctx* ctx = new gvl_ctx();
  ctx->cfg = *cfg;
  const gvl_config& f = ctx->cfg;
  ctx->has_clip = f.clip_hidden > 0 && f.clip_layers_run >= 0 && f.clip_heads > 0;
  ctx->has_iv2 = f.iv2_dim > 0 && f.iv2_heads > 0;
  ctx->has_llm = f.hidden > 0 && f.layers > 0;
  ctx->has_proj = ctx->has_clip && ctx->has_iv2 && f.hidden > 0;
  auto bad = [&](const char* m) { std::string s = m; delete ctx; return fail(nullptr, GVL_ERR_ARG, s); };
  if (ctx->has_clip) {
    if (f.clip_hidden % 64 || f.clip_inter % 64 || f.clip_hidden % f.clip_heads || f.clip_image % f.clip_patch) return bad("clip geometry: hidden/inter must be multiples of 64");
    const int g = f.clip_image / f.clip_patch;
    ctx->c_P = g * g; ctx->c_S = ctx->c_P + 1; ctx->c_Kp = round_up(3 * f.clip_patch * f.clip_patch, 64);
    ctx->c_Dr = f.clip_hidden / f.clip_heads; ctx->c_D = pad_head(ctx->c_Dr);
    if (ctx->c_D < 0 || (ctx->c_Dr & 7)) return bad("clip head dim unsupported");
  }
  if (ctx->has_iv2) {
    if (f.iv2_dim % 64 || f.iv2_inter % 64 || f.iv2_dim % f.iv2_heads || f.iv2_image % f.iv2_patch || f.iv2_frames_per_seg <= 0) return bad("iv2 geometry");
    const int g = f.iv2_image / f.iv2_patch;
    ctx->v_L = g * g; ctx->v_TL = ctx->v_L * f.iv2_frames_per_seg; ctx->v_S = ctx->v_TL + 1; ctx->v_Kp = round_up(3 * f.iv2_patch * f.iv2_patch, 64);
    ctx->v_Dr = f.iv2_dim / f.iv2_heads; ctx->v_D = pad_head(ctx->v_Dr);
    if (ctx->v_D < 0 || (ctx->v_Dr & 7)) return bad("iv2 head dim unsupported");
  }
  if (f.hidden > 0) {
    if (f.hidden % 64) return bad("llm hidden must be a multiple of 64");
    if (ctx->has_llm) {
      if (f.inter % 64 || f.hidden % f.heads || f.heads % f.kv_heads || f.vocab <= 0 || f.max_seq <= 0) return bad("llm geometry");
      ctx->l_Dr = f.hidden / f.heads; ctx->l_D = pad_head(ctx->l_Dr);
      if (ctx->l_D < 0 || (ctx->l_Dr & 15) || ((f.heads * ctx->l_Dr) % 64)) return bad("llm head dim unsupported");
    }
  }
  if (ctx->has_proj) {
    ctx->img_tok = f.llm_kind == GVL_LLM_PHI3 ? 156 : 64;
    ctx->seg_tok = f.iv2_frames_per_seg * 16;
    ctx->tok_per_seg = ctx->img_tok + ctx->seg_tok + 1;
  }
  // workspace arena
  const int ns = f.max_segs > 0 ? f.max_segs : 1;
  size_t need_b = 1 << 20;
  if (ctx->has_clip) need_b = std::max(need_b, clip_bytes(ctx, ns));
  if (ctx->has_iv2) need_b = std::max(need_b, iv2_bytes(ctx, ns));
  if (ctx->has_proj) need_b = std::max(need_b, visual_bytes(ctx, ns));
  if (ctx->has_clip && ctx->has_iv2) need_b += feats_bytes(ctx, ns);
  need_b += 64 << 20;   // slack for the operator-level test entry points
  ctx->arena_bytes = need_b;
  if (hipMalloc((void**)&ctx->arena, need_b) != hipSuccess) { delete ctx; return fail(nullptr, GVL_ERR_OOM, "hipMalloc(arena) failed"); }
  if (ctx->has_llm && f.max_prefill > 0) {
    ctx->arena_l_bytes = prefill_bytes(ctx, f.max_prefill) + (1 << 20);
    if (hipMalloc((void**)&ctx->arena_l, ctx->arena_l_bytes) != hipSuccess) { gvl_destroy(ctx); return fail(nullptr, GVL_ERR_OOM, "hipMalloc(LLM arena) failed"); }
  }
  // KV pool + decode buffers.  cfg.kv_pages > 0: a pool of exactly that many pages, now.  cfg.kv_pages <= 0: the pool is sized from
  // the HBM that is still free once the weights are resident (gvl_finalize_weights) -- "paged KV cache sized for 288 GB".
  if (ctx->has_llm) {
    if (f.kv_pages > 0) { const int rc = alloc_kv_pool(ctx, f.kv_pages); if (rc) { std::string e = ctx->err; gvl_destroy(ctx); return fail(nullptr, rc, e); } }
    const int qkvw = (f.heads + 2 * f.kv_heads) * ctx->l_Dr;
    bool ok = true;
    const size_t NB = GVL_MAX_DECODE_BATCH;
    ok &= hipMalloc((void**)&ctx->d_x, NB * f.hidden * 2) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_xn, NB * f.hidden * 2) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_seq_ngen, (size_t)gvl_ctx::kMaxSeqs * 4) == hipSuccess;
    ok &= hipHostMalloc((void**)&ctx->h_eos_flags, (size_t)gvl_ctx::kMaxSeqs * 4, hipHostMallocMapped) == hipSuccess;
    if (ok) { memset(ctx->h_eos_flags, 0, (size_t)gvl_ctx::kMaxSeqs * 4); ok &= hipHostGetDevicePointer((void**)&ctx->d_eos_flags, ctx->h_eos_flags, 0) == hipSuccess; }
    for (int i = 0; i < 3 && ok; ++i) ok &= hipEventCreateWithFlags(&ctx->step_ev[i], hipEventDisableTiming) == hipSuccess;
    {   // the skinny-GEMM decode path needs every projection's K to split over 8 waves x 32-wide MFMA steps, and rows that one wave normalises
      const char* e = gvl_lab_env("GVL_DECODE_VALU");
      ctx->decode_mfma = !(e && atoi(e)) && f.hidden % 256 == 0 && f.inter % 256 == 0 && (f.heads * ctx->l_Dr) % 256 == 0 && f.hidden <= 4096 && (ctx->l_Dr & 1) == 0;
    }
    ok &= hipMalloc((void**)&ctx->d_qkv, (size_t)qkvw * 2) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_q, NB * f.heads * ctx->l_D * 2) == hipSuccess && hipMemset(ctx->d_q, 0, NB * f.heads * ctx->l_D * 2) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_attn, NB * f.heads * ctx->l_Dr * 2) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_act, NB * f.inter * 2) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_logits, NB * f.vocab * 4) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_part, NB * f.heads * ctx->nsplit * (ctx->l_D + 2) * 4) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_counters, NB * f.heads * 4) == hipSuccess && hipMemset(ctx->d_counters, 0, NB * f.heads * 4) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_xt, NB * f.hidden * 2) == hipSuccess && hipMemset(ctx->d_xt, 0, NB * f.hidden * 2) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_sqpart, NB * ((f.hidden + 15) / 16) * 4) == hipSuccess && hipMemset(ctx->d_sqpart, 0, NB * ((f.hidden + 15) / 16) * 4) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_seq_tok, (size_t)gvl_ctx::kMaxSeqs * 4) == hipSuccess;
    // generated ids live in host-mapped memory: 4 bytes per token cross PCIe as they are produced, and gvl_decode_greedy* / gvl_seq_read read them after
    // their stream sync without a device -> host copy (the step's trace holds no runtime blit kernel)
    ok &= hipHostMalloc((void**)&ctx->h_seq_out, (size_t)gvl_ctx::kMaxSeqs * ctx->outlist_cap * 4, hipHostMallocMapped) == hipSuccess &&
          hipHostGetDevicePointer((void**)&ctx->d_seq_out, ctx->h_seq_out, 0) == hipSuccess;
    ctx->seq_table_cap = (f.max_seq + 63) / 64;
    ok &= hipMalloc((void**)&ctx->d_seq_tables, (size_t)gvl_ctx::kMaxSeqs * ctx->seq_table_cap * 4) == hipSuccess;
    ok &= hipMalloc((void**)&ctx->d_seq_pos, (size_t)gvl_ctx::kMaxSeqs * 4) == hipSuccess;
    if (!ok) { gvl_destroy(ctx); return fail(nullptr, GVL_ERR_OOM, "hipMalloc(decode buffers) failed"); }
  }
  *out = ctx;
  return 0;
}

int gvl_destroy(gvl_ctx* ctx) {
  if (!ctx) return 0;
  hipDeviceSynchronize();
  if (ctx->h_eos_flags) hipHostFree(ctx->h_eos_flags);
  if (ctx->h_seq_out) hipHostFree(ctx->h_seq_out);
  for (int i = 0; i < 3; ++i) if (ctx->step_ev[i]) hipEventDestroy(ctx->step_ev[i]);
  for (auto& kv : ctx->w) if (kv.second.p) hipFree(kv.second.p);
  for (void* p : ctx->dw_allocs) if (p) hipFree(p);
  for (void* p : ctx->pw_allocs) if (p) hipFree(p);
  ctx->pw_allocs.clear();
  for (void* p : ctx->nf_allocs) if (p) hipFree(p);
  ctx->nf_allocs.clear();
  if (ctx->comm) gvl_comm_destroy(ctx);
  void* ptrs[] = {ctx->d_xn, ctx->d_seq_ngen, ctx->arena, ctx->arena_l, ctx->kpool, ctx->vpool, ctx->d_x, ctx->d_qkv, ctx->d_q, ctx->d_attn, ctx->d_act, ctx->d_logits, ctx->d_part, ctx->d_counters, ctx->d_xt, ctx->d_sqpart, ctx->d_seq_tok, ctx->d_seq_tables, ctx->d_seq_pos, ctx->pre_scratch};
  for (void* p : ptrs) if (p) hipFree(p);
  for (auto& r : ctx->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  delete ctx;
  return 0;
}

int gvl_load_weight(gvl_ctx* ctx, const char* name, const void* data, int dtype, const int64_t* shape, int ndim, int is_device) {
  if (!ctx || !name || !data || ndim < 0 || ndim > 8) return fail(ctx, GVL_ERR_ARG, "gvl_load_weight: bad argument");
  if (dtype != GVL_F32 && dtype != GVL_BF16) return fail(ctx, GVL_ERR_ARG, "gvl_load_weight: dtype must be f32 or bf16");
  Tensor t; t.dtype = dtype; t.numel = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.numel *= shape[i]; }
  const size_t bytes = (size_t)t.numel * (dtype == GVL_F32 ? 4 : 2);
  auto it = ctx->w.find(name);
  if (it != ctx->w.end()) { hipFree(it->second.p); ctx->w.erase(it); }
  HIPCHK(ctx, hipMalloc(&t.p, bytes ? bytes : 16));
  HIPCHK(ctx, hipMemcpy(t.p, data, bytes, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  ctx->w[name] = t;
  ctx->finalized = false;
  return 0;
}

int gvl_finalize_weights(gvl_ctx* ctx) {
  if (!ctx) return GVL_ERR_ARG;
  const gvl_config& f = ctx->cfg;
  char nm[128];
  if (ctx->has_clip) {
    const int C = f.clip_hidden, I = f.clip_inter;
    NEED("clip.patch.w", GVL_BF16, (int64_t)C * ctx->c_Kp, &ctx->c_patchw);
    NEED("clip.cls", GVL_F32, C, &ctx->c_cls); NEED("clip.pos", GVL_F32, (int64_t)ctx->c_S * C, &ctx->c_pos);
    NEED("clip.preln.w", GVL_F32, C, &ctx->c_prelnw); NEED("clip.preln.b", GVL_F32, C, &ctx->c_prelnb);
    ctx->cl.assign(f.clip_layers_run, ClipLayerW());
    for (int l = 0; l < f.clip_layers_run; ++l) {
      ClipLayerW& w = ctx->cl[l];
#define CN(s) (snprintf(nm, sizeof nm, "clip.L%d." s, l), nm)
      NEED(CN("ln1.w"), GVL_F32, C, &w.ln1w); NEED(CN("ln1.b"), GVL_F32, C, &w.ln1b); NEED(CN("ln2.w"), GVL_F32, C, &w.ln2w); NEED(CN("ln2.b"), GVL_F32, C, &w.ln2b);
      NEED(CN("qkv.w"), GVL_BF16, (int64_t)3 * C * C, &w.qkvw); NEED(CN("qkv.b"), GVL_F32, 3 * C, &w.qkvb);
      NEED(CN("out.w"), GVL_BF16, (int64_t)C * C, &w.outw); NEED(CN("out.b"), GVL_F32, C, &w.outb);
      NEED(CN("fc1.w"), GVL_BF16, (int64_t)I * C, &w.fc1w); NEED(CN("fc1.b"), GVL_F32, I, &w.fc1b);
      NEED(CN("fc2.w"), GVL_BF16, (int64_t)C * I, &w.fc2w); NEED(CN("fc2.b"), GVL_F32, C, &w.fc2b);
#undef CN
    }
  }
  // fused RMSNorm (round 5): the norm weight is folded into the projection that consumes the norm -- W' = bf16(W diag(gamma)), a second copy beside the
  // original (the unfused path, gvl_debug_set("norm_fused", 0), and the decode tile copies keep reading the original).  Memory: the prefill copies are 1.1 GB
  // (InternVideo2-1B) / 5 GB (Phi-3.5) / 9 GB (Llama-3-8B); with bf16 decode weights their decode tile copies and the folded lm_head come on top: ~10 GB
  // (Phi-3.5) / ~18 GB (Llama-3-8B) of the 288 GB in all.  They are an OPTIMISATION: env GVL_NORM_FOLD=0 skips them, and a failed allocation releases every
  // folded copy made so far and carries on -- the launch sequences test the *_f pointers and take the separate norm pass.
  if (!ctx->nf_allocs.empty()) HIPCHK(ctx, hipDeviceSynchronize());
  for (void* p : ctx->nf_allocs) if (p) hipFree(p);
  ctx->nf_allocs.clear();
  bool nf_on = !(getenv("GVL_NORM_FOLD") && atoi(getenv("GVL_NORM_FOLD")) == 0);
  auto nf_give_up = [&]() {                          // out of memory: no folded copy at all (a half-folded model would mix the two rounding orders)
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();                     // fold / retile kernels may still be writing the copies
    for (void* p : ctx->nf_allocs) if (p) hipFree(p);
    ctx->nf_allocs.clear();
    for (auto& w : ctx->vb) w.qkvw_f = w.fc1w_f = nullptr;
    for (auto& w : ctx->ll) w.qkvw_f = w.guw_f = w.qkvd_f = w.gud_f = nullptr;
    ctx->l_headd_f = nullptr;
    nf_on = false;
    fprintf(stderr, "libgvl: no memory for the norm-folded weight copies: RMSNorm runs as separate passes\n");
  };
  auto folded = [&](const bf16_t* W, const bf16_t* gamma, long rows, int cols, const bf16_t** out) -> int {
    if (!nf_on) return 0;
    void* q = nullptr;
    if (hipMalloc(&q, (size_t)rows * cols * 2) != hipSuccess) { nf_give_up(); return 0; }
    ctx->nf_allocs.push_back(q);
    if (gvl_launch_fold_gamma(W, gamma, (bf16_t*)q, rows, cols, nullptr)) return fail(ctx, GVL_ERR_HIP, "fold_gamma launch failed");
    *out = (const bf16_t*)q;
    return 0;
  };
  // fused patch embedding (gvl_patch.hip): tile-order copies of the conv weights, for the geometries the kernel is built for.
  // A repeated finalize frees the previous copies: encodes still in flight on other streams may be reading them -- drain the device first.
  if (!ctx->pw_allocs.empty()) HIPCHK(ctx, hipDeviceSynchronize());
  for (void* p : ctx->pw_allocs) if (p) hipFree(p);
  ctx->pw_allocs.clear(); ctx->c_patchwt = ctx->v_patchwt = nullptr;
  auto patch_tiled = [&](const bf16_t* W, int C, int Kp, int p, int image, const bf16_t** out) -> int {
    if (p != 14 || image % p || (C != 1024 && C != 1408)) return 0;
    void* q = nullptr;
    if (hipMalloc(&q, (size_t)(C / 16) * (3 * p / 2) * 512 * 2) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, GVL_ERR_OOM, "hipMalloc(patch weight copy) failed"); }
    ctx->pw_allocs.push_back(q);
    if (gvl_retile_patch_weight(W, (bf16_t*)q, C, Kp, p, nullptr)) return fail(ctx, GVL_ERR_HIP, "patch weight retile launch failed");
    *out = (const bf16_t*)q;
    return 0;
  };
  if (ctx->has_clip && f.clip_hidden == 1024) { const int rc = patch_tiled(ctx->c_patchw, f.clip_hidden, ctx->c_Kp, f.clip_patch, f.clip_image, &ctx->c_patchwt); if (rc) return rc; }
  if (ctx->has_iv2) {
    const int C = f.iv2_dim, I = f.iv2_inter;
    NEED("iv2.patch.w", GVL_BF16, (int64_t)C * ctx->v_Kp, &ctx->v_patchw); NEED("iv2.patch.b", GVL_F32, C, &ctx->v_patchb);
    { const int rc = patch_tiled(ctx->v_patchw, C, ctx->v_Kp, f.iv2_patch, f.iv2_image, &ctx->v_patchwt); if (rc) return rc; }
    NEED("iv2.cls", GVL_BF16, C, &ctx->v_cls); NEED("iv2.pos", GVL_BF16, (int64_t)ctx->v_S * C, &ctx->v_pos);
    ctx->vb.assign(f.iv2_blocks_run, Iv2BlockW());
    for (int l = 0; l < f.iv2_blocks_run; ++l) {
      Iv2BlockW& w = ctx->vb[l];
#define VN(s) (snprintf(nm, sizeof nm, "iv2.B%d." s, l), nm)
      NEED(VN("n1.w"), GVL_BF16, C, &w.n1); NEED(VN("n2.w"), GVL_BF16, C, &w.n2);
      NEED(VN("qkv.w"), GVL_BF16, (int64_t)3 * C * C, &w.qkvw); NEED(VN("qn.w"), GVL_BF16, C, &w.qn); NEED(VN("kn.w"), GVL_BF16, C, &w.kn);
      NEED(VN("proj.w"), GVL_BF16, (int64_t)C * C, &w.projw); NEED(VN("proj.b"), GVL_F32, C, &w.projb);
      NEED(VN("ls1"), GVL_F32, C, &w.ls1); NEED(VN("ls2"), GVL_F32, C, &w.ls2);
      NEED(VN("fc1.w"), GVL_BF16, (int64_t)I * C, &w.fc1w); NEED(VN("fc1.b"), GVL_F32, I, &w.fc1b);
      NEED(VN("fc2.w"), GVL_BF16, (int64_t)C * I, &w.fc2w); NEED(VN("fc2.b"), GVL_F32, C, &w.fc2b);
#undef VN
      if (C % 64 == 0) {
        int rc = folded(w.qkvw, w.n1, (long)3 * C, C, &w.qkvw_f);
        if (!rc) rc = folded(w.fc1w, w.n2, (long)I, C, &w.fc1w_f);
        if (rc) return rc;
      }
    }
  }
  if (ctx->has_proj) {
    const int Hd = f.hidden; const bool phi = f.llm_kind == GVL_LLM_PHI3; const int cin = phi ? 4 * f.clip_hidden : f.clip_hidden;
    NEED("mm.0.w", GVL_BF16, (int64_t)Hd * cin, &ctx->mm0w); NEED("mm.0.b", GVL_F32, Hd, &ctx->mm0b);
    NEED("mm.1.w", GVL_BF16, (int64_t)Hd * Hd, &ctx->mm1w); NEED("mm.1.b", GVL_F32, Hd, &ctx->mm1b);
    NEED("vp.0.w", GVL_BF16, (int64_t)Hd * f.iv2_dim, &ctx->vp0w); NEED("vp.0.b", GVL_F32, Hd, &ctx->vp0b);
    NEED("vp.1.w", GVL_BF16, (int64_t)Hd * Hd, &ctx->vp1w); NEED("vp.1.b", GVL_F32, Hd, &ctx->vp1b);
    if (phi) { NEED("sub_gn", GVL_F32, cin, &ctx->sub_gn); NEED("glb_gn", GVL_BF16, cin, &ctx->glb_gn); }
    else NEED("newline", GVL_BF16, Hd, &ctx->newline);
  }
  if (ctx->has_llm) {
    const int Hd = f.hidden, I = f.inter, Dr = ctx->l_Dr, qkvw = (f.heads + 2 * f.kv_heads) * Dr;
    NEED("llm.embed", GVL_BF16, (int64_t)f.vocab * Hd, &ctx->l_embed); NEED("llm.norm.w", GVL_BF16, Hd, &ctx->l_norm);
    NEED("llm.head.w", GVL_BF16, (int64_t)f.vocab * Hd, &ctx->l_headw);
    if (f.lm_head_bias) NEED("llm.head.b", GVL_F32, f.vocab, &ctx->l_headb); else ctx->l_headb = nullptr;
    NEED("rope.cos_s", GVL_F32, (int64_t)f.max_seq * (Dr / 2), &ctx->cos_s); NEED("rope.sin_s", GVL_F32, (int64_t)f.max_seq * (Dr / 2), &ctx->sin_s);
    if (f.rope_orig_max_pos > 0) { NEED("rope.cos_l", GVL_F32, (int64_t)f.max_seq * (Dr / 2), &ctx->cos_l); NEED("rope.sin_l", GVL_F32, (int64_t)f.max_seq * (Dr / 2), &ctx->sin_l); }
    else { ctx->cos_l = ctx->sin_l = nullptr; }
    ctx->ll.assign(f.layers, LlmLayerW());
    for (int l = 0; l < f.layers; ++l) {
      LlmLayerW& w = ctx->ll[l];
#define LN(s) (snprintf(nm, sizeof nm, "llm.L%d." s, l), nm)
      NEED(LN("ln1.w"), GVL_BF16, Hd, &w.ln1); NEED(LN("ln2.w"), GVL_BF16, Hd, &w.ln2);
      NEED(LN("qkv.w"), GVL_BF16, (int64_t)qkvw * Hd, &w.qkvw); NEED(LN("o.w"), GVL_BF16, (int64_t)Hd * f.heads * Dr, &w.ow);
      NEED(LN("gu.w"), GVL_BF16, (int64_t)2 * I * Hd, &w.guw); NEED(LN("down.w"), GVL_BF16, (int64_t)Hd * I, &w.downw);
#undef LN
    }
    // decode copies in MFMA tile order (one wave load = 1 KiB of consecutive addresses; the rotate_half row permutation of qkv
    // is baked in): 288 GB of HBM pay for the second copy of the LLM -- the prefill GEMM keeps the row-major one
    for (void* p : ctx->dw_allocs) if (p) hipFree(p);
    ctx->dw_allocs.clear();
    ctx->fp8 = 0;
    if (f.decode_fp8) {
      if (f.decode_fp8 != 1 && f.decode_fp8 != 2) return fail(ctx, GVL_ERR_ARG, "cfg.decode_fp8: 0 = bf16, 1 = FP8 e4m3, 2 = MXFP4");
      const int mult = f.decode_fp8 == 2 ? 1024 : 512;
      if (!ctx->decode_mfma || Hd % mult || I % mult || (f.heads * Dr) % mult)
        return fail(ctx, GVL_ERR_ARG, "cfg.decode_fp8 needs hidden, inter and heads*head_dim to be multiples of 512 (FP8) / 1024 (MXFP4) (skinny-GEMM decode path)");
      ctx->fp8 = f.decode_fp8;
    }
    if (ctx->decode_mfma) {
      // bf16: a re-tiled copy.  FP8: per-row scales + the e4m3 tile copy.  MXFP4: E8M0 block scales + the E2M1 tile copy.  In both
      // quantised formats the row-major weight is replaced by its de-quantised values (prefill and decode evaluate ONE model).
      auto tiled = [&](const bf16_t* W, int N, int K, int dr, int nqk, const bf16_t** out, const float** sc_out) -> int {
        void* p = nullptr;
        const size_t n16 = (size_t)((N + 15) / 16) * 16;
        const size_t bytes = ctx->fp8 == 2 ? n16 * K / 2 : n16 * K * (ctx->fp8 ? 1 : 2);
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, GVL_ERR_OOM, "hipMalloc(decode weight copy) failed"); }
        ctx->dw_allocs.push_back(p);
        *out = (const bf16_t*)p;
        if (ctx->fp8) {
          void* sc = nullptr;
          const size_t sbytes = ctx->fp8 == 2 ? n16 * (K / 32) : (size_t)N * 4;
          if (hipMalloc(&sc, sbytes) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, GVL_ERR_OOM, "hipMalloc(weight scales) failed"); }
          ctx->dw_allocs.push_back(sc);
          *sc_out = (const float*)sc;
          const int qrc = ctx->fp8 == 2 ? gvl_mxfp4_quantise_decode_weight(const_cast<bf16_t*>(W), (unsigned char*)p, (unsigned*)sc, N, K, dr, nqk, nullptr)
                                        : gvl_fp8_quantise_decode_weight(const_cast<bf16_t*>(W), (unsigned char*)p, (float*)sc, N, K, dr, nqk, nullptr);
          if (qrc) return fail(ctx, GVL_ERR_HIP, "weight quantise launch failed");
        } else {
          *sc_out = nullptr;
          if (gvl_retile_decode_weight(W, (bf16_t*)p, N, K, dr, nqk, nullptr)) return fail(ctx, GVL_ERR_HIP, "retile launch failed");
        }
        return 0;
      };
      for (int l = 0; l < f.layers; ++l) {
        LlmLayerW& w = ctx->ll[l];
        int rc = tiled(w.qkvw, qkvw, Hd, Dr, f.heads + f.kv_heads, &w.qkvd, &w.qkvs);
        if (!rc) rc = tiled(w.ow, Hd, f.heads * Dr, 0, 0, &w.od, &w.os);
        if (!rc) rc = tiled(w.guw, 2 * I, Hd, 0, 0, &w.gud, &w.gus);
        if (!rc) rc = tiled(w.downw, Hd, I, 0, 0, &w.downd, &w.downs);
        if (rc) return rc;
      }
      { const int rc = tiled(ctx->l_headw, f.vocab, Hd, 0, 0, &ctx->l_headd, &ctx->l_heads); if (rc) return rc; }
    }
    // norm-folded prefill weights LAST: the quantised decode formats above replace the row-major weights by their de-quantised values (same stream, in
    // order), and the fused-RMSNorm prefill must evaluate that same model
    if (Hd % 64 == 0) {
      for (int l = 0; l < f.layers; ++l) {
        LlmLayerW& w = ctx->ll[l];
        int rc = folded(w.qkvw, w.ln1, (long)qkvw, Hd, &w.qkvw_f);
        if (!rc) rc = folded(w.guw, w.ln2, (long)2 * I, Hd, &w.guw_f);
        if (rc) return rc;
      }
      // the decode path's copies of the folded weights (bf16 decode weights only: a quantised format would quantise gamma * W, another model than
      // the prefill's): tile order like every decode weight; plus lm_head with the final norm weight
      ctx->l_headd_f = nullptr;
      if (ctx->decode_mfma && !ctx->fp8 && nf_on) {
        auto tiled_f = [&](const bf16_t* W, int N, int K, int dr, int nqk, const bf16_t** out) -> int {
          if (!nf_on || !W) return 0;
          void* p = nullptr;
          if (hipMalloc(&p, (size_t)((N + 15) / 16) * 16 * K * 2) != hipSuccess) { nf_give_up(); return 0; }
          ctx->nf_allocs.push_back(p);
          if (gvl_retile_decode_weight(W, (bf16_t*)p, N, K, dr, nqk, nullptr)) return fail(ctx, GVL_ERR_HIP, "retile launch failed");
          *out = (const bf16_t*)p;
          return 0;
        };
        for (int l = 0; l < f.layers; ++l) {
          LlmLayerW& w = ctx->ll[l];
          int rc = tiled_f(w.qkvw_f, qkvw, Hd, Dr, f.heads + f.kv_heads, &w.qkvd_f);
          if (!rc) rc = tiled_f(w.guw_f, 2 * I, Hd, 0, 0, &w.gud_f);
          if (rc) return rc;
        }
        const bf16_t* headf = nullptr;
        int rc = folded(ctx->l_headw, ctx->l_norm, (long)f.vocab, Hd, &headf);
        if (!rc) rc = tiled_f(headf, f.vocab, Hd, 0, 0, &ctx->l_headd_f);
        if (rc) return rc;
      }
    }
  }
  // the retile kernels above (patch weights, decode tile copies) ran on the null stream: a first encode / decode on a NON-blocking stream (torch pool
  // streams, the bench's sV / sL) is not ordered behind them -- finalize returns only when every derived copy is complete
  HIPCHK(ctx, hipDeviceSynchronize());
  if (ctx->has_llm && !ctx->kpool) {          // cfg.kv_pages <= 0: size the pool from what is free NOW (weights resident)
    const char* fe = getenv("GVL_KV_FRACTION");
    const double frac = fe ? atof(fe) : 0.85;
    const int pages = auto_kv_pages(ctx, frac > 0 && frac <= 1 ? frac : 0.85, (size_t)4 << 30);
    if (pages <= 0) return fail(ctx, GVL_ERR_HIP, "hipMemGetInfo failed");
    const int rc = alloc_kv_pool(ctx, pages);
    if (rc) return rc;
  }
  ctx->finalized = true;
  return 0;
}

#define REQUIRE_READY(cond, what) do { if (!ctx) return GVL_ERR_ARG; if (!ctx->finalized || !(cond)) return fail(ctx, GVL_ERR_STATE, what ": weights not finalized or tower not configured"); } while (0)

int gvl_clip_encode(gvl_ctx* ctx, const float* px, int n, float* out, void* stream) {
  REQUIRE_READY(ctx->has_clip, "gvl_clip_encode");
  if (!px || !out || n <= 0 || n > std::max(1, ctx->cfg.max_segs)) return fail(ctx, GVL_ERR_ARG, "gvl_clip_encode: bad n/pointers");
  return clip_encode(ctx, px, n, out, (hipStream_t)stream);
}
int gvl_iv2_encode(gvl_ctx* ctx, const float* px, int n, uint16_t* out, void* stream) {
  REQUIRE_READY(ctx->has_iv2, "gvl_iv2_encode");
  if (!px || !out || n <= 0 || n > std::max(1, ctx->cfg.max_segs)) return fail(ctx, GVL_ERR_ARG, "gvl_iv2_encode: bad n/pointers");
  return iv2_encode(ctx, px, n, out, (hipStream_t)stream);
}
int gvl_tokens_per_seg(const gvl_ctx* ctx) { return ctx ? ctx->tok_per_seg : 0; }
int gvl_build_visual(gvl_ctx* ctx, const float* clip_feats, const uint16_t* iv2_feats, int n, uint16_t* visual, void* stream) {
  REQUIRE_READY(ctx->has_proj, "gvl_build_visual");
  if (!clip_feats || !iv2_feats || !visual || n <= 0 || n > std::max(1, ctx->cfg.max_segs)) return fail(ctx, GVL_ERR_ARG, "gvl_build_visual: bad n/pointers");
  if (ctx->c_P != 576) return fail(ctx, GVL_ERR_ARG, "gvl_build_visual: needs the 24x24 CLIP grid (llava_next_video.py:460)");
  if (ctx->v_L != 256) return fail(ctx, GVL_ERR_ARG, "gvl_build_visual: needs the 16x16 InternVideo2 grid");
  return build_visual(ctx, clip_feats, iv2_feats, n, visual, (hipStream_t)stream);
}
int gvl_encode_segments(gvl_ctx* ctx, const float* spatial_px, const float* temporal_px, int n, uint16_t* visual, void* stream) {
  REQUIRE_READY(ctx->has_proj, "gvl_encode_segments");
  if (!spatial_px || !temporal_px || !visual || n <= 0 || n > std::max(1, ctx->cfg.max_segs)) return fail(ctx, GVL_ERR_ARG, "gvl_encode_segments: bad n/pointers");
  hipStream_t st = (hipStream_t)stream;
  ArenaScope arena_scope(ctx->arena_off);
  AALLOC(cf, float, (size_t)n * ctx->c_P * ctx->cfg.clip_hidden);
  AALLOC(vf, bf16_t, (size_t)n * ctx->v_TL * ctx->cfg.iv2_dim);
  int rc = clip_encode(ctx, spatial_px, n, cf, st);
  if (!rc) rc = iv2_encode(ctx, temporal_px, n, vf, st);
  if (!rc) rc = gvl_build_visual(ctx, cf, vf, n, visual, stream);
  return rc;
}

int gvl_splice(gvl_ctx* ctx, const int64_t* ids, int n_ids, const uint16_t* visual, int n_visual, uint16_t* embeds, int* seq_len_out, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_splice");
  if (!ids || (!visual && n_visual > 0) || !embeds || n_ids <= 0 || n_visual < 0 || n_ids > ctx->ids_cap) return fail(ctx, GVL_ERR_ARG, "gvl_splice: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int idx = -1, cnt = 0;
  std::vector<int> text; text.reserve(n_ids);
  for (int i = 0; i < n_ids; ++i) {
    if (ids[i] == -200) { if (idx < 0) idx = i; ++cnt; }
    else { if (ids[i] < 0 || ids[i] >= ctx->cfg.vocab) return fail(ctx, GVL_ERR_ARG, "gvl_splice: token id out of range"); text.push_back((int)ids[i]); }
  }
  if (cnt != 1) return fail(ctx, GVL_ERR_ARG, "gvl_splice: exactly one IMAGE_TOKEN_INDEX (-200) expected");
  const int Hd = ctx->cfg.hidden, n_post = n_ids - 1 - idx;
  // ids travel by value in the kernel arguments: stream ordered, nothing shared between back-to-back splices
  RUN(GVL_PROF_OTHER, 0, gvl_launch_gather_rows_host_ids(ctx->l_embed, text.data(), idx, embeds, Hd, st));
  if (n_visual > 0) RUN(GVL_PROF_OTHER, 0, gvl_launch_copy_bytes(visual, embeds + (size_t)idx * Hd, (size_t)n_visual * Hd * 2, st));
  RUN(GVL_PROF_OTHER, 0, gvl_launch_gather_rows_host_ids(ctx->l_embed, text.data() + idx, n_post, embeds + (size_t)(idx + n_visual) * Hd, Hd, st));
  if (seq_len_out) *seq_len_out = n_ids - 1 + n_visual;
  return 0;
}

int gvl_seq_alloc(gvl_ctx* ctx, int max_tokens, int* seq_id) {
  REQUIRE_READY(ctx->has_llm, "gvl_seq_alloc");
  if (max_tokens <= 0 || !seq_id) return fail(ctx, GVL_ERR_ARG, "gvl_seq_alloc: bad arguments");
  if (max_tokens > ctx->cfg.max_seq) return fail(ctx, GVL_ERR_ARG, "gvl_seq_alloc: max_tokens exceeds cfg.max_seq (rope tables)");
  const int np = (max_tokens + 63) / 64;
  if ((int)ctx->free_pages.size() < np) return fail(ctx, GVL_ERR_OOM, "gvl_seq_alloc: KV pages exhausted");
  int id = -1;
  for (size_t i = 0; i < ctx->seqs.size(); ++i) if (!ctx->seqs[i].used) { id = (int)i; break; }
  if (id < 0) {
    if ((int)ctx->seqs.size() >= gvl_ctx::kMaxSeqs) return fail(ctx, GVL_ERR_OOM, "gvl_seq_alloc: too many live sequences");
    ctx->seqs.emplace_back(); id = (int)ctx->seqs.size() - 1;
  }
  Seq& s = ctx->seqs[id];
  s.used = true; s.max_tokens = max_tokens; s.n_pages = np; s.pos = 0; s.n_gen = 0; s.pages.clear();
  for (int i = 0; i < np; ++i) { s.pages.push_back(ctx->free_pages.back()); ctx->free_pages.pop_back(); ctx->page_ref[s.pages.back()] = 1; }
  // preallocated slot: no hipMalloc / hipFree / device-wide sync per clip.  Work that uses the slot is stream ordered;
  // a freed slot or page may be handed out again only for work enqueued later on the same stream (one stream per ctx
  // for the LLM path -- the reference is single-stream too).
  s.d_block_table = ctx->d_seq_tables + (size_t)id * ctx->seq_table_cap;
  s.d_pos = ctx->d_seq_pos + id;
  s.d_tok = ctx->d_seq_tok + id;
  s.d_out = ctx->d_seq_out + (size_t)id * ctx->outlist_cap; s.h_out = ctx->h_seq_out + (size_t)id * ctx->outlist_cap;
  s.d_ngen = ctx->d_seq_ngen + id;
  s.d_eos = ctx->d_eos_flags + id; s.h_eos = ctx->h_eos_flags + id;
  s.table_dirty = true;                               // written by the first prefill / decode on ITS stream (upload_table); d_pos likewise
  *seq_id = id;
  return 0;
}
int gvl_seq_free(gvl_ctx* ctx, int seq_id) {
  if (!ctx || seq_id < 0 || seq_id >= (int)ctx->seqs.size() || !ctx->seqs[seq_id].used) return fail(ctx, GVL_ERR_ARG, "gvl_seq_free: bad seq");
  Seq& s = ctx->seqs[seq_id];
  for (int p : s.pages) if (--ctx->page_ref[p] == 0) ctx->free_pages.push_back(p);     // a page shared with a fork lives on until its last holder is freed
  s = Seq();
  return 0;
}
int gvl_seq_fork(gvl_ctx* ctx, int src_seq, int n_tokens, int max_tokens, int* dst_seq) {
  REQUIRE_READY(ctx->has_llm, "gvl_seq_fork");
  if (!dst_seq || src_seq < 0 || src_seq >= (int)ctx->seqs.size() || !ctx->seqs[src_seq].used) return fail(ctx, GVL_ERR_ARG, "gvl_seq_fork: bad arguments");
  if (n_tokens <= 0 || (n_tokens & 63) || n_tokens > ctx->seqs[src_seq].pos) return fail(ctx, GVL_ERR_ARG, "gvl_seq_fork: n_tokens must be a positive multiple of 64 within the source's tokens");
  if (max_tokens <= n_tokens || max_tokens > ctx->cfg.max_seq) return fail(ctx, GVL_ERR_ARG, "gvl_seq_fork: max_tokens must exceed n_tokens and fit cfg.max_seq");
  const int shared = n_tokens >> 6, np = (max_tokens + 63) / 64;
  if ((int)ctx->free_pages.size() < np - shared) return fail(ctx, GVL_ERR_OOM, "gvl_seq_fork: KV pages exhausted");
  int id = -1;
  for (size_t i = 0; i < ctx->seqs.size(); ++i) if (!ctx->seqs[i].used) { id = (int)i; break; }
  if (id < 0) {
    if ((int)ctx->seqs.size() >= gvl_ctx::kMaxSeqs) return fail(ctx, GVL_ERR_OOM, "gvl_seq_fork: too many live sequences");
    ctx->seqs.emplace_back(); id = (int)ctx->seqs.size() - 1;
  }
  const std::vector<int> src_pages(ctx->seqs[src_seq].pages.begin(), ctx->seqs[src_seq].pages.begin() + shared);   // (emplace_back may have moved the source)
  Seq& s = ctx->seqs[id];
  s.used = true; s.max_tokens = max_tokens; s.n_pages = np; s.pos = n_tokens; s.n_gen = 0; s.pages = src_pages;
  for (int p : s.pages) ++ctx->page_ref[p];          // whole pages of the prefix: immutable from now on for both holders (appends go to later pages)
  for (int i = shared; i < np; ++i) { s.pages.push_back(ctx->free_pages.back()); ctx->free_pages.pop_back(); ctx->page_ref[s.pages.back()] = 1; }
  s.d_block_table = ctx->d_seq_tables + (size_t)id * ctx->seq_table_cap;
  s.d_pos = ctx->d_seq_pos + id;
  s.d_tok = ctx->d_seq_tok + id;
  s.d_out = ctx->d_seq_out + (size_t)id * ctx->outlist_cap; s.h_out = ctx->h_seq_out + (size_t)id * ctx->outlist_cap;
  s.d_ngen = ctx->d_seq_ngen + id;
  s.d_eos = ctx->d_eos_flags + id; s.h_eos = ctx->h_eos_flags + id;
  s.table_dirty = true;
  *dst_seq = id;
  return 0;
}

int gvl_prefill(gvl_ctx* ctx, int seq_id, const uint16_t* embeds, int S, float* last_logits, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_prefill");
  if (seq_id < 0 || seq_id >= (int)ctx->seqs.size() || !ctx->seqs[seq_id].used) return fail(ctx, GVL_ERR_ARG, "gvl_prefill: bad seq");
  Seq& sq = ctx->seqs[seq_id];
  if (!embeds || S <= 0 || S > sq.max_tokens || S > ctx->cfg.max_prefill) return fail(ctx, GVL_ERR_ARG, "gvl_prefill: bad length");
  if (sq.pos != 0) return fail(ctx, GVL_ERR_STATE, "gvl_prefill: sequence already holds tokens");
  hipStream_t st = (hipStream_t)stream;
  Seq* one[1] = {&sq}; const bf16_t* e1[1] = {embeds};
  int rc = llm_prefill(ctx, one, 1, e1, &S, st);
  if (rc) return rc;
  if (last_logits) HIPCHK(ctx, hipMemcpyAsync(last_logits, ctx->d_logits, (size_t)ctx->cfg.vocab * 4, hipMemcpyDeviceToDevice, st));
  return 0;
}

int gvl_seq_clone(gvl_ctx* ctx, int src_seq, int max_tokens, int* dst_seq, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_seq_clone");
  if (!dst_seq || src_seq < 0 || src_seq >= (int)ctx->seqs.size() || !ctx->seqs[src_seq].used) return fail(ctx, GVL_ERR_ARG, "gvl_seq_clone: bad arguments");
  const int pos = ctx->seqs[src_seq].pos;
  if (pos <= 0 || max_tokens <= pos || max_tokens > ctx->cfg.max_seq) return fail(ctx, GVL_ERR_ARG, "gvl_seq_clone: the source must hold tokens and max_tokens must exceed them (and fit cfg.max_seq)");
  const int shared = pos >> 6, np = (max_tokens + 63) / 64;
  if ((int)ctx->free_pages.size() < np - shared) return fail(ctx, GVL_ERR_OOM, "gvl_seq_clone: KV pages exhausted");
  int id = -1;
  for (size_t i = 0; i < ctx->seqs.size(); ++i) if (!ctx->seqs[i].used) { id = (int)i; break; }
  if (id < 0) {
    if ((int)ctx->seqs.size() >= gvl_ctx::kMaxSeqs) return fail(ctx, GVL_ERR_OOM, "gvl_seq_clone: too many live sequences");
    ctx->seqs.emplace_back(); id = (int)ctx->seqs.size() - 1;
  }
  const std::vector<int> src_pages = ctx->seqs[src_seq].pages;
  const int src_ngen = ctx->seqs[src_seq].n_gen;
  Seq& s = ctx->seqs[id];
  s.used = true; s.max_tokens = max_tokens; s.n_pages = np; s.pos = pos; s.n_gen = src_ngen;
  s.pages.assign(src_pages.begin(), src_pages.begin() + shared);
  for (int p : s.pages) ++ctx->page_ref[p];
  for (int i = shared; i < np; ++i) { s.pages.push_back(ctx->free_pages.back()); ctx->free_pages.pop_back(); ctx->page_ref[s.pages.back()] = 1; }
  s.d_block_table = ctx->d_seq_tables + (size_t)id * ctx->seq_table_cap;
  s.d_pos = ctx->d_seq_pos + id;
  s.d_tok = ctx->d_seq_tok + id;
  s.d_out = ctx->d_seq_out + (size_t)id * ctx->outlist_cap; s.h_out = ctx->h_seq_out + (size_t)id * ctx->outlist_cap;
  s.d_ngen = ctx->d_seq_ngen + id;
  s.d_eos = ctx->d_eos_flags + id; s.h_eos = ctx->h_eos_flags + id;
  hipStream_t st = (hipStream_t)stream;
  s.table_dirty = true;
  { const int rc = upload_table(ctx, s, st); if (rc) return rc; }   // by value, on the clone's stream: ordered behind the source's pending steps there
  if (pos & 63)                                       // the partial last page is private: copy the source's (all layers, K and V^T)
    RUN(GVL_PROF_OTHER, 0, gvl_launch_kv_page_copy(ctx->kpool, ctx->vpool, ctx->layer_stride, (size_t)ctx->cfg.kv_heads * 64 * ctx->l_D, ctx->cfg.layers,
                                                   src_pages[shared], s.pages[shared], st));
  RUN(GVL_PROF_OTHER, 0, gvl_launch_set_int(s.d_pos, pos, st));
  RUN(GVL_PROF_OTHER, 0, gvl_launch_set_int(s.d_ngen, 0, st));
  s.n_gen = 0;
  *dst_seq = id;
  return 0;
}

int gvl_prefill_extend(gvl_ctx* ctx, int seq_id, const uint16_t* embeds, int n_new, float* last_logits, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_prefill_extend");
  if (seq_id < 0 || seq_id >= (int)ctx->seqs.size() || !ctx->seqs[seq_id].used) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_extend: bad seq");
  Seq& sq = ctx->seqs[seq_id];
  if (sq.pos <= 0 || (sq.pos & 63)) return fail(ctx, GVL_ERR_STATE, "gvl_prefill_extend: the sequence must hold a prefix of whole pages (gvl_seq_fork)");
  if (!embeds || n_new <= 0 || sq.pos + n_new > sq.max_tokens || n_new > ctx->cfg.max_prefill) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_extend: bad length");
  hipStream_t st = (hipStream_t)stream;
  Seq* one[1] = {&sq}; const bf16_t* e1[1] = {embeds};
  int rc = llm_prefill(ctx, one, 1, e1, &n_new, st, nullptr, sq.pos);
  if (rc) return rc;
  if (last_logits) HIPCHK(ctx, hipMemcpyAsync(last_logits, ctx->d_logits, (size_t)ctx->cfg.vocab * 4, hipMemcpyDeviceToDevice, st));
  return 0;
}

int gvl_forward_loss(gvl_ctx* ctx, int seq_id, const uint16_t* embeds, int S, const int64_t* labels, double* nll_sum, int* n_valid, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_forward_loss");
  if (seq_id < 0 || seq_id >= (int)ctx->seqs.size() || !ctx->seqs[seq_id].used) return fail(ctx, GVL_ERR_ARG, "gvl_forward_loss: bad seq");
  Seq& sq = ctx->seqs[seq_id];
  if (!embeds || !labels || !nll_sum || !n_valid || S <= 0 || S > sq.max_tokens || S > ctx->cfg.max_prefill) return fail(ctx, GVL_ERR_ARG, "gvl_forward_loss: bad arguments / length");
  if (sq.pos != 0) return fail(ctx, GVL_ERR_STATE, "gvl_forward_loss: sequence already holds tokens");
  // shift (logits[:-1] vs labels[1:]) and drop ignore_index (-100) on the host: integer work
  std::vector<int> rows, tgt;
  for (int t = 0; t + 1 < S; ++t) {
    const int64_t y = labels[t + 1];
    if (y == -100) continue;
    if (y < 0 || y >= ctx->cfg.vocab) return fail(ctx, GVL_ERR_ARG, "gvl_forward_loss: label out of range");   // torch CrossEntropyLoss raises too
    rows.push_back(t); tgt.push_back((int)y);
  }
  std::vector<float> nll(rows.size());
  LossReq lr{(int)rows.size(), rows.data(), tgt.data(), nll.data()};
  hipStream_t st = (hipStream_t)stream;
  Seq* one[1] = {&sq}; const bf16_t* e1[1] = {embeds};
  const int rc = llm_prefill(ctx, one, 1, e1, &S, st, &lr);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(st));                     // host vectors above are the copy endpoints
  double acc = 0;
  for (float v : nll) acc += (double)v;
  *nll_sum = acc; *n_valid = (int)rows.size();
  return 0;
}

int gvl_prefill_varlen(gvl_ctx* ctx, const int* seq_ids, int n_seqs, const uint16_t* const* embeds, const int* seq_lens, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_prefill_varlen");
  if (!seq_ids || !embeds || !seq_lens || n_seqs <= 0 || n_seqs > gvl_ctx::kMaxSeqs) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_varlen: bad arguments");
  for (int i = 0; i < n_seqs; ++i) {
    const int id = seq_ids[i];
    if (id < 0 || id >= (int)ctx->seqs.size() || !ctx->seqs[id].used || !embeds[i]) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_varlen: bad seq / embeds");
    if (seq_lens[i] <= 0 || seq_lens[i] > ctx->seqs[id].max_tokens) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_varlen: bad length");
    if (seq_lens[i] > ctx->cfg.max_prefill) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_varlen: seq_len exceeds cfg.max_prefill");
    if (ctx->seqs[id].pos != 0) return fail(ctx, GVL_ERR_STATE, "gvl_prefill_varlen: sequence already holds tokens");
    for (int j = 0; j < i; ++j) if (seq_ids[j] == id) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_varlen: duplicate seq");
  }
  hipStream_t st = (hipStream_t)stream;
  // groups of up to 8 sequences in call order -- as many as the prefill workspace (cfg.max_prefill rows in total) allows
  int i = 0;
  while (i < n_seqs) {
    const int left = n_seqs - i;
    int B = left >= ctx->dbg.prefill_group ? ctx->dbg.prefill_group : left;
    for (;;) {
      long rows = 0; for (int b = 0; b < B; ++b) rows += seq_lens[i + b];
      if (B == 1 || rows <= ctx->cfg.max_prefill) break;
      --B;
    }
    Seq* sqs[GVL_MAX_PREFILL_BATCH]; const bf16_t* es[GVL_MAX_PREFILL_BATCH];
    for (int b = 0; b < B; ++b) { sqs[b] = &ctx->seqs[seq_ids[i + b]]; es[b] = embeds[i + b]; }
    const int rc = llm_prefill(ctx, sqs, B, es, seq_lens + i, st);
    if (rc) return rc;
    i += B;
  }
  return 0;
}

int gvl_prefill_batch(gvl_ctx* ctx, const int* seq_ids, int n_seqs, const uint16_t* const* embeds, int S, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  if (n_seqs <= 0 || n_seqs > gvl_ctx::kMaxSeqs) return fail(ctx, GVL_ERR_ARG, "gvl_prefill_batch: bad arguments");
  int lens[gvl_ctx::kMaxSeqs];
  for (int i = 0; i < n_seqs; ++i) lens[i] = S;
  return gvl_prefill_varlen(ctx, seq_ids, n_seqs, embeds, lens, stream);
}

int gvl_decode_greedy(gvl_ctx* ctx, int seq_id, int max_new, int eos_id, int32_t* out_ids, int* n_out, void* stream) {
  return gvl_decode_greedy_batch(ctx, &seq_id, 1, max_new, eos_id, out_ids, n_out, stream);
}

int gvl_decode_greedy_batch(gvl_ctx* ctx, const int* seq_ids, int n_seqs, int max_new, int eos_id, int32_t* out_ids, int* n_out, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_decode_greedy_batch");
  if (!seq_ids || n_seqs <= 0 || n_seqs > gvl_ctx::kMaxSeqs || !out_ids || !n_out || max_new <= 0 || max_new > ctx->outlist_cap)
    return fail(ctx, GVL_ERR_ARG, "gvl_decode_greedy_batch: bad arguments");
  for (int i = 0; i < n_seqs; ++i) {
    const int id = seq_ids[i];
    if (id < 0 || id >= (int)ctx->seqs.size() || !ctx->seqs[id].used) return fail(ctx, GVL_ERR_ARG, "gvl_decode_greedy_batch: bad seq");
    if (ctx->seqs[id].n_gen < 1) return fail(ctx, GVL_ERR_STATE, "gvl_decode_greedy: call gvl_prefill first");   // members of a group must also be at the SAME step (checked per group)
    for (int j = 0; j < i; ++j) if (seq_ids[j] == id) return fail(ctx, GVL_ERR_ARG, "gvl_decode_greedy_batch: duplicate seq");
  }
  hipStream_t st = (hipStream_t)stream;
  // groups of up to 16 (VALU fallback: 4, 2, 1): a group streams the weights once per step for all of its members
  int i = 0;
  while (i < n_seqs) {
    const int left = n_seqs - i, B = decode_group_size(ctx, left);
    Seq* sqs[GVL_MAX_DECODE_BATCH]; int32_t* outs[GVL_MAX_DECODE_BATCH]; int* nouts[GVL_MAX_DECODE_BATCH];
    for (int b = 0; b < B; ++b) { sqs[b] = &ctx->seqs[seq_ids[i + b]]; outs[b] = out_ids + (size_t)(i + b) * max_new; nouts[b] = n_out + i + b; }
    const int rc = decode_group(ctx, sqs, B, max_new, eos_id, outs, nouts, st);
    if (rc) return rc;
    i += B;
  }
  return 0;
}

// ---- building blocks of a continuous-batching scheduler: sequences at DIFFERENT generation steps advance
// together, the host decides between chunks who joins (gvl_prefill*) and who leaves (gvl_seq_free).
int gvl_decode_steps(gvl_ctx* ctx, const int* seq_ids, int n_seqs, int n_steps, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_decode_steps");
  if (!seq_ids || n_seqs <= 0 || n_seqs > gvl_ctx::kMaxSeqs || n_steps <= 0) return fail(ctx, GVL_ERR_ARG, "gvl_decode_steps: bad arguments");
  for (int i = 0; i < n_seqs; ++i) {
    const int id = seq_ids[i];
    if (id < 0 || id >= (int)ctx->seqs.size() || !ctx->seqs[id].used) return fail(ctx, GVL_ERR_ARG, "gvl_decode_steps: bad seq");
    const Seq& sq = ctx->seqs[id];
    if (sq.n_gen < 1) return fail(ctx, GVL_ERR_STATE, "gvl_decode_steps: call gvl_prefill first");
    if (sq.pos + n_steps > sq.max_tokens || sq.n_gen + n_steps > ctx->outlist_cap) return fail(ctx, GVL_ERR_ARG, "gvl_decode_steps: sequence would exceed its capacity");
    for (int j = 0; j < i; ++j) if (seq_ids[j] == id) return fail(ctx, GVL_ERR_ARG, "gvl_decode_steps: duplicate seq");
  }
  hipStream_t st = (hipStream_t)stream;
  for (int s = 0; s < n_steps; ++s) {
    int i = 0;
    while (i < n_seqs) {                   // groups of up to 16 (VALU fallback: 4, 2, 1): one weight stream per step per group
      const int left = n_seqs - i, B = decode_group_size(ctx, left);
      Seq* sqs[GVL_MAX_DECODE_BATCH];
      for (int b = 0; b < B; ++b) sqs[b] = &ctx->seqs[seq_ids[i + b]];
      const int rc = decode_step(ctx, sqs, B, st);
      if (rc) return rc;
      i += B;
    }
  }
  return 0;
}

int gvl_seq_read(gvl_ctx* ctx, int seq_id, int first, int32_t* out_ids, int cap, int* n_gen, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_seq_read");
  if (seq_id < 0 || seq_id >= (int)ctx->seqs.size() || !ctx->seqs[seq_id].used || !n_gen || first < 0 || cap < 0 || (cap > 0 && !out_ids))
    return fail(ctx, GVL_ERR_ARG, "gvl_seq_read: bad arguments");
  const Seq& sq = ctx->seqs[seq_id];
  *n_gen = sq.n_gen;
  int n = sq.n_gen - first; if (n > cap) n = cap;
  if (n > 0) {
    HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream));
    memcpy(out_ids, sq.h_out + first, (size_t)n * 4);
  }
  return 0;
}

int gvl_decode_step_logits(gvl_ctx* ctx, int seq_id, int tok, float* logits, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_decode_step_logits");
  if (seq_id < 0 || seq_id >= (int)ctx->seqs.size() || !ctx->seqs[seq_id].used) return fail(ctx, GVL_ERR_ARG, "gvl_decode_step_logits: bad seq");
  Seq& sq = ctx->seqs[seq_id];
  if (tok < 0 || tok >= ctx->cfg.vocab || sq.pos >= sq.max_tokens || sq.pos == 0) return fail(ctx, GVL_ERR_ARG, "gvl_decode_step_logits: bad token / sequence full / not prefilled");
  hipStream_t st = (hipStream_t)stream;
  RUN(GVL_PROF_OTHER, 0, gvl_launch_set_int(sq.d_tok, tok, st));
  Seq* one[1] = {&sq};
  int rc = decode_step(ctx, one, 1, st);
  if (rc) return rc;
  if (logits) HIPCHK(ctx, hipMemcpyAsync(logits, ctx->d_logits, (size_t)ctx->cfg.vocab * 4, hipMemcpyDeviceToDevice, st));
  return 0;
}

int gvl_decode_step_logits_batch(gvl_ctx* ctx, const int* seq_ids, int n_seqs, const int32_t* toks, float* logits, void* stream) {
  REQUIRE_READY(ctx->has_llm, "gvl_decode_step_logits_batch");
  if (!seq_ids || !toks || n_seqs < 1 || n_seqs > decode_group_size(ctx, GVL_MAX_DECODE_BATCH) || (!ctx->decode_mfma && n_seqs == 3))
    return fail(ctx, GVL_ERR_ARG, "gvl_decode_step_logits_batch: 1 .. 16 sequences (VALU fallback geometries: 1, 2 or 4)");
  Seq* sqs[GVL_MAX_DECODE_BATCH];
  for (int i = 0; i < n_seqs; ++i) {
    const int id = seq_ids[i];
    if (id < 0 || id >= (int)ctx->seqs.size() || !ctx->seqs[id].used) return fail(ctx, GVL_ERR_ARG, "gvl_decode_step_logits_batch: bad seq");
    Seq& sq = ctx->seqs[id];
    if (toks[i] < 0 || toks[i] >= ctx->cfg.vocab || sq.pos >= sq.max_tokens || sq.pos == 0) return fail(ctx, GVL_ERR_ARG, "gvl_decode_step_logits_batch: bad token / sequence full / not prefilled");
    for (int j = 0; j < i; ++j) if (seq_ids[j] == id) return fail(ctx, GVL_ERR_ARG, "gvl_decode_step_logits_batch: duplicate seq");
    sqs[i] = &sq;
  }
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_seqs; ++i) RUN(GVL_PROF_OTHER, 0, gvl_launch_set_int(sqs[i]->d_tok, toks[i], st));
  const int rc = decode_step(ctx, sqs, n_seqs, st);       // ONE weight stream for all of them; row i depends on sequence i only (batch-invariant kernels)
  if (rc) return rc;
  if (logits) HIPCHK(ctx, hipMemcpyAsync(logits, ctx->d_logits, (size_t)n_seqs * ctx->cfg.vocab * 4, hipMemcpyDeviceToDevice, st));
  return 0;
}

int gvl_debug_set(gvl_ctx* ctx, const char* key, int value) {
  if (!ctx || !key) return GVL_ERR_ARG;
  const std::string k = key;
  if (k == "decode_attn_cpb") { if (value < 0 || value > 16) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: decode_attn_cpb must be 0 (default) .. 16"); ctx->dbg.decode_attn_cpb = value; }
  else if (k == "decode_attn_hpb") { if (value < 0) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: decode_attn_hpb must be >= 0"); ctx->dbg.decode_attn_hpb = value; }
  else if (k == "decode_graph") ctx->dbg.decode_graph = value != 0;
  else if (k == "attn_ring") { if (value != 0 && value != 2 && value != 3) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: attn_ring must be 0, 2 or 3"); ctx->dbg.attn_ring = value; }
  else if (k == "prefill_group") { if (value < 1 || value > GVL_MAX_PREFILL_BATCH) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: prefill_group must be 1 .. 8"); ctx->dbg.prefill_group = value; }
  else if (k == "attn_pipe") { if (value < 0 || value > 2) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: attn_pipe must be 0, 1 or 2"); ctx->dbg.attn_pipe = value; }
  else if (k == "attn_pipe_rows") { if (value != 128 && value != 256) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: attn_pipe_rows must be 128 (default) or 256"); ctx->dbg.attn_pipe_rows = value; }
  else if (k == "patch_fused") ctx->dbg.patch_fused = value != 0;
  else if (k == "varlen_attn") ctx->dbg.varlen_attn = value < 0 || value > 2 ? 1 : value;
  else if (k == "norm_fused") ctx->dbg.norm_fused = value != 0;
  else if (k == "last_layer_tail") ctx->dbg.last_layer_tail = value != 0;
  else if (k == "gemm_band") { if (value < 0 || value > 64) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: gemm_band must be 0 (automatic) .. 64"); gvl_gemm_set_band(value); }
  else if (k == "gemm_a4") { if (value < 0 || value > 3) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: gemm_a4 must be 0 .. 3"); gvl_gemm_set_a4(value); }
  else if (k == "gemm_narrow") gvl_gemm_set_narrow(value < 0 || value > 2 ? 1 : value);
  else if (k == "vision_in_place") { if (value < 0 || value > 2) return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: vision_in_place must be 0, 1 or 2"); ctx->dbg.vision_in_place = value; }
  else return fail(ctx, GVL_ERR_ARG, "gvl_debug_set: unknown key " + k);
  return 0;
}

int gvl_set_sampling(gvl_ctx* ctx, int do_sample, float temperature, int top_k, float top_p, uint64_t seed) {
  if (!ctx) return GVL_ERR_ARG;
  if (!do_sample) { ctx->sample.on = false; return 0; }
  if (!(temperature > 0.f) || top_k < 0 || !(top_p >= 0.f) || top_p > 1.f)
    return fail(ctx, GVL_ERR_ARG, "gvl_set_sampling: temperature must be > 0, top_k >= 0, 0 <= top_p <= 1");
  ctx->sample.on = true; ctx->sample.inv_temp = 1.0f / temperature; ctx->sample.top_k = top_k; ctx->sample.top_p = top_p;
  // stream numbering restarts with the call (same seed + same prefill order = same draws) -- unless sequences are LIVE: a scheduler that
  // changes the sampling parameters mid-flight must not hand the stream ids of running sequences to newcomers
  bool any_live = false;
  for (const Seq& q : ctx->seqs) any_live = any_live || q.used;
  if (!any_live || seed != ctx->sample.seed) ctx->sample.next_stream = 0;
  ctx->sample.seed = seed;
  return 0;
}

int gvl_op_sample(gvl_ctx* ctx, const float* logits, int n, int batch, float temperature, int top_k, float top_p, uint64_t seed,
                  const uint32_t* streams, const int32_t* steps_dev, int32_t* tokens_dev, void* stream) {
  if (!ctx || !logits || !streams || !steps_dev || !tokens_dev || n < 1 || batch < 1 || batch > GVL_MAX_DECODE_BATCH || !(temperature > 0.f) || top_k < 0 || !(top_p >= 0.f) || top_p > 1.f)
    return fail(ctx, GVL_ERR_ARG, "gvl_op_sample: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ArgmaxArgs am; memset(&am, 0, sizeof(am)); am.logits = logits; am.n = n; am.batch = batch;
  am.inv_temp = 1.0f / temperature; am.top_k = top_k; am.top_p = top_p; am.seed_lo = (unsigned)seed; am.seed_hi = (unsigned)(seed >> 32);
  am.step_override = steps_dev;
  for (int b = 0; b < batch; ++b) { am.tok_ptrs[b] = tokens_dev + b; am.stream[b] = streams[b]; }
  RUN(GVL_PROF_OTHER, 0, gvl_launch_sample(am, st));
  return 0;
}

int gvl_preprocess_frames(gvl_ctx* ctx, const uint8_t* frames, int n, int height, int width, int layout, int size,
                          const float* mean, const float* stdv, float* out, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  if (!frames || !out || !mean || !stdv || n <= 0 || height <= 0 || width <= 0 || size <= 0 || (layout != 0 && layout != 1))
    return fail(ctx, GVL_ERR_ARG, "gvl_preprocess_frames: bad arguments");
  for (int c = 0; c < 3; ++c) if (!(stdv[c] != 0.f)) return fail(ctx, GVL_ERR_ARG, "gvl_preprocess_frames: zero std");
  const int rc = gvl_launch_preprocess(frames, n, height, width, layout, size, mean, stdv, out, &ctx->pre_scratch, &ctx->pre_scratch_bytes, (hipStream_t)stream);
  if (rc == -2) return fail(ctx, GVL_ERR_OOM, "gvl_preprocess_frames: hipMalloc(scratch) failed");
  if (rc) return fail(ctx, rc == -1 ? GVL_ERR_ARG : GVL_ERR_HIP, "gvl_preprocess_frames: launch failed");
  return 0;
}

int gvl_prof_enable(gvl_ctx* ctx, int on) {
  if (!ctx) return GVL_ERR_ARG;
  hipDeviceSynchronize();
  for (auto& r : ctx->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  ctx->recs.clear();
  for (int i = 0; i < GVL_PROF_NCAT; ++i) { ctx->prof_ms[i] = 0; ctx->prof_work[i] = 0; ctx->prof_n[i] = 0; }
  ctx->prof = on != 0;
  return 0;
}
int gvl_prof_read(gvl_ctx* ctx, int cat, double* total_ms, int64_t* launches, double* work) {
  if (!ctx || cat < 0 || cat >= GVL_PROF_NCAT) return GVL_ERR_ARG;
  hipDeviceSynchronize();
  for (auto& r : ctx->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { ctx->prof_ms[r.cat] += ms; ctx->prof_work[r.cat] += r.work; ctx->prof_n[r.cat] += 1; }
    hipEventDestroy(r.e0); hipEventDestroy(r.e1);
  }
  ctx->recs.clear();
  if (total_ms) *total_ms = ctx->prof_ms[cat];
  if (launches) *launches = ctx->prof_n[cat];
  if (work) *work = ctx->prof_work[cat];
  return 0;
}

// ---- operator-level entry points -------------------------------------------------------------------
int gvl_op_gemm(gvl_ctx* ctx, const uint16_t* A, const uint16_t* W, void* C, int M, int N, int K, const float* bias, const float* gamma,
                const void* resid, int act, int out_f32, int tile_cfg, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g = gemm(A, K, W, C, act == GVL_ACT_SILU_MUL ? N / 2 : N, M, N, K);
  g.bias = bias; g.gamma = gamma; g.resid = resid; g.ldr = N; g.act = act; g.out_f32 = out_f32; g.round_pre_resid = 1; g.tile_cfg = tile_cfg;
  if (const char* e = gvl_lab_env("GVL_LAB_LD")) { int la = 0, lw = 0; if (sscanf(e, "%d,%d", &la, &lw) == 2) { g.lda = la; g.ldw = lw; } }   // LAB: operand row pitches (tools/gemm_lab.py)
  RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
  return 0;
}
// The fused-RMSNorm epilogues at operator level (bf16 output): rowscale [M] f32 or null multiplies the accumulator rows before bias / activation; rowsq
// [M][rowsq_ld] f32 or null receives the sums of squares of the rounded outputs per aligned 64-column block (N % 64 == 0).
int gvl_op_gemm_rows(gvl_ctx* ctx, const uint16_t* A, const uint16_t* W, uint16_t* C, int M, int N, int K, const float* bias, const float* gamma,
                     const uint16_t* resid, int act, const float* rowscale, float* rowsq, int rowsq_ld, int tile_cfg, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g = gemm(A, K, W, C, act == GVL_ACT_SILU_MUL ? N / 2 : N, M, N, K);
  g.bias = bias; g.gamma = gamma; g.resid = resid; g.ldr = N; g.act = act; g.round_pre_resid = 1; g.tile_cfg = tile_cfg;
  g.rowscale = rowscale; g.rowsq = rowsq; g.rowsq_ld = rowsq_ld;
  RUN(GVL_PROF_GEMM, gvl_gemm_flops(g), gvl_launch_gemm(g, st));
  return 0;
}
// W' = bf16(W diag(gamma)) and rs = rsqrt(sum of the blocks [b0, b0 + nblk) / cols + eps): the two small kernels around those epilogues
int gvl_op_fold_gamma(gvl_ctx* ctx, const uint16_t* W, const uint16_t* gamma, uint16_t* Wo, int64_t rows, int cols, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RUN(GVL_PROF_OTHER, 0, gvl_launch_fold_gamma(W, gamma, Wo, (long)rows, cols, st));
  return 0;
}
int gvl_op_rowsq_finish(gvl_ctx* ctx, const float* rowsq, int ld, int b0, int nblk, float* rs, int rows, int cols, float eps, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RUN(GVL_PROF_OTHER, 0, gvl_launch_rowsq_finish(rowsq, ld, b0, nblk, rs, rows, cols, eps, st));
  return 0;
}
int gvl_op_attention(gvl_ctx* ctx, const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out, int B, int S, int H, int KV, int Dr,
                     float scale, int causal, void* stream) {
  // q/k/v here are column blocks of ONE fused row-major qkv tensor [B*S][(H+2KV)*Dr]: q points at column 0;
  // k and v must equal q + H*Dr and q + (H+KV)*Dr (checked) -- the layout every tower produces.
  if (!ctx) return GVL_ERR_ARG;
  if (k != q + (size_t)H * Dr || v != q + (size_t)(H + KV) * Dr) return fail(ctx, GVL_ERR_ARG, "gvl_op_attention: q,k,v must be the column blocks of one fused qkv tensor");
  const int D = pad_head(Dr);
  if (D < 0 || (Dr & 7)) return fail(ctx, GVL_ERR_ARG, "gvl_op_attention: head dim unsupported");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = (S + 63) / 64;
  ArenaScope arena_scope(ctx->arena_off);
  AALLOC(Q, bf16_t, (size_t)B * H * S * D); AALLOC(Kt, bf16_t, (size_t)B * tiles * KV * 64 * D); AALLOC(Vt, bf16_t, (size_t)B * tiles * KV * 64 * D);
  // non-causal (vision) attention reads v -- and, when no head-dim padding is needed, q and k -- in place, exactly as the towers do
  const int ld = (H + 2 * KV) * Dr;
  const bool v_rows = ctx->dbg.vision_in_place && !causal && D != 128, qk_rows = ctx->dbg.vision_in_place == 1 && v_rows && D == Dr && D == 64;
  if (!qk_rows) { QkvPostArgs p; memset(&p, 0, sizeof(p)); p.qkv = q; p.ld = ld; p.Q = Q; p.Kt = Kt; p.Vt = v_rows ? nullptr : Vt; p.B = B; p.S = S; p.H = H; p.KV = KV; p.Dr = Dr; p.D = D; p.mode = 0;
    RUN(GVL_PROF_OTHER, 0, gvl_launch_qkv_post(p, st)); }
  { AttnArgs a; memset(&a, 0, sizeof(a)); a.Q = Q; a.Kt = Kt; a.Vt = Vt; a.O = out; a.B = B; a.H = H; a.KV = KV; a.S = S; a.D = D; a.Dout = Dr; a.scale = scale; a.causal = causal; a.ring = ctx->dbg.attn_ring;
    if (v_rows) { a.Vrows = v; a.v_ld = ld; }
    if (qk_rows) { a.Qrows = q; a.Krows = k; a.q_ld = a.k_ld = ld; }
    RUN(GVL_PROF_ATTN, gvl_attn_flops(a), gvl_launch_attention(a, st)); }
  return 0;
}
int gvl_op_layernorm(gvl_ctx* ctx, const float* x, const float* w, const float* b, uint16_t* y, int rows, int cols, float eps, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RUN(GVL_PROF_OTHER, 0, gvl_launch_layernorm_f32(x, w, b, y, rows, cols, eps, st));
  return 0;
}
int gvl_op_rmsnorm(gvl_ctx* ctx, const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int cols, float eps, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  RUN(GVL_PROF_OTHER, 0, gvl_launch_rmsnorm_bf16(x, w, y, rows, cols, eps, st));
  return 0;
}
int gvl_op_dgemm(gvl_ctx* ctx, const uint16_t* W, const uint16_t* x, const float* bias, float* y, int N, int K, int batch, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (N <= 0 || K <= 0 || K % 256 || batch < 1 || batch > GVL_MAX_DECODE_BATCH) return fail(ctx, GVL_ERR_ARG, "gvl_op_dgemm: K % 256 == 0 and 1 <= batch <= 16");
  // operator-level entry (tests / microbenchmarks): the operands arrive row-major and are re-tiled here; the model keeps tiled copies
  void *wt = nullptr, *xt = nullptr;
  HIPCHK(ctx, hipMalloc(&wt, (size_t)((N + 15) / 16) * 16 * K * 2));
  if (hipMalloc(&xt, (size_t)16 * K * 2) != hipSuccess) { hipFree(wt); return fail(ctx, GVL_ERR_OOM, "gvl_op_dgemm: hipMalloc"); }
  int rc = gvl_retile_decode_weight(W, (bf16_t*)wt, N, K, 0, 0, st);
  if (!rc) rc = gvl_launch_rows_to_tiled(x, (bf16_t*)xt, batch, K, K, st);
  GemvArgs g; memset(&g, 0, sizeof(g)); g.W = (const bf16_t*)wt; g.N = N; g.K = K; g.x = (const bf16_t*)xt; g.bias = bias; g.out_f32 = y; g.batch = batch; g.x_stride = K; g.out_stride = N;
  if (!rc) { ProfScope _ps(ctx, GVL_PROF_GEMV, 2.0 * N * K, st); rc = gvl_launch_dgemm(g, st); }
  hipStreamSynchronize(st);
  hipFree(wt); hipFree(xt);
  if (rc) return fail(ctx, rc == -1 ? GVL_ERR_ARG : GVL_ERR_HIP, "gvl_op_dgemm: launch failed");
  return 0;
}
// micro-benchmark of the decode projection kernels on synthetic operands (tools/decode_bench.py): mode 0 = skinny MFMA GEMM (variant:
// see gvl_launch_dgemm), 1 = round-1 VALU GEMV (batch 1, 2, 4), 2 / 3 = the same two with the fused RMSNorm prologue (batch <= 4).  `rounds` distinct weight matrices are cycled so that the
// Infinity Cache cannot hold the stream; returns the average microseconds per launch.
int gvl_op_decode_bench(gvl_ctx* ctx, int N, int K, int batch, int mode, int variant, int rounds, int iters, double* us_per_launch, void* stream) {
  if (!ctx || !us_per_launch || N <= 0 || K <= 0 || K % 256 || batch < 1 || batch > GVL_MAX_DECODE_BATCH || rounds < 1 || iters < 1) return fail(ctx, GVL_ERR_ARG, "gvl_op_decode_bench: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t wbytes = (size_t)((N + 15) / 16) * 16 * K * 2;
  char *w = nullptr, *x = nullptr, *y = nullptr;
  HIPCHK(ctx, hipMalloc((void**)&w, wbytes * rounds));
  HIPCHK(ctx, hipMalloc((void**)&x, (size_t)16 * K * 2));
  HIPCHK(ctx, hipMalloc((void**)&y, (size_t)16 * N * 4));
  hipMemsetAsync(w, 0x11, wbytes * rounds, st); hipMemsetAsync(x, 0x22, (size_t)16 * K * 2, st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int rc = 0;
  for (int pass = 0; pass < 2 && !rc; ++pass) {          // pass 0 = warm-up
    if (pass == 1) hipEventRecord(e0, st);
    for (int it = 0; it < (pass ? iters : 3) && !rc; ++it) {
      GemvArgs g; memset(&g, 0, sizeof(g)); g.W = (const bf16_t*)(w + wbytes * (it % rounds)); g.N = N; g.K = K; g.x = (const bf16_t*)x; g.out_f32 = (float*)y;
      g.batch = batch; g.x_stride = K; g.out_stride = N; g.variant = variant;
      if (mode == 2 || mode == 3) { g.norm_w = (const bf16_t*)x; g.eps = 1e-5f; }   // fused RMSNorm prologue: 2 = skinny GEMM (LDS), 3 = VALU GEMV
      rc = (mode == 1 || mode == 3) ? gvl_launch_gemv(g, st) : gvl_launch_dgemm(g, st);
    }
    if (pass == 1) hipEventRecord(e1, st);
  }
  hipStreamSynchronize(st);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(w); hipFree(x); hipFree(y);
  if (rc) return fail(ctx, rc == -1 ? GVL_ERR_ARG : GVL_ERR_HIP, "gvl_op_decode_bench: launch failed (unsupported variant / geometry)");
  *us_per_launch = 1e3 * ms / iters;
  return 0;
}
int gvl_op_gemv(gvl_ctx* ctx, const uint16_t* W, const uint16_t* x, const float* bias, float* y, int N, int K, void* stream) {
  if (!ctx) return GVL_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  GemvArgs g; memset(&g, 0, sizeof(g)); g.W = W; g.N = N; g.K = K; g.x = x; g.bias = bias; g.out_f32 = y;
  RUN(GVL_PROF_GEMV, 2.0 * N * K, gvl_launch_gemv(g, st));
  return 0;
}

}  // extern "C"
