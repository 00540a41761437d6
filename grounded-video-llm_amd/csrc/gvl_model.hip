// gvl_model.hip -- host side of libgvl.so: context, packed weights, workspace arenas, paged KV pool and the C ABI of include/gvl.h.
// The launch sequences behind the ABI live in gvl_vision.hip (CLIP, InternVideo2, glue / projectors) and gvl_llm.hip (prefill, decode step, decode loop); gvl_model.h is what
// the three share.  Restates (file:line in the reference):
//   splice          models/llava_next_video.py:568-596
//   generate()      models/llava_next_video.py:655-661 (greedy; transformers GenerationMixin [ext])
#include "gvl_model.h"

std::string& gvl_create_error() { static std::string e; return e; }

using namespace gvlm;

namespace {

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
