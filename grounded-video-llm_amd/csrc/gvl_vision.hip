// gvl_vision.hip -- the launch sequences of the two vision towers and of the glue / projectors (host code; the kernels live in gvl_gemm*.hip, gvl_attn.hip,
// gvl_elem.hip, gvl_patch.hip).  Restates (file:line in the reference):
//   CLIP tower      models/modeling_clip.py:182-191,355-393,626-651,851   (23 of 24 layers, hidden_states[-2])
//   InternVideo2    models/internvideo2.py:680-684,721-725,970-1040      (39 of 40 blocks)
//   glue/projectors models/llava_next_video.py:454-489,507-564
#include "gvl_model.h"

namespace gvlm {

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

}  // namespace gvlm
