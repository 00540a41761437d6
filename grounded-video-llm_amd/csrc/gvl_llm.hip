// gvl_llm.hip -- the decoder's launch sequences: ragged / batched prefill (+ the training-forward loss tail), the decode step and its hipGraph replay, the greedy decode
// loop over a group of sequences (host code; the kernels live in gvl_gemm*.hip, gvl_attn.hip, gvl_decode.hip, gvl_elem.hip).  Restates (file:line in the reference):
//   LLM             models/modeling_phi3.py:1034-1095,1249-1383,1512-1526 / models/modeling_llama.py:699-760
//   generate()      models/llava_next_video.py:655-661 (greedy; transformers GenerationMixin [ext])
#include "gvl_model.h"

namespace gvlm {

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
int llm_prefill(gvl_ctx* ctx, Seq* const* sqs, int nb, const bf16_t* const* embeds, const int* lens, hipStream_t st, const LossReq* loss, int pos0) {
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


}  // namespace gvlm
