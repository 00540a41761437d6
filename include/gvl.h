/*
 * gvl.h -- C ABI of libgvl.so, the MI355X-native Grounded-VideoLLM inference hot path.
 *
 * The reference (WHB139426/Grounded-Video-LLM) is pure Python/PyTorch and has no FFI seam; the
 * boundary this library sits behind is the set of Python calls made inside
 * LLAVA_NEXT_VIDEO.generate() (models/llava_next_video.py:616-666).  Each entry point below names
 * the reference call it replaces (paths relative to the reference root).  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; no torch types.  All tensor arguments are DEVICE pointers owned by the
 *     caller and valid for the duration of the call, unless the parameter says "host".
 *   - every call enqueues its work on the caller's hipStream_t (passed as void*) and returns
 *     without synchronising, except where noted.
 *   - return value: 0 = ok, < 0 = error (gvl_status); text via gvl_last_error().  Nothing throws.
 *   - bf16 tensors are raw uint16 bit patterns; "f32" is IEEE float.
 *   - one gvl_ctx per GPU / rank; a ctx is not thread-safe (the reference is single-threaded too).
 */
#ifndef GVL_H
#define GVL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gvl_ctx gvl_ctx;

typedef enum {
  GVL_OK = 0,
  GVL_ERR_ARG = -1,      /* bad argument / shape */
  GVL_ERR_STATE = -2,    /* call order (weights missing, sequence not allocated ...) */
  GVL_ERR_HIP = -3,      /* HIP runtime error */
  GVL_ERR_OOM = -4,      /* device memory or KV pages exhausted */
  GVL_ERR_NOGPU = -5     /* no gfx950 device */
} gvl_status;

enum { GVL_F32 = 0, GVL_BF16 = 1, GVL_I32 = 2, GVL_I64 = 3 };
enum { GVL_LLM_PHI3 = 0, GVL_LLM_LLAMA = 1 };

/* Geometry of the three towers.  Mirrors the hard-coded / config.json values of the reference:
 * CLIP  models/llava_next_video.py:56-71;  InternVideo2  models/internvideo2.py:1089-1114;
 * Phi-3 / Llama  models/modeling_phi3.py:132-156, models/modeling_llama.py (HF config [ext]). */
typedef struct {
  int32_t llm_kind;            /* GVL_LLM_PHI3 | GVL_LLM_LLAMA */
  /* CLIP ViT */
  int32_t clip_hidden, clip_inter, clip_layers_run, clip_heads, clip_image, clip_patch;
  /* InternVideo2 */
  int32_t iv2_dim, iv2_inter, iv2_blocks_run, iv2_heads, iv2_image, iv2_patch, iv2_frames_per_seg;
  /* LLM */
  int32_t hidden, inter, layers, heads, kv_heads, vocab;
  float rms_eps;
  int32_t lm_head_bias;        /* 1: logits = W h + b (llava_next_video.py:263) */
  int32_t rope_orig_max_pos;   /* LongRoPE switch point (4096); 0 = plain RoPE (one table) */
  int32_t max_seq;             /* rows of the rope tables / largest context of one sequence */
  /* limits */
  int32_t max_segs;            /* largest number of segments per encode call on this rank */
  int32_t kv_pages;            /* pages (64 tokens each, all layers) in the paged KV pool */
  int32_t max_prefill;         /* largest prefill length (rows of the activation workspace) */
  int32_t decode_fp8;          /* quantised weight variants of the LLM, opt-in, NOT the reference's numerics.  0: bf16.
                                  1: FP8 -- every decoder projection and lm_head is quantised at gvl_finalize_weights to OCP e4m3 with a
                                  per-row power-of-two scale; the decode path streams the FP8 copy (half the bytes).  2: MXFP4 (OCP
                                  Microscaling v1.0: E2M1 elements, one E8M0 scale per 32 consecutive k) -- a quarter of the bytes.
                                  In both the prefill path uses the de-quantised bf16 values: ONE model.  Needs hidden / inter /
                                  heads*head_dim multiples of 512 (FP8) / 1024 (MXFP4). */
} gvl_config;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int gvl_create(const gvl_config* cfg, gvl_ctx** out);
int gvl_destroy(gvl_ctx* ctx);
const char* gvl_last_error(const gvl_ctx* ctx);       /* ctx may be NULL: last create() error */
int gvl_device_info(char* arch_out, int arch_len, int* num_cus);

/* Packed weights.  `name` is one of the packed names produced by
 * grounded_video_llm_amd/weights.py from the reference state-dict keys;
 * replaces nn.Module.load_state_dict (inference.py:156-162).  `data` may be a host or device
 * pointer (is_device); the library converts to its internal dtype and keeps its own copy.
 * Synchronous. */
int gvl_load_weight(gvl_ctx* ctx, const char* name, const void* data, int dtype, const int64_t* shape,
                    int ndim, int is_device);
int gvl_finalize_weights(gvl_ctx* ctx);               /* checks every required tensor is present */
/* All packed tensors from ONE file written by tools/pack_checkpoint.py (`gvl-packed-1`: a safetensors container whose tensor
 * names are the packed names above; LoRA merged, q/k/v fused, K padded, pos-embed interpolated, RoPE tables built offline).
 * The C++ twin of `nn.Module.load_state_dict(torch.load(...))` (inference.py:156-162, models/llava_next_video.py:117-151) for a
 * host that has no Python: the file is mapped read-only and every tensor goes through gvl_load_weight.  Synchronous.  Call
 * gvl_finalize_weights afterwards.  *n_loaded (may be NULL) receives the number of tensors. */
int gvl_load_packed(gvl_ctx* ctx, const char* path, int* n_loaded);

/* ---- vision hot path ------------------------------------------------------------------------ */
/* vision_tower(px, output_hidden_states=True).hidden_states[-2][:, 1:]
 * (llava_next_video.py:504-505).  px f32 [n,3,336,336] -> out f32 [n,576,clip_hidden]. */
int gvl_clip_encode(gvl_ctx* ctx, const float* px, int n_images, float* out, void* stream);

/* video_encoder(x, None, False, x_vis_return_idx=-2, x_vis_only=True)[:, 1:, :]
 * (llava_next_video.py:532).  px f32 [n,3,T,224,224] -> out bf16 [n, T*256, iv2_dim]. */
int gvl_iv2_encode(gvl_ctx* ctx, const float* px, int n_segs, uint16_t* out, void* stream);

/* merge/pool + both projectors + newline + concat (llava_next_video.py:507-564) for n_segs
 * segments.  clip_feats f32 [n,576,clip_hidden], iv2_feats bf16 [n,T*256,iv2_dim]
 * -> visual bf16 [n * gvl_tokens_per_seg(), hidden]. */
int gvl_build_visual(gvl_ctx* ctx, const float* clip_feats, const uint16_t* iv2_feats, int n_segs,
                     uint16_t* visual, void* stream);
int gvl_tokens_per_seg(const gvl_ctx* ctx);           /* 285 (Phi-3.5) / 193 (Llama-3) at 8 f/seg */

/* encode_images() for the caller's segments = the three calls above on the ctx workspace.
 * spatial f32 [n,3,336,336], temporal f32 [n,3,T,224,224] (already "(b s) c f h w"). */
int gvl_encode_segments(gvl_ctx* ctx, const float* spatial_px, const float* temporal_px, int n_segs,
                        uint16_t* visual, void* stream);

/* prepare_multimodal_inputs() for one sample (llava_next_video.py:568-596): ids int64 host
 * [n_ids] with exactly one IMAGE_TOKEN_INDEX (-200); embeds bf16 [n_ids-1+n_visual, hidden]. */
int gvl_splice(gvl_ctx* ctx, const int64_t* ids_host, int n_ids, const uint16_t* visual, int n_visual,
               uint16_t* embeds, int* seq_len_out, void* stream);

/* ---- LLM hot path --------------------------------------------------------------------------- */
/* language_model.generate(inputs_embeds=...) (llava_next_video.py:655-661), greedy:
 *   seq_alloc  -> reserves KV pages for up to max_tokens (DynamicCache replacement, paged)
 *   prefill    -> step 0 of generate(): writes KV, returns last-position logits (f32 [vocab],
 *                 device pointer, may be NULL)
 *   decode_greedy -> steps 1..N on the device; out_ids int32 host [max_new]; stops after eos
 *                 (eos < 0 disables).  Synchronises the stream before returning. */
int gvl_seq_alloc(gvl_ctx* ctx, int max_tokens, int* seq_id);
int gvl_seq_free(gvl_ctx* ctx, int seq_id);
/* Paged KV pool of this ctx (replaces transformers' DynamicCache, models/modeling_phi3.py:1291 [ext]): pages of 64 tokens over all
 * layers.  cfg.kv_pages > 0 fixes the pool size at gvl_create; cfg.kv_pages <= 0 sizes it in gvl_finalize_weights from the HBM
 * that is free once the weights are resident (env GVL_KV_FRACTION, default 0.85 of it, minus 4 GiB) -- on a 288 GB MI355X about
 * 670 k Phi-3.5 tokens.  Any out pointer may be NULL.
 * The fused RMSNorm (gvl_debug_set "norm_fused") keeps norm-folded second copies of qkv / fc1 (InternVideo2) and qkv_proj / gate_up_proj / lm_head (LLM;
 * with bf16 decode weights also their decode tile copies): about 1.1 GB + 10 GB (Phi-3.5) / 18 GB (Llama-3-8B).  env GVL_NORM_FOLD=0 skips them (the norms
 * then run as separate passes); a failed allocation does the same by itself -- finalize does not fail for them. */
/* Prefix sharing (the reference asks three questions about ONE video, inference.py:178-182: the prompts share the system prompt and the
 * 3 420 visual tokens).  gvl_seq_fork makes a new sequence whose first n_tokens (a multiple of 64 = whole KV pages, <= the source's
 * tokens) ARE the source's pages -- referenced, not copied; a page returns to the pool when its last holder is freed -- and reserves
 * fresh pages up to max_tokens.  gvl_prefill_extend then runs the decoder over the remaining n_new prompt rows only: they take
 * positions prefix .. prefix + n_new - 1 and attend to the cached prefix plus themselves; last_logits / first token as gvl_prefill.
 * With a prefix that is a multiple of 128 tokens the result is BIT-IDENTICAL to a gvl_prefill of the whole prompt (same query blocks,
 * same page tiles, same k order); any multiple of 64 is within bf16 rounding of it.  LongRoPE models (cfg.rope_orig_max_pos > 0): a prefill
 * picks ONE factor set from the length it sees (short up to the original context, long beyond; modeling_phi3.py:381-385) -- the prefix's
 * cached K carries the choice made when IT was prefilled, so the identity holds only when prefix and whole prompt fall on the same side of
 * rope_orig_max_pos; the host must not share a prefix <= rope_orig_max_pos with a prompt longer than it (model.py falls back to full
 * prefills). */
int gvl_seq_fork(gvl_ctx* ctx, int src_seq, int n_tokens, int max_tokens, int* dst_seq);
int gvl_prefill_extend(gvl_ctx* ctx, int seq_id, const uint16_t* embeds_new, int n_new, float* last_logits, void* stream);
/* A copy of a sequence AT ITS CURRENT LENGTH (beam search: HF's cache reorder, transformers GenerationMixin._reorder_cache [ext]): whole pages
 * are shared by reference, the partial last page is copied on `stream` (one launch over all layers); the clone then appends to its own pages. */
int gvl_seq_clone(gvl_ctx* ctx, int src_seq, int max_tokens, int* dst_seq, void* stream);
int gvl_kv_info(const gvl_ctx* ctx, int* total_pages, int* free_pages, int64_t* pool_bytes, int* max_live_seqs);
/* The group sizes ONE batched decode step takes (gvl_decode_step_logits_batch, and the parts gvl_decode_greedy_batch / gvl_decode_steps
 * step together), valid after gvl_finalize_weights: *max_group = 16 on the skinny-MFMA decode path, 4 on the VALU fallback
 * (geometries whose projection widths are not multiples of 256); *any_size = 1 when every size 1 .. max_group is taken, 0 when only
 * 1, 2 and 4 are.  Hosts that form their own groups (beam search) ask here instead of restating the library's predicate. */
int gvl_decode_group_info(const gvl_ctx* ctx, int* max_group, int* any_size);
int gvl_prefill(gvl_ctx* ctx, int seq_id, const uint16_t* embeds, int seq_len, float* last_logits,
                void* stream);
int gvl_decode_greedy(gvl_ctx* ctx, int seq_id, int max_new, int eos_id, int32_t* out_ids_host,
                      int* n_out, void* stream);
/* Token selection of every later gvl_prefill* / gvl_decode_* call on this ctx.  The reference forwards do_sample / temperature /
 * top_p from generate(**kw) to HF generate (models/llava_next_video.py:655-661; inference.py:45-49 defaults: do_sample True,
 * temperature 0.2, top_p None; HF's GenerationConfig adds top_k 50 [ext]).  do_sample = 0: greedy argmax (the default of a new ctx).
 * do_sample = 1: scores / temperature -> top-k (0 = off; ties with the k-th score are kept) -> top-p (0 or 1 = off; a token is kept
 * iff the probability mass of strictly larger scores is < top_p) -> ONE draw from the softmax of what is left, all on the device
 * inside the decode step.  The draw of a sequence is a pure function of (seed, the order in which sequences were prefilled since
 * this call, generation step, logits): reproducible, and independent of how sequences are grouped into decode batches.  (A call
 * that repeats the current seed while sequences are live continues the numbering instead, so newcomers never share a stream with
 * a running sequence; callers that make several generate() calls per request pass a different seed per call.)
 * torch.multinomial's random stream is not reproduced (parity = same kept set + same distribution).  Beam search (num_beams > 1, do_sample = 0) is host
 * bookkeeping over gvl_seq_clone + gvl_decode_step_logits_batch (grounded_video_llm_amd/beam.py); beam-sample (num_beams > 1, do_sample = 1) is the
 * same bookkeeping with the 2 x num_beams candidates of a step drawn on the host from the warped beam distributions (beam.py). */
int gvl_set_sampling(gvl_ctx* ctx, int do_sample, float temperature, int top_k, float top_p, uint64_t seed);
/* Prefill of n_seqs sequences together, seq_lens[i] tokens each (ragged: prompts differ in length; the reference left-pads and
 * masks, llava_next_video.py:622-647 -- here the rows are packed back to back, no padding).  Groups of 4 / 2 / 1 sequences whose
 * rows fit cfg.max_prefill: the decoder GEMMs run over all rows of a group at once (better tile fill); RoPE / KV append / causal
 * attention stay per sequence on its own pages (one launch with a batch dimension when the lengths agree, one per sequence
 * otherwise).  Per sequence bit-identical to gvl_prefill.  embeds: host array of n_seqs device pointers (bf16 [len, hidden]). */
int gvl_prefill_varlen(gvl_ctx* ctx, const int* seq_ids, int n_seqs, const uint16_t* const* embeds,
                       const int* seq_lens, void* stream);
/* gvl_prefill_varlen with every length == seq_len. */
int gvl_prefill_batch(gvl_ctx* ctx, const int* seq_ids, int n_seqs, const uint16_t* const* embeds, int seq_len,
                      void* stream);
/* Batched greedy decode of n_seqs freshly prefilled sequences (the reference batches clips in generate() with
 * left padding, llava_next_video.py:622-647 -- here every sequence keeps its own pages and length, no padding).  Groups of
 * up to 16 sequences advance together (skinny MFMA GEMM, gvl_decode.hip; 4 / 2 / 1 on the VALU fallback for K % 256 != 0
 * geometries): every weight matrix is streamed ONCE per step for the whole group, so the HBM cost per
 * sequence falls as 1/group; the per-sequence arithmetic (and therefore the ids) is bit-identical to gvl_decode_greedy.
 * out_ids_host int32 [n_seqs][max_new]; n_out [n_seqs].  A group runs until all of its members hit eos / max_new. */
int gvl_decode_greedy_batch(gvl_ctx* ctx, const int* seq_ids, int n_seqs, int max_new, int eos_id,
                            int32_t* out_ids_host, int* n_out, void* stream);
/* Continuous batching: the two calls a scheduler needs besides prefill / seq_alloc / seq_free.
 * gvl_decode_steps advances every listed sequence by n_steps greedy tokens -- the sequences may be at DIFFERENT generation
 * steps (joined at different times); groups of up to 16 share one weight stream per step; asynchronous on `stream`, no eos
 * test (the host inspects the ids between chunks; tokens after an eos are discarded by the caller).
 * gvl_seq_read copies the ids generated so far, from index `first`, to the host (at most cap), reports the total count in
 * *n_gen and synchronises `stream`.  Free a sequence only after a gvl_seq_read / stream synchronise that follows its last step. */
int gvl_decode_steps(gvl_ctx* ctx, const int* seq_ids, int n_seqs, int n_steps, void* stream);
int gvl_seq_read(gvl_ctx* ctx, int seq_id, int first, int32_t* out_ids_host, int cap, int* n_gen, void* stream);
/* teacher-forced single step (parity tests): appends token `tok`, returns logits f32 [vocab]. */
int gvl_decode_step_logits(gvl_ctx* ctx, int seq_id, int tok, float* logits, void* stream);
/* The same for up to 16 sequences in ONE step (one stream of the weights): sequence i takes toks[i]; logits (may be null) receives
 * [n_seqs][vocab] fp32.  Row i is bit-identical to gvl_decode_step_logits on sequence i alone.  Beam search advances its k beams with it. */
int gvl_decode_step_logits_batch(gvl_ctx* ctx, const int* seq_ids, int n_seqs, const int32_t* toks, float* logits, void* stream);

/* ---- multi-GPU exchange (SURVEY.md §8 e) -------------------------------------------------------------------------------------
 * The reference's inference is single-GPU (inference.py:17); sharding the frame batch over the GPUs of a node is this build's
 * addition: every rank encodes its segments, ONE all-gather moves the per-segment token blocks (llava_next_video.py:563) to
 * every rank.  RCCL is loaded with dlopen("librccl.so.1") on first use; a host without RCCL gets GVL_ERR_STATE.
 *   gvl_comm_unique_id : ncclGetUniqueId on rank 0; the 128 bytes travel to the other ranks by the host's own means
 *   gvl_comm_init      : ncclCommInitRank on the ctx's device (collective over all ranks).  Env GVL_RCCL_LIB overrides the
 *                        library name (default: librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1)
 *   gvl_allgather_visual : local bf16 [rows_per_rank, hidden] -> all bf16 [world * rows_per_rank, hidden] (rank order) with
 *                        ncclAllGather on the caller's stream; `comm` = an ncclComm_t the host already owns, or NULL for the
 *                        ctx's communicator; world == 1 degenerates to one device copy. */
int gvl_comm_unique_id(char id_out[128]);
int gvl_comm_init(gvl_ctx* ctx, const char id[128], int rank, int world);
int gvl_comm_destroy(gvl_ctx* ctx);
/* ranks RCCL itself reports for the ctx's communicator (ncclCommCount); 1 when there is none.  bench.py prints it so that a
 * multi-GPU line shows how many ranks the collective really spanned. */
int gvl_comm_count(gvl_ctx* ctx, int* n_ranks);
int gvl_allgather_visual(gvl_ctx* ctx, void* comm, const uint16_t* local, int rows_per_rank, int hidden, uint16_t* all,
                         void* stream);
/* The same exchange for UNEVEN blocks, written straight into the segment-ordered prefix (llava_next_video.py:563: the per-segment blocks in segment
 * order): rank r contributes rows_per_rank[r] rows ([world] ints, host memory, the same on every rank) and they land at row offset
 * sum(rows_per_rank[0..r)) of `all` on every rank -- no padding to the largest block, no re-assembly copy.  One ncclGroup of per-rank broadcasts on
 * the caller's stream; `comm` must be NULL (the ctx's communicator).  Without a communicator: one device copy (or nothing when local == all). */
int gvl_allgatherv_visual(gvl_ctx* ctx, void* comm, const uint16_t* local, const int* rows_per_rank, int hidden, uint16_t* all, void* stream);

/* ---- training forward ---------------------------------------------------------- */
/* LLAVA_NEXT_VIDEO.forward(samples)["loss"] for ONE sample (llava_next_video.py:598-614): the causal-LM loss of
 * language_model(inputs_embeds, labels) -- Phi3ForCausalLM.forward labels branch (modeling_phi3.py:1512-1539; Llama alike):
 * logits[:-1] against labels[1:], ignore_index -100.  embeds bf16 [seq_len, hidden] (device, the spliced prefix WITHOUT the
 * masked right padding); labels int64 [seq_len] (host).  Returns the SUM of the token losses and their count, so that a batch
 * loss is sum(nll_sum) / sum(n_valid) exactly like CrossEntropyLoss(mean) over the flattened padded batch.  Only rows with a
 * label reach the lm_head.  Uses seq_id's KV pages as scratch (allocate >= seq_len tokens, free afterwards).  Forward only. */
int gvl_forward_loss(gvl_ctx* ctx, int seq_id, const uint16_t* embeds, int seq_len, const int64_t* labels_host,
                     double* nll_sum, int* n_valid, void* stream);

/* ---- frame pre-processing ---------------------------------------------------- */
/* frame_transform(image_size = size, mean, std) of the reference (mm_utils/utils.py:153-183, called per frame from
 * inference.py:69-88) for n uint8 RGB frames of one video: torchvision Resize(size, BICUBIC) [shortest edge -> size;
 * = PIL.Image.resize, Pillow Resample.c] -> CenterCrop(size) -> ToTensor -> Normalize, bit-exact to the CPU chain
 * (8-bit two-pass fixed-point resampler; three IEEE f32 operations for ToTensor/Normalize).
 * frames: device uint8, layout 0 = [n][H][W][3] (decoder order, mm_utils/video_utils.py:82) or 1 = [n][3][H][W]
 * (after the reference's permute, :91); out: device f32 [n][3][size][size].  Needs no weights (any ctx). */
int gvl_preprocess_frames(gvl_ctx* ctx, const uint8_t* frames, int n, int height, int width, int layout, int size,
                          const float* mean, const float* std3, float* out, void* stream);

/* ---- measurement ---------------------------------------------------------------------------- */
/* When enabled, every launch of a kernel family is bracketed by hipEvents on the caller's stream
 * (slow; bench.py uses it for ONE profiled step after the timed region). */
enum { GVL_PROF_GEMM = 0, GVL_PROF_ATTN = 1, GVL_PROF_GEMV = 2, GVL_PROF_DECODE_ATTN = 3, GVL_PROF_OTHER = 4,
       GVL_PROF_NCAT = 5 };
int gvl_prof_enable(gvl_ctx* ctx, int on);
int gvl_prof_read(gvl_ctx* ctx, int category, double* total_ms, int64_t* launches, double* work);

/* Result-neutral launch parameters, for tests and A/B measurements (the shipped library reads no kernel-selection environment variable):
 *   "decode_attn_cpb"  consecutive context splits one decode-attention block works through (0 = launcher's choice, 1..16)
 *   "decode_attn_hpb"  query heads of a KV head served by one block of the grouped-query decode kernel (0 = launcher's choice)
 *   "decode_graph"     1 (default): a decode group's step is captured once and replayed as a hipGraph for the following tokens; 0: eager
 *   "prefill_group"    sequences whose rows share one pass of the prefill GEMMs in gvl_prefill_varlen / _batch (1 .. 8, default 4: at the bench's 3.5 k-row prompts
 *                      8 measured neutral on clips/s and 0.7 % slower on the GEMM family -- 0.9 GB of activations per projection fall out of the
 *                      Infinity Cache; shorter prompts gain up to 4.7 % from 8, profiles/r03_gemm_prefill_group.txt)
 *   "vision_in_place"  1 (default): non-causal attention (vision towers, gvl_op_attention) reads V -- and Q, K when the head dim needs no padding
 *                      or transform; with InternVideo2's q RMSNorm applied in the kernel prologue -- straight from the fused-qkv matrix;
 *                      2: V only; 0: the round-2 path through Q / K pages and a V^T transpose pass
 *   "attn_pipe_rows"   128 (default): that kernel's 4-wave form (128 query rows per block) for every row; 256: whole 256-row query blocks on its 8-wave form (half
 *                      the DMA pieces per MFMA; measured 12 % slower), the remaining rows on the 4-wave form -- bit-identical
 *   "varlen_attn"      1 (default): the causal attention of a ragged prefill group (gvl_prefill_varlen) runs as ONE grid over the query blocks of all its sequences,
 *                      and so do its RoPE / KV-append and V^T-page passes (one launch each per layer); 2: the attention only (round 5), the passes per sequence;
 *                      0: one launch per sequence for all of them (rounds 2-4) -- bit-identical
 *   "norm_fused"       1 (default): RMSNorm in front of qkv / fc1 (InternVideo2) and qkv_proj / gate_up_proj / lm_head (LLM prefill AND, with bf16 decode weights,
 *                      the decode step) is fused into the GEMMs around it
 *                      (row statistics from the producing GEMM's epilogue, norm weight folded into the consuming GEMM's weight, row scale in its epilogue); 0: the
 *                      separate norm pass of rounds 1-4.  NOT bit-neutral -- the third stated exception below: two activation roundings of the reference
 *                      (x * rs and the gamma product, both to bf16) are gone and gamma * W is rounded once per weight instead
 *   "gemm_band"        0 (default): the ping-pong GEMM's rasterisation band is 8 tile rows; 1..64: that many rows for every later launch of the PROCESS (A/B only)
 *                      -- bit-identical
 *   "gemm_a4"          which form of the 256 x 256 GEMM kernel a launch takes: 1 (default) per fused epilogue, as measured -- the 4-wave kernel with the epilogue
 *                      pipelined into the next tile's main loop (gvl_gemm4p.hip) for erf-GELU / SwiGLU / residual + row statistics, the plain 4-wave kernel
 *                      (gvl_gemm4.hip: accumulators in AGPRs, hand-placed k loop) for store-only epilogues, the 8-wave ping-pong kernel for the rest;
 *                      0: always the 8-wave kernel; 2: the plain 4-wave kernel wherever it serves; 3: the pipelined one wherever it serves.  For every later
 *                      launch of the PROCESS -- bit-identical
 *   "gemm_narrow"      1 (default): the pipelined 4-wave kernel runs a column tile with <= 128 real columns (N = 1408 = 5.5 tile columns: every sixth tile of
 *                      InternVideo2's proj / fc2) as a NARROW tile -- 4 waves x (128 rows x 64 columns), no MFMA on the empty half; 0: as a full tile.  For every
 *                      later launch of the PROCESS -- bit-identical
 *   "last_layer_tail"  1 (default): a prefill without a loss request runs the LAST decoder layer's MLP on the sequences' last rows only (nothing else reads its
 *                      output); 0: on every row -- bit-identical
 *   "patch_fused"      1 (default): the patch embedding of a tower whose geometry the fused kernel covers (patch 14, width 1024 / 1408) runs as ONE kernel
 *                      (gvl_patch.hip: im2col in the operand loader + GEMM + CLS / position rows + CLIP's pre-LayerNorm); 0: the three-pass path (patchify, GEMM,
 *                      embed).  NOT bit-neutral: the fp32 accumulation order over k differs (agreement to fp32 rounding before the bf16 round; tests/test_gpu_towers.py)
 *   "attn_pipe"        1 (default): InternVideo2's attention (head dim 88, q in place) runs the software-pipelined key-tile loop
 *                      (attn_iv2_pipe_kernel, round 4); 0: the plain loop of attn_fwd_kernel; 2: the pipelined kernel's SAFE pass alone.  Bit-identical to 0
 *                      WHILE attn_fwd_kernel's lazy rule never fires after a row's first key tile (every golden, the bench: asserted in
 *                      tests/test_gpu_towers.py); the pipelined normal pass keeps the first tile's maximum as the softmax reference for the whole row, so on
 *                      scores where a later tile exceeds it by more than 2^8 the two differ at P-rounding level (bounded at 1.5e-2 of the output scale by
 *                      the sharp-score test there) -- the second stated exception below; mode 2 is bit-identical to 0 on any data
 * None of them may change a single output bit (asserted in tests/test_gpu_llm.py) -- with THREE stated exceptions: "norm_fused" and "attn_pipe" = 1 on
 * peaked scores (above), and "vision_in_place" = 1 on a head
 * dim that is padded (InternVideo2, 88 -> 96) folds the softmax scale and shift into q before its one rounding to bf16, a different (not larger)
 * set of rounding points: modes 0 and 2 are bit-identical to each other, mode 1 is bit-identical to them for CLIP (head dim 64) and agrees within
 * bf16 noise for InternVideo2 (one block 4.1e-3 of the output scale; 39 blocks vs the reference: the same error as the reference's own bf16,
 * tests/test_gpu_towers.py, tests/test_gpu_c0.py). */
int gvl_debug_set(gvl_ctx* ctx, const char* key, int value);

/* ---- operator-level entry points (parity tests call the kernels through these) --------------- */
/* C[M,N] = act(A[M,K] W[N,K]^T + bias) (+ resid); A,W bf16; bias/gamma f32 or NULL.
 * act: 0 none, 1 quick_gelu, 2 gelu(erf), 3 silu(gate)*up on interleaved (gate,up) column pairs
 * (output has N/2 columns).  out_f32: C is f32 else bf16; resid has C's dtype and layout. */
int gvl_op_gemm(gvl_ctx* ctx, const uint16_t* A, const uint16_t* W, void* C, int M, int N, int K,
                const float* bias, const float* gamma, const void* resid, int act, int out_f32,
                int tile_cfg, void* stream);
/* Fused RMSNorm at operator level (round 5; what the towers and the prefill run: models/internvideo2.py:443-448,590-603, modeling_phi3.py:319-324):
 *   RMSNorm(x) W^T  ==  rs[m] * (x (W diag(gamma))^T),   rs[m] = rsqrt(mean_k x[m][k]^2 + eps)
 * gvl_op_gemm_rows: gvl_op_gemm with bf16 output plus rowscale ([M] f32 or NULL: multiplies the accumulator rows before bias / activation) and rowsq
 * ([M][rowsq_ld] f32 or NULL: the GEMM that WRITES the residual stream leaves the sum of squares of its rounded outputs per aligned block of 64 columns;
 * N % 64 == 0).  gvl_op_fold_gamma: Wo = bf16(W diag(gamma)), W [rows][cols], gamma [cols] bf16.  gvl_op_rowsq_finish: rs[m] = rsqrt((sum of the blocks
 * [b0, b0 + nblk) of row m) / cols + eps). */
int gvl_op_gemm_rows(gvl_ctx* ctx, const uint16_t* A, const uint16_t* W, uint16_t* C, int M, int N, int K, const float* bias, const float* gamma,
                     const uint16_t* resid, int act, const float* rowscale, float* rowsq, int rowsq_ld, int tile_cfg, void* stream);
int gvl_op_fold_gamma(gvl_ctx* ctx, const uint16_t* W, const uint16_t* gamma, uint16_t* Wo, int64_t rows, int cols, void* stream);
int gvl_op_rowsq_finish(gvl_ctx* ctx, const float* rowsq, int ld, int b0, int nblk, float* rs, int rows, int cols, float eps, void* stream);
/* attention over q/k/v bf16 [B,S,H|KV,D] (plain layout; the library re-tiles internally).
 * out bf16 [B,S,H*D].  causal: 0/1. */
int gvl_op_attention(gvl_ctx* ctx, const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* out,
                     int B, int S, int H, int KV, int D, float scale, int causal, void* stream);
int gvl_op_layernorm(gvl_ctx* ctx, const float* x, const float* w, const float* b, uint16_t* y, int rows,
                     int cols, float eps, void* stream);
int gvl_op_rmsnorm(gvl_ctx* ctx, const uint16_t* x, const uint16_t* w, uint16_t* y, int rows, int cols,
                   float eps, void* stream);
/* y[N] = W[N,K] x[K] (+bias) -- the decode GEMV; x,W bf16, y f32. */
int gvl_op_gemv(gvl_ctx* ctx, const uint16_t* W, const uint16_t* x, const float* bias, float* y, int N, int K,
                void* stream);
/* The sampler on its own (operator test): logits f32 device [batch][n] -> tokens_dev int32 device [batch]; row b draws from
 * random stream streams[b] (host array) at generation step steps_dev[b] (device array).  batch <= 16. */
int gvl_op_sample(gvl_ctx* ctx, const float* logits, int n, int batch, float temperature, int top_k, float top_p, uint64_t seed,
                  const uint32_t* streams, const int32_t* steps_dev, int32_t* tokens_dev, void* stream);
/* y[b][N] = W[N,K] x[b][K] (+bias) for b < batch <= 16 -- the decode projections as ONE skinny MFMA GEMM (the weight stream is
 * read once for all sequences; K % 256 == 0).  x bf16 [batch][K], y f32 [batch][N]. */
int gvl_op_dgemm(gvl_ctx* ctx, const uint16_t* W, const uint16_t* x, const float* bias, float* y, int N, int K, int batch,
                 void* stream);
/* measurement only (tools/decode_bench.py): average microseconds per launch of one decode projection [N,K] x `batch` sequences on
 * synthetic operands; mode 0 = skinny MFMA GEMM (variant 0 = default), 1 = VALU GEMV; `rounds` distinct weight copies are cycled. */
int gvl_op_decode_bench(gvl_ctx* ctx, int N, int K, int batch, int mode, int variant, int rounds, int iters, double* us_per_launch,
                        void* stream);

/* measurement only (tools/mfma_probe.py): a register-only kernel that keeps every matrix pipe 100 % busy with
 * v_mfma_f32_32x32x16_bf16 -- mode 0 zero / 1 constant / 2 random operands -- and reports TFLOP/s and s_memtime ticks per ns of wall
 * time: the ceiling the power envelope leaves a PERFECT bf16 GEMM at a given switching activity.  Needs no ctx. */
int gvl_probe_mfma(int mode, int waves_per_simd, int iters, double* tflops, double* ghz, void* stream);
/* One named do-nothing dispatch (gvl_trace_marker_kernel) on `stream`: brackets a region of a rocprofv3 --kernel-trace (tools/rocpd_stats.py --between). */
int gvl_trace_marker(int tag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GVL_H */
