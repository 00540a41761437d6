/* A C99 host that runs the WHOLE hot path on the GPU through include/gvl.h and nothing else from this repository:
 *   gvl_create -> gvl_load_packed (one packed safetensors file) -> gvl_finalize_weights -> gvl_encode_segments -> gvl_splice -> gvl_seq_alloc -> gvl_prefill
 *   -> gvl_decode_greedy
 * i.e. LLAVA_NEXT_VIDEO.generate() of the reference (models/llava_next_video.py:616-666) for one sample, as a cgo / JNI / N-API host would drive it.
 * Device memory comes from the HIP runtime's C API; no Python, no torch.  Built with the system C compiler and run by tests/test_gpu_c_host.py, which
 * compares the printed ids with the Python host's (same library: must be equal) and with the CPU oracle's.
 *   e2e_host cfg.bin weights.safetensors spatial.f32 temporal.f32 ids.i64 n_segs n_ids max_new eos_id */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "gvl.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, gvl_last_error(ctx)); return 10; } } while (0)
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_)); return 11; } } while (0)

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(12); }
  fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  void* p = malloc(*n ? *n : 1);
  if (fread(p, 1, *n, f) != *n) { fprintf(stderr, "short read on %s\n", path); exit(12); }
  fclose(f);
  return p;
}

int main(int argc, char** argv) {
  if (argc != 10) { fprintf(stderr, "usage: e2e_host cfg.bin weights.safetensors spatial.f32 temporal.f32 ids.i64 n_segs n_ids max_new eos_id\n"); return 1; }
  gvl_ctx* ctx = NULL;
  size_t n = 0;
  gvl_config* cfg = (gvl_config*)slurp(argv[1], &n);
  if (n != sizeof(gvl_config)) { fprintf(stderr, "cfg.bin holds %zu bytes, gvl_config has %zu\n", n, sizeof(gvl_config)); return 2; }
  const int n_segs = atoi(argv[6]), n_ids = atoi(argv[7]), max_new = atoi(argv[8]), eos = atoi(argv[9]);
  size_t nsp, ntp, nid;
  float* sp = (float*)slurp(argv[3], &nsp);
  float* tp = (float*)slurp(argv[4], &ntp);
  int64_t* ids = (int64_t*)slurp(argv[5], &nid);
  if (nid != (size_t)n_ids * 8) { fprintf(stderr, "ids.i64 holds %zu bytes, expected %d ids\n", nid, n_ids); return 2; }

  int n_loaded = 0;
  CHECK(gvl_create(cfg, &ctx));
  CHECK(gvl_load_packed(ctx, argv[2], &n_loaded));
  CHECK(gvl_finalize_weights(ctx));
  const int tps = gvl_tokens_per_seg(ctx), n_vis = n_segs * tps, hidden = cfg->hidden;
  void *d_sp = NULL, *d_tp = NULL, *d_vis = NULL, *d_emb = NULL;
  HIPCHECK(hipMalloc(&d_sp, nsp)); HIPCHECK(hipMalloc(&d_tp, ntp));
  HIPCHECK(hipMalloc(&d_vis, (size_t)n_vis * hidden * 2)); HIPCHECK(hipMalloc(&d_emb, (size_t)(n_ids - 1 + n_vis) * hidden * 2));
  HIPCHECK(hipMemcpy(d_sp, sp, nsp, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(d_tp, tp, ntp, hipMemcpyHostToDevice));

  int S = 0, seq = -1, n_out = 0;
  CHECK(gvl_encode_segments(ctx, (const float*)d_sp, (const float*)d_tp, n_segs, (uint16_t*)d_vis, NULL));
  CHECK(gvl_splice(ctx, ids, n_ids, (const uint16_t*)d_vis, n_vis, (uint16_t*)d_emb, &S, NULL));
  CHECK(gvl_seq_alloc(ctx, S + max_new, &seq));
  CHECK(gvl_prefill(ctx, seq, (const uint16_t*)d_emb, S, NULL, NULL));
  int32_t* out = (int32_t*)calloc((size_t)max_new, sizeof(int32_t));
  CHECK(gvl_decode_greedy(ctx, seq, max_new, eos, out, &n_out, NULL));
  printf("LOADED %d TENSORS; S = %d; IDS:", n_loaded, S);
  for (int i = 0; i < n_out; ++i) printf(" %d", out[i]);
  printf("\n");
  CHECK(gvl_seq_free(ctx, seq));
  hipFree(d_sp); hipFree(d_tp); hipFree(d_vis); hipFree(d_emb);
  gvl_destroy(ctx);
  return 0;
}
