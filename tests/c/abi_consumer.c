/* A plain C99 consumer of include/gvl.h -- what a non-Python host (cgo / JNI / N-API shim) would compile.
 * Built and run by tests/test_host_logic.py::test_header_is_plain_c_and_links: proves the header needs nothing but <stdint.h>,
 * that libgvl.so links from C, and that on a host without a GPU the library refuses loudly instead of falling back to a CPU path. */
#include <stdio.h>
#include <string.h>
#include "gvl.h"

int main(void) {
  gvl_config cfg;
  gvl_ctx* ctx = NULL;
  char arch[64] = {0};
  int cus = 0;
  memset(&cfg, 0, sizeof cfg);
  int di = gvl_device_info(arch, (int)sizeof arch, &cus);
  int rc = gvl_create(&cfg, &ctx);
  printf("device_info=%d arch=%s cus=%d create=%d err=%s\n", di, arch, cus, rc, gvl_last_error(NULL));
  /* the entry points a non-Python host needs to run the whole path must link from C and reject a NULL ctx without crashing:
   * weights from ONE packed file, KV pool query, the visual-token exchange */
  {
    int n = -1, tp = -1;
    char uid[128];
    memset(uid, 0, sizeof uid);
    if (gvl_load_packed(NULL, "/nonexistent.gvl.safetensors", &n) != GVL_ERR_ARG) return 4;
    if (gvl_kv_info(NULL, &tp, NULL, NULL, NULL) != GVL_ERR_ARG) return 5;
    if (gvl_allgather_visual(NULL, NULL, NULL, 0, 0, NULL, NULL) != GVL_ERR_ARG) return 6;
    if (gvl_comm_init(NULL, uid, 0, 1) != GVL_ERR_ARG) return 7;
    if (gvl_comm_destroy(NULL) != GVL_ERR_ARG) return 8;
    if (gvl_set_sampling(NULL, 1, 0.2f, 50, 0.0f, 42u) != GVL_ERR_ARG) return 9;   /* HF generate's do_sample=True on the device */
  }
  if (di == GVL_ERR_NOGPU) return (rc == GVL_ERR_NOGPU && ctx == NULL) ? 0 : 2;   /* no GPU: must refuse */
  if (rc == 0 && ctx) gvl_destroy(ctx);                                           /* GPU present: an all-zero config is a valid empty ctx or an ARG error */
  return (rc == 0 || rc == GVL_ERR_ARG) ? 0 : 3;
}
