#!/bin/bash
# Round-3 parity ablation: the full-depth noise-class numbers of tests/ under each numerics-relevant LAB switch.
# Needs the LAB build:  GVL_BUILD_TAG=lab GVL_BUILD_DEFS=-DGVL_LAB python grounded-video-llm_amd/build.py
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r3; mkdir -p $O; OUT=$O/parity_ablation.txt; : > $OUT
LIB=$PWD/grounded-video-llm_amd/libgvl_lab.so
run() {  # name, env assignments...
  local name=$1; shift
  echo "=== $name" >> $OUT
  env GVL_LIB_PATH=$LIB "$@" timeout 1200 python -m pytest tests/test_gpu_c0.py tests/test_gpu_llama_fullsize.py -m gpu -q -s -k "internvideo2 or prefill_greedy or c1_headline or c4_full" 2>&1 \
    | grep -E "ratio max|passed|failed" | sed 's/^\.*//' | cut -c1-330 >> $OUT
}
run "shipped numerics (lazy softmax reference 2^8, row sum from the ones-row of V^T, table erf-GELU)" GVL_ATTN_LAZY=8
run "exact online-softmax rescale rule (GVL_ATTN_LAZY=0)" GVL_ATTN_LAZY=0
run "fp32 VALU row sum instead of the bf16 ones-row in the P.V MFMA (GVL_ATTN_NO_ONES=1)" GVL_ATTN_NO_ONES=1
run "both" GVL_ATTN_LAZY=0 GVL_ATTN_NO_ONES=1
cat $OUT
