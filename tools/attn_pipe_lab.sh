#!/bin/bash
# Round-4 LAB: what bounds attn_iv2_pipe_kernel.  LAB builds (GVL_BUILD_TAG=plN GVL_BUILD_DEFS=-DGVL_PIPE_LAB=N python grounded-video-llm_amd/build.py) remove ONE activity
# from the pipelined key-tile loop (wrong results on purpose); this script times InternVideo2's attention family (96 segments, 39 launches) under each, then (PMC=1) collects
# the PMC counters of the shipped kernel.   gpurun -- bash tools/attn_pipe_lab.sh "pl1 pl2 ..."
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pipe_lab; mkdir -p $OUT
for v in "" ${1:-pl1 pl2 pl3 pl4 pl8 pl15}; do
  lib=$R/grounded-video-llm_amd/libgvl${v:+_$v}.so
  [ -f $lib ] || continue
  echo "== ${v:-shipped}" | tee -a $OUT/lab.txt
  GVL_LIB_PATH=$lib timeout 300 python $R/tools/vision_ab.py 1 attn_pipe 1 0 2>&1 | grep vision_ab | tee -a $OUT/lab.txt
done
[ -n "$PMC" ] || exit 0
cd /tmp; export TMPDIR=/tmp
db() { find "$1" -name "*.db" | head -1; }
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -- python $R/tools/iv2_one.py 2 1 > $OUT/pmc$i.log 2>&1
  python $R/tools/rocpd_pmc.py "$(db $OUT/pmc$i)" attn >> $OUT/pmc.txt 2>&1
  rm -rf $OUT/pmc$i
done
cat $OUT/pmc.txt
