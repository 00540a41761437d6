#!/bin/bash
# Round-N profiling passes on the GPU box (run through gpurun): kernel traces of the default and the serial bench, the two PMC
# traffic passes, MFMA-utilisation counters per GEMM shape.  Outputs under gpurun_out/prof_$1/ ; summaries are copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r05}; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
WHAT=${2:-all}
db() { find "$1" -name "*.db" | head -1; }
if [[ $WHAT == all || $WHAT == trace ]]; then
  rocprofv3 --kernel-trace -d $OUT/default -- python $R/bench.py --no-cpu-baseline > $OUT/default.log 2>&1
  python $R/tools/rocpd_stats.py "$(db $OUT/default)" $OUT/${TAG}_bench_kernel_stats.txt > /dev/null
  rocprofv3 --kernel-trace -d $OUT/serial -- python $R/bench.py --plain --mode serial --steps 6 --warmup 1 > $OUT/serial.log 2>&1
  python $R/tools/rocpd_stats.py "$(db $OUT/serial)" $OUT/${TAG}_bench_kernel_stats_serial.txt > /dev/null
  # the launches of the TIMED step (8 clips: batched CLIP / InternVideo2 / ragged prefill / decode) serialised on one stream: 3 steps = 24 clips.
  # `roofline.gemm_ms_per_clip` of the bench line must equal (sum of the gemm_* rows) / 24 of this table.
  rocprofv3 --kernel-trace -d $OUT/serial_step -- python $R/bench.py --plain --mode serial_step --steps 3 --warmup 1 > $OUT/serial_step.log 2>&1
  # round 5: the table holds the TIMED steps only (3 steps = 24 clips between the two gvl_trace_marker_kernel dispatches bench.py --plain places): no weight
  # generation, no pool zeroing, and the header counts the at::native / rocclr dispatches inside the window (0)
  python $R/tools/rocpd_stats.py "$(db $OUT/serial_step)" $OUT/${TAG}_bench_kernel_stats_serial_step.txt --between gvl_trace_marker_kernel > /dev/null
fi
if [[ $WHAT == all || $WHAT == pmc ]]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python $R/bench.py --plain --mode serial_step --steps 1 --warmup 1 > $OUT/fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python $R/bench.py --plain --mode serial_step --steps 1 --warmup 1 > $OUT/write.log 2>&1
  python $R/tools/pmc_traffic.py "$(db $OUT/fetch)" "$(db $OUT/write)" $OUT/${TAG}_pmc_traffic.json 16 > /dev/null
fi
if [[ $WHAT == all || $WHAT == mfma ]]; then
  : > $OUT/${TAG}_gemm_mfma_util.txt
  while read name M N K mode; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/mfma_$name -- python $R/tools/gemm_one.py $M $N $K 0 5 $mode > $OUT/mfma_$name.log 2>&1
    python $R/tools/mfma_util.py "$(db $OUT/mfma_$name)" $name $M $N $K 5 >> $OUT/${TAG}_gemm_mfma_util.txt
  done <<SHAPES
clip.patch 27648 1024 640 plain
iv2.patch 24576 1408 640 bias
clip.patch.benchM 55296 1024 640 plain
iv2.patch.benchM 196608 1408 640 bias
clip.qkv 27696 3072 1024 bias
clip.fc1 27696 4096 1024 bias_qgelu
clip.fc2 27696 1024 4096 bias_resid32
iv2.qkv 24588 4224 1408 plain
iv2.proj 24588 1408 1408 bias_gamma_resid
iv2.fc1 24588 6144 1408 bias_gelu
iv2.fc2 24588 1408 6144 bias_gamma_resid
phi.qkv 3519 9216 3072 plain
phi.o 3519 3072 3072 resid
phi.gu 3519 16384 3072 silu
phi.down 3519 3072 8192 resid
sq8192 8192 8192 8192 plain
SHAPES
fi
if [[ $WHAT == mfma_bench ]]; then
  # the GEMMs of the TIMED step at the bench's M (8 clips / a 4-sequence prefill group), with the epilogues the model runs (fused RMSNorm forms): matrix-pipe busy
  # fraction and the shader clock under that load -- their product is what the board's power budget caps
  : > $OUT/${TAG}_gemm_mfma_util_bench.txt
  while read name M N K mode; do
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/mfmab_$name -- python $R/tools/gemm_one.py $M $N $K 0 5 $mode > $OUT/mfmab_$name.log 2>&1
    python $R/tools/mfma_util.py "$(db $OUT/mfmab_$name)" $name $M $N $K 5 >> $OUT/${TAG}_gemm_mfma_util_bench.txt
  done <<SHAPES
iv2.qkv 196704 4224 1408 rs
iv2.proj 196704 1408 1408 bias_gamma_resid_sq
iv2.fc1 196704 6144 1408 rs_bias_gelu
iv2.fc2 196704 1408 6144 bias_gamma_resid_sq
phi.qkv 14076 9216 3072 rs
phi.o 14076 3072 3072 resid_sq
phi.gu 14076 16384 3072 rs_silu
phi.down 14076 3072 8192 resid_sq
clip.qkv 221568 3072 1024 bias
clip.fc1 221568 4096 1024 bias_qgelu
sq8192 8192 8192 8192 plain
SHAPES
fi
# the rocprofv3 databases are hundreds of MB: only the summaries travel back (gpurun_out/ is capped at 64 MiB)
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
rm -f $OUT/*.log.big
ls -la $OUT/*.txt $OUT/*.json 2>/dev/null
