cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/t10.log
for m in plain bias_gelu bias_gamma_resid; do GVL_GEMM_TIMING=1 python tools/gemm_one.py 24588 6144 1408 82 1 $m; done > gpurun_out/tm10.log 2>&1
GVL_GEMM_TIMING=1 python tools/gemm_one.py 8192 8192 8192 82 1 plain >> gpurun_out/tm10.log 2>&1
GVL_BENCH_EPI=model python tools/gemm_bench.py 0,21,82 > gpurun_out/gb10.log 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b10.log 2>&1
