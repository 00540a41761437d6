cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/t7.log
GVL_BENCH_EPI=model python tools/gemm_bench.py 0,21,82,85 > gpurun_out/gb7.log 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b7.log 2>&1
