cd /root/repo
export GVL_GEMM_TIMING=1
GVL_GEMM_DBGFLAGS=8 python tools/gemm_one.py 8192 8192 8192 82 1 plain > gpurun_out/tm11.log 2>&1
GVL_GEMM_DBGFLAGS=0 python tools/gemm_one.py 8192 8192 8192 82 1 plain >> gpurun_out/tm11.log 2>&1
