#!/bin/bash
# full GPU validation: the -m gpu suite (summary + failures), smoke(), a short bench
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -40 > $O/gpu_tests.txt
python __graft_entry__.py smoke 2>&1 | grep graft > $O/smoke.txt
python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -3 > $O/bench_short.txt
tail -15 $O/gpu_tests.txt; cat $O/smoke.txt; cat $O/bench_short.txt
