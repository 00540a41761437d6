cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/t10.log
python tools/attn_one.py 12 2049 16 16 88 0 5 > gpurun_out/at10.log 2>&1; python tools/attn_one.py 1 3519 32 32 96 1 5 >> gpurun_out/at10.log 2>&1; python tools/attn_one.py 12 577 16 16 64 0 5 >> gpurun_out/at10.log 2>&1; python tools/attn_one.py 1 3519 32 8 128 1 5 >> gpurun_out/at10.log 2>&1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/b10.log 2>&1
