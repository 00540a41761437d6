"""LAB: does the decode projection stream faster when its weights are already in the 256 MB Infinity Cache (MALL)?  rounds = 1 repeats ONE weight copy
(resident after the first launch when it fits), rounds = 16 cycles > 1 GB of distinct copies (HBM)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

eng = E.Engine(E.TowerGeometry(max_segs=1), "cuda:0", towers=())
for name, N, K in [("phi.o", 3072, 3072), ("phi.down", 3072, 8192), ("phi.qkv", 9216, 3072), ("phi.gate_up", 16384, 3072), ("phi.lm_head", 32366, 3072)]:
    for B in (1, 8, 16):
        row = [f"{name:12s} B={B:2d} {N * K * 2 / 1e6:6.1f} MB"]
        for rounds in (1, 2, 16):
            us = C.c_double(0)
            rc = eng.lib.gvl_op_decode_bench(eng.ctx, N, K, B, 0, 0, rounds, 300, C.byref(us), eng.stream)
            row.append(f"rounds {rounds:2d}: {us.value:6.2f} us {N * K * 2 / us.value / 1e3:6.0f} GB/s" if rc == 0 else "n/a")
        print(" | ".join(row), flush=True)
