"""How fast can this MI355X issue bf16 MFMAs at all?  (gvl_probe_mfma: register-only kernel, every matrix pipe 100 % busy)
  python tools/mfma_probe.py   -> TFLOP/s and sustained clock for zero / constant / random operands, 1 and 2 waves per SIMD"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
import torch
from grounded_video_llm_amd import lib as L
lib = L.load()
torch.zeros(1, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for wps in (1, 2):
    for mode, name in ((0, "zero"), (1, "ones"), (2, "random"), (2, "random"), (0, "zero")):
        tf, ghz = C.c_double(0), C.c_double(0)
        rc = lib.gvl_probe_mfma(mode, wps, 200000, C.byref(tf), C.byref(ghz), st)
        print(f"waves/SIMD {wps} operands {name:6s}: rc {rc} {tf.value:8.1f} TFLOP/s = {tf.value / 2500:5.3f} of 2.5 PF; s_memtime ticks per ns {ghz.value:5.3f}; "
              f"implied clock if the pipe is 100 % busy: {tf.value * 1e12 / (1024 * 1024 * 1e9):5.2f} GHz", flush=True)
