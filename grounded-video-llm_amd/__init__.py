"""grounded_video_llm_amd -- MI355X-native Grounded-VideoLLM inference hot path.

Importable as `grounded_video_llm_amd` (see /_gvl_bootstrap.py: the directory name carries a
hyphen).  The heavy parts live in `csrc/` (HIP kernels + the C ABI of include/gvl.h, built into
libgvl.so); Python here is the host-side mirror of the reference's `inference.py` /
`LLAVA_NEXT_VIDEO.generate()` surface.
"""
__version__ = "0.1.0"
