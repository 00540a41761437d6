"""Register / scratch budgets of the shipped code objects, checked WITHOUT a GPU (tools/isa_budget.py reads the AMDGPU metadata of the ELFs libgvl.so embeds).

The occupancy classes DESIGN.md argues with are properties of the build, not of a run: a kernel that quietly starts spilling, or crosses 128 / 168 VGPRs and loses
a wave per SIMD, is a performance regression no parity test sees.  (gfx950: 512 VGPRs per SIMD lane -> <= 128 = 4 waves, <= 168 = 3, <= 256 = 2.)"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_budget  # noqa: E402

SO = os.path.join(ROOT, "grounded-video-llm_amd", "libgvl.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(SO) and os.path.exists(isa_budget.READELF)), reason="libgvl.so not built / no llvm-readelf")

# the one kernel that spills, by name and by how much (InternVideo2's fused patch embedding: 256 VGPRs + 23 spilled, 64 B of scratch)
KNOWN_SPILLS = {"_Z18patch_embed_kernelILi11ELi1EEv14PatchEmbedArgs": 23}


@pytest.fixture(scope="module")
def ks():
    k = isa_budget.kernels(SO)
    assert len(k) > 100, "metadata of the embedded code objects not found"
    return k


def test_no_kernel_spills_vector_registers_or_uses_scratch(ks):
    for name, r in ks.items():
        if name in KNOWN_SPILLS:
            assert r[".vgpr_spill_count"] <= KNOWN_SPILLS[name], (name, r)
            continue
        assert r[".vgpr_spill_count"] == 0 and r[".private_segment_fixed_size"] == 0, (name, r)


def test_decode_skinny_gemm_occupancy_classes(ks):
    """dgemm_kernel<RB, NW, U, NT, XN, W8>: 4-step load groups run 4 waves / SIMD (<= 128 VGPRs), the 8-step form of Phi-3.5's down_proj 3 (<= 168)."""
    seen = 0
    for name, r in ks.items():
        m = re.match(r"_Z12dgemm_kernelILi(\d)ELi(\d)ELi(\d)E", name)
        if not m:
            continue
        seen += 1
        u = int(m.group(3))
        assert r[".vgpr_count"] <= (168 if u >= 8 else 128), (name, r)
    assert seen >= 17


def test_pingpong_gemm_fits_two_waves_and_spills_scalars_only_where_known(ks):
    """gemm_pp_kernel<256, 256, EPI, 1>: 128 accumulator registers + staging; <= 256 VGPRs (2 waves / SIMD, 8 waves per workgroup on 4 SIMDs), no VGPR spill (above);
    SGPR spills (outside the k loop) only in the three widest epilogues."""
    seen = 0
    for name, r in ks.items():
        if not name.startswith("_Z14gemm_pp_kernel"):
            continue
        seen += 1
        assert r[".vgpr_count"] <= 256, (name, r)
        epi = re.search(r"ILi256ELi256ELi(n?\d+)E", name).group(1)
        assert r[".sgpr_spill_count"] <= {"56": 2, "184": 13, "n1": 5}.get(epi, 0), (name, r)
    assert seen >= 15


def test_attention_kernels_keep_their_wave_counts(ks):
    for name, r in ks.items():
        if name.startswith("_Z20attn_iv2_pipe_kernel"):
            assert r[".vgpr_count"] <= 256 and r[".sgpr_spill_count"] == 0, (name, r)
        m = re.match(r"_Z15attn_fwd_kernelILi(\d+)E", name)
        if m:                                       # head dim 64: 4 waves / SIMD; 96: 3; 128: 2
            assert r[".vgpr_count"] <= {64: 128, 96: 168, 128: 256}[int(m.group(1))], (name, r)
