"""CPU checks of the two hand-placed GEMM loops (no GPU): the committed .inc files are what the generators produce, and the built code objects keep the contract the
inline-asm statements rely on -- gemm_a4_kernel's accumulators live in a[0:255] ACROSS statements the compiler knows nothing about, so a compiler-inserted
v_accvgpr_write (an AGPR spill) anywhere in that kernel, or any scratch use, would silently corrupt a tile (cdna_hip_programming.md 5.7 item 4)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_budget  # noqa: E402

SO = os.path.join(ROOT, "grounded-video-llm_amd", "libgvl.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.parametrize("gen", ["gen_gemm4_loop.py", "gen_gemm4p.py"])
def test_generated_loops_are_up_to_date(gen):
    env = {k: v for k, v in os.environ.items() if not k.startswith("GVL_A4P_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--check"], env=env)
    assert r.returncode == 0, f"grounded-video-llm_amd/csrc/*.inc is stale: run python tools/{gen}"


@pytest.fixture(scope="module")
def disasm():
    if not (os.path.exists(SO) and os.path.exists(OBJDUMP)):
        pytest.skip("libgvl.so not built / no llvm-objdump")
    out = {}
    import tempfile
    for elf in isa_budget.code_objects(SO):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(\w+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", txt, re.S | re.M):
            if "gemm_a4" in m.group(1):
                out[m.group(1)] = m.group(2)
    assert out, "no 4-wave GEMM kernel found in libgvl.so"
    return out


def test_four_wave_kernels_keep_the_accumulator_file_to_the_asm(disasm):
    ks = isa_budget.kernels(SO)
    seen = {"a4": 0, "a4p": 0}
    for name, body in disasm.items():
        if name.endswith(".kd"):
            continue
        r = ks[name]
        assert r[".vgpr_spill_count"] == 0 and r[".private_segment_fixed_size"] == 0, (name, r)
        assert r[".agpr_count"] == 256, (name, r)
        reads, writes = body.count("v_accvgpr_read_b32"), body.count("v_accvgpr_write")
        assert writes == 0, f"{name}: {writes} compiler-inserted v_accvgpr_write"
        mfma = body.count("v_mfma_f32_32x32x16_bf16")
        if "gemm_a4p" in name:
            seen["a4p"] += 1
            # first-tile statement + pipelined statement: a drain of 256 each; the epilogues with NARROW statements (N = 1408's half column tile: 128 x 64 per wave)
            # carry two more drains of 128 and bodies of 32 MFMAs
            narrow = "Li184E" in name
            assert reads == (768 if narrow else 512), f"{name}: {reads} v_accvgpr_read"
            assert mfma % (32 if narrow else 64) == 0 and mfma >= 2 * 4 * 64, (name, mfma)
        else:
            seen["a4"] += 1
            assert reads == 256, f"{name}: {reads} v_accvgpr_read (one epilogue of 256 expected)"
            assert mfma == 4 * 64, (name, mfma)                                                         # FIRST, STEADY, PENULT, LAST bodies
    assert seen["a4"] >= 13 and seen["a4p"] >= 8, seen
