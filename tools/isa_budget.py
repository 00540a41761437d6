"""Register / LDS / scratch budget of every kernel in libgvl.so, read from the code objects the library embeds (no GPU needed):
  python tools/isa_budget.py [path/to/libgvl.so] [substring ...]
The .so carries one clang offload bundle per translation unit; each bundle's gfx950 entry is an ELF whose AMDGPU metadata note
(llvm-readelf --notes) lists, per kernel, .vgpr_count / .agpr_count / .sgpr_count, the spill counts, LDS (.group_segment_fixed_size) and
scratch (.private_segment_fixed_size).  tests/test_isa_budget.py pins the budgets DESIGN.md quotes (occupancy classes, no spills in the hot kernels)."""
import os, re, struct, subprocess, sys, tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(so_path, arch="gfx950"):
    """The device ELFs of `arch` embedded in so_path, one per translation unit."""
    d = open(so_path, "rb").read()
    out, at = [], d.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", d, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            triple = d[p + 24: p + 24 + tl].decode()
            p += 24 + tl
            if triple.startswith("hip") and arch in triple and size:
                out.append(d[at + off: at + off + size])
        at = d.find(MAGIC, at + 1)
    return out


FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size", ".private_segment_fixed_size",
          ".max_flat_workgroup_size")


def kernels(so_path):
    """{demangled-ish kernel name: {field: int}} over every code object of the library."""
    res = {}
    for elf in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + txt)[1:]:
            blk = "  - .agpr_count:" + blk
            nm = re.search(r"\.name:\s+(\S+)", blk)
            if not nm:
                continue
            rec = {}
            for k in FIELDS:
                m = re.search(re.escape(k) + r":\s+(\d+)", blk)
                rec[k] = int(m.group(1)) if m else 0
            res[nm.group(1)] = rec
    return res


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = sys.argv[1:]
    so = args.pop(0) if args and args[0].endswith(".so") else os.path.join(here, "grounded-video-llm_amd", "libgvl.so")
    ks = kernels(so)
    dm = demangle(sorted(ks))
    print(f"# {so}: {len(ks)} kernels in {len(code_objects(so))} code objects")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6} {'lds':>7} {'scratch':>7}  kernel")
    for n in sorted(ks, key=lambda n: dm[n]):
        if args and not any(a in dm[n] for a in args):
            continue
        r = ks[n]
        print(f"{r['.vgpr_count']:5d} {r['.agpr_count']:5d} {r['.sgpr_count']:5d} {r['.vgpr_spill_count']:6d} {r['.sgpr_spill_count']:6d} "
              f"{r['.group_segment_fixed_size']:7d} {r['.private_segment_fixed_size']:7d}  {re.sub(r'^void ', '', dm[n])[:110]}")
