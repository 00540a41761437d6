"""Compare the emitted instruction streams of kernels between two `hipcc -S --cuda-device-only` listings (labels normalised, comments and directives dropped).

usage: python tools/asm_diff.py <old.s> <new.s> <kernel-name-substring> [old-suffix=new-suffix]
Used when a template gains a parameter or a struct a field: the instantiations that must NOT change (the hot ones) are shown to be instruction-identical
without a GPU.  Listing recipe: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S csrc/<file>.hip -o <out>.s"""
import re
import sys


def kernels(path, key):
    out, cur, buf = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m and key in m.group(1):
            cur, buf = m.group(1), []
            continue
        if cur:
            if "s_endpgm" in line:
                out[cur] = buf
                cur = None
            else:
                t = re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r";.*", "", line).strip())
                if t and not t.startswith("."):
                    buf.append(t)
    return out


def main(old, new, key, rename=None):
    a, b = kernels(old, key), kernels(new, key)
    src, dst = rename.split("=") if rename else ("", "")
    same = True
    for k in sorted(a):
        k2 = k.replace(src, dst) if rename else k
        if k2 not in b:
            print(f"{k}: not in {new}")
            same = False
            continue
        d = [(x, y) for x, y in zip(a[k], b[k2]) if x != y]
        ok = not d and len(a[k]) == len(b[k2])
        same = same and ok
        print(f"{k2}: {len(a[k])} -> {len(b[k2])} instructions, {'IDENTICAL' if ok else f'{len(d)} differing lines, first: {d[:2]}'}")
    for k in sorted(b):
        if (k.replace(dst, src) if rename else k) not in a:
            print(f"{k}: new ({len(b[k])} instructions)")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:5]))
