#!/usr/bin/env python3
"""Generator of grounded-video-llm_amd/csrc/gvl_gemm4_loop.inc: the hand-placed main loop of gemm_a4_kernel (gvl_gemm4.hip).

    python tools/gen_gemm4_loop.py            # rewrites the .inc (committed; tests/test_gemm4_loop_gen.py checks it is up to date)

Why a generator: the loop is ONE inline-asm statement per output tile -- 64 v_mfma_f32_32x32x16_bf16 per k-tile and wave with every LDS fragment read, every
global->LDS DMA piece, every wait and the one barrier per k-tile placed BY HAND in the gaps between them (one wave per SIMD: nothing else covers a stall, and
hipcc neither keeps 256 accumulators in AGPRs next to 256 VGPRs of operands nor places loads into MFMA gaps).  Writing four near-identical bodies of ~180
instructions by hand invites exactly the slips (a wrong offset, a missed toggle) no test localises; the schedule below is the single source.

Structure (reference shapes: models/internvideo2.py:587,603,631-634, models/modeling_phi3.py:459-464,659-663 -- every nn.Linear of the towers / prefill):
  CU tile 256 x 256 x 64, 4 waves = 2 (m) x 2 (n), wave tile 128 x 128 = 4 x 4 blocks of 32 x 32: accumulators a[0:255], block (j = m block, i = n block) at
  a[(4 j + i) 16 ...].  LDS ring: 2 slots x 64 KiB = [W rows 0..255 | A rows 0..255] x 128 B (one k-tile), chunk-swizzled as in gvl_gemm.hip.
  k-tile t lives in slot (par + t) & 1.  A k-tile is 4 phases of 16 MFMAs (one 16-wide k step each) on fragment set X (even phases) / Y (odd phases):
    phase p   MFMA on k step p | ds_read the fragments of k step p + 1 (phase 3: k step 0 of the NEXT k-tile, from the other slot) into the other set
    phase 0   + the A pieces of the DMA of k-tile t + 1          phase 3   + the W pieces of the DMA of k-tile t + 2
    end of phase 2:  s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier   -- k-tile t + 1 has landed for everybody AND everybody has read the last of k-tile t,
                     so phase 3 may read k-tile t + 1 and refill the slot of k-tile t.  One barrier per k-tile.
  Bodies: FIRST (t = 0: srcC = 0 instead of a zeroing pass; all 16 pieces of k-tile 1), STEADY (loop), PENULT (t = nk - 2: its phase-3 DMA fetches the W half
  of the NEXT output tile's k-tile 0), LAST (t = nk - 1: the A half of that; no phase-3 reads).  nk >= 3.
  DMA piece i of an operand: buffer_load_dwordx4 ... offen lds; per-lane offset = row * pitch + swizzled chunk (rows beyond the matrix are >= num_records:
  the hardware returns zeros, nothing is clamped), running + 32 rows per piece; the k advance travels in the scalar offset.
Operands of the asm statement (see gvl_gemm4.hip): %0 rdW0 %1 rdA0 (v: LDS byte address of this lane's k-step-0 fragment chunk in slot par, W / A rows of the
wave) %2 %3 voff W / A of this tile (v) %4 %5 voff W / A of the next tile (v) %6 %7 buffer resources W / A (s[4]) %8 %9 byte step of 32 rows W / A (s)
%10 LDS byte address of slot par ^ 1 + wave * 1024 (s) %11 nk (s).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# Code placement of the hand-written stream (MI355X_MICROARCH.md, two-waves item 8): the steady loop's head sits on a 64-byte boundary (measured, same box,
# profiles/r06_gemm4_placement.txt: the pipelined kernel's InternVideo2 qkv shape 257.0 -> 249.0 us, the others 0.3 ... 0.7 %; 256-byte alignment and a 4-byte shift
# of the whole statement move nothing further).  LAB: GVL_A4_ALIGN=N overrides (0 = none), GVL_A4_SHIFT=k opens the statement with k s_nop (4 bytes each)
ALIGN_LOOP = [f".p2align {os.environ.get('GVL_A4_ALIGN', '6')}"] if os.environ.get("GVL_A4_ALIGN", "6") != "0" else []
SHIFT = ["s_nop 0"] * int(os.environ.get("GVL_A4_SHIFT", "0"))
OUT = os.path.join(HERE, "..", "grounded-video-llm_amd", "csrc", "gvl_gemm4_loop.inc")

MB, NB = 4, 4


class Frag:
    def __init__(self, base):
        self.w = [base + 4 * i for i in range(NB)]
        self.a = [base + 4 * NB + 4 * j for j in range(MB)]


X, Y = Frag(160), Frag(192)
RDW = [224, 225, 226, 227]
RDA = [228, 229, 230, 231]
RW, RA = 236, 237                      # running per-lane source offsets of the W / A pieces
S_KOFF, S_DMA, S_CNT, S_M0 = 72, 73, 76, 77
V_LO, V_HI = 160, 237
SLOT = 0x10000


def acc(j, i):
    b = (4 * j + i) * 16
    return f"a[{b}:{b + 15}]"


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]"


def mfma(j, i, f, czero):
    return f"v_mfma_f32_32x32x16_bf16 {acc(j, i)}, {vr(f.w[i])}, {vr(f.a[j])}, " + ("0" if czero else acc(j, i))


def reads(ph, dst):
    """the 8 fragment reads of k step ph (+ the slot toggles of the two address registers, each right behind its last use)"""
    out = []
    for i in range(NB):
        out.append([f"ds_read_b128 {vr(dst.w[i])}, v{RDW[ph]} offset:{i * 4096}"])
    out[-1].append(f"v_xor_b32 v{RDW[ph]}, 0x{SLOT:x}, v{RDW[ph]}")
    for j in range(MB):
        out.append([f"ds_read_b128 {vr(dst.a[j])}, v{RDA[ph]} offset:{j * 4096}"])
    out[-1].append(f"v_xor_b32 v{RDA[ph]}, 0x{SLOT:x}, v{RDA[ph]}")
    return out


def m0_for(op, i):
    return f"s_add_u32 m0, s{S_DMA}, 0x{(0x8000 if op == 'A' else 0) + i * 0x1000:x}"


def piece(op):
    r, rs, st = (RW, "%6", "%8") if op == "W" else (RA, "%7", "%9")
    return [f"buffer_load_dwordx4 v{r}, {rs}, s{S_KOFF} offen lds", f"v_add_u32 v{r}, {st}, v{r}"]


def body(kind, var):
    """one k-tile: list of asm lines"""
    L = []
    for p in range(4):
        use, nxt = (X, Y) if p % 2 == 0 else (Y, X)
        pre, gaps, post = [], [[] for _ in range(16)], []
        if p < 3:
            pre.append("s_waitcnt lgkmcnt(0)")
        # ---- fragment reads for the next k step
        if not (kind == "LAST" and p == 3):
            for g, ins in zip(var["read_gaps"], reads((p + 1) % 4, nxt)):
                gaps[g] += ins
        # ---- DMA pieces
        plan = []
        if p == 0:
            srcA = "%5" if kind == "LAST" else "%3"
            if kind == "FIRST":
                plan = [("W", i) for i in range(8)] + [("A", i) for i in range(8)]
                pre += [f"v_mov_b32 v{RW}, %2", f"v_mov_b32 v{RA}, {srcA}"]
            else:
                plan = [("A", i) for i in range(8)]
                pre += [f"v_mov_b32 v{RA}, {srcA}"]
        if p == 3 and kind != "LAST":
            plan = [("W", i) for i in range(8)]
            pre += [f"v_mov_b32 v{RW}, " + ("%4" if kind == "PENULT" else "%2")]
        if plan:
            dg = var["dma_gaps16"] if len(plan) == 16 else var["dma_gaps8"]
            pre.append(m0_for(*plan[0]))
            for k, (op, i) in enumerate(plan):
                gaps[dg[k]] += piece(op)
                if k + 1 < len(plan):
                    gaps[dg[k]].append(m0_for(*plan[k + 1]))
        if p == 1 and kind != "LAST":
            gaps[0] += [f"s_mov_b32 s{S_KOFF}, 0" if kind == "PENULT" else f"s_add_u32 s{S_KOFF}, s{S_KOFF}, 128", f"s_xor_b32 s{S_DMA}, s{S_DMA}, 0x{SLOT:x}"]
        if p == 2:
            post += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
        L += pre
        n = 0
        for j in range(MB):
            for i in range(NB):
                L.append(mfma(j, i, use, kind == "FIRST" and p == 0))
                L += gaps[n]
                n += 1
        L += post
    return L


VARIANTS = {
    0: dict(read_gaps=[0, 1, 2, 3, 4, 5, 6, 7], dma_gaps8=[8, 9, 10, 11, 12, 13, 14, 15], dma_gaps16=list(range(16))),
    1: dict(read_gaps=[0, 1, 2, 3, 4, 5, 6, 7], dma_gaps8=[0, 2, 4, 6, 8, 10, 12, 14], dma_gaps16=list(range(16))),
    2: dict(read_gaps=[1, 2, 3, 4, 5, 6, 7, 8], dma_gaps8=[0, 1, 2, 3, 4, 5, 6, 7], dma_gaps16=list(range(16))),
}


def tile_asm(var):
    L = SHIFT + [f"s_mov_b32 s{S_M0}, m0", f"s_mov_b32 s{S_KOFF}, 128", f"s_mov_b32 s{S_DMA}, %10", f"s_sub_u32 s{S_CNT}, %11, 3"]
    for regs, src in ((RDW, "%0"), (RDA, "%1")):
        L.append(f"v_mov_b32 v{regs[0]}, {src}")
        for ph in range(1, 4):
            L.append(f"v_xor_b32 v{regs[ph]}, 0x{ph << 5:x}, v{regs[0]}")
    # k-tile 0 of this tile HAS landed for everybody: phase 2 of the previous tile's LAST body waited for it in front of its barrier (the first tile of a
    # workgroup: the caller waits).  No vmcnt here -- it would wait for the epilogue operands the caller has just requested.  lgkmcnt: this wave's staging reads
    # of the previous epilogue; the barrier: everybody's, before k-tile 1 is DMA'd over the staging area.
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    for ins in reads(0, X):
        L += ins
    L += body("FIRST", var)
    L += [f"s_cmp_eq_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lgvl_a4_pen_%=", *ALIGN_LOOP, ".Lgvl_a4_loop_%=:"]
    L += body("STEADY", var)
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lgvl_a4_loop_%=", ".Lgvl_a4_pen_%=:"]
    L += body("PENULT", var)
    L += body("LAST", var)
    # the last MFMAs' results must be readable by the v_accvgpr_read of the epilogue (XDL write -> VALU read: 18 wait states for 16 passes)
    L += ["s_nop 15", "s_nop 3", f"s_mov_b32 m0, s{S_M0}"]
    return L


def dma_tile_asm():
    """all 16 pieces of ONE k-tile at scalar offset 0 (the first tile of a workgroup).  %0 %1 voff W / A (v), %2 %3 resources, %4 %5 row steps, %6 slot base + wave * 1024"""
    L = [f"s_mov_b32 s{S_M0}, m0", f"s_mov_b32 s{S_KOFF}, 0", f"v_mov_b32 v{RW}, %0", f"v_mov_b32 v{RA}, %1"]
    for op, rs, st, r in (("W", "%2", "%4", RW), ("A", "%3", "%5", RA)):
        for i in range(8):
            L += [f"s_add_u32 m0, %6, 0x{(0x8000 if op == 'A' else 0) + i * 0x1000:x}", "s_nop 0", f"buffer_load_dwordx4 v{r}, {rs}, s{S_KOFF} offen lds", f"v_add_u32 v{r}, {st}, v{r}"]
    L.append(f"s_mov_b32 m0, s{S_M0}")
    return L


def cstr(lines):
    return "\n".join(f'  "{l}\\n\\t"' for l in lines)


def render():
    o = ["// GENERATED by tools/gen_gemm4_loop.py -- do not edit; the schedule and its reasoning live there.", "#pragma once", ""]
    for v in sorted(VARIANTS):
        o += [f"#define GVL_A4_TILE_ASM_V{v} \\"]
        body_lines = tile_asm(VARIANTS[v])
        o += [f'  "{l}\\n\\t" \\' for l in body_lines[:-1]] + [f'  "{body_lines[-1]}"', ""]
    d = dma_tile_asm()
    o += ["#define GVL_A4_DMA_TILE_ASM \\"] + [f'  "{l}\\n\\t" \\' for l in d[:-1]] + [f'  "{d[-1]}"', ""]
    acl = ", ".join(f'"a{i}"' for i in range(256))
    vcl = ", ".join(f'"v{i}"' for i in range(V_LO, V_HI + 1))
    o += [f"#define GVL_A4_CLOBBER_AGPRS {acl}", f"#define GVL_A4_CLOBBER_VGPRS {vcl}",
          f'#define GVL_A4_CLOBBER_SGPRS "s{S_KOFF}", "s{S_DMA}", "s{S_CNT}", "s{S_M0}"', f"#define GVL_A4_FIRST_FREE_VGPR {V_LO}", ""]
    return "\n".join(o)


if __name__ == "__main__":
    txt = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == txt else 1)
    open(OUT, "w").write(txt)
    print(f"wrote {os.path.normpath(OUT)}: {len(txt.splitlines())} lines")
