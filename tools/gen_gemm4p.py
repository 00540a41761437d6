#!/usr/bin/env python3
"""Generator of grounded-video-llm_amd/csrc/gvl_gemm4p_loop.inc: the main loop of gemm_a4p_kernel (gvl_gemm4p.hip) -- the 4-wave GEMM whose EPILOGUE IS
SOFTWARE-PIPELINED ACROSS OUTPUT TILES.

    python tools/gen_gemm4p.py            # rewrites the .inc (committed; tests/test_gemm4_loop_gen.py checks it is up to date)

Why.  With one wave per SIMD nothing overlaps a tile's epilogue with matrix work: measured (profiles/r06_gemm4_anatomy.txt) 6 k ... 21 k cycles per tile beside a 47 k-cycle
K = 1408 main loop, i.e. 11 ... 31 % of InternVideo2's GEMM time with the matrix pipe idle (the 8-wave kernel: 4 k ... 13.7 k); skipping the epilogue altogether is worth
+8 ... +44 % per shape (profiles/r06_gemm4_noepi_bound.txt).  The accumulators cannot stay (the next tile needs all 256 AGPRs), so the epilogue is split:
  DRAIN   (exposed, ~0.5 k instructions): a[0:255] -> (x row scale, + bias) -> bf16 pairs P in 128 VGPRs.  Every fused epilogue of the library starts by rounding
          acc (* rs) (+ bias) to bf16, so P holds exactly the values the rest of the epilogue is defined on.
  DEFER   the rest -- activation / LayerScale in place on P, transposition through a private LDS staging area, residual add, row sums of squares, whole-row stores --
          runs INSIDE THE NEXT TILE'S MAIN LOOP, as fillers in the gaps between its MFMAs (<= CAP instructions per gap, program order kept).  A flush statement
          runs the last tile's program alone.
The main loop itself is the one of tools/gen_gemm4_loop.py (schedule variant 1): see there for the ring / phase / barrier protocol.  Differences: operands are
fixed physical registers (three "{v[0:15]}" / "{s[36:51]}" / "{s[52:67]}" operand blocks instead of 12 scalars: an asm statement takes 30 operands, the pipelined
form needs more), and the first U k-tiles behind FIRST are unrolled to carry the deferred program.

Synchronisation of the deferred program (everything else is plain in-order issue):
  * LDS reads (table, gamma, staging read-back) are consumed in a LATER PHASE than they were issued in: every phase of the main loop opens with s_waitcnt lgkmcnt(0)
    (marker PH in the program).  A wave's own ds_write -> ds_read of the staging area needs nothing: LDS operations of one wave execute in order.
  * global loads (residual) are consumed behind the NEXT end-of-phase-2 s_waitcnt vmcnt(0) of the main loop (marker KT).  Stores are never waited for.
Registers: v0-15 per-lane parameters, v16-31 the compiler's, v32-159 P, v160-233 the main loop, v234-255 the deferred program / drain; s36-67 parameters.
The C++ between two tile statements must not touch v32-v255 / a0-a255 (P and the accumulators live there across statements): tests/test_isa_budget.py audits the ISA.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# Code placement of the hand-written stream (MI355X_MICROARCH.md, two-waves item 8): the steady loop's head sits on a 64-byte boundary (measured, same box,
# profiles/r06_gemm4_placement.txt: the pipelined kernel's InternVideo2 qkv shape 257.0 -> 249.0 us, the others 0.3 ... 0.7 %; 256-byte alignment and a 4-byte shift
# of the whole statement move nothing further).  LAB: GVL_A4_ALIGN=N overrides (0 = none), GVL_A4_SHIFT=k opens the statement with k s_nop (4 bytes each)
ALIGN_LOOP = [f".p2align {os.environ.get('GVL_A4_ALIGN', '6')}"] if os.environ.get("GVL_A4_ALIGN", "6") != "0" else []
SHIFT = ["s_nop 0"] * int(os.environ.get("GVL_A4_SHIFT", "0"))
OUT = os.path.join(HERE, "..", "grounded-video-llm_amd", "csrc", "gvl_gemm4p_loop.inc")

MB, NB = 4, 4
SLOT = 0x10000

# ---- register map ---------------------------------------------------------------------------------------------------------------------------------------
# per-lane parameter block v[0:15]
V_RDW0, V_RDA0, V_VOW, V_VOA, V_VOWN, V_VOAN = 0, 1, 2, 3, 4, 5
V_VOC = [6, 7]          # output offsets of the PREVIOUS tile, column half 0 / 1 (0x80000000: columns beyond N)
V_VOR = [8, 9]          # residual offsets, same
V_VOQ = [10, 11]        # row-statistics offsets, same (valid on lanes & 7 == 0 only)
V_VOB = 11              # bias offset of THIS tile: (column of lane, + 64 for the second piece) * 4; the bias slice is DMA'd into the (then idle) staging area
V_RS = 12               # v12..v15: row scale of THIS tile's rows l31 + 32 j (drain)
P0 = 32                 # P[j][i][r] = v[P0 + (4 j + i) 8 + r]: r = 2 b + q holds columns 8 b + 4 h + 2 q, + 1 of block (j, i), row l31


class Frag:
    def __init__(self, base):
        self.w = [base + 4 * i for i in range(NB)]
        self.a = [base + 4 * NB + 4 * j for j in range(MB)]


X, Y = Frag(160), Frag(192)
RDW = [224, 225, 226, 227]
RDA = [228, 229, 230, 231]
RW, RA = 232, 233
V_WADDR, V_RADDR, V_BADDR, V_GADDR = 234, 235, 236, 237     # staging write base (row l31), staging read-back base (row lane >> 3), bias / gamma scratch + 16 h
T0 = 238                                                     # v238..v255: 18 temporaries of the deferred program / drain
NT = 18
V_LO, V_HI = 32, 255
# scalar parameter blocks
S_RSW, S_RSA, S_STEPW, S_STEPA, S_DMAO, S_NK, S_LBIAS, S_LGAMMA, S_LTAB, S_LSTG = 36, 40, 44, 45, 46, 47, 48, 49, 50, 51
S_RSC, S_RSR, S_RSQ, S_LDC8, S_LDR8, S_LDQ8, S_GLO = 52, 56, 60, 64, 65, 66, 67
S_RSB = 68              # s[68:71]: buffer resource over the bias vector
S_KOFF, S_DMA, S_CNT, S_M0, S_OC, S_OR, S_OQ = 72, 73, 76, 77, 78, 79, 80
S_WORK = [S_KOFF, S_DMA, S_CNT, S_M0, S_OC, S_OR, S_OQ]


def acc(j, i):
    b = (4 * j + i) * 16
    return f"a[{b}:{b + 15}]"


def vr(b, n=4):
    return f"v[{b}:{b + n - 1}]"


def sr(b, n=4):
    return f"s[{b}:{b + n - 1}]"


def preg(j, i, r):
    return P0 + (4 * j + i) * 8 + r


# ---- main loop (schedule variant 1 of gen_gemm4_loop.py, physical operands) -----------------------------------------------------------------------------
READ_GAPS = [0, 1, 2, 3, 4, 5, 6, 7]
DMA_GAPS8 = [int(x) for x in os.environ.get("GVL_A4P_DMAGAPS", "0,2,4,6,8,10,12,14").split(",")]     # LAB override: which MFMA gaps of a phase carry its 8 DMA pieces
DMA_GAPS16 = list(range(16))


def mfma(j, i, f, czero):
    return f"v_mfma_f32_32x32x16_bf16 {acc(j, i)}, {vr(f.w[i])}, {vr(f.a[j])}, " + ("0" if czero else acc(j, i))


def reads(ph, dst, nb=NB):
    out = []
    for i in range(nb):
        out.append([f"ds_read_b128 {vr(dst.w[i])}, v{RDW[ph]} offset:{i * 4096}"])
    out[-1].append(f"v_xor_b32 v{RDW[ph]}, 0x{SLOT:x}, v{RDW[ph]}")
    for j in range(MB):
        out.append([f"ds_read_b128 {vr(dst.a[j])}, v{RDA[ph]} offset:{j * 4096}"])
    out[-1].append(f"v_xor_b32 v{RDA[ph]}, 0x{SLOT:x}, v{RDA[ph]}")
    return out


def m0_for(op, i):
    return f"s_add_u32 m0, s{S_DMA}, 0x{(0x8000 if op == 'A' else 0) + i * 0x1000:x}"


def piece(op):
    r, rs, st = (RW, sr(S_RSW), f"s{S_STEPW}") if op == "W" else (RA, sr(S_RSA), f"s{S_STEPA}")
    return [f"buffer_load_dwordx4 v{r}, {rs}, s{S_KOFF} offen lds", f"v_add_u32 v{r}, {st}, v{r}"]


class Phase:
    def __init__(self, ngaps=16, cap=None):
        self.pre, self.gaps, self.post, self.mf = [], [[] for _ in range(ngaps)], [], []
        self.cap = cap               # instructions per gap (None: the module's CAP)

    def lines(self):
        L = list(self.pre)
        for n in range(len(self.gaps)):
            L.append(self.mf[n])
            L += self.gaps[n]
        return L + self.post


DMA_ORDER = os.environ.get("GVL_A4P_DMAORDER", "WA")      # LAB: "AW" = the A pieces in phase 3, the W pieces in phase 0 of the next k-tile
NARROW_CAP = int(os.environ.get("GVL_A4P_NCAP", "9"))     # instructions per MFMA gap of a NARROW body (half the gaps per k-tile carry the same deferred program)


def body(kind, epi=0, nb=NB):
    """one k-tile as 4 Phase objects.  nb = 4: the 128 x 128 wave tile (16 MFMAs per phase).  nb = 2 (NARROW, a tile with <= 128 valid columns -- N = 1408's last
    column tile): the wave tile is 128 rows x 64 columns (wave (wm, wn) -> columns 64 wn ...), 8 MFMAs per phase, two W fragments per k step, and the steady k-tiles
    fetch only the 4 W pieces that hold real rows; the prefetch of the NEXT tile's first k-tile (PENULT / LAST) stays complete -- that tile may be a wide one."""
    phs = []
    ng = MB * nb
    narrow = nb != NB
    for p in range(4):
        ph = Phase(ng, NARROW_CAP if narrow else None)
        use, nxt = (X, Y) if p % 2 == 0 else (Y, X)
        if p < 3:
            ph.pre.append("s_waitcnt lgkmcnt(0)")
        if not (kind == "LAST" and p == 3):
            for g, ins in zip(READ_GAPS, reads((p + 1) % 4, nxt, nb)):
                ph.gaps[g] += ins
        plan = []
        nw_own = 4 if narrow else 8             # W pieces of THIS tile's k-tiles
        # which operand's pieces go first (phase 3 of k-tile t: a full k-tile to land) and which second (phase 0 of k-tile t + 1: three phases): DMA_ORDER
        a_first = DMA_ORDER == "AW"
        if p == 0:
            if kind == "FIRST":
                plan = [("W", i) for i in range(nw_own)] + [("A", i) for i in range(8)]
                ph.pre += [f"v_mov_b32 v{RW}, v{V_VOW}", f"v_mov_b32 v{RA}, v{V_VOA}"]
            elif DMA_ORDER == "P3":
                plan = []
            elif a_first:
                plan = [("W", i) for i in range(8 if kind == "LAST" else nw_own)]
                ph.pre += [f"v_mov_b32 v{RW}, " + (f"v{V_VOWN}" if kind == "LAST" else f"v{V_VOW}")]
            else:
                plan = [("A", i) for i in range(8)]
                ph.pre += [f"v_mov_b32 v{RA}, " + (f"v{V_VOAN}" if kind == "LAST" else f"v{V_VOA}")]
        if p == 3 and kind != "LAST":
            if DMA_ORDER == "P3":           # LAB: both operands in phase 3 (a full k-tile to land for both)
                plan = [("W", i) for i in range(8 if kind == "PENULT" else nw_own)] + [("A", i) for i in range(8)]
                ph.pre += [f"v_mov_b32 v{RW}, " + (f"v{V_VOWN}" if kind == "PENULT" else f"v{V_VOW}"), f"v_mov_b32 v{RA}, " + (f"v{V_VOAN}" if kind == "PENULT" else f"v{V_VOA}")]
            elif a_first:
                plan = [("A", i) for i in range(8)]
                ph.pre += [f"v_mov_b32 v{RA}, " + (f"v{V_VOAN}" if kind == "PENULT" else f"v{V_VOA}")]
            else:
                plan = [("W", i) for i in range(8 if kind == "PENULT" else nw_own)]
                ph.pre += [f"v_mov_b32 v{RW}, " + (f"v{V_VOWN}" if kind == "PENULT" else f"v{V_VOW}")]
        if plan:
            if narrow:
                dg = [k * ng // len(plan) for k in range(len(plan))]        # spread over the 8 gaps (12 pieces: two in every other gap)
            else:
                dg = DMA_GAPS16 if len(plan) == 16 else DMA_GAPS8
            ph.pre.append(m0_for(*plan[0]))
            for k, (op, i) in enumerate(plan):
                ph.gaps[dg[k]] += piece(op)
                if k + 1 < len(plan):
                    ph.gaps[dg[k]].append(m0_for(*plan[k + 1]))
                    if dg[k + 1] == dg[k]:
                        ph.gaps[dg[k]].append("s_nop 0")       # M0 write -> LDS-DMA issue needs a wait state when no MFMA separates them
        if p == 1 and kind == "LAST" and (epi & 32):
            # the bias slice of this tile (128 floats of the wave's columns) -> the wave's staging area, idle since the deferred program ended; waited for by the
            # end-of-phase-2 vmcnt(0) below, read by the drain.  No DMA piece of the loop is pending in this phase: M0 is free.
            g0 = ng // 2
            ph.gaps[g0] += [f"s_mov_b32 m0, s{S_LSTG}"]
            ph.gaps[g0 + 1] += [f"buffer_load_dword v{V_VOB}, {sr(S_RSB)}, 0 offen lds"]
            ph.gaps[g0 + 2] += [f"buffer_load_dword v{V_VOB}, {sr(S_RSB)}, 0 offen offset:256 lds"]
        if p == 1 and kind != "LAST":
            ph.gaps[0] += [f"s_mov_b32 s{S_KOFF}, 0" if kind == "PENULT" else f"s_add_u32 s{S_KOFF}, s{S_KOFF}, 128", f"s_xor_b32 s{S_DMA}, s{S_DMA}, 0x{SLOT:x}"]
        if p == 2:
            ph.post += ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]
        for j in range(MB):
            for i in range(nb):
                ph.mf.append(mfma(j, i, use, kind == "FIRST" and p == 0))
        # LAB: GVL_A4P_LAB=synthK[d] -- K synthetic VALU fillers in EVERY gap of every body (d: one dependent chain instead of independent registers):
        # what a gap can really absorb (profiles/r06_gemm4p_filler_capacity.txt)
        lab = os.environ.get("GVL_A4P_LAB", "")
        if "synth" in lab:
            k = int(lab[lab.index("synth") + 5])
            dep = lab[lab.index("synth") + 6:lab.index("synth") + 7] == "d"
            for g in range(ng):
                for x in range(k):
                    r = T0 + (0 if dep else (g * k + x) % 16)
                    ph.gaps[g].append(f"v_add_u32 v{r}, v{r}, v{r}")
        phs.append(ph)
    return phs


# ---- the deferred program ---------------------------------------------------------------------------------------------------------------------------------
# items: ("i", text) instruction | ("VM", text) a vector-memory instruction: PHASE 3 ONLY | ("PH",) later-phase marker | ("KT",) next-k-tile marker.
# Why phase 3 only (measured, profiles/r06_gemm4p_lab.txt): the main loop ends phase 2 of every k-tile with s_waitcnt vmcnt(0) (the DMA of the next k-tile must have
# landed); a store still in flight there stalls the whole workgroup.  Issued right behind that wait a store has a full k-tile (~2 100 cycles) to retire: stores in
# phases 3 / 0 / 1 cost 8 % on InternVideo2's qkv shape, in phase 3 only < 1 %.
class Prog:
    def __init__(self):
        self.items = []

    def i(self, text):
        self.items.append(("i", text))

    def vm(self, text):
        self.items.append(("VM", text))

    def ph(self):
        if self.items and self.items[-1][0] not in ("PH", "KT"):
            self.items.append(("PH",))

    def kt(self):
        if self.items and self.items[-1][0] == "PH":
            self.items.pop()
        if not (self.items and self.items[-1][0] == "KT"):
            self.items.append(("KT",))


def epi_flags(epi):
    return dict(act=epi & 3, resid=bool(epi & 8), gamma=bool(epi & 16), bias=bool(epi & 32), rs=bool(epi & 64), sq=bool(epi & 128))


TSV = T0                  # v238..v253: 16 registers -- the read-back of one pass (residual epilogues) / scratch of the elementwise stage
TX = [T0 + 16, T0 + 17]   # two scratch registers


def unpack_add_pack(pr, dst, sv, r, tA, tB):
    """dst = pack2bf(lo_bf(r) + lo_bf(sv), hi_bf(r) + hi_bf(sv)) -- the bf16 residual add of the staged epilogue (gvl_gemm_epi.h), dword by dword"""
    pr.i(f"v_lshlrev_b32 v{tA}, 16, v{sv}")
    pr.i(f"v_lshlrev_b32 v{tB}, 16, v{r}")
    pr.i(f"v_add_f32 v{tA}, v{tB}, v{tA}")
    pr.i(f"v_and_b32 v{tB}, 0xffff0000, v{sv}")
    pr.i(f"v_and_b32 v{r}, 0xffff0000, v{r}")
    pr.i(f"v_add_f32 v{tB}, v{r}, v{tB}")
    pr.i(f"v_cvt_pk_bf16_f32 v{dst}, v{tA}, v{tB}")


STORES = os.environ.get("GVL_A4P_STORES", "end")     # where a tile's output stores go: "end" of the next tile's statement / "p3" = phase 3 of its early k-tiles
V_L7 = V_GADDR            # after the elementwise stage: lane & 7 (which pass a lane keeps the row statistic of)
V_SQ = V_RS               # v12..v15: the row statistics of piece k of EVERY pass -- lane L keeps the value of pass L & 7 (no epilogue has row scale AND statistics)


def deferred(epi):
    """The deferred program of one wave tile held in P (bf16 pairs of acc (* rs) (+ bias)), as a Prog.  It ends with the FINAL output rows in P -- pass q = (j, half)
    in its own 16 registers, piece k (rows 8 k + (lane >> 3), 16 bytes at column 8 (lane & 7) of the half) in registers 4 k .. 4 k + 3 -- and the row statistics in
    V_SQ; the stores are a separate section (end_stores): they run at the END of the tile statement, where no DMA wait is near."""
    f = epi_flags(epi)
    pr = Prog()
    assert f["act"] in (0, 2, 3), "quick-GELU: not generated (CLIP's fc1 stays on the 8-wave kernel)"
    assert not (f["sq"] and f["rs"])
    silu = f["act"] == 3
    assert not (silu and (f["resid"] or f["sq"] or f["gamma"]))
    # SwiGLU: a block row is ONE pass -- 64 output columns = 128-byte rows
    passes = [(j, 0) for j in range(MB)] if silu else [(j, half) for j in range(MB) for half in range(2)]
    n = len(passes)

    def pregs(q):                       # the 16 P registers of pass q: blocks (j, 2 half), (j, 2 half + 1)
        j, half = passes[q]
        return preg(j, 2 * half, 0)

    if f["act"] == 2:
        # erf-GELU by table on bf16 inputs (gvl_gemm_epi.h: gelu_tab_addr / TABLE path): p <- bf16(x * Phi(x)) for both halves of every dword of P.
        # address = tab - 4 GELU_LO + 4 clamp(|x| bits, LO, HI) + (sign << 13); 8 dwords (16 lookups in TSV) per batch, products one phase later
        allp = [preg(j, i, r) for j in range(MB) for i in range(NB) for r in range(8)]
        for b0 in range(0, len(allp), 8):
            for n_, p in enumerate(allp[b0:b0 + 8]):
                k0, k1, g = TSV + 2 * n_, TSV + 2 * n_ + 1, TX[n_ & 1]
                pr.i(f"v_and_b32 v{k0}, 0x7fff, v{p}")
                pr.i(f"v_bfe_u32 v{k1}, v{p}, 16, 15")
                pr.i(f"v_max_u32 v{k0}, 0x3980, v{k0}")
                pr.i(f"v_max_u32 v{k1}, 0x3980, v{k1}")
                pr.i(f"v_min_u32 v{k0}, 0x40b0, v{k0}")
                pr.i(f"v_min_u32 v{k1}, 0x40b0, v{k1}")
                pr.i(f"v_lshl_add_u32 v{k0}, v{k0}, 2, s{S_LTAB}")
                pr.i(f"v_lshl_add_u32 v{k1}, v{k1}, 2, s{S_LTAB}")
                pr.i(f"v_bfe_u32 v{g}, v{p}, 15, 1")
                pr.i(f"v_lshl_add_u32 v{k0}, v{g}, 13, v{k0}")
                pr.i(f"v_lshrrev_b32 v{g}, 31, v{p}")
                pr.i(f"v_lshl_add_u32 v{k1}, v{g}, 13, v{k1}")
                pr.i(f"ds_read_b32 v{k0}, v{k0}")
                pr.i(f"ds_read_b32 v{k1}, v{k1}")
            pr.ph()
            for n_, p in enumerate(allp[b0:b0 + 8]):
                k0, k1 = TSV + 2 * n_, TSV + 2 * n_ + 1
                pr.i(f"v_lshlrev_b32 v{TX[0]}, 16, v{p}")
                pr.i(f"v_and_b32 v{TX[1]}, 0xffff0000, v{p}")
                pr.i(f"v_mul_f32 v{k0}, v{TX[0]}, v{k0}")
                pr.i(f"v_mul_f32 v{k1}, v{TX[1]}, v{k1}")
                pr.i(f"v_cvt_pk_bf16_f32 v{p}, v{k0}, v{k1}")
    if silu:
        # SwiGLU on interleaved (gate, up) bf16 pairs: o = up * bf16(gate * sigmoid(gate)), sigmoid = rcp(1 + exp2(-log2(e) gate)) -- the instruction sequence hipcc
        # emits for the staged epilogue (same roundings).  Two dwords (the two chains interleave: a transcendental result is never read by the next instruction)
        # -> one packed dword: block (j, i) shrinks from 8 to 4 registers, P[j][i][b] = outputs 16 i + 4 b + 2 h, + 1
        for j in range(MB):
            for i in range(NB):
                for b in range(4):
                    pa, pb = preg(j, i, 2 * b), preg(j, i, 2 * b + 1)
                    A = [TSV, TSV + 1, TSV + 2]
                    Bt = [TSV + 4, TSV + 5, TSV + 6]
                    for step in range(10):
                        for (p, t) in ((pa, A), (pb, Bt)):
                            g, x, u_ = t
                            pr.i([f"v_lshlrev_b32 v{g}, 16, v{p}",
                                  f"v_mul_f32 v{x}, 0xbfb8aa3b, v{g}",
                                  f"v_exp_f32 v{x}, v{x}",
                                  f"v_add_f32 v{x}, 1.0, v{x}",
                                  f"v_rcp_f32 v{x}, v{x}",
                                  f"v_mul_f32 v{x}, v{x}, v{g}",
                                  f"v_cvt_pk_bf16_f32 v{x}, v{x}, v{x}",
                                  f"v_lshlrev_b32 v{x}, 16, v{x}",
                                  f"v_and_b32 v{u_}, 0xffff0000, v{p}",
                                  f"v_mul_f32 v{x}, v{u_}, v{x}"][step])
                    pr.i(f"v_cvt_pk_bf16_f32 v{preg(j, i, b)}, v{A[1]}, v{Bt[1]}")

    # ---- elementwise stage, in place on P, before any transposition (TSV is free then) ----
    if f["gamma"]:
        # LayerScale: p <- bf16(bf16(v) * gamma[col]); gamma of columns 32 i + 8 b + 4 h ... + 3 from the wave's scratch (v_gaddr = scratch + 16 h)
        groups = [(i, b) for i in range(NB) for b in range(4)]
        for g0 in range(0, 16, 4):
            for gi, (i, b) in enumerate(groups[g0:g0 + 4]):
                pr.i(f"ds_read_b128 {vr(TSV + 4 * gi)}, v{V_GADDR} offset:{(i * 32 + 8 * b) * 4}")
            pr.ph()
            for gi, (i, b) in enumerate(groups[g0:g0 + 4]):
                for j in range(MB):
                    for q in range(2):
                        p = preg(j, i, 2 * b + q)
                        pr.i(f"v_lshlrev_b32 v{TX[0]}, 16, v{p}")
                        pr.i(f"v_and_b32 v{TX[1]}, 0xffff0000, v{p}")
                        pr.i(f"v_mul_f32 v{TX[0]}, v{TX[0]}, v{TSV + 4 * gi + 2 * q}")
                        pr.i(f"v_mul_f32 v{TX[1]}, v{TX[1]}, v{TSV + 4 * gi + 2 * q + 1}")
                        pr.i(f"v_cvt_pk_bf16_f32 v{p}, v{TX[0]}, v{TX[1]}")
            pr.ph()
    if f["sq"]:
        pr.i(f"v_mbcnt_lo_u32_b32 v{TX[0]}, -1, 0")
        pr.i(f"v_mbcnt_hi_u32_b32 v{TX[0]}, -1, v{TX[0]}")
        pr.i(f"v_and_b32 v{V_L7}, 7, v{TX[0]}")

    # ---- transposition passes ----
    nw = [0]

    def s1a(q):                         # P(q) -> staging: rows l31, 128 B = the 64 columns of blocks 2 half, 2 half + 1; 16-byte chunk c of row r at chunk c ^ (r & 7)
        j, half = passes[q]
        if silu:                        # 4 bytes at byte column 32 i + 8 b + 4 h of row l31 (v_waddr carries 4 h): chunk 2 i + (b >> 1), + 8 (b & 1)
            for i in range(NB):
                for b in range(4):
                    t = TX[nw[0] & 1]
                    nw[0] += 1
                    pr.i(f"v_xor_b32 v{t}, 0x{(2 * i + (b >> 1)) << 4:x}, v{V_WADDR}")
                    pr.i(f"ds_write_b32 v{t}, v{preg(j, i, b)}" + (" offset:8" if b & 1 else ""))
            return
        for ii in range(2):
            for b in range(4):
                t = TX[nw[0] & 1]
                nw[0] += 1
                pr.i(f"v_xor_b32 v{t}, 0x{(ii * 4 + b) << 4:x}, v{V_WADDR}")
                pr.i(f"ds_write_b64 v{t}, {vr(preg(j, 2 * half + ii, 2 * b), 2)}")

    def s1b(q, dst):                    # staging -> dst[0:15]: lane L holds row 8 k + (L >> 3), columns 8 (L & 7) ... + 7 of the half
        for k in range(4):
            pr.i(f"ds_read_b128 {vr(dst + 4 * k)}, v{V_RADDR} offset:{k * 1024}")

    def s2(q):                          # residual pieces of pass q -> the P registers the pass has just left
        j, half = passes[q]
        for k in range(4):
            pr.i(f"s_mul_i32 s{S_OR}, s{S_LDR8}, {4 * j + k}")
            pr.vm(f"buffer_load_dwordx4 {vr(pregs(q) + 4 * k)}, v{V_VOR[half]}, {sr(S_RSR)}, s{S_OR} offen")

    def s3(q, sv):                      # residual add (result -> P(q): the residual registers it consumes) and / or row statistics of the read-back pieces in sv
        for k in range(4):
            if f["resid"]:
                for e in range(4):
                    unpack_add_pack(pr, pregs(q) + 4 * k + e, sv + 4 * k + e, pregs(q) + 4 * k + e, TX[0], TX[1])
            if f["sq"]:
                # sum of squares of the 8 bf16 of the piece, then over the 8 lanes of the aligned 64-column block, in the fixed order of stg_sumsq8
                t, o = TX[0], pregs(q) + 4 * k
                pr.i(f"v_mov_b32 v{t}, 0")
                for e in range(4):
                    pr.i(f"v_dot2c_f32_bf16 v{t}, v{o + e}, v{o + e}")
                # gfx940+: a DOT result read by a different VALU opcode needs 3 wait states (not interlocked: GCNHazardRecognizer DotWriteDifferentVALURead);
                # a VALU result read through DPP needs 2
                pr.i("s_nop 3")
                pr.i(f"v_add_f32_dpp v{t}, v{t}, v{t} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
                pr.i("s_nop 1")
                pr.i(f"v_add_f32_dpp v{t}, v{t}, v{t} quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
                pr.i("s_nop 1")
                pr.i(f"v_add_f32_dpp v{TX[1]}, v{t}, v{t} row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
                # all 8 lanes of the block hold the sum: lane (L & 7) == q keeps it
                pr.i(f"v_cmp_eq_u32 vcc, {q}, v{V_L7}")
                pr.i(f"v_cndmask_b32 v{V_SQ + k}, v{V_SQ + k}, v{TX[1]}, vcc")

    if f["resid"]:
        # cycle c (one k-tile each):  phase 3:  read-back(c - 1) -> TSV ; P(c) -> staging ; residual loads(c) -> P(c)'s registers
        #                             phases 0-2:  P(c - 1) <- TSV + residual(c - 1) (loaded a full k-tile earlier: behind the end-of-phase-2 vmcnt(0)), row statistics
        for c in range(n + 1):
            pr.kt()
            if 0 <= c - 1 < n:
                s1b(c - 1, TSV)
            if c < n:
                s1a(c)
                s2(c)
            pr.ph()
            if 0 <= c - 1 < n:
                s3(c - 1, TSV)
    elif STORES == "p3":
        # stores of pass q from its own registers, in the phase 3 behind its read-back; two passes per k-tile
        def s4(q):
            j, half = passes[q]
            for k in range(4):
                pr.i(f"s_mul_i32 s{S_OC}, s{S_LDC8}, {4 * j + k}")
                pr.vm(f"buffer_store_dwordx4 {vr(pregs(q) + 4 * k)}, v{V_VOC[half]}, {sr(S_RSC)}, s{S_OC} offen")
        B = 2
        for c0 in range(0, n + B, B):
            for q in range(c0 - B, c0):
                if 0 <= q < n:
                    s4(q)
            for q in range(c0, c0 + B):
                if q < n:
                    s1a(q)
                    s1b(q, pregs(q))
            pr.ph()
            if f["sq"]:
                for q in range(c0, c0 + B):
                    if q < n:
                        s3(q, pregs(q))
    else:
        for q in range(n):
            s1a(q)
            s1b(q, pregs(q))
            if f["sq"]:
                pr.ph()
                s3(q, pregs(q))
        pr.ph()
    LAB = os.environ.get("GVL_A4P_LAB", "")
    if "noresld" in LAB:
        pr.items = [it for it in pr.items if not (it[0] == "VM" and it[1].startswith("buffer_load_dwordx4"))]
    if "nolds" in LAB:
        pr.items = [it for it in pr.items if not (it[0] == "i" and it[1].startswith("ds_"))]
    return pr


def end_stores(epi):
    """the output stores of the tile the deferred program has finished, by block row j: lists of [scalar offset, store] pairs (+ the 4 row-statistics stores last)"""
    f = epi_flags(epi)
    if "nostore" in os.environ.get("GVL_A4P_LAB", ""):
        return [[] for _ in range(MB)]
    out = []
    for j in range(MB):
        S = []
        for half in range(1 if f["act"] == 3 else 2):
            if STORES == "p3" and not f["resid"]:
                break
            base = preg(j, 2 * half, 0)
            for k in range(4):
                S.append([f"s_mul_i32 s{S_OC}, s{S_LDC8}, {4 * j + k}", f"buffer_store_dwordx4 {vr(base + 4 * k)}, v{V_VOC[half]}, {sr(S_RSC)}, s{S_OC} offen"])
        out.append(S)
    if f["sq"]:
        for k in range(4):
            out[-1].append([f"s_mul_i32 s{S_OQ}, s{S_LDQ8}, {k}", f"buffer_store_dword v{V_SQ + k}, v{V_VOQ[0]}, {sr(S_RSQ)}, s{S_OQ} offen"])
    return out


CAP = int(os.environ.get("GVL_A4P_CAP", "5"))            # instructions per MFMA gap, main loop's own included


def place(bodies, prog):
    """distribute the deferred program over the gaps of `bodies` (lists of 4 Phases), in order.  Raises when it does not fit."""
    pos = [0, 0, 0]                     # body, phase, gap

    def advance_phase():
        pos[1] += 1
        pos[2] = 0
        if pos[1] == 4:
            pos[0] += 1
            pos[1] = 0

    def ok_here(vm):
        if pos[0] >= len(bodies):
            raise RuntimeError("deferred program does not fit the unrolled bodies")
        ph = bodies[pos[0]][pos[1]]
        if vm and pos[1] != 3:
            return False
        return len(ph.gaps[pos[2]]) < (ph.cap or CAP)

    for kind, *rest in prog.items:
        if kind == "PH":
            advance_phase()
            continue
        if kind == "KT":
            # the next phase 3 that lies behind an end-of-phase-2 wait not yet passed
            if pos[1] == 3 and pos[2] > 0:
                pos[0] += 1
            elif pos[1] == 3 and pos[2] == 0:
                pass
            pos[1], pos[2] = 3, 0
            continue
        vm = kind == "VM"
        while True:
            if pos[0] < len(bodies) and pos[2] >= len(bodies[pos[0]][pos[1]].gaps):
                advance_phase()
                continue
            if ok_here(vm):
                break
            pos[2] += 1
        bodies[pos[0]][pos[1]].gaps[pos[2]].append(rest[0])


# ---- drain --------------------------------------------------------------------------------------------------------------------------------------------------
BIAS0 = 160               # the tile's 64 bias floats of this lane, (i, b) -> v[BIAS0 + 16 i + 4 b ...]: the fragment registers, dead once the main loop has ended


def bias_reads(i_list):
    return [f"ds_read_b128 {vr(BIAS0 + 16 * i + 4 * b)}, v{V_BADDR} offset:{(i * 32 + 8 * b) * 4}" for i in i_list for b in range(4)]


def drain_row(epi, j, nb=NB):
    """a[block row j] -> (x row scale) (+ bias) -> bf16 pairs in P(j).  Batches of 4 pairs (8 reads, 4 packed ops, 4 converts): a dependent instruction sits >= 4
    issue slots behind its producer -- one wave per SIMD has nobody else to cover a VALU dependency stall.  The bias sits in registers (bias_reads)."""
    f = epi_flags(epi)
    L = []
    if "nodrain" in os.environ.get("GVL_A4P_LAB", ""):
        return L
    rsp, hi = V_RS + 2 * (j // 2), j % 2
    for i in range(nb):
        for b0 in range(0, 4, 2):
            rd, op, cv = [], [], []
            for n, (b, q) in enumerate([(b, q) for b in (b0, b0 + 1) for q in range(2)]):
                t = T0 + 2 * n
                a0 = (4 * j + i) * 16 + 4 * b + 2 * q
                bias = BIAS0 + 16 * i + 4 * b + 2 * q
                rd += [f"v_accvgpr_read_b32 v{t}, a{a0}", f"v_accvgpr_read_b32 v{t + 1}, a{a0 + 1}"]
                if f["rs"] and f["bias"]:
                    op.append(f"v_pk_fma_f32 {vr(t, 2)}, {vr(t, 2)}, {vr(rsp, 2)}, {vr(bias, 2)} op_sel:[0,{hi},0] op_sel_hi:[1,{hi},1]")
                elif f["rs"]:
                    op.append(f"v_pk_mul_f32 {vr(t, 2)}, {vr(t, 2)}, {vr(rsp, 2)} op_sel:[0,{hi}] op_sel_hi:[1,{hi}]")
                elif f["bias"]:
                    op.append(f"v_pk_add_f32 {vr(t, 2)}, {vr(t, 2)}, {vr(bias, 2)}")
                cv.append(f"v_cvt_pk_bf16_f32 v{preg(j, i, 2 * b + q)}, v{t}, v{t + 1}")
            L += rd + op + cv
    return L


def end_section(epi, stores, nb=NB):
    """behind the last MFMA: the drain, block row by block row, with the finished tile's stores of row j + 1 spread through the drain of row j (row 0's stores sit in the
    gaps of the LAST body's phase 3): a store is issued before the drain overwrites its registers, and no s_waitcnt vmcnt is near -- the next one is the end of phase 2
    of the next tile's first k-tile, > 4 k cycles away."""
    L = []
    if epi & 32:
        # bias of column blocks 2, 3 -> fragment set Y (blocks 0, 1 went to set X in the LAST body's phase 3, which multiplies out of Y): ONE exposed LDS round trip
        # for the tile's 64 values instead of one per (block row, column group) -- 32 of them, ~3 k cycles
        L += (bias_reads([2, 3]) if nb == NB else []) + ["s_waitcnt lgkmcnt(0)"]
    for j in range(MB):
        D = drain_row(epi, j, nb)
        S = stores[j + 1] if j + 1 < MB else []
        if not D:
            for pair in S:
                L += pair
            continue
        step = max(1, len(D) // (len(S) + 1))
        si = 0
        for n, ins in enumerate(D):
            L.append(ins)
            if si < len(S) and (n + 1) % step == 0 and not ins.startswith("ds_read"):
                L += S[si]
                si += 1
        while si < len(S):
            L += S[si]
            si += 1
    return L


def lane_setup(epi=0):
    """per-lane addresses of the deferred program / drain from the lane id (once per statement)"""
    t, u = T0, T0 + 1
    hs = 2 if (epi & 3) == 3 else 3            # SwiGLU writes 4-byte pieces at + 4 h, everything else 8-byte pieces at + 8 h
    return [f"v_mbcnt_lo_u32_b32 v{t}, -1, 0", f"v_mbcnt_hi_u32_b32 v{t}, -1, v{t}",
            # write base: staging + l31 * 128 + h * 8 + ((l31 & 7) << 4)
            f"v_and_b32 v{u}, 31, v{t}", f"v_lshlrev_b32 v{V_WADDR}, 7, v{u}", f"v_and_b32 v{u}, 7, v{u}", f"v_lshl_or_b32 v{V_WADDR}, v{u}, 4, v{V_WADDR}",
            f"v_lshrrev_b32 v{u}, 5, v{t}", f"v_lshl_or_b32 v{V_WADDR}, v{u}, {hs}, v{V_WADDR}", f"v_add_u32 v{V_WADDR}, s{S_LSTG}, v{V_WADDR}",
            f"v_lshl_add_u32 v{V_BADDR}, v{u}, 4, s{S_LSTG}", f"v_lshl_add_u32 v{V_GADDR}, v{u}, 4, s{S_LGAMMA}",
            # read-back base: staging + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4)
            f"v_lshrrev_b32 v{u}, 3, v{t}", f"v_and_b32 v{t}, 7, v{t}", f"v_xor_b32 v{t}, v{t}, v{u}", f"v_and_b32 v{t}, 7, v{t}",
            f"v_lshlrev_b32 v{V_RADDR}, 7, v{u}", f"v_lshl_or_b32 v{V_RADDR}, v{t}, 4, v{V_RADDR}", f"v_add_u32 v{V_RADDR}, s{S_LSTG}, v{V_RADDR}"]


def tile_asm(epi, with_deferred, nb=NB):
    """nb = 2: the NARROW statement (body()): same deferred program (it belongs to the PREVIOUS tile, whatever that tile's kind), same stores; the main loop and the drain
    cover column blocks 0, 1 only.  P of blocks 2, 3 keeps stale values: the next statement's program transposes and 'stores' them through offsets the host marked
    invalid (0x80000000: dropped by the buffer bounds check)."""
    prog = deferred(epi) if (with_deferred and "nodefer" not in os.environ.get("GVL_A4P_LAB", "")) else Prog()
    stores = end_stores(epi) if with_deferred else [[] for _ in range(MB)]
    ng = MB * nb
    # bodies that may carry deferred work: FIRST, E1 .. EU (unrolled STEADY), then the loop, PENULT, LAST
    u = 0
    while True:
        bodies = [body("FIRST", epi, nb)] + [body("STEADY", epi, nb) for _ in range(u)]
        try:
            place(bodies, prog)
            break
        except RuntimeError:
            u += 1
            assert u < 40, "deferred program too long"
    L = SHIFT + [f"s_mov_b32 s{S_M0}, m0", f"s_mov_b32 s{S_KOFF}, 128", f"s_mov_b32 s{S_DMA}, s{S_DMAO}", f"s_sub_u32 s{S_CNT}, s{S_NK}, {3 + u}"]
    for regs, src in ((RDW, V_RDW0), (RDA, V_RDA0)):
        L.append(f"v_mov_b32 v{regs[0]}, v{src}")
        for ph in range(1, 4):
            L.append(f"v_xor_b32 v{regs[ph]}, 0x{ph << 5:x}, v{regs[0]}")
    L += lane_setup(epi)
    L += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    for ins in reads(0, X, nb):
        L += ins
    for b in bodies:
        for ph in b:
            L += ph.lines()
    sfx = "" if nb == NB else "n"
    L += [f"s_cmp_eq_u32 s{S_CNT}, 0", f"s_cbranch_scc1 .Lgvl_a4p_pen{sfx}_%=", *ALIGN_LOOP, f".Lgvl_a4p_loop{sfx}_%=:"]
    for ph in body("STEADY", epi, nb):
        L += ph.lines()
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", f"s_cbranch_scc1 .Lgvl_a4p_loop{sfx}_%=", f".Lgvl_a4p_pen{sfx}_%=:"]
    for ph in body("PENULT", epi, nb):
        L += ph.lines()
    last = body("LAST", epi, nb)
    if epi & 32:
        for k, ins in enumerate(bias_reads([0, 1])):  # set X is dead in the last k-tile's phase 3 (no next k step to prefetch); the slice landed at the phase-2 wait
            last[3].gaps[(2 * k + 1 if k < 8 else 15) if nb == NB else k].append(ins)
    for k, pair in enumerate(stores[0]):              # the finished tile's block row 0: behind the end-of-phase-2 wait of the last k-tile, in front of the drain
        last[3].gaps[(2 * k) % 16 if nb == NB else k % ng] += pair
    for ph in last:
        L += ph.lines()
    # the last MFMAs' results must be readable by v_accvgpr_read (XDL write -> VALU read: 18 wait states for 16 passes)
    L += ["s_nop 15", "s_nop 3"]
    L += end_section(epi, stores, nb)
    L += [f"s_mov_b32 m0, s{S_M0}"]
    return L, u


def flush_asm(epi):
    """the deferred program alone, then its stores (behind the last tile of a workgroup)"""
    L = lane_setup(epi)
    for kind, *rest in deferred(epi).items:
        if kind == "PH":
            L.append("s_waitcnt lgkmcnt(0)")
        elif kind == "KT":
            L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
        else:
            L.append(rest[0])
    L.append("s_waitcnt lgkmcnt(0)")
    for S in end_stores(epi):
        for pair in S:
            L += pair
    return L


def dma_tile_asm():
    L = [f"s_mov_b32 s{S_M0}, m0", f"s_mov_b32 s{S_KOFF}, 0", f"v_mov_b32 v{RW}, v{V_VOW}", f"v_mov_b32 v{RA}, v{V_VOA}"]
    for op, rs, st, r in (("W", sr(S_RSW), S_STEPW, RW), ("A", sr(S_RSA), S_STEPA, RA)):
        for i in range(8):
            # s_dmao is the OTHER slot: the first tile's k-tile 0 goes to slot 0 = other ^ 0x10000
            L += [f"s_xor_b32 s{S_DMA}, s{S_DMAO}, 0x{SLOT:x}" if (op == "W" and i == 0) else None,
                  f"s_add_u32 m0, s{S_DMA}, 0x{(0x8000 if op == 'A' else 0) + i * 0x1000:x}", "s_nop 0", f"buffer_load_dwordx4 v{r}, {rs}, s{S_KOFF} offen lds", f"v_add_u32 v{r}, s{st}, v{r}"]
    L += ["s_waitcnt vmcnt(0)", f"s_mov_b32 m0, s{S_M0}"]
    return [x for x in L if x]


EPIS = [0, 32, 64, 128, 8, 136, 184, 98, 3, 67]
NARROW_EPIS = [184]       # epilogues that also get the narrow statements (InternVideo2 proj / fc2: N = 1408 = 5.5 tile columns)


def macro(name, lines):
    return [f"#define {name} \\"] + [f'  "{l}\\n\\t" \\' for l in lines[:-1]] + [f'  "{lines[-1]}"', ""]


def render():
    o = ["// GENERATED by tools/gen_gemm4p.py -- do not edit; the schedule and its reasoning live there.", "#pragma once", ""]
    for e in EPIS:
        t0, u0 = tile_asm(e, False)
        t1, u1 = tile_asm(e, True)
        un = None
        if e in NARROW_EPIS:
            # ONE statement holds both codes and branches on s_narrow: two asm statements under a C++ `if` made hipcc spill P (134 VGPRs) at the join
            n0_, _ = tile_asm(e, False, 2)
            n1_, un = tile_asm(e, True, 2)
            # the flag travels in v[V_RS] (these epilogues have no row scale; the statistics that live there are written later in the statement): a per-tile scalar
            # in the s[36:51] block made hipcc build that tuple in VGPRs ("V_READFIRSTLANE_B32 of a tuple")
            assert not epi_flags(e)["rs"]
            both = lambda wide, nar: ([f"v_readfirstlane_b32 s{S_OC}, v{V_RS}", f"s_cmp_eq_u32 s{S_OC}, 0", "s_cbranch_scc1 .Lgvl_a4p_wide_%="] + nar
                                      + ["s_branch .Lgvl_a4p_end_%=", ".Lgvl_a4p_wide_%=:"] + wide + [".Lgvl_a4p_end_%=:"])
            t0, t1 = both(t0, n0_), both(t1, n1_)
        o += macro(f"GVL_A4P_TILE0_E{e}", t0) + macro(f"GVL_A4P_TILE_E{e}", t1) + macro(f"GVL_A4P_FLUSH_E{e}", flush_asm(e))
        o += [f"#define GVL_A4P_MIN_NK_E{e} {3 + u1}", ""]
        if un is not None:
            o += [f"#define GVL_A4P_MIN_NK_N_E{e} {3 + un}", ""]
    o += macro("GVL_A4P_DMA_TILE_ASM", dma_tile_asm())
    acl = ", ".join(f'"a{i}"' for i in range(256))
    vcl = ", ".join(f'"v{i}"' for i in range(V_LO, V_HI + 1))
    scl = ", ".join(f'"s{i}"' for i in S_WORK) + ', "vcc"'
    o += [f"#define GVL_A4P_CLOBBER_AGPRS {acl}", f"#define GVL_A4P_CLOBBER_VGPRS {vcl}", f"#define GVL_A4P_CLOBBER_SGPRS {scl}", ""]
    o += ["#define GVL_A4P_EPI_LIST(X) " + " ".join(f"X({e})" for e in EPIS), "#define GVL_A4P_NARROW_LIST(X) " + " ".join(f"X({e})" for e in NARROW_EPIS), ""]
    return "\n".join(o)


if __name__ == "__main__":
    txt = render()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == txt else 1)
    open(OUT, "w").write(txt)
    print(f"wrote {os.path.normpath(OUT)}: {len(txt.splitlines())} lines")
