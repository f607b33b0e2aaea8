#!/usr/bin/env python3
"""Generator of the hand-scheduled K loops of the bf16 GEMM (ultravox_amd/csrc/gemm_a4_loop.inc).

A loop is ONE inline-asm statement: prologue (pipeline fill), steady-state body, three tail bodies and the drain.  hipcc sees
an opaque instruction; every register inside is named here.  Two geometries share the structure:

  a4: 256 threads = 4 waves as 2 (M) x 2 (N), one wave per SIMD, wave tile (16 MI) x 128: 8 x MI accumulator fragments
  a8: 512 threads = 8 waves as 2 (M) x 4 (N), two waves per SIMD (w and w + 4), wave tile (16 MI) x 64: 4 x MI fragments

of v_mfma_f32_16x16x32_bf16 in AGPRs a[0 : 4 NJ MI), block tile (32 MI) x 256 x 64.  Operand tiles are K = 64 deep (full 128-byte
rows, 16-byte chunks XOR-swizzled by row & 7 exactly as in the eight-phase kernels), staged by LDS-DMA into a RING OF FIVE
32 KiB SLOTS (all 160 KiB of the CU):

    unit 2t = W(t)  [256 rows],  unit 2t+1 = X(t)  [32 MI rows],  unit u lives in slot u mod 5.

While tile t is computed (slots 2t, 2t+1) the ring also holds W(t+1), X(t+1) (landed) and W(t+2) (in flight).  Half-way
through tile t - after the reads of its second k-half have returned - ONE barrier B_t frees its two slots, and
X(t+2) goes into W(t)'s slot, W(t+3) into X(t)'s: the weights (HBM, long latency) run 2 K-tiles ahead, the activations
(L2 / Infinity Cache) one.

Per K-tile and wave: phase A = NJ MI MFMAs on k-half 0 (fragments F0) with the NJ + MI ds_read_b128 of k-half 1 -> F1
in the gaps; [s_waitcnt vmcnt(NW) lgkmcnt(0); s_barrier]; phase B = NJ MI MFMAs on F1 with the LDS-DMA issue of X(t+2),
W(t+3) and the reads of tile t+1's k-half 0 -> F0 in the gaps.  Same MFMA, operand roles and k order as every other
variant of the family, so results are bit-identical across tile variants.

Why a8 exists (measured, profiles/r04_gemm_a4_ktile_probe.txt): with ONE wave per SIMD an LDS-DMA instruction blocks the wave's
issue for ~60 cycles while its MFMA pipe holds 16 cycles of work - the a4 K-tile costs 2660 cycles (2047 with the DMA removed:
the MFMA floor).  With two free-running waves per SIMD one wave's DMA / LDS issue sits under the other's MFMAs.

vmcnt bookkeeping (loads retire in order): at B_t's wait the youngest issued unit is W(t+2) (NW instructions per wave),
everything older - in particular X(t+1) - must have landed: vmcnt(NW); in the tails, where W(t+2) does not exist, vmcnt(0).
Bodies: FULL (t <= nk-4), T3 (t = nk-3: issues X(nk-1) only), T2 (t = nk-2: no DMA, vmcnt(0)), T1 (last tile: no DMA,
no barrier, no look-ahead reads).  The prologue always issues five units with k clamped to the last K-tile (units that
do not exist land in slots nobody reads), so it needs no branches and its wait is always vmcnt(2 NW + NX).

usage: python tools/gen_gemm_a4.py   (rewrites ultravox_amd/csrc/gemm_a4_loop.inc)
"""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ultravox_amd", "csrc", "gemm_a4_loop.inc")

SLOT = 32768
RING = 5 * SLOT
# literal SGPRs (shared by both geometries)
S_M0, S_CNT, S_UW, S_UX, S_UWN, S_UXN = 70, 71, 72, 73, 74, 75
S_PX, S_PW = 76, 78   # 64-bit running source pointers of X(t+2), W(t+3)
S_DX, S_DW, S_T = 80, 81, 86
S_PP, S_K1, S_K2 = 82, 84, 85
SGPR_CLOBBER = range(70, 88)


class Geo:
    """nwn = waves along N (2: a4, 4: a8); mi = X fragments (16 rows) per wave."""

    def __init__(self, nwn, mi):
        self.nwn, self.mi = nwn, mi
        self.waves = 2 * nwn
        self.nj = 256 // nwn // 16                 # W fragments per wave
        self.nw = 32 // self.waves                 # DMA instructions per wave and W unit
        assert (4 * mi) % self.waves == 0, "X unit must split evenly over the waves"
        self.nx = 4 * mi // self.waves             # ... per X unit
        self.dstep = self.waves * 1024             # LDS distance between a wave's consecutive DMA instructions
        self.nacc = 4 * self.nj * mi
        fr = 4 * (self.nj + mi)                    # registers of one fragment set
        if nwn == 2:
            self.vtmp = 120
            self.fw = (128, 128 + fr)
        else:                                      # 256 registers per wave: 128 accumulators leave 128 VGPRs
            assert self.nacc <= 128
            self.vtmp = 28
            self.fw = (32, 32 + fr)
        self.fx = (self.fw[0] + 4 * self.nj, self.fw[1] + 4 * self.nj)
        self.vclobber = range(self.vtmp, self.fw[1] + fr)
        assert self.fw[1] + fr <= (256 if nwn == 2 else 128)
        self.name = f"A{self.waves}_MI{mi}"

    def wfrag(self, s, j):
        b = self.fw[s] + 4 * j
        return f"v[{b}:{b + 3}]"

    def xfrag(self, s, i):
        b = self.fx[s] + 4 * i
        return f"v[{b}:{b + 3}]"

    def acc(self, j, i):
        b = (j * self.mi + i) * 4
        return f"a[{b}:{b + 3}]"

    def mfmas(self, s):
        """The NJ MI MFMAs of one k-half on fragment set s (W fragment = A operand, X fragment = B operand, as the family)."""
        return [f"v_mfma_f32_16x16x32_bf16 {self.acc(j, i)}, {self.wfrag(s, j)}, {self.xfrag(s, i)}, {self.acc(j, i)}"
                for j in range(self.nj) for i in range(self.mi)]

    def reads(self, s, vw, vx):
        """ds_read_b128 of one k-half into fragment set s; v{vw} (W) / v{vx} (X) already hold slot + lane offset."""
        out = [f"ds_read_b128 {self.xfrag(s, i)}, v{vx} offset:{i * 2048}" for i in range(self.mi)]   # X first: MFMAs walk i
        out += [f"ds_read_b128 {self.wfrag(s, j)}, v{vw} offset:{j * 2048}" for j in range(self.nj)]
        return out

    def dma_stream(self, n, off_fmt, ptr, sdst):
        """LDS-DMA of this wave's share of one unit: n instructions, destination s{sdst} + dstep i, source pointer s[ptr:ptr+1].
        Flat list [m0 write, load, m0 write, load, ...]: the interleaver keeps an MFMA (or s_nop) between the M0 write and the
        load that reads it (1 wait state required)."""
        out = []
        for i in range(n):
            out.append(f"s_add_u32 m0, s{sdst}, {i * self.dstep}" if i else f"s_mov_b32 m0, s{sdst}")
            out.append(f"global_load_lds_dwordx4 {off_fmt.format(i)}, s[{ptr}:{ptr + 1}]")
        return out


PROBE = dict(no_dma=False, no_reads=False, no_mfma=False, prio=True)     # probe builds (set per schedule): what the loop leaves out


def interleave(mf, streams):
    """streams: list of (first_gap, per_gap, stride, [instructions]); gap g = after MFMA g (g = -1: before the first).
    A stream puts per_gap instructions into every stride-th gap from first_gap on, in order; streams may share gaps.  An M0
    write never shares a gap with the load that reads it unless an instruction of another kind separates them (asserted)."""
    gaps = {}
    for first, per, stride, ins in streams:
        g, k = first, 0
        for x in ins:
            gaps.setdefault(g, []).append(x)
            k += 1
            if k == per:
                g, k = g + stride, 0

    def keep(x):
        if PROBE["no_dma"] and ("global_load_lds" in x or " m0," in x):
            return False
        if PROBE["no_reads"] and x.startswith("ds_read"):
            return False
        return True

    assert all(-1 <= g < len(mf) for g in gaps), (sorted(gaps), len(mf))
    out = [x for x in gaps.get(-1, []) if keep(x)]
    for g, m in enumerate(mf):
        out.append("s_nop 0" if PROBE["no_mfma"] else m)
        out.extend(x for x in gaps.get(g, []) if keep(x))
    for a, b in zip(out, out[1:]):
        assert not (" m0," in a and "global_load_lds" in b), "M0 write directly followed by its LDS-DMA"
    return out


def body(g, kind, sched):
    """One K-tile.  kind: FULL / T3 / T2 / T1."""
    nm = g.nj * g.mi
    vt = g.vtmp
    L = []
    # ---- phase A: MFMAs on F0; reads of (t, k-half 1) -> F1 ----
    pre = [f"v_add_u32 v{vt}, s{S_UW}, %[vw1]", f"v_add_u32 v{vt + 1}, s{S_UX}, %[vx1]"]
    L += interleave(g.mfmas(0), [(-1, 2, 1, pre), (sched["a_read0"], sched.get("a_read_per", 1), sched["a_read_stride"], g.reads(1, vt, vt + 1))])
    if kind == "T1":
        L.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    else:
        L.append(f"s_waitcnt vmcnt({g.nw if kind in ('FULL', 'T3') else 0}) lgkmcnt(0)")
        L.append("s_barrier")
    # ---- phase B: MFMAs on F1; DMA X(t+2) -> slot uW, W(t+3) -> slot uX; reads of (t+1, k-half 0) -> F0; bookkeeping ----
    streams = []
    dma = []
    if kind in ("FULL", "T3"):
        dma += [f"s_add_u32 s{S_DX}, s{S_UW}, %[wofs]"] + g.dma_stream(g.nx, "%[xo{}]", S_PX, S_DX)
        dma += [f"s_add_u32 s{S_PX}, s{S_PX}, 128", f"s_addc_u32 s{S_PX + 1}, s{S_PX + 1}, 0"]
    if kind == "FULL":
        dma += [f"s_add_u32 s{S_DW}, s{S_UX}, %[wofs]"] + g.dma_stream(g.nw, "%[wo{}]", S_PW, S_DW)
        dma += [f"s_add_u32 s{S_PW}, s{S_PW}, 128", f"s_addc_u32 s{S_PW + 1}, s{S_PW + 1}, 0"]
    if kind != "T1":
        streams.append((-1, 2, 1, [f"v_add_u32 v{vt + 2}, s{S_UWN}, %[vw0]", f"v_add_u32 v{vt + 3}, s{S_UXN}, %[vx0]"]))
        streams.append((sched["b_read0"], sched.get("b_read_per", 1), sched["b_read_stride"], g.reads(0, vt + 2, vt + 3)))
    if dma:
        streams.append((sched["b_dma0"], 1, 1, dma))
    if kind != "T1":
        # ring advance (after the DMA destinations and the read addresses were formed): tile t+1 becomes the current one.
        # (each s_sub_u32 / s_cselect pair communicates through SCC: no other SALU instruction may come between them)
        ring = [f"s_mov_b32 s{S_UW}, s{S_UWN}", f"s_mov_b32 s{S_UX}, s{S_UXN}",
                f"s_add_u32 s{S_UWN}, s{S_UWN}, {2 * SLOT}", f"s_sub_u32 s{S_T}, s{S_UWN}, {RING}",
                f"s_cselect_b32 s{S_UWN}, s{S_UWN}, s{S_T}",
                f"s_add_u32 s{S_UXN}, s{S_UXN}, {2 * SLOT}", f"s_sub_u32 s{S_T}, s{S_UXN}, {RING}",
                f"s_cselect_b32 s{S_UXN}, s{S_UXN}, s{S_T}"]
        per = sched.get("ring_per", 1)
        first = nm - 1 - (len(ring) + per - 1) // per
        assert first >= sched["b_dma0"] + len(dma), ("ring advance must follow the DMA stream", first, len(dma))
        streams.append((first, per, 1, ring))
    L += interleave(g.mfmas(1), streams)
    if kind != "T1":
        L.append("s_waitcnt lgkmcnt(0)")
    return L


def prologue(g):
    L = [f"s_mov_b32 s{S_M0}, m0",
         f"s_sub_u32 s{S_T}, %[nk], 1",
         f"s_min_u32 s{S_K1}, s{S_T}, 1", f"s_lshl_b32 s{S_K1}, s{S_K1}, 7",
         f"s_min_u32 s{S_K2}, s{S_T}, 2", f"s_lshl_b32 s{S_K2}, s{S_K2}, 7"]

    def unit(is_w, slot, kreg):
        lo, hi = ("%[sBlo]", "%[sBhi]") if is_w else ("%[sAlo]", "%[sAhi]")
        if kreg is None:
            L.extend([f"s_mov_b32 s{S_PP}, {lo}", f"s_mov_b32 s{S_PP + 1}, {hi}"])
        else:
            L.extend([f"s_add_u32 s{S_PP}, {lo}, s{kreg}", f"s_addc_u32 s{S_PP + 1}, {hi}, 0"])
        L.append(f"s_add_u32 s{S_DX}, %[wofs], {slot * SLOT}")
        for i in range(g.nw if is_w else g.nx):
            L.append(f"s_add_u32 m0, s{S_DX}, {i * g.dstep}")
            L.append("s_nop 0")
            L.append(f"global_load_lds_dwordx4 {('%[wo{}]' if is_w else '%[xo{}]').format(i)}, s[{S_PP}:{S_PP + 1}]")

    unit(True, 0, None)
    unit(False, 1, None)
    unit(True, 2, S_K1)
    unit(False, 3, S_K1)
    unit(True, 4, S_K2)
    L += [f"s_add_u32 s{S_PX}, %[sAlo], 256", f"s_addc_u32 s{S_PX + 1}, %[sAhi], 0",
          f"s_add_u32 s{S_PW}, %[sBlo], 384", f"s_addc_u32 s{S_PW + 1}, %[sBhi], 0",
          f"s_mov_b32 s{S_UW}, 0", f"s_mov_b32 s{S_UX}, {SLOT}", f"s_mov_b32 s{S_UWN}, {2 * SLOT}", f"s_mov_b32 s{S_UXN}, {3 * SLOT}"]
    for r in range(g.nacc):
        L.append(f"v_accvgpr_write_b32 a{r}, 0")
    # units 0, 1 have landed when at most the three younger ones (NW + NX + NW instructions) are outstanding
    L += [f"s_waitcnt vmcnt({2 * g.nw + g.nx})", "s_barrier",
          f"v_add_u32 v{g.vtmp + 2}, s{S_UW}, %[vw0]", f"v_add_u32 v{g.vtmp + 3}, s{S_UX}, %[vx0]"]
    L += g.reads(0, g.vtmp + 2, g.vtmp + 3)
    L.append("s_waitcnt lgkmcnt(0)")
    return L


def kernel_loop(g, sched):
    L = prologue(g)
    L += [f"s_sub_i32 s{S_CNT}, %[nk], 3", f"s_cmp_le_i32 s{S_CNT}, 0", "s_cbranch_scc1 .La4_tail_%="]
    L.append(".La4_full_%=:")
    L += body(g, "FULL", sched)
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 .La4_full_%="]
    L.append(".La4_tail_%=:")
    L += ["s_cmp_lt_u32 %[nk], 3", "s_cbranch_scc1 .La4_t2_%="]
    L += body(g, "T3", sched)
    L.append(".La4_t2_%=:")
    L += ["s_cmp_lt_u32 %[nk], 2", "s_cbranch_scc1 .La4_t1_%="]
    L += body(g, "T2", sched)
    L.append(".La4_t1_%=:")
    L += body(g, "T1", sched)
    # every wave is done with the operand slots (the epilogue stages its output through them); MFMA results are readable
    L += ["s_waitcnt vmcnt(0)", "s_barrier", "s_nop 7", "s_nop 7", f"s_mov_b32 m0, s{S_M0}"]
    return L


# gap positions: phase A reads from gap a_read0, a_read_per per gap, every a_read_stride gaps; phase B: DMA stream from b_dma0 (one
# per gap), look-ahead reads from b_read0 likewise; ring advance in the last gaps, ring_per per gap.
A4_S1 = dict(a_read0=0, a_read_stride=1, b_dma0=0, b_read0=36, b_read_stride=1)
BUILDS = [
    # (geometry, schedule name, schedule)
    (Geo(2, 8), "s0", dict(a_read0=0, a_read_stride=2, b_dma0=0, b_read0=1, b_read_stride=2)),
    (Geo(2, 8), "s1", A4_S1),
    (Geo(2, 8), "s2", dict(a_read0=4, a_read_stride=3, b_dma0=0, b_read0=2, b_read_stride=3)),
    # timing probes of s1: the K-tile without the in-loop DMA / without DMA and fragment reads / without MFMAs
    (Geo(2, 8), "s3", dict(A4_S1, probe=dict(no_dma=True))),
    (Geo(2, 8), "s4", dict(A4_S1, probe=dict(no_dma=True, no_reads=True))),
    (Geo(2, 8), "s5", dict(A4_S1, probe=dict(no_mfma=True))),
    # eight waves: 32 MFMAs per phase; DMA stream = 22 instructions
    (Geo(4, 8), "s0", dict(a_read0=0, a_read_stride=2, b_dma0=0, b_read0=1, b_read_stride=2, ring_per=2)),
    (Geo(4, 8), "s1", dict(a_read0=0, a_read_stride=1, b_dma0=0, b_read0=12, b_read_stride=1, ring_per=4)),
    (Geo(4, 8), "s2", dict(a_read0=2, a_read_per=2, a_read_stride=4, b_dma0=0, b_read0=3, b_read_per=2, b_read_stride=4, ring_per=4)),
    (Geo(4, 8), "s3", dict(a_read0=0, a_read_stride=2, b_dma0=0, b_read0=1, b_read_stride=2, ring_per=2, probe=dict(no_dma=True))),
    (Geo(4, 8), "s4", dict(a_read0=0, a_read_stride=2, b_dma0=0, b_read0=1, b_read_stride=2, ring_per=2, probe=dict(no_dma=True, no_reads=True))),
    (Geo(4, 8), "s5", dict(a_read0=0, a_read_stride=2, b_dma0=0, b_read0=1, b_read_stride=2, ring_per=2, probe=dict(no_mfma=True))),
]


# ---------------------------------------------------------------------------------------------------------------------------
# a8pp: eight waves, PING-PONG by wave row.  Same tile, ring, swizzle and MFMAs as a8; what changes is who does what when.
# A wave alternates a pure MFMA phase X(t) (all 64 MFMAs of K-tile t, fragments of the whole K-tile in 96 registers) and a
# load phase Y(t+1) (24 ds_read_b128 of tile t+1 + its share of the LDS-DMA), and the two wave rows - the two waves of every
# SIMD - run half a period apart: after barrier B_s row 0 loads (DMA of X(s+2), then Y(s+1)) while row 1 computes X(s), then
# row 0 computes X(s+1) while row 1 loads (DMA of W(s+3), Y(s+1)).  One barrier per K-tile; the X / Y hand-off inside an interval
# needs no synchronisation at all (each wave just moves on), so the matrix pipe always has a wave feeding it back to back.
# The DMA is split BY OPERAND between the rows: row 0 issues the activation units right after the barrier (they are needed by the
# next barrier: one interval of lead, L2 / Infinity-Cache data), row 1 issues the weight units half an interval later (needed two
# barriers on: 1.5 intervals of lead, HBM data).  vmcnt: row 0's wait before a barrier is vmcnt(0) (its youngest unit is the one
# needed), row 1's vmcnt(8) (W(s+3) stays in flight).  Units beyond the last K-tile are issued with k clamped to the last tile
# (they land in slots nobody reads again), so the bodies have no tail variants; the drain waits for them.
#   row 0:  P  Y(0) X(0) B_0  { DMA_X(s) Y(s+1) X(s+1) B_{s+1} }  s = 0 .. nk-2          F
#   row 1:  P  Y(0)      B_0  { X(s) DMA_W(s) Y(s+1)   B_{s+1} }  s = 0 .. nk-2  X(nk-1)  F
PP_VT = 28
PP_W, PP_X = 32, 64        # W fragment (kh, j) at v[32 + 4 (4 kh + j)], X fragment (kh, i) at v[64 + 4 (8 kh + i)]
S_K, S_KMAX = 76, 77       # byte offset of the unit this row issues next, its clamp
S_PTR = 78                 # s[78:79] = source pointer of the unit being issued


def pp_w(kh, j):
    b = PP_W + 4 * (4 * kh + j)
    return f"v[{b}:{b + 3}]"


def pp_x(kh, i):
    b = PP_X + 4 * (8 * kh + i)
    return f"v[{b}:{b + 3}]"


def pp_mfmas():
    out = []
    for kh in range(2):
        for j in range(4):
            for i in range(8):
                b = (j * 8 + i) * 4
                out.append("s_nop 0" if PROBE["no_mfma"] else f"v_mfma_f32_16x16x32_bf16 a[{b}:{b + 3}], {pp_w(kh, j)}, {pp_x(kh, i)}, a[{b}:{b + 3}]")
    return out


def pp_reads(sw, sx):
    """Y: v28..v31 = read addresses of (W, X) x (kh 0, 1) in the slots s{sw} / s{sx}; 24 reads."""
    pre = [f"v_add_u32 v{PP_VT}, s{sw}, %[vw0]", f"v_add_u32 v{PP_VT + 1}, s{sx}, %[vx0]",
           f"v_add_u32 v{PP_VT + 2}, s{sw}, %[vw1]", f"v_add_u32 v{PP_VT + 3}, s{sx}, %[vx1]"]
    rd = []
    for kh in range(2):
        rd += [f"ds_read_b128 {pp_x(kh, i)}, v{PP_VT + 1 + 2 * kh} offset:{i * 2048}" for i in range(8)]
        rd += [f"ds_read_b128 {pp_w(kh, j)}, v{PP_VT + 2 * kh} offset:{j * 2048}" for j in range(4)]
    return pre, rd


def pp_dma(sdst):
    """this wave's 8 instructions of one unit -> slot s{sdst}: source s[S_PTR:+1] = base + k"""
    if PROBE["no_dma"]:
        return [], []
    head = [f"s_add_u32 s{S_PTR}, %[gplo], s{S_K}", f"s_addc_u32 s{S_PTR + 1}, %[gphi], 0", f"s_add_u32 s{S_DX}, s{sdst}, %[wofs]"]
    ins = []
    for i in range(8):
        ins.append((f"s_add_u32 m0, s{S_DX}, {i * 4096}", f"global_load_lds_dwordx4 %[go{i}], s[{S_PTR}:{S_PTR + 1}]"))
    return head, ins


def pp_load_phase(sdst, sw, sx):
    """DMA of one unit (8 instructions) woven with the 24 fragment reads: [m0 write, read, load, read, read] x 8."""
    head, dma = pp_dma(sdst)
    pre, rd = pp_reads(sw, sx)
    L = head + pre
    rd = list(rd)
    for m0w, ld in dma:
        L += [m0w, rd.pop(0), ld, rd.pop(0), rd.pop(0)]
    if not dma and not PROBE["no_reads"]:
        L += rd
        rd = []
    if PROBE["no_reads"]:
        L = [x for x in L if not x.startswith("ds_read")]
        L = sum(([x, "s_nop 0"] if " m0," in x else [x] for x in L), [])
    assert not rd or PROBE["no_reads"]
    # next unit of this row: k <- min(k + 128, kmax)
    L += [f"s_add_u32 s{S_K}, s{S_K}, 128", f"s_min_u32 s{S_K}, s{S_K}, s{S_KMAX}"]
    return L


PP_RING = [f"s_mov_b32 s{S_UW}, s{S_UWN}", f"s_mov_b32 s{S_UX}, s{S_UXN}",
           f"s_add_u32 s{S_UWN}, s{S_UWN}, {2 * SLOT}", f"s_sub_u32 s{S_T}, s{S_UWN}, {RING}", f"s_cselect_b32 s{S_UWN}, s{S_UWN}, s{S_T}",
           f"s_add_u32 s{S_UXN}, s{S_UXN}, {2 * SLOT}", f"s_sub_u32 s{S_T}, s{S_UXN}, {RING}", f"s_cselect_b32 s{S_UXN}, s{S_UXN}, s{S_T}"]


def pp_prologue_unit(slot, kreg):
    L = []
    if kreg is None:
        L += [f"s_mov_b32 s{S_PTR}, %[gplo]", f"s_mov_b32 s{S_PTR + 1}, %[gphi]"]
    else:
        L += [f"s_add_u32 s{S_PTR}, %[gplo], s{kreg}", f"s_addc_u32 s{S_PTR + 1}, %[gphi], 0"]
    L.append(f"s_add_u32 s{S_DX}, %[wofs], {slot * SLOT}")
    for i in range(8):
        L += [f"s_add_u32 m0, s{S_DX}, {i * 4096}", "s_nop 0", f"global_load_lds_dwordx4 %[go{i}], s[{S_PTR}:{S_PTR + 1}]"]
    return L


def pp_loop():
    zero = [f"v_accvgpr_write_b32 a{r}, 0" for r in range(128)]
    ring0 = [f"s_mov_b32 s{S_UW}, 0", f"s_mov_b32 s{S_UX}, {SLOT}", f"s_mov_b32 s{S_UWN}, {2 * SLOT}", f"s_mov_b32 s{S_UXN}, {3 * SLOT}"]
    com = [f"s_mov_b32 s{S_M0}, m0",
           f"s_sub_u32 s{S_T}, %[nk], 1", f"s_lshl_b32 s{S_KMAX}, s{S_T}, 7",          # kmax = (nk - 1) * 128
           f"s_min_u32 s{S_K1}, s{S_T}, 1", f"s_lshl_b32 s{S_K1}, s{S_K1}, 7",
           f"s_min_u32 s{S_K2}, s{S_T}, 2", f"s_lshl_b32 s{S_K2}, s{S_K2}, 7",
           f"s_sub_u32 s{S_CNT}, %[nk], 1"] + ring0 + ["s_cmp_lg_u32 %[wr], 0", "s_cbranch_scc1 .Lpp_row1_%="]
    pre, rd = pp_reads(S_UW, S_UX)
    y0 = pre + ([] if PROBE["no_reads"] else rd) + ["s_waitcnt lgkmcnt(0)"]
    # ---- row 0: activation units (slots 1, 3 in the prologue; then X(s+2) -> W(s)'s slot) ----
    r0 = pp_prologue_unit(1, None) + pp_prologue_unit(3, S_K1) + zero
    r0 += [f"s_min_u32 s{S_K}, s{S_KMAX}, 256",                                       # next unit: X(2)
           "s_waitcnt vmcnt(8)", "s_barrier"] + y0 + pp_mfmas() + ["s_waitcnt vmcnt(0)", "s_barrier"]
    r0 += [f"s_cmp_eq_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lpp_end_%=", ".Lpp_loop0_%=:"]
    # (the MFMAs of k-half 0 start as soon as ITS 12 fragments are in - LDS reads return in order - k-half 1's arrive under them)
    mf = pp_mfmas()
    r0 += pp_load_phase(S_UW, S_UWN, S_UXN) + PP_RING + ["s_waitcnt lgkmcnt(12)"] + mf[:32] + ["s_waitcnt lgkmcnt(0)"] + mf[32:]
    r0 += ["s_waitcnt vmcnt(0)", "s_barrier"]
    r0 += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lpp_loop0_%=", "s_branch .Lpp_end_%="]
    # ---- row 1: weight units (slots 0, 2, 4 in the prologue; then W(s+3) -> X(s)'s slot) ----
    r1 = [".Lpp_row1_%=:"] + pp_prologue_unit(0, None) + pp_prologue_unit(2, S_K1) + pp_prologue_unit(4, S_K2) + zero
    r1 += [f"s_min_u32 s{S_K}, s{S_KMAX}, 384",                                       # next unit: W(3)
           "s_waitcnt vmcnt(16)", "s_barrier"] + y0[:-1] + ["s_waitcnt vmcnt(8) lgkmcnt(0)", "s_barrier"]
    r1 += [f"s_cmp_eq_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lpp_last1_%=", ".Lpp_loop1_%=:"]
    # Row 1 starts its MFMA phase AT the barrier, row 0 a load phase later: without help the two MFMA phases overlap, share the
    # pipe, row 1 finishes late and its load phase sticks out past row 0's MFMAs (measured: 2750 cycles per K-tile).  Row 1
    # therefore runs its MFMA phase at high issue priority - it owns the pipe for the first half of the interval, row 0 for the
    # second, and row 1's load phase sits entirely under row 0's MFMAs.
    hi, lo = (["s_setprio 3"], ["s_setprio 0"]) if PROBE.get("prio", True) else ([], [])
    r1 += hi + pp_mfmas() + lo + pp_load_phase(S_UX, S_UWN, S_UXN) + PP_RING + ["s_waitcnt vmcnt(8) lgkmcnt(0)", "s_barrier"]
    r1 += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 .Lpp_loop1_%=", ".Lpp_last1_%=:"]
    r1 += hi + pp_mfmas() + lo
    end = [".Lpp_end_%=:", "s_waitcnt vmcnt(0)", "s_barrier", "s_nop 7", "s_nop 7", f"s_mov_b32 m0, s{S_M0}"]
    L = com + r0 + r1 + end
    for a, b in zip(L, L[1:]):
        assert not (" m0," in a and "global_load_lds" in b), "M0 write directly followed by its LDS-DMA"
    return L


PP_BUILDS = [("s0", dict(prio=True)), ("s1", dict(prio=False)), ("s2", dict(prio=True, no_dma=True)), ("s3", dict(prio=True, no_dma=True, no_reads=True))]


def emit(f, name, lines):
    f.write(f"#define {name} \\\n")
    for ln in lines:
        f.write(f'  "{ln}\\n\\t" \\\n')
    f.write('  ""\n\n')


def main():
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_a4.py - do not edit (edit the generator and re-run it).\n")
        f.write("// The K loops of gemm_nt_bf16_a4_kernel / gemm_nt_bf16_a8_kernel as one inline-asm statement each; register map,\n")
        f.write("// ring and wait bookkeeping are documented in the generator.\n\n")
        seen = set()
        for g, sname, sched in BUILDS:
            for k in PROBE:
                PROBE[k] = bool(sched.get("probe", {}).get(k, False))
            emit(f, f"UVX_{g.name}_LOOP_{sname.upper()}", kernel_loop(g, sched))
            if g.name in seen:
                continue
            seen.add(g.name)
            cl = [f'"v{r}"' for r in g.vclobber] + [f'"s{r}"' for r in SGPR_CLOBBER] + ['"vcc"', '"scc"', '"memory"']
            cl += [f'"a{r}"' for r in range(g.nacc)]
            f.write(f"#define UVX_{g.name}_CLOBBER " + ", ".join(cl) + "\n\n")
        for sname, probe in PP_BUILDS:
            for k in PROBE:
                PROBE[k] = bool(probe.get(k, False))
            emit(f, f"UVX_A8PP_MI8_LOOP_{sname.upper()}", pp_loop())
        cl = [f'"v{r}"' for r in range(PP_VT, 128)] + [f'"s{r}"' for r in SGPR_CLOBBER] + ['"vcc"', '"scc"', '"memory"']
        cl += [f'"a{r}"' for r in range(128)]
        f.write("#define UVX_A8PP_MI8_CLOBBER " + ", ".join(cl) + "\n\n")
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
