#!/usr/bin/env python3
"""Generate clima_amd/csrc/rorr_xys_asm.inc: one random-overlap mixing step (reference: k_rorr,
/root/reference/src/radtran/clima_radtran_types.f90:823-852 -- the ng*ng sums, mrgrnk, weights_to_bins, rebin) for
8 g-points with x and y both ascending, as ONE block of gfx950 assembly with its registers named by this script.

Why assembly.  The step is a 64-key register sort and a rebin over the sorted keys, ~1000 instructions with 136
live f64 values.  Two pieces of work can be left out wave by wave (below), but both need conditional blocks around
code that holds all 64 keys -- and at every join hipcc's allocator wants the keys in the same registers on both paths,
which it repairs with v_mov storms and scratch spills (2.7x slower; tools/gen_sort_network_inplace.py tells the story).
Here the registers are fixed: v96..v223 hold key 0..63, v224..v239 are the sort's eight spares (the rebin's
scratch afterwards), v240..v255 one of the rebin's two weight buffers; the x / y operand registers are reused once the keys are built.

What is left out (all wave-uniform, decided from the operands of the 64 lanes):
  * merges whose two runs do not interleave.  Rows g and g+1 of the 8x8 sums are separate when
    x_(g+1) - x_g >= (y_7 - y_0) + 2^-40 (x_7 + y_7) (the margin covers the rounding of the sums and the pair index in
    their low mantissa bits, so the KEYS are in order): the merge that would join them is a no-op.  The same test
    with x and y exchanged describes the columns (the same 64 keys laid out transposed); the wave looks at ONE view --
    rows when y's range is the smaller of the two ranges in most of its lanes, columns otherwise (testing both and
    taking the better one cost 55 more instructions per step than it saved).
  * the rebin of rows that stand alone.  If the gaps r-1 .. 6 all hold, rows r .. 7 are each one ascending run above
    everything before them, the output edges E_(r+1) .. E_8 fall exactly between them (the pair weights of a row sum
    to the row's weight), and new_k = sum_j w_j key(k, j) for those rows -- 8 instructions instead of ~50.  The batch
    loop of the window rebin ends after row r-1; with r = 0 (every row separate: the sums are already in the
    reference's order) nothing is sorted or rebinned at all.  That closed form differs from the integral's difference
    quotient at the 1e-15 level (it is the more accurate of the two).
Census of how often each holds: tools/census_rorr.py, profiles/r04_census_rorr.txt.

Operands of the asm statement (kernels.hip, rorr_mix8): %0-%7 x[0..7] (+v; garbage afterwards), %8-%15 y[0..7] (+v; the
new coefficients come back in them), %16 LDS byte address of the pair-weight table s_wxy (512-byte aligned), %17
global pointer to the handle's table [E_1..E_8, w_0..w_7, 1/(E_(k+1)-E_k)] (24 doubles, OpacityParams::rorr_tab).
Registers: the block names v96..v255 itself (clobbers) and reuses the 32 operand registers, 192 in all, which leaves
the compiler 64 for what lives across the step (every spill around it costs a wave an L2 round trip).
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_sort_network_inplace as IP  # noqa: E402

# tight crossing windows of the window-form rebin (kernels.hip rb_lo/rb_hi<true>; tools/gen_rebin_windows.py)
RB_LO = [0, 5, 12, 19, 27, 35, 45, 55]
RB_HI = [0, 8, 18, 28, 36, 44, 51, 58]

KEY0, SP0, WA0 = 96, 224, 240
NTEMP = 8                                    # spares of the sort: v224..v239
# before and after the sort the spares are free: the test's temporaries, then the rebin's addresses (v224..v231),
# running sums and scratch
C, IC, TA, TB = (232, 233), (234, 235), (236, 237), (238, 239)
V_LO, V_HI = 96, 255
S_LO, S_HI = 40, 97
S_R, S_C, S_SKIP, S_ROWS, S_T0, S_T1, S_KEEP, S_IDX = 40, 41, 42, 43, 44, 45, 46, 47
S_E, S_W, S_RW = 48, 64, 80
S_A, S_B = 96, 97
GAP_OF = {(8, 0): 0, (8, 1): 2, (8, 2): 4, (8, 3): 6, (16, 0): 1, (16, 1): 5, (32, 0): 3}


def vp(lo):
    return "v[%d:%d]" % (lo, lo + 1)


def key(i):
    return vp(KEY0 + 2 * i)


def keylo(i):
    return "v%d" % (KEY0 + 2 * i)


def sp(j):
    return vp(SP0 + 2 * j)


def wa(u):
    return vp(WA0 + 2 * u)


def addr(u):
    return "v%d" % (SP0 + u)


def spair(lo):
    return "s[%d:%d]" % (lo, lo + 1)


def x(i):
    return "%%%d" % i


def y(j):
    return "%%%d" % (8 + j)


def out(k):
    """The new coefficients come back in the y operand registers (free once the last weights have been used)."""
    return y(k)


def ie(k):
    """I(E_k), k = 1..8, in the x operand registers (dead once the keys are built); E_8's in x_0's."""
    return x(k) if k < 8 else x(0)


def dbits(v):
    b = struct.unpack("<Q", struct.pack("<d", v))[0]
    return b & 0xffffffff, b >> 32


def loc(l):
    return key(l[1]) if l[0] == 'K' else sp(l[1])


class Asm:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)

    def label(self, name):
        self.lines.append("%s%%=:" % name)

    def text(self):
        return self.lines


import re

_REG = re.compile(r"v\[(\d+):(\d+)\]|s\[(\d+):(\d+)\]|\bv(\d+)\b|\bs(\d+)\b|(%\d+)|\b(vcc|scc|exec)\b")
_NO_SCC = ("s_mov_b32", "s_mul_i32", "s_cselect_b32", "s_flbit_i32_b32", "s_load_dwordx16")


def regs_of(tok):
    out = []
    for m in _REG.finditer(tok):
        if m.group(1):
            out += ["v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)]
        elif m.group(3):
            out += ["s%d" % i for i in range(int(m.group(3)), int(m.group(4)) + 1)]
        elif m.group(5):
            out.append("v" + m.group(5))
        elif m.group(6):
            out.append("s" + m.group(6))
        elif m.group(7):
            out.append(m.group(7))
        else:
            out.append(m.group(8))
    return out


def defs_uses(line):
    op, _, rest = line.partition(" ")
    toks = [t.strip() for t in rest.split(",")]
    if op.startswith("s_cmp") or op.startswith("s_bitcmp"):
        return ["scc"], [r for t in toks for r in regs_of(t)]
    d = regs_of(toks[0])
    u = [r for t in toks[1:] for r in regs_of(t)]
    if op.startswith("s_") and op not in _NO_SCC:
        d = d + ["scc"]
    if op in ("s_cselect_b32", "s_addc_u32"):
        u = u + ["scc"]
    return d, u


def is_barrier(line):
    return line.endswith(":") or line.startswith(("s_cbranch", "s_branch", "s_waitcnt"))


def schedule_region(lines):
    """List scheduling of a straight-line region: a wave issues a dependent f64 instruction ~2 cycles later than an
    independent one (profiles/r02_ubench_issue_rates.txt: 10.3 against 8.4 ticks), so no instruction should directly
    follow the one that produces its operand."""
    n = len(lines)
    if n < 3:
        return lines, 0
    du = [defs_uses(l) for l in lines]
    preds = [dict() for _ in range(n)]       # pred index -> latency in issue slots
    last_def, last_uses = {}, {}
    last_lds = None
    for i, (d, u) in enumerate(du):
        for r in u:
            if r in last_def:
                j = last_def[r]
                preds[i][j] = max(preds[i].get(j, 0), 2 if lines[j].startswith("v_") else 2)
        for r in d:
            if r in last_def:
                preds[i][last_def[r]] = max(preds[i].get(last_def[r], 0), 1)
            for j in last_uses.get(r, ()):
                if j != i:
                    preds[i][j] = max(preds[i].get(j, 0), 1)
        if lines[i].startswith("ds_"):
            if last_lds is not None:
                preds[i][last_lds] = max(preds[i].get(last_lds, 0), 1)
            last_lds = i
        for r in u:
            last_uses.setdefault(r, []).append(i)
        for r in d:
            last_def[r] = i
            last_uses[r] = []
    succs = [[] for _ in range(n)]
    for i in range(n):
        for j, lat in preds[i].items():
            succs[j].append((i, lat))
    crit = [1] * n
    for i in range(n - 1, -1, -1):
        for k, lat in succs[i]:
            crit[i] = max(crit[i], crit[k] + lat)
    npred = [len(p) for p in preds]
    earliest = [0] * n
    ready = [i for i in range(n) if npred[i] == 0]
    out, slot, stalls = [], 0, 0
    while ready:
        ok = [i for i in ready if earliest[i] <= slot]
        if ok:
            i = max(ok, key=lambda k: (crit[k], -k))
        else:
            i = min(ready, key=lambda k: (earliest[k], -crit[k]))
            stalls += earliest[i] - slot
            slot = earliest[i]
        ready.remove(i)
        out.append(lines[i])
        for k, lat in succs[i]:
            earliest[k] = max(earliest[k], slot + lat)
            npred[k] -= 1
            if npred[k] == 0:
                ready.append(k)
        slot += 1
    assert len(out) == n
    return out, stalls


def schedule(lines):
    out, region, stalls, before = [], [], 0, 0

    def count_adjacent(ls):
        c = 0
        for a_, b_ in zip(ls, ls[1:]):
            da, _ = defs_uses(a_)
            _, ub = defs_uses(b_)
            if set(da) & set(ub) - {"scc"}:
                c += 1
        return c
    for l in lines + ["END:"]:
        if is_barrier(l):
            before += count_adjacent(region)
            r, st = schedule_region(region)
            stalls += count_adjacent(r)
            out += r
            if l != "END:":
                out.append(l)
            region = []
        else:
            region.append(l)
    return out, before, stalls


EXPERIMENT = os.environ.get("CLIMA_RORR_EXPERIMENT", "")   # timing experiments only: "noskip" (tests run, nothing left out), "notest"
DEBUG = False     # the diagnostic variant (-DCLIMA_STAMPS): s_memtime at the phase boundaries, handed out with rows / skip mask


def build():
    a = Asm()
    L = lambda n: "%s%%=" % n   # noqa: E731  (a label unique to each copy of the block)
    if DEBUG:
        a("s_memtime s[98:99]")
    # ---- the handle's tables into scalar registers (waited for before the rebin)
    a("s_load_dwordx16 s[%d:%d], %%17, 0x0" % (S_E, S_E + 15))
    a("s_load_dwordx16 s[%d:%d], %%17, 0x40" % (S_W, S_W + 15))
    a("s_load_dwordx16 s[%d:%d], %%17, 0x80" % (S_RW, S_RW + 15))
    # ---- which view: rows (range of y against the gaps of x) when y's range is the smaller one in most lanes, else columns
    lo, hi = dbits(2.0 ** -40)
    a("s_mov_b32 s%d, 0x%x" % (S_T0, lo))
    a("s_mov_b32 s%d, 0x%x" % (S_T1, hi))
    a("s_mov_b32 s%d, 0xfffffe07" % S_KEEP)
    a("s_mov_b32 s%d, 0" % S_SKIP)
    a("v_add_f64 %s, %s, %s" % (vp(TA[0]), x(7), y(7)))
    a("v_add_f64 %s, %s, -%s" % (vp(TB[0]), y(7), y(0)))                             # range of y
    a("v_add_f64 %s, %s, -%s" % (vp(C[0]), x(7), x(0)))                              # range of x
    a("v_mul_f64 %s, %s, %s" % (vp(TA[0]), vp(TA[0]), spair(S_T0)))                 # margin 2^-40 (x_7 + y_7)
    if EXPERIMENT != "notest":
        a("v_cmp_le_f64 vcc, %s, %s" % (vp(TB[0]), vp(C[0])))
        a("s_bcnt1_i32_b64 s%d, vcc" % S_A)
        a("s_cmp_lt_u32 s%d, 32" % S_A)
        a("s_cbranch_scc1 %s" % L("Lcol"))

    def view(xs, ys, rng, swapped):
        """gap tests of one view (xs: the operand whose gaps are tested, rng: the other operand's range), then the keys.
        S_SKIP collects one bit per gap, gap g in bit 6-g (each test shifts the mask left and adds its result)."""
        a("v_add_f64 %s, %s, %s" % (vp(rng[0]), vp(rng[0]), vp(TA[0])))             # threshold: range + margin
        tmp = [IC, (WA0, WA0 + 1)]
        for g in range(7 if EXPERIMENT != "notest" else 0):
            t = tmp[g % 2]
            a("v_add_f64 %s, %s, -%s" % (vp(t[0]), xs(g + 1), xs(g)))
            a("v_cmp_ge_f64 vcc, %s, %s" % (vp(t[0]), vp(rng[0])))
            a("s_cmp_eq_u64 vcc, exec")
            a("s_addc_u32 s%d, s%d, s%d" % (S_SKIP, S_SKIP, S_SKIP))
        # rows the sort / rebin have to deal with: 0 if every gap holds, else (highest failing gap) + 2 = 8 - (lowest clear bit)
        a("s_andn2_b32 s%d, 0x7f, s%d" % (S_A, S_SKIP))
        a("s_ff1_i32_b32 s%d, s%d" % (S_B, S_A))
        a("s_sub_i32 s%d, 8, s%d" % (S_ROWS, S_B))
        a("s_cmp_eq_u32 s%d, 0" % S_A)
        a("s_cselect_b32 s%d, 0, s%d" % (S_ROWS, S_ROWS))
        # the 64 keys: key(position 8a+b) = x_a + y_b (row view) or x_b + y_a (column view), the PAIR's index 8i+j in
        # mantissa bits 3-8 either way (KEY_IDX_MASK)
        for p in range(64):
            aa, bb = divmod(p, 8)
            i, j = (bb, aa) if swapped else (aa, bb)
            a("v_add_f64 %s, %s, %s" % (key(p), x(i), y(j)))
        for p in range(64):
            aa, bb = divmod(p, 8)
            i, j = (bb, aa) if swapped else (aa, bb)
            idx = (i * 8 + j) << 3
            if idx <= 64:
                a("v_and_or_b32 %s, %s, s%d, %d" % (keylo(p), keylo(p), S_KEEP, idx))
            else:
                a("v_and_b32 %s, 0xfffffe07, %s" % (keylo(p), keylo(p)))
                a("v_or_b32 %s, 0x%x, %s" % (keylo(p), idx, keylo(p)))
    view(x, y, TB, False)
    if EXPERIMENT in ("noskip", "notest"):
        a("s_mov_b32 s%d, 0" % S_SKIP)
        a("s_mov_b32 s%d, 8" % S_ROWS)
    a.label("Lbuilt")        # (the column view is at the end of the block and comes back here)
    if DEBUG:
        a("s_memtime s[100:101]")
    # ---- the sort: per-merge pruned odd-even merges as in-place exchanges; a merge whose gap holds is jumped over
    nce = nmov = 0
    for p in (8, 16, 32):
        net, _ = IP.pruned_merge(p)
        for m in range(64 // (2 * p)):
            ops, peak, moves = IP.orient(IP.shift(net, 2 * p * m), NTEMP, seed=p * 10 + m)
            assert peak <= NTEMP
            # check the block by itself
            import random
            rnd = random.Random(p + m)
            for _ in range(50):
                v = [rnd.random() for _ in range(64)]
                for h in (0, 1):
                    s0 = 2 * p * m + h * p
                    v[s0:s0 + p] = sorted(v[s0:s0 + p])
                got = IP.simulate(ops, v, NTEMP)
                ref = list(v)
                for (ca, cb) in IP.shift(net, 2 * p * m):
                    if ref[ca] > ref[cb]:
                        ref[ca], ref[cb] = ref[cb], ref[ca]
                assert got == ref
            lab = "Lm%d_%d" % (p, m)
            a("s_bitcmp1_b32 s%d, %d" % (S_SKIP, 6 - GAP_OF[(p, m)]))
            a("s_cbranch_scc1 %s" % L(lab))
            for op, dst, s1, s2 in ops:
                if op == 'MOV':
                    a("v_mov_b64 %s, %s" % (loc(dst), loc(s1)))
                    nmov += 1
                elif op == 'LO':
                    a("v_min_f64 %s, %s, %s" % (loc(dst), loc(s1), loc(s2)))
                    a("v_max_f64 %s, %s, %s" % (loc(s2), loc(s1), loc(s2)))
                    nce += 1
                else:
                    a("v_max_f64 %s, %s, %s" % (loc(dst), loc(s1), loc(s2)))
                    a("v_min_f64 %s, %s, %s" % (loc(s1), loc(s1), loc(s2)))
                    nce += 1
            a.label(lab)
    # ---- the rebin (window form): I(E_k) = max over the elements j of the window of IC_(j-1) + v_j (E_k - C_(j-1))
    if DEBUG:
        a("s_memtime s[40:41]")
    a("s_waitcnt lgkmcnt(0)")
    a("s_cmp_eq_u32 s%d, 0" % S_ROWS)
    a("s_cbranch_scc1 %s" % L("Louts"))
    lo, hi = dbits(-1.0e300)
    a("s_mov_b32 s%d, 0x%x" % (S_T0, lo))
    a("s_mov_b32 s%d, 0x%x" % (S_T1, hi))
    a("s_mov_b32 s%d, 0x1f8" % S_IDX)
    a("v_mov_b64 %s, 0" % vp(C[0]))
    a("v_mov_b64 %s, 0" % vp(IC[0]))
    for k in range(1, 8):
        a("v_mov_b64 %s, %s" % (ie(k), spair(S_T0)))

    def wset(b, u):          # weights of batch b: even batches in the fixed registers, odd ones in the y operands
        return wa(u) if b % 2 == 0 else y(u)

    def prefetch(b):
        for u in range(8):
            a("v_and_or_b32 %s, %s, s%d, %%16" % (addr(u), keylo(8 * b + u), S_IDX))
        for u in range(8):
            a("ds_read_b64 %s, %s" % (wset(b, u), addr(u)))
    prefetch(0)
    tt = [TA, TB]
    nt = 0
    for b in range(8):
        if b > 0:
            a("s_cmp_le_u32 s%d, %d" % (S_ROWS, b))
            a("s_cbranch_scc1 %s" % L("Lexit%d" % b))
        if b < 7:
            prefetch(b + 1)        # (also when this is the last row to rebin: a branch around it costs more than the 16 instructions)
            a("s_waitcnt lgkmcnt(8)")
        else:
            a("s_waitcnt lgkmcnt(0)")
        for u in range(8):
            j = 8 * b + u
            for k in range(1, 8):
                if RB_LO[k] <= j <= RB_HI[k]:
                    t = tt[nt % 2]
                    nt += 1
                    a("v_add_f64 %s, %s, -%s" % (vp(t[0]), spair(S_E + 2 * (k - 1)), vp(C[0])))
                    a("v_fma_f64 %s, %s, %s, %s" % (vp(t[0]), key(j), vp(t[0]), vp(IC[0])))
                    a("v_max_f64 %s, %s, %s" % (ie(k), ie(k), vp(t[0])))
            a("v_fma_f64 %s, %s, %s, %s" % (vp(IC[0]), key(j), wset(b, u), vp(IC[0])))
            a("v_add_f64 %s, %s, %s" % (vp(C[0]), vp(C[0]), wset(b, u)))
    a("v_mov_b64 %s, %s" % (ie(8), vp(IC[0])))
    a("s_branch %s" % L("Louts"))
    for b in range(1, 8):      # the batch loop ended after row b-1: I(E_b) is the integral so far
        a.label("Lexit%d" % b)
        a("v_mov_b64 %s, %s" % (ie(b), vp(IC[0])))
        if b < 7:
            a("s_branch %s" % L("Louts"))
    # ---- new coefficients: difference quotients of I for the rows that were rebinned, the weighted row sum for the others
    a.label("Louts")
    if DEBUG:
        a("s_memtime s[44:45]")
    a("s_waitcnt lgkmcnt(0)")     # (weights requested ahead for a row that was not rebinned may still be on their way)
    # rows r..7 stand alone: weighted row sums, from row 7 down to the first row that was rebinned
    for k in range(7, -1, -1):
        a("s_cmp_gt_u32 s%d, %d" % (S_ROWS, k))
        a("s_cbranch_scc1 %s" % L("Lopen%d" % k))
        a("v_mul_f64 %s, %s, %s" % (vp(TA[0]), key(8 * k), spair(S_W)))
        for j in range(1, 7):
            a("v_fma_f64 %s, %s, %s, %s" % (vp(TA[0]), key(8 * k + j), spair(S_W + 2 * j), vp(TA[0])))
        a("v_fma_f64 %s, %s, %s, %s" % (out(k), key(8 * k + 7), spair(S_W + 14), vp(TA[0])))
    a("s_branch %s" % L("Ldone"))
    # rows 0..k were rebinned: difference quotients of I, falling through from row k to row 0
    for k in range(7, -1, -1):
        a.label("Lopen%d" % k)
        if k == 0:
            a("v_mul_f64 %s, %s, %s" % (out(0), ie(1), spair(S_RW)))
        else:
            a("v_add_f64 %s, %s, -%s" % (vp(TA[0]), ie(k + 1), ie(k)))
            a("v_mul_f64 %s, %s, %s" % (out(k), vp(TA[0]), spair(S_RW + 2 * k)))
    a("s_branch %s" % L("Ldone"))
    a.label("Lcol")
    view(y, x, C, True)
    a("s_branch %s" % L("Lbuilt"))
    a.label("Ldone")
    if DEBUG:
        a("s_memtime s[46:47]")
        a("s_waitcnt lgkmcnt(0)")
    lines, before, after = schedule(a.text())
    print("instructions that directly follow their operand's producer: %d before scheduling, %d after" % (before, after))
    return lines, nce, nmov


def main():
    global DEBUG
    emit(False, "rorr_xys_asm.inc", "RORR_XYS_ASM")
    if os.environ.get("CLIMA_RORR_DEBUG") == "1":   # (a variant with s_memtime samples at the phase boundaries: not used by any build)
        DEBUG = True
        emit(True, "rorr_xys_asm_dbg.inc", "RORR_XYS_ASM_DBG")


def emit(dbg, fname, macro):
    lines, nce, nmov = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clima_amd", "csrc", fname)
    clob = ["v%d" % i for i in range(V_LO, V_HI + 1)] + ["s%d" % i for i in range(S_LO, (101 if dbg else S_HI) + 1)] + ["vcc", "scc"]
    with open(path, "w") as f:
        f.write("// Generated by tools/gen_rorr_asm.py -- do not edit.  One random-overlap mixing step (8 g-points, x and y\n")
        f.write("// ascending) as a block of gfx950 assembly: %d instructions, sort %d exchanges + %d moves.\n" % (
            sum(1 for l in lines if not l.endswith(":")), nce, nmov))
        f.write("#define %s_TEXT \\\n" % macro)
        for l in lines:
            f.write('  "%s\\n\\t" \\\n' % l)
        f.write('  ""\n')
        f.write("#define %s_CLOBBERS %s\n" % (macro, ", ".join('"%s"' % c for c in clob)))
    print("%d lines, %d exchanges, %d moves -> %s" % (len(lines), nce, nmov, os.path.abspath(path)))


if __name__ == "__main__":
    main()
