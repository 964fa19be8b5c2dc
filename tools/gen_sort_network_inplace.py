#!/usr/bin/env python3
"""Generate clima_amd/csrc/sort_network_64_ip.inc: the register sort of the random-overlap mixing step
(reference: k_rorr, src/radtran/clima_radtran_types.f90:826-852) as IN-PLACE compare-exchanges, grouped by merge.

Why in place.  A compare-exchange is two instructions, v_min_f64 + v_max_f64, and one of its two results needs a
register that is neither operand's (both instructions read both operands).  Written as `lo = min(a, b); hi = max(a, b)`
the compiler rotates registers freely, which is fine in straight-line code -- but a merge that can be LEFT OUT (its two
runs are known not to interleave, kernels.hip rorr_mix8) is a conditional block, and at the join behind it every key
must sit in the same register on both paths: hipcc repairs the rotation with a storm of v_mov_b64 + scratch spills
(a 2.7x slower kernel).  Here every exchange names its registers: the key that stays is updated where it is
(`v_max_f64 b, a, b`), the key that moves goes to a named free register -- a spare T(j), or its own home K(i) when it
is coming back -- and every block ends with all 64 keys at home, so a block that is skipped and a block that ran leave
the same layout and the join costs nothing.

  CE_LO(dst, a, b):  dst = min(a, b);  b = max(a, b)      (key a moves to dst; a's old register is free)
  CE_HI(dst, a, b):  dst = max(a, b);  a = min(a, b)      (key b moves to dst)
  CE_MOV(dst, src):  dst = src                             (only where the parities leave a key displaced)

Sections (the includer defines exactly one of them, and K(i), T(j), CE_MERGE_BEGIN(level, merge), CE_MERGE_END):
  CE_IP_XYS      x and y both ascending: 8 sorted runs of 8, elementwise ordered.  Merge levels 8, 16, 32 without
                 their first stage; every merge is pruned BY ITSELF against its own input class (all monotone 0/1
                 matrices of its rows, halves sorted), so it is a complete merge whatever later levels do -- which is
                 what leaving out single merges needs (the whole-network pruning of round 2 let level 16 lean on
                 level 32 for 6 exchanges).
  CE_IP_HEAD     the 8 runs themselves (only when y does not ascend).
  CE_IP_GENERAL  merge levels 8, 16, 32 in full (x or y not ascending).
"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_sort_network as G  # noqa: E402

N, RUN = 64, 8
NTEMP = int(os.environ.get("CLIMA_SORT_NTEMP", "4"))


def merge_class_inputs(rows):
    """0/1 inputs of a merge of two runs of rows/2 rows each: monotone 0/1 matrices (rows x 8), each half sorted."""
    M = G.monotone01(8)            # 8x8, row-major; take the first `rows` rows: every monotone rows x 8 matrix occurs
    M = np.unique(M[:, :rows * 8], axis=0)
    half = rows * 4
    out = np.concatenate([np.sort(M[:, :half], axis=1), np.sort(M[:, half:], axis=1)], axis=1)
    return np.unique(out, axis=0)


def sorts(net, V):
    V = V.copy()
    for a, b in net:
        lo = np.minimum(V[:, a], V[:, b])
        hi = np.maximum(V[:, a], V[:, b])
        V[:, a] = lo
        V[:, b] = hi
    return bool(np.all(V[:, :-1] <= V[:, 1:]))


def level_merge(p, first_stage):
    """Exchanges of ONE merge of level p (two runs of p keys, wires 0..2p-1) from Knuth's merge exchange."""
    out = []
    for pp, k, ces in G.stages(N):
        if pp != p or (k == p and not first_stage):
            continue
        out += [(a, b) for a, b in ces if b < 2 * p]
    return out


def pruned_merge(p):
    net = level_merge(p, False)
    V = merge_class_inputs(2 * p // RUN)
    assert sorts(net, V), "level %d without its first stage does not merge its class" % p
    alive = [True] * len(net)
    for i in range(len(net) - 1, -1, -1):
        alive[i] = False
        if not sorts([c for c, a in zip(net, alive) if a], V):
            alive[i] = True
    out = [c for c, a in zip(net, alive) if a]
    assert sorts(out, V)
    return out, len(V)


def orient(block, ntemp, tries=400, seed=1):
    """Schedule one block of exchanges (all on keys at home before and after) as in-place operations.
    Returns (ops, peak temps, moves): ops = ('LO'|'HI'|'MOV', dst, a, b) with locations ('K', i) / ('T', j)."""
    rnd = random.Random(seed)
    n = len(block)
    # per key: the exchanges it takes part in, in order
    uses = {}
    for idx, (a, b) in enumerate(block):
        uses.setdefault(a, []).append(idx)
        uses.setdefault(b, []).append(idx)
    preds = [set() for _ in range(n)]
    for k, lst in uses.items():
        for u, v in zip(lst, lst[1:]):
            preds[v].add(u)
    best = None
    for t in range(tries):
        loc = {}                       # key -> temp index while displaced
        free = list(range(ntemp))
        done = [False] * n
        npred = [len(p) for p in preds]
        succs = [[] for _ in range(n)]
        for v in range(n):
            for u in preds[v]:
                succs[u].append(v)
        ready = [i for i in range(n) if npred[i] == 0]
        pos = {k: 0 for k in uses}     # next use index per key
        ops, moves, peak, fail = [], 0, 0, False
        order_noise = 0 if t == 0 else rnd.choice([0, 2, 6, 16])
        while ready:
            # prefer exchanges that bring a displaced key home (they free a spare); then the original order
            def prio(i):
                a, b = block[i]
                back = (a in loc) + (b in loc)
                return (-back, i + (rnd.random() * order_noise if order_noise else 0))
            ready.sort(key=prio)
            pick = None
            for i in ready:
                a, b = block[i]
                if a in loc or b in loc or len(free) > 1:      # (the last spare is kept for exchanges with no way back)
                    pick = i
                    break
            forced = pick is None
            if forced:
                pick = ready[0]
            i = pick
            ready.remove(i)
            a, b = block[i]
            la = ('T', loc[a]) if a in loc else ('K', a)
            lb = ('T', loc[b]) if b in loc else ('K', b)
            last_a = uses[a][-1] == i or forced
            last_b = uses[b][-1] == i or forced
            if a in loc and b in loc:
                # one of them goes home; prefer the one at its last use (it must end at home)
                mover = a if (last_a or not last_b) else b
                if t and not last_a and not last_b and rnd.random() < 0.5:
                    mover = b
            elif a in loc:
                mover = a
            elif b in loc:
                mover = b
            else:
                # both at home: one must leave; it needs a later exchange to come back with
                if last_a and last_b:
                    mover = None
                elif last_a:
                    mover = b
                elif last_b:
                    mover = a
                else:
                    mover = a if (t == 0 or rnd.random() < 0.5) else b
            if mover is None:
                # no way back: exchange through a spare and move (3 instructions)
                j = free[-1]
                ops.append(('LO', ('T', j), la, lb))
                ops.append(('MOV', ('K', a), ('T', j), None))
                moves += 1
                peak = max(peak, ntemp - len(free) + 1)
            else:
                if mover in loc:
                    j = loc.pop(mover)
                    dst = ('K', mover)
                    free.append(j)
                else:
                    j = free.pop()
                    loc[mover] = j
                    dst = ('T', j)
                    peak = max(peak, ntemp - len(free))
                ops.append(('LO' if mover == a else 'HI', dst, la, lb))
            done[i] = True
            for v in succs[i]:
                npred[v] -= 1
                if npred[v] == 0:
                    ready.append(v)
        if fail:
            continue
        for k, j in sorted(loc.items()):
            ops.append(('MOV', ('K', k), ('T', j), None))
            moves += 1
        cand = (moves, peak, ops)
        if best is None or cand[:2] < best[:2]:
            best = cand
            if moves == 0:
                break
    assert best is not None, "no schedule with %d spares" % ntemp
    assert all(l is None or l[0] == 'K' or l[1] < ntemp for op in best[2] for l in op[1:])
    return best[2], best[1], best[0]


def simulate(ops, vals, ntemp):
    """Run in-place ops on a dict of key values; returns the 64 values at home."""
    K = list(vals)
    T = [None] * ntemp

    def get(l):
        return K[l[1]] if l[0] == 'K' else T[l[1]]

    def put(l, v):
        if l[0] == 'K':
            K[l[1]] = v
        else:
            T[l[1]] = v
    for op, dst, a, b in ops:
        if op == 'MOV':
            put(dst, get(a))
        else:
            va, vb = get(a), get(b)
            lo, hi = min(va, vb), max(va, vb)
            if op == 'LO':
                put(dst, lo)
                put(b, hi)
            else:
                put(dst, hi)
                put(a, lo)
    return K


def fmt(l):
    return "%s(%d)" % (l[0], l[1])


def emit(f, ops):
    for op, dst, a, b in ops:
        if op == 'MOV':
            f.write("CE_MOV(%s,%s)\n" % (fmt(dst), fmt(a)))
        else:
            f.write("CE_%s(%s,%s,%s)\n" % (op, fmt(dst), fmt(a), fmt(b)))


def shift(net, off):
    return [(a + off, b + off) for a, b in net]


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clima_amd", "csrc", "sort_network_64_ip.inc")
    rnd = random.Random(7)
    stats = []
    sections = {}
    # ---- x and y ascending: per-merge pruned, first stages left out
    xys_blocks = []
    for p in (8, 16, 32):
        net, ncases = pruned_merge(p)
        stats.append("level %d: %d exchanges per merge (%d without pruning), complete on its %d class inputs"
                     % (p, len(net), len(level_merge(p, False)), ncases))
        for m in range(N // (2 * p)):
            xys_blocks.append((p, m, shift(net, 2 * p * m)))
    sections["CE_IP_XYS"] = xys_blocks
    # ---- the runs themselves, and the full merge levels
    head = [(a, b) for pp, k, ces in G.stages(N) if pp < RUN for a, b in ces]
    sections["CE_IP_HEAD"] = [(0, 0, head)]
    sections["CE_IP_GENERAL"] = [(p, m, shift(level_merge(p, True), 2 * p * m)) for p in (8, 16, 32) for m in range(N // (2 * p))]

    text = {}
    total_moves = {}
    peak_all = 0
    for name, blocks in sections.items():
        lines = []
        total_moves[name] = 0
        for p, m, net in blocks:
            ops, peak, moves = orient(net, NTEMP, seed=p * 10 + m)
            peak_all = max(peak_all, peak)
            total_moves[name] += moves
            lines.append((p, m, ops, len(net)))
        text[name] = lines

    # ---- checks: every section sorts what it must, through the in-place ops, with every combination of left-out merges
    def run_section(name, vals, skip=()):
        K = list(vals)
        for p, m, ops, _ in text[name]:
            if (p, m) in skip:
                continue
            K = simulate(ops, K, NTEMP)
        return K
    for _ in range(300):          # general path on arbitrary keys
        v = [rnd.random() if rnd.random() < 0.7 else round(rnd.random(), 1) for _ in range(N)]
        r = run_section("CE_IP_GENERAL", run_section("CE_IP_HEAD", v))
        assert r == sorted(v)
    gap_of = {(8, 0): 0, (8, 1): 2, (8, 2): 4, (8, 3): 6, (16, 0): 1, (16, 1): 5, (32, 0): 3}
    for trial in range(3000):     # x, y ascending; merges left out wherever their gap allows it (and random subsets of those)
        x = sorted(rnd.choice([rnd.random(), round(rnd.random(), 1), rnd.random() * 10 ** rnd.randint(-3, 3)]) for _ in range(8))
        y = sorted(rnd.choice([rnd.random(), round(rnd.random(), 1), rnd.random() * 10 ** rnd.randint(-3, 3)]) for _ in range(8))
        v = [x[i] + y[j] for i in range(8) for j in range(8)]
        allowed = [pm for pm, g in gap_of.items() if x[g] + y[7] <= x[g + 1] + y[0]]
        skip = set(pm for pm in allowed if trial % 3 == 0 or rnd.random() < 0.5)
        assert run_section("CE_IP_XYS", v, skip) == sorted(v), (x, y, skip)
    # exhaustive 0/1: every monotone matrix, every set of merges its gaps allow left out
    M01 = G.monotone01(8)
    R = M01.reshape(-1, 8, 8)
    sep = np.stack([~((R[:, g, :].max(axis=1) == 1) & (R[:, g + 1, :].min(axis=1) == 0)) for g in range(7)], axis=1)
    plain = {(p, m): [(('K', 0), 0)] for p, m, _, _ in text["CE_IP_XYS"]}
    nets = {(p, m): [c for c in blk] for p, m, blk in sections["CE_IP_XYS"]}
    for S in range(128):
        ok = np.ones(len(M01), bool)
        for g in range(7):
            if S >> g & 1:
                ok &= sep[:, g]
        V = M01[ok].copy()
        for (p, m, net) in sections["CE_IP_XYS"]:
            if S >> gap_of[(p, m)] & 1:
                continue
            for a, b in net:
                lo = np.minimum(V[:, a], V[:, b]); hi = np.maximum(V[:, a], V[:, b])
                V[:, a] = lo; V[:, b] = hi
        assert bool(np.all(V[:, :-1] <= V[:, 1:])), "gap set %s" % bin(S)
    del plain, nets

    with open(out, "w") as f:
        f.write("// Generated by tools/gen_sort_network_inplace.py -- do not edit.\n")
        f.write("// In-place compare-exchanges for the 64-key register sort, grouped by merge; %d spare registers T(j).\n" % NTEMP)
        f.write("//   CE_LO(dst,a,b): dst = min(a,b), b = max(a,b)    CE_HI(dst,a,b): dst = max(a,b), a = min(a,b)    CE_MOV(dst,src)\n")
        f.write("// Every block starts and ends with all keys in K(0..63).\n")
        for s in stats:
            f.write("//   %s\n" % s)
        for name in ("CE_IP_XYS", "CE_IP_HEAD", "CE_IP_GENERAL"):
            nce = sum(n for _, _, _, n in text[name])
            f.write("//   %-13s: %d exchanges, %d extra moves\n" % (name, nce, total_moves[name]))
        for name in ("CE_IP_XYS", "CE_IP_HEAD", "CE_IP_GENERAL"):
            f.write("#ifdef %s\n" % name)
            for p, m, ops, _ in text[name]:
                if name == "CE_IP_HEAD":
                    emit(f, ops)
                else:
                    f.write("CE_MERGE_BEGIN(%d,%d)\n" % (p, m))
                    emit(f, ops)
                    f.write("CE_MERGE_END\n")
            f.write("#endif\n")
    for s in stats:
        print(s)
    for name in text:
        print("%-13s %4d exchanges, %d extra moves" % (name, sum(n for _, _, _, n in text[name]), total_moves[name]))
    print("peak spares in use:", peak_all, "of", NTEMP, "->", os.path.abspath(out))


if __name__ == "__main__":
    main()
