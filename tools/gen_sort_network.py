#!/usr/bin/env python3
"""Generate the compare-exchange list of Batcher's odd-even merge sort for n keys.

The random-overlap resort step (reference: k_rorr, src/radtran/clima_radtran_types.f90:832-852)
sorts ng*ng = 64 optical depths per (layer, bin, species).  On the GPU each lane owns one
(layer, bin) and keeps its 64 keys in registers, so the sort must be a fixed network with
compile-time indices: every compare-exchange is one v_min_f64 + one v_max_f64.

Writes clima_amd/csrc/sort_network_<n>.inc containing `CE(a,b)` lines (a < b), and
`CE_MERGE(a,b)` lines for the sub-network that only merges n/ng pre-sorted runs of ng keys.
"""
import os
import random
import sys


def oddeven_merge_sort(n, presorted_run=1):
    """Knuth TAOCP 5.2.2 Algorithm M (merge exchange), n a power of two.
    With presorted_run = r, stages with p < r are skipped (runs of r already sorted)."""
    ces = []
    p = 1
    while p < n:
        if p >= presorted_run:
            k = p
            while k >= 1:
                j = k % p
                while j <= n - 1 - k:
                    for i in range(0, min(k - 1, n - j - k - 1) + 1):
                        if (i + j) // (p * 2) == (i + j + k) // (p * 2):
                            ces.append((i + j, i + j + k))
                    j += 2 * k
                k //= 2
        p *= 2
    return ces


def check(n, ces, presorted_run=1, trials=2000):
    rnd = random.Random(1)
    for _ in range(trials):
        v = [rnd.choice([0, 1]) if rnd.random() < 0.5 else rnd.random() for _ in range(n)]
        if presorted_run > 1:
            for s in range(0, n, presorted_run):
                v[s:s + presorted_run] = sorted(v[s:s + presorted_run])
        ref = sorted(v)
        for a, b in ces:
            if v[a] > v[b]:
                v[a], v[b] = v[b], v[a]
        assert v == ref


def stages(n):
    """Knuth merge exchange as a list of (p, k, [compare-exchanges]) stages."""
    out = []
    p = 1
    while p < n:
        k = p
        while k >= 1:
            ces = []
            j = k % p
            while j <= n - 1 - k:
                for i in range(0, min(k - 1, n - j - k - 1) + 1):
                    if (i + j) // (p * 2) == (i + j + k) // (p * 2):
                        ces.append((i + j, i + j + k))
                j += 2 * k
            out.append((p, k, ces))
            k //= 2
        p *= 2
    return out


def check_tableau(n, run, st, trials=5000):
    """x and y both ascending: the k == p stage of every merge level is redundant."""
    rnd = random.Random(2)
    m = n // run
    for _ in range(trials):
        x = sorted(rnd.choice([rnd.random(), round(rnd.random(), 1)]) for _ in range(m))
        y = sorted(rnd.choice([rnd.random(), round(rnd.random(), 1)]) for _ in range(run))
        v = [x[i] + y[j] for i in range(m) for j in range(run)]
        for p, k, ces in st:
            if p < run or k == p:
                continue
            for a, b in ces:
                if v[a] > v[b]:
                    v[a], v[b] = v[b], v[a]
        assert v == sorted(v)


def monotone01(run):
    """All 0/1 matrices (run x run, row-major = key order i*run+j) that are non-decreasing along
    rows and columns: C(2*run, run) of them (12 870 for run = 8)."""
    import numpy as np
    out = []

    def rec(i, prev, cs):
        if i == run:
            out.append([1 if j >= cs[r] else 0 for r in range(run) for j in range(run)])
            return
        for c in range(0, prev + 1):
            rec(i + 1, c, cs + [c])
    rec(0, run, [])
    return np.array(out, dtype=np.uint8)


def sorts_all(net, M):
    import numpy as np
    V = M.copy()
    for a, b in net:
        lo = np.minimum(V[:, a], V[:, b])
        hi = np.maximum(V[:, a], V[:, b])
        V[:, a] = lo
        V[:, b] = hi
    return bool(np.all(V[:, :-1] <= V[:, 1:]))


def tableau_prune(n, run, st):
    """Which exchanges of the merge levels (first stages already left out) can be dropped when the
    keys are x_i + y_j with x and y both ascending, i.e. a matrix sorted along rows and columns.
    A comparison network sorts every such real matrix iff it sorts every such 0/1 matrix
    (thresholding commutes with compare-exchange) and there are only C(2 run, run) of those, so
    every step of the greedy pruning (last exchange to first) is checked exhaustively.
    Returns {(p, k, position in stage): True if droppable}."""
    assert n == run * run
    M = monotone01(run)
    items = [(p, k, i, ce) for p, k, ces in st if p >= run and k != p for i, ce in enumerate(ces)]
    assert sorts_all([it[3] for it in items], M)
    alive = [True] * len(items)
    for idx in range(len(items) - 1, -1, -1):
        alive[idx] = False
        if not sorts_all([it[3] for it, a in zip(items, alive) if a], M):
            alive[idx] = True
    assert sorts_all([it[3] for it, a in zip(items, alive) if a], M)
    return {(p, k, i): (not a) for (p, k, i, ce), a in zip(items, alive)}, len(M)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    run = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    full = oddeven_merge_sort(n)
    merge = oddeven_merge_sort(n, run)
    check(n, full)
    check(n, merge, run)
    st = stages(n)
    assert [ce for _, _, c in st for ce in c] == full
    check_tableau(n, run, st)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clima_amd", "csrc",
                       "sort_network_%d.inc" % n)
    nfirst = sum(len(c) for p, k, c in st if p >= run and k == p)
    drop, ncases = tableau_prune(n, run, st) if n == run * run else ({}, 0)
    ndrop = sum(drop.values())
    with open(out, "w") as f:
        f.write("// Generated by tools/gen_sort_network.py %d %d -- do not edit.\n" % (n, run))
        f.write("// Batcher odd-even merge sort (Knuth's merge exchange), %d keys: %d compare-exchanges.\n" % (n, len(full)))
        f.write("//   CE_FULL_HEAD      : %d exchanges that sort the %d runs of %d keys (skip when the runs are sorted)\n"
                % (len(full) - len(merge), n // run, run))
        f.write("//   CE_L<p>_FIRST     : first stage of merge level p (%d exchanges in all); redundant when the keys\n" % nfirst)
        f.write("//                       are x_i + y_j with BOTH x and y ascending (elementwise-ordered runs)\n")
        f.write("//   CE_L<p>_REST      : the remaining stages of level p (%d exchanges in all)\n" % (len(merge) - nfirst))
        f.write("//   CE_X              : %d of those exchanges are never needed when BOTH x and y are ascending (keys sorted\n" % ndrop)
        f.write("//                       along rows and columns): pruned greedily, every step checked on all %d such 0/1\n" % ncases)
        f.write("//                       matrices (zero-one principle for that input class); the includer defines CE_X\n")
        f.write("#ifdef CE_FULL_HEAD\n")
        for p, k, ces in st:
            if p < run:
                for a, b in ces:
                    f.write("CE(%d,%d)\n" % (a, b))
        f.write("#endif\n")
        p = run
        while p < n:
            f.write("#ifdef CE_L%d_FIRST\n" % p)
            for pp, k, ces in st:
                if pp == p and k == p:
                    for a, b in ces:
                        f.write("CE(%d,%d)\n" % (a, b))
            f.write("#endif\n#ifdef CE_L%d_REST\n" % p)
            for pp, k, ces in st:
                if pp == p and k != p:
                    for i, (a, b) in enumerate(ces):
                        f.write("%s(%d,%d)\n" % ("CE_X" if drop.get((pp, k, i)) else "CE", a, b))
            f.write("#endif\n")
            p *= 2
    print("n=%d full=%d merge_tail=%d (first stages %d, droppable for ordered operands %d) -> %s" % (n, len(full), len(merge), nfirst, ndrop, os.path.abspath(out)))


if __name__ == "__main__":
    main()
