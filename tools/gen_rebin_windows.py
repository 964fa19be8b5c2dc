#!/usr/bin/env python3
"""Crossing windows of the window-form rebin (clima_amd/csrc/kernels.hip: rb_lo / rb_hi;
radtran_dev.h: RB_WIN_LO / RB_WIN_HI) for 8 Gauss-Legendre g-point weights on [0, 1].

Element j* of the sorted 64 sums crosses output edge E_k when C_{j*-1} < E_k <= C_{j*}, C being the
running sum of the pair weights in sorted order.  Whatever that order is, C_j lies between the sum
of the j+1 smallest and the j+1 largest pair weights, which confines j* to [jlo, jhi].  The tables in
the kernel are these windows padded by one element on either side."""
import numpy as np


def windows(w):
    w = np.asarray(w, dtype=float)
    wxy = np.sort(np.outer(w, w).ravel())
    lo, hi = np.cumsum(wxy), np.cumsum(wxy[::-1])
    E = np.concatenate([[0.0], np.cumsum(w)])
    out = []
    for k in range(1, len(w)):
        jlo = int(np.searchsorted(hi, E[k] * (1 - 1e-9)))   # first j whose largest-possible C_j reaches E_k
        jhi = int(np.searchsorted(lo, E[k] * (1 + 1e-9)))   # first j whose smallest-possible C_j reaches E_k
        out.append((jlo, min(jhi, len(wxy) - 1)))
    return out


def tight_windows(w):
    """x and y ascending: the sorted order is a linear extension of the product order of the 8x8 grid
    of pairs, so every prefix is a down-set (a Young diagram); bounds over the down-sets of each size."""
    w = np.asarray(w, dtype=float)
    n = len(w)
    W = np.outer(w, w)
    lo, hi = {}, {}

    def rec(i, maxlen, size, wt):
        if i == n:
            lo[size] = min(lo.get(size, 9.0), wt)
            hi[size] = max(hi.get(size, -1.0), wt)
            return
        cs = np.concatenate([[0.0], np.cumsum(W[i])])
        for r in range(maxlen + 1):
            rec(i + 1, r, size + r, wt + cs[r])

    rec(0, n, 0, 0.0)
    E = np.concatenate([[0.0], np.cumsum(w)])
    out = []
    for k in range(1, n):
        jlo = min(j for j in range(n * n) if hi[j + 1] >= E[k] * (1 - 1e-9))
        jhi = min(j for j in range(n * n) if lo[j + 1] >= E[k] * (1 + 1e-9))
        out.append((jlo, jhi))
    return out


if __name__ == "__main__":
    x, w = np.polynomial.legendre.leggauss(8)
    for name, win in (("wide (any order)", windows(w / 2.0)), ("tight (x, y ascending)", tight_windows(w / 2.0))):
        print(name, ":", win, "pairs:", sum(b - a + 1 for a, b in win))
        print("  lo = {0, %s}" % ", ".join(str(a) for a, _ in win))
        print("  hi = {0, %s}" % ", ".join(str(b) for _, b in win))
