"""Work-balanced partition of the opacity bins over ranks.

Spectral bins are independent through opacity assembly and the two-stream solves
(the reference's two OpenMP loops are over bins: src/radtran/clima_radtran_types.f90:638-771,
clima_radtran_radiate.f90:50-158); the only cross-bin step is the frequency integration
(radiate.f90:184-192).  Each rank takes a contiguous range of opacity bins and the run
needs ONE all-reduce of the 4*(nz+1) partial level fluxes.

This is the host-side statement of the rule implemented in csrc/radtran_api.hip
(`compute_shard`); tests check that the two agree.
"""
import bisect

W_OPACITY, W_IR, W_SOLAR_BASE, W_SOLAR_PER_ZENITH = 3.0, 1.0, 0.6, 0.24


def bin_costs(nw, ir_range, sol_range, nzen):
    """Relative cost of every opacity bin: opacity 3, IR solve 1, solar solve 0.6 + 0.24 nzen
    (measured device-time ratios on MI355X, tools/gpu_balance.py).
    ir_range / sol_range are (first, last) opacity-bin indices, inclusive, 0-based."""
    cost = []
    for l in range(nw):
        c = W_OPACITY
        if ir_range[0] <= l <= ir_range[1]:
            c += W_IR
        if sol_range[0] <= l <= sol_range[1]:
            c += W_SOLAR_BASE + W_SOLAR_PER_ZENITH * nzen
        cost.append(c)
    return cost


def bin_shard(nw, ir_range, sol_range, nzen, rank, world):
    """-> (op_lo, op_n, ir_lo, ir_n, sol_lo, sol_n); ir_lo / sol_lo are channel-local."""
    cum = [0.0]
    for c in bin_costs(nw, ir_range, sol_range, nzen):
        cum.append(cum[-1] + c)

    def cut(k):
        if k <= 0:
            return 0
        if k >= world:
            return nw
        return bisect.bisect_left(cum, cum[nw] * k / world)

    lo, hi = (0, nw) if world == 1 else (cut(rank), cut(rank + 1))

    def clip(rng):
        a, b = max(lo, rng[0]), min(hi - 1, rng[1])
        return (0, 0) if b < a else (a - rng[0], b - a + 1)

    ir_lo, ir_n = clip(ir_range)
    sol_lo, sol_n = clip(sol_range)
    return lo, hi - lo, ir_lo, ir_n, sol_lo, sol_n
