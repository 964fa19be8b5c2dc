"""Expected table set of tests/golden/datadir_c/ built WITHOUT clima_amd/data_loader.py, from the
arrays that went into the files (tests/golden/datadir_c_written.npz).  k-tables, axes and channel
edges pass through (rounded to float32 where the file stores float32); everything the reference
regrids onto the bins (types_create.f90:1171-1257, :1407-1468, :734-866: addpnt + inter2 /
interp_discrete_to_bins of futils) is regridded here by an independent routine -- the exact average
over each bin of the piecewise-linear curve through the (padded) points.  Test infrastructure."""
import os

import numpy as np

from clima_amd import synthetic as S

HERE = os.path.dirname(os.path.abspath(__file__))
DATADIR_C = os.path.join(HERE, "golden", "datadir_c")
WRITTEN = os.path.join(HERE, "golden", "datadir_c_written.npz")
LOG10TINY = float(np.log10(np.sqrt(np.finfo(np.float64).tiny)))   # src/clima_const.f90: log10tiny
HUGE = float(np.finfo(np.float64).max)
C_LIGHT = 299792458.0


def bin_average(edges, x, y):
    """Mean over [edges[i], edges[i+1]] of the piecewise-linear curve through (x, y)."""
    out = np.empty(len(edges) - 1)
    for i in range(len(edges) - 1):
        a, b = edges[i], edges[i + 1]
        m = (x > a) & (x < b)
        xs = np.concatenate([[a], x[m], [b]])
        ys = np.concatenate([[np.interp(a, x, y)], y[m], [np.interp(b, x, y)]])
        out[i] = np.sum(0.5 * (ys[1:] + ys[:-1]) * np.diff(xs)) / (b - a)
    return out


def _padded(x, y, pad):
    xx = np.concatenate([[0.0, x[0] * (1 - 1e-4)], x, [x[-1] * (1 + 1e-4), HUGE]])
    return xx, np.concatenate([[pad, pad], y, [pad, pad]])


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def expected_tables():
    from datadir_fixture import RAY_PAR
    d = np.load(WRITTEN)
    g = lambda rel, name: d["file:%s:%s" % (rel, name)]
    sp = [str(s) for s in d["species"]]
    parts = [str(s) for s in d["particles"]]
    t = S.TableSet()
    t.species_names, t.particle_names = tuple(sp), tuple(parts)
    wavl = d["wavl"]
    t.wavl = wavl
    t.ir_wavl = g("kdistributions/bins.h5", "ir_wavl") * 1.0e3
    t.sol_wavl = g("kdistributions/bins.h5", "sol_wavl") * 1.0e3
    # settings.yaml of the fixture: k-distributions of the species that have a file, in species order
    for i, name in enumerate(sp):
        rel = "kdistributions/%s.h5" % name
        if "file:%s:log10k" % rel not in d:
            continue
        k = g(rel, "log10k")
        if name == "CO2":
            k = _f32(k)                                   # stored as float32 in the file
        t.ktables.append(dict(sp_ind=i, weights=g(rel, "weights"), log10P=g(rel, "log10P"), temp=g(rel, "T"), log10k=k))
    # CIA (every pair file; H2O pairs do not exist in the fixture), in the loader's order: the settings say
    # `CIA: true`, i.e. every pair (i <= j in species order) that has a file (types_create.f90:404-447)
    for i in range(len(sp)):
        for j in range(i, len(sp)):
            for name, (a, b) in (("%s-%s" % (sp[i], sp[j]), (i, j)), ("%s-%s" % (sp[j], sp[i]), (j, i))):
                rel = "CIA/%s.h5" % name
                if "file:%s:log10xs" % rel in d:
                    xf = g(rel, "wavelengths") * 1.0e3
                    temp, vals = g(rel, "T"), g(rel, "log10xs")
                    data = np.stack([bin_average(wavl, *_padded(xf, vals[:, q], LOG10TINY)) for q in range(len(temp))], axis=1)
                    t.xsections.append(dict(xs_type=S.XS_CIA, dim=1, sp1=a, sp2=b, temp=temp, data=data))
                    break
    ray = [ln for ln in open(os.path.join(DATADIR_C, "settings.yaml")).read().split("rayleigh: [")[1].split("]")[0].split(", ")]
    for name in ray:
        A, B, Dl = RAY_PAR[name]
        t.xsections.append(dict(xs_type=S.XS_RAYLEIGH, dim=0, sp1=sp.index(name), sp2=-1, temp=None,
                                data=S._rayleigh_vardavas(A, B, Dl, wavl[:-1])))
    for i, name in enumerate(sp):
        rel = "xsections/%s.h5" % name
        if "file:%s:photoabsorption" % rel in d:
            xf, xs = g(rel, "wavelengths"), g(rel, "photoabsorption")
            t.xsections.append(dict(xs_type=S.XS_PHOTOLYSIS, dim=0, sp1=i, sp2=-1, temp=None,
                                    data=10.0 ** bin_average(wavl, *_padded(xf, np.log10(xs), LOG10TINY))))
    rel = "water_continuum/MT_CKD.h5"
    xf, temp = g(rel, "wavelengths") * 1.0e3, g(rel, "T")
    regr = lambda v: np.stack([bin_average(wavl, *_padded(xf, v[:, q], LOG10TINY)) for q in range(len(temp))], axis=1)
    t.continuum = dict(LH2O=sp.index("H2O"), temp=temp, log10_H2O=regr(g(rel, "log10xs_H2O")),
                       log10_foreign=regr(g(rel, "log10xs_foreign")), model="MT_CKD")
    rel = "aerosol_xsections/khare1984/mie_khare1984.h5"
    xf, rad_um = g(rel, "wavelengths"), g(rel, "radii")
    xx = np.concatenate([[0.0], xf, [HUGE]])

    def mie(name):
        v = _f32(g(rel, name))                            # stored as float32 in the file
        return np.stack([bin_average(wavl, xx, np.concatenate([[v[0, q]], v[:, q], [v[-1, q]]])) for q in range(len(rad_um))], axis=1)

    t.particles.append(dict(p_ind=0, radii=rad_um / 1.0e4, w0=mie("w0"), qext=mie("qext"), gt=mie("g0"), dat_name="khare1984"))
    ws, flux = d["star_w"], d["star_f"]
    # the text file holds 11 significant digits
    ws = np.array([float("%.10e" % v) for v in ws])
    flux = np.array([float("%.10e" % v) for v in flux])
    xx = np.concatenate([[0.0, ws[0] * (1 - 1e-4)], ws, [ws[-1] * (1 + 1e-4), HUGE]])
    yy = np.concatenate([[0.0, 0.0], flux, [0.0, 0.0]])
    wav = 0.5 * (t.sol_wavl[:-1] + t.sol_wavl[1:])
    t.photons_sol = bin_average(t.sol_wavl, xx, yy) * (wav * 1e-9 * wav / C_LIGHT)
    return t
