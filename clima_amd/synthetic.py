"""Seeded synthetic opacity tables and atmospheric columns.

The reference's opacity data (`photochem_clima_data` v0.3.1) is a network download that is
absent here, so every benchmark and parity case runs on synthetic tables
with the reference's logical layout:

  * k-distributions  log10k(ngauss, npress, ntemp, nwav) column-major
    (src/radtran/clima_radtran_types_create.f90:1349-1358) == C order [nw][nT][nP][ng]
  * CIA / photolysis / continuum cross-sections regridded to the bin grid: 0-D ``xs[nw]`` or
    1-D ``log10xs[nw][nT]`` (types_create.f90:1171-1257)
  * Rayleigh ``xs[nw]`` from rayleigh_vardavas (src/clima_eqns.f90:240-246)
  * Mie particles ``w0, qext, g [nw][nrad]`` (types_create.f90:734-866)
  * wavelength grid (nm, ascending) and the IR / solar channels as index sub-ranges of it
    (types_create.f90:250-268)

Columns follow tests/test_radtran.f90:27-67 (ModernEarth) and SURVEY.md 8(d) configs 3-5.
"""
import os

import numpy as np

SEED = 20260515
K_BOLTZ = 1.380649e-16  # src/clima_const.f90:10
C_LIGHT = 299792458.0
PLANK = 6.62607004e-34
K_BOLTZ_SI = 1.380649e-23
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

XS_CIA, XS_RAYLEIGH, XS_ABSORPTION, XS_PHOTOLYSIS = 0, 1, 2, 3

MODERN_EARTH_SPECIES = ("H2O", "CO2", "O2", "N2", "O3", "CH4")  # templates/ModernEarth/settings.yaml
EARLY_MARS_SPECIES = ("H2O", "CO2", "N2", "H2", "CH4", "CO", "O2")


def nominal_wavl(nw=1000):
    """nw log-spaced bins, 100 nm ... 1e6 nm."""
    return np.logspace(2.0, 6.0, nw + 1)


def gauss_weights01(ng):
    x, w = np.polynomial.legendre.leggauss(ng)
    return w / 2.0


class TableSet:
    """Plain container: what `create_OpticalProperties` + `create_RTChannel` +
    `read_stellar_flux` leave in a `Radtran` (clima_radtran.f90:171-196)."""

    def __init__(self):
        self.species_names = ()
        self.particle_names = ()
        self.wavl = None
        self.ktables = []
        self.xsections = []
        self.continuum = None
        self.particles = []
        self.ir_wavl = None
        self.sol_wavl = None
        self.photons_sol = None

    @property
    def nw(self):
        return len(self.wavl) - 1

    @property
    def nsp(self):
        return len(self.species_names)

    @property
    def np_(self):
        return len(self.particle_names)

    @property
    def ng(self):
        return len(self.ktables[0]["weights"])

    def table_bytes(self):
        n = sum(k["log10k"].nbytes for k in self.ktables)
        n += sum(x["data"].nbytes for x in self.xsections)
        if self.continuum is not None:
            n += self.continuum["log10_H2O"].nbytes + self.continuum["log10_foreign"].nbytes
        n += sum(p["w0"].nbytes * 3 for p in self.particles)
        return n


def _rayleigh_vardavas(A, B, Delta, lam_nm):
    """src/clima_eqns.f90:240-246"""
    return (4.577e-21 * ((6.0 + 3.0 * Delta) / (6.0 - 7.0 * Delta)) *
            (A * (1.0 + B / (lam_nm * 1.0e-3) ** 2.0)) ** 2.0 * (1.0 / (lam_nm * 1.0e-3) ** 4.0))


def blackbody_photons(wavl, T=5772.0, scale=2.16e-5):
    """Star-like mW/m^2/Hz per bin (used when the binned template spectrum does not apply)."""
    nu = C_LIGHT / (0.5 * (wavl[:-1] + wavl[1:]) * 1e-9)
    x = np.minimum(PLANK * nu / (K_BOLTZ_SI * T), 700.0)
    B = 1.0e3 * 2.0 * PLANK * nu ** 3 / C_LIGHT ** 2 / np.expm1(x)
    return np.pi * B * scale


def make_tables(nw=1000, ng=8, nP=20, nT=20, species=MODERN_EARTH_SPECIES,
                k_species=("H2O", "CO2", "O2", "O3", "CH4"),
                cia_pairs=(("N2", "N2"), ("O2", "O2"), ("CO2", "CO2"), ("O2", "N2"), ("CH4", "CH4"), ("CO2", "CH4")),
                ray_species=("CO2", "O2", "N2", "CH4", "H2O"),
                pxs_species=("H2O", "CO2", "O2", "O3", "CH4"),
                particles=("HCaer1",), water_continuum=True, nT_cia=10, nrad=20,
                star="sun_now", seed=SEED, sorted_k=True, sol_frac=0.6, ir_frac=0.4, weights=None):
    """Synthetic `TableSet`.  `sorted_k=False` scrambles the g ordering of
    the k-coefficients to exercise the general (unsorted) resort path."""
    rng = np.random.default_rng(seed)
    t = TableSet()
    t.species_names = tuple(species)
    t.particle_names = tuple(particles)
    t.wavl = nominal_wavl(nw)
    lam = np.sqrt(t.wavl[:-1] * t.wavl[1:])  # bin centres, nm
    x = np.log10(lam)  # 2..6
    weights = gauss_weights01(ng) if weights is None else np.asarray(weights, dtype=float)
    log10P = np.linspace(-6.0, 2.0, nP)
    temp = np.linspace(50.0, 1000.0, nT)
    gq = np.cumsum(weights) - 0.5 * weights  # g mid-points

    typ_col = {"H2O": 5e22, "CO2": 8e21, "O2": 4.5e24, "O3": 1e19, "CH4": 4e19, "CO": 1e21, "H2": 1e23, "N2": 1.6e25}
    for si, sp in enumerate(k_species):
        # smooth band structure per species + monotone-in-g ramp + noise; the floor is set
        # so that tau ~ 1e-4 in windows and up to ~1e6 in band centres at the highest g
        centres = 2.3 + 3.4 * rng.random(3)
        widths = 0.06 + 0.2 * rng.random(3)
        amps = 2.0 + 4.0 * rng.random(3)
        base = -np.log10(typ_col.get(sp, 1e22)) - 4.0 + np.zeros(nw)
        for c, wd, a in zip(centres, widths, amps):
            base = base + a * np.exp(-0.5 * ((x - c) / wd) ** 2)
        ramp = 3.0 * (gq ** 3) + 1.0 * gq  # steep tail at high g like real k-distributions
        arr = (base[:, None, None, None]
               + (0.002 * (temp - 300.0))[None, :, None, None]
               + (0.25 * (log10P + 2.0))[None, None, :, None] * (0.3 + 0.7 * (1 - gq))[None, None, None, :]
               + ramp[None, None, None, :]
               + rng.uniform(-0.2, 0.2, size=(nw, nT, nP, ng)))
        arr = np.clip(arr, -30.0, -18.0)
        if sorted_k:
            arr = np.sort(arr, axis=3)
        else:
            arr = arr[..., rng.permutation(ng)]
        t.ktables.append(dict(sp_ind=species.index(sp), weights=weights.copy(), log10P=log10P.copy(),
                              temp=temp.copy(), log10k=np.ascontiguousarray(arr)))

    has_h2o = "H2O" in species
    Tc = np.linspace(100.0, 600.0, nT_cia)
    for a, b in cia_pairs:
        if a not in species or b not in species:
            continue
        if water_continuum and has_h2o and "H2O" in (a, b):
            continue  # types_create.f90:431-433
        c0 = 2.5 + 3.0 * rng.random(3)
        data = -49.5 + np.zeros((nw, nT_cia))
        for c in c0:
            data = data + (3.0 * np.exp(-0.5 * ((x - c) / 0.3) ** 2))[:, None]
        data = data + (0.004 * (Tc - 300.0))[None, :] + rng.uniform(-0.1, 0.1, size=(nw, nT_cia))
        data = np.clip(data, -50.0, -43.0)
        t.xsections.append(dict(xs_type=XS_CIA, dim=1, sp1=species.index(a), sp2=species.index(b),
                                temp=Tc.copy(), data=np.ascontiguousarray(data)))
    ray_par = {"CO2": (43.9e-5, 6.4e-3, 0.0805), "O2": (26.63e-5, 5.07e-3, 0.054), "N2": (29.06e-5, 7.7e-3, 0.0305),
               "CH4": (42.6e-5, 14.41e-3, 0.0), "H2O": (28.0e-5, 5.0e-3, 0.17), "H2": (13.58e-5, 7.52e-3, 0.0),
               "CO": (32.7e-5, 8.1e-3, 0.0)}
    for sp in ray_species:
        if sp in species:
            A, B, D = ray_par[sp]
            t.xsections.append(dict(xs_type=XS_RAYLEIGH, dim=0, sp1=species.index(sp), sp2=-1, temp=None,
                                    data=_rayleigh_vardavas(A, B, D, lam)))
    for sp in pxs_species:
        if sp in species:
            edge = {"O2": 180.0, "CO2": 170.0, "H2O": 190.0, "O3": 300.0, "CH4": 140.0}.get(sp, 160.0)
            edge = edge * (0.9 + 0.2 * rng.random())
            xs = 1.0e-17 * np.exp(-((lam / edge) ** 6)) * 10 ** rng.uniform(-0.3, 0.3, size=nw)
            xs[lam > 3.0 * edge] = 0.0
            t.xsections.append(dict(xs_type=XS_PHOTOLYSIS, dim=0, sp1=species.index(sp), sp2=-1, temp=None,
                                    data=xs))
    if water_continuum and has_h2o:
        Tw = np.linspace(200.0, 400.0, nT_cia)
        shape = -42.5 - 0.6 * (x - 2.0) + 1.5 * np.exp(-0.5 * ((x - 4.3) / 0.4) ** 2)
        h2o = shape[:, None] - 0.006 * (Tw - 296.0)[None, :] + rng.uniform(-0.1, 0.1, size=(nw, nT_cia))
        frn = shape[:, None] - 2.3 - 0.002 * (Tw - 296.0)[None, :] + rng.uniform(-0.1, 0.1, size=(nw, nT_cia))
        t.continuum = dict(LH2O=species.index("H2O"), temp=Tw, log10_H2O=np.ascontiguousarray(h2o),
                           log10_foreign=np.ascontiguousarray(frn))
    rad = np.logspace(-7.0, -3.0, nrad)
    for pi_, _ in enumerate(particles):
        size = 2.0 * np.pi * rad[None, :] / (lam[:, None] * 1.0e-7)  # size parameter (lam nm -> cm)
        qext = 2.0 * size ** 4 / (1.0 + size ** 4) * (1.0 + 0.3 / (1.0 + size)) + 1.0e-12
        w0 = np.clip(0.2 + 0.75 * size ** 2 / (1.0 + size ** 2) + rng.uniform(-0.02, 0.02, size=size.shape), 0.0, 0.999)
        gt = np.clip(0.85 * size ** 2 / (1.0 + size ** 2) + rng.uniform(-0.02, 0.02, size=size.shape), 0.0, 0.95)
        t.particles.append(dict(p_ind=pi_, radii=rad.copy(), w0=np.ascontiguousarray(w0),
                                qext=np.ascontiguousarray(qext), gt=np.ascontiguousarray(gt)))

    n_sol_end = int(round(sol_frac * nw))
    n_ir_start = int(round(ir_frac * nw))
    t.sol_wavl = t.wavl[: n_sol_end + 1].copy()
    t.ir_wavl = t.wavl[n_ir_start:].copy()
    binned = os.path.join(_DATA, "stellar_binned.npz")
    if nw == 1000 and star is not None and os.path.exists(binned):
        t.photons_sol = np.load(binned)[star][:n_sol_end].copy()
    else:
        t.photons_sol = blackbody_photons(t.sol_wavl)
    return t


def modern_earth_tables(nw=1000, **kw):
    """BASELINE.json configs[0..1] / SURVEY 8(d): ModernEarth species, nk=5, 6 CIA pairs,
    5 Rayleigh, 5 photolysis, MT_CKD-like continuum, one Mie particle (HCaer1)."""
    return make_tables(nw=nw, **kw)


def early_mars_tables(nw=1000, **kw):
    """CO2-dominated, CIA-heavy (10 pairs)."""
    args = dict(species=EARLY_MARS_SPECIES, k_species=("H2O", "CO2", "CH4", "CO", "O2"),
                cia_pairs=(("CO2", "CO2"), ("N2", "N2"), ("H2", "H2"), ("CO2", "H2"), ("CO2", "CH4"), ("N2", "H2"),
                           ("CH4", "CH4"), ("O2", "O2"), ("N2", "O2"), ("CO2", "N2")),
                ray_species=("CO2", "N2", "H2", "CH4", "CO", "O2", "H2O"),
                pxs_species=("H2O", "CO2", "CH4", "CO", "O2"), particles=(), star="sun_3p8Ga", seed=SEED + 3)
    args.update(kw)
    return make_tables(nw=nw, **args)


# --------------------------------------------------------------------------- columns

class Column(dict):
    """T_surface, T(nz), P(nz) bar, densities(nz,nsp) cm^-3, dz(nz) cm, pdensities, radii."""

    def args(self):
        return (self["T_surface"], self["T"], self["P"], self["densities"], self["dz"],
                self.get("pdensities"), self.get("radii"))


def _vertical_grid(bottom, top, nz):
    """src/clima_eqns.f90:172-184"""
    dz = np.full(nz, (top - bottom) / nz)
    z = np.empty(nz)
    z[0] = dz[0] / 2.0
    for i in range(1, nz):
        z[i] = z[i - 1] + dz[i]
    return z, dz


def modern_earth_column(nz=200, species=MODERN_EARTH_SPECIES, n_particles=1, top=1.0e7):
    """tests/test_radtran.f90:27-67: uniform grid 0..100 km, log-interp of mixing ratios and
    P, linear T (unpack_atmospherefile, src/clima_types_create.f90:468-505),
    densities = mix*P*1e6/(k T), dummy particles (pdensities=1, radii=1e-5)."""
    d = np.load(os.path.join(_DATA, "modern_earth_atmosphere.npz"))
    z_f = d["alt_km"] * 1.0e5
    z, dz = _vertical_grid(0.0, top, nz)
    names = [str(s) for s in d["species"]]
    mix = np.empty((nz, len(species)))
    for i, sp in enumerate(species):
        mix[:, i] = 10.0 ** np.interp(z, z_f, np.log10(d["mix"][:, names.index(sp)]))
    T = np.interp(z, z_f, d["temp_K"])
    P = 10.0 ** np.interp(z, z_f, np.log10(d["press_bar"]))
    density = (P * 1.0e6) / (K_BOLTZ * T)
    col = Column(T_surface=float(T[0]), T=T, P=P, densities=np.asfortranarray(mix * density[:, None]), dz=dz)
    if n_particles > 0:
        col["pdensities"] = np.asfortranarray(np.full((nz, n_particles), 1.0))
        col["radii"] = np.asfortranarray(np.full((nz, n_particles), 1.0e-5))
    return col


def early_mars_column(nz=200, species=EARLY_MARS_SPECIES):
    """2 bar CO2, T_surf 250 K, dry adiabat to a 150 K isothermal
    top, layers log-spaced in P from P_surf to 1e-6 bar."""
    P_surf, T_surf = 2.0, 250.0
    edges = np.logspace(np.log10(P_surf), -6.0, nz + 1)
    P = np.sqrt(edges[:-1] * edges[1:])
    T = np.maximum(150.0, T_surf * (P / P_surf) ** 0.22)
    mu, grav = 44.0 * 1.66054e-24, 371.0
    H = K_BOLTZ * T / (mu * grav)
    dz = H * np.log(edges[:-1] / edges[1:])
    mixd = {"H2O": np.minimum(1.0e-3, 1.0e-3 * (P / P_surf)), "CO2": 0.95, "N2": 0.027, "H2": 0.02,
            "CH4": 1.0e-3, "CO": 1.0e-3, "O2": 1.0e-5}
    density = (P * 1.0e6) / (K_BOLTZ * T)
    dens = np.stack([np.broadcast_to(mixd[s], (nz,)) * density for s in species], axis=1)
    return Column(T_surface=T_surf, T=T, P=P, densities=np.asfortranarray(dens), dz=dz)


def perturbed_columns(ncol=1024, nz=200, seed=7, species=MODERN_EARTH_SPECIES, n_particles=1):
    """ModernEarth with whole-column dT ~ U(-20,20) + per-layer
    N(0,2 K), P x U(0.5,2), H2O and CO2 mixing ratios x 10^U(-1,1)."""
    rng = np.random.default_rng(seed)
    base = modern_earth_column(nz, species, n_particles)
    out = []
    for _ in range(ncol):
        c = Column({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()})
        T = base["T"] + rng.uniform(-20.0, 20.0) + rng.normal(0.0, 2.0, nz)
        P = base["P"] * rng.uniform(0.5, 2.0)
        mix = base["densities"] / ((base["P"] * 1.0e6) / (K_BOLTZ * base["T"]))[:, None]
        mix[:, species.index("H2O")] *= 10.0 ** rng.uniform(-1.0, 1.0)
        mix[:, species.index("CO2")] *= 10.0 ** rng.uniform(-1.0, 1.0)
        density = (P * 1.0e6) / (K_BOLTZ * T)
        c.update(T=T, P=P, T_surface=float(T[0]), densities=np.asfortranarray(mix * density[:, None]))
        out.append(c)
    return out


def doubled_column(col):
    """Radiative grid of AdiabatClimate (copy_atm_to_radiative_grid,
    src/adiabat/clima_adiabat.f90:729-773, nz_r = 2*nz): each layer split into two
    identical half-thickness layers -- the input pattern `pair_reuse` detects
    (clima_radtran_types.f90:621-632)."""
    rep = lambda a: np.repeat(a, 2, axis=0)
    c = Column(T_surface=col["T_surface"], T=rep(col["T"]), P=rep(col["P"]),
               densities=np.asfortranarray(rep(col["densities"])), dz=rep(col["dz"]) / 2.0)
    if col.get("radii") is not None:
        c["pdensities"] = np.asfortranarray(rep(col["pdensities"]))
        c["radii"] = np.asfortranarray(rep(col["radii"]))
    return c
