/*
 * clima_oracle.c -- CPU restatement (plain C, IEEE double) of Clima's hot path.
 * TEST INFRASTRUCTURE ONLY -- see clima_oracle.h for the rules and the pin status.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference repository root).  Operation order is kept as written in the reference so
 * that, compiled without FMA contraction, results track the Fortran bit-for-bit up to
 * libm differences.
 */
#include "clima_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/clima_const.f90:9-21 */
static const double PLANK = 6.62607004e-34;
static const double C_LIGHT = 299792458.0;
static const double K_BOLTZ_SI = 1.380649e-23;
static const double PI = 3.14159265358979323846e0;
/* src/radtran/clima_radtran_types.f90:9-11 */
static const double MAX_W0 = 0.99999;
static const double MAX_GT = 0.999999;
static const double TAU_MIN = 1.0e-20;

/* ------------------------------------------------------------------ data model */

typedef struct { /* Ktable, clima_radtran_types.f90:23-38 */
  int sp_ind, ng, nP, nT;
  double *weights, *log10P, *temp, *log10k; /* log10k [nw][nT][nP][ng] */
  double log10P_min, log10P_max, T_min, T_max;
} OrcKtable;

typedef struct { /* Xsection, clima_radtran_types.f90:44-55 */
  int xs_type, dim, sp1, sp2, nT;
  double *temp, *data; /* dim0: xs_0d[nw]; dim1: log10 xs [nw][nT] */
  double T_min, T_max;
} OrcXs;

typedef struct { /* ParticleXsection, clima_radtran_types.f90:57-66 */
  int p_ind, nrad;
  double *radii, *w0, *qext, *gt; /* [nw][nrad] */
  double r_min, r_max;
} OrcPart;

typedef struct { /* RTChannel, clima_radtran_types.f90:263-269 */
  int ind_start, ind_end, nw; /* 0-based ind_start, inclusive ind_end */
  double *wavl, *freq;        /* nw+1 */
} OrcChannel;

typedef struct { /* ClimaRadtranWrk, clima_radtran.f90:11-25 */
  double *fup_a, *fdn_a, *amean, *tau_band, *fup_n, *fdn_n;
} OrcWrk;

struct OrcRadtran {
  int nz, nsp, np, nw;
  double *wavl, *freq;
  int nk, ncia, nray, npxs, npart, ng;
  OrcKtable *k;
  OrcXs *cia, *ray, *pxs;
  OrcPart *part;
  int has_cont, LH2O, cont_nT;
  /* custom optical properties (types.f90:432-538): per bin l, linear in log10(P cgs) */
  int cust_on, cust_nP;
  double *cust_log10P, *cust_dtau, *cust_w0, *cust_g0; /* axis [nP] ascending; tables [nw][nP] */
  double *cont_temp, *cont_H2O, *cont_foreign, cont_Tmin, cont_Tmax;
  /* Ksettings, clima_radtran_types.f90:84-94 */
  double *wbin, *wbin_e, *wxy;
  OrcChannel ir, sol;
  int nzen;
  double *zenith_u, *zenith_w, *surface_albedo, *surface_emissivity, *photons_sol;
  double diurnal_fac, ir_tau_min, photon_scale_factor;
  int has_hard_surface;
  /* OpticalPropertiesResult, clima_radtran_types.f90:242-247 */
  double *tau, *w0, *g, *tau_band;
  OrcWrk wrk_ir, wrk_sol;
  double *f_total;
  int finalized;
};

static void set_err(char *err, const char *msg) {
  if (err) {
    strncpy(err, msg, ORC_ERR_LEN);
    err[ORC_ERR_LEN] = 0;
  }
}

static double *dupd(const double *src, size_t n) {
  double *p = (double *)malloc((n ? n : 1) * sizeof(double));
  if (src) memcpy(p, src, n * sizeof(double));
  return p;
}

static double minval(const double *x, int n) {
  double m = x[0];
  for (int i = 1; i < n; i++) if (x[i] < m) m = x[i];
  return m;
}
static double maxval(const double *x, int n) {
  double m = x[0];
  for (int i = 1; i < n; i++) if (x[i] > m) m = x[i];
  return m;
}

/* futils is_close (v0.1.14, taken from fortran-stdlib): |a-b| <= tol*max(|a|,|b|).
 * Call sites clima_radtran_types.f90:625-629, types_create.f90:259. */
static int is_close(double a, double b, double tol) {
  double m = fmax(fabs(a), fabs(b));
  return fabs(a - b) <= fabs(tol * m);
}

/* ------------------------------------------------------------------ small equations */

/* src/clima_eqns.f90:64-73 */
double orc_planck_fcn(double nu, double T) {
  return 1.0e3 * ((2.0 * PLANK * pow(nu, 3.0)) / (C_LIGHT * C_LIGHT)) *
         ((1.0) / (exp((PLANK * nu) / (K_BOLTZ_SI * T)) - 1.0));
}

/* src/clima_eqns.f90:75-80 */
double orc_ten2power(double y) {
  const double c = 2.302585092994045684017991454684364207601; /* log(10) */
  return exp(y * c);
}

/* src/clima_eqns.f90:43-54 */
static void weights_to_bins(int n, const double *weights, double *bins) {
  bins[0] = 0.0;
  for (int i = 1; i < n + 1; i++) bins[i] = weights[i - 1] + bins[i - 1];
}

/* Bracketing semantics of dintrv, src/dependencies/linear_interpolation_module.F90:348-350:
 *   x < xt(1)            -> (1,2)
 *   xt(i) <= x < xt(i+1) -> (i,i+1)
 *   xt(n) <= x           -> (n-1,n)
 * (stateless: the cached ilo only accelerates the search, :364-501). 0-based left index. */
static int bracket(int n, const double *xt, double x) {
  if (x < xt[0]) return 0;
  if (x >= xt[n - 1]) return n - 2;
  int lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    int mid = (lo + hi) / 2;
    if (x < xt[mid]) hi = mid; else lo = mid;
  }
  return lo;
}

/* linear_interp_1d%evaluate, linear_interpolation_module.F90:233-271 */
double orc_interp1d(int n, const double *x, const double *f, double xv) {
  int i = bracket(n, x, xv);
  double q1 = (xv - x[i]) / (x[i + 1] - x[i]);
  double p1 = 1.0 - q1;
  return p1 * f[i] + q1 * f[i + 1];
}

/* linear_interp_2d%evaluate, linear_interpolation_module.F90:297-337.
 * f is f(nx,ny) column-major. */
static double interp2d_strided(int nx, int ny, const double *x, const double *y,
                               const double *f, long sx, long sy, double xv, double yv) {
  int ix = bracket(nx, x, xv);
  int iy = bracket(ny, y, yv);
  double q1 = (xv - x[ix]) / (x[ix + 1] - x[ix]);
  double q2 = (yv - y[iy]) / (y[iy + 1] - y[iy]);
  double p1 = 1.0 - q1;
  double p2 = 1.0 - q2;
  double fx1 = p1 * f[ix * sx + iy * sy] + q1 * f[(ix + 1) * sx + iy * sy];
  double fx2 = p1 * f[ix * sx + (iy + 1) * sy] + q1 * f[(ix + 1) * sx + (iy + 1) * sy];
  return p2 * fx1 + q2 * fx2;
}
double orc_interp2d(int nx, int ny, const double *x, const double *y, const double *f,
                    double xv, double yv) {
  return interp2d_strided(nx, ny, x, y, f, 1, nx, xv, yv);
}

/* futils mrgrnk (ORDERPACK): ascending rank, ties keep original order (stable).
 * Call site clima_radtran_types.f90:840.  irank 0-based. */
void orc_mrgrnk(int n, const double *x, int *irank) {
  int *a = irank;
  int *b = (int *)malloc((n ? n : 1) * sizeof(int));
  for (int i = 0; i < n; i++) a[i] = i;
  for (int w = 1; w < n; w *= 2) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n;
      int hi = lo + 2 * w < n ? lo + 2 * w : n;
      int i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
        if (x[a[j]] < x[a[i]]) b[k++] = a[j++]; else b[k++] = a[i++];
      }
      while (i < mid) b[k++] = a[i++];
      while (j < hi) b[k++] = a[j++];
    }
    memcpy(a, b, n * sizeof(int));
  }
  free(b);
}

/* futils rebin: conservative rebinning of piecewise-constant old_vals on old_bins onto
 * new_bins: new_vals(k) = sum_j overlap(j,k)*old_vals(j) / (new_bins(k+1)-new_bins(k)),
 * accumulated in ascending j.  Documented semantics clima/cython/futils.pyx:16-35;
 * call site clima_radtran_types.f90:847. */
void orc_rebin(int n_old, const double *old_bins, const double *old_vals, int n_new,
               const double *new_bins, double *new_vals) {
  int l = 0;
  for (int k = 0; k < n_new; k++) {
    double b0 = new_bins[k], b1 = new_bins[k + 1];
    double acc = 0.0;
    while (l < n_old && old_bins[l + 1] <= b0) l++;
    for (int j = l; j < n_old; j++) {
      if (old_bins[j] >= b1) break;
      double lo = old_bins[j] > b0 ? old_bins[j] : b0;
      double hi = old_bins[j + 1] < b1 ? old_bins[j + 1] : b1;
      if (hi > lo) acc = acc + (hi - lo) * old_vals[j];
    }
    new_vals[k] = acc / (b1 - b0);
  }
}

/* futils gauss_legendre: nodes/weights on [-1,1] (setup-time; clima_eqns.f90:26-41). */
void orc_gauss_legendre(int n, double *x, double *w) {
  for (int i = 0; i < n; i++) {
    double z = cos(PI * (i + 0.75) / (n + 0.5));
    double pp = 0.0;
    for (int it = 0; it < 100; it++) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 0; j < n; j++) {
        double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      double z1 = z;
      z = z1 - p1 / pp;
      if (fabs(z - z1) < 1e-16) break;
    }
    x[n - 1 - i] = z; /* ascending */
    w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
  }
}

/* ------------------------------------------------------------------ two-stream */

/* src/radtran/clima_radtran_twostream.f90:297-316 */
void orc_tridiag(int n, double *a, double *b, double *c, double *d) {
  c[0] = c[0] / b[0];
  d[0] = d[0] / b[0];
  for (int i = 1; i < n - 1; i++) {
    c[i] = c[i] / (b[i] - a[i] * c[i - 1]);
    d[i] = (d[i] - a[i] * d[i - 1]) / (b[i] - a[i] * c[i - 1]);
  }
  d[n - 1] = (d[n - 1] - a[n - 1] * d[n - 2]) / (b[n - 1] - a[n - 1] * c[n - 2]);
  for (int i = n - 2; i >= 0; i--) d[i] = d[i] - c[i] * d[i + 1];
}

/* Coefficients of the tridiagonal system, identical in both solvers
 * (clima_radtran_twostream.f90:91-117 and :249-275). */
static void assemble(int nz, const double *e1, const double *e2, const double *e3,
                     const double *e4, const double *cp0, const double *cpb,
                     const double *cm0, const double *cmb, double Rsfc, double Ssfc,
                     double *A, double *B, double *D, double *E) {
  A[0] = 0.0;
  B[0] = e1[0];
  D[0] = -e2[0];
  E[0] = 0.0 - cm0[0];
  for (int i = 0; i < nz - 1; i++) {
    int l = 2 * i + 2; /* Fortran l = 2*i+1 (1-based i), 0-based row 2i+2 */
    A[l] = e2[i] * e3[i] - e4[i] * e1[i];
    B[l] = e1[i] * e1[i + 1] - e3[i] * e3[i + 1];
    D[l] = e3[i] * e4[i + 1] - e1[i] * e2[i + 1];
    E[l] = e3[i] * (cp0[i + 1] - cpb[i]) + e1[i] * (cmb[i] - cm0[i + 1]);
  }
  for (int i = 0; i < nz - 1; i++) {
    int l = 2 * i + 1; /* Fortran l = 2*i */
    A[l] = e2[i + 1] * e1[i] - e3[i] * e4[i + 1];
    B[l] = e2[i] * e2[i + 1] - e4[i] * e4[i + 1];
    D[l] = e1[i + 1] * e4[i + 1] - e2[i + 1] * e3[i + 1];
    E[l] = e2[i + 1] * (cp0[i + 1] - cpb[i]) - e4[i + 1] * (cm0[i + 1] - cmb[i]);
  }
  int l = 2 * nz - 1;
  A[l] = e1[nz - 1] - Rsfc * e3[nz - 1];
  B[l] = e2[nz - 1] - Rsfc * e4[nz - 1];
  D[l] = 0.0;
  E[l] = Ssfc - cpb[nz - 1] + Rsfc * cmb[nz - 1];
}

/* src/radtran/clima_radtran_twostream.f90:10-154 */
void orc_two_stream_solar(int nz, const double *tau_in, const double *w0_in,
                          const double *gt_in, double u0, double Rsfc, double *amean,
                          double *surface_radiance, double *fup, double *fdn) {
  double *buf = (double *)malloc(sizeof(double) * (size_t)(nz * 19 + 2 * (nz + 1) + 8 * nz));
  double *tau = buf, *w0 = tau + nz, *gt = w0 + nz, *gam1 = gt + nz, *gam2 = gam1 + nz,
         *gam3 = gam2 + nz, *gam4 = gam3 + nz, *lambda = gam4 + nz, *cap_gam = lambda + nz,
         *e1 = cap_gam + nz, *e2 = e1 + nz, *e3 = e2 + nz, *e4 = e3 + nz, *cp0 = e4 + nz,
         *cpb = cp0 + nz, *cm0 = cpb + nz, *cmb = cm0 + nz, *y1 = cmb + nz, *y2 = y1 + nz,
         *tauc = y2 + nz, *direct = tauc + nz + 1, *A = direct + nz + 1, *B = A + 2 * nz,
         *D = B + 2 * nz, *E = D + 2 * nz;
  const double sqrt3 = sqrt(3.0);
  const double u1 = 1.0 / sqrt(3.0);
  const double Fs_pi = 1.0;

  for (int i = 0; i < nz; i++) { /* :38-51 */
    tau[i] = tau_in[i] * (1.0 - w0_in[i] * gt_in[i] * gt_in[i]);
    w0[i] = w0_in[i] * (1.0 - gt_in[i] * gt_in[i]) / (1.0 - w0_in[i] * gt_in[i] * gt_in[i]);
    gt[i] = gt_in[i] / (1.0 + gt_in[i]);
    gam1[i] = sqrt3 * (2.0 - w0[i] * (1 + gt[i])) / 2.0;
    gam2[i] = sqrt3 * w0[i] * (1.0 - gt[i]) / 2.0;
    gam3[i] = (1.0 - sqrt3 * gt[i] * u0) / 2.0;
    gam4[i] = 1.0 - gam3[i];
    lambda[i] = sqrt(gam1[i] * gam1[i] - gam2[i] * gam2[i]);
    cap_gam[i] = gam2[i] / (gam1[i] + lambda[i]);
  }
  for (int i = 0; i < nz; i++) { /* :55-61 */
    double wrk = exp(-lambda[i] * tau[i]);
    e1[i] = 1.0 + cap_gam[i] * wrk;
    e2[i] = 1.0 - cap_gam[i] * wrk;
    e3[i] = cap_gam[i] + wrk;
    e4[i] = cap_gam[i] - wrk;
  }
  tauc[0] = 0.0; /* :64-67 */
  for (int i = 1; i < nz + 1; i++) tauc[i] = tauc[i - 1] + tau[i - 1];
  direct[0] = u0 * Fs_pi; /* :73-87 */
  for (int i = 0; i < nz; i++) {
    double facp = w0[i] * Fs_pi * ((gam1[i] - 1.0 / u0) * gam3[i] + gam4[i] * gam2[i]);
    double facm = w0[i] * Fs_pi * ((gam1[i] + 1.0 / u0) * gam4[i] + gam2[i] * gam3[i]);
    double et0 = exp(-tauc[i] / u0);
    double etb = et0 * exp(-tau[i] / u0);
    double denom = lambda[i] * lambda[i] - 1.0 / (u0 * u0);
    direct[i + 1] = u0 * Fs_pi * etb;
    cp0[i] = et0 * facp / denom;
    cpb[i] = etb * facp / denom;
    cm0[i] = et0 * facm / denom;
    cmb[i] = etb * facm / denom;
  }
  double Ssfc = Rsfc * direct[nz]; /* :89 */
  assemble(nz, e1, e2, e3, e4, cp0, cpb, cm0, cmb, Rsfc, Ssfc, A, B, D, E);
  orc_tridiag(nz * 2, A, B, D, E); /* :120 */
  for (int i = 0; i < nz; i++) { y1[i] = E[2 * i]; y2[i] = E[2 * i + 1]; }
  /* :135-148 */
  amean[0] = (1.0 / u1) * (y1[0] * e3[0] - y2[0] * e4[0] + cp0[0]) + direct[0] / u0;
  for (int i = 0; i < nz; i++)
    amean[i + 1] = (1.0 / u1) * (y1[i] * (e1[i] + e3[i]) + y2[i] * (e2[i] + e4[i]) + cpb[i] + cmb[i]) +
                   direct[i + 1] / u0;
  fup[0] = ((y1[0] * e3[0] - y2[0] * e4[0]) + cp0[0]);
  fdn[0] = direct[0];
  for (int i = 0; i < nz; i++) {
    fup[i + 1] = (y1[i] * e1[i] + y2[i] * e2[i] + cpb[i]);
    fdn[i + 1] = (y1[i] * e3[i] + y2[i] * e4[i] + cmb[i]) + direct[i + 1];
  }
  int i = nz - 1; /* :151-152 */
  *surface_radiance = (y1[i] * e3[i] + y2[i] * e4[i] + cmb[i]) / u1 + exp(-tauc[i + 1] / u0);
  free(buf);
}

/* src/radtran/clima_radtran_twostream.f90:156-295 */
void orc_two_stream_ir(int nz, const double *tau, const double *w0, const double *gt,
                       double emissivity, int has_hard_surface, double tau_min,
                       const double *bplanck, double *fup, double *fdn) {
  double *buf = (double *)malloc(sizeof(double) * (size_t)(nz * 14 + 8 * nz));
  double *gam1 = buf, *gam2 = gam1 + nz, *lambda = gam2 + nz, *cap_gam = lambda + nz,
         *e1 = cap_gam + nz, *e2 = e1 + nz, *e3 = e2 + nz, *e4 = e3 + nz, *cp0 = e4 + nz,
         *cpb = cp0 + nz, *cm0 = cpb + nz, *cmb = cm0 + nz, *y1 = cmb + nz, *y2 = y1 + nz,
         *A = y2 + nz, *B = A + 2 * nz, *D = B + 2 * nz, *E = D + 2 * nz;
  const double u1 = 0.5;
  const double norm = 2.0 * PI * u1;
  double Rsfc, Ssfc;
  if (has_hard_surface) Rsfc = 1.0 - emissivity; else Rsfc = 0.0; /* :186-190 */
  for (int i = 0; i < nz; i++) { /* :195-201 */
    gam1[i] = 2.0 - w0[i] * (1.0 + gt[i]);
    gam2[i] = w0[i] * (1.0 - gt[i]);
    lambda[i] = sqrt(gam1[i] * gam1[i] - gam2[i] * gam2[i]);
    cap_gam[i] = gam2[i] / (gam1[i] + lambda[i]);
  }
  for (int i = 0; i < nz; i++) { /* :205-211 */
    double wrk = exp(-lambda[i] * tau[i]);
    e1[i] = 1.0 + cap_gam[i] * wrk;
    e2[i] = 1.0 - cap_gam[i] * wrk;
    e3[i] = cap_gam[i] + wrk;
    e4[i] = cap_gam[i] - wrk;
  }
  for (int i = 0; i < nz; i++) { /* :215-234 */
    double b0n, b1n;
    if (tau[i] <= tau_min) {
      double b_avg = 0.5 * (bplanck[i] + bplanck[i + 1]);
      b0n = b_avg;
      b1n = 0.0;
    } else {
      b0n = bplanck[i];
      b1n = (bplanck[i + 1] - b0n) / tau[i];
    }
    cp0[i] = norm * (b0n + b1n * (1.0 / (gam1[i] + gam2[i])));
    cpb[i] = norm * (b0n + b1n * (tau[i] + 1.0 / (gam1[i] + gam2[i])));
    cm0[i] = norm * (b0n + b1n * (-1.0 / (gam1[i] + gam2[i])));
    cmb[i] = norm * (b0n + b1n * (tau[i] - 1.0 / (gam1[i] + gam2[i])));
  }
  if (has_hard_surface) { /* :236-247 */
    Ssfc = emissivity * PI * bplanck[nz];
  } else {
    double b1_bot;
    if (tau[nz - 1] <= tau_min) b1_bot = 0.0;
    else b1_bot = (bplanck[nz] - bplanck[nz - 1]) / tau[nz - 1];
    Ssfc = PI * (bplanck[nz] + u1 * b1_bot);
  }
  assemble(nz, e1, e2, e3, e4, cp0, cpb, cm0, cmb, Rsfc, Ssfc, A, B, D, E);
  orc_tridiag(nz * 2, A, B, D, E); /* :278 */
  for (int i = 0; i < nz; i++) { y1[i] = E[2 * i]; y2[i] = E[2 * i + 1]; }
  fup[0] = ((y1[0] * e3[0] - y2[0] * e4[0]) + cp0[0]); /* :288-293 */
  fdn[0] = 0.0;
  for (int i = 0; i < nz; i++) {
    fup[i + 1] = (y1[i] * e1[i] + y2[i] * e2[i] + cpb[i]);
    fdn[i + 1] = (y1[i] * e3[i] + y2[i] * e4[i] + cmb[i]);
  }
  free(buf);
}

/* ------------------------------------------------------------------ construction */

OrcRadtran *orc_create(int nz, int nsp, int np, int nw, const double *wavl) {
  OrcRadtran *r = (OrcRadtran *)calloc(1, sizeof(OrcRadtran));
  r->nz = nz; r->nsp = nsp; r->np = np; r->nw = nw;
  r->wavl = dupd(wavl, nw + 1);
  r->freq = dupd(NULL, nw + 1);
  /* types_create.f90:361: freq = c_light/(wavl*1e-9) */
  for (int i = 0; i < nw + 1; i++) r->freq[i] = C_LIGHT / (wavl[i] * 1.0e-9);
  r->k = NULL; r->cia = r->ray = r->pxs = NULL; r->part = NULL;
  r->diurnal_fac = 0.5;           /* clima_radtran.f90:51 */
  r->has_hard_surface = 1;        /* :60 */
  r->ir_tau_min = 1.0e-6;         /* :62 */
  r->photon_scale_factor = 1.0;   /* :68 */
  return r;
}

static void free_wrk(OrcWrk *w) {
  free(w->fup_a); free(w->fdn_a); free(w->amean); free(w->tau_band); free(w->fup_n); free(w->fdn_n);
}

void orc_destroy(OrcRadtran *r) {
  if (!r) return;
  for (int i = 0; i < r->nk; i++) { free(r->k[i].weights); free(r->k[i].log10P); free(r->k[i].temp); free(r->k[i].log10k); }
  OrcXs *lists[3] = {r->cia, r->ray, r->pxs};
  int ns[3] = {r->ncia, r->nray, r->npxs};
  for (int t = 0; t < 3; t++) { for (int i = 0; i < ns[t]; i++) { free(lists[t][i].temp); free(lists[t][i].data); } free(lists[t]); }
  for (int i = 0; i < r->npart; i++) { free(r->part[i].radii); free(r->part[i].w0); free(r->part[i].qext); free(r->part[i].gt); }
  free(r->k); free(r->part);
  free(r->cont_temp); free(r->cont_H2O); free(r->cont_foreign);
  free(r->cust_log10P); free(r->cust_dtau); free(r->cust_w0); free(r->cust_g0);
  free(r->wavl); free(r->freq); free(r->wbin); free(r->wbin_e); free(r->wxy);
  free(r->ir.wavl); free(r->ir.freq); free(r->sol.wavl); free(r->sol.freq);
  free(r->zenith_u); free(r->zenith_w); free(r->surface_albedo); free(r->surface_emissivity); free(r->photons_sol);
  free(r->tau); free(r->w0); free(r->g); free(r->tau_band);
  free_wrk(&r->wrk_ir); free_wrk(&r->wrk_sol); free(r->f_total);
  free(r);
}

/* futils v0.1.14 `interp(xg, x, y, yg, ierr=)` as called at clima_radtran_types.f90:487-497:
 * piecewise-linear in x (ascending), constant beyond both ends (linear_extrap defaults to
 * false).  Third-party, absent from /root/reference: restated from its published
 * behaviour -- parity unpinned.  Returns nonzero when x is not strictly ascending. */
static int futils_interp(int ng, const double *xg, int n, const double *x, const double *y, double *yg) {
  if (n < 1) return -1;
  for (int i = 1; i < n; i++) if (!(x[i] > x[i - 1])) return -2;
  for (int i = 0; i < ng; i++) {
    double xv = xg[i];
    if (xv <= x[0]) yg[i] = y[0];
    else if (xv >= x[n - 1]) yg[i] = y[n - 1];
    else {
      int lo = 0, hi = n - 1;
      while (hi - lo > 1) { int mid = (lo + hi) / 2; if (xv < x[mid]) hi = mid; else lo = mid; }
      double slope = (y[lo + 1] - y[lo]) / (x[lo + 1] - x[lo]);
      yg[i] = y[lo] + slope * (xv - x[lo]);
    }
  }
  return 0;
}

/* OpticalProperties_set_custom_optical_properties, clima_radtran_types.f90:432-538.
 * dtau_dz, w0, g0 are (nP, nwv) column-major; P in dynes/cm^2, decreasing. */
int orc_set_custom_optical_properties(OrcRadtran *r, int nwv, const double *wv, int nP, const double *P,
                                      int d1_t, int d2_t, const double *dtau_dz, int d1_w, int d2_w,
                                      const double *w0, int d1_g, int d2_g, const double *g0, char *err) {
  for (int i = 0; i < nwv; i++) if (wv[i] <= 0.0) { set_err(err, "All elements of `wv` must be larger than zero"); return 1; }
  for (int i = 0; i < nP; i++) if (P[i] <= 0.0) { set_err(err, "All elements of `P` must be larger than zero"); return 1; }
  if (nP != d1_t) { set_err(err, "`P` and `dtau_dz` have incompatible shapes"); return 1; }
  if (nwv != d2_t) { set_err(err, "`wv` and `dtau_dz` have incompatible shapes"); return 1; }
  if (nP != d1_w) { set_err(err, "`P` and `w0` have incompatible shapes"); return 1; }
  if (nwv != d2_w) { set_err(err, "`wv` and `w0` have incompatible shapes"); return 1; }
  if (nP != d1_g) { set_err(err, "`P` and `g0` have incompatible shapes"); return 1; }
  if (nwv != d2_g) { set_err(err, "`wv` and `g0` have incompatible shapes"); return 1; }
  const int nw = r->nw;
  double *wv1 = dupd(NULL, nw), *row = dupd(NULL, nwv);
  double *t = dupd(NULL, (size_t)nw * nP), *w = dupd(NULL, (size_t)nw * nP), *g = dupd(NULL, (size_t)nw * nP);
  double *lp = dupd(NULL, nP);
  int bad = 0;
  for (int i = 0; i < nw; i++) wv1[i] = 0.5 * (r->wavl[i + 1] + r->wavl[i]); /* :479 */
  const double *src[3] = {dtau_dz, w0, g0};
  double *dst[3] = {t, w, g};
  double *tmp = dupd(NULL, nw);
  for (int i = 0; i < nP && !bad; i++) {
    int j = nP - 1 - i; /* :483 */
    for (int a = 0; a < 3 && !bad; a++) {
      for (int k = 0; k < nwv; k++) row[k] = src[a][i + (size_t)k * nP];
      if (futils_interp(nw, wv1, nwv, wv, row, tmp) != 0) bad = 1;
      for (int l = 0; l < nw; l++) dst[a][(size_t)l * nP + j] = tmp[l];
    }
  }
  free(tmp); free(row); free(wv1);
  if (bad) { free(t); free(w); free(g); free(lp); set_err(err, "Interpolation error in `set_custom_optical_properties`"); return 1; }
  for (int i = 0; i < nP; i++) lp[nP - 1 - i] = log10(P[i]); /* :505-506 */
  /* linear_interp_1d%initialize: at least two nodes, strictly increasing */
  int ok = nP >= 2;
  for (int i = 1; i < nP && ok; i++) if (!(lp[i] > lp[i - 1])) ok = 0;
  if (!ok) { free(t); free(w); free(g); free(lp); set_err(err, "Interpolation initialization error in `set_custom_optical_properties`"); return 1; }
  free(r->cust_log10P); free(r->cust_dtau); free(r->cust_w0); free(r->cust_g0);
  r->cust_log10P = lp; r->cust_dtau = t; r->cust_w0 = w; r->cust_g0 = g;
  r->cust_nP = nP; r->cust_on = 1;
  return 0;
}

void orc_unset_custom_optical_properties(OrcRadtran *r) { r->cust_on = 0; } /* :541-548 */

int orc_add_ktable(OrcRadtran *r, int sp_ind, int ng, const double *weights, int nP,
                   const double *log10P, int nT, const double *temp, const double *log10k,
                   char *err) {
  if (r->nk > 0 && ng != r->ng) { set_err(err, "all k-distributions must share the same number of g-points"); return 1; }
  r->k = (OrcKtable *)realloc(r->k, sizeof(OrcKtable) * (r->nk + 1));
  OrcKtable *k = &r->k[r->nk++];
  k->sp_ind = sp_ind; k->ng = ng; k->nP = nP; k->nT = nT;
  k->weights = dupd(weights, ng);
  k->log10P = dupd(log10P, nP);
  k->temp = dupd(temp, nT);
  k->log10k = dupd(log10k, (size_t)r->nw * nT * nP * ng);
  k->log10P_min = minval(log10P, nP); k->log10P_max = maxval(log10P, nP); /* types_create.f90:1371-1375 */
  k->T_min = minval(temp, nT); k->T_max = maxval(temp, nT);
  if (r->nk == 1) { /* create_Ksettings, types_create.f90:191-224 from the first k-table */
    r->ng = ng;
    r->wbin = dupd(weights, ng);
    r->wbin_e = dupd(NULL, ng + 1);
    weights_to_bins(ng, weights, r->wbin_e); /* weight_e, types_create.f90:1303-1304 */
    r->wxy = dupd(NULL, ng * ng);
    for (int i = 0; i < ng; i++)
      for (int j = 0; j < ng; j++) r->wxy[j + i * ng] = r->wbin[i] * r->wbin[j];
  }
  return 0;
}

int orc_add_xsection(OrcRadtran *r, int xs_type, int dim, int sp1, int sp2, int nT,
                     const double *temp, const double *data, char *err) {
  OrcXs **list; int *n;
  if (xs_type == ORC_XS_CIA) { list = &r->cia; n = &r->ncia; }
  else if (xs_type == ORC_XS_RAYLEIGH) { list = &r->ray; n = &r->nray; }
  else if (xs_type == ORC_XS_PHOTOLYSIS || xs_type == ORC_XS_ABSORPTION) { list = &r->pxs; n = &r->npxs; }
  else { set_err(err, "unknown cross-section type"); return 1; }
  if (dim != 0 && dim != 1) { set_err(err, "cross-section dim must be 0 or 1"); return 1; }
  *list = (OrcXs *)realloc(*list, sizeof(OrcXs) * (*n + 1));
  OrcXs *x = &(*list)[(*n)++];
  x->xs_type = xs_type; x->dim = dim; x->sp1 = sp1; x->sp2 = sp2; x->nT = dim ? nT : 0;
  x->temp = dim ? dupd(temp, nT) : NULL;
  x->data = dupd(data, dim ? (size_t)r->nw * nT : (size_t)r->nw);
  x->T_min = dim ? minval(temp, nT) : 0; x->T_max = dim ? maxval(temp, nT) : 0;
  return 0;
}

int orc_set_water_continuum(OrcRadtran *r, int LH2O, int nT, const double *temp,
                            const double *log10_H2O, const double *log10_foreign, char *err) {
  (void)err;
  r->has_cont = 1; r->LH2O = LH2O; r->cont_nT = nT;
  r->cont_temp = dupd(temp, nT);
  r->cont_H2O = dupd(log10_H2O, (size_t)r->nw * nT);
  r->cont_foreign = dupd(log10_foreign, (size_t)r->nw * nT);
  r->cont_Tmin = minval(temp, nT); r->cont_Tmax = maxval(temp, nT);
  return 0;
}

int orc_add_particle(OrcRadtran *r, int p_ind, int nrad, const double *radii,
                     const double *w0, const double *qext, const double *gt, char *err) {
  (void)err;
  r->part = (OrcPart *)realloc(r->part, sizeof(OrcPart) * (r->npart + 1));
  OrcPart *p = &r->part[r->npart++];
  p->p_ind = p_ind; p->nrad = nrad;
  p->radii = dupd(radii, nrad);
  size_t n = (size_t)r->nw * nrad;
  p->w0 = dupd(w0, n); p->qext = dupd(qext, n); p->gt = dupd(gt, n);
  p->r_min = minval(radii, nrad); p->r_max = maxval(radii, nrad);
  return 0;
}

/* create_RTChannel, types_create.f90:226-270 */
static int make_channel(OrcRadtran *r, OrcChannel *c, int n, const double *wavl, char *err) {
  int ind1 = 0, ind2 = 0;
  double best1 = INFINITY, best2 = INFINITY;
  for (int i = 0; i < r->nw + 1; i++) { /* minloc: first minimum */
    double d1 = fabs(wavl[0] - r->wavl[i]), d2 = fabs(wavl[n - 1] - r->wavl[i]);
    if (d1 < best1) { best1 = d1; ind1 = i; }
    if (d2 < best2) { best2 = d2; ind2 = i; }
  }
  if (n != ind2 - ind1 + 1) { set_err(err, "The wavelength bins are not compatible with the k-distribution wavelength bins."); return 1; }
  for (int i = 0; i < n; i++)
    if (!is_close(wavl[i], r->wavl[ind1 + i], 1.0e-7)) { set_err(err, "The wavelength bins are not compatible with the k-distribution wavelength bins."); return 1; }
  c->nw = n - 1;
  c->wavl = dupd(wavl, n);
  c->freq = dupd(NULL, n);
  for (int i = 0; i < n; i++) c->freq[i] = C_LIGHT / (wavl[i] * 1.0e-9);
  c->ind_start = ind1;
  c->ind_end = ind2 - 1;
  return 0;
}

int orc_set_channels(OrcRadtran *r, int n_ir, const double *ir_wavl, int n_sol,
                     const double *sol_wavl, char *err) {
  if (make_channel(r, &r->ir, n_ir, ir_wavl, err)) return 1;
  if (make_channel(r, &r->sol, n_sol, sol_wavl, err)) return 1;
  return 0;
}

int orc_set_photons_sol(OrcRadtran *r, int n, const double *photons_sol, char *err) {
  if (n != r->sol.nw) { set_err(err, "\"photons_sol\" has the wrong size"); return 1; }
  free(r->photons_sol);
  r->photons_sol = dupd(photons_sol, n);
  return 0;
}

static void alloc_wrk(OrcWrk *w, int nz, int nw) {
  w->fup_a = (double *)calloc((size_t)(nz + 1) * nw, sizeof(double));
  w->fdn_a = (double *)calloc((size_t)(nz + 1) * nw, sizeof(double));
  w->amean = (double *)calloc((size_t)(nz + 1) * nw, sizeof(double));
  w->tau_band = (double *)calloc((size_t)nz * nw, sizeof(double));
  w->fup_n = (double *)calloc(nz + 1, sizeof(double));
  w->fdn_n = (double *)calloc(nz + 1, sizeof(double));
}

/* remainder of create_Radtran_2, clima_radtran.f90:149-217 */
int orc_finalize(OrcRadtran *r, int nzen, double surface_albedo, char *err) {
  if (r->nz < 1) { set_err(err, "\"nz\" can not be less than 1."); return 1; }
  if (!r->ir.wavl || !r->sol.wavl) { set_err(err, "channels are not set"); return 1; }
  int nz = r->nz;
  /* zenith_angles_and_weights, clima_eqns.f90:26-41; then cos(deg*pi/180), clima_radtran.f90:165 */
  r->nzen = nzen;
  r->zenith_u = dupd(NULL, nzen); r->zenith_w = dupd(NULL, nzen);
  double *x = dupd(NULL, nzen), *w = dupd(NULL, nzen);
  orc_gauss_legendre(nzen, x, w);
  for (int i = 0; i < nzen; i++) {
    double mu = x[i] / 2.0 + 1.0 / 2.0;
    double ang = acos(mu) * 180.0 / PI;
    r->zenith_w[i] = w[i] / 2.0;
    r->zenith_u[i] = cos(ang * PI / 180.0);
  }
  free(x); free(w);
  r->surface_albedo = dupd(NULL, r->sol.nw);
  for (int i = 0; i < r->sol.nw; i++) r->surface_albedo[i] = surface_albedo;
  r->surface_emissivity = dupd(NULL, r->ir.nw);
  for (int i = 0; i < r->ir.nw; i++) r->surface_emissivity[i] = 1.0;
  if (!r->photons_sol) { r->photons_sol = (double *)calloc(r->sol.nw, sizeof(double)); }
  size_t n3 = (size_t)nz * r->ng * r->nw, n2 = (size_t)nz * r->nw;
  r->tau = (double *)calloc(n3 ? n3 : 1, sizeof(double));
  r->w0 = (double *)calloc(n3 ? n3 : 1, sizeof(double));
  r->g = (double *)calloc(n2, sizeof(double));
  r->tau_band = (double *)calloc(n2, sizeof(double));
  alloc_wrk(&r->wrk_ir, nz, r->ir.nw);
  alloc_wrk(&r->wrk_sol, nz, r->sol.nw);
  r->f_total = (double *)calloc(nz + 1, sizeof(double));
  r->finalized = 1;
  return 0;
}

void orc_set_zenith(OrcRadtran *r, int n, const double *u, const double *w) {
  free(r->zenith_u); free(r->zenith_w);
  r->nzen = n; r->zenith_u = dupd(u, n); r->zenith_w = dupd(w, n);
}
void orc_get_zenith(OrcRadtran *r, double *u, double *w) {
  memcpy(u, r->zenith_u, r->nzen * sizeof(double));
  memcpy(w, r->zenith_w, r->nzen * sizeof(double));
}
void orc_set_surface_albedo(OrcRadtran *r, const double *a) { memcpy(r->surface_albedo, a, r->sol.nw * sizeof(double)); }
void orc_set_surface_emissivity(OrcRadtran *r, const double *e) { memcpy(r->surface_emissivity, e, r->ir.nw * sizeof(double)); }
void orc_set_scalars(OrcRadtran *r, double diurnal_fac, int has_hard_surface, double ir_tau_min, double photon_scale_factor) {
  r->diurnal_fac = diurnal_fac; r->has_hard_surface = has_hard_surface;
  r->ir_tau_min = ir_tau_min; r->photon_scale_factor = photon_scale_factor;
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Census hook (diagnostics for the build's kernel design, not part of the restated algorithm): when a
 * buffer is set, every random-overlap mixing step records, per (bin l, step jj-1, layer i) in
 * buf[((l*(nk-1) + jj-1)*nz + i)*4 + 0..3]:
 *   [0] flags: 1 = y ascending, 2 = x ascending, 4 = max(y)-min(y) <= smallest gap of x (the ng*ng sums are
 *       already ascending in the order of types.f90:826-830, "row-major"), 8 = max(x)-min(x) <= smallest gap
 *       of y ("column-major"), 16 / 32 = the same two with the range also allowed up to 2^-46 of the smallest
 *       sum, 64 = layer copied by pair_reuse (no sort)
 *   [1] bit g-1 set: x(g) - x(g-1) >= range of y (rows g-1 and g of the sums do not interleave)   [2] the same
 *       with x and y exchanged
 *   [3] number of inversions of the row-major order against the sorted one, saturated at 255 */
static unsigned char *g_census = NULL;
void orc_census_set(unsigned char *buf) { g_census = buf; }

static void census_step(unsigned char *o, int ng, const double *x, int xs, const double *y, const double *xy, const int *inds) {
  int ys_ = 1, xs_ = 1, nx = 0, ny = 0;
  double ymin = y[0], ymax = y[0], xmin = x[0], xmax = x[0], gx = 1e300, gy = 1e300;
  for (int g = 1; g < ng; g++) {
    if (y[g] < y[g - 1]) ys_ = 0;
    if (x[g * xs] < x[(g - 1) * xs]) xs_ = 0;
    ymin = fmin(ymin, y[g]); ymax = fmax(ymax, y[g]);
    xmin = fmin(xmin, x[g * xs]); xmax = fmax(xmax, x[g * xs]);
    gx = fmin(gx, x[g * xs] - x[(g - 1) * xs]);
    gy = fmin(gy, y[g] - y[g - 1]);
  }
  for (int g = 1; g < ng; g++) {
    if (g <= 8 && x[g * xs] - x[(g - 1) * xs] >= ymax - ymin) nx |= 1 << (g - 1);
    if (g <= 8 && y[g] - y[g - 1] >= xmax - xmin) ny |= 1 << (g - 1);
  }
  const double tiny = ldexp(xmin + ymin, -46);
  int f = ys_ | (xs_ << 1);
  if (ys_ && xs_) {
    if (ymax - ymin <= gx) f |= 4;
    if (xmax - xmin <= gy) f |= 8;
    if (ymax - ymin <= fmax(gx, tiny)) f |= 16;
    if (xmax - xmin <= fmax(gy, tiny)) f |= 32;
  }
  int inv = 0;
  for (int a = 0; a < ng * ng && inv < 255; a++)
    for (int b = a + 1; b < ng * ng; b++)
      if (inds[a] > inds[b]) { inv++; if (inv >= 255) break; }
  (void)xy;
  o[0] = (unsigned char)f; o[1] = (unsigned char)nx; o[2] = (unsigned char)ny; o[3] = (unsigned char)inv;
}

/* ------------------------------------------------------------------ opacity */

/* interpolate_Xsection, clima_radtran_types.f90:890-917 (res indexed by layer) */
static void interpolate_xs(const OrcRadtran *r, const OrcXs *xs, int l, const double *T,
                           double *res, const unsigned char *pair_reuse) {
  int nz = r->nz;
  if (xs->dim == 0) {
    for (int j = 0; j < nz; j++) res[j] = xs->data[l];
  } else {
    for (int j = 0; j < nz; j++) {
      if (pair_reuse[j]) {
        res[j] = res[j - 1];
      } else {
        double TT = fmin(fmax(T[j], xs->T_min), xs->T_max);
        double val = orc_interp1d(xs->nT, xs->temp, xs->data + (size_t)l * xs->nT, TT);
        res[j] = orc_ten2power(val);
      }
    }
  }
}

/* OpticalProperties_compute_opacity + k_rorr, clima_radtran_types.f90:574-888 */
static int compute_opacity(OrcRadtran *r, const double *P, const double *T,
                           const double *densities, const double *dz,
                           const double *pdensities, const double *radii, char *err) {
  const int nz = r->nz, nsp = r->nsp, ng = r->ng, nw = r->nw;
  if (r->nk == 0) { set_err(err, "There are no k-distributions, yet there must be some to compute total opacity."); return 1; }

  /* pre-pass :599-633 */
  double *log10P = dupd(NULL, nz), *cols = dupd(NULL, (size_t)nz * nsp), *foreign_col = dupd(NULL, nz);
  unsigned char *pair_reuse = (unsigned char *)calloc(nz, 1);
  int *ierrs = (int *)calloc(nw, sizeof(int));
  const double pair_tol = 1.0e-12;
  for (int j = 0; j < nz; j++) log10P[j] = log10(P[j]);
  for (int i = 0; i < nsp; i++)
    for (int j = 0; j < nz; j++) cols[j + (size_t)i * nz] = densities[j + (size_t)i * nz] * dz[j];
  if (r->has_cont) {
    for (int j = 0; j < nz; j++) {
      foreign_col[j] = 0.0;
      for (int i = 0; i < nsp; i++)
        if (i != r->LH2O) foreign_col[j] = foreign_col[j] + cols[j + (size_t)i * nz];
    }
  }
  if (nz % 2 == 0) {
    for (int j = 1; j < nz; j += 2) {
      int ok = 1;
      ok = ok && is_close(P[j], P[j - 1], pair_tol);
      ok = ok && is_close(T[j], T[j - 1], pair_tol);
      for (int i = 0; i < nsp; i++) ok = ok && is_close(cols[j + (size_t)i * nz], cols[j - 1 + (size_t)i * nz], pair_tol);
      if (radii && r->npart > 0)
        for (int i = 0; i < r->np; i++) ok = ok && is_close(radii[j + (size_t)i * nz], radii[j - 1 + (size_t)i * nz], pair_tol);
      pair_reuse[j] = (unsigned char)ok;
    }
  }

#pragma omp parallel for schedule(dynamic, 4)
  for (int l = 0; l < nw; l++) { /* :640-769 */
    const int nxy = ng * ng;
    double *ks = dupd(NULL, (size_t)r->nk * ng * nz); /* ks(i)%k(j,k) -> [i][k][j] */
    double *cia = dupd(NULL, (size_t)(r->ncia ? r->ncia : 1) * nz);
    double *pxs = dupd(NULL, (size_t)(r->npxs ? r->npxs : 1) * nz);
    double *H2O = dupd(NULL, nz), *foreign = dupd(NULL, nz);
    int npa = r->npart ? r->npart : 1;
    double *w0p = dupd(NULL, (size_t)npa * nz), *qextp = dupd(NULL, (size_t)npa * nz), *gtp = dupd(NULL, (size_t)npa * nz);
    double *tausp_1 = dupd(NULL, (size_t)npa * nz);
    double *tausg = dupd(NULL, nz), *taua = dupd(NULL, nz), *tauc = dupd(NULL, nz), *tausc = dupd(NULL, nz),
           *w0c = dupd(NULL, nz), *g0c = dupd(NULL, nz), *tausp = dupd(NULL, nz), *taup = dupd(NULL, nz),
           *taua_1 = dupd(NULL, nz), *tau = dupd(NULL, nz), *w0 = dupd(NULL, nz), *gt = dupd(NULL, nz);
    double *tau_k = dupd(NULL, (size_t)nz * ng), *tau_k1 = dupd(NULL, ng), *tau_xy = dupd(NULL, (size_t)nz * nxy),
           *tau_xy1 = dupd(NULL, nxy), *tau_xy2 = dupd(NULL, nxy), *wxy1 = dupd(NULL, nxy), *wxy_e = dupd(NULL, nxy + 1);
    int *inds = (int *)malloc(sizeof(int) * nxy);

    /* k-distributions :649-662 */
    for (int i = 0; i < r->nk; i++) {
      const OrcKtable *k = &r->k[i];
      const double *slab = k->log10k + (size_t)l * k->nT * k->nP * k->ng;
      for (int kk = 0; kk < k->ng; kk++) {
        double *kv = ks + ((size_t)i * ng + kk) * nz;
        for (int j = 0; j < nz; j++) {
          if (pair_reuse[j]) {
            kv[j] = kv[j - 1];
          } else {
            double TT = fmin(fmax(T[j], k->T_min), k->T_max);
            double log10PP = fmin(fmax(log10P[j], k->log10P_min), k->log10P_max);
            /* f(iP,iT) for g-point kk: element stride ng in P, ng*nP in T */
            double v = interp2d_strided(k->nP, k->nT, k->log10P, k->temp, slab + kk, k->ng,
                                        (long)k->ng * k->nP, log10PP, TT);
            kv[j] = orc_ten2power(v);
          }
        }
      }
    }
    /* CIA / photolysis / continuum / particles interpolation :665-683 */
    for (int i = 0; i < r->ncia; i++) interpolate_xs(r, &r->cia[i], l, T, cia + (size_t)i * nz, pair_reuse);
    for (int i = 0; i < r->npxs; i++) interpolate_xs(r, &r->pxs[i], l, T, pxs + (size_t)i * nz, pair_reuse);
    if (r->has_cont) { /* interpolate_WaterContinuum :919-945 */
      for (int j = 0; j < nz; j++) {
        if (pair_reuse[j]) {
          H2O[j] = H2O[j - 1];
          foreign[j] = foreign[j - 1];
        } else {
          double TT = fmin(fmax(T[j], r->cont_Tmin), r->cont_Tmax);
          H2O[j] = orc_ten2power(orc_interp1d(r->cont_nT, r->cont_temp, r->cont_H2O + (size_t)l * r->cont_nT, TT));
          foreign[j] = orc_ten2power(orc_interp1d(r->cont_nT, r->cont_temp, r->cont_foreign + (size_t)l * r->cont_nT, TT));
        }
      }
    }
    for (int i = 0; i < r->npart; i++) { /* interpolate_Particle :947-983 */
      const OrcPart *p = &r->part[i];
      int ierr = 0;
      for (int j = 0; j < nz; j++) {
        if (pair_reuse[j]) {
          w0p[j + (size_t)i * nz] = w0p[j - 1 + (size_t)i * nz];
          qextp[j + (size_t)i * nz] = qextp[j - 1 + (size_t)i * nz];
          gtp[j + (size_t)i * nz] = gtp[j - 1 + (size_t)i * nz];
          continue;
        }
        double rp = radii[j + (size_t)p->p_ind * nz];
        if (rp < p->r_min || rp > p->r_max) {
          rp = fmin(fmax(rp, p->r_min), p->r_max);
          ierr = 1;
        }
        w0p[j + (size_t)i * nz] = orc_interp1d(p->nrad, p->radii, p->w0 + (size_t)l * p->nrad, rp);
        qextp[j + (size_t)i * nz] = orc_interp1d(p->nrad, p->radii, p->qext + (size_t)l * p->nrad, rp);
        gtp[j + (size_t)i * nz] = orc_interp1d(p->nrad, p->radii, p->gt + (size_t)l * p->nrad, rp);
      }
      ierrs[l] += ierr;
    }
    /* Rayleigh :686-693 (TOA-first index n) */
    for (int n = 0; n < nz; n++) tausg[n] = 0.0;
    for (int i = 0; i < r->nray; i++) {
      int j = r->ray[i].sp1;
      for (int k = 0; k < nz; k++) {
        int n = nz - 1 - k;
        tausg[n] = tausg[n] + r->ray[i].data[l] * cols[k + (size_t)j * nz];
      }
    }
    /* CIA :696-704 */
    for (int n = 0; n < nz; n++) taua[n] = 0.0;
    for (int i = 0; i < r->ncia; i++) {
      int j = r->cia[i].sp1, jj = r->cia[i].sp2;
      for (int k = 0; k < nz; k++) {
        int n = nz - 1 - k;
        taua[n] = taua[n] + cia[k + (size_t)i * nz] * densities[k + (size_t)j * nz] * densities[k + (size_t)jj * nz] * dz[k];
      }
    }
    /* photolysis/absorption :707-713 */
    for (int i = 0; i < r->npxs; i++) {
      int j = r->pxs[i].sp1;
      for (int k = 0; k < nz; k++) {
        int n = nz - 1 - k;
        taua[n] = taua[n] + pxs[k + (size_t)i * nz] * cols[k + (size_t)j * nz];
      }
    }
    /* continuum :716-723 */
    if (r->has_cont) {
      int L = r->LH2O;
      for (int k = 0; k < nz; k++) {
        int n = nz - 1 - k;
        taua[n] = taua[n] + H2O[k] * densities[k + (size_t)L * nz] * cols[k + (size_t)L * nz] +
                  foreign[k] * densities[k + (size_t)L * nz] * foreign_col[k];
      }
    }
    /* custom opacity :540-572, :726-730; tiny when unset :558-562 */
    for (int k = 0; k < nz; k++) {
      int n = nz - 1 - k; /* reversed to TOA-first :727-729 */
      if (!r->cust_on) {
        tauc[n] = 2.2250738585072014e-308; w0c[n] = 2.2250738585072014e-308; g0c[n] = 2.2250738585072014e-308;
      } else {
        /* evaluate without clamping: dintrv extrapolates from the end intervals (:348-350) */
        double x = log10(P[k] * 1.0e6);
        const size_t o = (size_t)l * r->cust_nP;
        tauc[n] = orc_interp1d(r->cust_nP, r->cust_log10P, r->cust_dtau + o, x) * dz[k];
        w0c[n] = orc_interp1d(r->cust_nP, r->cust_log10P, r->cust_w0 + o, x);
        g0c[n] = orc_interp1d(r->cust_nP, r->cust_log10P, r->cust_g0 + o, x);
      }
      tausc[n] = w0c[n] * tauc[n];
    }
    /* particles :733-757 */
    for (int n = 0; n < nz; n++) { tausp[n] = 0.0; taup[n] = 0.0; }
    for (int i = 0; i < r->npart; i++) {
      int j = r->part[i].p_ind;
      for (int k = 0; k < nz; k++) {
        int n = nz - 1 - k;
        double rr = radii[k + (size_t)j * nz];
        double taup_1 = qextp[k + (size_t)i * nz] * PI * (rr * rr) * pdensities[k + (size_t)j * nz] * dz[k];
        taup[n] = taup[n] + taup_1;
        tausp_1[n + (size_t)i * nz] = w0p[k + (size_t)i * nz] * taup_1;
        tausp[n] = tausp[n] + tausp_1[n + (size_t)i * nz];
      }
    }
    for (int n = 0; n < nz; n++) gt[n] = 0.0;
    for (int i = 0; i < r->npart; i++) {
      for (int k = 0; k < nz; k++) {
        int n = nz - 1 - k;
        gt[n] = gt[n] + gtp[k + (size_t)i * nz] * tausp_1[n + (size_t)i * nz] / fmax(TAU_MIN, (tausp[n] + tausg[n] + tausc[n]));
      }
    }
    for (int n = 0; n < nz; n++) gt[n] = gt[n] + g0c[n] * tausc[n] / fmax(TAU_MIN, (tausp[n] + tausg[n] + tausc[n]));
    for (int n = 0; n < nz; n++) gt[n] = fmin(gt[n], MAX_GT);

    /* k_rorr :780-888 */
    {
      int j1 = r->k[0].sp_ind;
      for (int i = 0; i < ng; i++)
        for (int j = 0; j < nz; j++) tau_k[j + (size_t)i * nz] = ks[((size_t)0 * ng + i) * nz + j] * cols[j + (size_t)j1 * nz];
      for (int jj = 1; jj < r->nk; jj++) {
        int j2 = r->k[jj].sp_ind;
        for (int i = 0; i < ng; i++)
          for (int j = 0; j < ng; j++)
            for (int z = 0; z < nz; z++)
              tau_xy[z + (size_t)(j + i * ng) * nz] = tau_k[z + (size_t)i * nz] + ks[((size_t)jj * ng + j) * nz + z] * cols[z + (size_t)j2 * nz];
        for (int i = 0; i < nz; i++) {
          if (pair_reuse[i]) {
            for (int j = 0; j < ng; j++) tau_k[i + (size_t)j * nz] = tau_k[i - 1 + (size_t)j * nz];
            if (g_census) g_census[(((size_t)l * (r->nk - 1) + jj - 1) * nz + i) * 4] = 64;
          } else {
            for (int j = 0; j < nxy; j++) tau_xy1[j] = tau_xy[i + (size_t)j * nz];
            orc_mrgrnk(nxy, tau_xy1, inds);
            if (g_census) {
              double yv[64] = {0};
              for (int j = 0; j < ng && j < 64; j++) yv[j] = ks[((size_t)jj * ng + j) * nz + i] * cols[i + (size_t)j2 * nz];
              census_step(g_census + (((size_t)l * (r->nk - 1) + jj - 1) * nz + i) * 4, ng, tau_k + i, nz, yv, tau_xy1, inds);
            }
            for (int j = 0; j < nxy; j++) {
              tau_xy2[j] = tau_xy1[inds[j]];
              wxy1[j] = r->wxy[inds[j]];
            }
            weights_to_bins(nxy, wxy1, wxy_e);
            orc_rebin(nxy, wxy_e, tau_xy2, ng, r->wbin_e, tau_k1);
            for (int j = 0; j < ng; j++) tau_k[i + (size_t)j * nz] = tau_k1[j];
          }
        }
      }
      double *res_tau_band = r->tau_band + (size_t)l * nz;
      for (int n = 0; n < nz; n++) res_tau_band[n] = 0.0;
      for (int i = 0; i < ng; i++) {
        for (int k = 0; k < nz; k++) taua_1[nz - 1 - k] = tau_k[k + (size_t)i * nz];
        for (int n = 0; n < nz; n++) tau[n] = tausg[n] + taua[n] + taup[n] + taua_1[n] + tauc[n];
        for (int j = 0; j < nz; j++) {
          if (tau[j] <= TAU_MIN) w0[j] = 0.0;
          else w0[j] = fmin(MAX_W0, (tausg[j] + tausp[j] + tausc[j]) / tau[j]);
        }
        double *rt = r->tau + ((size_t)l * ng + i) * nz, *rw = r->w0 + ((size_t)l * ng + i) * nz;
        for (int n = 0; n < nz; n++) {
          rt[n] = tau[n];
          res_tau_band[n] = res_tau_band[n] + tau[n] * r->wbin[i];
          rw[n] = w0[n];
        }
      }
      for (int n = 0; n < nz; n++) r->g[n + (size_t)l * nz] = gt[n];
    }

    free(ks); free(cia); free(pxs); free(H2O); free(foreign); free(w0p); free(qextp); free(gtp); free(tausp_1);
    free(tausg); free(taua); free(tauc); free(tausc); free(w0c); free(g0c); free(tausp); free(taup);
    free(taua_1); free(tau); free(w0); free(gt);
    free(tau_k); free(tau_k1); free(tau_xy); free(tau_xy1); free(tau_xy2); free(wxy1); free(wxy_e); free(inds);
  }

  int bad = 0;
  for (int l = 0; l < nw; l++) if (ierrs[l] != 0) bad = 1;
  free(log10P); free(cols); free(foreign_col); free(pair_reuse); free(ierrs);
  if (bad) { set_err(err, "Opacity computation failed in one or more wavelength bins."); return 1; }
  return 0;
}

/* ------------------------------------------------------------------ radiate */

/* radiate, src/radtran/clima_radtran_radiate.f90:7-196 */
static void radiate(OrcRadtran *r, const OrcChannel *rtc, int is_ir, double diurnal_fac,
                    const double *photons_sol, int nzen, const double *zenith_u,
                    const double *zenith_weights, double T_surface, const double *T,
                    OrcWrk *out) {
  const int nz = r->nz, ng = r->ng;
#pragma omp parallel for schedule(dynamic, 4)
  for (int l = rtc->ind_start; l <= rtc->ind_end; l++) { /* :52-156 */
    int ll = l - rtc->ind_start;
    double *bplanck = dupd(NULL, nz + 1);
    double *f = (double *)calloc((size_t)9 * (nz + 1), sizeof(double));
    double *fup0 = f, *fup1 = f + (nz + 1), *fup2 = f + 2 * (nz + 1), *fdn0 = f + 3 * (nz + 1),
           *fdn1 = f + 4 * (nz + 1), *fdn2 = f + 5 * (nz + 1), *amean0 = f + 6 * (nz + 1),
           *amean1 = f + 7 * (nz + 1), *amean2 = f + 8 * (nz + 1);
    double emissivity = 0.0, albedo = 0.0, surf_rad;
    if (is_ir) { /* :63-71 */
      double avg_freq = 0.5 * (r->freq[l] + r->freq[l + 1]);
      bplanck[nz] = orc_planck_fcn(avg_freq, T_surface);
      for (int j = 0; j < nz; j++) bplanck[nz - 1 - j] = orc_planck_fcn(avg_freq, T[j]);
      emissivity = r->surface_emissivity[ll];
      albedo = 0.0;
    } else {
      emissivity = 0.0;
      albedo = r->surface_albedo[ll];
    }
    for (int i = 0; i < nz + 1; i++) { fup2[i] = 0.0; fdn2[i] = 0.0; amean2[i] = 0.0; }
    for (int ii = 0; ii < nzen; ii++) { /* :83-136 */
      for (int i = 0; i < nz + 1; i++) { fup1[i] = 0.0; fdn1[i] = 0.0; amean1[i] = 0.0; }
      for (int i = 0; i < ng; i++) {
        const double *tau = r->tau + ((size_t)l * ng + i) * nz;
        const double *w0 = r->w0 + ((size_t)l * ng + i) * nz;
        const double *g = r->g + (size_t)l * nz;
        if (!is_ir)
          orc_two_stream_solar(nz, tau, w0, g, zenith_u[ii], albedo, amean0, &surf_rad, fup0, fdn0);
        else
          orc_two_stream_ir(nz, tau, w0, g, emissivity, r->has_hard_surface, r->ir_tau_min, bplanck, fup0, fdn0);
        for (int n = 0; n < nz + 1; n++) {
          fup1[n] = fup1[n] + fup0[n] * r->wbin[i];
          fdn1[n] = fdn1[n] + fdn0[n] * r->wbin[i];
          if (!is_ir) amean1[n] = amean1[n] + amean0[n] * r->wbin[i];
        }
      }
      for (int n = 0; n < nz + 1; n++) {
        fup2[n] = fup2[n] + fup1[n] * zenith_weights[ii];
        fdn2[n] = fdn2[n] + fdn1[n] * zenith_weights[ii];
        if (!is_ir) amean2[n] = amean2[n] + amean1[n] * zenith_weights[ii];
      }
    }
    for (int i = 0; i < nz + 1; i++) { /* :140-154 */
      int n = nz - i;
      out->fup_a[i + (size_t)ll * (nz + 1)] = fup2[n];
      out->fdn_a[i + (size_t)ll * (nz + 1)] = fdn2[n];
      if (!is_ir) out->amean[i + (size_t)ll * (nz + 1)] = amean2[n];
    }
    for (int i = 0; i < nz; i++) out->tau_band[i + (size_t)ll * nz] = r->tau_band[(nz - 1 - i) + (size_t)l * nz];
    free(bplanck); free(f);
  }

  if (!is_ir) { /* :167-180 */
    for (int l = 0; l < rtc->nw; l++) {
      double avg_freq = 0.5 * (rtc->freq[l] + rtc->freq[l + 1]);
      double avg_wavl = 1.0e9 * C_LIGHT / avg_freq;
      for (int i = 0; i < nz + 1; i++) {
        size_t ix = i + (size_t)l * (nz + 1);
        out->fup_a[ix] = out->fup_a[ix] * photons_sol[l] * diurnal_fac;
        out->fdn_a[ix] = out->fdn_a[ix] * photons_sol[l] * diurnal_fac;
        out->amean[ix] = out->amean[ix] * photons_sol[l] * diurnal_fac;
        out->amean[ix] = out->amean[ix] * (avg_freq / avg_wavl);
        out->amean[ix] = out->amean[ix] * (avg_wavl / (PLANK * C_LIGHT * 1.0e16)) * (rtc->wavl[l + 1] - rtc->wavl[l]);
      }
    }
  }
  for (int i = 0; i < nz + 1; i++) { out->fup_n[i] = 0.0; out->fdn_n[i] = 0.0; } /* :184-192 */
  for (int l = 0; l < rtc->nw; l++) {
    double dfreq = rtc->freq[l] - rtc->freq[l + 1];
    for (int i = 0; i < nz + 1; i++) {
      out->fup_n[i] = out->fup_n[i] + out->fup_a[i + (size_t)l * (nz + 1)] * dfreq;
      out->fdn_n[i] = out->fdn_n[i] + out->fdn_a[i + (size_t)l * (nz + 1)] * dfreq;
    }
  }
}

/* check_inputs, clima_radtran.f90:417-445 (dimension checks :447-491 are the
 * caller's job at this pointer-based boundary; sizes are implied by nz/nsp/np) */
static int check_inputs(const OrcRadtran *r, const double *pdensities, const double *radii, char *err) {
  if ((pdensities && !radii) || (radii && !pdensities)) { set_err(err, "Both pdensities and radii must be arguments."); return 1; }
  if (r->np > 0 && !radii) { set_err(err, "The model contains particles but \"pdensities\" and \"radii\" are not arguments."); return 1; }
  return 0;
}

/* Radtran_radiate, clima_radtran.f90:221-318 */
int orc_radiate(OrcRadtran *r, double T_surface, const double *T, const double *P,
                const double *densities, const double *dz, const double *pdensities,
                const double *radii, int compute_solar, int compute_opacity_, char *err) {
  if (err) err[0] = 0;
  if (!r->finalized) { set_err(err, "Radtran is not finalized"); return 1; }
  if (check_inputs(r, pdensities, radii, err)) return 1;
  const int nz = r->nz;
  if (compute_opacity_)
    if (compute_opacity(r, P, T, densities, dz, pdensities, radii, err)) return 1;
  const double zero = 0.0, one = 1.0;
  radiate(r, &r->ir, 1, 0.0, &zero, 1, &zero, &one, T_surface, T, &r->wrk_ir); /* :262-283 */
  if (!compute_solar) { /* :286-289 */
    for (int i = 0; i < nz + 1; i++)
      r->f_total[i] = (r->wrk_sol.fdn_n[i] - r->wrk_sol.fup_n[i]) + (r->wrk_ir.fdn_n[i] - r->wrk_ir.fup_n[i]);
    return 0;
  }
  double *ps = dupd(NULL, r->sol.nw); /* :302 */
  for (int i = 0; i < r->sol.nw; i++) ps[i] = r->photons_sol[i] * r->photon_scale_factor;
  radiate(r, &r->sol, 0, r->diurnal_fac, ps, r->nzen, r->zenith_u, r->zenith_w, T_surface, T, &r->wrk_sol);
  free(ps);
  for (int i = 0; i < nz + 1; i++) /* :316 */
    r->f_total[i] = (r->wrk_sol.fdn_n[i] - r->wrk_sol.fup_n[i]) + (r->wrk_ir.fdn_n[i] - r->wrk_ir.fup_n[i]);
  return 0;
}

/* Radtran_TOA_fluxes, clima_radtran.f90:320-342 */
int orc_toa_fluxes(OrcRadtran *r, double T_surface, const double *T, const double *P,
                   const double *densities, const double *dz, const double *pdensities,
                   const double *radii, int compute_solar, int compute_opacity_,
                   double *ISR, double *OLR, char *err) {
  if (orc_radiate(r, T_surface, T, P, densities, dz, pdensities, radii, compute_solar, compute_opacity_, err)) return 1;
  int nz = r->nz;
  *ISR = (r->wrk_sol.fdn_n[nz] - r->wrk_sol.fup_n[nz]);
  *OLR = -(r->wrk_ir.fdn_n[nz] - r->wrk_ir.fup_n[nz]);
  return 0;
}

/* ------------------------------------------------------------------ getters */

void orc_dims(OrcRadtran *r, int *nz, int *nw, int *ng, int *nw_ir, int *nw_sol, int *ir_start, int *sol_start) {
  *nz = r->nz; *nw = r->nw; *ng = r->ng; *nw_ir = r->ir.nw; *nw_sol = r->sol.nw;
  *ir_start = r->ir.ind_start; *sol_start = r->sol.ind_start;
}

void orc_get_wrk(OrcRadtran *r, int which, double *fup_a, double *fdn_a, double *fup_n,
                 double *fdn_n, double *amean, double *tau_band) {
  OrcWrk *w = which ? &r->wrk_sol : &r->wrk_ir;
  int nwc = which ? r->sol.nw : r->ir.nw, nz = r->nz;
  if (fup_a) memcpy(fup_a, w->fup_a, sizeof(double) * (size_t)(nz + 1) * nwc);
  if (fdn_a) memcpy(fdn_a, w->fdn_a, sizeof(double) * (size_t)(nz + 1) * nwc);
  if (amean) memcpy(amean, w->amean, sizeof(double) * (size_t)(nz + 1) * nwc);
  if (tau_band) memcpy(tau_band, w->tau_band, sizeof(double) * (size_t)nz * nwc);
  if (fup_n) memcpy(fup_n, w->fup_n, sizeof(double) * (nz + 1));
  if (fdn_n) memcpy(fdn_n, w->fdn_n, sizeof(double) * (nz + 1));
}
void orc_get_f_total(OrcRadtran *r, double *f_total) { memcpy(f_total, r->f_total, sizeof(double) * (r->nz + 1)); }
void orc_get_opr(OrcRadtran *r, double *tau, double *w0, double *g, double *tau_band) {
  size_t n3 = (size_t)r->nz * r->ng * r->nw, n2 = (size_t)r->nz * r->nw;
  if (tau) memcpy(tau, r->tau, n3 * sizeof(double));
  if (w0) memcpy(w0, r->w0, n3 * sizeof(double));
  if (g) memcpy(g, r->g, n2 * sizeof(double));
  if (tau_band) memcpy(tau_band, r->tau_band, n2 * sizeof(double));
}
void orc_get_channel(OrcRadtran *r, int which, double *wavl, double *freq) {
  OrcChannel *c = which ? &r->sol : &r->ir;
  if (wavl) memcpy(wavl, c->wavl, sizeof(double) * (c->nw + 1));
  if (freq) memcpy(freq, c->freq, sizeof(double) * (c->nw + 1));
}
