/* Sanitizer driver for the CPU oracle (TEST INFRASTRUCTURE): build a small synthetic Radtran through the
 * orc_* API, run radiate / TOA_fluxes in every call pattern, read every result back, destroy -- under
 * AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer (tools/sanitize.sh).  The reference's CI runs its
 * own test_radtran under valgrind (.github/workflows/test.yaml:48-55); this is the counterpart for the
 * restatement.  Exit code 0 and no sanitizer report = clean. */
#include "../../oracle/clima_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NZ 24
#define NSP 3
#define NP 1
#define NW 12
#define NG 8
#define NPR 4
#define NT 5
#define NRAD 6

static double frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (*s >> 8) / 16777216.0; }
#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "oracle_driver: %s\n", err); return 1; } } while (0)

int main(void) {
  char err[ORC_ERR_LEN + 1];
  unsigned seed = 7;
  double wavl[NW + 1];
  for (int i = 0; i <= NW; i++) wavl[i] = 100.0 * pow(1.0e4, (double)i / NW);
  for (int rep = 0; rep < 2; rep++) {
    OrcRadtran *r = orc_create(NZ, NSP, NP, NW, wavl);
    const double gx[NG] = {0.0199, 0.1017, 0.2372, 0.4083, 0.5917, 0.7628, 0.8983, 0.9801};
    double wts[NG] = {0.0506, 0.1112, 0.1569, 0.1813, 0.1813, 0.1569, 0.1112, 0.0506};
    double log10P[NPR] = {-5.0, -3.0, -1.0, 1.0}, temp[NT] = {100.0, 200.0, 300.0, 400.0, 600.0};
    (void)gx;
    for (int sp = 0; sp < 2; sp++) {
      double *k = malloc(sizeof(double) * NW * NT * NPR * NG);
      for (int l = 0; l < NW; l++)
        for (int it = 0; it < NT; it++)
          for (int ip = 0; ip < NPR; ip++) {
            double base = -24.0 + 3.0 * frand(&seed);
            for (int g = 0; g < NG; g++) k[((l * NT + it) * NPR + ip) * NG + g] = base + 0.4 * g + 0.002 * temp[it];
          }
      CHECK(orc_add_ktable(r, sp, NG, wts, NPR, log10P, NT, temp, k, err));
      free(k);
    }
    double xs0[NW], xs1[NW * NT];
    for (int l = 0; l < NW; l++) { xs0[l] = 1e-27 * (1.0 + l); for (int it = 0; it < NT; it++) xs1[l * NT + it] = -46.0 + 0.1 * it + 0.05 * l; }
    CHECK(orc_add_xsection(r, ORC_XS_RAYLEIGH, 0, 2, -1, 0, NULL, xs0, err));
    CHECK(orc_add_xsection(r, ORC_XS_CIA, 1, 2, 2, NT, temp, xs1, err));
    CHECK(orc_add_xsection(r, ORC_XS_PHOTOLYSIS, 0, 0, -1, 0, NULL, xs0, err));
    CHECK(orc_set_water_continuum(r, 0, NT, temp, xs1, xs1, err));
    double radii[NRAD], pw0[NW * NRAD], pq[NW * NRAD], pg[NW * NRAD];
    for (int i = 0; i < NRAD; i++) radii[i] = 1e-6 * pow(10.0, 0.5 * i);
    for (int i = 0; i < NW * NRAD; i++) { pw0[i] = 0.5 + 0.4 * frand(&seed); pq[i] = 1.0 + frand(&seed); pg[i] = 0.7 * frand(&seed); }
    CHECK(orc_add_particle(r, 0, NRAD, radii, pw0, pq, pg, err));
    /* a channel that does not fit the grid is refused with the reference's text; the object stays usable */
    double bad[3] = {150.0, 300.0, 999.0};
    if (orc_set_channels(r, 3, bad, 3, bad, err) == 0) { fprintf(stderr, "oracle_driver: bad channels accepted\n"); return 1; }
    CHECK(orc_set_channels(r, NW - 3, wavl + 4, NW - 1, wavl, err));
    double photons[NW];
    for (int l = 0; l < NW; l++) photons[l] = 1e-8 * (1.0 + l);
    CHECK(orc_set_photons_sol(r, NW - 2, photons, err));
    CHECK(orc_finalize(r, 4, 0.25, err));

    double T[NZ], P[NZ], dz[NZ], dens[NZ * NSP], pd[NZ], ra[NZ];
    for (int j = 0; j < NZ; j++) {
      P[j] = 1.0 * exp(-0.4 * j); T[j] = 290.0 - 4.0 * j; if (T[j] < 180.0) T[j] = 180.0; dz[j] = 1.0e5;
      const double n = P[j] * 1.0e6 / (1.380649e-16 * T[j]);
      dens[j] = 1e-3 * n; dens[NZ + j] = 4e-4 * n; dens[2 * NZ + j] = 0.78 * n;
      pd[j] = 10.0; ra[j] = 2e-5;
    }
    double isr, olr, isr2, olr2;
    CHECK(orc_toa_fluxes(r, 295.0, T, P, dens, dz, pd, ra, 1, 1, &isr, &olr, err));
    CHECK(orc_toa_fluxes(r, 295.0, T, P, dens, dz, pd, ra, 0, 0, &isr2, &olr2, err));   /* the RCE-Jacobian pattern */
    if (!(isr == isr2 && olr == olr2) || !(olr > 0.0) || !(isr > 0.0)) { fprintf(stderr, "oracle_driver: state not carried\n"); return 1; }
    /* the particle-radius clamp is an error (types.f90:973-976) */
    ra[3] = 1.0;
    if (orc_radiate(r, 295.0, T, P, dens, dz, pd, ra, 1, 1, err) == 0) { fprintf(stderr, "oracle_driver: radius outside the table accepted\n"); return 1; }
    ra[3] = 2e-5;
    int nz, nw, ng, nwi, nws, si, ss;
    orc_dims(r, &nz, &nw, &ng, &nwi, &nws, &si, &ss);
    double *a = malloc(sizeof(double) * (size_t)(nz + 1) * nw * 3 + sizeof(double) * (size_t)nz * nw + sizeof(double) * 2 * (nz + 1));
    for (int which = 0; which < 2; which++) {
      const int nc = which ? nws : nwi;
      double *fup = a, *fdn = fup + (nz + 1) * nc, *am = fdn + (nz + 1) * nc, *tb = am + (nz + 1) * nc, *un = tb + nz * nc, *dn = un + nz + 1;
      orc_get_wrk(r, which, fup, fdn, un, dn, am, tb);
    }
    free(a);
    double *tau = malloc(sizeof(double) * (size_t)nz * ng * nw * 2 + sizeof(double) * (size_t)nz * nw * 2);
    orc_get_opr(r, tau, tau + (size_t)nz * ng * nw, tau + (size_t)2 * nz * ng * nw, tau + (size_t)2 * nz * ng * nw + (size_t)nz * nw);
    free(tau);
    /* custom optical properties, then unset */
    double wv[3] = {2e2, 1e3, 1e5}, Pc[3] = {1e6, 1e4, 1e2}, dt[9], w0c[9], g0c[9];
    for (int i = 0; i < 9; i++) { dt[i] = 3e-8; w0c[i] = 0.5; g0c[i] = 0.3; }
    CHECK(orc_set_custom_optical_properties(r, 3, wv, 3, Pc, 3, 3, dt, 3, 3, w0c, 3, 3, g0c, err));
    CHECK(orc_radiate(r, 295.0, T, P, dens, dz, pd, ra, 1, 1, err));
    orc_unset_custom_optical_properties(r);
    double ft[NZ + 1];
    orc_get_f_total(r, ft);
    orc_destroy(r);
  }
  /* unit-level entry points on their edge cases */
  {
    double x[5] = {3, 1, 2, 1, 0}; int rk[5];
    orc_mrgrnk(5, x, rk);
    double ob[4] = {0, 1, 2, 3}, ov[3] = {1, 2, 3}, nb[3] = {0, 1.5, 3}, nv[2];
    orc_rebin(3, ob, ov, 2, nb, nv);
    if (fabs(nv[0] - (1.0 + 1.0) / 1.5) > 1e-15) { fprintf(stderr, "oracle_driver: rebin\n"); return 1; }
    double tau1[1] = {1e-9}, w1[1] = {0.3}, g1[1] = {0.1}, bp[2] = {1e-10, 2e-10}, fu[2], fd[2], am[2], sr;
    orc_two_stream_ir(1, tau1, w1, g1, 1.0, 1, 1e-6, bp, fu, fd);
    orc_two_stream_solar(1, tau1, w1, g1, 0.5, 0.3, am, &sr, fu, fd);
  }
  printf("oracle_driver: ok\n");
  return 0;
}
