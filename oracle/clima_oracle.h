/*
 * clima_oracle.h -- CPU restatement of Clima's correlated-k two-stream radiative
 * transfer hot path (Radtran%radiate / TOA_fluxes).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (clima_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * Parity pin status (see DESIGN.md "Oracle"):
 *   - two_stream_ir / two_stream_solar / tridiag: PINNED against the reference's own
 *     src/radtran/clima_radtran_twostream.f90 compiled unmodified with amdflang into
 *     oracle/_ref/ (tests/golden/twostream_*.npz were generated from it).
 *   - everything else (compute_opacity, k_rorr, interpolation, radiate orchestration,
 *     planck): "parity unpinned" -- the reference holds no golden vectors or numerical
 *     assertions for this path (tests/test_radtran.f90:73-81 only prints) and its
 *     remaining sources cannot be compiled here without stand-ins for absent
 *     third-party modules (futils v0.1.14, fypp-generated code).  These functions are
 *     line-by-line restatements citing the reference file:line they follow.
 *   - futils v0.1.14 (Nicholaswogan/futils, not vendored): mrgrnk = ORDERPACK stable
 *     merge-sort ranking; rebin = conservative rebinning of piecewise-constant data;
 *     is_close = |a-b| <= tol*|b| style relative test.  Restated from their published
 *     definitions; call sites clima_radtran_types.f90:625-629, 840, 847.
 */
#ifndef CLIMA_ORACLE_H
#define CLIMA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_ERR_LEN 1024

/* xs_type enumerators: clima_radtran_types.f90:40-42 */
enum { ORC_XS_CIA = 0, ORC_XS_RAYLEIGH = 1, ORC_XS_ABSORPTION = 2, ORC_XS_PHOTOLYSIS = 3 };

typedef struct OrcRadtran OrcRadtran;

/* ---- construction (state after create_Radtran_2, clima_radtran.f90:128-219) ---- */
OrcRadtran *orc_create(int nz, int nsp, int np, int nw, const double *wavl /* nw+1, nm */);
void orc_destroy(OrcRadtran *r);
/* log10k in the on-disk layout log10k(ngauss,npress,ntemp,nwav) column-major
 * (types_create.f90:1349-1358), i.e. C [nw][nT][nP][ng]. sp_ind 0-based. */
int orc_add_ktable(OrcRadtran *r, int sp_ind, int ng, const double *weights, int nP,
                   const double *log10P, int nT, const double *temp, const double *log10k,
                   char *err);
/* dim 0: data[nw] = xs (linear); dim 1: data[nw][nT] = log10 xs (types_create.f90:1171-1257) */
int orc_add_xsection(OrcRadtran *r, int xs_type, int dim, int sp1, int sp2, int nT,
                     const double *temp, const double *data, char *err);
int orc_set_water_continuum(OrcRadtran *r, int LH2O, int nT, const double *temp,
                            const double *log10_H2O, const double *log10_foreign, char *err);
/* w0,qext,gt each [nw][nrad] */
int orc_add_particle(OrcRadtran *r, int p_ind, int nrad, const double *radii,
                     const double *w0, const double *qext, const double *gt, char *err);
/* channel wavelength edges (nm); index ranges resolved as create_RTChannel, types_create.f90:226-270 */
int orc_set_channels(OrcRadtran *r, int n_ir, const double *ir_wavl, int n_sol,
                     const double *sol_wavl, char *err);
int orc_set_photons_sol(OrcRadtran *r, int n, const double *photons_sol, char *err);
/* clima_radtran_types.f90:432-548; arrays (nP, nwv) column-major, P in dynes/cm^2 decreasing */
int orc_set_custom_optical_properties(OrcRadtran *r, int nwv, const double *wv, int nP, const double *P,
                                      int d1_t, int d2_t, const double *dtau_dz, int d1_w, int d2_w,
                                      const double *w0, int d1_g, int d2_g, const double *g0, char *err);
void orc_unset_custom_optical_properties(OrcRadtran *r);
int orc_finalize(OrcRadtran *r, int num_zenith_angles, double surface_albedo, char *err);

/* ---- public mutable fields (clima_radtran.f90:51-68) ---- */
void orc_set_zenith(OrcRadtran *r, int n, const double *u, const double *w);
void orc_get_zenith(OrcRadtran *r, double *u, double *w);
void orc_set_surface_albedo(OrcRadtran *r, const double *a /* nw_sol */);
void orc_set_surface_emissivity(OrcRadtran *r, const double *e /* nw_ir */);
void orc_set_scalars(OrcRadtran *r, double diurnal_fac, int has_hard_surface,
                     double ir_tau_min, double photon_scale_factor);
void orc_set_num_threads(int n);
void orc_census_set(unsigned char *buf); /* diagnostics: per mixing step order statistics, see clima_oracle.c */
int orc_get_max_threads(void);

/* ---- the path ---- */
/* Radtran_radiate, clima_radtran.f90:221-318. densities (nz,nsp) column-major,
 * pdensities/radii (nz,np) column-major or NULL.  returns 0 ok, else err filled. */
int orc_radiate(OrcRadtran *r, double T_surface, const double *T, const double *P,
                const double *densities, const double *dz, const double *pdensities,
                const double *radii, int compute_solar, int compute_opacity, char *err);
/* Radtran_TOA_fluxes, clima_radtran.f90:320-342 */
int orc_toa_fluxes(OrcRadtran *r, double T_surface, const double *T, const double *P,
                   const double *densities, const double *dz, const double *pdensities,
                   const double *radii, int compute_solar, int compute_opacity,
                   double *ISR, double *OLR, char *err);

/* ---- results ---- */
void orc_dims(OrcRadtran *r, int *nz, int *nw, int *ng, int *nw_ir, int *nw_sol,
              int *ir_start, int *sol_start);
/* which: 0 = ir, 1 = sol.  arrays column-major (nz+1,nw_ch) / (nz,nw_ch) */
void orc_get_wrk(OrcRadtran *r, int which, double *fup_a, double *fdn_a, double *fup_n,
                 double *fdn_n, double *amean, double *tau_band);
void orc_get_f_total(OrcRadtran *r, double *f_total);
/* OpticalPropertiesResult: tau,w0 (nz,ng,nw) column-major TOA-first; g,tau_band (nz,nw) */
void orc_get_opr(OrcRadtran *r, double *tau, double *w0, double *g, double *tau_band);
void orc_get_channel(OrcRadtran *r, int which, double *wavl, double *freq);

/* ---- unit-level entry points (pinned against oracle/_ref where noted) ---- */
void orc_two_stream_solar(int nz, const double *tau_in, const double *w0_in,
                          const double *gt_in, double u0, double Rsfc, double *amean,
                          double *surface_radiance, double *fup, double *fdn);
void orc_two_stream_ir(int nz, const double *tau, const double *w0, const double *gt,
                       double emissivity, int has_hard_surface, double tau_min,
                       const double *bplanck, double *fup, double *fdn);
void orc_tridiag(int n, double *a, double *b, double *c, double *d);
double orc_planck_fcn(double nu, double T);
double orc_ten2power(double y);
double orc_interp1d(int n, const double *x, const double *f, double xv);
double orc_interp2d(int nx, int ny, const double *x, const double *y, const double *f,
                    double xv, double yv);
void orc_mrgrnk(int n, const double *x, int *irank /* 0-based */);
void orc_rebin(int n_old, const double *old_bins, const double *old_vals, int n_new,
               const double *new_bins, double *new_vals);
void orc_gauss_legendre(int n, double *x, double *w);

#ifdef __cplusplus
}
#endif
#endif
