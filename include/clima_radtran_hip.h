/*
 * clima_radtran_hip.h -- C ABI of the MI355X-native Radtran hot path.
 *
 * Drop-in boundary for Clima's `type Radtran` (src/radtran/clima_radtran.f90:31-85) and the
 * bind(c) shim above it (clima/fortran/Radtran.f90, ClimaRadtranWrk.f90, RTChannel.f90).
 * Conventions are the reference's (clima/fortran/clima_c_api.f90:5, AdiabatClimate.f90:169-190):
 *   - handles are opaque `void*` passed BY VALUE; every scalar is passed BY REFERENCE;
 *   - arrays are bare pointers with explicit extents; 2-D arrays are column-major;
 *   - errors: `char err[CLIMA_ERR_LEN+1]`, err[0]==0 <=> success; message text is API;
 *   - species / particle indices are 1-based (Fortran `sp_ind`, `p_ind`);
 *   - the object is not thread-safe; calls on one handle are serialised on its HIP stream.
 * No torch / C++ types cross this boundary.  All symbols are implemented by
 * clima_amd/csrc/libclima_radtran_hip.so; there is no CPU fallback: every entry point
 * that needs the GPU fails with a message in `err` when HIP is unavailable.
 */
#ifndef CLIMA_RADTRAN_HIP_H
#define CLIMA_RADTRAN_HIP_H

#ifndef __cplusplus
#include <stdbool.h>  /* logical(c_bool) <-> bool (1 byte), clima/fortran/Radtran.f90:211-227 */
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define CLIMA_ERR_LEN 1024

/* Xsection kinds: enum at src/radtran/clima_radtran_types.f90:40-42 */
#define CLIMA_XS_CIA 0
#define CLIMA_XS_RAYLEIGH 1
#define CLIMA_XS_ABSORPTION 2
#define CLIMA_XS_PHOTOLYSIS 3

/* ------------------------------------------------------------------------------------
 * Construction.  Replaces create_Radtran_2 (src/radtran/clima_radtran.f90:128-219): the
 * HDF5/YAML loaders (clima_radtran_types_create.f90) stay on the host side of the
 * boundary and hand over the tables they produce.
 * ---------------------------------------------------------------------------------- */

/* allocate / free an empty handle (pattern of allocate_adiabatclimate /
 * deallocate_adiabatclimate, clima/fortran/AdiabatClimate.f90:7-22) */
void allocate_radtran(void **ptr);
void deallocate_radtran(void *ptr);

/* nz layers, nsp gases (rad%ng), np particles, nw opacity bins, wavl[nw+1] nm ascending
 * (OpticalProperties%wavl, clima_radtran_types.f90:96-99) */
void radtran_create_begin(void *ptr, const int *nz, const int *nsp, const int *np,
                          const int *nw, const double *wavl, char *err);
/* Ktable (clima_radtran_types.f90:23-38).  log10k is the on-disk array
 * log10k(ngauss,npress,ntemp,nwav), column-major (types_create.f90:1349-1358). */
void radtran_add_ktable(void *ptr, const int *sp_ind, const int *ngauss,
                        const double *weights, const int *npress, const double *log10P,
                        const int *ntemp, const double *temp, const double *log10k,
                        char *err);
/* Xsection (clima_radtran_types.f90:44-55) after regridding to the bin grid
 * (types_create.f90:1171-1257): dim 0 -> data(nw) = xs; dim 1 -> data(ntemp,nw) = log10 xs.
 * sp_ind2 is used by CIA only. */
void radtran_add_xsection(void *ptr, const int *xs_type, const int *dim, const int *sp_ind1,
                          const int *sp_ind2, const int *ntemp, const double *temp,
                          const double *data, char *err);
/* WaterContinuum (clima_radtran_types.f90:68-77): log10_xs_*(ntemp,nw) */
void radtran_set_water_continuum(void *ptr, const int *LH2O, const int *ntemp,
                                 const double *temp, const double *log10_xs_H2O,
                                 const double *log10_xs_foreign, char *err);
/* ParticleXsection (clima_radtran_types.f90:57-66): w0,qext,gt (nrad,nw) */
void radtran_add_particle(void *ptr, const int *p_ind, const int *nrad, const double *radii,
                          const double *w0, const double *qext, const double *gt, char *err);
/* RTChannel edges in nm; index ranges resolved like create_RTChannel
 * (types_create.f90:226-270), same error text on mismatch. */
void radtran_set_channels(void *ptr, const int *n_ir_edges, const double *ir_wavl,
                          const int *n_sol_edges, const double *sol_wavl, char *err);
/* rad%photons_sol(sol%nw), mW/m^2/Hz -- output of read_stellar_flux (types_create.f90:9-78) */
void radtran_set_photons_sol(void *ptr, const int *n, const double *photons_sol, char *err);
/* zenith angles (Gauss-Legendre, clima_eqns.f90:26-41), albedo/emissivity defaults,
 * result arrays (clima_radtran.f90:162-217); uploads the tables to HBM. */
void radtran_create_end(void *ptr, const int *num_zenith_angles, const double *surface_albedo,
                        char *err);

/* ------------------------------------------------------------------------------------
 * The path.  The reference exposes no C symbol for Radtran%radiate itself (it is reached
 * through adiabatclimate_* only); these two are the ones a maintainer binds instead.
 * ---------------------------------------------------------------------------------- */

/* Radtran%radiate (clima_radtran.f90:221-318).  densities(dim1_d,dim2_d),
 * pdensities(dim1_p,dim2_p), radii(dim1_r,dim2_r) column-major; pass has_particles=0 and NULLs
 * when the optional arguments are absent.  Dimension errors reproduce check_inputs (:417-491),
 * including the separate `"radii" has the wrong input dimension.` of check_dimensions_p (:459). */
void radtran_radiate_wrapper(void *ptr, const double *T_surface, const int *dim_T,
                             const double *T, const int *dim_P, const double *P,
                             const int *dim1_d, const int *dim2_d, const double *densities,
                             const int *dim_dz, const double *dz, const int *has_particles,
                             const int *dim1_p, const int *dim2_p, const double *pdensities,
                             const int *dim1_r, const int *dim2_r, const double *radii,
                             const int *compute_solar,
                             const int *compute_opacity, char *err);
/* Radtran%TOA_fluxes (clima_radtran.f90:320-342) */
void radtran_toa_fluxes_wrapper(void *ptr, const double *T_surface, const int *dim_T,
                                const double *T, const int *dim_P, const double *P,
                                const int *dim1_d, const int *dim2_d, const double *densities,
                                const int *dim_dz, const double *dz, const int *has_particles,
                                const int *dim1_p, const int *dim2_p, const double *pdensities,
                                const int *dim1_r, const int *dim2_r, const double *radii,
                                const int *compute_solar,
                                const int *compute_opacity, double *ISR, double *OLR,
                                char *err);
/* Radtran%apply_radiation_enhancement (clima_radtran.f90:402-411) */
void radtran_apply_radiation_enhancement(void *ptr, const double *rad_enhancement);
/* clima/fortran/Radtran.f90:77-118 (same names and argument lists): custom optical properties,
 * wv nm, P dynes/cm^2 decreasing, dtau_dz / w0 / g0 (size(P), size(wv)) column-major
 * (src/radtran/clima_radtran_types.f90:432-548) */
void radtran_set_custom_optical_properties(void *ptr, const int *dim_wv, const double *wv, const int *dim_P,
                                           const double *P, const int *dim1_dtau_dz, const int *dim2_dtau_dz,
                                           const double *dtau_dz, const int *dim1_w0, const int *dim2_w0,
                                           const double *w0, const int *dim1_g0, const int *dim2_g0,
                                           const double *g0, char *err);
void radtran_unset_custom_optical_properties(void *ptr);

/* ---- HBM-resident form of the same call (no PCIe inside the timed region) ----
 * upload_column copies the column into the handle's device buffers; radiate_resident
 * enqueues opacity + IR + solar + integration on the handle's stream and returns without
 * a host sync; synchronize waits and surfaces device-side failures (e.g. the particle
 * radius clamp of interpolate_Particle, clima_radtran_types.f90:973-976). */
void radtran_upload_column(void *ptr, const double *T_surface, const double *T, const double *P,
                           const double *densities, const double *dz, const double *pdensities,
                           const double *radii, char *err);
void radtran_radiate_resident(void *ptr, const int *compute_solar, const int *compute_opacity,
                              char *err);
void radtran_synchronize(void *ptr, char *err);
/* device pointer to the packed level fluxes [ir_up, ir_dn, sol_up, sol_dn][nz+1] (f64):
 * the buffer a bin-sharded run all-reduces over RCCL. */
void radtran_flux_device_ptr(void *ptr, void **dptr, int *count);
/* restrict this handle to opacity bins of shard `rank` out of `world` (work-balanced
 * contiguous ranges; rank 0-based).  world=1 restores the full grid. */
void radtran_set_bin_shard(void *ptr, const int *rank, const int *world, char *err);
/* the ranges this handle owns: first opacity bin / count (0-based), then the channel-local
 * IR and solar sub-ranges */
void radtran_bin_shard_get(void *ptr, int *op_lo, int *op_n, int *ir_lo, int *ir_n, int *sol_lo,
                           int *sol_n);
/* after an external all-reduce of the flux buffer: the level rows have changed; f_total is formed
 * from them when the results are next read */
void radtran_finish_reduced(void *ptr, char *err);

/* ---- The multi-GPU step owned by the library (bins shard over the GPUs of a node, one RCCL
 * all-reduce of the per-layer integrated fluxes).  One process per GPU; every rank constructs the same Radtran
 * (after radtran_set_device) and joins a communicator.  From then on radtran_radiate_wrapper /
 * radtran_toa_fluxes_wrapper / radtran_radiate_resident work on the rank's own spectral bins and end with ONE
 * ncclAllReduce (sum, f64, 4 (nz+1) + 1 values) enqueued on the handle's stream: the level fluxes, f_total, ISR
 * and OLR every rank reads are those of the whole spectrum; per-bin spectra stay sharded (zeros outside the
 * rank's bins).  What is distributed is the sum over bins of src/radtran/clima_radtran_radiate.f90:184-192 and
 * f_total of clima_radtran.f90:287, 316.  A fused hand-off timeout on any rank is repeated by all of them
 * (the flag travels on the same all-reduce), never reported as an error. */
#define CLIMA_COMM_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */
/* select the HIP device of this process (before radtran_create_end) */
void radtran_set_device(const int *device, char *err);
/* rank 0: a fresh communicator id, to be handed to every rank by the caller's own means (MPI_Bcast, a file ...) */
void radtran_comm_unique_id(char *id /* [CLIMA_COMM_ID_BYTES] */, char *err);
/* collective over the nranks processes; also restricts the handle to the bins of shard (rank, nranks) */
void radtran_comm_init_rank(void *ptr, const int *nranks, const int *rank, const char *id, char *err);
/* the same with the id exchanged through a file that rank 0 creates (hosts without a message layer);
 * `path` (NUL-terminated) must be new for every job (a file older than ten minutes is taken for a leftover and ignored) */
void radtran_comm_init_file(void *ptr, const int *nranks, const int *rank, const char *path, char *err);
/* nranks (0: no communicator), rank, all-reduces enqueued so far */
void radtran_comm_get(void *ptr, int *nranks, int *rank, int *reduces);
/* leave the communicator; the handle works on the whole spectrum again */
void radtran_comm_destroy(void *ptr);
/* Column batch (BASELINE.json config 4): ncol independent Radtran%TOA_fluxes calls
 * (src/radtran/clima_radtran.f90:320-342), moved to HBM in one copy and enqueued back to back.
 * Inputs as radtran_toa_fluxes_wrapper with the column as the last dimension: T, P, dz (nz, ncol),
 * densities (nz, nsp, ncol), pdensities / radii (nz, np, ncol), T_surface (ncol).  ISR, OLR (ncol);
 * fluxes (nz+1, 5, ncol) = ir up, ir down, solar up, solar down, f_total, or NULL. */
void radtran_toa_fluxes_batch(void *ptr, const int *ncol, const double *T_surface, const double *T, const double *P,
                              const double *densities, const double *dz, const int *has_particles,
                              const double *pdensities, const double *radii, double *ISR, double *OLR,
                              double *fluxes, char *err);
/* Batched shared-opacity IR calls: what the RCE Jacobian does one call at a time
 * (src/adiabat/clima_adiabat_solve.f90:798-812 -> clima_radtran.f90:221-318 with
 * compute_solar = compute_opacity = .false.).  T is (nz, ncol) column-major, T_surface (ncol);
 * column c receives that call's wrk_ir%fup_n, wrk_ir%fdn_n and f_total in (nz+1, ncol) arrays.
 * Uses the opacities of the last compute_opacity call and the solar fluxes of the last solar
 * call; the handle's own wrk_ir / f_total are left untouched.  On a handle with a communicator
 * (radtran_comm_init_rank) every rank passes the same columns, works on its share of the bins and the library
 * all-reduces the batch's up / down arrays once (2 (nz+1) ncol doubles) before f_total is formed: every rank
 * receives the whole result.  A bin shard without a communicator (radtran_set_bin_shard) is refused.
 * The results come back through the handle's pinned block, in pieces the host copies out while the next one is on the
 * link.  After radtran_batch_pin_results_set(handle, 1) a caller that passes the same three result arrays as in its
 * previous batch call (the Jacobian's work arrays) has them page-locked from that second call on, like the arrays of
 * radtran_spectra_get_all, and filled by the device directly (402 x 403: 0.66 instead of 0.80 ms): they must then stay
 * allocated until radtran_spectra_release or the handle's end.  Off by default. */
void radtran_radiate_ir_batch(void *ptr, const int *ncol, const double *T_surface, const int *dim1_T,
                              const int *dim2_T, const double *T, double *fup_n, double *fdn_n,
                              double *f_total, char *err);
/* clima/fortran/Radtran.f90:41-75 (same names and argument lists): the YAML text of
 * OpticalProperties_opacities2yaml (src/radtran/clima_radtran_types.f90:328-430).  _1 allocates
 * the string and returns its length, _2 copies it into out_c (out_len + 1 chars) and frees it.
 * The names it prints come from radtran_set_names / radtran_set_opacity_labels (what the loaders
 * know: species and particle names one per line in index order; k-method, water-continuum model,
 * one data-set name per added particle cross section). */
void radtran_opacities2yaml_wrapper_1(void *ptr, int *out_len, void **out_cp);
void radtran_opacities2yaml_wrapper_2(void *ptr, void **out_cp, const int *out_len, char *out_c);
void radtran_set_names(void *ptr, const char *species_names, const char *particle_names, char *err);
void radtran_set_opacity_labels(void *ptr, const char *k_method, const char *water_continuum_model,
                                const char *particle_data, char *err);
/* Launch form of a compute_opacity call: 1 (default) = opacity and two-stream work in one grid
 * (k_fused: two-stream blocks start as soon as the opacity blocks of their bin are done),
 * 0 = one launch per kernel.  Same results to rounding; CLIMA_HIP_FUSED=0 sets the default off. */
void radtran_fused_set(void *ptr, const int *enable);
void radtran_fused_get(void *ptr, int *enabled);
/* Construction from files: `Radtran(settings_f, star_f, num_zenith_angles, surface_albedo, nz, datadir, err)`
 * (src/radtran/clima_radtran.f90:98-126) for hosts that do not link the reference's loaders
 * (src/radtran/clima_radtran_types_create.f90): the settings YAML's optical-properties block, the stellar spectrum and a
 * `photochem_clima_data`-style directory (kdistributions/, CIA/, xsections/, water_continuum/, rayleigh/,
 * aerosol_xsections/) are read on the host (HDF5 C library opened at run time, CLIMA_HDF5_LIB names it) and handed to
 * radtran_create_begin ... radtran_create_end.  Strings are NUL-terminated; error texts are the reference's.
 * radtran_load_from_files is the same without the upload (radtran_create_end): the handle is left in the "begun" state. */
void radtran_create_from_files(void *ptr, const char *settings_file, const char *star_file, const int *num_zenith_angles,
                               const double *surface_albedo, const int *nz, const char *datadir, char *err);
void radtran_load_from_files(void *ptr, const char *settings_file, const char *star_file, const int *nz, const char *datadir, char *err);
/* extents of a handle (layers, gases, particles, opacity bins, g-points) and the names it holds (newline-separated, into
 * caller buffers of `cap` bytes each): what a host needs to size its arrays after radtran_create_from_files */
void radtran_dims_get(void *ptr, int *nz, int *nsp, int *np, int *nw, int *ngauss);
void radtran_names_get(void *ptr, const int *cap, char *species_names, char *particle_names);
/* All per-bin spectra of the last call in one go: the seven arrays of the two result holders (`ClimaRadtranWrk`,
 * src/radtran/clima_radtran.f90:11-25: fup_a, fdn_a (nz+1, nw), tau_band (nz, nw) per channel, amean for the solar
 * one), column-major as the reference-named getters fill them.  The caller's arrays are page-locked on first use
 * (hipHostRegister) and stay so until the handle is destroyed or radtran_spectra_release is called -- call that
 * before freeing them; they are filled by asynchronous copies on the handle's stream and one synchronise
 * (config 2: 6.7 MB in ~0.16 ms where the seven getters take ~0.53).  do_solar false: the IR arrays only. */
void radtran_spectra_get_all(void *ptr, const bool *do_solar, const int *nlev, const int *nw_ir, const int *nw_sol,
                             double *ir_fup_a, double *ir_fdn_a, double *ir_tau_band,
                             double *sol_fup_a, double *sol_fdn_a, double *sol_amean, double *sol_tau_band, char *err);
void radtran_spectra_release(void *ptr);
void radtran_batch_pin_results_set(void *ptr, const int *flag);
void radtran_batch_pin_results_get(void *ptr, int *flag);
/* With 8 g-points, calls with at most `items` (bin, source layer) items -- a bin-sharded rank, a short
 * column -- run the opacity work in the group-of-lanes kernel (8 lanes per item: a fifth of the
 * dependent chain of the lane-per-item kernel at 2.4x its total work) with one launch per kernel;
 * larger calls take the lane-per-item kernel inside the fused grid.  Default 34816
 * (CLIMA_HIP_COOP_ITEMS); 0 turns the group-of-lanes form off.  Same results to rounding (6e-14). */
void radtran_coop_items_set(void *ptr, const int *items);
void radtran_coop_items_get(void *ptr, int *items);
/* How many polls a two-stream block of the fused grid spends on its opacity tiles' flags before it gives up and the call
 * is repeated through separate launches (radtran_fused_fallbacks_get counts those).  Default 400000 (~0.2 s;
 * CLIMA_HIP_FUSED_SPINS); 0 makes every wait expire -- how the tests and bench.py's multi-GPU probe force the repeat. */
void radtran_fused_spins_set(void *ptr, const int *spins);
void radtran_fused_spins_get(void *ptr, int *spins);
/* radtran_radiate_ir_batch, response form.  With the opacities fixed two_stream_ir
 * (src/radtran/clima_radtran_twostream.f90:156-295) is linear in the Planck values of the levels, and the columns of
 * the RCE Jacobian (src/adiabat/clima_adiabat_solve.f90:798-812) are one base profile with one or a few temperatures
 * changed each: such a column is F(base) + unit responses x Planck differences, which costs one exp and two FMAs per
 * (level, deviation, bin, g-point) instead of a solve per (column, bin, g-point).  mode 1 (default): taken when at least
 * 48 columns differ from the profile the batch's columns share in at most 8 temperatures and a cost model of the two
 * forms favours it -- tall grids, few changes per column -- (the others, and the base profile itself, go through the
 * general kernel); not on handles whose ir_tau_min was lowered below 1e-7 (the source slope dB / tau of such thin layers
 * costs the response form digits first: 1e-8 against 5e-10 of a row's maximum in the fuzz sweep); 0: never; 2: whenever
 * any column qualifies (tests).  CLIMA_HIP_IR_GREEN sets the default.  Same results to rounding (1e-12 of the level
 * fluxes at the reference's ir_tau_min).  `batches` counts the batches that took it. */
void radtran_ir_green_set(void *ptr, const int *mode);
void radtran_ir_green_get(void *ptr, int *mode, int *batches);
/* A two-stream block of the fused grid waits (bounded) for the opacity blocks of its bin.  If that
 * wait ever expires the call is NOT failed: the library computes it again through the separate
 * launches before results are handed out.  This counts such re-issues on the handle (0 in normal
 * operation; CLIMA_HIP_FUSED_SPINS=0 forces them, for tests). */
void radtran_fused_fallbacks_get(void *ptr, int *count);
/* HIP stream the handle launches on (for callers that order other work against it) */
void radtran_stream_get(void *ptr, void **stream);
/* per-kernel device time (HIP events on the handle's stream).  enable = 1 records events
 * around every kernel, enable = 2 around the dominant kernel (id 1, opacity) only, 0 turns
 * them off; kernel_time_get returns accumulated ms and launch count for kernel id
 * (0 prep, 1 opacity, 2 twostream, 3 integrate) and resets nothing. */
void radtran_profile_set(void *ptr, const int *enable);
/* record the events on every stride-th call only (default 1): an event pair drains the queue for a
 * few microseconds, which a throughput measurement should not pay on every call */
void radtran_profile_stride_set(void *ptr, const int *stride);
void radtran_kernel_time_get(void *ptr, const int *kernel_id, double *ms_total, int *launches,
                             char *err);
void radtran_profile_reset(void *ptr);
/* algorithmic-byte accounting of SURVEY.md 8(d) for the last uploaded column */
void radtran_algorithmic_bytes(void *ptr, double *bytes_tables_distinct, double *bytes_in,
                               double *bytes_out, double *bytes_tables_full, char *err);

/* N_PT (distinct (P,T) k-table nodes the last uploaded column touches, mean over the
 * k-tables, and the table's nP*nT) and N_T (the same for the 1-D temperature tables) */
void radtran_algorithmic_nodes(void *ptr, double *n_pt, double *n_pt_full, double *n_t, double *n_t_full, char *err);

/* bench hooks: n calls timed one by one with the host's steady clock INSIDE the library (us[n]) -- what a
 * Fortran / C caller sees, without a foreign-function layer's own cost.  clima_bench_toa_fluxes: the synchronous
 * drop-in call radtran_toa_fluxes_wrapper (arguments as there; compute_solar = compute_opacity = 1);
 * clima_bench_resident_sync: radtran_radiate_resident + radtran_synchronize per call (column already in HBM). */
void clima_bench_toa_fluxes(void *ptr, const int *n, const double *T_surface, const int *dim_T, const double *T,
                            const int *dim_P, const double *P, const int *dim1_d, const int *dim2_d,
                            const double *densities, const int *dim_dz, const double *dz, const int *has_particles,
                            const int *dim1_p, const int *dim2_p, const double *pdensities, const int *dim1_r,
                            const int *dim2_r, const double *radii, double *us, double *ISR, double *OLR, char *err);
void clima_bench_resident_sync(void *ptr, const int *n, double *us, char *err);
/* Timing only: one resident call's launches captured into a hipGraph and replayed (k launches per synchronise), us[n]
 * per pass; the replayed results are not valid (the captured call id makes the fused grid's waits trivial). */
void clima_bench_resident_graph(void *ptr, const int *n, const int *k, double *us, char *err);

/* test hook: y[i] = the kernels' device exp(x[i]) (used where the reference calls exp) */
void clima_test_device_exp(const int *n, const double *x, double *y, char *err);
/* FNV-1a (64 bit) over the handle's host-side tables (metadata as int32, values as float64 bytes, in the order they were
 * handed over), one digest per group -- digest[0..7]: extents + grid, k-tables, CIA, Rayleigh, absorption / photolysis,
 * continuum, particles, channels + stellar photons: how the from-files loader is held to the Python one without a device */
void clima_test_host_tables_digest(void *ptr, unsigned long long *digest);
void clima_test_device_exp_table(const int *n, const int *base10, const double *x, double *y, char *err);
/* test hook: the device reciprocal with 0, 1 and 2 Newton steps, and the device sqrt of |x|
 * (y: 4 arrays of n) */
void clima_test_device_rcp(const int *n, const double *x, double *y, char *err);
/* test hook: the DPP wave scans of the kernels on nwaves*64 values (out: 4 arrays of that length:
 * affine inclusive scan, the same through build+apply, shift up by one lane, lane reversal) */
void clima_test_wave_scan(const int *nwaves, const double *a, const double *b, double *out, char *err);

/* test hook: one column through the PRODUCTION two-stream kernels with tau, w0, g (nz, TOA-first) and
 * the Planck values of the levels (nz+1) given directly -- the arguments of the reference's
 * two_stream_ir / two_stream_solar (src/radtran/clima_radtran_twostream.f90:10-295).  ir_par =
 * {emissivity, has_hard_surface, ir_tau_min}, sol_par = {u0, Rsfc}; ng g-points of weights wbin (sum
 * 1) carry the same column.  form 0: k_twostream_w<slots>, 1: k_twostream, 2: two-stream part of
 * k_fused in its whole-wave form (slots 2..8, ng 8), 4: the same in the half-wave form (two g-point columns
 * per wave, slots = ceil(nz/32) = 3..7), 5: the same in the paired form (a column of pairwise identical
 * layers -- AdiabatClimate's doubled radiative grid -- even nz, slots 2, 4, 6, 8),
 * 3: k_twostream_ir_batch<slots, NW> (IR only; slots 1..4 in blocks of 8 g-point waves, 5..8 in blocks of 4),
 * 6: the same in blocks of 4 waves at every slot count.  Outputs (nz+1) TOA-first. */
void clima_test_two_stream(const int *nz, const int *ng, const int *form, const int *slots, const double *tau,
                           const double *w0, const double *g, const double *bplanck, const double *ir_par,
                           const double *sol_par, const double *wbin, double *ir_fup, double *ir_fdn,
                           double *sol_fup, double *sol_fdn, double *sol_amean, char *err);
/* The same for the response form of radtran_radiate_ir_batch (ir_green.inc): the column of clima_test_two_stream and
 * `ndev` changes of the Planck value at levels dev_k (TOA-first, nz = the surface) -> per change the change of the level
 * fluxes, resp_up / resp_dn (nz+1, ndev) TOA-first, summed over the g-points with the weights wbin -- produced by the
 * production kernels (k_green_factor, k_green_unit, k_green_local, k_green_accum_far, k_green_accum_mixed).
 * tests/test_gpu_golden.py holds F(base) + the changes to what the reference's two_stream_ir
 * (src/radtran/clima_radtran_twostream.f90:156-295) returns for the changed Planck profile.  nz >= 4. */
void clima_test_ir_response(const int *nz, const int *ng, const double *tau, const double *w0, const double *g,
                            const double *ir_par, const double *wbin, const int *ndev, const int *dev_k,
                            const double *dev_db, double *resp_up, double *resp_dn, char *err);

/* The far accumulation of the response form has two kernels: 0 (default) the matrix-core one (v_mfma_f64_16x16x4_f64),
 * 1 the vector one it replaced (also CLIMA_HIP_GREEN_MFMA=0).  Process-wide; tests hold the two against each other. */
void clima_test_green_far_form_set(const int *vector_form);

/* OpticalPropertiesResult (clima_radtran_types.f90:242-247), for parity checks:
 * tau,w0 (nz,ngauss,nw) and g,tau_band (nz,nw), column-major, TOA-first. */
void radtran_opr_get(void *ptr, double *tau, double *w0, double *g, double *tau_band, char *err);

/* ------------------------------------------------------------------------------------
 * Getters / setters with the reference's names and signatures
 * (clima/fortran/Radtran.f90:3-299).
 * ---------------------------------------------------------------------------------- */
void radtran_set_bolometric_flux_wrapper(void *ptr, const double *flux);             /* :3  */
void radtran_bolometric_flux_wrapper(void *ptr, double *flux);                       /* :12 */
void radtran_skin_temperature_wrapper(void *ptr, const double *bond_albedo, double *T_skin); /* :21 */
void radtran_equilibrium_temperature_wrapper(void *ptr, const double *bond_albedo, double *T_eq); /* :31 */
void radtran_zenith_u_get_size(void *ptr, int *dim1);                                /* :124 */
void radtran_zenith_u_get(void *ptr, const int *dim1, double *arr);                  /* :133 */
void radtran_zenith_u_set(void *ptr, const int *dim1, const double *arr);            /* :143 */
void radtran_zenith_weights_get(void *ptr, const int *dim1, double *arr);            /* field :53 */
void radtran_zenith_weights_set(void *ptr, const int *dim1, const double *arr);
void radtran_surface_albedo_get_size(void *ptr, int *dim1);                          /* :153 */
void radtran_surface_albedo_get(void *ptr, const int *dim1, double *arr);            /* :162 */
void radtran_surface_albedo_set(void *ptr, const int *dim1, const double *arr);      /* :172 */
void radtran_surface_emissivity_get_size(void *ptr, int *dim1);                      /* :182 */
void radtran_surface_emissivity_get(void *ptr, const int *dim1, double *arr);        /* :191 */
void radtran_surface_emissivity_set(void *ptr, const int *dim1, const double *arr);  /* :201 */
void radtran_has_hard_surface_get(void *ptr, bool *val);                             /* :211, logical(c_bool) */
void radtran_has_hard_surface_set(void *ptr, const bool *val);                       /* :220, logical(c_bool) */
void radtran_photon_scale_factor_get(void *ptr, double *val);                        /* :229 */
void radtran_photon_scale_factor_set(void *ptr, const double *val);                  /* :238 */
void radtran_ir_tau_min_get(void *ptr, double *val);                                 /* :247 */
void radtran_ir_tau_min_set(void *ptr, const double *val);                           /* :256 */
void radtran_diurnal_fac_get(void *ptr, double *val);                                /* field :51 */
void radtran_diurnal_fac_set(void *ptr, const double *val);
void radtran_ir_get(void *ptr, void **ptr1);                                         /* :265 */
void radtran_sol_get(void *ptr, void **ptr1);                                        /* :274 */
void radtran_wrk_ir_get(void *ptr, void **ptr1);                                     /* :283 */
void radtran_wrk_sol_get(void *ptr, void **ptr1);                                    /* :292 */
void radtran_f_total_get_size(void *ptr, int *dim1);                                 /* field :72 */
void radtran_f_total_get(void *ptr, const int *dim1, double *arr);
void radtran_photons_sol_get_size(void *ptr, int *dim1);                             /* field :65 */
void radtran_photons_sol_get(void *ptr, const int *dim1, double *arr);

/* ClimaRadtranWrk (clima/fortran/ClimaRadtranWrk.f90:7-123); ptr from radtran_wrk_*_get.
 * Getters copy device results out on first use after a radiate (lazy D2H). */
void climaradtranwrk_fup_a_get_size(void *ptr, int *dim1, int *dim2);
void climaradtranwrk_fup_a_get(void *ptr, const int *dim1, const int *dim2, double *arr);
void climaradtranwrk_fdn_a_get_size(void *ptr, int *dim1, int *dim2);
void climaradtranwrk_fdn_a_get(void *ptr, const int *dim1, const int *dim2, double *arr);
void climaradtranwrk_fup_n_get_size(void *ptr, int *dim1);
void climaradtranwrk_fup_n_get(void *ptr, const int *dim1, double *arr);
void climaradtranwrk_fdn_n_get_size(void *ptr, int *dim1);
void climaradtranwrk_fdn_n_get(void *ptr, const int *dim1, double *arr);
void climaradtranwrk_amean_get_size(void *ptr, int *dim1, int *dim2);
void climaradtranwrk_amean_get(void *ptr, const int *dim1, const int *dim2, double *arr);
void climaradtranwrk_tau_band_get_size(void *ptr, int *dim1, int *dim2);
void climaradtranwrk_tau_band_get(void *ptr, const int *dim1, const int *dim2, double *arr);

/* RTChannel (clima/fortran/RTChannel.f90:3-38); ptr from radtran_ir_get / radtran_sol_get */
void rtchannel_wavl_get_size(void *ptr, int *dim1);
void rtchannel_wavl_get(void *ptr, const int *dim1, double *arr);
void rtchannel_freq_get_size(void *ptr, int *dim1);
void rtchannel_freq_get(void *ptr, const int *dim1, double *arr);

#ifdef __cplusplus
}
#endif
#endif
