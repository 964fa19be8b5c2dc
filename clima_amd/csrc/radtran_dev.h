// radtran_dev.h -- device-facing parameter blocks shared by the kernels and the host API.
//
// HBM layout (all f64 unless noted; "TOA-first" = index 0 is the top layer, as
// OpticalPropertiesResult in src/radtran/clima_radtran_types.f90:242-247, :862-865):
//
//   tables   log10k[s]      [nw][nT][nP][ng]   (the on-disk order, types_create.f90:1349-1358;
//                                               one (P,T) node = ng contiguous doubles = 64 B)
//            xs 0-D         [nw]               Rayleigh / photolysis / constant CIA
//            xs 1-D         [nw][nT]           log10 CIA, photolysis, H2O self/foreign continuum
//            particles      [nw][nrad] x3      w0, qext, g
//   column   T,P,dz [nz]; dens [nsp][nz]; pdens,radii [np][nz]   (ground-first, as the API)
//   prep     log10P, cols [nsp][nz], foreign_col, src (pair_reuse source layer),
//            per interpolation slot: left index ix[slot][nz] (i32) and weight q[slot][nz]
//   opr      tau, w0        [nw][ng][nz]       TOA-first; == Fortran (nz,ng,nw) column-major
//            g, tau_band    [nw][nz]
//   results  fup_a, fdn_a, amean [nw_ch][nz+1]; tau_band [nw_ch][nz]   ground-first,
//            == Fortran (nz+1,nw_ch) column-major (ClimaRadtranWrk, clima_radtran.f90:11-25)
//            flux_n         [4][nz+1]          ir_up, ir_dn, sol_up, sol_dn (the all-reduce payload)
//            f_total        [nz+1]
#pragma once

#include <hip/hip_runtime.h>

namespace clima {

// Crossing windows of the window-form rebin (kernels.hip rb_lo / rb_hi), indexed by the output edge
// 1..7: element range that can contain the crossing of that edge.  The host checks a handle's weights
// against them (rebin_windows_fit, radtran_api.hip).
constexpr int RB_WIN_LO[8] = {0, 1, 5, 10, 18, 27, 39, 52};        // any order of the sums
constexpr int RB_WIN_HI[8] = {0, 11, 24, 36, 45, 53, 58, 62};
constexpr int RB_TIGHT_LO[8] = {0, 5, 12, 19, 27, 35, 45, 55};     // x and y ascending: prefixes are down-sets of the 8x8 grid
constexpr int RB_TIGHT_HI[8] = {0, 8, 18, 28, 36, 44, 51, 58};

constexpr int MAX_K = 8;      // k-distribution species
constexpr int MAX_XS = 16;    // per Xsection kind
constexpr int MAX_PART = 4;   // particle species
constexpr int MAX_ZEN = 16;   // zenith angles passed by value to the two-stream kernel
constexpr int MAX_SLOTS = 2 * MAX_K + 2 * MAX_XS + 1 + MAX_PART + 1;  // + custom optical properties axis

// src/clima_const.f90:9-21
constexpr double PLANK = 6.62607004e-34;
constexpr double C_LIGHT = 299792458.0;
constexpr double K_BOLTZ_SI = 1.380649e-23;
constexpr double PI = 3.14159265358979323846e0;
constexpr double SIGMA_SI = 5.670374419e-8;
// src/radtran/clima_radtran_types.f90:9-11
constexpr double MAX_W0 = 0.99999;
constexpr double MAX_GT = 0.999999;
constexpr double TAU_MIN = 1.0e-20;
constexpr double LN10 = 2.302585092994045684017991454684364207601;
constexpr double TINY = 2.2250738585072014e-308;  // tiny(0.0_dp), clima_radtran_types.f90:558-562

struct XsDev {
  const double *data;  // dim 0: [nw]; dim 1: [nw][nT]
  int dim, sp1, sp2, nT, slot;
};

// One term of the continuum absorption taua (types.f90:696-723), in the reference's
// summation order: CIA pairs, photolysis/absorption, H2O self and foreign continuum.
// tau = sigma(bin, T_layer) * weight(layer); the weight is bin-independent and is formed
// once per call by the prep kernel.
constexpr int ABS_BATCH = 8;          // the opacity tile takes continuum terms in batches of this many; the host pads the list
constexpr int MAX_ABS = 2 * 16 + 2 + 6;  // (2 MAX_XS + 2 real terms, rounded up to a multiple of ABS_BATCH)
enum { ABS_CIA = 0, ABS_COLUMN = 1, ABS_H2O_SELF = 2, ABS_H2O_FOREIGN = 3, ABS_ZERO = 4 };  // ABS_ZERO: padding, weight 0
struct AbsEntry {
  const double *data;  // nT == 0: xs[nw]; else log10 xs [nw][nT]
  int nT;              // 0 for constant (0-D) cross sections
  int slot;            // interpolation slot (any valid slot when nT == 0)
  int kind, a, b;      // weight: CIA dens_a*dens_b*dz | cols_a | dens_a*cols_a | dens_a*foreign_col
};

struct KDev {
  const double *log10k;  // [nw][nT][nP][ng]
  int sp, nP, nT, slotP, slotT;
};

struct PartDev {
  const double *w0, *qext, *gt;  // [nw][nrad]
  int p_ind, nrad, slot;
};

// One interpolation axis evaluated per layer by the prep kernel: clamp, bracket
// (dintrv semantics, linear_interpolation_module.F90:348-350), weight.
struct SlotDev {
  const double *axis;
  int n;
  int source;  // 0 = log10(P), 1 = T, 2+p = radius of particle column p, -1 = log10(P*1e6) of the layer itself (custom opacity)
  double lo, hi;
  int flag_clamp;  // particles: out-of-range radius is an error (types.f90:973-976)
};

// Flags of an entry of the compact source-layer list (ColumnDev::meta)
constexpr int SRC_PAIR = 1 << 30;    // layer j+1 reuses this layer (pair_reuse, types.f90:621-632)
constexpr int SRC_EXACT = 1 << 29;   // ... and every input of layer j+1 is bitwise equal to layer j's
constexpr int SRC_LAYER = 0xffff;

// Element strides from one column of a batch to the next (0 everywhere for a single call).  Every
// per-column array of a group lives in ONE block per column with the same internal layout, so a
// group needs one stride: the column inputs, the prep results, the optical properties, the per-bin
// spectra, the level fluxes, the hand-off flags.
struct BatchStrides {
  size_t col, prep, opr, res, flux;
  int done;
};

struct ColumnDev {
  const double *T, *P, *dz, *dens, *pdens, *radii;  // dens [nsp][nz], pdens/radii [np][nz]
  // pair_reuse decided on the host when the column is uploaded (one definition of is_close for the
  // grid size and the kernels): meta[0] = nsrc, meta[1 + m] = source layer m | SRC_* flags
  // (ascending), meta[1 + nz + j] = source layer of layer j (j, or j-1 for the second of a pair)
  const int *meta;
  double *log10P, *cols, *foreign_col;
  double *absw;    // [nabs][nz] per-layer weights of the absorption entries
  int *ix;         // [nslots][nz]
  double *q;       // [nslots][nz]
  int *err_flag;   // device error word: id of the last call that clamped a particle radius
  const double *T_surface;  // device scalar
};

// custom optical properties (types.f90:432-572): per bin, linear in log10(P cgs)
struct CustomDev {
  const double *dtau, *w0, *g0;  // [nw][nP]
  int nP, slot, on;
};

struct OpacityParams {
  int nz, nw, ng, nsp, np;
  int bin_lo, nbins;  // opacity bins handled by this launch
  int nsrc;           // source layers of the column (host count, == meta[0]); nz for a batch (upper bound)
  int coop;           // ng = 8: use the group-of-lanes kernel (k_opacity_coop<8>) -- few items, latency matters
  int generic;        // ng != 8: the wave-per-item generic kernel instead of the group-of-lanes one (cross-check)
  int nk, nray, npart;
  KDev k[MAX_K];
  XsDev ray[MAX_XS];
  int nabs;
  AbsEntry abs[MAX_ABS];
  PartDev part[MAX_PART];
  const double *wbin, *wbin_e, *wxy;      // Ksettings (types.f90:84-94)
  const double *wbin_e_pad;               // wbin_e followed by +inf sentinels (edge stream of the rebin)
  const double *rorr_tab;                 // ng = 8: [E_1..E_8, w_0..w_7, 1/(E_(k+1)-E_k)], read into scalar registers by rorr_xys_asm.inc
  ColumnDev col;
  double *tau, *w0, *g, *tau_band;        // opr
  double *scat;                           // [nw][nz] scattering optical depth tausg + tausp + tausc of a layer (every g-point's
                                          // w0 is min(MAX_W0, scat / tau)): written always
  int write_w0;                           // 0: the two-stream part of the same fused grid forms w0 from scat itself, the
                                          // 8 x nw x nz values are neither written nor read back (the host materialises
                                          // them when something else asks for them)
  long long *stamps;                      // diagnostic build only (-DCLIMA_STAMPS); null otherwise
  int rebin_mode;                         // 0 window form (the weights fit the compiled crossing windows), 1 streaming,
                                          // 2 streaming with max(wxy) > min(wbin): an element may span several output edges
  CustomDev cust;
};

struct PrepParams {
  int ncol;                  // gridDim.y: columns of a batch (1 for a single call)
  BatchStrides bs;
  int nz, nsp, np, nslots, has_cont, LH2O;
  int call_id;  // stamped into err_flag by a failing call (monotonic, so the flag never needs a reset)
  SlotDev slots[MAX_SLOTS];
  int nabs;
  int abs_kind[MAX_ABS], abs_a[MAX_ABS], abs_b[MAX_ABS];
  int nzero;                 // output arrays cleared by spare blocks of the prep launch (per column: + c*bs.res)
  double *zero_ptr[6];
  size_t zero_count[6];
  ColumnDev col;
};

struct TwoStreamParams {
  int nz, ng;
  int ncols, nchunks, nc_shift;            // g-point columns per block, chunks per column, log2(ncols) or -1 (launcher)
  int col_base, accumulate;                // wave kernel: first g-point of this launch; add into the outputs (launcher)
  // batched shared-opacity IR launches (wave kernel): gridDim.z = b_ncol temperature columns,
  // T + z*b_T, T_surface + z*b_Ts, IR spectra + z*b_out; all 0 for a single column
  int b_ncol, b_T, b_Ts;
  size_t b_out;
  // task list: blocks [0, n_sol) are solar bins sol_lo.., blocks [n_sol, n_sol+n_ir) IR bins
  int n_sol, sol_lo, n_ir, ir_lo;          // channel-local first bin of this launch
  int sol_start, ir_start;                 // channel -> opacity-bin offset (RTChannel%ind_start)
  const double *tau, *w0, *g, *tau_band;   // opr
  const double *scat;                      // see OpacityParams
  int w0_from_scat;                        // the layers' w0 = min(MAX_W0, scat / tau) instead of the stored array
  const double *wbin;
  const double *freq;                      // opacity grid [nw+1]
  // test hook (clima_test_two_stream): Planck values at the levels given directly [nz+1] TOA-first
  // instead of computed from T (null in production), and a slot count above ceil(nz/64) (0: none)
  const double *bplanck;
  int force_slots;
  // every layer 2m+1 (ground-first) carries exactly the optical properties of layer 2m: pair_reuse marked
  // all pairs as exact copies (AdiabatClimate's doubled radiative grid); the fused grid then takes the
  // paired form of the two-stream part
  int paired;
  // IR
  const double *T, *T_surface;
  const double *emissivity;                // [nw_ir]
  int has_hard_surface;
  double ir_tau_min;
  // solar
  int nzen;
  const double *zen_u, *zen_w, *zen_iu;    // cos(zenith), weights, 1/cos (device arrays)
  // the same by value (kernarg segment -> scalar loads in the zenith loop); valid for nzen <= MAX_ZEN
  double zen_u_v[MAX_ZEN], zen_w_v[MAX_ZEN], zen_iu_v[MAX_ZEN];
  const double *albedo;                    // [nw_sol]
  const double *photons_sol;               // [nw_sol], unscaled
  double photon_scale_factor, diurnal_fac;
  const double *am_f1, *am_f2, *am_dw;     // per solar bin amean unit factors (radiate.f90:174-178)
  // outputs (channel-local bin index)
  double *ir_fup_a, *ir_fdn_a, *ir_tau_band;
  double *sol_fup_a, *sol_fdn_a, *sol_amean, *sol_tau_band;
};

// fused opacity + two-stream launch (k_fused), one workgroup per item.  Items of column c are the
// blocks [c*(n_op+n_ts), (c+1)*(n_op+n_ts)): first its n_op opacity tiles, then its n_ts two-stream
// items (one per (bin, g-point group), ordered by readiness).
struct FusedParams {
  int ncol;        // columns in this launch
  int n_op;        // opacity tiles per column (launcher; an upper bound when layers are reused)
  int n_ts;        // two-stream items per column (launcher)
  int call_id;     // value an opacity tile publishes in done[c][tile] when its results are out
  int max_spins;   // bound of a two-stream item's wait
  int *done;       // [ncol][n_op]
  int *timeout_flag;  // id of the last call in which a two-stream item's wait expired
  int slots;       // layers per lane of the two-stream part: ceil(nz/64) (launcher)
  int sol_early;   // solar bins whose opacities the first round of opacity tiles produces (launcher)
  BatchStrides bs;
};

struct IntegrateParams {
  int ncol;                                // gridDim.z
  BatchStrides bs;
  int nz;
  int nw_ir, nw_sol;
  int ir_lo, ir_n, sol_lo, sol_n;          // bins owned by this rank (all when unsharded)
  int do_solar;
  const double *ir_fup_a, *ir_fdn_a, *sol_fup_a, *sol_fdn_a;
  const double *ir_freq, *sol_freq;        // channel freq [nw_ch+1]
  double *flux_n;                          // [4][nz+1]
  double *flux_part;                       // bin-sharded handles: this rank's partial rows (flux_n is all-reduced in place); else null
  double *f_total;
  double *partial;                         // [4][nchunk][nz+1] chunk sums
  int nchunk;
  // handles with a communicator (radtran_comm_init_rank): one more word rides on the all-reduce behind the four
  // level rows -- 1 when this rank's fused hand-off timed out in the opacity pass `id_opr`, + 1024 when it did in
  // the pass of the last solar computation `id_sol` -- so that EVERY rank learns it and repeats the step unfused
  double *timeout_out;
  const int *timeout_flag;
  int id_opr, id_sol;
  // synchronous entry points (radtran_radiate_wrapper): the four level rows and the two device error words are
  // ALSO stored straight into the host's pinned result block (device address of it), so that the call ends with a
  // stream synchronise instead of a copy launch + synchronise.  Null otherwise.
  double *host_out;
  const int *err_words;
};

struct BatchIntegrateParams {
  int nz, ir_lo, ir_n, nchunk, col0;
  const double *fup_a, *fdn_a;   // [ncol][nw_ir][nz+1]
  size_t spec_stride;
  const double *freq;            // IR channel freq [nw_ir+1]
  double *partial;               // [ncol][2][nchunk][nz+1]
  const double *flux_n;          // the handle's [4][nz+1] (solar rows of the last solar call)
  double *out;                   // three arrays [ncol_total][nz+1], out_arr elements apart: fup_n, fdn_n, f_total
  size_t out_arr;
};

constexpr int GREEN_LB = 16;                 // levels per block of the far form (ir_green.inc)
// Which form a level takes for a deviation at k.  A unit change of bplanck[k] reaches the source terms of layers k-1 and
// k, i.e. the rows 2k-3 .. 2k+2 of E (those that exist); the levels k and k+1 (the bottoms of these two layers) and, for
// k <= 1, level 0 are evaluated explicitly; the levels above take the above form, those below the below form.
//   0 above, 1 below, 2 explicit (slot: 0 level k, 1 level k+1, 2 level 0)
__host__ __device__ inline int green_class(int lv, int k, int &slot) {
  slot = 0;
  if (lv >= k + 2) return 1;
  if (lv == 0 && k <= 1) { slot = 2; return 2; }
  if (lv <= k - 1) return 0;
  slot = lv - k;
  return 2;
}
// class of (deviation k, level block): 0 every level of the block takes the above form, 1 the below form, 2 mixed
__host__ __device__ inline int green_block_class(int k, int blk, int nz) {
  const int lv0 = blk * GREEN_LB, lv1 = lv0 + GREEN_LB - 1 < nz ? lv0 + GREEN_LB - 1 : nz;
  if (k >= 2 && lv1 <= k - 1) return 0;
  if (lv0 >= k + 2) return 1;
  return 2;
}

// The RCE Jacobian's batch as a response problem (ir_green.inc): columns that differ from a base profile in a few
// temperatures are F(base) + sum of unit responses times Planck differences.  q = (bin - ir_lo) * ng + g.
struct GreenParams {
  int nz, ng, n_ir, ir_lo, ir_start, NQ;
  const double *tau, *w0, *g, *wbin;       // opr (TOA-first layers), g-point weights
  const double *freq, *ir_freq;            // opacity grid [nw+1], IR channel [nw_ir+1]
  const double *emissivity;                // [nw_ir]
  int has_hard_surface;
  double ir_tau_min;
  // from the opacities alone (k_green_factor, k_green_unit, k_green_local), all q-major
  double *RW;                              // [NQ][7][2 nz] per row: c', 1/den, z, running products of phi and of -c' (mantissa, exponent)
  double *IS;                              // [NQ][6][nz+1]: per level, above form (log, up, down) and below form
  double *FS;                              // [NQ][2][blocks of 16 levels][34]: the same relative to a block's reference level
  double *DS;                              // [NQ][8][nz+1]: per k, (amplitude, log) pairs of the two forms, explicit values of the levels k, k+1
  double *D0;                              // [NQ][2][2]: level 0's explicit values (up, down) for k = 0, 1
  // deviations (sorted by k), padded to a multiple of 16
  int ndev, ndev_pad;
  const int *dev_k;
  const double *dev_T, *base_T;            // base_T[k], k TOA-first, k = nz: surface
  double *DB;                              // [n_ir][ndev_pad] + 64 doubles of slack (k_green_accum_far_mfma's tiles past the list)
  int qsplit;
  int nmix;                                // (deviation, level block) pairs of mixed class
  const int *mix_dev, *mix_blk;
  double *partial;                         // [qsplit][ndev_pad][2][nz+1] (levels TOA-first)
  int msplit;                              // bin splits of the mixed blocks' kernel
  double *partial_m;                       // [msplit][ndev_pad][2][nz+1]: its sums (only the mixed (deviation, level) pairs are written)
  // columns
  const int *col_src, *col_ptr, *col_dev;  // row of gen_out (or -1: base + responses), CSR of a column's deviations
  const double *gen_out;                   // [3][1 + dense columns][nz+1] (arrays gen_arr apart): column 0 = the base profile
  size_t gen_arr, out_arr;
  const double *flux_n;                    // the handle's level rows (solar rows of the last solar call)
  double *out;                             // [3][ncol][nz+1] (arrays out_arr apart)
};
void launch_green_factor(const GreenParams &p, hipStream_t s);
void launch_green_columns(const GreenParams &p, int ncol, hipStream_t s);
void launch_batch_ftotal(double *out, size_t out_arr, int ncol, int nz, const double *flux_n, hipStream_t s);
void launch_green_accumulate(const GreenParams &p, hipStream_t s);   // the accumulation alone (test hook: DB given)
int green_far_resident_waves();
void green_vector_form_set(int vector_form);   // test hook: the far accumulation's vector form (1) or matrix form (0)
int green_far_waves(int ndev, int nl);   // per bin split
int green_far_splits(int n_ir, int waves);

// launchers (kernels.hip)
void launch_prep(const PrepParams &p, hipStream_t s);
// returns false when ng is unsupported by the compiled kernels (ng = 8 tuned; 1..32 generic)
bool launch_opacity(const OpacityParams &p, hipStream_t s);
bool launch_twostream(TwoStreamParams &p, hipStream_t s, size_t *lds_bytes);
bool launch_twostream_w(TwoStreamParams &p, hipStream_t s, size_t *lds_bytes, bool zeroed);
// T + c*b_T, T_surface + c*b_Ts, IR spectra + c*b_out for column c of ncol
bool launch_twostream_ir_batch(TwoStreamParams &p, int ncol, hipStream_t s, int force_nw = 0);
bool fused_supported(const OpacityParams &op, const TwoStreamParams &ts);
int fused_half_form(const OpacityParams &op, const TwoStreamParams &ts, int ncol);
int fused_tiles(const OpacityParams &op);   // opacity tiles per column (size of a column's done[] slice)
bool launch_fused(const OpacityParams &op, TwoStreamParams &ts, FusedParams fp, hipStream_t s);
// test hook: the two-stream blocks of the fused grid alone (no opacity blocks), on opacities already in HBM
// (meta_nsrc: a device int, any value >= 1)
bool launch_fused_twostream_only(TwoStreamParams &ts, int slots, const int *meta_nsrc, hipStream_t s, bool half = false, bool paired = false);
int twostream_w_groups(int ng);
int twostream_w_half_slots(const TwoStreamParams &p);   // > 0: launch_twostream_w takes k_twostream_h (with 8 g-points: stores whole values)
void launch_integrate(const IntegrateParams &p, hipStream_t s);
bool integrate_one_launch(const IntegrateParams &p);
void launch_integrate_batch(const BatchIntegrateParams &p, int ncol, hipStream_t s);
int integrate_chunks(int nbins);
void launch_f_total(int nz, const double *flux_n, double *f_total, hipStream_t s);
void launch_scale(double *a, size_t n, double f, hipStream_t s);
void launch_copy(double *dst, const double *src, size_t n, hipStream_t s);  // src may be pinned host memory
void launch_test_rcp(const double *x, double *y, int n, hipStream_t s);
void launch_test_wscan(const double *a, const double *b, double *out, int nwaves, hipStream_t s);
void launch_w0_from_scat(const double *tau, const double *scat, double *w0, int nw, int ng, int nz, hipStream_t s);
void launch_test_exp(const double *x, double *y, int n, hipStream_t s);
void launch_test_exp_tab(const double *x, double *y, int n, int base10, hipStream_t s);

}  // namespace clima
