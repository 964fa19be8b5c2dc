// radtran_api.hip -- host side of the C ABI declared in include/clima_radtran_hip.h.
//
// Mirrors `type Radtran` (src/radtran/clima_radtran.f90:31-85): the handle owns the
// tables and work arrays in HBM, a HIP stream, and host mirrors of the small results.
// There is no CPU fallback anywhere in this file: if HIP is unavailable every compute
// entry point reports the HIP error in `err`.
#include "../../include/clima_radtran_hip.h"
#include "radtran_dev.h"

#include <rccl/rccl.h>   // the bin-sharded step's one collective; backend "nccl" on ROCm IS RCCL
#include <sys/stat.h>
#include <unistd.h>
#include <ctime>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <set>
#include <string>
#include <utility>
#include <limits>
#include <vector>

using namespace clima;

namespace {

constexpr unsigned MAGIC = 0xC11AAD17u;

void set_err(char *err, const std::string &msg) {
  if (!err) return;
  std::strncpy(err, msg.c_str(), CLIMA_ERR_LEN);
  err[CLIMA_ERR_LEN] = 0;
}
void clear_err(char *err) {
  if (err) err[0] = 0;
}

struct HipFail {
  std::string msg;
};
#define HIPCHK(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      throw HipFail{std::string("HIP error in " #call ": ") + hipGetErrorString(e_)};        \
  } while (0)
#define NCCLCHK(call)                                                                        \
  do {                                                                                       \
    ncclResult_t e_ = (call);                                                                \
    if (e_ != ncclSuccess)                                                                   \
      throw HipFail{std::string("RCCL error in " #call ": ") + ncclGetErrorString(e_)};      \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  bool owned = true;
  void view(T *ptr, size_t count) {  // a window into another buffer (not freed here)
    release();
    p = ptr; n = count; owned = false;
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
  }
  void upload(const std::vector<T> &v) {
    if (!(p && owned && n == v.size())) alloc(v.size());   // (same size: the allocation is kept)
    if (!v.empty()) HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  void zero(hipStream_t s = nullptr) {
    if (n) HIPCHK(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    n = 0;
    owned = true;
  }
  ~DevBuf() { release(); }
};

struct KTabHost {
  int sp, ng, nP, nT;
  std::vector<double> weights, log10P, temp, log10k;
  DevBuf<double> d_log10k, d_log10P, d_temp;
};
struct XsHost {
  int type, dim, sp1, sp2, nT;
  std::vector<double> temp, data;
  DevBuf<double> d_data, d_temp;
};
struct PartHost {
  int p_ind, nrad;
  std::vector<double> radii, w0, qext, gt;
  DevBuf<double> d_radii, d_w0, d_qext, d_gt;
};

struct Radtran;
struct ChannelObj {  // RTChannel, clima_radtran_types.f90:263-269
  Radtran *parent = nullptr;
  int which = 0;
  int ind_start = 0, ind_end = -1, nw = 0;  // 0-based, inclusive end
  std::vector<double> wavl, freq;
  DevBuf<double> d_freq;
};
struct WrkObj {  // ClimaRadtranWrk, clima_radtran.f90:11-25
  Radtran *parent = nullptr;
  int which = 0;
  DevBuf<double> fup_a, fdn_a, amean, tau_band;
};

struct Radtran {
  unsigned magic = MAGIC;
  int state = 0;  // 0 allocated, 1 begun, 2 finalized
  int nz = 0, nsp = 0, np = 0, nw = 0, ng = 0;
  std::vector<double> wavl, freq;
  std::vector<KTabHost *> k;
  std::vector<XsHost *> cia, ray, pxs;
  std::vector<PartHost *> part;
  std::vector<int> part_slot;
  bool has_cont = false;
  int LH2O = -1, cont_nT = 0;
  std::vector<double> cont_temp, cont_H2O, cont_foreign;
  DevBuf<double> d_cont_temp, d_cont_H2O, d_cont_foreign;
  std::vector<double> wbin, wbin_e, wxy;
  DevBuf<double> d_wbin, d_wbin_e, d_wbin_e_pad, d_wxy, d_freq, d_rorr_tab;
  ChannelObj ir, sol;
  WrkObj wrk_ir, wrk_sol;
  // public fields (clima_radtran.f90:51-68)
  double diurnal_fac = 0.5;
  std::vector<double> zenith_u, zenith_w, surface_albedo, surface_emissivity, photons_sol;
  bool has_hard_surface = true;
  double ir_tau_min = 1.0e-6;
  double photon_scale_factor = 1.0;
  bool fields_dirty = true;
  DevBuf<double> d_zen_u, d_zen_w, d_zen_iu, d_partial, d_albedo, d_emis, d_photons, d_am_f1, d_am_f2, d_am_dw;
  // column + prep
  int nslots = 0;
  // names for opacities2yaml (the loaders know them; optional: radtran_set_names / radtran_set_opacity_labels)
  std::vector<std::string> species_names, particle_names, particle_data;
  std::string k_method_name = "RandomOverlapResortRebin", continuum_model = "MT_CKD";
  std::vector<SlotDev> slots;
  // custom optical properties (types.f90:432-548): tables [nw][nP] on the device, axis log10(P cgs) ascending
  bool cust_on = false;
  int cust_nP = 0;
  std::vector<double> cust_axis;
  DevBuf<double> d_cust_axis, d_cust_dtau, d_cust_w0, d_cust_g0;
  // batched shared-opacity IR calls (radtran_radiate_ir_batch)
  DevBuf<double> d_bT, d_bTs, d_bup, d_bdn, d_bpartial, d_bout;
  // ... and its response form (ir_green.inc): work arrays, deviation lists, the general sub-batch's rows
  DevBuf<double> d_green, d_green_acc, d_green_in, d_gen_out;
  int green_last_n = -1;           // columns of the last batch, when it took the response form (the next one of that size starts its opacity-only part early)
  double *h_bout = nullptr;        // pinned: the batch's three result arrays on their way to the caller's
  size_t h_bout_n = 0;
  char *h_green = nullptr;         // pinned: what the host hands the response form (base profile, deviation lists)
  size_t h_green_n = 0;
  DevBuf<int> d_green_idx;
  int ir_green_mode = 1;           // CLIMA_HIP_IR_GREEN: 0 never, 1 when enough columns are sparse deviations of one profile, 2 whenever any is
  long ir_green_batches = 0;       // batches that took the response form (radtran_ir_green_batches_get)
  DevBuf<double> d_col;  // [T_surface | T | P | dz | dens | pdens | radii | meta (ints: nsrc, source list, source of every layer)]
  size_t meta_ofs = 0;   // doubles before the meta ints in a column block
  int nsrc = 0;          // source layers of the resident column (pair_reuse decided at upload)
  bool all_pairs_exact = false;  // ... and every layer is half of an exact pair (the doubled radiative grid)
  DevBuf<double> d_prep;  // one block: [log10P | cols | foreign_col | absw | q | ix (ints)]
  size_t prep_count = 0;
  std::vector<AbsEntry> abs_entries;  // continuum terms in the reference's summation order (+ zero-weight padding)
  DevBuf<double> d_zero_xs;
  DevBuf<int> d_err;
#ifdef CLIMA_STAMPS
  DevBuf<long long> d_stamps;
#endif
  double *h_col = nullptr;  // pinned staging
  double *h_col_dev = nullptr;  // the same buffer as the device addresses it (null: not mapped, use the copy engine)
  size_t col_count = 0;
  bool column_has_particles = false;
  bool column_loaded = false;
  hipEvent_t ev_upload = nullptr;   // marks the end of the last column copy out of the pinned staging buffer
  bool upload_pending = false;
  int call_id = 0, checked_id = 0;  // opacity passes enqueued / already checked for device errors
  long upload_id = 0, opr_upload_id = 0;   // columns uploaded so far / the upload the stored opacities were computed from
  std::string deferred_err;         // a failure met inside a getter without an `err` argument: reported by the next call that has one
  // fused hand-off (k_fused): bound of a two-stream block's wait, the last timed-out pass already
  // handled, the pass of the last solar computation, flags of the last call, re-issued calls so far
  int fused_max_spins = 400000, checked_timeout = 0, solar_id = 0, fused_fallbacks = 0;
  bool last_cs = true;
  std::vector<double> last_T, last_P, last_radii;  // host copy for byte accounting
  // opr: one block [tau | w0 | g | tau_band]; the four are views into it
  DevBuf<double> d_opr;
  size_t opr_count = 0;
  DevBuf<double> d_tau, d_w0, d_g, d_tau_band, d_scat;
  bool opr_valid = false;
  // results
  DevBuf<double> d_small;     // flux_n[4*(nz+1)] | f_total[nz+1] | err flag slot: one D2H copy per call
  DevBuf<double> d_flux_n, d_f_total;   // views into d_small
  std::vector<std::pair<void *, size_t>> host_registered;   // caller arrays page-locked by radtran_spectra_get_all / radtran_radiate_ir_batch
  double *batch_out[3] = {nullptr, nullptr, nullptr};       // the result arrays of the last radtran_radiate_ir_batch call
  size_t batch_out_n = 0;
  int batch_shared_min = 2;                                 // CLIMA_HIP_BATCH_SHARED_MIN: batches of at most so many columns take the per-column kernel
  bool batch_pin_results = false;                           // radtran_batch_pin_results_set
  hipEvent_t bout_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // the batch's result pieces
  hipStream_t copy_streams[3] = {nullptr, nullptr, nullptr}; // radtran_spectra_get_all: the seven copies go out over four queues
  double *h_small = nullptr;  // pinned: flux_n[4*(nz+1)] | f_total[nz+1] | err flag (as double slot)
  double *h_small_dev = nullptr;   // the same block as the device addresses it (null: not mapped)
  bool want_host_out = false;      // set by the synchronous wrappers around their enqueue_radiate
  bool small_in_host = false;      // the last call's integration stored its rows into h_small itself: no copy to fetch them
  int *h_errflag = nullptr;
  std::vector<double> f_total;
  bool small_valid = false;
  bool w0_valid = true;            // false after a fused call whose tiles left w0 unwritten (ensure_w0 materialises it)
  // sharding
  DevBuf<double> d_flux_part;  // this rank's partial level rows (d_flux_n is all-reduced in place)
  int shard_rank = 0, shard_world = 1;
  // the library's own multi-GPU step (radtran_comm_init_rank): with a communicator every radiate() ends with ONE
  // ncclAllReduce of the 4 (nz+1) partial level fluxes (+ one status word) on the handle's stream
  ncclComm_t comm = nullptr;
  int comm_n = 1, comm_rank = 0, device = 0;
  long comm_reduces = 0;       // all-reduces enqueued so far (tests)
  double comm_status = 0.0;    // the reduced status word of the last fetched step (see IntegrateParams::timeout_out)
  int op_lo = 0, op_n = 0, ir_lo = 0, ir_n = 0, sol_lo = 0, sol_n = 0;
  // stream + profiling
  hipStream_t stream = nullptr;
  // column batches (radtran_toa_fluxes_batch): the column / level-flux buffers a call works on
  double *col_override = nullptr, *flux_override = nullptr, *ftot_override = nullptr;
  int nsrc_override = 0;
  DevBuf<double> d_cols_arena, d_flux_arena;
  // one-launch batches: per-column prep / opr / spectra blocks for the columns in flight
  DevBuf<double> d_prep_arena, d_opr_arena, d_res_arena;
  int batch_cols_in_flight = 64;
  bool batch_shared = true;        // radiate_ir_batch: temperature-independent work shared by the columns (CLIMA_HIP_BATCH_SHARED=0: one full solve per column)
  int rebin_mode = 1;              // 0 window form, 1 streaming, 2 streaming multi-edge (rebin_mode_for)
  long coop_items = 34816;         // ng = 8: at most this many (bin, source layer) items go to k_opacity_coop<8> (CLIMA_HIP_COOP_ITEMS)
  bool fused = true;               // opacity + two-stream in one grid (k_fused); CLIMA_HIP_FUSED=0 or radtran_fused_set turns it off
  bool generic_opacity = false;    // g-point counts other than 8: k_opacity_generic instead of the group-of-lanes kernel (CLIMA_HIP_GENERIC=1)
  bool ts_block_mode = false;      // CLIMA_HIP_TS_MODE=block when the handle was made: the workgroup-per-bin two-stream kernel
  int ts_ncols_env = 0;            // CLIMA_HIP_TS_NCOLS (that kernel's columns per block), 0: its own choice
  DevBuf<int> d_done;              // per opacity block: call id of its last completed run
  int profile = 0;   // 0 off, 1 HIP events around every kernel, 2 around the dominant kernel (id 1) only
  long timer_calls = 0;
  int profile_stride = 1;   // events on every profile_stride-th call only (bounds the cost of measuring)
  struct Ev { hipEvent_t a, b; int id; };
  std::vector<Ev> pending;
  std::vector<hipEvent_t> pool;
  double k_ms[4] = {0, 0, 0, 0};
  int k_n[4] = {0, 0, 0, 0};
  size_t ts_lds = 0;

  ~Radtran() {
    for (auto *x : k) delete x;
    for (auto *x : cia) delete x;
    for (auto *x : ray) delete x;
    for (auto *x : pxs) delete x;
    for (auto *x : part) delete x;
    for (auto &e : pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : pool) (void)hipEventDestroy(e);
    if (h_col) (void)hipHostFree(h_col);
    for (auto &e : host_registered) (void)hipHostUnregister(e.first);
    for (auto &cs : copy_streams) if (cs) (void)hipStreamDestroy(cs);
    for (auto &e : bout_ev) if (e) (void)hipEventDestroy(e);
    if (h_small) (void)hipHostFree(h_small);
    if (h_bout) (void)hipHostFree(h_bout);
    if (h_green) (void)hipHostFree(h_green);
    if (ev_upload) (void)hipEventDestroy(ev_upload);
    if (comm) (void)ncclCommDestroy(comm);
    if (stream) (void)hipStreamDestroy(stream);
    magic = 0;
  }
};

Radtran *as_rad(void *ptr) {
  Radtran *r = reinterpret_cast<Radtran *>(ptr);
  if (!r || r->magic != MAGIC) return nullptr;
  return r;
}

// futils is_close (fortran-stdlib form)
bool is_close(double a, double b, double tol) { return std::fabs(a - b) <= std::fabs(tol * std::max(std::fabs(a), std::fabs(b))); }

// futils gauss_legendre (setup only, clima_eqns.f90:26-41)
void gauss_legendre(int n, std::vector<double> &x, std::vector<double> &w) {
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  for (int i = 0; i < n; i++) {
    double z = std::cos(PI * (i + 0.75) / (n + 0.5)), pp = 0.0;
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
      if (std::fabs(z - z1) < 1e-16) break;
    }
    x[n - 1 - i] = z;
    w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
  }
}

int host_bracket(const std::vector<double> &xt, double x) {
  int n = (int)xt.size();
  if (x < xt[0]) return 0;
  if (x >= xt[n - 1]) return n - 2;
  int lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    int mid = (lo + hi) / 2;
    if (x < xt[mid]) hi = mid; else lo = mid;
  }
  return lo;
}

// create_RTChannel, types_create.f90:226-270
bool make_channel(Radtran *r, ChannelObj &c, int which, int n, const double *wavl, char *err) {
  int ind1 = 0, ind2 = 0;
  double best1 = INFINITY, best2 = INFINITY;
  for (int i = 0; i < r->nw + 1; i++) {
    double d1 = std::fabs(wavl[0] - r->wavl[i]), d2 = std::fabs(wavl[n - 1] - r->wavl[i]);
    if (d1 < best1) { best1 = d1; ind1 = i; }
    if (d2 < best2) { best2 = d2; ind2 = i; }
  }
  const char *msg = "The wavelength bins are not compatible with the k-distribution wavelength bins.";
  if (n != ind2 - ind1 + 1) { set_err(err, msg); return false; }
  for (int i = 0; i < n; i++)
    if (!is_close(wavl[i], r->wavl[ind1 + i], 1.0e-7)) { set_err(err, msg); return false; }
  c.parent = r;
  c.which = which;
  c.nw = n - 1;
  c.wavl.assign(wavl, wavl + n);
  c.freq.resize(n);
  for (int i = 0; i < n; i++) c.freq[i] = C_LIGHT / (wavl[i] * 1.0e-9);
  c.ind_start = ind1;
  c.ind_end = ind2 - 1;
  return true;
}

void upload_fields(Radtran *r) {
  if (!r->fields_dirty) return;
  // kernels of earlier, unsynchronised calls may still be reading the old values
  if (r->stream) HIPCHK(hipStreamSynchronize(r->stream));
  r->d_zen_u.upload(r->zenith_u);
  r->d_zen_w.upload(r->zenith_w);
  {
    std::vector<double> iu(r->zenith_u.size());
    for (size_t i = 0; i < iu.size(); i++) iu[i] = 1.0 / r->zenith_u[i];
    r->d_zen_iu.upload(iu);
  }
  r->d_albedo.upload(r->surface_albedo);
  r->d_emis.upload(r->surface_emissivity);
  r->d_photons.upload(r->photons_sol);
  r->fields_dirty = false;
}

// Which rebin form the opacity kernels may use for these g-point weights (ng = 8).  The window form
// evaluates, for output edge k, only the sorted elements [RB_WIN_LO[k], RB_WIN_HI[k]]; the element j*
// that crosses E_k (C_{j*-1} < E_k <= C_{j*}) must be among them WHATEVER order the sort produces.
// C_j is bounded by the sums of the j+1 smallest / largest pair weights, so
//   j* >= first j with (sum of the j+1 largest)  >= E_k,   j* <= first j with (sum of the j+1 smallest) >= E_k.
// Both bounds are taken with a relative slack far above the rounding of a 64-term sum.
int rebin_mode_for(const std::vector<double> &wbin, const std::vector<double> &wbin_e, const std::vector<double> &wxy) {
  const int multi = (*std::max_element(wxy.begin(), wxy.end()) > *std::min_element(wbin.begin(), wbin.end())) ? 2 : 1;
  if (wbin.size() != 8) return multi;
  if (const char *e = getenv("CLIMA_HIP_REBIN")) { if (std::strcmp(e, "stream") == 0) return multi; }
  std::vector<double> s(wxy);
  std::sort(s.begin(), s.end());
  if (s[0] <= 0.0) return multi;
  double lo[64], hi[64], a = 0.0, b = 0.0;
  for (int j = 0; j < 64; j++) { a += s[j]; b += s[63 - j]; lo[j] = a; hi[j] = b; }
  for (int k = 1; k < 8; k++) {
    const double E = wbin_e[k];
    int jlo = 0, jhi = 0;
    while (jlo < 63 && hi[jlo] < E * (1.0 - 1e-9)) jlo++;
    while (jhi < 63 && lo[jhi] < E * (1.0 + 1e-9)) jhi++;
    if (jlo < RB_WIN_LO[k] || jhi > RB_WIN_HI[k]) return multi;
  }
  // the tight table (x and y ascending): lightest / heaviest down-set (Young diagram within the 8x8
  // grid of pairs (i, j), weight wbin(i)*wbin(j)) of every size, by enumeration of the 12 870 diagrams
  double dlo[65], dhi[65];
  for (int n = 0; n <= 64; n++) { dlo[n] = 1e300; dhi[n] = -1.0; }
  double rowcum[8][9];
  for (int i = 0; i < 8; i++) { rowcum[i][0] = 0.0; for (int j = 0; j < 8; j++) rowcum[i][j + 1] = rowcum[i][j] + wbin[i] * wbin[j]; }
  int len[8];
  std::function<void(int, int, int, double)> rec = [&](int i, int maxlen, int size, double wt) {
    if (i == 8) { dlo[size] = std::min(dlo[size], wt); dhi[size] = std::max(dhi[size], wt); return; }
    for (int rl = 0; rl <= maxlen; rl++) { len[i] = rl; rec(i + 1, rl, size + rl, wt + rowcum[i][rl]); }
  };
  rec(0, 8, 0, 0.0);
  for (int k = 1; k < 8; k++) {
    const double E = wbin_e[k];
    int jlo = 0, jhi = 0;
    while (jlo < 63 && dhi[jlo + 1] < E * (1.0 - 1e-9)) jlo++;
    while (jhi < 63 && dlo[jhi + 1] < E * (1.0 + 1e-9)) jhi++;
    if (jlo < RB_TIGHT_LO[k] || jhi > RB_TIGHT_HI[k]) return multi;
  }
  return 0;
}

// relative cost of a bin's opacity work, IR solve, solar solve (base + per zenith angle)
constexpr double SHARD_W_OP = 3.0, SHARD_W_IR = 1.0, SHARD_W_SOL0 = 0.6, SHARD_W_SOLZ = 0.24;

void compute_shard(Radtran *r) {
  // contiguous opacity-bin ranges balanced by work: opacity 3, IR solve 1, solar
  // solve 0.6 + 0.24 nzen -- the measured device-time ratios of the three kinds of work on MI355X
  // (tools/gpu_balance.py); rank 0-based.
  const int nw = r->nw, W = r->shard_world, R = r->shard_rank;
  std::vector<double> cost(nw + 1, 0.0);
  const int nzen = (int)r->zenith_u.size();
  double w_op = SHARD_W_OP, w_ir = SHARD_W_IR, w_s0 = SHARD_W_SOL0, w_sz = SHARD_W_SOLZ;
  if (const char *e = getenv("CLIMA_HIP_SHARD_COST")) (void)sscanf(e, "%lf,%lf,%lf,%lf", &w_op, &w_ir, &w_s0, &w_sz);  // tuning aid
  for (int l = 0; l < nw; l++) {
    double c = w_op;
    if (l >= r->ir.ind_start && l <= r->ir.ind_end) c += w_ir;
    if (l >= r->sol.ind_start && l <= r->sol.ind_end) c += w_s0 + w_sz * nzen;
    cost[l + 1] = cost[l] + c;
  }
  auto cut = [&](int k) {
    if (k <= 0) return 0;
    if (k >= W) return nw;
    double target = cost[nw] * k / W;
    return (int)(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin());
  };
  int lo = cut(R), hi = cut(R + 1);
  if (W == 1) { lo = 0; hi = nw; }
  r->op_lo = lo;
  r->op_n = hi - lo;
  auto clip = [&](const ChannelObj &c, int &clo, int &cn) {
    int a = std::max(lo, c.ind_start), b = std::min(hi - 1, c.ind_end);
    if (b < a) { clo = 0; cn = 0; } else { clo = a - c.ind_start; cn = b - a + 1; }
  };
  clip(r->ir, r->ir_lo, r->ir_n);
  clip(r->sol, r->sol_lo, r->sol_n);
}

hipEvent_t get_event(Radtran *r) {
  if (!r->pool.empty()) {
    hipEvent_t e = r->pool.back();
    r->pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIPCHK(hipEventCreate(&e));
  return e;
}

struct KernelTimer {
  Radtran *r;
  int id;
  hipEvent_t a = nullptr, b = nullptr;
  bool on() const {
    if (r->profile_stride > 1 && (r->timer_calls % r->profile_stride) != 0) return false;
    return r->profile == 1 || (r->profile == 2 && id == 1);
  }
  KernelTimer(Radtran *r_, int id_) : r(r_), id(id_) {
    if (on()) {
      a = get_event(r);
      b = get_event(r);
      HIPCHK(hipEventRecord(a, r->stream));
    }
  }
  void stop() {
    if (on()) {
      HIPCHK(hipEventRecord(b, r->stream));
      r->pending.push_back({a, b, id});
    }
  }
};

void resolve_events(Radtran *r) {
  for (auto &e : r->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      r->k_ms[e.id] += ms;
      r->k_n[e.id] += 1;
    }
    r->pool.push_back(e.a);
    r->pool.push_back(e.b);
  }
  r->pending.clear();
}

// Prep block of one column (d_prep or a slice of the batch arena)
void prep_views(Radtran *r, double *base, ColumnDev &c) {
  const size_t nz = r->nz, nab = std::max<size_t>(1, r->abs_entries.size()), ns = (size_t)r->nslots + 1;
  c.log10P = base;
  c.cols = c.log10P + nz;
  c.foreign_col = c.cols + nz * r->nsp;
  c.absw = c.foreign_col + nz;
  c.q = c.absw + nab * nz;
  c.ix = reinterpret_cast<int *>(c.q + ns * nz);
}
size_t prep_block_count(Radtran *r) {
  const size_t nz = r->nz, nab = std::max<size_t>(1, r->abs_entries.size()), ns = (size_t)r->nslots + 1;
  return nz * (2 + r->nsp + nab + ns) + (ns * nz + 1) / 2;
}

ColumnDev column_dev_at(Radtran *r, double *col_base, double *prep_base) {
  ColumnDev c;
  const int nz = r->nz;
  c.T_surface = col_base;
  c.T = col_base + 1;
  c.P = c.T + nz;
  c.dz = c.P + nz;
  c.dens = c.dz + nz;
  c.pdens = c.dens + (size_t)r->nsp * nz;
  c.radii = c.pdens + (size_t)r->np * nz;
  c.meta = reinterpret_cast<const int *>(col_base + r->meta_ofs);
  prep_views(r, prep_base, c);
  c.err_flag = r->d_err.p;
  return c;
}

ColumnDev column_dev(Radtran *r) {
  return column_dev_at(r, r->col_override ? r->col_override : r->d_col.p, r->d_prep.p);
}

// pair_reuse (clima_radtran_types.f90:621-632), decided here once per column so that the grid size
// and every kernel work from ONE definition: for even nz, layer j (1-based even) reuses layer j-1
// when P, T, every column densities*dz and (with particles) every radius agree to 1e-12 (is_close).
// meta: [0] = nsrc, [1 + m] = m-th source layer (0-based, ascending) | SRC_PAIR | SRC_EXACT,
// [1 + nz + j] = source layer of layer j.  Returns nsrc.
int build_meta(Radtran *r, const double *T, const double *P, const double *dz, const double *dens,
               const double *pdens, const double *radii, bool has_particles, int *meta) {
#pragma clang fp contract(off)  // the columns are stored products in the reference (opw%cols), never fused into the comparison
  const int nz = r->nz, nsp = r->nsp, np = r->np;
  const double tol = 1.0e-12;
  const bool use_radii = has_particles && radii && !r->part.empty();  // present(radii) .and. self%npart > 0
  int nsrc = 0;
  int *srcl = meta + 1, *src = meta + 1 + nz;
  for (int j = 0; j < nz; j++) {
    bool reuse = false, exact = false;
    if ((nz & 1) == 0 && (j & 1) == 1) {
      reuse = is_close(P[j], P[j - 1], tol) && is_close(T[j], T[j - 1], tol);
      exact = P[j] == P[j - 1] && T[j] == T[j - 1] && dz[j] == dz[j - 1];
      for (int i = 0; i < nsp && reuse; i++) {
        const volatile double ca = dens[(size_t)i * nz + j] * dz[j], cb = dens[(size_t)i * nz + j - 1] * dz[j - 1];  // opw%cols
        reuse = is_close(ca, cb, tol);
        exact = exact && dens[(size_t)i * nz + j] == dens[(size_t)i * nz + j - 1];
      }
      if (use_radii)
        for (int i = 0; i < np && reuse; i++) reuse = is_close(radii[(size_t)i * nz + j], radii[(size_t)i * nz + j - 1], tol);
      if (reuse && np > 0 && has_particles && pdens && radii)
        for (int i = 0; i < np; i++)
          exact = exact && pdens[(size_t)i * nz + j] == pdens[(size_t)i * nz + j - 1] &&
                  radii[(size_t)i * nz + j] == radii[(size_t)i * nz + j - 1];
      exact = exact && reuse;
    }
    if (reuse) {
      srcl[nsrc - 1] |= SRC_PAIR | (exact ? SRC_EXACT : 0);  // layer j-1 is the entry just written
      src[j] = j - 1;
    } else {
      srcl[nsrc++] = j;
      src[j] = j;
    }
  }
  for (int m = nsrc; m < nz; m++) srcl[m] = nz - 1;
  meta[0] = nsrc;
  return nsrc;
}

TwoStreamParams make_twostream_params(Radtran *r, const ColumnDev &col, bool compute_solar) {
  TwoStreamParams ts;
  const int nz = r->nz;
  std::memset(&ts, 0, sizeof(ts));
  ts.nz = nz; ts.ng = r->ng;
  if (r->ts_ncols_env) ts.ncols = r->ts_ncols_env;   // (the switches are read once, when the handle is made: a getenv per
                                                      //  call is a walk through the whole environment)
  ts.n_sol = compute_solar ? r->sol_n : 0; ts.sol_lo = r->sol_lo;
  ts.n_ir = r->ir_n; ts.ir_lo = r->ir_lo;
  ts.sol_start = r->sol.ind_start; ts.ir_start = r->ir.ind_start;
  ts.tau = r->d_tau.p; ts.w0 = r->d_w0.p; ts.g = r->d_g.p; ts.tau_band = r->d_tau_band.p;
  ts.wbin = r->d_wbin.p; ts.freq = r->d_freq.p;
  ts.T = col.T; ts.T_surface = col.T_surface;
  ts.emissivity = r->d_emis.p; ts.has_hard_surface = r->has_hard_surface ? 1 : 0; ts.ir_tau_min = r->ir_tau_min;
  ts.nzen = (int)r->zenith_u.size(); ts.zen_u = r->d_zen_u.p; ts.zen_w = r->d_zen_w.p; ts.zen_iu = r->d_zen_iu.p;
  for (int z = 0; z < ts.nzen && z < MAX_ZEN; z++) {
    ts.zen_u_v[z] = r->zenith_u[z]; ts.zen_w_v[z] = r->zenith_w[z]; ts.zen_iu_v[z] = 1.0 / r->zenith_u[z];
  }
  ts.albedo = r->d_albedo.p; ts.photons_sol = r->d_photons.p;
  ts.photon_scale_factor = r->photon_scale_factor; ts.diurnal_fac = r->diurnal_fac;
  ts.am_f1 = r->d_am_f1.p; ts.am_f2 = r->d_am_f2.p; ts.am_dw = r->d_am_dw.p;
  ts.ir_fup_a = r->wrk_ir.fup_a.p; ts.ir_fdn_a = r->wrk_ir.fdn_a.p; ts.ir_tau_band = r->wrk_ir.tau_band.p;
  ts.sol_fup_a = r->wrk_sol.fup_a.p; ts.sol_fdn_a = r->wrk_sol.fdn_a.p; ts.sol_amean = r->wrk_sol.amean.p;
  ts.sol_tau_band = r->wrk_sol.tau_band.p;
  return ts;
}

// A batch of columns worked on by ONE launch of each kernel (radtran_toa_fluxes_batch): per-column
// blocks in arenas, column c at base + c * stride.  Null for a single call on the handle's own buffers.
struct BatchCtx {
  int ncol;
  double *col, *prep, *opr, *res, *flux;
  int *done;
  BatchStrides bs;
};

// spectra block of one column of a batch: [ir fup_a | ir fdn_a | ir tau_band | sol fup_a | sol fdn_a | sol amean | sol tau_band]
size_t res_block_count(Radtran *r) {
  const size_t nl = (size_t)r->nz + 1, nz = r->nz;
  return (size_t)r->ir.nw * (2 * nl + nz) + (size_t)r->sol.nw * (3 * nl + nz);
}

// The stored w0 array is current (a fused call's tiles leave it to the two-stream part of the same grid:
// whoever else wants it -- IR-only calls on the stored opacities, the batched IR kernel, radtran_opr_get --
// gets it formed from tau and the layers' scattering optical depth first)
void ensure_w0(Radtran *r) {
  if (r->w0_valid) return;
  launch_w0_from_scat(r->d_tau.p, r->d_scat.p, r->d_w0.p, r->nw, r->ng, r->nz, r->stream);
  HIPCHK(hipGetLastError());
  r->w0_valid = true;
}

void enqueue_radiate(Radtran *r, bool compute_solar, bool compute_opacity, bool allow_fused = true,
                     const BatchCtx *bc = nullptr) {
  r->timer_calls++;
  upload_fields(r);
  const int nz = r->nz;
  const size_t nl = (size_t)nz + 1;
  ColumnDev col = bc ? column_dev_at(r, bc->col, bc->prep) : column_dev(r);
  BatchStrides bs;
  std::memset(&bs, 0, sizeof(bs));
  if (bc) bs = bc->bs;
  const int ncol = bc ? bc->ncol : 1;
  const int nsrc = bc ? nz : (r->col_override ? r->nsrc_override : r->nsrc);
  // where this call's optical properties and spectra live
  double *o_tau = r->d_tau.p, *o_w0 = r->d_w0.p, *o_g = r->d_g.p, *o_tb = r->d_tau_band.p, *o_sc = r->d_scat.p;
  double *ir_fup = r->wrk_ir.fup_a.p, *ir_fdn = r->wrk_ir.fdn_a.p, *ir_tb = r->wrk_ir.tau_band.p;
  double *sol_fup = r->wrk_sol.fup_a.p, *sol_fdn = r->wrk_sol.fdn_a.p, *sol_am = r->wrk_sol.amean.p, *sol_tb = r->wrk_sol.tau_band.p;
  if (bc) {
    o_tau = bc->opr; o_w0 = o_tau + r->d_tau.n; o_g = o_w0 + r->d_w0.n; o_tb = o_g + r->d_g.n; o_sc = o_tb + r->d_tau_band.n;
    ir_fup = bc->res; ir_fdn = ir_fup + r->ir.nw * nl; ir_tb = ir_fdn + r->ir.nw * nl;
    sol_fup = ir_tb + (size_t)r->ir.nw * nz; sol_fdn = sol_fup + r->sol.nw * nl; sol_am = sol_fdn + r->sol.nw * nl;
    sol_tb = sol_am + r->sol.nw * nl;
  }
  auto ts_params = [&]() {
    TwoStreamParams ts = make_twostream_params(r, col, compute_solar);
    ts.tau = o_tau; ts.w0 = o_w0; ts.g = o_g; ts.tau_band = o_tb; ts.scat = o_sc; ts.w0_from_scat = 0;
    ts.ir_fup_a = ir_fup; ts.ir_fdn_a = ir_fdn; ts.ir_tau_band = ir_tb;
    ts.sol_fup_a = sol_fup; ts.sol_fdn_a = sol_fdn; ts.sol_amean = sol_am; ts.sol_tau_band = sol_tb;
    // exact pairs in the column (and this call computes the opacities from it) -> exact pairs in opr
    static const bool allow_paired = [] { const char *e = getenv("CLIMA_HIP_PAIRED"); return !(e && e[0] == '0'); }();
    ts.paired = (allow_paired && !bc && !r->col_override && compute_opacity && r->all_pairs_exact) ? 1 : 0;
    return ts;
  };
  bool pre_zeroed = false, fused_done = false, whole_stores = false;
  if (compute_opacity) {
    PrepParams pp;
    std::memset(&pp, 0, sizeof(pp));
    pp.ncol = ncol; pp.bs = bs;
    pp.nz = nz; pp.nsp = r->nsp; pp.np = r->np; pp.nslots = r->nslots;
    pp.has_cont = r->has_cont; pp.LH2O = r->LH2O;
    for (int s = 0; s < r->nslots; s++) pp.slots[s] = r->slots[s];
    if (r->cust_on) {  // evaluated without clamping: the end intervals extrapolate (linear_interpolation_module.F90:348-350)
      SlotDev cs;
      cs.axis = r->d_cust_axis.p; cs.n = r->cust_nP; cs.source = -1;
      cs.lo = -std::numeric_limits<double>::infinity(); cs.hi = std::numeric_limits<double>::infinity();
      cs.flag_clamp = 0;
      pp.slots[pp.nslots++] = cs;
    }
    pp.nabs = (int)r->abs_entries.size();
    for (int e = 0; e < pp.nabs; e++) { pp.abs_kind[e] = r->abs_entries[e].kind; pp.abs_a[e] = r->abs_entries[e].a; pp.abs_b[e] = r->abs_entries[e].b; }
    pp.col = col;
    pp.call_id = ++r->call_id;
    {  // when the wave-per-column two-stream kernel will add two g-point groups into its
       // outputs, let spare blocks of this launch clear them (saves a launch)
      const bool wave_mode = !r->ts_block_mode && (nz + 63) / 64 <= 8;
      pre_zeroed = false;
      // (not when the fused grid will run its half-wave form: those blocks store whole values -- 7.7 MB of
      // zeros per config-2 call that nobody reads)
      {
        OpacityParams oq;
        std::memset(&oq, 0, sizeof(oq));
        oq.nz = nz; oq.ng = r->ng; oq.nbins = r->op_n; oq.rebin_mode = r->rebin_mode; oq.cust.on = r->cust_on ? 1 : 0;
        const bool coop = !bc && r->ng == 8 && (long)r->op_n * nsrc <= r->coop_items;
        if ((bc || (r->fused && allow_fused && !coop)) && wave_mode && twostream_w_groups(r->ng) >= 2) {
          const TwoStreamParams tq = ts_params();
          whole_stores = fused_half_form(oq, tq, ncol) != 0;
        } else if (!bc && wave_mode && r->ng == 8) {
          // one launch per kernel: the stand-alone half-wave two-stream kernel stores whole values too
          const TwoStreamParams tq = ts_params();
          whole_stores = tq.nzen <= MAX_ZEN && twostream_w_half_slots(tq) != 0;
        }
      }
      if (wave_mode && twostream_w_groups(r->ng) >= 2 && !whole_stores) {
        int n = 0;
        if (r->ir_n > 0) {
          pp.zero_ptr[n] = ir_fup + (size_t)r->ir_lo * nl; pp.zero_count[n++] = nl * r->ir_n;
          pp.zero_ptr[n] = ir_fdn + (size_t)r->ir_lo * nl; pp.zero_count[n++] = nl * r->ir_n;
        }
        if (compute_solar && r->sol_n > 0) {
          pp.zero_ptr[n] = sol_fup + (size_t)r->sol_lo * nl; pp.zero_count[n++] = nl * r->sol_n;
          pp.zero_ptr[n] = sol_fdn + (size_t)r->sol_lo * nl; pp.zero_count[n++] = nl * r->sol_n;
          pp.zero_ptr[n] = sol_am + (size_t)r->sol_lo * nl; pp.zero_count[n++] = nl * r->sol_n;
        }
        pp.nzero = n;
        pre_zeroed = true;
      }
    }
    { KernelTimer t(r, 0); launch_prep(pp, r->stream); HIPCHK(hipGetLastError()); t.stop(); }

    OpacityParams op;
    std::memset(&op, 0, sizeof(op));
    op.nz = nz; op.nw = r->nw; op.ng = r->ng; op.nsp = r->nsp; op.np = r->np;
    op.bin_lo = r->op_lo; op.nbins = r->op_n; op.nsrc = nsrc;
    op.nk = (int)r->k.size(); op.nray = (int)r->ray.size(); op.npart = (int)r->part.size();
    for (size_t i = 0; i < r->k.size(); i++)
      op.k[i] = KDev{r->k[i]->d_log10k.p, r->k[i]->sp, r->k[i]->nP, r->k[i]->nT, (int)(2 * i), (int)(2 * i + 1)};
    for (size_t i = 0; i < r->ray.size(); i++) op.ray[i] = XsDev{r->ray[i]->d_data.p, 0, r->ray[i]->sp1, -1, 0, -1};
    op.nabs = (int)r->abs_entries.size();
    for (int e = 0; e < op.nabs; e++) op.abs[e] = r->abs_entries[e];
    for (size_t i = 0; i < r->part.size(); i++)
      op.part[i] = PartDev{r->part[i]->d_w0.p, r->part[i]->d_qext.p, r->part[i]->d_gt.p, r->part[i]->p_ind, r->part[i]->nrad, r->part_slot[i]};
    op.wbin = r->d_wbin.p; op.wbin_e = r->d_wbin_e.p; op.wxy = r->d_wxy.p; op.wbin_e_pad = r->d_wbin_e_pad.p; op.rorr_tab = r->d_rorr_tab.p;
    op.col = col;
    op.cust = CustomDev{r->d_cust_dtau.p, r->d_cust_w0.p, r->d_cust_g0.p, r->cust_nP, r->nslots, r->cust_on ? 1 : 0};
    op.rebin_mode = r->rebin_mode;
    // few (bin, source layer) items -- a bin-sharded rank, a short or all-pairs column: the group-of-lanes
    // kernel (a fifth of the lane-per-item kernel's dependent chain) and one launch per kernel
    op.coop = (!bc && r->ng == 8 && (long)r->op_n * nsrc <= r->coop_items) ? 1 : 0;
    op.generic = r->generic_opacity ? 1 : 0;
#ifdef CLIMA_STAMPS
    op.stamps = r->d_stamps.p;
#endif
    op.tau = o_tau; op.w0 = o_w0; op.g = o_g; op.tau_band = o_tb; op.scat = o_sc; op.write_w0 = 1;
    if (bc || (r->fused && allow_fused && (pre_zeroed || whole_stores) && !op.coop)) {
      TwoStreamParams tsf = ts_params();
      if (fused_supported(op, tsf) && (pre_zeroed || whole_stores)) {
        FusedParams fp;
        std::memset(&fp, 0, sizeof(fp));
        fp.ncol = ncol; fp.bs = bs;
        fp.call_id = pp.call_id; fp.max_spins = r->fused_max_spins;
        fp.done = bc ? bc->done : r->d_done.p; fp.timeout_flag = r->d_err.p + 1;
        KernelTimer t(r, 1);
        // the fused grid's two-stream part forms w0 from the layers' scattering optical depth itself: the
        // tiles leave the 8 nw nz values unwritten (26 MB per config-2 call neither stored nor read back);
        // ensure_w0() materialises them if something else asks (CLIMA_HIP_W0_SCAT=0: always written)
        static const bool w0_scat = [] { const char *e = getenv("CLIMA_HIP_W0_SCAT"); return !(e && e[0] == '0'); }();
        op.write_w0 = w0_scat ? 0 : 1;
        fused_done = launch_fused(op, tsf, fp, r->stream);
        HIPCHK(hipGetLastError());
        t.stop();
        if (!fused_done) op.write_w0 = 1;
      }
    }
    if (bc && !fused_done) throw HipFail{"internal: a one-launch batch needs the fused grid"};
    if (!fused_done) {
      KernelTimer t(r, 1);
      const bool ok = launch_opacity(op, r->stream);
      HIPCHK(hipGetLastError());
      if (!ok)
        throw HipFail{"k-distributions with " + std::to_string(r->ng) + " g-points are not supported (1..32)"};
      t.stop();
    }
    r->opr_valid = true;
    if (!bc) r->w0_valid = op.write_w0 != 0;
    if (!bc && !r->col_override) r->opr_upload_id = r->upload_id;
  } else if (!bc) {
    ensure_w0(r);   // this call's two-stream kernels read the stored optical properties
  }
  r->last_cs = compute_solar;
  if (compute_solar) r->solar_id = r->call_id;

  TwoStreamParams ts = ts_params();
  if (!fused_done) {
    KernelTimer t(r, 2);
    // default: wave-per-column kernel; CLIMA_HIP_TS_MODE=block selects the workgroup-per-bin form
    bool ok = false;
    if (!r->ts_block_mode && ts.nzen <= MAX_ZEN) ok = launch_twostream_w(ts, r->stream, &r->ts_lds, pre_zeroed);
    if (!ok) { HIPCHK(hipGetLastError()); ok = launch_twostream(ts, r->stream, &r->ts_lds); }
    HIPCHK(hipGetLastError());
    if (!ok)
      throw HipFail{"nz*ngauss = " + std::to_string(nz * r->ng) + " exceeds what the two-stream kernels can stage"};
    t.stop();
  }

  IntegrateParams ip;
  std::memset(&ip, 0, sizeof(ip));
  ip.ncol = ncol; ip.bs = bs;
  ip.nz = nz; ip.nw_ir = r->ir.nw; ip.nw_sol = r->sol.nw;
  ip.ir_lo = r->ir_lo; ip.ir_n = r->ir_n; ip.sol_lo = r->sol_lo; ip.sol_n = r->sol_n;
  ip.do_solar = compute_solar ? 1 : 0;
  ip.ir_fup_a = ir_fup; ip.ir_fdn_a = ir_fdn;
  ip.sol_fup_a = sol_fup; ip.sol_fdn_a = sol_fdn;
  ip.ir_freq = r->ir.d_freq.p; ip.sol_freq = r->sol.d_freq.p;
  ip.flux_n = bc ? bc->flux : (r->flux_override ? r->flux_override : r->d_flux_n.p);
  ip.flux_part = r->shard_world > 1 ? r->d_flux_part.p : nullptr;
  ip.f_total = r->shard_world == 1 ? (r->ftot_override ? r->ftot_override : r->d_f_total.p) : nullptr;
  ip.nchunk = integrate_chunks(std::max(r->ir_n, r->sol_n));
  ip.partial = r->d_partial.p;
  const bool reduce = r->comm && !bc && !r->col_override;
  // a synchronous call on an unsharded handle: the integration kernel stores the level rows and the error words into
  // the host's pinned block as well, and the call ends with a stream synchronise -- no copy launch in front of it
  r->small_in_host = false;
  if (r->want_host_out && r->h_small_dev && !bc && !r->col_override && !reduce && r->shard_world == 1 && integrate_one_launch(ip)) {
    ip.host_out = r->h_small_dev;
    ip.err_words = r->d_err.p;
    r->small_in_host = true;
  }
  if (reduce) {
    // the status word rides on the all-reduce in the slot behind the four level rows (f_total's first element:
    // on such a handle f_total is formed on the host from the REDUCED rows, fetch_small)
    ip.f_total = nullptr;
    ip.timeout_out = r->d_small.p + 4 * nl; ip.timeout_flag = r->d_err.p + 1;
    ip.id_opr = r->call_id; ip.id_sol = r->solar_id;
  }
  { KernelTimer t(r, 3); launch_integrate(ip, r->stream); HIPCHK(hipGetLastError()); t.stop(); }
  if (reduce) {
    // the step's single collective (src/radtran/clima_radtran_radiate.f90:184-192 summed over the bins of all
    // ranks): in place, on the handle's stream, no host round trip
    NCCLCHK(ncclAllReduce(r->d_small.p, r->d_small.p, 4 * nl + 1, ncclDouble, ncclSum, r->comm, r->stream));
    r->comm_reduces++;
  }
  r->small_valid = false;
}

bool recover_fused_timeout(Radtran *r);

void fetch_small(Radtran *r) {
  if (r->small_valid) return;
  const int nl = r->nz + 1;
  for (int pass = 0; pass < 2; pass++) {
    if (!r->small_in_host)
      HIPCHK(hipMemcpyAsync(r->h_small, r->d_small.p, sizeof(double) * (5 * nl + 1), hipMemcpyDeviceToHost, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    resolve_events(r);
    if (r->comm) r->comm_status = r->h_small[4 * nl];
    if (r->col_override || !r->column_loaded || !recover_fused_timeout(r)) break;  // re-issued unfused: fetch again
  }
  // f_total from the four level rows (clima_radtran.f90:287); the one-launch integration leaves it
  // to the host, the other forms computed the same expression on the device
  double *h = r->h_small;
  for (int i = 0; i < nl; i++) h[4 * nl + i] = (h[3 * nl + i] - h[2 * nl + i]) + (h[1 * nl + i] - h[0 * nl + i]);
  r->small_valid = true;
}

bool check_dims(Radtran *r, int dim_T, int dim_P, int d1, int d2, int dim_dz, int has_p, int p1, int p2,
                int r1, int r2, const double *pdens, const double *radii, char *err) {
  // check_inputs / check_dimensions(_p), clima_radtran.f90:417-491 (same texts)
  if (has_p && ((pdens && !radii) || (radii && !pdens))) { set_err(err, "Both pdensities and radii must be arguments."); return false; }
  if (r->np > 0 && (!has_p || !radii)) { set_err(err, "The model contains particles but \"pdensities\" and \"radii\" are not arguments."); return false; }
  if (dim_T != r->nz) { set_err(err, "\"T\" has the wrong input dimension."); return false; }
  if (dim_P != r->nz) { set_err(err, "\"P\" has the wrong input dimension."); return false; }
  if (d1 != r->nz || d2 != r->nsp) { set_err(err, "\"densities\" has the wrong input dimension."); return false; }
  if (dim_dz != r->nz) { set_err(err, "\"dz\" has the wrong input dimension."); return false; }
  if (has_p && radii) {
    if (p1 != r->nz || p2 != r->np) { set_err(err, "\"pdensities\" has the wrong input dimension."); return false; }
    if (r1 != r->nz || r2 != r->np) { set_err(err, "\"radii\" has the wrong input dimension."); return false; }  // :458-461
  }
  return true;
}

void do_upload(Radtran *r, double T_surface, const double *T, const double *P, const double *dens,
               const double *dz, const double *pdens, const double *radii) {
  const int nz = r->nz;
  double *h = r->h_col;
  // the pinned staging buffer is reused: wait (lazily, here) for the previous upload's copy
  if (r->upload_pending) { HIPCHK(hipEventSynchronize(r->ev_upload)); r->upload_pending = false; }
  h[0] = T_surface;
  std::memcpy(h + 1, T, sizeof(double) * nz);
  std::memcpy(h + 1 + nz, P, sizeof(double) * nz);
  std::memcpy(h + 1 + 2 * nz, dz, sizeof(double) * nz);
  std::memcpy(h + 1 + 3 * nz, dens, sizeof(double) * (size_t)nz * r->nsp);
  double *hp = h + 1 + 3 * nz + (size_t)nz * r->nsp;
  if (r->np > 0 && pdens && radii) {
    std::memcpy(hp, pdens, sizeof(double) * (size_t)nz * r->np);
    std::memcpy(hp + (size_t)nz * r->np, radii, sizeof(double) * (size_t)nz * r->np);
  }
  r->column_has_particles = (pdens && radii);
  r->upload_id++;
  {
    int *meta = reinterpret_cast<int *>(h + r->meta_ofs);
    r->nsrc = build_meta(r, T, P, dz, dens, pdens, radii, r->column_has_particles, meta);
    bool all = (nz % 2 == 0) && r->nsrc * 2 == nz;
    for (int m = 0; m < r->nsrc && all; m++) all = (meta[1 + m] & SRC_EXACT) != 0;
    r->all_pairs_exact = all;
  }
  {
    // ~18 KB: a kernel that reads the pinned buffer over PCIe gets the column into HBM 4 us sooner
    // than the copy engine does (CLIMA_HIP_COPY_KERNEL=0 selects hipMemcpyAsync)
    static const bool kcopy = [] { const char *e = getenv("CLIMA_HIP_COPY_KERNEL"); return !(e && e[0] == '0'); }();
    if (kcopy && r->h_col_dev) launch_copy(r->d_col.p, r->h_col_dev, r->col_count, r->stream);
    else HIPCHK(hipMemcpyAsync(r->d_col.p, h, sizeof(double) * r->col_count, hipMemcpyHostToDevice, r->stream));
  }
  HIPCHK(hipEventRecord(r->ev_upload, r->stream));
  r->upload_pending = true;
  r->last_T.assign(T, T + nz);
  r->last_P.assign(P, P + nz);
  if (r->np > 0 && radii) r->last_radii.assign(radii, radii + (size_t)nz * r->np); else r->last_radii.clear();
  r->column_loaded = true;
}

// A two-stream block of the fused grid gave up waiting for its bin's opacity blocks (k_fused: the
// wait is bounded; it can expire when the device is time-sliced with another process, or if blocks
// were ever dispatched out of index order).  Nothing is wrong with the inputs: the stale parts are
// computed again through the separate launches.  h_errflag[1] must be current (fetch_small /
// radtran_synchronize).  Opacities and IR results are stale when the last opacity pass timed out,
// the solar results when the pass of the last solar computation did.
const char *const REPLACED_MSG = "The fused opacity/two-stream hand-off of an earlier opacity pass timed out and the column has "
                                 "been replaced since: repeat the steps from that opacity pass on.";
bool recover_fused_timeout(Radtran *r) {
  if (r->comm) {
    // With a communicator the partial rows are already summed when the host looks: the step is repeated by
    // EVERY rank -- each learns of any rank's expired wait from the reduced status word -- through the separate
    // launches, collective included.  The repeat's own status word is 0 (its passes have new ids).
    const double st = r->comm_status;
    if (!(st > 0.0)) return false;
    r->comm_status = 0.0;
    r->checked_timeout = r->call_id;
    if (std::fmod(st, 1024.0) > 0.0 && r->upload_id != r->opr_upload_id) throw HipFail{REPLACED_MSG};
    r->fused_fallbacks++;
    enqueue_radiate(r, st >= 1024.0 || r->last_cs, true, false);
    return true;
  }
  const int t = r->h_errflag[1];
  if (t <= r->checked_timeout) return false;
  r->checked_timeout = r->call_id;
  const bool stale_opr = t == r->call_id, stale_sol = t == r->solar_id;
  if (!stale_opr && !stale_sol) return false;
  if (r->shard_world > 1)  // reduced by the CALLER (no communicator on the handle): this step cannot be redone here
    throw HipFail{"The fused opacity/two-stream hand-off timed out on a bin-sharded handle whose all-reduce is the caller's; "
                  "repeat the step (radtran_fused_set(0) selects the separate launches; with radtran_comm_init_rank the "
                  "library repeats it itself)."};
  // the stored opacities came from a column that has been replaced since (upload A, opacity pass, upload B, a
  // compute_opacity = .false. pass): computing them again from column B is not what the caller asked for
  if (stale_opr && r->upload_id != r->opr_upload_id) throw HipFail{REPLACED_MSG};
  r->fused_fallbacks++;
  enqueue_radiate(r, stale_sol || r->last_cs, true, false);
  return true;
}

bool surface_device_error(Radtran *r, char *err) {
  const bool failed = *r->h_errflag > r->checked_id;  // a call since the last check flagged
  r->checked_id = r->call_id;
  if (failed) {
    // clima_radtran_types.f90:773-776
    set_err(err, "Opacity computation failed in one or more wavelength bins.");
    return true;
  }
  return false;
}

// Everything enqueued so far has run, and a fused hand-off that timed out has been repaired: after this the
// device buffers hold the call's results (what fetch_small does for the level rows; the per-bin spectra and the
// optical properties are read out after this)
void settle(Radtran *r) {
  for (int pass = 0; pass < 2; pass++) {
    HIPCHK(hipMemcpyAsync(r->h_errflag, r->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, r->stream));
    if (r->comm && !r->small_valid)   // the reduced status word (the rows themselves are fetched when they are read)
      HIPCHK(hipMemcpyAsync(&r->comm_status, r->d_small.p + 4 * (r->nz + 1), sizeof(double), hipMemcpyDeviceToHost, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    resolve_events(r);
    if (r->col_override || !r->column_loaded || !recover_fused_timeout(r)) break;
  }
}

void defer_err(Radtran *r, const std::string &msg) {
  if (r && r->deferred_err.empty()) r->deferred_err = msg;
}

void get2d(WrkObj *w, DevBuf<double> &buf, int dim1, int dim2, double *arr) {
  Radtran *r = w->parent;
  if (!r->small_valid) settle(r);   // (valid rows: the call was synchronised and checked when they were fetched, nothing enqueued since)
  size_t n = std::min((size_t)dim1 * dim2, buf.n);
  if (n) HIPCHK(hipMemcpy(arr, buf.p, n * sizeof(double), hipMemcpyDeviceToHost));
}

}  // namespace

#define GUARD(r, ptr, err)                                   \
  Radtran *r = as_rad(ptr);                                  \
  if (!r) { set_err(err, "invalid Radtran handle"); return; }
#define TRY try {
#define CATCH(err) } catch (const HipFail &f) { set_err(err, f.msg); } catch (const std::exception &e) { set_err(err, e.what()); }

extern "C" {

void allocate_radtran(void **ptr) { *ptr = new Radtran(); }
void deallocate_radtran(void *ptr) {
  Radtran *r = as_rad(ptr);
  if (r) delete r;
}

void radtran_create_begin(void *ptr, const int *nz, const int *nsp, const int *np, const int *nw,
                          const double *wavl, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (*nz < 1) { set_err(err, "\"nz\" can not be less than 1."); return; }  // clima_radtran.f90:149-152
  if (*nw < 1 || *nsp < 1 || *np < 0) { set_err(err, "invalid dimensions"); return; }
  if (*np > MAX_PART) { set_err(err, "too many particle species for this build"); return; }
  r->nz = *nz; r->nsp = *nsp; r->np = *np; r->nw = *nw;
  r->wavl.assign(wavl, wavl + *nw + 1);
  r->freq.resize(*nw + 1);
  for (int i = 0; i < *nw + 1; i++) r->freq[i] = C_LIGHT / (wavl[i] * 1.0e-9);  // types_create.f90:361
  r->state = 1;
}

void radtran_add_ktable(void *ptr, const int *sp_ind, const int *ngauss, const double *weights,
                        const int *npress, const double *log10P, const int *ntemp, const double *temp,
                        const double *log10k, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 1) { set_err(err, "radtran_add_ktable: call between create_begin and create_end"); return; }
  if (*sp_ind < 1 || *sp_ind > r->nsp) { set_err(err, "k-distribution species index out of range"); return; }
  if ((int)r->k.size() >= MAX_K) { set_err(err, "too many k-distributions for this build"); return; }
  if (!r->k.empty() && *ngauss != r->ng) { set_err(err, "all k-distributions must share the same g-points"); return; }
  if (*npress < 2 || *ntemp < 2) { set_err(err, "k-distribution grids need at least 2 nodes"); return; }
  auto *k = new KTabHost();
  k->sp = *sp_ind - 1; k->ng = *ngauss; k->nP = *npress; k->nT = *ntemp;
  k->weights.assign(weights, weights + *ngauss);
  k->log10P.assign(log10P, log10P + *npress);
  k->temp.assign(temp, temp + *ntemp);
  k->log10k.assign(log10k, log10k + (size_t)r->nw * *ntemp * *npress * *ngauss);
  if (r->k.empty()) {  // create_Ksettings, types_create.f90:191-224; weight_e :1303-1304
    r->ng = *ngauss;
    r->wbin = k->weights;
    r->wbin_e.assign(r->ng + 1, 0.0);
    for (int i = 1; i < r->ng + 1; i++) r->wbin_e[i] = r->wbin[i - 1] + r->wbin_e[i - 1];
    r->wxy.assign((size_t)r->ng * r->ng, 0.0);
    for (int i = 0; i < r->ng; i++)
      for (int j = 0; j < r->ng; j++) r->wxy[j + (size_t)i * r->ng] = r->wbin[i] * r->wbin[j];
    r->rebin_mode = rebin_mode_for(r->wbin, r->wbin_e, r->wxy);
  }
  r->k.push_back(k);
}

void radtran_add_xsection(void *ptr, const int *xs_type, const int *dim, const int *sp_ind1,
                          const int *sp_ind2, const int *ntemp, const double *temp, const double *data,
                          char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 1) { set_err(err, "radtran_add_xsection: call between create_begin and create_end"); return; }
  std::vector<XsHost *> *list;
  if (*xs_type == CLIMA_XS_CIA) list = &r->cia;
  else if (*xs_type == CLIMA_XS_RAYLEIGH) list = &r->ray;
  else if (*xs_type == CLIMA_XS_PHOTOLYSIS || *xs_type == CLIMA_XS_ABSORPTION) list = &r->pxs;
  else { set_err(err, "unknown cross-section type"); return; }
  if ((int)list->size() >= MAX_XS) { set_err(err, "too many cross sections for this build"); return; }
  if (*dim != 0 && *dim != 1) { set_err(err, "cross-section dim must be 0 or 1"); return; }
  if (*xs_type == CLIMA_XS_RAYLEIGH && *dim != 0) { set_err(err, "Rayleigh cross sections are 0-D"); return; }
  if (*sp_ind1 < 1 || *sp_ind1 > r->nsp) { set_err(err, "cross-section species index out of range"); return; }
  if (*xs_type == CLIMA_XS_CIA && (*sp_ind2 < 1 || *sp_ind2 > r->nsp)) { set_err(err, "CIA species index out of range"); return; }
  if (*dim == 1 && *ntemp < 2) { set_err(err, "1-D cross sections need at least 2 temperatures"); return; }
  auto *x = new XsHost();
  x->type = *xs_type; x->dim = *dim; x->sp1 = *sp_ind1 - 1; x->sp2 = (*xs_type == CLIMA_XS_CIA) ? *sp_ind2 - 1 : -1;
  x->nT = *dim ? *ntemp : 0;
  if (*dim) x->temp.assign(temp, temp + *ntemp);
  x->data.assign(data, data + (*dim ? (size_t)r->nw * *ntemp : (size_t)r->nw));
  list->push_back(x);
}

void radtran_set_water_continuum(void *ptr, const int *LH2O, const int *ntemp, const double *temp,
                                 const double *log10_xs_H2O, const double *log10_xs_foreign, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 1) { set_err(err, "radtran_set_water_continuum: call between create_begin and create_end"); return; }
  if (*LH2O < 1 || *LH2O > r->nsp || *ntemp < 2) { set_err(err, "invalid water continuum arguments"); return; }
  r->has_cont = true; r->LH2O = *LH2O - 1; r->cont_nT = *ntemp;
  r->cont_temp.assign(temp, temp + *ntemp);
  r->cont_H2O.assign(log10_xs_H2O, log10_xs_H2O + (size_t)r->nw * *ntemp);
  r->cont_foreign.assign(log10_xs_foreign, log10_xs_foreign + (size_t)r->nw * *ntemp);
}

void radtran_add_particle(void *ptr, const int *p_ind, const int *nrad, const double *radii,
                          const double *w0, const double *qext, const double *gt, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 1) { set_err(err, "radtran_add_particle: call between create_begin and create_end"); return; }
  if (*p_ind < 1 || *p_ind > r->np || *nrad < 2) { set_err(err, "invalid particle arguments"); return; }
  if ((int)r->part.size() >= MAX_PART) { set_err(err, "too many particles for this build"); return; }
  auto *p = new PartHost();
  p->p_ind = *p_ind - 1; p->nrad = *nrad;
  size_t n = (size_t)r->nw * *nrad;
  p->radii.assign(radii, radii + *nrad);
  p->w0.assign(w0, w0 + n); p->qext.assign(qext, qext + n); p->gt.assign(gt, gt + n);
  r->part.push_back(p);
}

void radtran_set_channels(void *ptr, const int *n_ir_edges, const double *ir_wavl, const int *n_sol_edges,
                          const double *sol_wavl, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 1) { set_err(err, "radtran_set_channels: call between create_begin and create_end"); return; }
  if (*n_ir_edges < 2 || *n_sol_edges < 2) { set_err(err, "channels need at least one bin"); return; }
  if (!make_channel(r, r->ir, 0, *n_ir_edges, ir_wavl, err)) return;
  if (!make_channel(r, r->sol, 1, *n_sol_edges, sol_wavl, err)) return;
}

void radtran_set_photons_sol(void *ptr, const int *n, const double *photons_sol, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->sol.nw == 0 || *n != r->sol.nw) { set_err(err, "\"photons_sol\" has the wrong size"); return; }
  r->photons_sol.assign(photons_sol, photons_sol + *n);
  r->fields_dirty = true;
}

// futils v0.1.14 interp(xg, x, y, yg, ierr=) as called at types.f90:487-497: piecewise linear,
// constant beyond both ends (linear_extrap defaults to false); nonzero when x is not ascending
static int futils_interp(const std::vector<double> &xg, int n, const double *x, const double *y, double *yg) {
  if (n < 1) return -1;
  for (int i = 1; i < n; i++) if (!(x[i] > x[i - 1])) return -2;
  for (size_t i = 0; i < xg.size(); i++) {
    const double xv = xg[i];
    if (xv <= x[0]) yg[i] = y[0];
    else if (xv >= x[n - 1]) yg[i] = y[n - 1];
    else {
      int lo = 0, hi = n - 1;
      while (hi - lo > 1) { const int mid = (lo + hi) / 2; if (xv < x[mid]) hi = mid; else lo = mid; }
      const double slope = (y[lo + 1] - y[lo]) / (x[lo + 1] - x[lo]);
      yg[i] = y[lo] + slope * (xv - x[lo]);
    }
  }
  return 0;
}

// clima/fortran/Radtran.f90:77-107 -> Radtran_set_custom_optical_properties (clima_radtran.f90:494-506)
// -> OpticalProperties_set_custom_optical_properties (clima_radtran_types.f90:432-538)
void radtran_set_custom_optical_properties(void *ptr, const int *dim_wv, const double *wv, const int *dim_P,
                                           const double *P, const int *dim1_dtau_dz, const int *dim2_dtau_dz,
                                           const double *dtau_dz, const int *dim1_w0, const int *dim2_w0,
                                           const double *w0, const int *dim1_g0, const int *dim2_g0,
                                           const double *g0, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  const int nwv = *dim_wv, nP = *dim_P;
  for (int i = 0; i < nwv; i++) if (wv[i] <= 0.0) { set_err(err, "All elements of `wv` must be larger than zero"); return; }
  for (int i = 0; i < nP; i++) if (P[i] <= 0.0) { set_err(err, "All elements of `P` must be larger than zero"); return; }
  if (nP != *dim1_dtau_dz) { set_err(err, "`P` and `dtau_dz` have incompatible shapes"); return; }
  if (nwv != *dim2_dtau_dz) { set_err(err, "`wv` and `dtau_dz` have incompatible shapes"); return; }
  if (nP != *dim1_w0) { set_err(err, "`P` and `w0` have incompatible shapes"); return; }
  if (nwv != *dim2_w0) { set_err(err, "`wv` and `w0` have incompatible shapes"); return; }
  if (nP != *dim1_g0) { set_err(err, "`P` and `g0` have incompatible shapes"); return; }
  if (nwv != *dim2_g0) { set_err(err, "`wv` and `g0` have incompatible shapes"); return; }
  TRY
  const int nw = r->nw;
  std::vector<double> wv1(nw), row(nwv), tmp(nw);
  for (int i = 0; i < nw; i++) wv1[i] = 0.5 * (r->wavl[i + 1] + r->wavl[i]);  // :479
  std::vector<double> tab[3];
  const double *src[3] = {dtau_dz, w0, g0};
  for (auto &t : tab) t.assign((size_t)nw * nP, 0.0);
  for (int i = 0; i < nP; i++) {
    const int j = nP - 1 - i;  // :483
    for (int a = 0; a < 3; a++) {
      for (int k = 0; k < nwv; k++) row[k] = src[a][i + (size_t)k * nP];
      if (futils_interp(wv1, nwv, wv, row.data(), tmp.data()) != 0) {
        set_err(err, "Interpolation error in `set_custom_optical_properties`");
        return;
      }
      for (int l = 0; l < nw; l++) tab[a][(size_t)l * nP + j] = tmp[l];
    }
  }
  std::vector<double> lp(nP);
  for (int i = 0; i < nP; i++) lp[nP - 1 - i] = std::log10(P[i]);  // :505-506
  bool ok = nP >= 2;  // linear_interp_1d%initialize: two or more strictly increasing nodes
  for (int i = 1; i < nP && ok; i++) ok = lp[i] > lp[i - 1];
  if (!ok) { set_err(err, "Interpolation initialization error in `set_custom_optical_properties`"); return; }
  HIPCHK(hipStreamSynchronize(r->stream));  // a previous call may still read the old tables
  r->cust_axis = lp; r->cust_nP = nP;
  r->d_cust_axis.upload(lp); r->d_cust_dtau.upload(tab[0]); r->d_cust_w0.upload(tab[1]); r->d_cust_g0.upload(tab[2]);
  r->cust_on = true;
  CATCH(err)
}

static std::vector<std::string> split_lines(const char *s) {
  std::vector<std::string> out;
  if (!s) return out;
  std::string cur;
  for (const char *c = s; *c; c++) {
    if (*c == '\n') { out.push_back(cur); cur.clear(); } else cur.push_back(*c);
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

// Names the loaders know and opacities2yaml prints: species / particles in index order, one per
// line (OpticalProperties%species_names, %particle_names, clima_radtran_types.f90:100-101)
void radtran_set_names(void *ptr, const char *species_names, const char *particle_names, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  auto sp = split_lines(species_names), pa = split_lines(particle_names);
  if (r->state >= 1 && ((int)sp.size() != r->nsp || (int)pa.size() != r->np)) {
    set_err(err, "radtran_set_names: the number of names does not match the number of species / particles");
    return;
  }
  r->species_names = sp; r->particle_names = pa;
}

// k-method name (Ksettings%k_method_name), water-continuum model (WaterContinuum%model) and the
// data-set name of every particle cross section in the order they were added (ParticleXsection%dat_name)
void radtran_set_opacity_labels(void *ptr, const char *k_method, const char *water_continuum_model,
                                const char *particle_data, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (k_method && k_method[0]) r->k_method_name = k_method;
  if (water_continuum_model && water_continuum_model[0]) r->continuum_model = water_continuum_model;
  r->particle_data = split_lines(particle_data);
}

// OpticalProperties_opacities2yaml, clima_radtran_types.f90:328-430 (line by line)
static std::string opacities2yaml(Radtran *r) {
  auto sp = [&](int i) { return (i >= 0 && i < (int)r->species_names.size()) ? r->species_names[i] : ("species" + std::to_string(i + 1)); };
  auto join = [](const std::vector<std::string> &v) {
    std::string o;
    for (size_t i = 0; i < v.size(); i++) { o += v[i]; if (i + 1 != v.size()) o += ", "; }
    return o;
  };
  std::string out = "  k-method: " + r->k_method_name;
  out += "\n  opacities:";
  if (!r->k.empty()) {
    std::vector<std::string> v;
    for (auto *k : r->k) v.push_back(sp(k->sp));
    out += "\n    k-distributions: [" + join(v) + "]";
  }
  if (!r->cia.empty()) {
    std::vector<std::string> v;
    for (auto *x : r->cia) v.push_back(sp(x->sp1) + "-" + sp(x->sp2));
    out += "\n    CIA: [" + join(v) + "]";
  }
  if (!r->ray.empty()) {
    std::vector<std::string> v;
    for (auto *x : r->ray) v.push_back(sp(x->sp1));
    out += "\n    rayleigh: [" + join(v) + "]";
  }
  if (!r->pxs.empty()) {
    std::vector<std::string> v;
    for (auto *x : r->pxs) v.push_back(sp(x->sp1));
    out += "\n    photolysis-xs: [" + join(v) + "]";
  }
  if (r->has_cont) out += "\n    water-continuum: " + r->continuum_model;
  if (!r->part.empty()) {
    std::vector<std::string> v;
    for (size_t i = 0; i < r->part.size(); i++) {
      const int pi = r->part[i]->p_ind;
      const std::string name = (pi >= 0 && pi < (int)r->particle_names.size()) ? r->particle_names[pi] : ("particle" + std::to_string(pi + 1));
      const std::string dat = i < r->particle_data.size() ? r->particle_data[i] : std::string("unknown");
      v.push_back("{name: " + name + ", data: " + dat + "}");
    }
    out += "\n    particle-xs: [" + join(v) + "]";
  }
  return out;
}

// clima/fortran/Radtran.f90:41-75: _1 allocates the C string and returns its length, _2 copies it
// into the caller's buffer (out_len + 1 chars) and frees it
void radtran_opacities2yaml_wrapper_1(void *ptr, int *out_len, void **out_cp) {
  Radtran *r = as_rad(ptr);
  const std::string s = r ? opacities2yaml(r) : std::string();
  char *buf = (char *)std::malloc(s.size() + 1);
  std::memcpy(buf, s.c_str(), s.size() + 1);
  *out_len = (int)s.size();
  *out_cp = buf;
}
void radtran_opacities2yaml_wrapper_2(void *ptr, void **out_cp, const int *out_len, char *out_c) {
  (void)ptr;
  if (!out_cp || !*out_cp) return;
  std::memcpy(out_c, *out_cp, (size_t)*out_len + 1);
  std::free(*out_cp);
  *out_cp = nullptr;
}

void radtran_fused_set(void *ptr, const int *enable) {
  Radtran *r = as_rad(ptr);
  if (r) r->fused = (*enable != 0);
}
void radtran_fused_get(void *ptr, int *enabled) {
  Radtran *r = as_rad(ptr);
  *enabled = (r && r->fused) ? 1 : 0;
}
void radtran_coop_items_set(void *ptr, const int *items) {
  Radtran *r = as_rad(ptr);
  if (r) r->coop_items = *items;
}
void radtran_coop_items_get(void *ptr, int *items) {
  Radtran *r = as_rad(ptr);
  *items = r ? (int)std::min<long>(r->coop_items, 2147483647L) : 0;
}
void radtran_fused_spins_set(void *ptr, const int *spins) {
  Radtran *r = as_rad(ptr);
  if (r) r->fused_max_spins = std::max(0, *spins);
}
void radtran_fused_spins_get(void *ptr, int *spins) {
  Radtran *r = as_rad(ptr);
  *spins = r ? r->fused_max_spins : 0;
}
void radtran_ir_green_set(void *ptr, const int *mode) {
  Radtran *r = as_rad(ptr);
  if (r) r->ir_green_mode = std::max(0, std::min(2, *mode));
}
void radtran_ir_green_get(void *ptr, int *mode, int *batches) {
  Radtran *r = as_rad(ptr);
  *mode = r ? r->ir_green_mode : 0;
  *batches = r ? (int)std::min<long>(r->ir_green_batches, 2147483647L) : 0;
}
void radtran_fused_fallbacks_get(void *ptr, int *count) {
  Radtran *r = as_rad(ptr);
  *count = r ? r->fused_fallbacks : 0;
}

// clima/fortran/Radtran.f90:109-118
void radtran_unset_custom_optical_properties(void *ptr) {
  Radtran *r = as_rad(ptr);
  if (r) r->cust_on = false;
}

void radtran_create_end(void *ptr, const int *num_zenith_angles, const double *surface_albedo, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 1) { set_err(err, "radtran_create_end: create_begin has not been called"); return; }
  if (r->k.empty()) { set_err(err, "There are no k-distributions, yet there must be some to compute total opacity."); return; }
  if (r->ir.nw == 0 || r->sol.nw == 0) { set_err(err, "wavelength channels are not set"); return; }
  if (*num_zenith_angles < 1) { set_err(err, "number of zenith angles must be >= 1"); return; }
  TRY
  const int nz = r->nz, nw = r->nw, ng = r->ng;
  // zenith_angles_and_weights (clima_eqns.f90:26-41) then cos(deg*pi/180) (clima_radtran.f90:165)
  std::vector<double> x, w;
  gauss_legendre(*num_zenith_angles, x, w);
  r->zenith_u.resize(*num_zenith_angles);
  r->zenith_w.resize(*num_zenith_angles);
  for (int i = 0; i < *num_zenith_angles; i++) {
    double mu = x[i] / 2.0 + 1.0 / 2.0;
    double ang = std::acos(mu) * 180.0 / PI;
    r->zenith_w[i] = w[i] / 2.0;
    r->zenith_u[i] = std::cos(ang * PI / 180.0);
  }
  r->surface_albedo.assign(r->sol.nw, *surface_albedo);   // :182-183
  r->surface_emissivity.assign(r->ir.nw, 1.0);            // :185-186
  if (r->photons_sol.empty()) r->photons_sol.assign(r->sol.nw, 0.0);

  int dev_count = 0;
  HIPCHK(hipGetDeviceCount(&dev_count));
  if (dev_count < 1) throw HipFail{"no HIP device available: the Radtran hot path has no CPU fallback"};
  HIPCHK(hipGetDevice(&r->device));
  HIPCHK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&r->ev_upload, hipEventDisableTiming));
  if (const char *f = getenv("CLIMA_HIP_FUSED")) r->fused = atoi(f) != 0;
  if (const char *f = getenv("CLIMA_HIP_GENERIC")) r->generic_opacity = atoi(f) != 0;
  if (const char *f = getenv("CLIMA_HIP_TS_MODE")) r->ts_block_mode = std::strcmp(f, "block") == 0;
  if (const char *f = getenv("CLIMA_HIP_TS_NCOLS")) r->ts_ncols_env = atoi(f);
  if (const char *f = getenv("CLIMA_HIP_BATCH_SHARED")) r->batch_shared = atoi(f) != 0;
  if (const char *f = getenv("CLIMA_HIP_IR_GREEN")) r->ir_green_mode = std::max(0, std::min(2, atoi(f)));
  if (const char *f = getenv("CLIMA_HIP_BATCH_SHARED_MIN")) r->batch_shared_min = std::max(0, atoi(f));
  if (const char *f = getenv("CLIMA_HIP_FUSED_SPINS")) r->fused_max_spins = std::max(0, atoi(f));  // test aid: 0 makes waits expire

  // ---- tables to HBM + interpolation slots
  r->slots.clear();
  auto add_slot = [&](const double *axis_dev, const std::vector<double> &axis, int source, bool flag) {
    SlotDev s;
    s.axis = axis_dev; s.n = (int)axis.size(); s.source = source;
    s.lo = *std::min_element(axis.begin(), axis.end());   // types_create.f90:1371-1375
    s.hi = *std::max_element(axis.begin(), axis.end());
    s.flag_clamp = flag ? 1 : 0;
    r->slots.push_back(s);
  };
  for (auto *k : r->k) {
    k->d_log10k.upload(k->log10k); k->d_log10P.upload(k->log10P); k->d_temp.upload(k->temp);
    add_slot(k->d_log10P.p, k->log10P, 0, false);
    add_slot(k->d_temp.p, k->temp, 1, false);
  }
  r->abs_entries.clear();
  for (auto *v : {&r->cia, &r->pxs})
    for (auto *xs : *v) {
      xs->d_data.upload(xs->data);
      AbsEntry e;
      e.data = xs->d_data.p; e.nT = xs->dim ? xs->nT : 0; e.slot = 0;
      e.kind = (v == &r->cia) ? ABS_CIA : ABS_COLUMN; e.a = xs->sp1; e.b = xs->sp2 < 0 ? 0 : xs->sp2;
      if (xs->dim) { xs->d_temp.upload(xs->temp); e.slot = (int)r->slots.size(); add_slot(xs->d_temp.p, xs->temp, 1, false); }
      r->abs_entries.push_back(e);
    }
  for (auto *xs : r->ray) xs->d_data.upload(xs->data);
  if (r->has_cont) {  // H2O self then foreign (types.f90:719-721)
    r->d_cont_temp.upload(r->cont_temp); r->d_cont_H2O.upload(r->cont_H2O); r->d_cont_foreign.upload(r->cont_foreign);
    const int slot = (int)r->slots.size();
    add_slot(r->d_cont_temp.p, r->cont_temp, 1, false);
    r->abs_entries.push_back(AbsEntry{r->d_cont_H2O.p, r->cont_nT, slot, ABS_H2O_SELF, r->LH2O, 0});
    r->abs_entries.push_back(AbsEntry{r->d_cont_foreign.p, r->cont_nT, slot, ABS_H2O_FOREIGN, r->LH2O, 0});
  }
  if ((int)r->abs_entries.size() > MAX_ABS - (ABS_BATCH - 2)) throw HipFail{"too many continuum terms for this build"};
  // the opacity tile takes the terms in batches of ABS_BATCH without per-term tests: the list is padded with
  // terms of weight 0 on a table of zeros (they add +0.0 at the end of the sum)
  if (r->abs_entries.size() % ABS_BATCH) {
    r->d_zero_xs.upload(std::vector<double>((size_t)r->nw + 1, 0.0));
    while (r->abs_entries.size() % ABS_BATCH) r->abs_entries.push_back(AbsEntry{r->d_zero_xs.p, 0, 0, ABS_ZERO, 0, 0});
  }
  r->part_slot.clear();
  for (auto *p : r->part) {
    p->d_radii.upload(p->radii); p->d_w0.upload(p->w0); p->d_qext.upload(p->qext); p->d_gt.upload(p->gt);
    r->part_slot.push_back((int)r->slots.size());
    add_slot(p->d_radii.p, p->radii, 2 + p->p_ind, true);
  }
  r->nslots = (int)r->slots.size();
  if (r->nslots + 1 > MAX_SLOTS) throw HipFail{"too many interpolated tables for this build"};  // one kept for custom opacity
  r->d_wbin.upload(r->wbin); r->d_wbin_e.upload(r->wbin_e); r->d_wxy.upload(r->wxy);
  {
    std::vector<double> pad = r->wbin_e;
    for (int i = 0; i < 4; i++) pad.push_back(INFINITY);
    r->d_wbin_e_pad.upload(pad);
    // the tables the assembly form of the mixing step keeps in scalar registers (8 g-points): E_1..E_8, w, 1/width --
    // the reciprocal formed as the opacity tile forms it (1.0 / (E_(k+1) - E_k))
    std::vector<double> tab(24, 0.0);
    if (r->ng == 8)
      for (int k = 0; k < 8; k++) {
        tab[k] = r->wbin_e[k + 1];
        tab[8 + k] = r->wbin[k];
        tab[16 + k] = 1.0 / (r->wbin_e[k + 1] - r->wbin_e[k]);
      }
    r->d_rorr_tab.upload(tab);
  }
  r->d_freq.upload(r->freq);
  r->ir.d_freq.upload(r->ir.freq); r->sol.d_freq.upload(r->sol.freq);
  // amean unit factors per solar bin (radiate.f90:174-178)
  std::vector<double> f1(r->sol.nw), f2(r->sol.nw), dw(r->sol.nw);
  for (int l = 0; l < r->sol.nw; l++) {
    double avg_freq = 0.5 * (r->sol.freq[l] + r->sol.freq[l + 1]);
    double avg_wavl = 1.0e9 * C_LIGHT / avg_freq;
    f1[l] = (avg_freq / avg_wavl);
    f2[l] = (avg_wavl / (PLANK * C_LIGHT * 1.0e16));
    dw[l] = (r->sol.wavl[l + 1] - r->sol.wavl[l]);
  }
  r->d_am_f1.upload(f1); r->d_am_f2.upload(f2); r->d_am_dw.upload(dw);

  // ---- column, prep, opr, results
  r->meta_ofs = 1 + (size_t)3 * nz + (size_t)nz * r->nsp + (size_t)2 * nz * r->np;
  r->col_count = r->meta_ofs + (size_t)nz + 1;   // + (2 nz + 1) ints
  r->d_col.alloc(r->col_count);
  r->d_col.zero();
  HIPCHK(hipHostMalloc((void **)&r->h_col, sizeof(double) * r->col_count, hipHostMallocMapped));
  std::memset(r->h_col, 0, sizeof(double) * r->col_count);
  if (hipHostGetDevicePointer((void **)&r->h_col_dev, r->h_col, 0) != hipSuccess) { r->h_col_dev = nullptr; (void)hipGetLastError(); }
  r->prep_count = prep_block_count(r);
  r->d_prep.alloc(r->prep_count); r->d_prep.zero();
  r->d_done.alloc(((size_t)nw * nz + 255) / 256 + 1); r->d_done.zero();
  if (const char *f = getenv("CLIMA_HIP_BATCH_COLS")) r->batch_cols_in_flight = std::max(1, atoi(f));
  if (const char *f = getenv("CLIMA_HIP_COOP_ITEMS")) r->coop_items = atol(f);
#ifdef CLIMA_STAMPS
  r->d_stamps.alloc(64 + 2 * 8192); r->d_stamps.zero();
#endif
  r->opr_count = (size_t)2 * nw * ng * nz + (size_t)3 * nw * nz;
  r->d_opr.alloc(r->opr_count); r->d_opr.zero();
  r->d_tau.view(r->d_opr.p, (size_t)nw * ng * nz); r->d_w0.view(r->d_tau.p + r->d_tau.n, (size_t)nw * ng * nz);
  r->d_g.view(r->d_w0.p + r->d_w0.n, (size_t)nw * nz); r->d_tau_band.view(r->d_g.p + r->d_g.n, (size_t)nw * nz);
  r->d_scat.view(r->d_tau_band.p + r->d_tau_band.n, (size_t)nw * nz);
  auto mk = [&](WrkObj &wk, int which, int nwc) {  // clima_radtran.f90:199-214
    wk.parent = r; wk.which = which;
    wk.fup_a.alloc((size_t)(nz + 1) * nwc); wk.fdn_a.alloc((size_t)(nz + 1) * nwc);
    wk.amean.alloc((size_t)(nz + 1) * nwc); wk.tau_band.alloc((size_t)nz * nwc);
    wk.fup_a.zero(); wk.fdn_a.zero(); wk.amean.zero(); wk.tau_band.zero();
  };
  mk(r->wrk_ir, 0, r->ir.nw);
  mk(r->wrk_sol, 1, r->sol.nw);
  r->d_small.alloc((size_t)5 * (nz + 1) + 1); r->d_small.zero();
  r->d_flux_n.view(r->d_small.p, (size_t)4 * (nz + 1));
  // two ints share the last double slot: [0] particle-radius clamp, [1] fused hand-off timeout
  r->d_err.view(reinterpret_cast<int *>(r->d_small.p + (size_t)5 * (nz + 1)), 2);
  r->d_partial.alloc((size_t)4 * integrate_chunks(std::max(r->ir.nw, r->sol.nw)) * (nz + 1)); r->d_partial.zero();
  r->d_f_total.view(r->d_small.p + (size_t)4 * (nz + 1), nz + 1);
  HIPCHK(hipHostMalloc((void **)&r->h_small, sizeof(double) * (5 * (nz + 1) + 1), hipHostMallocMapped));
  std::memset(r->h_small, 0, sizeof(double) * (5 * (nz + 1) + 1));
  {
    static const bool direct = [] { const char *e = getenv("CLIMA_HIP_HOST_OUT"); return !(e && e[0] == '0'); }();
    if (!direct || hipHostGetDevicePointer((void **)&r->h_small_dev, r->h_small, 0) != hipSuccess) { r->h_small_dev = nullptr; (void)hipGetLastError(); }
  }
  r->h_errflag = reinterpret_cast<int *>(r->h_small + 5 * (nz + 1));
  r->f_total.assign(nz + 1, 0.0);
  HIPCHK(hipDeviceSynchronize());
  r->fields_dirty = true;
  compute_shard(r);
  r->small_valid = true;
  r->state = 2;
  CATCH(err)
}

void radtran_upload_column(void *ptr, const double *T_surface, const double *T, const double *P,
                           const double *densities, const double *dz, const double *pdensities,
                           const double *radii, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  if (r->np > 0 && (!pdensities || !radii)) { set_err(err, "The model contains particles but \"pdensities\" and \"radii\" are not arguments."); return; }
  TRY
  do_upload(r, *T_surface, T, P, densities, dz, pdensities, radii);
  CATCH(err)
}

void radtran_radiate_resident(void *ptr, const int *compute_solar, const int *compute_opacity, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  if (!r->column_loaded) { set_err(err, "no column has been uploaded"); return; }
  TRY
  enqueue_radiate(r, *compute_solar != 0, *compute_opacity != 0);
  CATCH(err)
}

// Batched form of the RCE Jacobian's radiative calls (src/adiabat/clima_adiabat_solve.f90:798-812:
// one `radiate(..., compute_solar=.false., compute_opacity=.false.)` per perturbed temperature
// profile).  Column c gets what that call would leave in wrk_ir%fup_n, wrk_ir%fdn_n and f_total
// (clima_radtran.f90:262-289: IR with the resident opr, solar terms of the last solar call).
//
// The general form: n columns at d_T [n][nz], d_Ts [n] (device) -> three arrays [n][nz+1] at d_out, out_arr elements apart.
static void ir_batch_general(Radtran *r, const double *d_T, const double *d_Ts, int n, double *d_out, size_t out_arr) {
  const int nz = r->nz, nl = nz + 1, nw_ir = r->ir.nw;
  // columns per launch: bounds the per-column spectra held in HBM (2.5 GB at 512 layers: 288 GB make that a
  // non-issue) and sets how many columns share one evaluation of the temperature-independent part -- a block of
  // the shared-matrix kernel takes a quarter of them: at 64 per launch that part was a fifth of the kernel
  const int CH = std::min(n, 512);
  const int nchunk = integrate_chunks(r->ir_n);
  const size_t spec = (size_t)nw_ir * nl;
  auto ensure = [](DevBuf<double> &b, size_t count) { if (b.n < count) b.alloc(count); };  // grow-only
  ensure(r->d_bup, spec * CH); ensure(r->d_bdn, spec * CH); ensure(r->d_bpartial, (size_t)CH * 2 * nchunk * nl);
  ColumnDev col = column_dev(r);
  TwoStreamParams ts = make_twostream_params(r, col, false);
  ts.ir_fup_a = r->d_bup.p; ts.ir_fdn_a = r->d_bdn.p;
  ts.b_T = nz; ts.b_Ts = 1; ts.b_out = spec;
  const bool split = twostream_w_groups(r->ng) > 1;
  for (int c0 = 0; c0 < n; c0 += CH) {
    const int nc = std::min(CH, n - c0);
    TwoStreamParams tb = ts;
    tb.T = d_T + (size_t)c0 * nz; tb.T_surface = d_Ts + c0; tb.b_ncol = nc;
    // shared-matrix batch kernel (up to 512 layers); otherwise one full solve per column
    // (one or two columns of more than 256 layers -- the response form's base profile and a dense column -- take the
    // single call's kernel: the 5-8 slot forms of the shared-matrix kernel run one wave per SIMD and amortise their
    // temperature-independent part over the columns of a block; measured at 402 layers, one column: 26 us less.  At
    // 202 layers the shared-matrix kernel is the faster one even alone: 20 us.)
    const bool tiny = nc <= r->batch_shared_min && nz > 256;
    const bool shared_ok = r->batch_shared && !tiny && launch_twostream_ir_batch(tb, nc, r->stream);
    HIPCHK(hipGetLastError());
    if (!shared_ok) {
      if (split) {  // g-point groups add into zeroed spectra
        HIPCHK(hipMemsetAsync(r->d_bup.p, 0, sizeof(double) * spec * nc, r->stream));
        HIPCHK(hipMemsetAsync(r->d_bdn.p, 0, sizeof(double) * spec * nc, r->stream));
      }
      const bool ok = launch_twostream_w(tb, r->stream, &r->ts_lds, true);
      HIPCHK(hipGetLastError());
      if (!ok)
        throw HipFail{"radiate_ir_batch: nz = " + std::to_string(nz) + " exceeds what the wave two-stream kernel holds (512)"};
    }
    BatchIntegrateParams bp;
    std::memset(&bp, 0, sizeof(bp));
    bp.nz = nz; bp.ir_lo = r->ir_lo; bp.ir_n = r->ir_n; bp.nchunk = nchunk; bp.col0 = c0;
    bp.fup_a = r->d_bup.p; bp.fdn_a = r->d_bdn.p; bp.spec_stride = spec;
    bp.freq = r->ir.d_freq.p; bp.partial = r->d_bpartial.p; bp.flux_n = r->d_flux_n.p; bp.out = d_out; bp.out_arr = out_arr;
    launch_integrate_batch(bp, nc, r->stream);
    HIPCHK(hipGetLastError());
  }
}

// The response form (ir_green.inc).  The Jacobian's columns are one base profile with one (on the doubled radiative
// grid: a few) temperatures changed each; with the opacities fixed the IR solve is linear in the Planck values, so
// such a column is F(base) + unit responses x Planck differences.  The plan below finds the base (per level the
// temperature two of three sample columns share) and every column's deviations from it; a column with more than
// green_max_dev(nz) of them (the surface-temperature column of a convective profile moves every layer) goes through the
// general kernel together with the base profile itself.
// (8 in round 3.  With the accumulation on the matrix cores a deviation costs ~0.5 us at 402 layers where a column of the
// general kernel costs 17: the bound grows with the height of the grid, nz / 16 between 8 and 32.)
static int green_max_dev(int nz) { return std::min(32, std::max(8, nz / 16)); }
struct GreenPlan {
  std::vector<double> base;              // [nz] ground-first, then the surface temperature
  std::vector<int> dense;                // columns for the general kernel
  std::vector<int> col_src, col_ptr, col_dev, dev_k;
  std::vector<double> dev_T;
  int n_sparse = 0;
};
static void green_plan(const double *T, const double *Ts, int n, int nz, GreenPlan &pl) {
  // the base: per level what two of three sample columns agree on (a Jacobian's batch changes a level in one column
  // only, so any two columns agree nearly everywhere); a poor guess costs speed, never correctness -- columns far from
  // it go through the general kernel
  pl.base.assign(nz + 1, 0.0);
  {
    const int c0 = 0, c1 = n / 2, c2 = n - 1;
    for (int j = 0; j <= nz; j++) {
      const double a = j < nz ? T[(size_t)c0 * nz + j] : Ts[c0], b = j < nz ? T[(size_t)c1 * nz + j] : Ts[c1],
                   c = j < nz ? T[(size_t)c2 * nz + j] : Ts[c2];
      pl.base[j] = (b == c) ? b : a;
    }
  }
  pl.col_src.assign(n, -1);
  const int max_dev = green_max_dev(nz);
  std::vector<int> dk, dc; std::vector<double> dT;
  const double *base = pl.base.data();
  for (int c = 0; c < n; c++) {
    const double *Tc = T + (size_t)c * nz;
    int cnt = Ts[c] != base[nz] ? 1 : 0;
    for (int j = 0; j < nz; j++) cnt += Tc[j] != base[j] ? 1 : 0;      // (no early exit: this loop vectorises)
    if (cnt > max_dev) { pl.col_src[c] = 1 + (int)pl.dense.size(); pl.dense.push_back(c); continue; }
    pl.n_sparse++;
    if (cnt == 0) continue;
    for (int j = 0; j <= nz; j++) {
      const double v = j < nz ? Tc[j] : Ts[c];
      if (v != base[j]) { dk.push_back(j < nz ? nz - 1 - j : nz); dc.push_back(c); dT.push_back(v); }   // (radiate.f90:65-69: level k is layer nz-1-k)
    }
  }
  std::vector<int> order(dk.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dk[a] < dk[b]; });
  pl.dev_k.resize(order.size()); pl.dev_T.resize(order.size());
  std::vector<int> cnt(n + 1, 0);
  for (size_t s2 = 0; s2 < order.size(); s2++) { pl.dev_k[s2] = dk[order[s2]]; pl.dev_T[s2] = dT[order[s2]]; cnt[dc[order[s2]] + 1]++; }
  pl.col_ptr.assign(n + 1, 0);
  for (int c = 0; c < n; c++) pl.col_ptr[c + 1] = pl.col_ptr[c] + cnt[c + 1];
  pl.col_dev.resize(order.size());
  std::vector<int> fill(pl.col_ptr.begin(), pl.col_ptr.end() - 1);
  for (size_t s2 = 0; s2 < order.size(); s2++) pl.col_dev[fill[dc[order[s2]]]++] = (int)s2;
}

// The part of the response form that sees the opacities alone (k_green_factor, k_green_unit, k_green_local: ~345 of the
// ~700 us of GPU work of a 402-layer Jacobian batch): its arrays and launches.  radtran_radiate_ir_batch issues it BEFORE
// the host looks at the columns when the handle's last batch of this size took the response form -- the plan (~100 us of
// host time at 403 columns) then runs beside it instead of in front of it.
// (Measured and not kept: a second queue for the kernels that do not depend on each other -- k_green_local beside
// k_green_unit, the base profile's general kernel beside k_green_factor, the mixed blocks' sums beside the far-form
// ones.  rocprofv3's kernel trace shows them side by side and each that much slower: the far-form kernel holds 2 x 232
// registers per SIMD, the general kernel one wave of 512, and the batch took 828-840 us either way.)
static void green_factor_part(Radtran *r, GreenParams &g) {
  const int nz = r->nz, nl = nz + 1, N = 2 * nz, ng = r->ng, n_ir = r->ir_n;
  const int NQ = n_ir * ng;
  const size_t RQ = (size_t)N * NQ, LQ = (size_t)nl * NQ;
  const size_t FQ = (size_t)2 * ((nl + 15) / 16) * 34 * NQ;    // (GREEN_LB, GREEN_FS of ir_green.inc)
  const size_t total = 7 * RQ + 6 * LQ + FQ + 8 * LQ + 4 * (size_t)NQ;
  if (r->d_green.n < total) r->d_green.alloc(total);
  std::memset(&g, 0, sizeof(g));
  g.nz = nz; g.ng = ng; g.n_ir = n_ir; g.ir_lo = r->ir_lo; g.ir_start = r->ir.ind_start; g.NQ = NQ;
  g.tau = r->d_tau.p; g.w0 = r->d_w0.p; g.g = r->d_g.p; g.wbin = r->d_wbin.p;
  g.freq = r->d_freq.p; g.ir_freq = r->ir.d_freq.p; g.emissivity = r->d_emis.p;
  g.has_hard_surface = r->has_hard_surface ? 1 : 0; g.ir_tau_min = r->ir_tau_min;
  double *w = r->d_green.p;
  auto take = [&](size_t cnt) { double *p0 = w; w += cnt; return p0; };
  g.RW = take(7 * RQ);
  g.IS = take(6 * LQ); g.FS = take(FQ); g.DS = take(8 * LQ); g.D0 = take(4 * (size_t)NQ);
  launch_green_factor(g, r->stream);
  HIPCHK(hipGetLastError());
}

// `pre`: the opacity-only part has been issued already (green_factor_part's parameter block)
static void ir_batch_green(Radtran *r, const GreenPlan &pl, const double *T, const double *Ts, int n, double *d_out, const GreenParams *pre) {
  const int nz = r->nz, nl = nz + 1, n_ir = r->ir_n;
  auto ensure = [](DevBuf<double> &b, size_t count) { if (b.n < count) b.alloc(count); };
  // 1. the base profile and the dense columns through the general kernel
  const int ngen = 1 + (int)pl.dense.size();
  const int ndev = (int)pl.dev_k.size(), ndev_pad = std::max(16, (ndev + 15) / 16 * 16);
  std::vector<int> mdev, mblk;           // the level blocks around a deviation's own levels (green_block_class == 2)
  for (int d = 0; d < ndev; d++)
    for (int blk = std::max(0, (pl.dev_k[d] - 1) / 16 - 1); blk < (nl + 15) / 16; blk++) {
      const int cls = green_block_class(pl.dev_k[d], blk, nz);
      if (cls == 2) { mdev.push_back(d); mblk.push_back(blk); }
      if (cls == 1) break;
    }
  const int nmix = (int)mdev.size();
  // everything the host hands over goes through ONE pinned block (no pageable copies, no synchronise before the
  // kernels): doubles T [ngen][nz] | Ts [ngen] | dev_T [ndev_pad] | base_T by level k [nl], then ints
  // dev_k [ndev_pad] | col_src [n] | col_ptr [n+1] | col_dev [ndev] | mix_dev [nmix] | mix_blk [nmix]
  const size_t nd = (size_t)ngen * nz + ngen + ndev_pad + nl, ni = (size_t)ndev_pad + n + (n + 1) + ndev + 2 * (size_t)nmix;
  const size_t stage_bytes = sizeof(double) * nd + sizeof(int) * ni;
  if (r->h_green_n < stage_bytes) {
    if (r->h_green) (void)hipHostFree(r->h_green);
    r->h_green = nullptr; r->h_green_n = 0;
    HIPCHK(hipHostMalloc((void **)&r->h_green, stage_bytes + stage_bytes / 2, hipHostMallocDefault));
    r->h_green_n = stage_bytes + stage_bytes / 2;
  }
  double *hd = reinterpret_cast<double *>(r->h_green);
  int *hi = reinterpret_cast<int *>(hd + nd);
  {
    double *hT = hd, *hTs = hd + (size_t)ngen * nz, *hdev = hTs + ngen, *hbase = hdev + ndev_pad;
    std::memcpy(hT, pl.base.data(), sizeof(double) * nz);
    hTs[0] = pl.base[nz];
    for (int i = 1; i < ngen; i++) {
      std::memcpy(hT + (size_t)i * nz, T + (size_t)pl.dense[i - 1] * nz, sizeof(double) * nz);
      hTs[i] = Ts[pl.dense[i - 1]];
    }
    for (int d = 0; d < ndev_pad; d++) hdev[d] = d < ndev ? pl.dev_T[d] : 0.0;
    for (int k = 0; k < nz; k++) hbase[k] = pl.base[nz - 1 - k];
    hbase[nz] = pl.base[nz];
    int *w2 = hi;
    for (int d = 0; d < ndev_pad; d++) *w2++ = d < ndev ? pl.dev_k[d] : 0;
    w2 = std::copy(pl.col_src.begin(), pl.col_src.end(), w2);
    w2 = std::copy(pl.col_ptr.begin(), pl.col_ptr.end(), w2);
    w2 = std::copy(pl.col_dev.begin(), pl.col_dev.end(), w2);
    w2 = std::copy(mdev.begin(), mdev.end(), w2);
    w2 = std::copy(mblk.begin(), mblk.end(), w2);
  }
  ensure(r->d_green_in, nd); ensure(r->d_gen_out, (size_t)ngen * 3 * nl);
  if (r->d_green_idx.n < ni) r->d_green_idx.alloc(ni);
  HIPCHK(hipMemcpyAsync(r->d_green_in.p, hd, sizeof(double) * nd, hipMemcpyHostToDevice, r->stream));
  HIPCHK(hipMemcpyAsync(r->d_green_idx.p, hi, sizeof(int) * ni, hipMemcpyHostToDevice, r->stream));
  const double *d_T = r->d_green_in.p, *d_Ts = d_T + (size_t)ngen * nz;
  ir_batch_general(r, d_T, d_Ts, ngen, r->d_gen_out.p, (size_t)ngen * nl);
  // 3. the opacity-only part (unless it is under way already) and the accumulation's arrays
  GreenParams g;
  if (pre) g = *pre;
  else green_factor_part(r, g);
  const int waves = green_far_waves(ndev, nl);    // of the far-form accumulation, per bin split
  // bin splits: the accumulation's waves should fill the machine ONCE (green_far_resident_waves): a few waves more
  // than that and the kernel takes two rounds
  const int qsplit = green_far_splits(n_ir, waves);
  const int msplit = std::max(qsplit, std::min(n_ir, 4096 / std::max((nmix + 3) / 4, 1)));   // the mixed blocks: few pairs, finer splits
  ensure(r->d_green_acc, (size_t)n_ir * ndev_pad + 64 + (size_t)(qsplit + msplit) * ndev_pad * 2 * nl);
  double *w = r->d_green_acc.p;
  auto take = [&](size_t cnt) { double *p0 = w; w += cnt; return p0; };
  g.DB = take((size_t)n_ir * ndev_pad + 64); g.partial = take((size_t)qsplit * ndev_pad * 2 * nl);
  g.msplit = msplit; g.partial_m = take((size_t)msplit * ndev_pad * 2 * nl);
  g.ndev = ndev; g.ndev_pad = ndev_pad; g.qsplit = qsplit;
  g.dev_k = r->d_green_idx.p; g.col_src = g.dev_k + ndev_pad; g.col_ptr = g.col_src + n; g.col_dev = g.col_ptr + n + 1;
  g.nmix = nmix; g.mix_dev = g.col_dev + ndev; g.mix_blk = g.mix_dev + nmix;
  g.dev_T = d_Ts + ngen; g.base_T = g.dev_T + ndev_pad;
  g.gen_out = r->d_gen_out.p; g.gen_arr = (size_t)ngen * nl; g.flux_n = r->d_flux_n.p; g.out = d_out; g.out_arr = (size_t)n * nl;
  launch_green_columns(g, n, r->stream);
  HIPCHK(hipGetLastError());
  r->ir_green_batches++;
}

static void register_host(Radtran *r, void *p, size_t bytes);
static bool host_is_registered(const Radtran *r, const void *p, size_t bytes);
void radtran_radiate_ir_batch(void *ptr, const int *ncol, const double *T_surface, const int *dim1_T,
                              const int *dim2_T, const double *T, double *fup_n, double *fdn_n,
                              double *f_total, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  if (*dim1_T != r->nz || *dim2_T != *ncol || *ncol < 1) { set_err(err, "\"T\" has the wrong input dimension."); return; }
  if (!r->opr_valid) { set_err(err, "radiate_ir_batch needs opacities: call radiate with compute_opacity first"); return; }
  if (r->shard_world != 1 && !r->comm) { set_err(err, "radiate_ir_batch is not available on a bin-sharded handle"); return; }
  TRY
  // CLIMA_HIP_BATCH_TIMES=1: the call's host-side phases on stderr (a diagnostic: tools/gpu_ir_batch.py)
  static const bool times = [] { const char *e = getenv("CLIMA_HIP_BATCH_TIMES"); return e && e[0] == '1'; }();
  const auto t_begin = std::chrono::steady_clock::now();
  auto since = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
  settle(r);          // (a communicator handle: the level rows of the last step are the reduced ones from here on)
  upload_fields(r);
  ensure_w0(r);
  const int nz = r->nz, nl = nz + 1, n = *ncol;
  auto ensure = [](DevBuf<double> &b, size_t count) { if (b.n < count) b.alloc(count); };  // grow-only
  ensure(r->d_bout, (size_t)n * 3 * nl);
  bool green = false;
  if (r->ir_green_mode != 0 && r->batch_shared && nz >= 4 && nz <= 512 && r->ir_n > 0) {   // (CLIMA_HIP_BATCH_SHARED=0: one full solve per column, bit for bit the single call)
    // k_green_unit / k_green_local take one (bin, g-point) pair per blockIdx.y: a grid dimension of at most 65535
    const bool fits = (long)r->ir_n * r->ng <= 65535;
    GreenParams gpre;
    bool pre = false;
    if (fits && r->green_last_n == n) { green_factor_part(r, gpre); pre = true; }
    GreenPlan pl;
    green_plan(T, T_surface, n, nz, pl);
    // worth it from a few dozen sparse columns of a tall grid on (measured: 203 columns x 202 layers 0.60 against 1.29 ms,
    // 103 x 102: 0.54 against 0.46 -- the general kernel's 1-2 slot forms are cheap and the opacity-only pass is not free)
    if (r->ir_green_mode == 2) green = pl.n_sparse > 0;
    // (ir_tau_min far below the reference's 1e-6: layers of tau ~ 1e-8 keep the source slope dB / tau, and a single changed
    // level is the worst case for it -- the response form stays correct but loses digits faster than the general kernel
    // there, 1e-8 against 5e-10 of the row's maximum in the fuzz sweep: left to mode 2)
    else if (pl.n_sparse >= 48 && r->ir_tau_min >= 1.0e-7) {
      // cost model (ms on an MI355X, from profiles/r03_ir_batch.txt and the rocprofv3 runs behind DESIGN section 4; both
      // sides scale with the number of (bin, g-point) pairs): the general kernel per column against the opacity-only
      // pass + the accumulation over deviations x levels + the extra launches
      const double scale = (double)r->ir_n * r->ng / 4800.0;
      const double t_general = pl.n_sparse * (nz <= 256 ? 0.032e-3 : 0.044e-3) * nz * scale;
      const double t_green = 0.05 + scale * (0.35 * nz / 402.0 + 0.28 * (double)pl.dev_k.size() * nl / 162409.0);   // (round 4: the accumulation on the matrix cores)
      green = t_green < 0.8 * t_general;
    }
    // not with a base that is not a number anywhere (every column would "deviate" there and inherit it), nor when the
    // work arrays (~150 KB per (bin, g-point) at 500 layers) would take more than 16 GB
    for (double v : pl.base) if (!std::isfinite(v)) green = false;
    // (the partial sums of the two accumulation kernels are part of that: their bin splits x deviations x 2 x levels, with
    // ir_batch_green's split counts -- the mixed blocks' from a lower bound of their number, one per deviation)
    {
      const int ndev = (int)pl.dev_k.size();
      const int qs = green_far_splits(r->ir_n, green_far_waves(ndev, nl));
      const int ms = std::max(qs, std::min(r->ir_n, 4096 / std::max((ndev + 3) / 4, 1)));
      if (((double)r->ir_n * r->ng * (13.0 * nl + 14.0 * nz + 4.3 * nl) +
           (double)(qs + ms) * ((double)ndev + 64.0) * 2.0 * nl) * 8.0 > 16.0e9) green = false;
    }
    if (!fits) green = false;
    if (green) ir_batch_green(r, pl, T, T_surface, n, r->d_bout.p, pre ? &gpre : nullptr);
    r->green_last_n = green ? n : -1;
  }
  if (!green) {
    ensure(r->d_bT, (size_t)n * nz); ensure(r->d_bTs, n);
    HIPCHK(hipMemcpyAsync(r->d_bT.p, T, sizeof(double) * (size_t)n * nz, hipMemcpyHostToDevice, r->stream));
    HIPCHK(hipMemcpyAsync(r->d_bTs.p, T_surface, sizeof(double) * n, hipMemcpyHostToDevice, r->stream));
    ir_batch_general(r, r->d_bT.p, r->d_bTs.p, n, r->d_bout.p, (size_t)n * nl);
  }
  const size_t arr = (size_t)n * nl;
  if (r->comm) {
    // a communicator handle worked on its share of the bins: one all-reduce of the batch's up / down arrays
    // (src/radtran/clima_radtran_radiate.f90:184-192 summed over the bins of all ranks), f_total from the reduced rows
    NCCLCHK(ncclAllReduce(r->d_bout.p, r->d_bout.p, 2 * arr, ncclDouble, ncclSum, r->comm, r->stream));
    r->comm_reduces++;
    launch_batch_ftotal(r->d_bout.p, arr, n, nz, r->d_flux_n.p, r->stream);
    HIPCHK(hipGetLastError());
  }
  // the three result arrays come back through the handle's pinned block
  if (r->h_bout_n < 3 * arr) {
    if (r->h_bout) (void)hipHostFree(r->h_bout);
    r->h_bout = nullptr; r->h_bout_n = 0;
    HIPCHK(hipHostMalloc((void **)&r->h_bout, sizeof(double) * 3 * arr, hipHostMallocDefault));
    r->h_bout_n = 3 * arr;
  }
  const double t_enq = times ? since() : 0.0;
  if (times) HIPCHK(hipStreamSynchronize(r->stream));
  const double t_kern = times ? since() : 0.0;
  // The three result arrays (3.9 MB at 403 columns x 403 levels).  Default: through the handle's pinned block in pieces,
  // the host copying piece i into the caller's arrays while piece i + 1 is still on the link (one copy and then one
  // memcpy of the whole took 95 + 140 us of a 0.93 ms call).  radtran_batch_pin_results_set(1): the caller's arrays are
  // page-locked when it passes the same three as in its previous batch (the Jacobian's work arrays: the RCE solver fills
  // one set per iteration, src/adiabat/clima_adiabat_solve.f90:768-822) and the device fills them directly -- opt-in,
  // because the arrays must then stay allocated until radtran_spectra_release or the handle's end.
  double *outs[3] = {fup_n, fdn_n, f_total};
  bool direct = false;
  if (r->batch_pin_results) {
    const bool same = r->batch_out_n == arr && r->batch_out[0] == fup_n && r->batch_out[1] == fdn_n && r->batch_out[2] == f_total;
    r->batch_out[0] = fup_n; r->batch_out[1] = fdn_n; r->batch_out[2] = f_total; r->batch_out_n = arr;
    direct = same;
    if (same)
      for (double *o : outs) { register_host(r, o, sizeof(double) * arr); direct = direct && host_is_registered(r, o, sizeof(double) * arr); }
  }
  double t_d2h = 0.0;
  if (direct) {
    // (one queue: the three arrays over three copy queues behind an event took 86 instead of 95 us at 403 x 403 and
    // 50-150 us MORE at 201 x 201 -- the extra queues' start-up)
    for (int i = 0; i < 3; i++)
      HIPCHK(hipMemcpyAsync(outs[i], r->d_bout.p + (size_t)i * arr, sizeof(double) * arr, hipMemcpyDeviceToHost, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    t_d2h = times ? since() : 0.0;
  } else {
    constexpr int NPIECE = 6;
    const size_t total = 3 * arr, piece = (total + NPIECE - 1) / NPIECE;
    for (auto &e : r->bout_ev)
      if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int k = 0; k < NPIECE; k++) {
      const size_t lo = std::min(total, (size_t)k * piece), hi = std::min(total, lo + piece);
      if (hi > lo) HIPCHK(hipMemcpyAsync(r->h_bout + lo, r->d_bout.p + lo, sizeof(double) * (hi - lo), hipMemcpyDeviceToHost, r->stream));
      HIPCHK(hipEventRecord(r->bout_ev[k], r->stream));
    }
    for (int k = 0; k < NPIECE; k++) {
      const size_t lo = std::min(total, (size_t)k * piece), hi = std::min(total, lo + piece);
      HIPCHK(hipEventSynchronize(r->bout_ev[k]));
      for (size_t x = lo; x < hi;) {      // (a piece may straddle two of the three arrays)
        const size_t i = x / arr, n_here = std::min(hi, (i + 1) * arr) - x;
        std::memcpy(outs[i] + (x - i * arr), r->h_bout + x, sizeof(double) * n_here);
        x += n_here;
      }
    }
    t_d2h = times ? since() : 0.0;
  }
  if (times)
    fprintf(stderr, "radiate_ir_batch: %d columns%s: plan + enqueue %.0f us, kernels done at %.0f, results with the caller at %.0f%s\n",
            n, green ? " (response form)" : "", t_enq, t_kern, t_d2h, direct ? " (its arrays page-locked)" : " (through the pinned block, six pieces)");
  CATCH(err)
}

// Config 4 of BASELINE.json (many independent columns): every column is one full
// Radtran%TOA_fluxes (clima_radtran.f90:320-342), but the batch is moved to HBM in one copy, the
// calls are enqueued back to back with no host round trip between them, and the level fluxes
// come back in one copy.  Arrays carry the column as their LAST (slowest) dimension:
// T, P, dz (nz, ncol); densities (nz, nsp, ncol); pdensities, radii (nz, np, ncol).
// Outputs: ISR, OLR (ncol); fluxes (nz+1, 5, ncol) = ir up, ir down, solar up, solar down, f_total,
// or NULL.  The handle's own wrk / f_total hold the last column afterwards.
void radtran_toa_fluxes_batch(void *ptr, const int *ncol, const double *T_surface, const double *T, const double *P,
                              const double *densities, const double *dz, const int *has_particles,
                              const double *pdensities, const double *radii, double *ISR, double *OLR,
                              double *fluxes, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  if (*ncol < 1) { set_err(err, "\"T\" has the wrong input dimension."); return; }
  if (r->shard_world != 1) { set_err(err, "toa_fluxes_batch is not available on a bin-sharded handle"); return; }
  const int hp = has_particles ? *has_particles : 0;
  if (r->np > 0 && !hp) { set_err(err, "\"pdensities\" and \"radii\" are required arguments."); return; }
  TRY
  const int nz = r->nz, nl = nz + 1, n = *ncol;
  const size_t cc = r->col_count;
  std::vector<double> h((size_t)n * cc, 0.0);
  std::vector<int> nsrc_h(n);
  for (int c = 0; c < n; c++) {  // the device layout of one column (do_upload)
    double *d = h.data() + (size_t)c * cc;
    d[0] = T_surface[c];
    std::memcpy(d + 1, T + (size_t)c * nz, sizeof(double) * nz);
    std::memcpy(d + 1 + nz, P + (size_t)c * nz, sizeof(double) * nz);
    std::memcpy(d + 1 + 2 * nz, dz + (size_t)c * nz, sizeof(double) * nz);
    std::memcpy(d + 1 + 3 * nz, densities + (size_t)c * nz * r->nsp, sizeof(double) * (size_t)nz * r->nsp);
    double *dp = d + 1 + 3 * nz + (size_t)nz * r->nsp;
    if (r->np > 0) {
      std::memcpy(dp, pdensities + (size_t)c * nz * r->np, sizeof(double) * (size_t)nz * r->np);
      std::memcpy(dp + (size_t)nz * r->np, radii + (size_t)c * nz * r->np, sizeof(double) * (size_t)nz * r->np);
    }
    nsrc_h[c] = build_meta(r, d + 1, d + 1 + nz, d + 1 + 2 * nz, d + 1 + 3 * nz, r->np > 0 ? dp : nullptr,
                           r->np > 0 ? dp + (size_t)nz * r->np : nullptr, r->np > 0, reinterpret_cast<int *>(d + r->meta_ofs));
  }
  if (r->d_cols_arena.n < h.size()) r->d_cols_arena.alloc(h.size());
  if (r->d_flux_arena.n < (size_t)n * 5 * nl) r->d_flux_arena.alloc((size_t)n * 5 * nl);
  HIPCHK(hipMemcpyAsync(r->d_cols_arena.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, r->stream));
  r->column_has_particles = r->np > 0;
  const int first_call = r->call_id + 1;
  std::vector<double> out((size_t)n * 5 * nl);
  // One launch of each kernel per chunk of columns (the fused grid takes the columns' work items in
  // turn, so one column's two-stream tail runs beside the next column's opacity tiles) where the fused
  // form covers the configuration; otherwise the calls of the columns are enqueued back to back.
  const int CH = std::min(n, r->batch_cols_in_flight);
  bool one_launch = r->fused && r->ng == 8 && !r->ts_block_mode && (nz + 63) / 64 >= 2 && (nz + 63) / 64 <= 8 &&
                    (int)r->zenith_u.size() <= MAX_ZEN && !((nz + 63) / 64 > 4 && (r->rebin_mode != 0 || r->cust_on)) &&
                    integrate_chunks(std::max(r->ir_n, r->sol_n)) * (32 + 16) * sizeof(double) <= 64 * 1024;
  if (const char *e = getenv("CLIMA_HIP_BATCH_ONE_LAUNCH")) one_launch = one_launch && atoi(e) != 0;
  const size_t tiles = ((size_t)r->op_n * nz + 255) / 256;
  if (one_launch) {
    const size_t pc = r->prep_count, oc = r->opr_count, rc = res_block_count(r);
    if (r->d_prep_arena.n < pc * CH) r->d_prep_arena.alloc(pc * CH);
    if (r->d_opr_arena.n < oc * CH) r->d_opr_arena.alloc(oc * CH);
    if (r->d_res_arena.n < rc * CH) { r->d_res_arena.alloc(rc * CH); r->d_res_arena.zero(r->stream); }
    if (r->d_done.n < tiles * CH + 1) { r->d_done.alloc(tiles * CH + 1); r->d_done.zero(r->stream); }
  }
  auto run_all = [&](bool allow_fused) {
    if (one_launch && allow_fused) {
      for (int c0 = 0; c0 < n; c0 += CH) {
        BatchCtx bc;
        bc.ncol = std::min(CH, n - c0);
        bc.col = r->d_cols_arena.p + (size_t)c0 * cc;
        bc.prep = r->d_prep_arena.p; bc.opr = r->d_opr_arena.p; bc.res = r->d_res_arena.p;
        bc.flux = r->d_flux_arena.p + (size_t)c0 * 5 * nl;
        bc.done = r->d_done.p;
        bc.bs = BatchStrides{cc, r->prep_count, r->opr_count, res_block_count(r), (size_t)5 * nl, (int)tiles};
        enqueue_radiate(r, true, true, true, &bc);
      }
    } else {
      for (int c = 0; c < n; c++) {
        r->col_override = r->d_cols_arena.p + (size_t)c * cc;
        r->nsrc_override = nsrc_h[c];
        r->flux_override = r->d_flux_arena.p + (size_t)c * 5 * nl;
        r->ftot_override = r->flux_override + 4 * nl;
        try {
          enqueue_radiate(r, true, true, allow_fused);
        } catch (...) {
          r->col_override = r->flux_override = r->ftot_override = nullptr;
          throw;
        }
      }
      r->col_override = r->flux_override = r->ftot_override = nullptr;
    }
    HIPCHK(hipMemcpyAsync(out.data(), r->d_flux_arena.p, sizeof(double) * out.size(), hipMemcpyDeviceToHost, r->stream));
    HIPCHK(hipMemcpyAsync(r->h_errflag, r->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, r->stream));
    // the handle's own level fluxes = the last column's
    HIPCHK(hipMemcpyAsync(r->d_flux_n.p, r->d_flux_arena.p + (size_t)(n - 1) * 5 * nl, sizeof(double) * 4 * nl, hipMemcpyDeviceToDevice, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    resolve_events(r);
  };
  run_all(true);
  bool fell_back = false;
  if (r->h_errflag[1] >= first_call) {  // a fused hand-off wait expired in some column: the batch again, unfused
    r->fused_fallbacks++;
    run_all(false);
    fell_back = true;
  }
  r->checked_timeout = r->call_id;
  r->small_valid = false; r->small_in_host = false;   // (the device rows changed: fetch them)
  r->column_loaded = false;   // d_col does not hold the last column: a resident call needs an upload first
  // What the handle holds afterwards is the LAST column's, like after n single calls: its level rows (copied above),
  // and its spectra / band optical depths -- the one-launch form left those in the batch arena (copied here);
  // its optical properties stay in the arena (opr_valid false: an IR-only call needs a compute_opacity call first),
  // unless the batch fell back to one call per column, which works in the handle's own buffers.
  const bool in_arena = one_launch && !fell_back;
  r->opr_valid = !in_arena;
  if (in_arena) {
    const double *res = r->d_res_arena.p + (size_t)((n - 1) % CH) * res_block_count(r);
    const size_t nli = (size_t)r->ir.nw * nl, nls = (size_t)r->sol.nw * nl, nzi = (size_t)r->ir.nw * nz, nzs = (size_t)r->sol.nw * nz;
    auto d2d = [&](DevBuf<double> &dst, const double *src, size_t cnt) {
      HIPCHK(hipMemcpyAsync(dst.p, src, sizeof(double) * cnt, hipMemcpyDeviceToDevice, r->stream));
    };
    d2d(r->wrk_ir.fup_a, res, nli); d2d(r->wrk_ir.fdn_a, res + nli, nli); d2d(r->wrk_ir.tau_band, res + 2 * nli, nzi);
    const double *rs = res + 2 * nli + nzi;
    d2d(r->wrk_sol.fup_a, rs, nls); d2d(r->wrk_sol.fdn_a, rs + nls, nls); d2d(r->wrk_sol.amean, rs + 2 * nls, nls);
    d2d(r->wrk_sol.tau_band, rs + 3 * nls, nzs);
    HIPCHK(hipStreamSynchronize(r->stream));
  }
  if (*r->h_errflag >= first_call) {
    r->checked_id = r->call_id;
    set_err(err, "Opacity computation failed in one or more wavelength bins.");  // clima_radtran_types.f90:773-776
    return;
  }
  r->checked_id = r->call_id;
  for (int c = 0; c < n; c++) {
    double *f = out.data() + (size_t)c * 5 * nl;
    for (int i = 0; i < nl; i++) f[4 * nl + i] = (f[3 * nl + i] - f[2 * nl + i]) + (f[1 * nl + i] - f[0 * nl + i]);  // :287
    ISR[c] = f[3 * nl + nz] - f[2 * nl + nz];        // clima_radtran.f90:339-340
    OLR[c] = -(f[1 * nl + nz] - f[0 * nl + nz]);
    if (fluxes) std::memcpy(fluxes + (size_t)c * 5 * nl, f, sizeof(double) * 5 * nl);
  }
  CATCH(err)
}

void radtran_synchronize(void *ptr, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  TRY
  settle(r);
  if (!r->deferred_err.empty()) {   // a getter without an `err` argument met a failure since the last check
    const std::string m = r->deferred_err;
    r->deferred_err.clear();
    throw HipFail{m};
  }
  surface_device_error(r, err);
  CATCH(err)
}

void radtran_radiate_wrapper(void *ptr, const double *T_surface, const int *dim_T, const double *T,
                             const int *dim_P, const double *P, const int *dim1_d, const int *dim2_d,
                             const double *densities, const int *dim_dz, const double *dz,
                             const int *has_particles, const int *dim1_p, const int *dim2_p,
                             const double *pdensities, const int *dim1_r, const int *dim2_r, const double *radii,
                             const int *compute_solar, const int *compute_opacity, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  const int hp = has_particles ? *has_particles : 0;
  if (!check_dims(r, *dim_T, *dim_P, *dim1_d, *dim2_d, *dim_dz, hp, dim1_p ? *dim1_p : 0, dim2_p ? *dim2_p : 0,
                  dim1_r ? *dim1_r : 0, dim2_r ? *dim2_r : 0, hp ? pdensities : nullptr, hp ? radii : nullptr, err))
    return;
  TRY
#ifdef CLIMA_TRACE_SYNC
  static double acc[4] = {0, 0, 0, 0};
  static long ncalls = 0;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
#endif
  do_upload(r, *T_surface, T, P, densities, dz, hp ? pdensities : nullptr, hp ? radii : nullptr);
#ifdef CLIMA_TRACE_SYNC
  const double t1 = now();
#endif
  r->want_host_out = true;
  try {
    enqueue_radiate(r, *compute_solar != 0, *compute_opacity != 0);
  } catch (...) {
    r->want_host_out = false;
    throw;
  }
  r->want_host_out = false;
#ifdef CLIMA_TRACE_SYNC
  const double t2 = now();
#endif
  fetch_small(r);
#ifdef CLIMA_TRACE_SYNC
  const double t3 = now();
  acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; ncalls++;
  if (ncalls % 100 == 0) { fprintf(stderr, "sync trace over 100 calls: upload %.1f us, enqueue %.1f us, fetch+sync %.1f us\n", acc[0] / 100, acc[1] / 100, acc[2] / 100); acc[0] = acc[1] = acc[2] = 0; }
#endif
  if (surface_device_error(r, err)) return;
  const int nl = r->nz + 1;
  for (int i = 0; i < nl; i++) r->f_total[i] = r->h_small[4 * nl + i];
  CATCH(err)
}

void radtran_toa_fluxes_wrapper(void *ptr, const double *T_surface, const int *dim_T, const double *T,
                                const int *dim_P, const double *P, const int *dim1_d, const int *dim2_d,
                                const double *densities, const int *dim_dz, const double *dz,
                                const int *has_particles, const int *dim1_p, const int *dim2_p,
                                const double *pdensities, const int *dim1_r, const int *dim2_r, const double *radii,
                                const int *compute_solar, const int *compute_opacity, double *ISR, double *OLR,
                                char *err) {
  radtran_radiate_wrapper(ptr, T_surface, dim_T, T, dim_P, P, dim1_d, dim2_d, densities, dim_dz, dz,
                          has_particles, dim1_p, dim2_p, pdensities, dim1_r, dim2_r, radii, compute_solar,
                          compute_opacity, err);
  if (err && err[0]) return;
  Radtran *r = as_rad(ptr);
  if (!r) return;
  const int nl = r->nz + 1, nz = r->nz;
  // clima_radtran.f90:339-340
  *ISR = (r->h_small[3 * nl + nz] - r->h_small[2 * nl + nz]);
  *OLR = -(r->h_small[1 * nl + nz] - r->h_small[0 * nl + nz]);
}

// Bench hook: `n` synchronous radtran_toa_fluxes_wrapper calls (host arrays in, ISR / OLR out, one stream
// synchronise each) timed one by one with the host's steady clock INSIDE the library -- the figure a Fortran or C
// caller of TOA_fluxes sees, without the ~14 us a ctypes call adds (SURVEY.md 8(d) "Metric").  us[n].
void clima_bench_toa_fluxes(void *ptr, const int *n, const double *T_surface, const int *dim_T, const double *T,
                            const int *dim_P, const double *P, const int *dim1_d, const int *dim2_d,
                            const double *densities, const int *dim_dz, const double *dz, const int *has_particles,
                            const int *dim1_p, const int *dim2_p, const double *pdensities, const int *dim1_r,
                            const int *dim2_r, const double *radii, double *us, double *ISR, double *OLR, char *err) {
  clear_err(err);
  for (int i = 0; i < *n; i++) {
    const int one = 1;
    const auto t0 = std::chrono::steady_clock::now();
    radtran_toa_fluxes_wrapper(ptr, T_surface, dim_T, T, dim_P, P, dim1_d, dim2_d, densities, dim_dz, dz, has_particles,
                               dim1_p, dim2_p, pdensities, dim1_r, dim2_r, radii, &one, &one, ISR, OLR, err);
    us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (err && err[0]) return;
  }
}

// Bench hook: `n` resident calls, each followed by its own radtran_synchronize (no PCIe for the column, one host
// round trip per call), timed one by one inside the library.  us[n].
void clima_bench_resident_sync(void *ptr, const int *n, double *us, char *err) {
  clear_err(err);
  const int one = 1;
  for (int i = 0; i < *n; i++) {
    const auto t0 = std::chrono::steady_clock::now();
    radtran_radiate_resident(ptr, &one, &one, err);
    if (err && err[0]) return;
    radtran_synchronize(ptr, err);
    us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (err && err[0]) return;
  }
}

// Bench hook (timing only): ONE resident call's launches captured into a hipGraph, then `n` times hipGraphLaunch +
// stream synchronise, timed one by one -- what a graph would make of clima_bench_resident_sync's loop.  The replayed
// launches carry the captured call's id, so the two-stream blocks of the fused grid find their flags already set and
// do not wait: results are NOT valid (the handle is marked for a fresh call afterwards), the time is a lower bound
// (the missing wait is worth ~1 us).  With mode = 1 each pass replays `k` launches of the graph
// before it synchronises (k calls back to back).  us[n].
void clima_bench_resident_graph(void *ptr, const int *n, const int *k, double *us, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2 || !r->column_loaded) { set_err(err, "no column has been uploaded"); return; }
  TRY
  const int one = 1;
  radtran_radiate_resident(ptr, &one, &one, err);     // steady state: buffers allocated, fields uploaded
  if (err && err[0]) return;
  HIPCHK(hipStreamSynchronize(r->stream));
  // (nothing that synchronises or allocates may run inside the capture: stale fields are uploaded by the call above;
  // per-kernel event profiling records events of its own)
  if (r->fields_dirty || r->profile != 0) throw HipFail{"clima_bench_resident_graph: not with stale fields or profiling on"};
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  HIPCHK(hipStreamBeginCapture(r->stream, hipStreamCaptureModeThreadLocal));
  try {
    enqueue_radiate(r, true, true);
  } catch (...) {
    // a failure inside the captured section must not leave the handle's stream in capture mode (every later call
    // on it would fail): close the capture, drop what was captured, pass the error on
    hipGraph_t g2 = nullptr;
    (void)hipStreamEndCapture(r->stream, &g2);
    if (g2) (void)hipGraphDestroy(g2);
    (void)hipGetLastError();
    throw;
  }
  HIPCHK(hipStreamEndCapture(r->stream, &graph));
  try {
    HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 10; i++) HIPCHK(hipGraphLaunch(exec, r->stream));
    HIPCHK(hipStreamSynchronize(r->stream));
    for (int i = 0; i < *n; i++) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int j = 0; j < std::max(1, *k); j++) HIPCHK(hipGraphLaunch(exec, r->stream));
      HIPCHK(hipStreamSynchronize(r->stream));
      us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
  } catch (...) {
    if (exec) (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    throw;
  }
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  r->small_valid = false; r->small_in_host = false; r->opr_valid = false;
  radtran_radiate_resident(ptr, &one, &one, err);     // leave the handle with a valid call
  if (err && err[0]) return;
  HIPCHK(hipStreamSynchronize(r->stream));
  CATCH(err)
}

void radtran_apply_radiation_enhancement(void *ptr, const double *rad_enhancement) {
  Radtran *r = as_rad(ptr);
  if (!r || r->state != 2) return;
  try {  // clima_radtran.f90:402-411
    const int nl = r->nz + 1;
    settle(r);   // the results being scaled are final (a hand-off that timed out has been repaired BEFORE, not after)
    launch_scale(r->wrk_sol.fdn_a.p, r->wrk_sol.fdn_a.n, *rad_enhancement, r->stream);
    launch_scale(r->wrk_sol.fup_a.p, r->wrk_sol.fup_a.n, *rad_enhancement, r->stream);
    launch_scale(r->d_flux_n.p + 2 * nl, (size_t)2 * nl, *rad_enhancement, r->stream);
    // a bin-sharded handle keeps its PARTIAL solar rows for the IR-only steps that follow: they are scaled too
    if (r->shard_world > 1 && r->d_flux_part.p) launch_scale(r->d_flux_part.p + 2 * nl, (size_t)2 * nl, *rad_enhancement, r->stream);
    // (with a communicator the slot behind the rows carries the step's status word, and f_total is formed on the host
    // from the rows in any case: fetch_small)
    if (!r->comm) launch_f_total(r->nz, r->d_flux_n.p, r->d_f_total.p, r->stream);
    r->small_valid = false; r->small_in_host = false;   // (the device rows changed: fetch them)
    fetch_small(r);
    for (int i = 0; i < nl; i++) r->f_total[i] = r->h_small[4 * nl + i];
  } catch (...) {
  }
}

void radtran_flux_device_ptr(void *ptr, void **dptr, int *count) {
  Radtran *r = as_rad(ptr);
  *dptr = r ? (void *)r->d_flux_n.p : nullptr;
  *count = r ? 4 * (r->nz + 1) : 0;
}

void radtran_set_bin_shard(void *ptr, const int *rank, const int *world, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  if (*world < 1 || *rank < 0 || *rank >= *world) { set_err(err, "invalid shard (rank, world)"); return; }
  // a communicator fixes the shard (rank, nranks) -- except a one-rank communicator, on which any shard may be
  // rehearsed (one rank's share of an N-GPU step on one GPU: bench.py CLIMA_BENCH_FAKE_SHARD)
  if (r->comm && r->comm_n > 1 && (*world != r->comm_n || *rank != r->comm_rank)) {
    set_err(err, "the bin shard of a handle with a communicator is (rank, nranks) of the communicator");
    return;
  }
  TRY
  HIPCHK(hipStreamSynchronize(r->stream));
  r->shard_rank = *rank; r->shard_world = *world;
  compute_shard(r);
  // bins outside the shard hold zeros so that sharded per-bin spectra add up across ranks
  for (WrkObj *w : {&r->wrk_ir, &r->wrk_sol}) { w->fup_a.zero(r->stream); w->fdn_a.zero(r->stream); w->amean.zero(r->stream); w->tau_band.zero(r->stream); }
  r->d_flux_n.zero(r->stream);
  if (*world > 1) {
    if (r->d_flux_part.n < (size_t)4 * (r->nz + 1)) r->d_flux_part.alloc((size_t)4 * (r->nz + 1));
    r->d_flux_part.zero(r->stream);
  }
  HIPCHK(hipStreamSynchronize(r->stream));
  CATCH(err)
}

// ---- the library's own multi-GPU step ---------------------------------------
// One process per GPU.  Each rank builds the same Radtran, then joins a communicator; from then on every
// radiate() / TOA_fluxes() / radiate_resident() of the handle works on the rank's own spectral bins and ends
// with one ncclAllReduce (sum, f64, 4 (nz+1) + 1 values) on the handle's stream, so that wrk_ir%fup_n ...
// f_total, ISR and OLR are those of the whole spectrum on every rank (per-bin spectra stay sharded: zeros
// outside the rank's bins).  The reduction being distributed: clima_radtran_radiate.f90:184-192.

void radtran_set_device(const int *device, char *err) {
  clear_err(err);
  TRY
  int n = 0;
  HIPCHK(hipGetDeviceCount(&n));
  if (*device < 0 || *device >= n) throw HipFail{"radtran_set_device: no such HIP device"};
  HIPCHK(hipSetDevice(*device));
  CATCH(err)
}

void radtran_comm_unique_id(char *id, char *err) {
  clear_err(err);
  TRY
  static_assert(sizeof(ncclUniqueId) == CLIMA_COMM_ID_BYTES, "CLIMA_COMM_ID_BYTES is RCCL's NCCL_UNIQUE_ID_BYTES");
  ncclUniqueId u;
  NCCLCHK(ncclGetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  CATCH(err)
}

static void comm_attach(Radtran *r, int nranks, int rank, const char *id) {
  if (r->comm) throw HipFail{"this handle already has a communicator"};
  HIPCHK(hipSetDevice(r->device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  NCCLCHK(ncclCommInitRank(&r->comm, nranks, u, rank));
  r->comm_n = nranks; r->comm_rank = rank;
}

void radtran_comm_init_rank(void *ptr, const int *nranks, const int *rank, const char *id, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  if (*nranks < 1 || *rank < 0 || *rank >= *nranks) { set_err(err, "invalid communicator (rank, nranks)"); return; }
  TRY
  comm_attach(r, *nranks, *rank, id);
  CATCH(err)
  if (err && err[0]) return;
  radtran_set_bin_shard(ptr, rank, nranks, err);
}

// The same for hosts without a message layer of their own (a plain Fortran program started once per GPU): rank 0
// writes the id to `path` (created atomically; removed again once every rank has joined), the others wait for it.
// `path` must be new for every job.
void radtran_comm_init_file(void *ptr, const int *nranks, const int *rank, const char *path, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  if (*nranks < 1 || *rank < 0 || *rank >= *nranks) { set_err(err, "invalid communicator (rank, nranks)"); return; }
  if (!path || !path[0]) { set_err(err, "radtran_comm_init_file: empty path"); return; }
  TRY
  char id[CLIMA_COMM_ID_BYTES];
  const std::string file(path), tmp = file + ".tmp";
  // What is published is a record, not the bare id: magic, the communicator's size and a job nonce (CLIMA_COMM_NONCE,
  // else SLURM_JOB_ID, else empty) in front of it.  A rank other than 0 joins only on a record whose size and nonce are
  // its own -- a leftover of a crashed or re-launched job on the same path (which rank 0 also removes before it
  // publishes) is not taken for this job's id, and a rank that finds nothing valid in time returns an error instead of
  // blocking in ncclCommInitRank on a dead id.  Without a nonce the file's age is the only guard (60 s against this
  // call's start, by this host's clock): hosts sharing a path across jobs should set one.
  struct Record { char magic[8]; int nranks; char nonce[64]; char id[CLIMA_COMM_ID_BYTES]; } rec;
  std::memset(&rec, 0, sizeof(rec));
  std::memcpy(rec.magic, "CLRCOMM1", 8);
  rec.nranks = *nranks;
  const char *nonce = getenv("CLIMA_COMM_NONCE");
  if (!nonce || !nonce[0]) nonce = getenv("SLURM_JOB_ID");
  if (nonce) std::strncpy(rec.nonce, nonce, sizeof(rec.nonce) - 1);
  if (*rank == 0) {
    (void)std::remove(file.c_str());   // whatever an earlier job left there
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    std::memcpy(rec.id, id, sizeof(id));
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(&rec, 1, sizeof(rec), f) != sizeof(rec) || std::fclose(f) != 0 || std::rename(tmp.c_str(), file.c_str()) != 0)
      throw HipFail{"radtran_comm_init_file: cannot write " + file};
  } else {
    bool got = false, stale = false;
    const time_t t_enter = time(nullptr);
    int max_tries = 6000;   // up to ~120 s (CLIMA_COMM_WAIT_S: another bound, in seconds -- the tests use 1)
    if (const char *w = getenv("CLIMA_COMM_WAIT_S")) max_tries = std::max(1, atoi(w) * 50);
    for (int tries = 0; tries < max_tries && !got; tries++) {
      struct stat st;
      if (stat(file.c_str(), &st) == 0 && st.st_size == (off_t)sizeof(rec)) {
        Record got_rec;
        FILE *f = std::fopen(file.c_str(), "rb");
        const bool read_ok = f && std::fread(&got_rec, 1, sizeof(got_rec), f) == sizeof(got_rec);
        if (f) std::fclose(f);
        const bool mine = read_ok && std::memcmp(got_rec.magic, rec.magic, 8) == 0 && got_rec.nranks == rec.nranks &&
                          std::memcmp(got_rec.nonce, rec.nonce, sizeof(rec.nonce)) == 0 &&
                          (rec.nonce[0] || st.st_mtime >= t_enter - 60);
        if (mine) { std::memcpy(id, got_rec.id, sizeof(id)); got = true; }
        else if (read_ok) stale = true;
      }
      if (!got) usleep(20000);
    }
    if (!got) throw HipFail{std::string("radtran_comm_init_file: rank 0 did not publish a record for this job at ") + file +
                            (stale ? " (a record of another job, communicator size or nonce is there)" : "")};
  }
  comm_attach(r, *nranks, *rank, id);   // collective: returns once every rank has joined
  if (*rank == 0) (void)std::remove(file.c_str());
  CATCH(err)
  if (err && err[0]) return;
  radtran_set_bin_shard(ptr, rank, nranks, err);
}

void radtran_comm_get(void *ptr, int *nranks, int *rank, int *reduces) {
  Radtran *r = as_rad(ptr);
  *nranks = (r && r->comm) ? r->comm_n : 0;
  *rank = (r && r->comm) ? r->comm_rank : 0;
  if (reduces) *reduces = r ? (int)std::min<long>(r->comm_reduces, 2147483647L) : 0;
}

void radtran_comm_destroy(void *ptr) {
  Radtran *r = as_rad(ptr);
  if (!r || !r->comm) return;
  (void)hipStreamSynchronize(r->stream);
  (void)ncclCommDestroy(r->comm);
  r->comm = nullptr; r->comm_n = 1; r->comm_rank = 0;
  const int zero = 0, one = 1;
  char e[CLIMA_ERR_LEN + 1];
  radtran_set_bin_shard(ptr, &zero, &one, e);   // back to the whole spectrum
}

void radtran_bin_shard_get(void *ptr, int *op_lo, int *op_n, int *ir_lo, int *ir_n, int *sol_lo, int *sol_n) {
  Radtran *r = as_rad(ptr);
  if (!r) return;
  *op_lo = r->op_lo; *op_n = r->op_n; *ir_lo = r->ir_lo; *ir_n = r->ir_n; *sol_lo = r->sol_lo; *sol_n = r->sol_n;
}

void radtran_finish_reduced(void *ptr, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  // f_total = (sol_dn - sol_up) + (ir_dn - ir_up) of the reduced rows is formed on the host when the
  // results are fetched (fetch_small), like after every call: nothing to launch here, the level
  // rows just have to be read again
  r->small_valid = false; r->small_in_host = false;   // (the device rows changed: fetch them)
}

void radtran_stream_get(void *ptr, void **stream) {
  Radtran *r = as_rad(ptr);
  *stream = r ? (void *)r->stream : nullptr;
}

void radtran_profile_set(void *ptr, const int *enable) {
  Radtran *r = as_rad(ptr);
  if (r) r->profile = (*enable == 2) ? 2 : (*enable != 0 ? 1 : 0);
}
void radtran_profile_stride_set(void *ptr, const int *stride) {
  Radtran *r = as_rad(ptr);
  if (r) r->profile_stride = std::max(1, *stride);
}
void radtran_profile_reset(void *ptr) {
  Radtran *r = as_rad(ptr);
  if (!r) return;
  for (int i = 0; i < 4; i++) { r->k_ms[i] = 0; r->k_n[i] = 0; }
  r->timer_calls = r->profile_stride - 1;  // with a sampling stride, the next call is a sampled one
}
void radtran_kernel_time_get(void *ptr, const int *kernel_id, double *ms_total, int *launches, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (*kernel_id < 0 || *kernel_id > 3) { set_err(err, "kernel id out of range"); return; }
  TRY
  if (!r->pending.empty()) { HIPCHK(hipStreamSynchronize(r->stream)); resolve_events(r); }
  *ms_total = r->k_ms[*kernel_id];
  *launches = r->k_n[*kernel_id];
  CATCH(err)
}

void radtran_algorithmic_bytes(void *ptr, double *bytes_tables_distinct, double *bytes_in, double *bytes_out,
                               double *bytes_tables_full, char *err) {
  // B_alg = B_tab(distinct interpolation nodes touched by the column)
  // + B_in + B_out; intermediates (opr) get no credit.
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2 || !r->column_loaded) { set_err(err, "no column has been uploaded"); return; }
  const int nz = r->nz;
  const double nw = r->nw;
  double distinct = 0.0, full = 0.0;
  auto nodes1d = [&](const std::vector<double> &axis, const std::vector<double> &vals) {
    std::set<int> s;
    double lo = *std::min_element(axis.begin(), axis.end()), hi = *std::max_element(axis.begin(), axis.end());
    for (double v : vals) { int i = host_bracket(axis, std::min(std::max(v, lo), hi)); s.insert(i); s.insert(i + 1); }
    return (double)s.size();
  };
  std::vector<double> lp(nz);
  for (int j = 0; j < nz; j++) lp[j] = std::log10(r->last_P[j]);
  for (auto *k : r->k) {
    std::set<std::pair<int, int>> s;
    double plo = *std::min_element(k->log10P.begin(), k->log10P.end()), phi = *std::max_element(k->log10P.begin(), k->log10P.end());
    double tlo = *std::min_element(k->temp.begin(), k->temp.end()), thi = *std::max_element(k->temp.begin(), k->temp.end());
    for (int j = 0; j < nz; j++) {
      int iP = host_bracket(k->log10P, std::min(std::max(lp[j], plo), phi));
      int iT = host_bracket(k->temp, std::min(std::max(r->last_T[j], tlo), thi));
      for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) s.insert({iP + a, iT + b});
    }
    distinct += 8.0 * k->ng * nw * (double)s.size();
    full += 8.0 * k->ng * nw * k->nP * k->nT;
  }
  for (auto *v : {&r->cia, &r->pxs, &r->ray})
    for (auto *xs : *v) {
      if (xs->dim) { distinct += 8.0 * nw * nodes1d(xs->temp, r->last_T); full += 8.0 * nw * xs->nT; }
      else { distinct += 8.0 * nw; full += 8.0 * nw; }
    }
  if (r->has_cont) { distinct += 2 * 8.0 * nw * nodes1d(r->cont_temp, r->last_T); full += 2 * 8.0 * nw * r->cont_nT; }
  for (auto *p : r->part) {
    if (!r->last_radii.empty()) {
      std::vector<double> rr(r->last_radii.begin() + (size_t)p->p_ind * nz, r->last_radii.begin() + (size_t)(p->p_ind + 1) * nz);
      distinct += 3 * 8.0 * nw * nodes1d(p->radii, rr);
    }
    full += 3 * 8.0 * nw * p->nrad;
  }
  const int nzen = (int)r->zenith_u.size();
  *bytes_tables_distinct = distinct;
  *bytes_tables_full = full;
  *bytes_in = 8.0 * (nz * (3.0 + r->nsp + 2.0 * r->np) + 2.0 * nw + nzen * 2.0 + r->sol.nw);
  *bytes_out = 8.0 * ((2.0 * (nz + 1) + nz) * r->ir.nw + (3.0 * (nz + 1) + nz) * r->sol.nw + 5.0 * (nz + 1));
}

// N_PT = distinct (P,T) nodes of a k-table that the last uploaded column's bilinear
// stencils touch (mean over the k-tables; <= nP*nT), N_T likewise for the 1-D (temperature) tables.
void radtran_algorithmic_nodes(void *ptr, double *n_pt, double *n_pt_full, double *n_t, double *n_t_full, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2 || !r->column_loaded) { set_err(err, "no column has been uploaded"); return; }
  const int nz = r->nz;
  std::vector<double> lp(nz);
  for (int j = 0; j < nz; j++) lp[j] = std::log10(r->last_P[j]);
  double spt = 0.0, sptf = 0.0;
  for (auto *k : r->k) {
    std::set<std::pair<int, int>> s;
    const double plo = *std::min_element(k->log10P.begin(), k->log10P.end()), phi = *std::max_element(k->log10P.begin(), k->log10P.end());
    const double tlo = *std::min_element(k->temp.begin(), k->temp.end()), thi = *std::max_element(k->temp.begin(), k->temp.end());
    for (int j = 0; j < nz; j++) {
      const int iP = host_bracket(k->log10P, std::min(std::max(lp[j], plo), phi));
      const int iT = host_bracket(k->temp, std::min(std::max(r->last_T[j], tlo), thi));
      for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) s.insert({iP + a, iT + b});
    }
    spt += (double)s.size(); sptf += (double)k->nP * k->nT;
  }
  auto nodes1d = [&](const std::vector<double> &axis) {
    std::set<int> s;
    const double lo = *std::min_element(axis.begin(), axis.end()), hi = *std::max_element(axis.begin(), axis.end());
    for (double v : r->last_T) { const int i = host_bracket(axis, std::min(std::max(v, lo), hi)); s.insert(i); s.insert(i + 1); }
    return (double)s.size();
  };
  double st = 0.0, stf = 0.0;
  int n1 = 0;
  for (auto *v : {&r->cia, &r->pxs})
    for (auto *xs : *v)
      if (xs->dim) { st += nodes1d(xs->temp); stf += xs->nT; n1++; }
  if (r->has_cont) { st += 2 * nodes1d(r->cont_temp); stf += 2 * r->cont_nT; n1 += 2; }
  *n_pt = r->k.empty() ? 0.0 : spt / r->k.size();
  *n_pt_full = r->k.empty() ? 0.0 : sptf / r->k.size();
  *n_t = n1 ? st / n1 : 0.0;
  *n_t_full = n1 ? stf / n1 : 0.0;
}

#ifdef CLIMA_STAMPS
extern "C" void clima_debug_stamps(void *ptr, long long *out) {
  Radtran *r = as_rad(ptr);
  (void)hipStreamSynchronize(r->stream);
  (void)hipMemcpy(out, r->d_stamps.p, (64 + 2 * 8192) * sizeof(long long), hipMemcpyDeviceToHost);
}
#endif

void radtran_opr_get(void *ptr, double *tau, double *w0, double *g, double *tau_band, char *err) {
  clear_err(err);
  GUARD(r, ptr, err);
  if (r->state != 2) { set_err(err, "Radtran is not constructed"); return; }
  TRY
  settle(r);       // (a repaired hand-off recomputes the optical properties)
  ensure_w0(r);
  HIPCHK(hipStreamSynchronize(r->stream));
  if (tau) HIPCHK(hipMemcpy(tau, r->d_tau.p, r->d_tau.n * sizeof(double), hipMemcpyDeviceToHost));
  if (w0) HIPCHK(hipMemcpy(w0, r->d_w0.p, r->d_w0.n * sizeof(double), hipMemcpyDeviceToHost));
  if (g) HIPCHK(hipMemcpy(g, r->d_g.p, r->d_g.n * sizeof(double), hipMemcpyDeviceToHost));
  if (tau_band) HIPCHK(hipMemcpy(tau_band, r->d_tau_band.p, r->d_tau_band.n * sizeof(double), hipMemcpyDeviceToHost));
  CATCH(err)
}

void clima_test_device_exp(const int *n, const double *x, double *y, char *err) {
  clear_err(err);
  TRY
  DevBuf<double> dx, dy;
  dx.alloc(*n); dy.alloc(*n);
  HIPCHK(hipMemcpy(dx.p, x, sizeof(double) * *n, hipMemcpyHostToDevice));
  launch_test_exp(dx.p, dy.p, *n, nullptr);
  HIPCHK(hipMemcpy(y, dy.p, sizeof(double) * *n, hipMemcpyDeviceToHost));
  CATCH(err)
}

// the table exp of the zenith-angle loop (base10 = 0: e^x, arguments <= 0) and of the opacity tile's
// table interpolations (base10 = 1: 10^x)
void clima_test_device_exp_table(const int *n, const int *base10, const double *x, double *y, char *err) {
  clear_err(err);
  TRY
  DevBuf<double> dx, dy;
  dx.alloc(*n); dy.alloc(*n);
  HIPCHK(hipMemcpy(dx.p, x, sizeof(double) * *n, hipMemcpyHostToDevice));
  launch_test_exp_tab(dx.p, dy.p, *n, *base10, nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(y, dy.p, sizeof(double) * *n, hipMemcpyDeviceToHost));
  CATCH(err)
}

void clima_test_device_rcp(const int *n, const double *x, double *y, char *err) {
  clear_err(err);
  TRY
  DevBuf<double> dx, dy;
  dx.alloc(*n); dy.alloc((size_t)4 * *n);
  HIPCHK(hipMemcpy(dx.p, x, sizeof(double) * *n, hipMemcpyHostToDevice));
  launch_test_rcp(dx.p, dy.p, *n, nullptr);
  HIPCHK(hipMemcpy(y, dy.p, sizeof(double) * 4 * *n, hipMemcpyDeviceToHost));
  CATCH(err)
}

void clima_test_wave_scan(const int *nwaves, const double *a, const double *b, double *out, char *err) {
  clear_err(err);
  TRY
  const size_t n = (size_t)*nwaves * 64;
  DevBuf<double> da, db, dout;
  da.alloc(n); db.alloc(n); dout.alloc(4 * n);
  HIPCHK(hipMemcpy(da.p, a, sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(db.p, b, sizeof(double) * n, hipMemcpyHostToDevice));
  launch_test_wscan(da.p, db.p, dout.p, *nwaves, nullptr);
  HIPCHK(hipMemcpy(out, dout.p, sizeof(double) * 4 * n, hipMemcpyDeviceToHost));
  CATCH(err)
}


// Test hook: ONE column through the production two-stream kernels with its optical properties
// and Planck values given directly -- the inputs and outputs of the reference's two_stream_ir /
// two_stream_solar (src/radtran/clima_radtran_twostream.f90:10-295), which is what
// tests/golden/twostream_golden.npz holds.  The column is presented as one bin that lies in both
// channels, `ng` g-points carrying the same tau/w0 with weights wbin (sum 1), one zenith angle u0 of
// weight 1, unit stellar flux and unit factors, so the kernels' weighted sums reproduce the solver's
// own outputs.  form: 0 wave-per-column kernel (k_twostream_w<slots>), 1 workgroup-per-bin kernel
// (k_twostream), 2 the two-stream part of the fused grid (k_fused, whole-wave form, slots 2..8, ng = 8), 4 the
// same in the half-wave form (two g-point columns per wave, slots = ceil(nz/32) = 3..7), 5 the same in the
// paired form (every layer 2m+1 a copy of layer 2m: even nz, slots 2, 4, 6, 8; the caller passes such a column),
// 3 batched shared-opacity IR kernel (k_twostream_ir_batch<slots, NW>, IR outputs only: slots 1..4 in blocks of 8
// g-point waves, 5..8 in blocks of 4), 6 the same with blocks of 4 waves at every slot count.
// slots = layer slots per lane (>= ceil(nz/64)).  Outputs are TOA-first like the solver's.
void clima_test_two_stream(const int *nz_, const int *ng_, const int *form, const int *slots, const double *tau,
                           const double *w0, const double *g, const double *bplanck, const double *ir_par,
                           const double *sol_par, const double *wbin, double *ir_fup, double *ir_fdn,
                           double *sol_fup, double *sol_fdn, double *sol_amean, char *err) {
  clear_err(err);
  TRY
  const int nz = *nz_, ng = *ng_, nl = nz + 1;
  if (nz < 1 || ng < 1 || ng > 32) throw HipFail{"clima_test_two_stream: bad nz / ng"};
  std::vector<double> h_tau((size_t)ng * nz), h_w0((size_t)ng * nz);
  for (int c = 0; c < ng; c++)
    for (int i = 0; i < nz; i++) { h_tau[(size_t)c * nz + i] = tau[i]; h_w0[(size_t)c * nz + i] = w0[i]; }
  DevBuf<double> d_tau, d_w0, d_g, d_tb, d_bp, d_wbin, d_freq, d_T, d_one, d_em, d_alb, d_zu, d_zw, d_ziu, d_out;
  d_tau.upload(h_tau); d_w0.upload(h_w0);
  d_g.upload(std::vector<double>(g, g + nz));
  d_tb.upload(std::vector<double>(nz, 0.0));
  d_bp.upload(std::vector<double>(bplanck, bplanck + nl));
  d_wbin.upload(std::vector<double>(wbin, wbin + ng));
  d_freq.upload(std::vector<double>{2.0e13, 1.0e13});
  d_T.upload(std::vector<double>(nl, 250.0));
  d_one.upload(std::vector<double>{1.0});
  d_em.upload(std::vector<double>{ir_par[0]});
  d_alb.upload(std::vector<double>{sol_par[1]});
  d_zu.upload(std::vector<double>{sol_par[0]});
  d_zw.upload(std::vector<double>{1.0});
  d_ziu.upload(std::vector<double>{1.0 / sol_par[0]});
  d_out.alloc((size_t)7 * nl); d_out.zero();
  DevBuf<int> d_qm;
  TwoStreamParams ts;
  std::memset(&ts, 0, sizeof(ts));
  ts.nz = nz; ts.ng = ng;
  ts.n_sol = (*form == 3 || *form == 6) ? 0 : 1; ts.n_ir = 1;
  ts.tau = d_tau.p; ts.w0 = d_w0.p; ts.g = d_g.p; ts.tau_band = d_tb.p;
  ts.wbin = d_wbin.p; ts.freq = d_freq.p;
  ts.bplanck = d_bp.p; ts.force_slots = *slots;
  ts.T = d_T.p; ts.T_surface = d_T.p;
  ts.emissivity = d_em.p; ts.has_hard_surface = ir_par[1] != 0.0 ? 1 : 0; ts.ir_tau_min = ir_par[2];
  ts.nzen = 1; ts.zen_u = d_zu.p; ts.zen_w = d_zw.p; ts.zen_iu = d_ziu.p;
  ts.zen_u_v[0] = sol_par[0]; ts.zen_w_v[0] = 1.0; ts.zen_iu_v[0] = 1.0 / sol_par[0];
  ts.albedo = d_alb.p; ts.photons_sol = d_one.p; ts.photon_scale_factor = 1.0; ts.diurnal_fac = 1.0;
  ts.am_f1 = d_one.p; ts.am_f2 = d_one.p; ts.am_dw = d_one.p;
  ts.ir_fup_a = d_out.p; ts.ir_fdn_a = d_out.p + nl; ts.sol_fup_a = d_out.p + 2 * nl; ts.sol_fdn_a = d_out.p + 3 * nl;
  ts.sol_amean = d_out.p + 4 * nl; ts.ir_tau_band = d_out.p + 5 * nl; ts.sol_tau_band = d_out.p + 6 * nl;
  bool ok = false;
  size_t lds = 0;
  if (*form == 0) ok = launch_twostream_w(ts, nullptr, &lds, false);
  else if (*form == 1) ok = launch_twostream(ts, nullptr, &lds);
  else if (*form == 2 || *form == 4 || *form == 5) {   // 4: the half-wave form (two g-point columns per wave), 5: the paired form
    if (*form == 5)
      for (int i = 0; i + 1 < nz; i += 2)
        if (tau[i] != tau[i + 1] || w0[i] != w0[i + 1] || g[i] != g[i + 1])
          throw HipFail{"clima_test_two_stream: form 5 needs a column of pairwise identical layers"};
    d_qm.upload(std::vector<int>{nz});  // the column's source-layer count
    ok = launch_fused_twostream_only(ts, *slots, d_qm.p, nullptr, *form == 4, *form == 5);
  }
  else if (*form == 3 || *form == 6) { ts.b_T = 0; ts.b_Ts = 0; ts.b_out = 0; ok = launch_twostream_ir_batch(ts, 1, nullptr, *form == 6 ? 4 : 0); }
  HIPCHK(hipGetLastError());
  if (!ok) throw HipFail{"clima_test_two_stream: this form does not cover the requested shape"};
  HIPCHK(hipDeviceSynchronize());
  std::vector<double> out((size_t)5 * nl);
  HIPCHK(hipMemcpy(out.data(), d_out.p, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
  double *dst[5] = {ir_fup, ir_fdn, sol_fup, sol_fdn, sol_amean};
  for (int a = 0; a < 5; a++)
    for (int n = 0; n < nl; n++) dst[a][n] = out[(size_t)a * nl + (nz - n)];  // the kernels store ground-first (radiate.f90:140-154)
  CATCH(err)
}

// Test hook of the response form (ir_green.inc), the counterpart of clima_test_two_stream for it: one column's tau / w0 /
// g (the same for every g-point, weights wbin) and ndev deviations (level k TOA-first, k = nz the surface; change of the
// Planck value there) -> per deviation the change of the level fluxes (TOA-first), sum over the g-points, exactly as the
// production kernels form it (k_green_factor / unit / local / accum_far / accum_mixed); only the Planck differences come
// from the caller instead of k_green_db.
void clima_test_ir_response(const int *nz_, const int *ng_, const double *tau, const double *w0, const double *g,
                            const double *ir_par, const double *wbin, const int *ndev_, const int *dev_k,
                            const double *dev_db, double *resp_up, double *resp_dn, char *err) {
  clear_err(err);
  TRY
  const int nz = *nz_, ng = *ng_, nl = nz + 1, N = 2 * nz, ndev = *ndev_;
  if (nz < 4 || ng < 1 || ng > 32 || ndev < 1) throw HipFail{"clima_test_ir_response: bad nz / ng / ndev"};
  std::vector<int> order(ndev);
  for (int i = 0; i < ndev; i++) { order[i] = i; if (dev_k[i] < 0 || dev_k[i] > nz) throw HipFail{"clima_test_ir_response: bad level"}; }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dev_k[a] < dev_k[b]; });
  const int ndev_pad = std::max(16, (ndev + 15) / 16 * 16), nblk = (nl + 15) / 16;
  std::vector<double> h_tau((size_t)ng * nz), h_w0((size_t)ng * nz), h_db(ndev_pad + 64, 0.0);
  for (int c = 0; c < ng; c++)
    for (int i = 0; i < nz; i++) { h_tau[(size_t)c * nz + i] = tau[i]; h_w0[(size_t)c * nz + i] = w0[i]; }
  std::vector<int> hi(ndev_pad, 0), mdev, mblk;
  for (int d = 0; d < ndev; d++) { hi[d] = dev_k[order[d]]; h_db[d] = dev_db[order[d]]; }
  for (int d = 0; d < ndev; d++)
    for (int blk = 0; blk < nblk; blk++)
      if (green_block_class(hi[d], blk, nz) == 2) { mdev.push_back(d); mblk.push_back(blk); }
  const int nmix = (int)mdev.size();
  hi.insert(hi.end(), mdev.begin(), mdev.end());
  hi.insert(hi.end(), mblk.begin(), mblk.end());
  DevBuf<double> d_tau, d_w0, d_g, d_wbin, d_em, d_db, d_work;
  DevBuf<int> d_idx;
  d_tau.upload(h_tau); d_w0.upload(h_w0); d_g.upload(std::vector<double>(g, g + nz));
  d_wbin.upload(std::vector<double>(wbin, wbin + ng)); d_em.upload(std::vector<double>{ir_par[0]});
  d_db.upload(h_db); d_idx.upload(hi);
  const size_t RQ = (size_t)N * ng, LQ = (size_t)nl * ng, FQ = (size_t)2 * nblk * 34 * ng;
  d_work.alloc(7 * RQ + 6 * LQ + FQ + 8 * LQ + 4 * (size_t)ng + (size_t)ndev_pad * 2 * nl); d_work.zero();
  GreenParams gp;
  std::memset(&gp, 0, sizeof(gp));
  gp.nz = nz; gp.ng = ng; gp.n_ir = 1; gp.ir_lo = 0; gp.ir_start = 0; gp.NQ = ng;
  gp.tau = d_tau.p; gp.w0 = d_w0.p; gp.g = d_g.p; gp.wbin = d_wbin.p; gp.emissivity = d_em.p;
  gp.has_hard_surface = ir_par[1] != 0.0 ? 1 : 0; gp.ir_tau_min = ir_par[2];
  double *w = d_work.p;
  auto take = [&](size_t cnt) { double *p0 = w; w += cnt; return p0; };
  gp.RW = take(7 * RQ); gp.IS = take(6 * LQ); gp.FS = take(FQ); gp.DS = take(8 * LQ); gp.D0 = take(4 * (size_t)ng);
  gp.partial = take((size_t)ndev_pad * 2 * nl);
  gp.DB = d_db.p;
  gp.ndev = ndev; gp.ndev_pad = ndev_pad; gp.qsplit = 1; gp.msplit = 1; gp.partial_m = gp.partial;   // (disjoint pairs: one array)
  gp.dev_k = d_idx.p; gp.nmix = nmix; gp.mix_dev = gp.dev_k + ndev_pad; gp.mix_blk = gp.mix_dev + nmix;
  launch_green_factor(gp, nullptr);
  HIPCHK(hipGetLastError());
  launch_green_accumulate(gp, nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  std::vector<double> out((size_t)ndev_pad * 2 * nl);
  HIPCHK(hipMemcpy(out.data(), gp.partial, sizeof(double) * out.size(), hipMemcpyDeviceToHost));
  for (int d = 0; d < ndev; d++)
    for (int lv = 0; lv < nl; lv++) {
      resp_up[(size_t)order[d] * nl + lv] = out[((size_t)d * 2 + 0) * nl + lv];
      resp_dn[(size_t)order[d] * nl + lv] = out[((size_t)d * 2 + 1) * nl + lv];
    }
  CATCH(err)
}

// which form of the response form's far accumulation the process uses from here on: 0 the matrix-core kernel
// (k_green_accum_far_mfma, the default), 1 the vector kernel (k_green_accum_far; CLIMA_HIP_GREEN_MFMA=0 selects it too)
void clima_test_green_far_form_set(const int *vector_form) { green_vector_form_set(*vector_form); }

// ---- reference-named getters / setters (clima/fortran/Radtran.f90) -------------------

static double bolometric(Radtran *r) {  // Radtran_bolometric_flux, clima_radtran.f90:353-364
  double flux = 0.0;
  for (int i = 0; i < r->sol.nw; i++) flux = flux + r->photons_sol[i] * (r->sol.freq[i] - r->sol.freq[i + 1]);
  return r->photon_scale_factor * flux / 1.0e3;
}
void radtran_set_bolometric_flux_wrapper(void *ptr, const double *flux) {  // :345-350
  Radtran *r = as_rad(ptr);
  if (!r) return;
  r->photon_scale_factor = 1.0;
  r->photon_scale_factor = *flux / bolometric(r);
}
void radtran_bolometric_flux_wrapper(void *ptr, double *flux) {
  Radtran *r = as_rad(ptr);
  if (r) *flux = bolometric(r);
}
static double equilibrium_temperature(double stellar_radiation, double bond_albedo) {  // clima_eqns.f90:248-254
  return std::pow((stellar_radiation * (1.0 - bond_albedo)) / (4.0 * SIGMA_SI), 0.25);
}
void radtran_skin_temperature_wrapper(void *ptr, const double *bond_albedo, double *T_skin) {  // clima_eqns.f90:256-261
  Radtran *r = as_rad(ptr);
  if (r) *T_skin = equilibrium_temperature(bolometric(r), *bond_albedo) * std::pow(0.5, 0.25);
}
void radtran_equilibrium_temperature_wrapper(void *ptr, const double *bond_albedo, double *T_eq) {
  Radtran *r = as_rad(ptr);
  if (r) *T_eq = equilibrium_temperature(bolometric(r), *bond_albedo);
}

#define VEC_GETSET(name, field)                                                                 \
  void radtran_##name##_get_size(void *ptr, int *dim1) {                                        \
    Radtran *r = as_rad(ptr);                                                                   \
    *dim1 = r ? (int)r->field.size() : 0;                                                       \
  }                                                                                             \
  void radtran_##name##_get(void *ptr, const int *dim1, double *arr) {                          \
    Radtran *r = as_rad(ptr);                                                                   \
    if (!r) return;                                                                             \
    for (int i = 0; i < *dim1 && i < (int)r->field.size(); i++) arr[i] = r->field[i];           \
  }                                                                                             \
  void radtran_##name##_set(void *ptr, const int *dim1, const double *arr) {                    \
    Radtran *r = as_rad(ptr);                                                                   \
    if (!r) return;                                                                             \
    /* dirty only when a value changes: the Fortran module pushes every public field before every radiate, \
       and an upload costs a stream synchronise and six copies */                                \
    for (int i = 0; i < *dim1 && i < (int)r->field.size(); i++)                                 \
      if (r->field[i] != arr[i]) { r->field[i] = arr[i]; r->fields_dirty = true; }              \
  }
VEC_GETSET(zenith_u, zenith_u)
VEC_GETSET(surface_albedo, surface_albedo)
VEC_GETSET(surface_emissivity, surface_emissivity)

void radtran_zenith_weights_get(void *ptr, const int *dim1, double *arr) {
  Radtran *r = as_rad(ptr);
  if (!r) return;
  for (int i = 0; i < *dim1 && i < (int)r->zenith_w.size(); i++) arr[i] = r->zenith_w[i];
}
void radtran_zenith_weights_set(void *ptr, const int *dim1, const double *arr) {
  Radtran *r = as_rad(ptr);
  if (!r) return;
  for (int i = 0; i < *dim1 && i < (int)r->zenith_w.size(); i++)
    if (r->zenith_w[i] != arr[i]) { r->zenith_w[i] = arr[i]; r->fields_dirty = true; }
}
// logical(c_bool) in the reference (clima/fortran/Radtran.f90:211-227; Radtran_pxd.pxd:45-46 binds bool*):
// exactly one byte is read or written
static_assert(sizeof(bool) == 1, "logical(c_bool) is one byte");
void radtran_has_hard_surface_get(void *ptr, bool *val) { Radtran *r = as_rad(ptr); if (r) *val = r->has_hard_surface; }
void radtran_has_hard_surface_set(void *ptr, const bool *val) {
  Radtran *r = as_rad(ptr);
  if (r) r->has_hard_surface = (*reinterpret_cast<const unsigned char *>(val) != 0);
}
void radtran_photon_scale_factor_get(void *ptr, double *val) { Radtran *r = as_rad(ptr); if (r) *val = r->photon_scale_factor; }
void radtran_photon_scale_factor_set(void *ptr, const double *val) { Radtran *r = as_rad(ptr); if (r) r->photon_scale_factor = *val; }
void radtran_ir_tau_min_get(void *ptr, double *val) { Radtran *r = as_rad(ptr); if (r) *val = r->ir_tau_min; }
void radtran_ir_tau_min_set(void *ptr, const double *val) { Radtran *r = as_rad(ptr); if (r) r->ir_tau_min = *val; }
void radtran_diurnal_fac_get(void *ptr, double *val) { Radtran *r = as_rad(ptr); if (r) *val = r->diurnal_fac; }
void radtran_diurnal_fac_set(void *ptr, const double *val) { Radtran *r = as_rad(ptr); if (r) r->diurnal_fac = *val; }
void radtran_ir_get(void *ptr, void **ptr1) { Radtran *r = as_rad(ptr); *ptr1 = r ? (void *)&r->ir : nullptr; }
void radtran_sol_get(void *ptr, void **ptr1) { Radtran *r = as_rad(ptr); *ptr1 = r ? (void *)&r->sol : nullptr; }
void radtran_wrk_ir_get(void *ptr, void **ptr1) { Radtran *r = as_rad(ptr); *ptr1 = r ? (void *)&r->wrk_ir : nullptr; }
void radtran_wrk_sol_get(void *ptr, void **ptr1) { Radtran *r = as_rad(ptr); *ptr1 = r ? (void *)&r->wrk_sol : nullptr; }
void radtran_f_total_get_size(void *ptr, int *dim1) { Radtran *r = as_rad(ptr); *dim1 = r ? r->nz + 1 : 0; }
void radtran_f_total_get(void *ptr, const int *dim1, double *arr) {
  Radtran *r = as_rad(ptr);
  if (!r || r->state != 2) return;
  try {
    fetch_small(r);
    for (int i = 0; i < *dim1 && i < r->nz + 1; i++) arr[i] = r->h_small[4 * (r->nz + 1) + i];
  } catch (const HipFail &f) {
    defer_err(r, f.msg);   // no `err` here (clima/fortran/Radtran.f90 getters have none): radtran_synchronize reports it
  } catch (...) {
    defer_err(r, "radtran_f_total_get failed");
  }
}
void radtran_photons_sol_get_size(void *ptr, int *dim1) { Radtran *r = as_rad(ptr); *dim1 = r ? (int)r->photons_sol.size() : 0; }
void radtran_photons_sol_get(void *ptr, const int *dim1, double *arr) {
  Radtran *r = as_rad(ptr);
  if (!r) return;
  for (int i = 0; i < *dim1 && i < (int)r->photons_sol.size(); i++) arr[i] = r->photons_sol[i];
}

// extents of a handle (whatever built it): layers, gases, particles, opacity bins, g-points
void radtran_dims_get(void *ptr, int *nz, int *nsp, int *np, int *nw, int *ngauss) {
  Radtran *r = as_rad(ptr);
  *nz = r ? r->nz : 0; *nsp = r ? r->nsp : 0; *np = r ? r->np : 0; *nw = r ? r->nw : 0; *ngauss = r ? r->ng : 0;
}
// the species / particle names the handle holds (radtran_set_names), newline-separated, into caller buffers of `cap` bytes
void radtran_names_get(void *ptr, const int *cap, char *species_names, char *particle_names) {
  Radtran *r = as_rad(ptr);
  auto put = [&](const std::vector<std::string> &v, char *out) {
    std::string o;
    for (size_t i = 0; i < v.size(); i++) o += (i ? "\n" : "") + v[i];
    std::strncpy(out, o.c_str(), (size_t)std::max(*cap - 1, 0));
    if (*cap > 0) out[*cap - 1] = 0;
  };
  if (r) { put(r->species_names, species_names); put(r->particle_names, particle_names); }
  else if (*cap > 0) { species_names[0] = 0; particle_names[0] = 0; }
}

// Test hook: FNV-1a (64 bit) over the handle's HOST-side tables in the order they were handed over -- metadata as
// int32, values as float64 bytes.  The from-files loader (radtran_loader.hip) is held to clima_amd/data_loader.py with
// it where there is no device: identical tables <=> identical digests (tests/test_loader_cabi.py).
void clima_test_host_tables_digest(void *ptr, unsigned long long *digest) {
  // digest[0..7]: extents + opacity grid | k-tables | CIA | Rayleigh | absorption / photolysis | continuum | particles |
  // channels + stellar photons (each hash starts afresh: a difference names its group)
  Radtran *r = as_rad(ptr);
  unsigned long long h = 0;
  auto start = [&] { h = 1469598103934665603ULL; };
  auto bytes = [&](const void *p0, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p0);
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ULL; }
  };
  auto ints = [&](std::initializer_list<int> v) { for (int x : v) bytes(&x, sizeof(int)); };
  auto vals = [&](const std::vector<double> &v) { if (!v.empty()) bytes(v.data(), v.size() * sizeof(double)); };
  for (int g = 0; g < 8; g++) digest[g] = 0;
  if (!r) return;
  start(); ints({r->nz, r->nsp, r->np, r->nw}); vals(r->wavl); digest[0] = h;
  start();
  for (auto *k : r->k) { ints({k->sp, k->ng, k->nP, k->nT}); vals(k->weights); vals(k->log10P); vals(k->temp); vals(k->log10k); }
  digest[1] = h;
  int g = 2;
  for (auto *lst : {&r->cia, &r->ray, &r->pxs}) {
    start();
    for (auto *x : *lst) { ints({x->type, x->dim, x->sp1, x->sp2, x->nT}); vals(x->temp); vals(x->data); }
    digest[g++] = h;
  }
  start(); ints({r->has_cont ? 1 : 0, r->LH2O, r->cont_nT}); vals(r->cont_temp); vals(r->cont_H2O); vals(r->cont_foreign); digest[5] = h;
  start();
  for (auto *q : r->part) { ints({q->p_ind, q->nrad}); vals(q->radii); vals(q->w0); vals(q->qext); vals(q->gt); }
  digest[6] = h;
  start(); vals(r->ir.wavl); vals(r->sol.wavl); vals(r->photons_sol); digest[7] = h;
}

// ---- all per-bin spectra of the last call in one go ------------------------------------
// The reference's result holder is plain allocatables the caller reads after every call (clima_radtran.f90:11-25); a
// host that does the same through the seven reference-named getters pays seven synchronous copies to pageable memory
// (6.7 MB at ~12 GB/s: 530 of a 648 us call at config 2).  Here the caller's seven arrays are page-locked on first
// use (hipHostRegister; they stay so until the handle is destroyed or radtran_spectra_release) and filled by seven
// asynchronous copies on the handle's stream and ONE synchronise.  An array that cannot be registered is still
// filled (staged by the runtime, slower).  do_solar false: the IR channel's three arrays only.
static void register_host(Radtran *r, void *p, size_t bytes) {
  if (!p || !bytes) return;
  for (auto &e : r->host_registered)
    if (e.first == p && e.second >= bytes) return;
  for (auto it = r->host_registered.begin(); it != r->host_registered.end(); ++it)
    if (it->first == p) { (void)hipHostUnregister(p); r->host_registered.erase(it); break; }   // (re-allocated larger at the same address)
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) r->host_registered.emplace_back(p, bytes);
  else (void)hipGetLastError();
}
static bool host_is_registered(const Radtran *r, const void *p, size_t bytes) {
  for (auto &e : r->host_registered)
    if (e.first == p && e.second >= bytes) return true;
  return false;
}
void radtran_spectra_get_all(void *ptr, const bool *do_solar, const int *nlev, const int *nw_ir, const int *nw_sol,
                             double *ir_fup_a, double *ir_fdn_a, double *ir_tau_band,
                             double *sol_fup_a, double *sol_fdn_a, double *sol_amean, double *sol_tau_band, char *err) {
  err[0] = 0;
  GUARD(r, ptr, err)
  TRY
  if (r->state != 2) throw HipFail{"radtran_spectra_get_all: the handle is not constructed"};
  if (*nlev != r->nz + 1 || *nw_ir != r->ir.nw || *nw_sol != r->sol.nw)
    throw HipFail{"radtran_spectra_get_all: array extents do not match the handle (nz+1, ir%nw, sol%nw)"};
  if (!r->small_valid) settle(r);   // (after a synchronous call the results are final already)
  struct Job { double *dst; DevBuf<double> *src; };
  Job jobs[7] = {{ir_fup_a, &r->wrk_ir.fup_a}, {ir_fdn_a, &r->wrk_ir.fdn_a}, {ir_tau_band, &r->wrk_ir.tau_band},
                 {sol_fup_a, &r->wrk_sol.fup_a}, {sol_fdn_a, &r->wrk_sol.fdn_a}, {sol_amean, &r->wrk_sol.amean},
                 {sol_tau_band, &r->wrk_sol.tau_band}};
  const int n = *do_solar ? 7 : 3;
  for (int i = 0; i < n; i++) register_host(r, jobs[i].dst, jobs[i].src->n * sizeof(double));
  // the results are final (settled above): the copies need no ordering among themselves and go out over four
  // queues -- one copy engine sustains ~34 GB/s of the link, several of them more
  HIPCHK(hipStreamSynchronize(r->stream));
  for (auto &cs : r->copy_streams)
    if (!cs) HIPCHK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipStream_t qs[4] = {r->stream, r->copy_streams[0], r->copy_streams[1], r->copy_streams[2]};
  const int order[7] = {3, 4, 5, 0, 1, 6, 2};   // the four large solar / IR arrays first, one per queue
  int q = 0;
  for (int k = 0; k < 7; k++) {
    const int i = order[k];
    if (i >= n || !jobs[i].dst || !jobs[i].src->n) continue;
    HIPCHK(hipMemcpyAsync(jobs[i].dst, jobs[i].src->p, jobs[i].src->n * sizeof(double), hipMemcpyDeviceToHost, qs[q & 3]));
    q++;
  }
  for (int k = 0; k < 4; k++) HIPCHK(hipStreamSynchronize(qs[k]));
  CATCH(err)
}
void radtran_batch_pin_results_set(void *ptr, const int *flag) {
  Radtran *r = as_rad(ptr);
  if (r && r->batch_pin_results != (*flag != 0)) { r->batch_pin_results = *flag != 0; r->batch_out_n = 0; }
}
void radtran_batch_pin_results_get(void *ptr, int *flag) {
  Radtran *r = as_rad(ptr);
  *flag = r && r->batch_pin_results ? 1 : 0;
}
void radtran_spectra_release(void *ptr) {
  Radtran *r = as_rad(ptr);
  if (!r) return;
  for (auto &e : r->host_registered) (void)hipHostUnregister(e.first);
  r->host_registered.clear();
  r->batch_out_n = 0;
}

// ---- ClimaRadtranWrk (clima/fortran/ClimaRadtranWrk.f90) ------------------------------
static int ch_nw(WrkObj *w) { return w->which ? w->parent->sol.nw : w->parent->ir.nw; }
#define WRK2D(name, field, rows)                                                          \
  void climaradtranwrk_##name##_get_size(void *ptr, int *dim1, int *dim2) {               \
    WrkObj *w = reinterpret_cast<WrkObj *>(ptr);                                          \
    *dim1 = w->parent->nz + (rows);                                                       \
    *dim2 = ch_nw(w);                                                                     \
  }                                                                                       \
  void climaradtranwrk_##name##_get(void *ptr, const int *dim1, const int *dim2, double *arr) { \
    WrkObj *w = reinterpret_cast<WrkObj *>(ptr);                                          \
    try { get2d(w, w->field, *dim1, *dim2, arr); }                                        \
    catch (const HipFail &f) { defer_err(w->parent, f.msg); }                             \
    catch (...) { defer_err(w->parent, "climaradtranwrk getter failed"); }                \
  }
WRK2D(fup_a, fup_a, 1)
WRK2D(fdn_a, fdn_a, 1)
WRK2D(amean, amean, 1)
WRK2D(tau_band, tau_band, 0)
#define WRK1D(name, up)                                                                   \
  void climaradtranwrk_##name##_get_size(void *ptr, int *dim1) {                          \
    WrkObj *w = reinterpret_cast<WrkObj *>(ptr);                                          \
    *dim1 = w->parent->nz + 1;                                                            \
  }                                                                                       \
  void climaradtranwrk_##name##_get(void *ptr, const int *dim1, double *arr) {            \
    WrkObj *w = reinterpret_cast<WrkObj *>(ptr);                                          \
    Radtran *r = w->parent;                                                               \
    try {                                                                                 \
      fetch_small(r);                                                                     \
      const int nl = r->nz + 1;                                                           \
      const int a = (w->which ? 2 : 0) + ((up) ? 0 : 1);                                  \
      for (int i = 0; i < *dim1 && i < nl; i++) arr[i] = r->h_small[a * nl + i];          \
    } catch (const HipFail &f) { defer_err(r, f.msg);                                     \
    } catch (...) { defer_err(r, "climaradtranwrk getter failed"); }                      \
  }
WRK1D(fup_n, 1)
WRK1D(fdn_n, 0)

// ---- RTChannel (clima/fortran/RTChannel.f90) ------------------------------------------
void rtchannel_wavl_get_size(void *ptr, int *dim1) { *dim1 = (int)reinterpret_cast<ChannelObj *>(ptr)->wavl.size(); }
void rtchannel_wavl_get(void *ptr, const int *dim1, double *arr) {
  ChannelObj *c = reinterpret_cast<ChannelObj *>(ptr);
  for (int i = 0; i < *dim1 && i < (int)c->wavl.size(); i++) arr[i] = c->wavl[i];
}
void rtchannel_freq_get_size(void *ptr, int *dim1) { *dim1 = (int)reinterpret_cast<ChannelObj *>(ptr)->freq.size(); }
void rtchannel_freq_get(void *ptr, const int *dim1, double *arr) {
  ChannelObj *c = reinterpret_cast<ChannelObj *>(ptr);
  for (int i = 0; i < *dim1 && i < (int)c->freq.size(); i++) arr[i] = c->freq[i];
}

}  // extern "C"
