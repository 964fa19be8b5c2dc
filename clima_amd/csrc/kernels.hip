// kernels.hip -- hand-written gfx950 (CDNA4) kernels for Clima's radiate() hot path.
//
//   k_prep        per-column pre-pass shared by every bin: log10P, columns, pair_reuse,
//                 interpolation brackets/weights (compute_opacity pre-pass,
//                 src/radtran/clima_radtran_types.f90:599-633 + dintrv bracketing)
//   k_opacity     one LANE per (bin, layer): k-table / CIA / continuum / Rayleigh / Mie
//                 gather + random-overlap resort-rebin (types.f90:640-888).  The 64-key
//                 resort is a register-resident Batcher network (v_min_f64/v_max_f64).
//   k_twostream   one WORKGROUP per (channel, bin): Toon two-stream, layer-parallel
//                 coefficient assembly staged in LDS, Thomas solve per g-point column,
//                 g-/zenith-weighted level fluxes (radiate.f90:50-158, twostream.f90)
//   k_integrate   spectral integration + f_total (radiate.f90:184-192, clima_radtran.f90:316)
//
// All arithmetic is IEEE binary64.  No MFMA: there is no dense contraction on this path.
#include "radtran_dev.h"

namespace clima {

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------

// ten2power, src/clima_eqns.f90:75-80
__device__ __forceinline__ double ten2power(double y) { return exp(y * LN10); }

// planck_fcn, src/clima_eqns.f90:64-73
__device__ __forceinline__ double planck_fcn(double nu, double T) {
  return 1.0e3 * ((2.0 * PLANK * (nu * nu * nu)) / (C_LIGHT * C_LIGHT)) *
         ((1.0) / (exp((PLANK * nu) / (K_BOLTZ_SI * T)) - 1.0));
}

// futils is_close (fortran-stdlib form): |a-b| <= tol*max(|a|,|b|)
__device__ __forceinline__ bool is_close(double a, double b, double tol) {
  return fabs(a - b) <= fabs(tol * fmax(fabs(a), fabs(b)));
}

// dintrv bracketing, linear_interpolation_module.F90:348-350 (stateless form)
__device__ __forceinline__ int bracket(const double *xt, int n, double x) {
  if (x < xt[0]) return 0;
  if (x >= xt[n - 1]) return n - 2;
  int lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (x < xt[mid]) hi = mid; else lo = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------
// k_prep: one block, threads stride over layers
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(PrepParams p) {
  const int nz = p.nz;
  const ColumnDev &c = p.col;
  for (int j = threadIdx.x; j < nz; j += blockDim.x) {
    c.log10P[j] = log10(c.P[j]);  // types.f90:605
    double fc = 0.0;
    for (int i = 0; i < p.nsp; i++) {  // :607-619
      double col = c.dens[i * nz + j] * c.dz[j];
      c.cols[i * nz + j] = col;
      if (p.has_cont && i != p.LH2O) fc = fc + col;
    }
    c.foreign_col[j] = fc;
  }
  __syncthreads();
  // pair_reuse (:621-632): even nz only, second layer of each pair
  for (int j = threadIdx.x; j < nz; j += blockDim.x) {
    int src = j;
    if ((nz & 1) == 0 && (j & 1) == 1) {
      const double tol = 1.0e-12;
      bool ok = is_close(c.P[j], c.P[j - 1], tol) && is_close(c.T[j], c.T[j - 1], tol);
      for (int i = 0; i < p.nsp; i++) ok = ok && is_close(c.cols[i * nz + j], c.cols[i * nz + j - 1], tol);
      if (p.check_radii)
        for (int i = 0; i < p.np; i++) ok = ok && is_close(c.radii[i * nz + j], c.radii[i * nz + j - 1], tol);
      if (ok) src = j - 1;
    }
    c.src[j] = src;
  }
  __syncthreads();
  // interpolation brackets and weights per (slot, layer); reuse layers take their
  // source layer's inputs, which is what copying its interpolated value amounts to
  // (:652-653, :907-908, :933-935, :963-968)
  for (int idx = threadIdx.x; idx < p.nslots * nz; idx += blockDim.x) {
    const int s = idx / nz, j = idx - s * nz;
    const SlotDev &sl = p.slots[s];
    const int js = c.src[j];
    double x;
    if (sl.source == 0) x = c.log10P[js];
    else if (sl.source == 1) x = c.T[js];
    else x = c.radii[(sl.source - 2) * nz + js];
    if (sl.flag_clamp && (x < sl.lo || x > sl.hi)) atomicOr(c.err_flag, 1);
    x = fmin(fmax(x, sl.lo), sl.hi);  // :655-656, :910, :937, :974
    const int i = bracket(sl.axis, sl.n, x);
    c.ix[idx] = i;
    c.q[idx] = (x - sl.axis[i]) / (sl.axis[i + 1] - sl.axis[i]);  // linear_interpolation_module.F90:256, :319-320
  }
}

void launch_prep(const PrepParams &p, hipStream_t s) { hipLaunchKernelGGL(k_prep, dim3(1), dim3(256), 0, s, p); }

// ------------------------------------------------------------------------------------
// k_opacity
// ------------------------------------------------------------------------------------

// linear_interp_1d%evaluate (linear_interpolation_module.F90:256-259)
__device__ __forceinline__ double lerp1(const double *f, int i, double q) {
  const double p1 = 1.0 - q;
  return p1 * f[i] + q * f[i + 1];
}

constexpr int OP_THREADS = 256;

// Random-overlap resort + rebin for NG = 8 (k_rorr, types.f90:826-852), one lane per
// (bin, layer).  X = current mixture tau_k(8), Y = new species' k*col (8), both in the
// lane's private LDS slots sm[slot][tid].  The 64 sums X_i+Y_j are sorted by a Batcher
// network held in registers; each key carries its pair index in the 6 low mantissa bits
// (ordering ties exactly like a stable sort on (value,index) whenever the values differ
// above 2^-46 relative), and the exact value is re-formed from X,Y when consumed.  The
// sorted stream is deposited straight into the ng output bins (weights_to_bins + futils
// rebin, types.f90:846-847) and written to sm[out+k][tid].
__device__ __forceinline__ void rorr_mix8(double (*sm)[OP_THREADS], const int tid, const int xo,
                                          const int yo, const int oo, const double *s_wxy,
                                          const double *s_E) {
  double key[64];
  {
    double y[8];
#pragma unroll
    for (int j = 0; j < 8; j++) y[j] = sm[yo + j][tid];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const double xi = sm[xo + i][tid];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const double v = xi + y[j];  // tau_xy(:, j+(i-1)*ng), types.f90:828
        unsigned long long b = (unsigned long long)__double_as_longlong(v);
        b = (b & ~63ULL) | (unsigned long long)(i * 8 + j);
        key[i * 8 + j] = __longlong_as_double((long long)b);
      }
    }
  }
#define CE(a, b)                                     \
  {                                                  \
    const double lo_ = __builtin_fmin(key[a], key[b]); \
    const double hi_ = __builtin_fmax(key[a], key[b]); \
    key[a] = lo_;                                    \
    key[b] = hi_;                                    \
  }
#define CE_FULL_HEAD
#define CE_MERGE_TAIL
#include "sort_network_64.inc"
#undef CE_FULL_HEAD
#undef CE_MERGE_TAIL
#undef CE
  int k = 0;
  double b0 = s_E[0], b1 = s_E[1];
  double acc = 0.0, c0 = 0.0;
#pragma unroll
  for (int p = 0; p < 64; p++) {
    const int idx = (int)((unsigned long long)__double_as_longlong(key[p]) & 63ULL);
    const double v = sm[xo + (idx >> 3)][tid] + sm[yo + (idx & 7)][tid];
    const double c1 = c0 + s_wxy[idx];  // weights_to_bins (clima_eqns.f90:43-54) on wxy(inds)
    for (;;) {
      const double lo = fmax(c0, b0), hi = fmin(c1, b1);
      if (hi > lo) acc = acc + (hi - lo) * v;
      if (c1 > b1 && k < 7) {
        sm[oo + k][tid] = acc / (b1 - b0);
        k++;
        b0 = b1;
        b1 = s_E[k + 1];
        acc = 0.0;
      } else {
        break;
      }
    }
    c0 = c1;
  }
  sm[oo + k][tid] = acc / (b1 - b0);
  for (k = k + 1; k < 8; k++) sm[oo + k][tid] = 0.0;
}

__global__ __launch_bounds__(OP_THREADS) void k_opacity8(OpacityParams p) {
  constexpr int NG = 8;
  __shared__ double sm[3 * NG][OP_THREADS];  // per-lane private slots: X, Y, out
  __shared__ double s_wxy[NG * NG];
  __shared__ double s_E[NG + 1];
  __shared__ double s_wbin[NG];
  const int tid = threadIdx.x;
  if (tid < NG * NG) s_wxy[tid] = p.wxy[tid];
  if (tid < NG + 1) s_E[tid] = p.wbin_e[tid];
  if (tid < NG) s_wbin[tid] = p.wbin[tid];
  __syncthreads();

  const int nz = p.nz;
  const long total = (long)p.nbins * nz;
  long t = (long)blockIdx.x * OP_THREADS + tid;
  const bool valid = t < total;
  if (!valid) t = total - 1;
  const int l = p.bin_lo + (int)(t / nz);
  const int j = (int)(t % nz);  // ground-first layer
  const int n = nz - 1 - j;     // TOA-first index (types.f90:690-691, :862-865)
  const ColumnDev &c = p.col;
  const bool reuse = c.src[j] != j;
  const double dzj = c.dz[j];

  // ---- Rayleigh (:686-693)
  double tausg = 0.0;
  for (int i = 0; i < p.nray; i++) tausg = tausg + p.ray[i].data[l] * c.cols[p.ray[i].sp1 * nz + j];
  // ---- CIA (:665-667, :696-704)
  double taua = 0.0;
  for (int i = 0; i < p.ncia; i++) {
    const XsDev &x = p.cia[i];
    double s;
    if (x.dim == 0) s = x.data[l];
    else s = ten2power(lerp1(x.data + (size_t)l * x.nT, c.ix[x.slot * nz + j], c.q[x.slot * nz + j]));
    taua = taua + s * c.dens[x.sp1 * nz + j] * c.dens[x.sp2 * nz + j] * dzj;
  }
  // ---- photolysis / absorption (:670-672, :707-713)
  for (int i = 0; i < p.npxs; i++) {
    const XsDev &x = p.pxs[i];
    double s;
    if (x.dim == 0) s = x.data[l];
    else s = ten2power(lerp1(x.data + (size_t)l * x.nT, c.ix[x.slot * nz + j], c.q[x.slot * nz + j]));
    taua = taua + s * c.cols[x.sp1 * nz + j];
  }
  // ---- water continuum (:675-677, :716-723)
  if (p.has_cont) {
    const int ix = c.ix[p.cont_slot * nz + j];
    const double q = c.q[p.cont_slot * nz + j];
    const double h2o = ten2power(lerp1(p.cont_H2O + (size_t)l * p.cont_nT, ix, q));
    const double frn = ten2power(lerp1(p.cont_foreign + (size_t)l * p.cont_nT, ix, q));
    const double dL = c.dens[p.LH2O * nz + j];
    taua = taua + h2o * dL * c.cols[p.LH2O * nz + j] + frn * dL * c.foreign_col[j];
  }
  // ---- custom opacity unset (:558-562, :726-730): tiny everywhere
  const double tauc = TINY, tausc = TINY * TINY;
  // ---- particles (:680-683, :733-757)
  double tausp = 0.0, taup = 0.0;
  double tausp_1[MAX_PART], gtp[MAX_PART];
  for (int i = 0; i < p.npart; i++) {
    const PartDev &pt = p.part[i];
    const int ix = c.ix[pt.slot * nz + j];
    const double q = c.q[pt.slot * nz + j];
    const double w0p = lerp1(pt.w0 + (size_t)l * pt.nrad, ix, q);
    const double qext = lerp1(pt.qext + (size_t)l * pt.nrad, ix, q);
    gtp[i] = lerp1(pt.gt + (size_t)l * pt.nrad, ix, q);
    const double rr = c.radii[pt.p_ind * nz + j];
    const double taup_1 = qext * PI * (rr * rr) * c.pdens[pt.p_ind * nz + j] * dzj;
    taup = taup + taup_1;
    tausp_1[i] = w0p * taup_1;
    tausp = tausp + tausp_1[i];
  }
  double gt = 0.0;
  for (int i = 0; i < p.npart; i++) gt = gt + gtp[i] * tausp_1[i] / fmax(TAU_MIN, (tausp + tausg + tausc));
  gt = gt + TINY * tausc / fmax(TAU_MIN, (tausp + tausg + tausc));
  gt = fmin(gt, MAX_GT);

  // ---- k-distributions (:649-662) and random-overlap mixing (k_rorr :816-854)
  int xo = 0, oo = 2 * NG;
  for (int s = 0; s < p.nk; s++) {
    const KDev &kd = p.k[s];
    const int iP = c.ix[kd.slotP * nz + j], iT = c.ix[kd.slotT * nz + j];
    const double q1 = c.q[kd.slotP * nz + j], q2 = c.q[kd.slotT * nz + j];
    const double p1 = 1.0 - q1, p2 = 1.0 - q2;
    const double *slab = kd.log10k + (size_t)l * kd.nT * kd.nP * NG;
    const double *f11 = slab + ((size_t)iT * kd.nP + iP) * NG;
    const double *f21 = f11 + NG;                    // iP+1
    const double *f12 = f11 + (size_t)kd.nP * NG;    // iT+1
    const double *f22 = f12 + NG;
    const double col = c.cols[kd.sp * nz + j];
    const int dst = (s == 0) ? xo : NG;
#pragma unroll
    for (int g = 0; g < NG; g++) {
      // linear_interp_2d%evaluate, linear_interpolation_module.F90:319-327
      const double fx1 = p1 * f11[g] + q1 * f21[g];
      const double fx2 = p1 * f12[g] + q1 * f22[g];
      const double kk = ten2power(p2 * fx1 + q2 * fx2);
      sm[dst + g][tid] = kk * col;  // :818 / :828
    }
    if (s > 0) {
      rorr_mix8(sm, tid, xo, NG, oo, s_wxy, s_E);
      // pair_reuse: the second layer of a pair copies the first layer's rebinned
      // mixture (:833-834).  Layer j-1 of the same bin lives in lane-1 (nz even).
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const double mine = sm[oo + g][tid];
        const double prev = __shfl_up(mine, 1);
        if (reuse) sm[oo + g][tid] = prev;
      }
      const int tmp = xo;
      xo = oo;
      oo = tmp;
    }
  }

  // ---- totals (:856-886)
  if (valid) {
    double tb = 0.0;
    const size_t base = ((size_t)l * NG) * nz + n;
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const double tau = tausg + taua + taup + sm[xo + g][tid] + tauc;
      double w0;
      if (tau <= TAU_MIN) w0 = 0.0;
      else w0 = fmin(MAX_W0, (tausg + tausp + tausc) / tau);
      p.tau[base + (size_t)g * nz] = tau;
      p.w0[base + (size_t)g * nz] = w0;
      tb = tb + tau * s_wbin[g];
    }
    p.tau_band[(size_t)l * nz + n] = tb;
    p.g[(size_t)l * nz + n] = gt;
  }
}

bool launch_opacity(const OpacityParams &p, hipStream_t s) {
  if (p.ng != 8) return false;
  const long total = (long)p.nbins * p.nz;
  if (total <= 0) return true;
  const int grid = (int)((total + OP_THREADS - 1) / OP_THREADS);
  hipLaunchKernelGGL(k_opacity8, dim3(grid), dim3(OP_THREADS), 0, s, p);
  return true;
}

// ------------------------------------------------------------------------------------
// k_twostream
// ------------------------------------------------------------------------------------
//
// LDS image per block (bin): 4 arrays [nz][ng] f64 (pair index p = i*ng + c, layer i
// TOA-first, g-point column c) that are recycled through the phases:
//   G : cap_gamma      -> c' of row 2i+1      -> sum-over-g staging (fup)
//   X : exp(-lambda t) -> c' of row 2i+2      -> staging (fdn)
//   A : tau'/tauc, cp0 -> E of row 2i+1 -> d' -> Y(2i+1)=y2_i -> staging (amean)
//   B : cm0            -> E of row 2i+2 -> d' -> Y(2i+2)=y1_{i+1}
// Rows 0 and 2nz-1 live in layer nz-1's otherwise unused A/B slots.
// Tridiagonal layout: SURVEY.md Appendix A / twostream.f90:91-117, :249-275.

constexpr int TS_MAXP = 4;  // (layer, g) pairs per thread

struct E4 {
  double e1, e2, e3, e4;
};
// e's, twostream.f90:55-61 / :205-211
__device__ __forceinline__ E4 make_e(double G, double x) {
  E4 e;
  e.e1 = 1.0 + G * x;
  e.e2 = 1.0 - G * x;
  e.e3 = G + x;
  e.e4 = G - x;
  return e;
}

// Thomas algorithm (tridiag, twostream.f90:297-316) for one column; rows are formed on
// the fly from (G,X) of adjacent layers, inputs/outputs through the LDS image.
__device__ void thomas_column(double *sG, double *sX, double *sA, double *sB, const int nz,
                              const int ng, const int c, const double Rsfc) {
  E4 e = make_e(sG[c], sX[c]);
  // row 0 (:93-96): B=e1, D=-e2, E=-cm0(1) (staged in B[nz-1])
  double cp = (-e.e2) / e.e1;
  double dp = sB[(nz - 1) * ng + c] / e.e1;
  const double c0p = cp, d0p = dp;
  double Gn = 0.0, xn = 0.0, Ea = 0.0, Eb = 0.0;
  if (nz > 1) {
    Gn = sG[ng + c];
    xn = sX[ng + c];
    Ea = sA[c];
    Eb = sB[c];
  }
  for (int i = 0; i < nz - 1; i++) {
    const E4 f = make_e(Gn, xn);
    const double Ea_i = Ea, Eb_i = Eb;
    if (i + 1 < nz - 1) {  // prefetch next iteration's operands
      Gn = sG[(i + 2) * ng + c];
      xn = sX[(i + 2) * ng + c];
      Ea = sA[(i + 1) * ng + c];
      Eb = sB[(i + 1) * ng + c];
    }
    // row 2i+1 (Fortran even rows l=2i, :106-112)
    double A = f.e2 * e.e1 - e.e3 * f.e4;
    double B = e.e2 * f.e2 - e.e4 * f.e4;
    double D = f.e1 * f.e4 - f.e2 * f.e3;
    double den = B - A * cp;
    double cn = D / den;
    double dn = (Ea_i - A * dp) / den;
    sG[i * ng + c] = cn;
    sA[i * ng + c] = dn;
    cp = cn;
    dp = dn;
    // row 2i+2 (Fortran odd rows l=2i+1, :97-103)
    A = e.e2 * e.e3 - e.e4 * e.e1;
    B = e.e1 * f.e1 - e.e3 * f.e3;
    D = e.e3 * f.e4 - e.e1 * f.e2;
    den = B - A * cp;
    cn = D / den;
    dn = (Eb_i - A * dp) / den;
    sX[i * ng + c] = cn;
    sB[i * ng + c] = dn;
    cp = cn;
    dp = dn;
    e = f;
  }
  // last row (:113-117), E staged in A[nz-1]
  {
    const double A = e.e1 - Rsfc * e.e3;
    const double B = e.e2 - Rsfc * e.e4;
    const double ylast = (sA[(nz - 1) * ng + c] - A * dp) / (B - A * cp);
    sA[(nz - 1) * ng + c] = ylast;  // y2_{nz-1}
    double ynext = ylast;
    for (int i = nz - 2; i >= 0; i--) {  // back substitution (:313-315)
      const double yb = sB[i * ng + c] - sX[i * ng + c] * ynext;
      sB[i * ng + c] = yb;  // y1_{i+1}
      const double ya = sA[i * ng + c] - sG[i * ng + c] * yb;
      sA[i * ng + c] = ya;  // y2_i
      ynext = ya;
    }
    sB[(nz - 1) * ng + c] = d0p - c0p * ynext;  // y1_0
  }
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_twostream(TwoStreamParams p) {
  extern __shared__ __align__(16) double lds[];
  const int nz = p.nz, ng = p.ng;
  const int npairs = nz * ng;
  double *sG = lds, *sX = lds + npairs, *sA = lds + 2 * npairs, *sB = lds + 3 * npairs;
  double *sBp = lds + 4 * npairs;       // [nz+1] Planck (IR) / scratch
  double *sL0 = sBp + (nz + 1);         // [3][ng] level-0 values per column
  const int tid = threadIdx.x, nt = blockDim.x;
  const bool solar = (int)blockIdx.x < p.n_sol;
  const int ll = solar ? p.sol_lo + (int)blockIdx.x : p.ir_lo + ((int)blockIdx.x - p.n_sol);
  const int l = (solar ? p.sol_start : p.ir_start) + ll;  // opacity bin (radiate.f90:57)
  const double *tauL = p.tau + (size_t)l * ng * nz;
  const double *w0L = p.w0 + (size_t)l * ng * nz;
  const double *gL = p.g + (size_t)l * nz;

  // values each thread keeps for its pairs across the solve
  double kG[TS_MAXP], kx[TS_MAXP], kcpb[TS_MAXP], kcmb[TS_MAXP], kdir[TS_MAXP], kdiru[TS_MAXP];
  double cp0_first = 0.0;  // cp0 of layer 0 (pair k=0 of threads tid<ng)
  double Rsfc;

  if (solar) {
    // =========================== solar (two_stream_solar, twostream.f90:10-154) ======
    Rsfc = p.albedo[ll];
    double ktau[TS_MAXP], kw0[TS_MAXP], kgt[TS_MAXP], kg1[TS_MAXP], kg2[TS_MAXP], klam[TS_MAXP];
    const double sqrt3 = 1.7320508075688772;  // sqrt(3.0_dp)
#pragma unroll
    for (int k = 0; k < TS_MAXP; k++) {
      const int pr = tid + k * nt;
      if (pr < npairs) {
        const int i = pr / ng, c = pr - i * ng;
        const double tau_in = tauL[c * nz + i], w0_in = w0L[c * nz + i], gt_in = gL[i];
        // delta-Eddington (:38-40)
        const double tau = tau_in * (1.0 - w0_in * gt_in * gt_in);
        const double w0 = w0_in * (1.0 - gt_in * gt_in) / (1.0 - w0_in * gt_in * gt_in);
        const double gt = gt_in / (1.0 + gt_in);
        // quadrature coefficients (:43-44), lambda, Gamma (:50-51), exp (:56)
        const double gam1 = sqrt3 * (2.0 - w0 * (1 + gt)) / 2.0;
        const double gam2 = sqrt3 * w0 * (1.0 - gt) / 2.0;
        const double lam = sqrt(gam1 * gam1 - gam2 * gam2);
        const double G = gam2 / (gam1 + lam);
        const double x = exp(-lam * tau);
        ktau[k] = tau; kw0[k] = w0; kgt[k] = gt; kg1[k] = gam1; kg2[k] = gam2; klam[k] = lam;
        kG[k] = G; kx[k] = x;
        sA[pr] = tau;
      }
    }
    __syncthreads();
    // tauc (:64-67): cumulative optical depth at the top of each layer, sequential order
    if (tid < ng) {
      double cum = 0.0;
      for (int i = 0; i < nz; i++) {
        const double t = sA[i * ng + tid];
        sA[i * ng + tid] = cum;
        cum = cum + t;
      }
    }
    __syncthreads();
    // C+/C- and direct beam (:73-87), summed over zenith angles with their weights: the
    // system matrix does not depend on u0, so sum_z w_z * solve(E_z) == solve(sum_z w_z E_z)
    // (radiate.f90:83-136 applies the same weights to the solved fluxes).
    double wsum = 0.0, dir0 = 0.0;
    for (int z = 0; z < p.nzen; z++) { wsum = wsum + p.zen_w[z]; dir0 = dir0 + p.zen_w[z] * p.zen_u[z]; }
#pragma unroll
    for (int k = 0; k < TS_MAXP; k++) {
      const int pr = tid + k * nt;
      if (pr < npairs) {
        const double tauc = sA[pr];
        double CP0 = 0.0, CPB = 0.0, CM0 = 0.0, CMB = 0.0, DIR = 0.0, DIRU = 0.0;
        for (int z = 0; z < p.nzen; z++) {
          const double u0 = p.zen_u[z], wz = p.zen_w[z];
          const double gam3 = (1.0 - sqrt3 * kgt[k] * u0) / 2.0;
          const double gam4 = 1.0 - gam3;
          const double facp = kw0[k] * ((kg1[k] - 1.0 / u0) * gam3 + gam4 * kg2[k]);
          const double facm = kw0[k] * ((kg1[k] + 1.0 / u0) * gam4 + kg2[k] * gam3);
          const double et0 = exp(-tauc / u0);
          const double etb = et0 * exp(-ktau[k] / u0);
          const double denom = klam[k] * klam[k] - 1.0 / (u0 * u0);
          const double direct = u0 * etb;
          CP0 = CP0 + wz * (et0 * facp / denom);
          CPB = CPB + wz * (etb * facp / denom);
          CM0 = CM0 + wz * (et0 * facm / denom);
          CMB = CMB + wz * (etb * facm / denom);
          DIR = DIR + wz * direct;
          DIRU = DIRU + wz * (direct / u0);
        }
        kcpb[k] = CPB; kcmb[k] = CMB; kdir[k] = DIR; kdiru[k] = DIRU;
        sG[pr] = kG[k];
        sX[pr] = kx[k];
        sA[pr] = CP0;
        sB[pr] = CM0;
        if (pr < ng) cp0_first = CP0;
      }
    }
    if (tid < ng) {  // level-0 direct terms: direct(1)=u0 (:73), direct(1)/u0 = 1
      sL0[1 * ng + tid] = dir0;
      sL0[2 * ng + tid] = wsum;
    }
  } else {
    // =========================== IR (two_stream_ir, twostream.f90:156-295) ===========
    const double emis = p.emissivity[ll];
    Rsfc = p.has_hard_surface ? 1.0 - emis : 0.0;  // :186-190
    const double avg_freq = 0.5 * (p.freq[l] + p.freq[l + 1]);  // radiate.f90:64
    for (int n = tid; n < nz + 1; n += nt)                        // radiate.f90:65-69
      sBp[n] = planck_fcn(avg_freq, n == nz ? *p.T_surface : p.T[nz - 1 - n]);
    __syncthreads();
    const double norm = 2.0 * PI * 0.5;
#pragma unroll
    for (int k = 0; k < TS_MAXP; k++) {
      const int pr = tid + k * nt;
      if (pr < npairs) {
        const int i = pr / ng, c = pr - i * ng;
        const double tau = tauL[c * nz + i], w0 = w0L[c * nz + i], gt = gL[i];
        const double gam1 = 2.0 - w0 * (1.0 + gt);  // :195-201
        const double gam2 = w0 * (1.0 - gt);
        const double lam = sqrt(gam1 * gam1 - gam2 * gam2);
        const double G = gam2 / (gam1 + lam);
        const double x = exp(-lam * tau);
        double b0n, b1n;  // :216-227
        if (tau <= p.ir_tau_min) {
          b0n = 0.5 * (sBp[i] + sBp[i + 1]);
          b1n = 0.0;
        } else {
          b0n = sBp[i];
          b1n = (sBp[i + 1] - b0n) / tau;
        }
        const double r = 1.0 / (gam1 + gam2);
        const double cp0 = norm * (b0n + b1n * (r));  // :229-232
        const double cpb = norm * (b0n + b1n * (tau + r));
        const double cm0 = norm * (b0n + b1n * (-r));
        const double cmb = norm * (b0n + b1n * (tau - r));
        kG[k] = G; kx[k] = x; kcpb[k] = cpb; kcmb[k] = cmb; kdir[k] = 0.0; kdiru[k] = 0.0;
        sG[pr] = G;
        sX[pr] = x;
        sA[pr] = cp0;
        sB[pr] = cm0;
        if (pr < ng) cp0_first = cp0;
      }
    }
    if (tid < ng) {
      sL0[1 * ng + tid] = 0.0;  // fdn(1) = 0 (:289)
      sL0[2 * ng + tid] = 0.0;
    }
  }
  __syncthreads();

  // ---- right-hand sides (E of :102, :111, :96, :117) from neighbouring layers
  double EA[TS_MAXP], EB[TS_MAXP];
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    EA[k] = 0.0; EB[k] = 0.0;
    if (pr < npairs) {
      const int i = pr / ng, c = pr - i * ng;
      const E4 e = make_e(kG[k], kx[k]);
      if (i < nz - 1) {
        const E4 f = make_e(sG[pr + ng], sX[pr + ng]);
        const double cp0n = sA[pr + ng], cm0n = sB[pr + ng];
        EA[k] = f.e2 * (cp0n - kcpb[k]) - f.e4 * (cm0n - kcmb[k]);  // row 2i+1 (:111)
        EB[k] = e.e3 * (cp0n - kcpb[k]) + e.e1 * (kcmb[k] - cm0n);  // row 2i+2 (:102)
      } else {
        double Ssfc;
        if (solar) {
          Ssfc = Rsfc * kdir[k];  // :89 (zenith-weighted direct beam at the ground)
        } else if (p.has_hard_surface) {
          Ssfc = p.emissivity[ll] * PI * sBp[nz];  // :237
        } else {  // :241-246
          const double tau = tauL[c * nz + i];
          const double b1_bot = (tau <= p.ir_tau_min) ? 0.0 : (sBp[nz] - sBp[nz - 1]) / tau;
          Ssfc = PI * (sBp[nz] + 0.5 * b1_bot);
        }
        EA[k] = Ssfc - kcpb[k] + Rsfc * kcmb[k];  // last row (:117)
        EB[k] = 0.0 - sB[c];                      // row 0 (:96): -cm0 of the top layer
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    if (pr < npairs) { sA[pr] = EA[k]; sB[pr] = EB[k]; }
  }
  __syncthreads();

  // ---- tridiagonal solve, one lane per g-point column
  if (tid < ng) thomas_column(sG, sX, sA, sB, nz, ng, tid, Rsfc);
  __syncthreads();

  // ---- level fluxes (:143-148, :288-293) and mean intensity (:135-140)
  const double inv_u1 = solar ? 1.7320508075688772 : 0.0;  // 1/u1, u1 = 1/sqrt(3)
  double ofu[TS_MAXP], ofd[TS_MAXP], oam[TS_MAXP];
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    ofu[k] = ofd[k] = oam[k] = 0.0;
    if (pr < npairs) {
      const int i = pr / ng, c = pr - i * ng;
      const E4 e = make_e(kG[k], kx[k]);
      const double y1 = (i == 0) ? sB[(nz - 1) * ng + c] : sB[pr - ng];
      const double y2 = sA[pr];
      ofu[k] = (y1 * e.e1 + y2 * e.e2 + kcpb[k]);
      ofd[k] = (y1 * e.e3 + y2 * e.e4 + kcmb[k]) + kdir[k];
      oam[k] = inv_u1 * (y1 * (e.e1 + e.e3) + y2 * (e.e2 + e.e4) + kcpb[k] + kcmb[k]) + kdiru[k];
      if (i == 0) {
        const double top = (y1 * e.e3 - y2 * e.e4) + cp0_first;
        sL0[c] = top;                                         // fup(1)
        sL0[2 * ng + c] = inv_u1 * top + sL0[2 * ng + c];     // amean(1)
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    if (pr < npairs) { sG[pr] = ofu[k]; sX[pr] = ofd[k]; sA[pr] = oam[k]; }
  }
  __syncthreads();

  // ---- g-point weights (radiate.f90:122-126), unit factors (:167-180), reversal to
  //      ground-first (:140-154)
  double scale = 1.0, am_scale = 0.0;
  if (solar) {
    scale = p.photons_sol[ll] * p.photon_scale_factor;  // clima_radtran.f90:302
  }
  for (int n = tid; n < nz + 1; n += nt) {
    double fu = 0.0, fd = 0.0, am = 0.0;
    for (int c = 0; c < ng; c++) {
      const double w = p.wbin[c];
      if (n == 0) {
        fu = fu + sL0[c] * w;
        fd = fd + sL0[ng + c] * w;
        am = am + sL0[2 * ng + c] * w;
      } else {
        fu = fu + sG[(n - 1) * ng + c] * w;
        fd = fd + sX[(n - 1) * ng + c] * w;
        am = am + sA[(n - 1) * ng + c] * w;
      }
    }
    const size_t o = (size_t)ll * (nz + 1) + (nz - n);
    if (solar) {
      p.sol_fup_a[o] = fu * scale * p.diurnal_fac;
      p.sol_fdn_a[o] = fd * scale * p.diurnal_fac;
      am = am * scale * p.diurnal_fac;
      am = am * p.am_f1[ll];
      am = am * p.am_f2[ll] * p.am_dw[ll];
      p.sol_amean[o] = am;
    } else {
      p.ir_fup_a[o] = fu;
      p.ir_fdn_a[o] = fd;
    }
  }
  (void)am_scale;
  double *tb = solar ? p.sol_tau_band : p.ir_tau_band;
  for (int i = tid; i < nz; i += nt) tb[(size_t)ll * nz + i] = p.tau_band[(size_t)l * nz + (nz - 1 - i)];
}

bool launch_twostream(const TwoStreamParams &p, hipStream_t s, size_t *lds_bytes) {
  const int npairs = p.nz * p.ng;
  const size_t lds = sizeof(double) * ((size_t)4 * npairs + (p.nz + 1) + 3 * p.ng);
  if (lds_bytes) *lds_bytes = lds;
  if (lds > 160 * 1024) return false;
  int threads = (npairs + TS_MAXP - 1) / TS_MAXP;
  threads = ((threads + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  if (threads > 1024) return false;
  const int grid = p.n_sol + p.n_ir;
  if (grid <= 0) return true;
  if (threads <= 512) {
    static bool attr512 = false;
    if (!attr512) { (void)hipFuncSetAttribute((const void *)k_twostream<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr512 = true; }
    hipLaunchKernelGGL(k_twostream<512>, dim3(grid), dim3(threads), lds, s, p);
  } else {
    static bool attr1024 = false;
    if (!attr1024) { (void)hipFuncSetAttribute((const void *)k_twostream<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1024 = true; }
    hipLaunchKernelGGL(k_twostream<1024>, dim3(grid), dim3(threads), lds, s, p);
  }
  return true;
}

// ------------------------------------------------------------------------------------
// k_integrate: fup_n(i) = sum_l fup_a(i,l)*(freq(l)-freq(l+1)) in bin order
// (radiate.f90:184-192); one thread per (array, level); then f_total (clima_radtran.f90:316)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_integrate(IntegrateParams p) {
  const int nl = p.nz + 1;
  for (int t = threadIdx.x; t < 4 * nl; t += blockDim.x) {
    const int a = t / nl, i = t - a * nl;
    const bool sol = a >= 2;
    if (sol && !p.do_solar) continue;
    const double *src = a == 0 ? p.ir_fup_a : a == 1 ? p.ir_fdn_a : a == 2 ? p.sol_fup_a : p.sol_fdn_a;
    const double *freq = sol ? p.sol_freq : p.ir_freq;
    const int lo = sol ? p.sol_lo : p.ir_lo, cnt = sol ? p.sol_n : p.ir_n;
    double acc = 0.0;
    for (int l = lo; l < lo + cnt; l++) {
      const double dfreq = freq[l] - freq[l + 1];
      acc = acc + src[(size_t)l * nl + i] * dfreq;
    }
    p.flux_n[a * nl + i] = acc;
  }
  __syncthreads();
  if (p.f_total)
    for (int i = threadIdx.x; i < nl; i += blockDim.x)
      p.f_total[i] = (p.flux_n[3 * nl + i] - p.flux_n[2 * nl + i]) + (p.flux_n[1 * nl + i] - p.flux_n[0 * nl + i]);
}

void launch_integrate(const IntegrateParams &p, hipStream_t s) {
  hipLaunchKernelGGL(k_integrate, dim3(1), dim3(1024), 0, s, p);
}

__global__ void k_f_total(int nl, const double *flux_n, double *f_total) {
  for (int i = threadIdx.x; i < nl; i += blockDim.x)
    f_total[i] = (flux_n[3 * nl + i] - flux_n[2 * nl + i]) + (flux_n[1 * nl + i] - flux_n[0 * nl + i]);
}
void launch_f_total(int nz, const double *flux_n, double *f_total, hipStream_t s) {
  hipLaunchKernelGGL(k_f_total, dim3(1), dim3(256), 0, s, nz + 1, flux_n, f_total);
}

__global__ void k_scale(double *a, size_t n, double f) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = a[i] * f;
}
void launch_scale(double *a, size_t n, double f, hipStream_t s) {
  if (n == 0) return;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_scale, dim3(grid), dim3(256), 0, s, a, n, f);
}

}  // namespace clima
