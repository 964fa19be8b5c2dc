// kernels.hip -- hand-written gfx950 (CDNA4) kernels for Clima's radiate() hot path.
//
//   k_prep             per-column pre-pass shared by every bin: log10P, columns, continuum weights, interpolation
//                      brackets / weights of every table axis (compute_opacity pre-pass,
//                      src/radtran/clima_radtran_types.f90:599-633 + dintrv bracketing); spare blocks clear what the
//                      two-stream forms that add partial sums accumulate into.  One launch, a column dimension.
//   k_opacity8         one LANE per (bin, source layer): k-table / CIA / continuum / Rayleigh / Mie gather +
//                      random-overlap resort-rebin (types.f90:640-888); the 64-key resort is a register-resident
//                      Batcher network (v_min_f64 / v_max_f64), the rebin the "window" form.
//   k_opacity_coop     NG = 8, 16 or 32 lanes per (bin, source layer): the sort across the group (DPP / ds_swizzle);
//                      few-item calls (a bin-sharded rank) and EVERY g-point count other than 8 (1..32: the lanes
//                      beyond ng are padded).
//   k_opacity_generic  any g-point count 1..32, one wave per item, LDS bitonic sort in the reference's arithmetic
//                      order: a cross-check (CLIMA_HIP_GENERIC=1), on no default path.
//   k_twostream_w      one WAVE per (channel, bin, g-point) -- twostream_p_body: lane q owns a chunk of layers in
//                      registers, per-lane elimination with flux boundary conditions, the chunks joined by DPP wave
//                      scans (3x3 projective suffix scan + affine prefix scan); half-wave (two columns per wave) and
//                      paired (AdiabatClimate's doubled grid) forms (radiate.f90:50-158, twostream.f90:10-295).
//   k_twostream_h      the half-wave form of twostream_p_body as a launch of its own (16 / 24 / 32 g-points).
//   k_twostream        one WORKGROUP per (channel, bin), LDS image + 16-chunk decomposition: beyond 512 layers.
//   k_twostream_ir_batch  many temperature columns on one set of opacities (the RCE Jacobian,
//                      src/adiabat/clima_adiabat_solve.f90:798-812): the temperature-independent part once per
//                      (bin, g-point); blocks of 8 waves up to 256 layers, of 4 (one per SIMD) up to 512.
//   k_green_*          (ir_green.inc) the same batch as a response problem: columns that are one profile with a few
//                      temperatures changed = F(profile) + unit responses x Planck differences; what depends on the
//                      opacities alone once per (bin, g-point), two FMAs per (level, deviation, bin, g-point).
//   k_fused            the production grid of a compute_opacity call: the opacity tiles of k_opacity8 followed, in
//                      the SAME launch, by the two-stream blocks of twostream_p_body, each waiting (bounded) for the
//                      tiles of its bin -- write-through hand-off, no cache-wide fence.
//   k_integrate_one    spectral integration (radiate.f90:184-192); f_total (clima_radtran.f90:316) on the host from
//                      the four level rows it fetches anyway.
//
// All arithmetic is IEEE binary64.  No MFMA: there is no dense contraction on this path.
#include "radtran_dev.h"
#include <algorithm>
#include <cstring>
#include <mutex>
#include <set>
#include <utility>

namespace clima {

// hipFuncAttributeMaxDynamicSharedMemorySize = 160 KiB, once per (device, kernel).  false when the
// runtime refuses it (the error stays in hipGetLastError for the caller to report).
static bool ensure_max_lds(const void *fn, int bytes = 160 * 1024) {
  static std::mutex mu;
  static std::set<std::pair<int, const void *>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({dev, fn})) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done.insert({dev, fn});
  return true;
}
// compute units of the current device (256 when the query fails)
static int device_cus() {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lk(mu);
  if (!cus[dev]) {
    hipDeviceProp_t pr;
    cus[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  return cus[dev];
}

// Diagnostic build only (-DCLIMA_STAMPS): s_memtime stamps of one wave, written to a buffer
// nothing else reads.  The production build contains no stamp.
#ifdef CLIMA_STAMPS
#define STAMP(buf, slot)                                                          \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    if ((buf) && tile == 100 && threadIdx.x == 0) (buf)[slot] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)
#else
#define STAMP(buf, slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------

// v_min_f64 / v_max_f64 without the sNaN-quieting canonicalisation hipcc wraps around
// fmin/fmax (the operands here are never NaN)
__device__ __forceinline__ double dmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double dmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// 1/x to ~1 ulp: v_rcp_f64 + two Newton steps (no scaling: |x| is O(1) where this is used)
// v_rcp_f64 alone is good to ~2^-24; one Newton step leaves up to 2e-15, two give the correctly
// rounded reciprocal on 4e5 test values (tests/devtools/gpu_rcp_accuracy.py)
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// sqrt(x) for x of ordinary magnitude (no scaling, no special cases: lambda^2 is O(1) and positive
// where this is used): v_rsq_f64 + coupled Goldschmidt step + two residual corrections, 10
// instructions where the library form with its range scaling and class tests takes ~20.
// <= 1 ulp on 4e5 values in [1e-6, 1e6] (tests/test_gpu_parity.py::test_device_rcp_and_sqrt).
__device__ __forceinline__ double sqrt_nr(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return g;
}

// exp(x) in ~23 f64 operations (the ocml exp is ~2x that): k = rint(x*log2e),
// r = x - k*ln2 (two-term Cody-Waite), degree-13 Taylor polynomial in r (|r| <= 0.35,
// truncation 5e-18), scaled by 2^k (two v_ldexp_f64).  Error <= 1 ulp over the arguments of this
// path; underflows to 0 like exp().  tests/test_gpu_parity.py::test_device_exp checks it.
__device__ __forceinline__ double exp_scale(double p, double k) {
  // scale by 2^k with v_ldexp_f64, in two steps because it returns inf for an exponent argument of
  // 1024 even where p*2^1024 is representable (p < 1); both steps are exact (or round once, into
  // the subnormals).  The convert saturates, so huge |x| end in 0 / inf as they should.
  const int ki = (int)k;
  const int kk = min(ki, 1023);
  return __builtin_ldexp(__builtin_ldexp(p, kk), ki - kk);
}
__device__ __forceinline__ double fast_exp(double x) {
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
  r = __builtin_fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                 // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);     // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);    // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);    // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);   // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);     // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);    // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);    // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);    // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02);   // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01);   // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return exp_scale(p, k);
}

// The same exp with its 14 non-inline constants held in VGPRs by the caller (ExpK::load() makes
// them opaque, so hipcc cannot fall back to materialising them again).  With literal constants every
// Horner step compiles to "two v_mov_b32 (or one v_mov_b64) of the constant into the destination,
// then v_fmac_f64": 2-3 instructions where the three-address v_fma_f64 below needs one.  A wave
// of this code issues roughly one instruction per 9-10 cycles whatever its type, so it is the
// instruction count of the wave, not the VALU's, that sets its run time.  Same operations, same
// rounding: results are bitwise those of fast_exp().
struct ExpK {
  double l2e, ln2h, ln2l, c[11];
  __device__ __forceinline__ void load() {
    const double v[14] = {1.4426950408889634074, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
                          1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08,
                          2.755731922398589e-07, 2.7557319223985893e-06, 2.48015873015873e-05,
                          1.984126984126984e-04, 1.388888888888889e-03, 8.333333333333333e-03,
                          4.1666666666666664e-02, 1.6666666666666666e-01};
    l2e = v[0]; ln2h = v[1]; ln2l = v[2];
    asm volatile("" : "+v"(l2e), "+v"(ln2h), "+v"(ln2l));
#pragma unroll
    for (int i = 0; i < 11; i++) {
      c[i] = v[3 + i];
      asm volatile("" : "+v"(c[i]));
    }
  }
};
__device__ __forceinline__ double fma3(double a, double b, double c) {
#ifdef CLIMA_FMA3_ASM
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
#else
  return __builtin_fma(a, b, c);
#endif
}
__device__ __forceinline__ double fast_exp(double x, const ExpK &K) {
  const double k = __builtin_rint(x * K.l2e);
  double r = fma3(k, K.ln2h, x);
  r = fma3(k, K.ln2l, r);
  double p = fma3(K.c[0], r, K.c[1]);
#pragma unroll
  for (int i = 2; i < 11; i++) p = fma3(p, r, K.c[i]);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return exp_scale(p, k);
}

// exp(x) for attenuation factors (x <= 0), table form:
//   f = x*256/ln2,  n = rint(f),  r = f - n (exact, |r| <= 1/2),  exp(x) = 2^(n>>8) * 2^((n&255)/256) * 2^(r/256),
// with the 256 correctly rounded values 2^(i/256) in LDS and a degree-4 polynomial in r for 2^(r/256)
// (truncation 4e-17): 14 instructions + one LDS read where fast_exp() takes 23.  The only error beyond
// ~1 ulp is the rounding of f, |x| * 1.1e-16 relative -- that of an argument known to one rounding, which
// x (a product) is anyway.  Needs no range clamp: for |f| >= 2^52 r is 0, the conversion of n saturates and
// the ldexp underflows to 0.  Used where a wave evaluates many exponentials whose results are then summed
// with weights (the zenith-angle loop); tests/test_gpu_parity.py::test_device_exp_table.
constexpr int EXP2_N = 256;
__device__ const double EXP2_TAB[EXP2_N] = {
    1.0, 1.0027112750502025, 1.0054299011128027, 1.0081558981184175,
    1.0108892860517005, 1.0136300849514894, 1.016378314910953, 1.019133996077738,
    1.0218971486541166, 1.0246677928971357, 1.0274459491187637, 1.030231637686041,
    1.0330248790212284, 1.0358256936019572, 1.0386341019613787, 1.041450124688316,
    1.0442737824274138, 1.0471050958792898, 1.0499440858006872, 1.0527907730046264,
    1.0556451783605572, 1.0585073227945128, 1.061377227289262, 1.0642549128844645,
    1.0671404006768237, 1.0700337118202419, 1.0729348675259756, 1.075843889062791,
    1.0787607977571199, 1.0816856149932152, 1.0846183622133092, 1.0875590609177697,
    1.0905077326652577, 1.0934643990728858, 1.0964290818163769, 1.099401802630222,
    1.102382583307841, 1.1053714457017412, 1.1083684117236787, 1.1113735033448175,
    1.1143867425958924, 1.1174081515673693, 1.1204377524096067, 1.12347556733302,
    1.1265216186082418, 1.129575928566288, 1.1326385195987192, 1.1357094141578055,
    1.1387886347566916, 1.1418762039695616, 1.1449721444318042, 1.148076478840179,
    1.1511892299529827, 1.154310420590216, 1.1574400736337511, 1.1605782120274988,
    1.1637248587775775, 1.1668800369524817, 1.1700437696832502, 1.1732160801636373,
    1.1763969916502812, 1.1795865274628758, 1.182784710984341, 1.1859915656609938,
    1.189207115002721, 1.1924313825831512, 1.1956643920398273, 1.1989061670743806,
    1.202156731452703, 1.2054161090051239, 1.2086843236265816, 1.2119613992768012,
    1.215247359980469, 1.2185422298274085, 1.2218460329727576, 1.2251587936371455,
    1.22848053610687, 1.2318112847340759, 1.2351510639369334, 1.2384998981998165,
    1.241857812073484, 1.245224830175258, 1.2486009771892048, 1.2519862778663162,
    1.255380757024691, 1.2587844395497165, 1.2621973503942507, 1.2656195145788063,
    1.2690509571917332, 1.2724917033894028, 1.275941778396392, 1.2794012075056693,
    1.2828700160787783, 1.2863482295460256, 1.2898358734066657, 1.2933329732290895,
    1.2968395546510096, 1.3003556433796506, 1.3038812651919358, 1.3074164459346773,
    1.3109612115247644, 1.3145155879493546, 1.318079601266064, 1.3216532776031575,
    1.3252366431597413, 1.3288297242059544, 1.3324325470831615, 1.3360451382041458,
    1.339667524053303, 1.3432997311868353, 1.3469417862329458, 1.3505937158920345,
    1.3542555469368927, 1.3579273062129011, 1.3616090206382248, 1.365300717204012,
    1.3690024229745905, 1.3727141650876684, 1.3764359707545302, 1.380167867260238,
    1.383909881963832, 1.387662042298529, 1.3914243757719262, 1.3951969099662003,
    1.3989796725383112, 1.4027726912202048, 1.4065759938190154, 1.4103896082172707,
    1.4142135623730951, 1.4180478843204152, 1.4218926021691656, 1.4257477441054942,
    1.42961333839197, 1.433489413367789, 1.4373759974489824, 1.4412731191286257,
    1.4451808069770467, 1.449099089642035, 1.4530279958490526, 1.4569675544014438,
    1.460917794180647, 1.4648787441464057, 1.4688504333369818, 1.4728328908693675,
    1.4768261459394993, 1.4808302278224719, 1.4848451658727524, 1.488870989524397,
    1.4929077282912648, 1.4969554117672355, 1.5010140696264256, 1.5050837316234065,
    1.5091644275934228, 1.5132561874526098, 1.5173590411982147, 1.5214730189088146,
    1.5255981507445384, 1.529734466947287, 1.533881997840956, 1.5380407738316568,
    1.5422108254079407, 1.5463921831410214, 1.550584877685, 1.5547889397770887,
    1.559004400237837, 1.5632312899713576, 1.567469639965553, 1.5717194812923414,
    1.5759808451078865, 1.5802537626528246, 1.5845382652524937, 1.588834384317164,
    1.593142151342267, 1.597461597908627, 1.6017927556826934, 1.606135656416771,
    1.6104903319492543, 1.6148568142048607, 1.6192351351948637, 1.6236253270173289,
    1.6280274218573478, 1.632441451987275, 1.6368674497669644, 1.6413054476440063,
    1.645755478153965, 1.6502175739206177, 1.6546917676561943, 1.6591780921616162,
    1.6636765803267364, 1.6681872651305825, 1.6727101796415966, 1.6772453570178785,
    1.681792830507429, 1.6863526334483934, 1.6909247992693053, 1.6955093614893326,
    1.7001063537185235, 1.7047158096580513, 1.709337763100463, 1.713972247929926,
    1.718619298122478, 1.723278947746274, 1.7279512309618377, 1.732636182022311,
    1.7373338352737062, 1.7420442251551564, 1.746767386199169, 1.7515033530318782,
    1.7562521603732995, 1.761013843037584, 1.7657884359332727, 1.7705759740635547,
    1.7753764925265212, 1.7801900265154245, 1.785016611318935, 1.789856282321401,
    1.7947090750031072, 1.7995750249405351, 1.804454167806624, 1.809346539371032,
    1.8142521755003989, 1.8191711121586085, 1.8241033854070534, 1.8290490314048973,
    1.8340080864093424, 1.8389805867758937, 1.843966568958626, 1.8489660695104508,
    1.8539791250833855, 1.8590057724288205, 1.864046048397789, 1.8690999899412386,
    1.8741676341103, 1.8792490180565602, 1.8843441790323345, 1.8894531543909392,
    1.8945759815869656, 1.8997126981765553, 1.9048633418176741, 1.9100279502703899,
    1.9152065613971474, 1.9203992131630474, 1.925605943636125, 1.930826790987627,
    1.9360617934922943, 1.9413109895286405, 1.9465744175792332, 1.9518521162309783,
    1.9571441241754002, 1.9624504802089273, 1.9677712232331759, 1.9731063922552343,
    1.978456026387951, 1.9838201648502194, 1.9891988469672663, 1.9945921121709402};
constexpr double EXP2_PER_E = 369.3299304675746, EXP2_PER_10 = 850.4135922911647;   // 256/ln2, 256 log2(10)
// 2^(x k / 256): k = EXP2_PER_E for e^x, EXP2_PER_10 for 10^x -- or that times a factor of the argument
// the caller has at hand (the zenith-angle loop passes tau' and -EXP2_PER_E/u0: one product instead of two)
__device__ __forceinline__ double exp_tab_any(double x, double k, const double *s_tab) {
  double f, nf, r;
  {
    // (contracted into fma(x, k, -nf), r would be the rounding residual of the product when |f| >= 2^52:
    // huge, and the result inf instead of 0)
#pragma clang fp contract(off)
    f = x * k;
    nf = __builtin_rint(f);
    r = f - nf;
  }
  const int n = (int)nf;
  const double t = s_tab[n & (EXP2_N - 1)];
  double p = 2.239395190875157e-12;              // (ln2/256)^k / k!, k = 4 .. 1
  p = __builtin_fma(p, r, 3.3083026805413713e-09);
  p = __builtin_fma(p, r, 3.6655655969101062e-06);
  p = __builtin_fma(p, r, 0.0027076061740622863);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(t * p, n >> 8);
}
__device__ __forceinline__ double exp_tab(double x, const double *s_tab) { return exp_tab_any(x, EXP2_PER_E, s_tab); }
// 10^y the same way (ten2power, src/clima_eqns.f90:75-80, for the opacity tile's table interpolations:
// 14 instructions where ten2power() below takes 24); relative error <= 1.5 ulp + |y| * 2.6e-16
__device__ __forceinline__ double ten2power_tab(double y, const double *s_tab) { return exp_tab_any(y, EXP2_PER_10, s_tab); }
// 1/x with one Newton step on v_rcp_f64: relative error <= 2e-15 (tests/devtools/gpu_rcp_accuracy.py),
// for factors that enter sums of weighted source terms
__device__ __forceinline__ double rcp_n1(double x) {
  double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// ten2power, src/clima_eqns.f90:75-80
__device__ __forceinline__ double ten2power(double y) { return fast_exp(y * LN10); }

// planck_fcn, src/clima_eqns.f90:64-73
__device__ __forceinline__ double planck_fcn(double nu, double T, const ExpK &K) {
  return 1.0e3 * ((2.0 * PLANK * (nu * nu * nu)) / (C_LIGHT * C_LIGHT)) *
         ((1.0) / (fast_exp((PLANK * nu) / (K_BOLTZ_SI * T), K) - 1.0));
}
__device__ __forceinline__ double planck_fcn(double nu, double T) {
  return 1.0e3 * ((2.0 * PLANK * (nu * nu * nu)) / (C_LIGHT * C_LIGHT)) *
         ((1.0) / (fast_exp((PLANK * nu) / (K_BOLTZ_SI * T)) - 1.0));
}

// A double moved between lanes by DPP (data-parallel primitives: the operand routing of a VALU move, a few cycles)
// instead of a ds_bpermute round trip through the LDS crossbar (~100+).  Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_mov(double old, double src) {
  const long long o = __double_as_longlong(old), v = __double_as_longlong(src);
  const int lo = __builtin_amdgcn_update_dpp((int)o, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(v >> 32), CTRL, ROW_MASK, BANK_MASK, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
// x of lane - D / lane + D within the 16-lane row (D = 1..15); `old` where the row has no such lane
template <int D>
__device__ __forceinline__ double row_shr(double old, double x) { return dpp_mov<DPP_ROW_SHR + D, 0xf, 0xf>(old, x); }
template <int D>
__device__ __forceinline__ double row_shl(double old, double x) { return dpp_mov<DPP_ROW_SHL + D, 0xf, 0xf>(old, x); }

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
// k_prep: block 0 writes the per-layer column quantities; block b>=1 evaluates
// interpolation slot b-1 for every layer (axis staged in LDS).  pair_reuse (types.f90:621-632)
// arrives with the column (decided on the host at upload, ColumnDev::meta), so there is no
// inter-block dependency.  gridDim.y = columns of a batch.
// ------------------------------------------------------------------------------------
constexpr int PREP_AXIS_MAX = 1024;
constexpr int PREP_ZERO_BLOCKS = 32;  // blocks per output array cleared by the prep launch

// column `cb` of a batch: every per-column pointer moved by that column's stride
__device__ __forceinline__ ColumnDev column_at(const ColumnDev &c0, const BatchStrides &bs, const int cb) {
  ColumnDev c = c0;
  const size_t oc = (size_t)cb * bs.col, op = (size_t)cb * bs.prep;
  c.T += oc; c.P += oc; c.dz += oc; c.dens += oc; c.pdens += oc; c.radii += oc; c.T_surface += oc;
  c.meta += 2 * oc;
  c.log10P += op; c.cols += op; c.foreign_col += op; c.absw += op; c.q += op;
  c.ix += 2 * op;
  return c;
}

__global__ __launch_bounds__(256) void k_prep(PrepParams p) {
  __shared__ double s_axis[PREP_AXIS_MAX];
  const int nz = p.nz;
  const int cb = blockIdx.y;
  const ColumnDev c = column_at(p.col, p.bs, cb);
  const int *src = c.meta + 1 + nz;  // source layer of every layer (pair_reuse, decided on the host)
  if (blockIdx.x == 0) {
    for (int j = threadIdx.x; j < nz; j += blockDim.x) {
      const double Pj = c.P[j], dzj = c.dz[j];
      double fc = 0.0;
      for (int i0 = 0; i0 < p.nsp; i0 += 8) {  // :607-619, loads of a batch issued together
        double d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = c.dens[min(i0 + k, p.nsp - 1) * nz + j];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int i = i0 + k;
          if (i < p.nsp) {
            const double col = d[k] * dzj;
            c.cols[i * nz + j] = col;
            if (p.has_cont && i != p.LH2O) fc = fc + col;
          }
        }
      }
      c.log10P[j] = log10(Pj);  // types.f90:605
      c.foreign_col[j] = fc;
    }
    return;
  }
  if ((int)blockIdx.x > p.nslots + p.nabs) {
    // spare blocks clear the output spectra that the two-stream kernel accumulates into
    const int zb = (int)blockIdx.x - (p.nslots + p.nabs + 1);
    const int arr = zb / PREP_ZERO_BLOCKS, part = zb - arr * PREP_ZERO_BLOCKS;
    double *dst = p.zero_ptr[arr] + (size_t)cb * p.bs.res;
    const size_t n = p.zero_count[arr];
    for (size_t i = (size_t)part * blockDim.x + threadIdx.x; i < n; i += (size_t)PREP_ZERO_BLOCKS * blockDim.x) dst[i] = 0.0;
    return;
  }
  if ((int)blockIdx.x > p.nslots) {
    // bin-independent weight of one continuum term for every layer (types.f90:696-723)
    const int e = (int)blockIdx.x - p.nslots - 1;
    const int a = p.abs_a[e], b = p.abs_b[e], kind = p.abs_kind[e];
    for (int j = threadIdx.x; j < nz; j += blockDim.x) {
      double w;
      if (kind == ABS_CIA) w = c.dens[a * nz + j] * c.dens[b * nz + j] * c.dz[j];
      else if (kind == ABS_COLUMN) w = c.dens[a * nz + j] * c.dz[j];
      else if (kind == ABS_H2O_SELF) w = c.dens[a * nz + j] * (c.dens[a * nz + j] * c.dz[j]);
      else if (kind == ABS_ZERO) w = 0.0;
      else {
        double fc = 0.0;  // foreign column (:610-619)
        for (int i = 0; i < p.nsp; i++)
          if (i != p.LH2O) fc = fc + c.dens[i * nz + j] * c.dz[j];
        w = c.dens[a * nz + j] * fc;
      }
      c.absw[e * nz + j] = w;
    }
    return;
  }
  // interpolation bracket and weight of one slot for every layer; reuse layers take their
  // source layer's inputs, which is what copying its interpolated value amounts to
  // (:652-653, :907-908, :933-935, :963-968)
  const int s = blockIdx.x - 1;
  const SlotDev &sl = p.slots[s];
  const bool in_lds = sl.n <= PREP_AXIS_MAX;
  // this thread's first layer: its input goes out together with the axis (and, for a layer that is its own source -- all of
  // them unless the column has reuse pairs -- with the source index itself): one memory round trip where the axis, the
  // index and the value took three, one behind the other
  const double *in = sl.source <= 0 ? c.P : sl.source == 1 ? c.T : c.radii + (size_t)(sl.source - 2) * nz;
  const int j0 = threadIdx.x;
  int js0 = 0;
  double x0 = 0.0;
  if (j0 < nz) { js0 = sl.source < 0 ? j0 : src[j0]; x0 = in[j0]; }
  if (in_lds)
    for (int i = threadIdx.x; i < sl.n; i += blockDim.x) s_axis[i] = sl.axis[i];
  __syncthreads();
  const double *axis = in_lds ? s_axis : sl.axis;
  for (int j = threadIdx.x; j < nz; j += blockDim.x) {
    // custom optical properties are evaluated for every layer itself (types.f90:564-569)
    const int js = j == j0 ? js0 : (sl.source < 0 ? j : src[j]);
    const double xin = (j == j0 && js == j) ? x0 : in[js];
    double x;
    if (sl.source < 0) x = log10(xin * 1.0e6);  // log10P_cgs, types.f90:606
    else if (sl.source == 0) x = log10(xin);
    else x = xin;
    if (sl.flag_clamp && (x < sl.lo || x > sl.hi)) atomicMax(c.err_flag, p.call_id);  // stamped, never reset
    x = fmin(fmax(x, sl.lo), sl.hi);  // :655-656, :910, :937, :974
    const int i = bracket(axis, sl.n, x);
    c.ix[s * nz + j] = i;
    c.q[s * nz + j] = (x - axis[i]) / (axis[i + 1] - axis[i]);  // linear_interpolation_module.F90:256, :319-320
  }
}

void launch_prep(const PrepParams &p, hipStream_t s) {
  hipLaunchKernelGGL(k_prep, dim3(1 + p.nslots + p.nabs + p.nzero * PREP_ZERO_BLOCKS, p.ncol > 0 ? p.ncol : 1), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------
// k_opacity
// ------------------------------------------------------------------------------------

// linear_interp_1d%evaluate (linear_interpolation_module.F90:256-259)
__device__ __forceinline__ double lerp1(const double *f, int i, double q) {
  const double p1 = 1.0 - q;
  return p1 * f[i] + q * f[i + 1];
}

constexpr int OP_THREADS = 256;
// s_getreg operand: HW_REG_HW_ID (id 4), field WAVE_ID (offset 0, 4 bits) -- the wave's slot on its SIMD
constexpr int HWREG_HW_ID_WAVE_ID = (3 << 11) | (0 << 6) | 4;

// Device-scope accesses for data handed from one block to another INSIDE a launch (k_fused):
// stores write through and loads bypass the per-XCD L2, so no cache-wide writeback /
// invalidate (what an agent-scope fence costs on this part) is needed.
template <bool COHERENT>
__device__ __forceinline__ void st_opr(double *p, double v) {
  if constexpr (COHERENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COHERENT>
__device__ __forceinline__ double ld_opr(const double *p) {
  if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

#ifdef CLIMA_STAMPS
__device__ long long *g_stamp_buf = nullptr;   // (two-stream blocks of the diagnostic build find the buffer here)
#endif

// Random-overlap resort + rebin for NG = 8 (k_rorr, types.f90:826-852), one lane per
// (bin, layer).  x = current mixture tau_k(8), y = new species' k*col (8), in registers.
// The 64 sums x_i+y_j are sorted by a Batcher odd-even merge network held in registers
// (one v_min_f64 + one v_max_f64 per compare-exchange).  Each key carries its pair index
// in mantissa bits 3-8 (KEY_IDX_MASK): that orders ties exactly like the stable rank on
// (value,index) whenever two values differ above 2^-44 relative, and it is how the sorted
// stream finds its weight wxy(idx).  The value used downstream is the key itself
// (relative perturbation <= 2^-44 = 5.7e-14, the size of the fast_exp() argument
// rounding that the k-table interpolation already carries).  When y is ascending (the normal
// case for k-distributions) the 8 runs of 8 keys are pre-sorted and only the merge tail of
// the network runs; when x is ascending too (it is, up to rounding, after the first mixing
// step) the first stage of every merge level is redundant as well: 295 of 543 exchanges.
// Rebin (weights_to_bins + futils rebin, types.f90:846-847) as a stream over the sorted
// keys: c = running sum of the sorted weights, S = running integral of the sorted step
// function; whenever c passes an output edge E_k the integral up to that edge,
// I_k = S + v*(E_k - c0), is written to the lane's private LDS slot k.  The new coefficients
// are (I_k - I_{k-1}) / (E_k - E_{k-1}).
// Rebin form RM = 0 ("window" form, the default): the integral of the sorted step function is convex
// and piecewise linear in the cumulative weight c, with ascending slopes (the sorted keys), so it is
// the maximum of the lines through its pieces,
//     I(E_k) = max_j [ IC_{j-1} + v_j * (E_k - C_{j-1}) ],   C = running weight, IC = running integral,
// and the maximising j is the element that crosses E_k -- whose term is exactly the expression the
// streaming form evaluates there.  No search, no branch, no LDS slots: one subtract, one fma, one max
// per (element, edge) pair.  Only pairs that CAN be a crossing are evaluated: whatever order the sort
// produces, C_j lies between the sum of the j+1 smallest and the j+1 largest pair weights, which
// bounds the crossing element of edge k to a window [rb_lo(k), rb_hi(k)] -- 144 pairs instead of 448
// for 8 Gauss-Legendre weights (tools/gen_rebin_windows.py prints the tables; the host checks the
// handle's actual weights against them and otherwise selects the streaming form, RM = 1 or 2).  A
// window wider than necessary is harmless (every line lies below the integral).
// Two window tables.  WIDE: any order of the 64 sums (bounds over arbitrary subsets of the pair
// weights): 144 pairs.  TIGHT: x and y both ascending (the wave-uniform `xys` case, which is the
// normal one: k-coefficients ascend in g, and so does a rebinned mixture): the sorted order is then a
// linear extension of the 8x8 product order, every prefix is a down-set (a Young diagram) of the
// grid, and the bounds over down-sets of each size confine the crossings to 52 pairs.
template <bool TIGHT>
constexpr int rb_lo(int k) {
  constexpr int wide[8] = {0, 1, 5, 10, 18, 27, 39, 52};
  constexpr int tight[8] = {0, 5, 12, 19, 27, 35, 45, 55};
  return TIGHT ? tight[k] : wide[k];
}
template <bool TIGHT>
constexpr int rb_hi(int k) {
  constexpr int wide[8] = {0, 11, 24, 36, 45, 53, 58, 62};
  constexpr int tight[8] = {0, 8, 18, 28, 36, 44, 51, 58};
  return TIGHT ? tight[k] : wide[k];
}

// A sort key carries the index i*8+j of its (x_i, y_j) pair in mantissa bits 3-8 -- as the byte offset
// of the pair's weight in the LDS table, so that the lookup is one AND and the read (in bits 0-5 it
// took a shift as well: 64 instructions per mixing step).  The key is the value that enters the
// integral, so the sum x_i + y_j is perturbed by at most 2^-44 of itself.
constexpr unsigned long long KEY_IDX_MASK = 0x1f8ULL;
__device__ __forceinline__ double key_weight(const double key, const double *s_wxy) {
  const unsigned int off = (unsigned int)__double_as_longlong(key) & (unsigned int)KEY_IDX_MASK;
  return *(const double *)((const char *)s_wxy + off);
}

// One straight-line pass serves both tables: the pairs of the tight table always, the pairs that only
// the wide table holds under a wave-uniform `if (!xys)` per element (two separate unrolled passes
// behind one branch cost the register allocator 40 spilled registers).
__device__ __forceinline__ void rebin_window(const double (&key)[64], const bool xys, const double *s_wxy,
                                             const double *s_Ew, const double *rW, double (&out)[8]) {
  // the interior edges, wave-uniform: fetched from LDS for this rebin only (held in vector registers for
  // the whole tile they cost 18 of them, which were then spilled around the sorts)
  asm volatile("" ::: "memory");
  double E[8];
#pragma unroll
  for (int k = 1; k < 8; k++) E[k] = s_Ew[k];
  double C = 0.0, IC = 0.0, Ie[9];
#pragma unroll
  for (int k = 1; k < 8; k++) Ie[k] = -1.0e300;  // below every candidate (the sums are finite)
  constexpr int RB = 8;
  double wn[RB];
#pragma unroll
  for (int u = 0; u < RB; u++) wn[u] = key_weight(key[u], s_wxy);
#pragma unroll
  for (int pb = 0; pb < 64; pb += RB) {
    double wv[RB];
#pragma unroll
    for (int u = 0; u < RB; u++) wv[u] = wn[u];
    if (pb + RB < 64) {
#pragma unroll
      for (int u = 0; u < RB; u++)
        wn[u] = key_weight(key[pb + RB + u], s_wxy);
    }
    const double Cb = C, ICb = IC;   // the running sums at the batch's first element (for the wide-only pass below)
#pragma unroll
    for (int u = 0; u < RB; u++) {
      const int j = pb + u;
      const double v = key[j];
#pragma unroll
      for (int k = 1; k < 8; k++) {
        if (j >= rb_lo<true>(k) && j <= rb_hi<true>(k))
          Ie[k] = dmax(Ie[k], __builtin_fma(v, E[k] - C, IC));
      }
      IC = __builtin_fma(v, wv[u], IC);  // weights_to_bins (clima_eqns.f90:43-54) and the integral, in sorted order
      C = C + wv[u];
    }
    // the pairs only the wide table holds, for the batch's eight elements behind ONE wave-uniform test (a
    // test per element cost ~90 scalar and branch instructions per rebin): the running sums are formed
    // again from the batch's start, same operations in the same order.  (One such pass over all 64
    // elements after the loop, a single test per rebin, made hipcc spill 110 registers: 165 us per call.)
    if (!xys) {
      double c2 = Cb, ic2 = ICb;
#pragma unroll
      for (int u = 0; u < RB; u++) {
        const int j = pb + u;
        const double v = key[j];
#pragma unroll
        for (int k = 1; k < 8; k++) {
          if (j >= rb_lo<false>(k) && j <= rb_hi<false>(k) && !(j >= rb_lo<true>(k) && j <= rb_hi<true>(k)))
            Ie[k] = dmax(Ie[k], __builtin_fma(v, E[k] - c2, ic2));
        }
        ic2 = __builtin_fma(v, wv[u], ic2);
        c2 = c2 + wv[u];
      }
    }
  }
  Ie[8] = IC;  // the last edge is the total weight
  out[0] = Ie[1] * rW[0];
#pragma unroll
  for (int q = 1; q < 8; q++) out[q] = (Ie[q + 1] - Ie[q]) * rW[q];
}

#include "rorr_xys_asm.inc"
#ifndef CLIMA_RORR_ASM
#define CLIMA_RORR_ASM 1
#endif
constexpr bool RORR_ASM = CLIMA_RORR_ASM != 0;   // (0: the compiler-scheduled step of rounds 1-3, for A/B timing)

template <int RM>
__device__ __forceinline__ void rorr_mix8(const double (&x)[8], const double (&y)[8],
                                          double (*sI)[OP_THREADS], const int tid, const int tile,
                                          const double *s_wxy, const double *s_E,
                                          const double *s_Ew, const double *rW,
                                          double (&out)[8], const double *rorr_tab, long long *stamps = nullptr, const int stamp_slot = 0) {
  (void)stamps; (void)stamp_slot;   // diagnostic build only: where the "sort done" stamp of this mixing step goes
  double key[64];
  bool ysorted = true, xsorted = true;
#pragma unroll
  for (int j = 0; j < 7; j++) {
    ysorted = ysorted && (y[j] <= y[j + 1]);
    xsorted = xsorted && (x[j] <= x[j + 1]);
  }
  const bool ys = __all(ysorted), xys = ys && __all(xsorted);
  if constexpr (RM == 0 && RORR_ASM) {
    // x and y ascending (the normal case: k-coefficients ascend in g, and so does a rebinned mixture) with the
    // window-form rebin: the whole step is one block of generated assembly (rorr_xys_asm.inc,
    // tools/gen_rorr_asm.py) that leaves out the merges and the rebin rows the wave's operands make unnecessary.
    if (__builtin_expect(xys, 1)) {
      double xx[8], yy[8];
#pragma unroll
      for (int g = 0; g < 8; g++) { xx[g] = x[g]; yy[g] = y[g]; }
      typedef __attribute__((address_space(3))) const double lds_cdouble;
      const unsigned wxy_addr = (unsigned)(unsigned long)(lds_cdouble *)s_wxy;   // 512-byte aligned (opacity8_body)
      asm volatile(RORR_XYS_ASM_TEXT
                   : "+v"(xx[0]), "+v"(xx[1]), "+v"(xx[2]), "+v"(xx[3]), "+v"(xx[4]), "+v"(xx[5]), "+v"(xx[6]), "+v"(xx[7]),
                     "+v"(yy[0]), "+v"(yy[1]), "+v"(yy[2]), "+v"(yy[3]), "+v"(yy[4]), "+v"(yy[5]), "+v"(yy[6]), "+v"(yy[7])
                   : "v"(wxy_addr), "s"(rorr_tab)
                   : RORR_XYS_ASM_CLOBBERS);
#pragma unroll
      for (int g = 0; g < 8; g++) out[g] = yy[g];   // the new coefficients come back in the y operands
      STAMP(stamps, stamp_slot);
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const double v = x[i] + y[j];  // tau_xy(:, j+(i-1)*ng), types.f90:828
      unsigned long long b = (unsigned long long)__double_as_longlong(v);
      b = (b & ~KEY_IDX_MASK) | (unsigned long long)((i * 8 + j) << 3);
      key[i * 8 + j] = __longlong_as_double((long long)b);
    }
  }
#define CE(a, b)                              \
  {                                           \
    const double lo_ = dmin(key[a], key[b]);  \
    const double hi_ = dmax(key[a], key[b]);  \
    key[a] = lo_;                             \
    key[b] = hi_;                             \
  }
// exchanges that operands ascending in both x and y never need (sort_network_64.inc)
#define CE_X(a, b) \
  if (!xys) CE(a, b)
#define CE_MERGE_BEGIN(P, M)
#define CE_MERGE_END
  if (!ys) {
#define CE_FULL_HEAD
#include "sort_network_64.inc"
#undef CE_FULL_HEAD
  }
  // merge levels; their first stage is redundant when x and y are both ascending
  if (!xys) {
#define CE_L8_FIRST
#include "sort_network_64.inc"
#undef CE_L8_FIRST
  }
#define CE_L8_REST
#include "sort_network_64.inc"
#undef CE_L8_REST
  if (!xys) {
#define CE_L16_FIRST
#include "sort_network_64.inc"
#undef CE_L16_FIRST
  }
#define CE_L16_REST
#include "sort_network_64.inc"
#undef CE_L16_REST
  if (!xys) {
#define CE_L32_FIRST
#include "sort_network_64.inc"
#undef CE_L32_FIRST
  }
#define CE_L32_REST
#include "sort_network_64.inc"
#undef CE_L32_REST
#undef CE_MERGE_BEGIN
#undef CE_MERGE_END
#undef CE_X
#undef CE
  STAMP(stamps, stamp_slot);
  if constexpr (RM == 0) {
    rebin_window(key, xys, s_wxy, s_Ew, rW, out);
    return;
  }
  const double *E = s_Ew;
  constexpr bool MULTI = RM == 2;
  double S = 0.0, c0 = 0.0;
  // Next output edge to pass is E[k]: bk = E[k], `en` points at E[k+1] in the LDS edge table and
  // bn = *en is prefetched; `slot` is where I(E_k) goes.  Running pointers instead of k keep the
  // crossing block (which some lane of the wave needs at nearly every element) short.
  double bk = E[1], bn = E[2];
  const double *en = s_E + 2;
  double *slot = &sI[0][tid];
  double *const slot_end = &sI[0][tid] + 8 * OP_THREADS;
  // weights_to_bins (clima_eqns.f90:43-54) on wxy(inds): the LDS lookups run one batch of
  // RB elements ahead of their use, so a batch pays one wait instead of one per element
  constexpr int RB = 8;
  double wn[RB];
#pragma unroll
  for (int u = 0; u < RB; u++) wn[u] = key_weight(key[u], s_wxy);
#pragma unroll
  for (int pb = 0; pb < 64; pb += RB) {
    double wv[RB];
#pragma unroll
    for (int u = 0; u < RB; u++) wv[u] = wn[u];
    if (pb + RB < 64) {
#pragma unroll
      for (int u = 0; u < RB; u++)
        wn[u] = key_weight(key[pb + RB + u], s_wxy);
    }
#pragma unroll
    for (int u = 0; u < RB; u++) {
      const double v = key[pb + u];  // value with the pair index in mantissa bits 3-8 (<= 2^-44 relative)
      const double w = wv[u];
      const double c1 = c0 + w;
      if (c1 > bk) {                    // this element reaches past E[k]
        *slot = __builtin_fma(v, bk - c0, S);
        slot += OP_THREADS;
        bk = bn;
        en++;
        if constexpr (MULTI) {
          while (c1 > bk) {             // one element spanning a whole output bin: only possible
            *slot = __builtin_fma(v, bk - c0, S);  // when max(wxy) > min(wbin)
            slot += OP_THREADS;
            bk = *en;
            en++;
          }
        }
      }
      bn = *en;  // refreshed unconditionally (unchanged if nothing was crossed): the lookup is
                 // then consumed one element later instead of at the branch join
      S = __builtin_fma(v, w, S);
      c0 = c1;
    }
  }
  for (; slot < slot_end; slot += OP_THREADS) *slot = S;  // edges at or beyond the total weight (rounding)
  double Ik[8];
#pragma unroll
  for (int q = 0; q < 8; q++) Ik[q] = sI[q][tid];
  out[0] = Ik[0] * rW[0];  // rW = 1/(E_{k+1}-E_k), formed once per kernel
#pragma unroll
  for (int q = 1; q < 8; q++) out[q] = (Ik[q] - Ik[q - 1]) * rW[q];
}

// Layer terms that are not k-distributions (types.f90:665-757)
struct LayerTerms {
  double tausg, taua, tauc, tausc, taup, tausp, gt;
};

// One tile of 256 lanes, one lane per (bin, SOURCE layer): a layer that pair_reuse marks as a copy of
// the layer below it (types.f90:621-632; AdiabatClimate's doubled radiative grid is all such pairs,
// src/adiabat/clima_adiabat.f90:729-773) has no lane of its own -- its source's lane writes both
// layers.  What the reference copies for the second layer is the interpolated k-coefficients and
// cross sections and the rebinned mixture (:652-653, :833-834, :907-908, :933-935); its columns and
// particle terms are its own.  When every input of the two layers is bitwise equal (SRC_EXACT: what
// the doubled grid produces) all of that is equal too and the lane stores its results twice;
// otherwise it evaluates the second layer's own terms before storing.
// `c` and the opr pointers are those of the tile's column (column_at / + c*bs.opr in a batch).
template <int RM, bool CUSTOM, bool COHERENT>
__device__ __forceinline__ void opacity8_body(const OpacityParams &p, const int tile, const size_t oc,
                                              const size_t opf, const size_t oo) {
  // column arrays at c.X[oc + ...] (ints: 2*oc), prep arrays at c.X[opf + ...] (ints: 2*opf), optical
  // properties at p.X[oo + ...]: the offsets of the tile's column in a batch (0 for a single call).
  // Offsets at the point of use rather than 15 shifted pointers up front: those would sit in scalar
  // registers for the whole tile, and the tile has none to spare.
  const ColumnDev &c = p.col;
  constexpr int NG = 8;
  __shared__ double sI[RM == 0 ? 1 : NG][OP_THREADS];  // streaming rebin: per-lane private slots (slot-major: conflict-free)
  __shared__ __align__(512) double s_wxy[NG * NG];   // (512: the assembly forms a weight's address as base | key bits 3-8)
  __shared__ double s_E[NG + 4];  // output edges followed by +inf sentinels
  const int tid = threadIdx.x;
  // wave-uniform tables read where they are used: the g-point edges E_0..E_8, then the g-point weights
  __shared__ double s_Ew[2 * NG + 1];
  __shared__ double s_e2[EXP2_N];   // exp table of ten2power_tab
  // 1/(E_{k+1}-E_k), read at the end of every rebin: wave-uniform values that the compiler kept in (and
  // spilled from) vector registers across the sorts
  __shared__ double rW[NG];
  {
    // Every table's loads go out together, at clamped indices, and only the LDS writes are predicated: written
    // as five `if (tid < n) table[tid] = source[tid]` these were five memory round trips one after the other
    // (each test's block waits for its own load) before a tile's first useful instruction.
    static_assert(EXP2_N == OP_THREADS, "one table entry per thread");
    const double v_wxy = p.wxy[tid & (NG * NG - 1)];
    const double v_Ep = p.wbin_e_pad[min(tid, NG + 3)];
    const double v_e2 = EXP2_TAB[tid];
    const double v_e = p.wbin_e[min(tid, NG)], v_e1 = p.wbin_e[min(tid + 1, NG)];
    const double v_w = p.wbin[min(max(tid - (NG + 1), 0), NG - 1)];
    if (tid < NG * NG) s_wxy[tid] = v_wxy;
    if (tid < NG + 4) s_E[tid] = v_Ep;
    s_e2[tid] = v_e2;
    if (tid < 2 * NG + 1) s_Ew[tid] = tid < NG + 1 ? v_e : v_w;
    if (tid < NG) rW[tid] = 1.0 / (v_e1 - v_e);
  }
  __syncthreads();

#ifdef CLIMA_STAMPS
  const long long wt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int nz = p.nz;
  const int nsrc = c.meta[2 * oc + 0];
  const long total = (long)p.nbins * nsrc;
  long t = (long)tile * OP_THREADS + tid;
  const bool valid = t < total;
  if (!valid) t = total - 1;
  const int l = p.bin_lo + (int)(t / nsrc);
  const int ent = c.meta[2 * oc + 1 + (int)(t % nsrc)];
  const int j = ent & SRC_LAYER;  // ground-first layer
  const bool pair = (ent & SRC_PAIR) != 0, exact = (ent & SRC_EXACT) != 0;
  const int n = nz - 1 - j;       // TOA-first index (types.f90:690-691, :862-865)

  STAMP(p.stamps, 0);
  // Everything of the layer that is not a k-distribution (Rayleigh, continuum, custom and particle
  // opacity) is needed only by the totals at the end, so it can run before or after the mixing
  // loop.  The two waves that share a SIMD take opposite orders (hardware wave-slot parity): one
  // is in this load-latency-bound part while the other is in the VALU-bound sort/rebin, instead of
  // both stalling on memory at the same time.
  auto layer_terms = [&](const int jl, LayerTerms &o) {
#ifdef CLIMA_EXP_NOTERMS   // timing experiment only (WRONG results): what the tile costs without its layer terms
    o.tausg = 1e-3; o.taua = 1e-3; o.tauc = TINY; o.tausc = TINY * TINY; o.taup = 0.0; o.tausp = 0.0; o.gt = 0.0;
    if (jl >= 0) return;
#endif
    const double dzj = c.dz[oc + jl];
    // ---- Rayleigh (:686-693)
    double tausg = 0.0;
  #pragma unroll 4
    for (int i = 0; i < p.nray; i++) tausg = tausg + p.ray[i].data[l] * c.cols[opf + p.ray[i].sp1 * nz + jl];
    // ---- continuum absorption: CIA, photolysis/absorption, H2O continuum (:665-677, :696-723).
    // Entries are processed eight at a time with every load of the batch issued before the
    // first use, so the dependent index -> table round trips overlap instead of queueing.
    // The host pads the list to a multiple of ABS_BATCH with terms of weight 0, so a batch needs no
    // per-term test (a branch costs a wave about four instruction slots).
    double taua = 0.0;
    constexpr int AB = ABS_BATCH;
    for (int e0 = 0; e0 < p.nabs; e0 += AB) {
      int ixx[AB];
      double qq[AB], ww[AB];
  #pragma unroll
      for (int u = 0; u < AB; u++) {
        const AbsEntry &x = p.abs[e0 + u];
        ixx[u] = c.ix[2 * opf + x.slot * nz + jl];
        qq[u] = c.q[opf + x.slot * nz + jl];
        ww[u] = c.absw[opf + (e0 + u) * nz + jl];
      }
      double v0[AB], v1[AB];
  #pragma unroll
      for (int u = 0; u < AB; u++) {
        const AbsEntry &x = p.abs[e0 + u];
#ifdef CLIMA_EXP_NODEP   // timing experiment only (WRONG results): the table loads do not wait for the bracket index
        const double *base = x.data + (x.nT ? (size_t)l * x.nT : (size_t)l);
#else
        const double *base = x.data + (x.nT ? (size_t)l * x.nT + ixx[u] : (size_t)l);
#endif
        v0[u] = base[0];
        v1[u] = base[x.nT ? 1 : 0];
      }
  #pragma unroll
      for (int u = 0; u < AB; u++) {
        const AbsEntry &x = p.abs[e0 + u];
        double sgm = v0[u];
        if (x.nT) sgm = ten2power_tab((1.0 - qq[u]) * v0[u] + qq[u] * v1[u], s_e2);  // lerp1 + ten2power (:910-912)
        taua = taua + sgm * ww[u];
      }
    }
    // ---- custom opacity (:540-572, :726-730); tiny everywhere when unset (:558-562)
    double tauc = TINY, tausc = TINY * TINY;
    double g0c = TINY;
    if constexpr (CUSTOM) {
      const int ix = c.ix[2 * opf + p.cust.slot * nz + jl];
      const double q = c.q[opf + p.cust.slot * nz + jl];
      const size_t o = (size_t)l * p.cust.nP;
      tauc = lerp1(p.cust.dtau + o, ix, q) * dzj;
      const double w0c = lerp1(p.cust.w0 + o, ix, q);
      g0c = lerp1(p.cust.g0 + o, ix, q);
      tausc = w0c * tauc;
    }
    // ---- particles (:680-683, :733-757)
    double tausp = 0.0, taup = 0.0;
    double tausp_1[MAX_PART], gtp[MAX_PART];
    for (int i = 0; i < p.npart; i++) {
      const PartDev &pt = p.part[i];
      const int ix = c.ix[2 * opf + pt.slot * nz + jl];
      const double q = c.q[opf + pt.slot * nz + jl];
      const double w0p = lerp1(pt.w0 + (size_t)l * pt.nrad, ix, q);
      const double qext = lerp1(pt.qext + (size_t)l * pt.nrad, ix, q);
      gtp[i] = lerp1(pt.gt + (size_t)l * pt.nrad, ix, q);
      const double rr = c.radii[oc + pt.p_ind * nz + jl];
      const double taup_1 = qext * PI * (rr * rr) * c.pdens[oc + pt.p_ind * nz + jl] * dzj;
      taup = taup + taup_1;
      tausp_1[i] = w0p * taup_1;
      tausp = tausp + tausp_1[i];
    }
    double gt = 0.0;
    for (int i = 0; i < p.npart; i++) gt = gt + gtp[i] * tausp_1[i] / fmax(TAU_MIN, (tausp + tausg + tausc));
    gt = gt + g0c * tausc / fmax(TAU_MIN, (tausp + tausg + tausc));
    gt = fmin(gt, MAX_GT);
    o.tausg = tausg; o.taua = taua; o.tauc = tauc; o.tausc = tausc; o.taup = taup; o.tausp = tausp; o.gt = gt;
  };
  LayerTerms lt;
  const bool terms_first = (__builtin_amdgcn_s_getreg(HWREG_HW_ID_WAVE_ID) & 1) == 0;
  // A wave that takes them first parks them in LDS for the mixing loop: the assembly block of a mixing step
  // (rorr_xys_asm.inc) names 192 registers, and every value the compiler has to carry across it beyond the 64 that
  // are left is spilled and reloaded through L2 with the wave stalled.
  __shared__ double s_lt[7][OP_THREADS];
  if (terms_first) {
    layer_terms(j, lt);
    if constexpr (RM == 0 && RORR_ASM) {
      s_lt[0][tid] = lt.tausg; s_lt[1][tid] = lt.taua; s_lt[2][tid] = lt.tauc; s_lt[3][tid] = lt.tausc;
      s_lt[4][tid] = lt.taup; s_lt[5][tid] = lt.tausp; s_lt[6][tid] = lt.gt;
    }
  }

  STAMP(p.stamps, 1);
  // ---- k-distributions (:649-662) and random-overlap mixing (k_rorr :816-854)
  // g-point coefficients of species s times the column of layer jl
  auto k_times_col = [&](const int s, const int jl, const int iP, const int iT, const double q1, const double q2,
                         double (&kc)[NG]) {
    const KDev &kd = p.k[s];
    const double p1 = 1.0 - q1, p2 = 1.0 - q2;
    const double *slab = kd.log10k + (size_t)l * kd.nT * kd.nP * NG;
    const double *f11 = slab + ((size_t)iT * kd.nP + iP) * NG;
    const double *f21 = f11 + NG;                    // iP+1
    const double *f12 = f11 + (size_t)kd.nP * NG;    // iT+1
    const double *f22 = f12 + NG;
    const double col = c.cols[opf + kd.sp * nz + jl];
#pragma unroll
    for (int g = 0; g < NG; g++) {
      // linear_interp_2d%evaluate, linear_interpolation_module.F90:319-327
      const double fx1 = p1 * f11[g] + q1 * f21[g];
      const double fx2 = p1 * f12[g] + q1 * f22[g];
      kc[g] = ten2power_tab(p2 * fx1 + q2 * fx2, s_e2) * col;  // :818 / :828
    }
  };
  double tk[NG];  // tau_k of the running mixture
#pragma unroll
  for (int g = 0; g < NG; g++) tk[g] = 0.0;
  // interpolation brackets of the next species are fetched while the current one is mixed
  int iP_n = c.ix[2 * opf + p.k[0].slotP * nz + j], iT_n = c.ix[2 * opf + p.k[0].slotT * nz + j];
  double q1_n = c.q[opf + p.k[0].slotP * nz + j], q2_n = c.q[opf + p.k[0].slotT * nz + j];
  for (int s = 0; s < p.nk; s++) {
    const int iP = iP_n, iT = iT_n;
    const double q1 = q1_n, q2 = q2_n;
    // the layer index as the loop body sees it: opaque, so that the addresses formed from it are formed here and not
    // once before the loop -- hoisted, they are 64-bit values that have to live across the mixing step's assembly
    // block (192 named registers), and the ones that do not fit are spilled and reloaded every iteration
    int jv = j;
    if constexpr (RM == 0 && RORR_ASM) asm volatile("" : "+v"(jv));
    if (s + 1 < p.nk) {
      const KDev &kn = p.k[s + 1];
      iP_n = c.ix[2 * opf + kn.slotP * nz + jv]; iT_n = c.ix[2 * opf + kn.slotT * nz + jv];
      q1_n = c.q[opf + kn.slotP * nz + jv]; q2_n = c.q[opf + kn.slotT * nz + jv];
    }
    double kc[NG];
    k_times_col(s, jv, iP, iT, q1, q2, kc);
    STAMP(p.stamps, 2 + 3 * s);
    if (s == 0) {
#pragma unroll
      for (int g = 0; g < NG; g++) tk[g] = kc[g];
    } else {
#ifdef CLIMA_STAMPS
      if (tile == 100 && threadIdx.x == 0) g_stamp_buf = p.stamps;
#endif
      double out[NG];
      rorr_mix8<RM>(tk, kc, sI, tid, tile, s_wxy, s_E, s_Ew, rW, out, p.rorr_tab, p.stamps, 21 + s);   // (slots 22..25: 32.. belong to the two-stream blocks)
#pragma unroll
      for (int g = 0; g < NG; g++) tk[g] = out[g];
      STAMP(p.stamps, 4 + 3 * s);
    }
  }
  if (!terms_first) {
    layer_terms(j, lt);
  } else if constexpr (RM == 0 && RORR_ASM) {
    lt.tausg = s_lt[0][tid]; lt.taua = s_lt[1][tid]; lt.tauc = s_lt[2][tid]; lt.tausc = s_lt[3][tid];
    lt.taup = s_lt[4][tid]; lt.tausp = s_lt[5][tid]; lt.gt = s_lt[6][tid];
  }
  STAMP(p.stamps, 20);
#ifdef CLIMA_STAMPS
  if (p.stamps && (tid & 63) == 0) {
    const long w = (long)tile * (OP_THREADS / 64) + (tid >> 6);
    p.stamps[64 + 2 * w] = wt0;
    p.stamps[64 + 2 * w + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // ---- totals (:856-886)
  auto store_layer = [&](const int nn, const LayerTerms &T, const double (&tkv)[NG]) {
    asm volatile("" ::: "memory");
    const double *wbin = s_Ew + NG + 1;
    double tb = 0.0;
    const size_t base = ((size_t)l * NG) * nz + nn;
    const double scat = T.tausg + T.tausp + T.tausc;
    if (p.write_w0) {
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const double tau = T.tausg + T.taua + T.taup + tkv[g] + T.tauc;
        double w0;
        if (tau <= TAU_MIN) w0 = 0.0;
        else w0 = fmin(MAX_W0, scat / tau);
        st_opr<COHERENT>(&p.w0[oo + base + (size_t)g * nz], w0);
      }
    }
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const double tau = T.tausg + T.taua + T.taup + tkv[g] + T.tauc;
      st_opr<COHERENT>(&p.tau[oo + base + (size_t)g * nz], tau);
      tb = tb + tau * wbin[g];
    }
    st_opr<COHERENT>(&p.scat[oo + (size_t)l * nz + nn], scat);
    st_opr<COHERENT>(&p.tau_band[oo + (size_t)l * nz + nn], tb);
    st_opr<COHERENT>(&p.g[oo + (size_t)l * nz + nn], T.gt);
  };
  if (valid) {
    store_layer(n, lt, tk);
    if (pair) {
      // the second layer of the pair (ground-first j+1, TOA-first n-1)
      if (exact) {
        store_layer(n - 1, lt, tk);
      } else {
        LayerTerms lt2;
        layer_terms(j + 1, lt2);
        if (p.nk == 1) {  // no mixing step to copy: k of the source layer times this layer's own column (:818)
          double tk2[NG];
          k_times_col(0, j + 1, c.ix[2 * opf + p.k[0].slotP * nz + j], c.ix[2 * opf + p.k[0].slotT * nz + j], c.q[opf + p.k[0].slotP * nz + j],
                      c.q[opf + p.k[0].slotT * nz + j], tk2);
          store_layer(n - 1, lt2, tk2);
        } else {
          store_layer(n - 1, lt2, tk);
        }
      }
    }
  }
}

template <int RM, bool CUSTOM>
__global__ __launch_bounds__(OP_THREADS, 2) void k_opacity8(OpacityParams p) {
  opacity8_body<RM, CUSTOM, false>(p, (int)blockIdx.x, 0, 0, 0);
}

// ------------------------------------------------------------------------------------
// k_opacity_generic: any g-point count ng <= OPG_MAX_NG (the k-distribution settings allow
// `new_num_k_bins` other than 8).  One wave per (bin, layer); the ng*ng sums of the
// random-overlap step live in LDS and are sorted by a wave-cooperative bitonic network on
// (value, pair index) -- lexicographic, i.e. exactly the stable rank of mrgrnk
// (types.f90:840) -- then weights_to_bins (clima_eqns.f90:43-54) runs serially in rank
// order and every output g-point is rebinned by its own lane with the overlap sum of the
// futils rebin (types.f90:847).  Same arithmetic order as the reference; this is the
// completeness path, k_opacity8 is the tuned one.
// ------------------------------------------------------------------------------------
constexpr int OPG_MAX_NG = 32;

__global__ __launch_bounds__(64) void k_opacity_generic(OpacityParams p, int N2) {
  extern __shared__ __align__(16) double sm[];
  const int ng = p.ng, n2 = ng * ng, nz = p.nz;
  double *s_key = sm;                 // [N2]
  double *s_cum = s_key + N2;         // [n2 + 1] cumulative sorted weights
  double *s_tk = s_cum + (n2 + 1);    // [ng] running mixture
  double *s_kc = s_tk + ng;           // [ng] new species
  double *s_out = s_kc + ng;          // [ng]
  double *s_wxy = s_out + ng;         // [n2] weights of the pairs
  int *s_idx = (int *)(s_wxy + n2);   // [N2]
  const int lane = threadIdx.x;
  for (int m = lane; m < n2; m += 64) s_wxy[m] = p.wxy[m];
  const ColumnDev &c = p.col;
  const long t = blockIdx.x;
  const int l = p.bin_lo + (int)(t / nz);
  const int j = (int)(t % nz);   // ground-first layer
  const int n = nz - 1 - j;      // TOA-first index
  const double dzj = c.dz[j];
  // the second layer of a reused pair takes the rebinned mixture of the first (:833-834);
  // with nk >= 2 its final mixture depends on the first layer's inputs only
  const int jk = (p.nk >= 2) ? c.meta[1 + nz + j] : j;

  // ---- wave-uniform terms (same statements as k_opacity8)
  double tausg = 0.0;
  for (int i = 0; i < p.nray; i++) tausg = tausg + p.ray[i].data[l] * c.cols[p.ray[i].sp1 * nz + j];
  double taua = 0.0;
  for (int e = 0; e < p.nabs; e++) {
    const AbsEntry &x = p.abs[e];
    double sgm;
    if (x.nT) {
      const int ix = c.ix[x.slot * nz + j];
      const double q = c.q[x.slot * nz + j];
      const double *base = x.data + (size_t)l * x.nT + ix;
      sgm = ten2power((1.0 - q) * base[0] + q * base[1]);
    } else {
      sgm = x.data[l];
    }
    taua = taua + sgm * c.absw[e * nz + j];
  }
  double tauc = TINY, tausc = TINY * TINY, g0c = TINY;
  if (p.cust.on) {
    const int ix = c.ix[p.cust.slot * nz + j];
    const double q = c.q[p.cust.slot * nz + j];
    const size_t o = (size_t)l * p.cust.nP;
    tauc = lerp1(p.cust.dtau + o, ix, q) * dzj;
    const double w0c = lerp1(p.cust.w0 + o, ix, q);
    g0c = lerp1(p.cust.g0 + o, ix, q);
    tausc = w0c * tauc;
  }
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
  gt = gt + g0c * tausc / fmax(TAU_MIN, (tausp + tausg + tausc));
  gt = fmin(gt, MAX_GT);

  // ---- k-distributions and random-overlap mixing
  for (int s = 0; s < p.nk; s++) {
    const KDev &kd = p.k[s];
    if (lane < ng) {
      const int iP = c.ix[kd.slotP * nz + jk], iT = c.ix[kd.slotT * nz + jk];
      const double q1 = c.q[kd.slotP * nz + jk], q2 = c.q[kd.slotT * nz + jk];
      const double p1 = 1.0 - q1, p2 = 1.0 - q2;
      const double *f11 = kd.log10k + (size_t)l * kd.nT * kd.nP * ng + ((size_t)iT * kd.nP + iP) * ng;
      const double *f21 = f11 + ng, *f12 = f11 + (size_t)kd.nP * ng, *f22 = f12 + ng;
      const double fx1 = p1 * f11[lane] + q1 * f21[lane];
      const double fx2 = p1 * f12[lane] + q1 * f22[lane];
      const double v = ten2power(p2 * fx1 + q2 * fx2) * c.cols[kd.sp * nz + jk];
      if (s == 0) s_tk[lane] = v;
      else s_kc[lane] = v;
    }
    __syncthreads();
    if (s == 0) continue;
    for (int m = lane; m < N2; m += 64) {
      double v = __longlong_as_double(0x7ff0000000000000LL);  // +inf padding sorts last
      if (m < n2) v = s_tk[m / ng] + s_kc[m % ng];            // tau_xy(:, j+(i-1)*ng), :828
      s_key[m] = v;
      s_idx[m] = m;
    }
    __syncthreads();
    for (int k = 2; k <= N2; k <<= 1) {
      for (int d = k >> 1; d > 0; d >>= 1) {
        for (int u = lane; u < (N2 >> 1); u += 64) {
          const int i = ((u & ~(d - 1)) << 1) | (u & (d - 1));
          const int x = i + d;
          const bool up = (i & k) == 0;
          const double ka = s_key[i], kb = s_key[x];
          const int ia = s_idx[i], ib = s_idx[x];
          const bool gtr = (ka > kb) || (ka == kb && ia > ib);
          if (gtr == up) { s_key[i] = kb; s_key[x] = ka; s_idx[i] = ib; s_idx[x] = ia; }
        }
        __syncthreads();
      }
    }
    {
      // weights_to_bins (clima_eqns.f90:43-54): cumulative weights in rank order.  Every lane sums a
      // run of consecutive ranks, an exclusive wave scan of the run totals gives its offset (a
      // serial sum over all ng^2 ranks by one lane was 95 % of this kernel's time at ng = 16)
      const int seg = (N2 + 63) / 64;
      double tot = 0.0;
      for (int q = 0; q < seg; q++) {
        const int m = lane * seg + q;
        if (m < n2) tot = tot + s_wxy[s_idx[m]];
      }
      double incl = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double nb = __shfl_up(incl, d);
        if (lane >= d) incl = incl + nb;
      }
      double cum = incl - tot;
      if (lane == 0) { cum = 0.0; s_cum[0] = 0.0; }
      for (int q = 0; q < seg; q++) {
        const int m = lane * seg + q;
        if (m < n2) { cum = cum + s_wxy[s_idx[m]]; s_cum[m + 1] = cum; }
      }
    }
    __syncthreads();
    if (lane < ng) {
      const double b0 = p.wbin_e[lane], b1 = p.wbin_e[lane + 1];
      int lo = 0, hi = n2;  // first old bin whose upper edge is above b0
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_cum[mid + 1] <= b0) lo = mid + 1; else hi = mid; }
      double acc = 0.0;
      for (int m = lo; m < n2; m++) {
        const double e0 = s_cum[m], e1 = s_cum[m + 1];
        if (e0 >= b1) break;
        const double a = e0 > b0 ? e0 : b0, b = e1 < b1 ? e1 : b1;
        if (b > a) acc = acc + (b - a) * s_key[m];
      }
      s_out[lane] = acc / (b1 - b0);
    }
    __syncthreads();
    if (lane < ng) s_tk[lane] = s_out[lane];
    __syncthreads();
  }

  // ---- totals (:856-886)
  if (lane < ng) {
    const double tau = tausg + taua + taup + s_tk[lane] + tauc;
    double w0;
    if (tau <= TAU_MIN) w0 = 0.0;
    else w0 = fmin(MAX_W0, (tausg + tausp + tausc) / tau);
    const size_t o = ((size_t)l * ng + lane) * nz + n;
    p.tau[o] = tau;
    p.w0[o] = w0;
    s_out[lane] = tau;
  }
  __syncthreads();
  if (lane == 0) {
    double tb = 0.0;
    for (int g = 0; g < ng; g++) tb = tb + s_out[g] * p.wbin[g];
    p.tau_band[(size_t)l * nz + n] = tb;
    p.g[(size_t)l * nz + n] = gt;
  }
}

// ------------------------------------------------------------------------------------
// k_opacity_coop<NG>: NG lanes per (bin, source layer), NG = 8, 16 or 32 g-points.  Lane g of a
// group owns g-point g: it interpolates that g-point's k-coefficient, and in a mixing step it holds
// row g of the NG x NG sums (x_g + y_r, r = 0..NG-1) in registers.  The NG*NG keys are sorted ACROSS
// the group by a bitonic network in its direction-free form (the first step of every merge level
// mirrors the partner index, so all runs ascend): position = lane*NG + register; partners at a
// distance below NG are registers of the same lane (one v_min + one v_max), partners further away sit
// in the same (or the mirrored) register of lane^m and come over ds_swizzle.  Rows that already
// ascend (k-coefficients ascending in g: the normal case) skip the levels that sort within a lane.
// Rebin (weights_to_bins + futils rebin, types.f90:846-847) on the distributed sorted sequence: a
// lane's keys are a contiguous run of ranks, so the running weight C and integral IC at its first key
// come from an exclusive scan of the lanes' totals; output edge E_k is crossed inside exactly one
// lane's run ((C_first, C_next_lane]), which evaluates I(E_k) = max_r [IC_r + v_r (E_k - C_r)] over its
// keys (the window form of rorr_mix8) and hands it to lanes k-1 and k through LDS.
// Two uses: (1) g-point counts 16 and 32 (NG^2 = 256 / 1024 keys do not fit one lane's registers);
// (2) NG = 8 when a call has few (bin, layer) items -- a bin-sharded rank, a short column: the
// lane-per-item kernel's 4 mixing steps are a ~30 us dependent chain however few items there are, this
// form's chain is a fifth of that at 2.4x the total work.
// Keys carry their pair index in the 2*log2(NG) low mantissa bits (ties ordered as the stable rank).
// ------------------------------------------------------------------------------------
// v of lane (lane ^ M), M < 32.  Masks 1, 2, 3 are quad permutations, 7 and 15 the half-row / row mirrors:
// DPP moves in the VALU; the others go over the LDS crossbar (ds_swizzle, bit-mask mode) -- which is
// what bounded the 16-g-point kernel when every exchange went that way.
template <int M>
__device__ __forceinline__ double swz_xor(double v) {
  const long long b = __double_as_longlong(v);
  int lo, hi;
  constexpr int ctrl = M == 1 ? 0xB1 : M == 2 ? 0x4E : M == 3 ? 0x1B : M == 7 ? 0x141 : M == 15 ? 0x140 : -1;
  if constexpr (ctrl >= 0) {
    lo = __builtin_amdgcn_mov_dpp((int)b, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), ctrl, 0xf, 0xf, true);
  } else {
    lo = __builtin_amdgcn_ds_swizzle((int)b, (M << 10) | 0x1f);
    hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), (M << 10) | 0x1f);
  }
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// butterfly sum over the NG lanes of a group
template <int NG, int M = 1>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (M < NG) return group_sum<NG, 2 * M>(v + swz_xor<M>(v));
  else return v;
}

// waves per SIMD the kernel is compiled for (__launch_bounds__' second argument): its lane exchanges make it
// latency-bound, so they count
// (measured, profiles/r03_coop_ab.txt: 16 g-points at three waves per SIMD instead of two -- 168 registers, 14-22
// scratch accesses outside the sort -- 610 -> 541 us; 8 g-points forced from three to four: slower, left alone;
// 32 g-points at two instead of one: 4.67 -> 2.92 ms)
#ifndef COOP_WAVES_8
#define COOP_WAVES_8 1
#endif
#ifndef COOP_WAVES_16
#define COOP_WAVES_16 3
#endif
#ifndef COOP_WAVES_32
#define COOP_WAVES_32 2
#endif
#define COOP_MIN_WAVES(NG) ((NG) == 8 ? COOP_WAVES_8 : (NG) == 16 ? COOP_WAVES_16 : COOP_WAVES_32)
template <int NG>
struct CoopSort {
  // one step with partner position pos ^ X restricted to partners in another lane: lane ^ LM, register
  // r ^ RM (RM = NG-1 mirrors the register index, 0 keeps it); `lower`: this lane holds the lower position
  template <int LM, int RM>
  static __device__ __forceinline__ void cross(double (&key)[NG], const bool lower) {
    double b[NG];
#pragma unroll
    for (int r = 0; r < NG; r++) b[r] = swz_xor<LM>(key[r ^ RM]);
    // the lane of the lower position keeps the minima, the other one the maxima: one v_min_f64 or one v_max_f64
    // per key under the lanes' exec masks (a compare and a 64-bit select per key took twice the instructions)
    if (lower) {
#pragma unroll
      for (int r = 0; r < NG; r++) key[r] = dmin(key[r], b[r]);
    } else {
#pragma unroll
      for (int r = 0; r < NG; r++) key[r] = dmax(key[r], b[r]);
    }
  }
  template <int X>
  static __device__ __forceinline__ void intra(double (&key)[NG]) {  // partner register r ^ X, X < NG
#pragma unroll
    for (int r = 0; r < NG; r++) {
      if ((r ^ X) > r) {
        const double lo = dmin(key[r], key[r ^ X]), hi = dmax(key[r], key[r ^ X]);
        key[r] = lo;
        key[r ^ X] = hi;
      }
    }
  }
  // half-cleaner steps of level K: distances K/4, K/8, ..., 1
  template <int K, int D>
  static __device__ __forceinline__ void halves(double (&key)[NG], const int g) {
    if constexpr (D >= 1) {
      if constexpr (D >= NG) cross<D / NG, 0>(key, (g & (D / NG)) == 0);
      else intra<D>(key);
      halves<K, D / 2>(key, g);
    }
  }
  template <int K>
  static __device__ __forceinline__ void level(double (&key)[NG], const int g) {
    // flip step: partner = pos ^ (K-1)
    if constexpr (K <= NG) intra<K - 1>(key);
    else cross<K / NG - 1, NG - 1>(key, (g & (K / (2 * NG))) == 0);
    halves<K, K / 4>(key, g);
  }
  template <int K, int KMAX>
  static __device__ __forceinline__ void levels(double (&key)[NG], const int g) {
    if constexpr (K <= KMAX) {
      level<K>(key, g);
      levels<2 * K, KMAX>(key, g);
    }
  }
};

template <int NG, bool CUSTOM>
__global__ __launch_bounds__(OP_THREADS, COOP_MIN_WAVES(NG)) void k_opacity_coop(OpacityParams p) {
  constexpr int N2 = NG * NG, GROUPS = OP_THREADS / NG;
  constexpr unsigned long long IDX_MASK = (unsigned long long)(N2 - 1);
  __shared__ double s_wxy[N2];
  __shared__ double s_E[NG + 1];
  __shared__ double sIe[GROUPS][NG + 1];
  const int tid = threadIdx.x;
  // ng <= NG g-points (round 3: 12 runs in the 16-lane kernel, 4 and 6 in the 8-lane one, 20-28 in the 32-lane one
  // instead of the wave-per-item generic kernel): the lanes g >= ng of a group carry coefficients of PAD_K -- finite, far
  // above any optical depth -- and pair weights of 0, so their sums sort behind the real ones, add nothing to the running
  // weight or integral, and every output edge k <= ng is crossed inside the real data; they store nothing.
  const int ng = p.ng;
  constexpr double PAD_K = 1.0e290;
  for (int m = tid; m < N2; m += OP_THREADS) {
    const int gi = m / NG, ri = m % NG;
    s_wxy[m] = (gi < ng && ri < ng) ? p.wxy[gi * ng + ri] : 0.0;
  }
  if (tid <= NG) s_E[tid] = tid <= ng ? p.wbin_e[tid] : __longlong_as_double(0x7ff0000000000000LL);
  __syncthreads();
  const ColumnDev &c = p.col;
  const int nz = p.nz;
  const int g = tid & (NG - 1), grp = tid / NG;
  const bool g_on = g < ng;
  const int nsrc = c.meta[0];
  const long total = (long)p.nbins * nsrc;
  long t = ((long)blockIdx.x * OP_THREADS + tid) / NG;
  const bool valid = t < total;
  if (!valid) t = total - 1;
  const int l = p.bin_lo + (int)(t / nsrc);
  const int ent = c.meta[1 + (int)(t % nsrc)];
  const int j = ent & SRC_LAYER;
  const bool pair = (ent & SRC_PAIR) != 0, exact = (ent & SRC_EXACT) != 0;
  const int n = nz - 1 - j;
  const double wg = g_on ? p.wbin[g] : 0.0;
  const double rWg = g_on ? 1.0 / (s_E[g + 1] - s_E[g]) : 0.0;

  // ---- layer terms (types.f90:665-757), shared out over the lanes of the group: lane g takes the
  //      Rayleigh species and continuum entries g, g+NG, ...; the partial sums meet in a butterfly
  //      (their order of addition differs from the reference's: rounding only).  Particle and custom
  //      terms are few and every lane evaluates them.
  auto layer_terms = [&](const int jl, LayerTerms &o) {
    const double dzj = c.dz[jl];
    double tausg = 0.0;
    for (int i = g; i < p.nray; i += NG) tausg = tausg + p.ray[i].data[l] * c.cols[p.ray[i].sp1 * nz + jl];
    tausg = group_sum<NG>(tausg);
    double taua = 0.0;
    for (int e = g; e < p.nabs; e += NG) {
      const AbsEntry &x = p.abs[e];
      double sgm;
      if (x.nT) {
        const int ix = c.ix[x.slot * nz + jl];
        const double q = c.q[x.slot * nz + jl];
        const double *base = x.data + (size_t)l * x.nT + ix;
        sgm = ten2power((1.0 - q) * base[0] + q * base[1]);
      } else {
        sgm = x.data[l];
      }
      taua = taua + sgm * c.absw[e * nz + jl];
    }
    taua = group_sum<NG>(taua);
    double tauc = TINY, tausc = TINY * TINY, g0c = TINY;
    if constexpr (CUSTOM) {
      const int ix = c.ix[p.cust.slot * nz + jl];
      const double q = c.q[p.cust.slot * nz + jl];
      const size_t o2 = (size_t)l * p.cust.nP;
      tauc = lerp1(p.cust.dtau + o2, ix, q) * dzj;
      const double w0c = lerp1(p.cust.w0 + o2, ix, q);
      g0c = lerp1(p.cust.g0 + o2, ix, q);
      tausc = w0c * tauc;
    }
    double tausp = 0.0, taup = 0.0;
    double tausp_1[MAX_PART], gtp[MAX_PART];
    for (int i = 0; i < p.npart; i++) {
      const PartDev &pt = p.part[i];
      const int ix = c.ix[pt.slot * nz + jl];
      const double q = c.q[pt.slot * nz + jl];
      const double w0p = lerp1(pt.w0 + (size_t)l * pt.nrad, ix, q);
      const double qext = lerp1(pt.qext + (size_t)l * pt.nrad, ix, q);
      gtp[i] = lerp1(pt.gt + (size_t)l * pt.nrad, ix, q);
      const double rr = c.radii[pt.p_ind * nz + jl];
      const double taup_1 = qext * PI * (rr * rr) * c.pdens[pt.p_ind * nz + jl] * dzj;
      taup = taup + taup_1;
      tausp_1[i] = w0p * taup_1;
      tausp = tausp + tausp_1[i];
    }
    double gt = 0.0;
    for (int i = 0; i < p.npart; i++) gt = gt + gtp[i] * tausp_1[i] / fmax(TAU_MIN, (tausp + tausg + tausc));
    gt = gt + g0c * tausc / fmax(TAU_MIN, (tausp + tausg + tausc));
    gt = fmin(gt, MAX_GT);
    o.tausg = tausg; o.taua = taua; o.tauc = tauc; o.tausc = tausc; o.taup = taup; o.tausp = tausp; o.gt = gt;
  };
  // k-coefficient of species s at this lane's g-point times the column of layer jl (:649-662, :818 / :828)
  auto k_times_col = [&](const int s, const int jl) {
    const KDev &kd = p.k[s];
    const int iP = c.ix[kd.slotP * nz + j], iT = c.ix[kd.slotT * nz + j];
    const double q1 = c.q[kd.slotP * nz + j], q2 = c.q[kd.slotT * nz + j];
    const double p1 = 1.0 - q1, p2 = 1.0 - q2;
    const double *f11 = kd.log10k + (size_t)l * kd.nT * kd.nP * ng + ((size_t)iT * kd.nP + iP) * ng;
    const double *f21 = f11 + ng, *f12 = f11 + (size_t)kd.nP * ng, *f22 = f12 + ng;
    const int gq = min(g, ng - 1);
    const double fx1 = p1 * f11[gq] + q1 * f21[gq];
    const double fx2 = p1 * f12[gq] + q1 * f22[gq];
    const double kv = ten2power(p2 * fx1 + q2 * fx2) * c.cols[kd.sp * nz + jl];
    return g_on ? kv : PAD_K;
  };
  const int gbase = (tid & 63) & ~(NG - 1);  // first lane of the group within the wave
  double tk = k_times_col(0, j);
  for (int s = 1; s < p.nk; s++) {
    const double kc = k_times_col(s, j);
    // ---- the row of this lane: x_g + y_r (tau_xy(:, r+(g-1)*ng), types.f90:828), pair index in the low bits
    double key[NG];
#pragma unroll
    for (int r = 0; r < NG; r++) {
      const double v = tk + __shfl(kc, gbase + r);
      unsigned long long b = (unsigned long long)__double_as_longlong(v);
      b = (b & ~IDX_MASK) | (unsigned long long)(g * NG + r);
      key[r] = __longlong_as_double((long long)b);
    }
    // rows ascend when y does; otherwise sort within the lanes first
    double kc_next;
    if constexpr (NG <= 16) kc_next = row_shl<1>(kc, kc);   // (lane 15 of a row keeps its own: it is a group's last lane)
    else kc_next = __shfl_down(kc, 1);
    const bool ys = __all(g == NG - 1 || kc <= kc_next);
    if (!ys) CoopSort<NG>::template levels<2, NG>(key, g);
    CoopSort<NG>::template levels<2 * NG, N2>(key, g);
    // ---- rebin: weights in sorted order, running weight / integral at this lane's first key
    double w[NG];
#pragma unroll
    for (int r = 0; r < NG; r++) w[r] = s_wxy[(int)((unsigned long long)__double_as_longlong(key[r]) & IDX_MASK)];
    double Cw = 0.0, ICw = 0.0;
#pragma unroll
    for (int r = 0; r < NG; r++) { ICw = __builtin_fma(key[r], w[r], ICw); Cw = Cw + w[r]; }
    double Cs = Cw, ICs = ICw;  // inclusive scan over the lanes of the group: DPP shifts within the 16-lane row (a
                                // group is half a row, a row, or two rows: the second row of NG = 32 then adds the
                                // first row's total, fetched over the crossbar)
    auto scan_step = [&](const double a, const double b2, const int d) {
      if ((g & 15) >= d) { Cs = Cs + a; ICs = ICs + b2; }
    };
    scan_step(row_shr<1>(0.0, Cs), row_shr<1>(0.0, ICs), 1);
    scan_step(row_shr<2>(0.0, Cs), row_shr<2>(0.0, ICs), 2);
    scan_step(row_shr<4>(0.0, Cs), row_shr<4>(0.0, ICs), 4);
    if constexpr (NG >= 16) scan_step(row_shr<8>(0.0, Cs), row_shr<8>(0.0, ICs), 8);
    if constexpr (NG >= 32) {
      const double c15 = __shfl(Cs, gbase + 15), ic15 = __shfl(ICs, gbase + 15);
      if (g >= 16) { Cs = Cs + c15; ICs = ICs + ic15; }
    }
    const double Cbase = Cs - Cw, ICbase = ICs - ICw;
    double hiC;   // the next lane's first running weight
    if constexpr (NG <= 16) hiC = row_shl<1>(0.0, Cbase);
    else hiC = __shfl_down(Cbase, 1);
    if (g == NG - 1) hiC = __longlong_as_double(0x7ff0000000000000LL);
    if (g == NG - 1) sIe[grp][ng] = ICs;   // the last edge is the total weight
    if (g == 0) sIe[grp][0] = 0.0;
    // the output edges crossed inside this lane's run of ranks: Cbase < E_k <= hiC
    int k = 1;
    while (k < ng && s_E[k] <= Cbase) k++;
    for (; k < ng; k++) {
      const double Ek = s_E[k];
      if (!(Ek <= hiC)) break;
      double C = Cbase, IC = ICbase, I = -1.0e300;
#pragma unroll
      for (int r = 0; r < NG; r++) {
        I = dmax(I, __builtin_fma(key[r], Ek - C, IC));
        IC = __builtin_fma(key[r], w[r], IC);
        C = C + w[r];
      }
      sIe[grp][k] = I;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the group's lanes are lanes of this wave
    tk = g_on ? (sIe[grp][g + 1] - sIe[grp][g]) * rWg : PAD_K;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the next mixing step rewrites the slots
  }

  // the layer's other terms are needed by the totals only: evaluated here, they do not sit in registers through the
  // mixing steps
  LayerTerms lt;
  layer_terms(j, lt);

  // ---- totals (:856-886)
  auto store_layer = [&](const int nn, const LayerTerms &T, const double tkv) {
    const double tau = T.tausg + T.taua + T.taup + tkv + T.tauc;
    double w0;
    if (tau <= TAU_MIN) w0 = 0.0;
    else w0 = fmin(MAX_W0, (T.tausg + T.tausp + T.tausc) / tau);
    const double tb = group_sum<NG>(g_on ? tau * wg : 0.0);
    if (valid && g_on) {
      const size_t o = ((size_t)l * ng + g) * nz + nn;
      p.tau[o] = tau;
      p.w0[o] = w0;
      if (g == 0) {
        p.tau_band[(size_t)l * nz + nn] = tb;
        p.g[(size_t)l * nz + nn] = T.gt;
      }
    }
  };
  store_layer(n, lt, tk);
  if (__any(pair)) {
    LayerTerms lt2 = lt;
    double tk2 = tk;
    const int j2 = pair ? j + 1 : j;
    if (pair && !exact) {
      layer_terms(j2, lt2);
      if (p.nk == 1) tk2 = k_times_col(0, j2);   // no mixing step to copy: k of the source layer times this layer's own column
    }
    const bool keep = valid && pair;
    // (store_layer shuffles: every lane takes part; lanes without a pair write nothing)
    const double tau = lt2.tausg + lt2.taua + lt2.taup + tk2 + lt2.tauc;
    double w0;
    if (tau <= TAU_MIN) w0 = 0.0;
    else w0 = fmin(MAX_W0, (lt2.tausg + lt2.tausp + lt2.tausc) / tau);
    const double tb = group_sum<NG>(g_on ? tau * wg : 0.0);
    if (keep && g_on) {
      const size_t o = ((size_t)l * ng + g) * nz + (n - 1);
      p.tau[o] = tau;
      p.w0[o] = w0;
      if (g == 0) {
        p.tau_band[(size_t)l * nz + (n - 1)] = tb;
        p.g[(size_t)l * nz + (n - 1)] = lt2.gt;
      }
    }
  }
}

template <int NG>
static void launch_coop(const OpacityParams &p, hipStream_t s) {
  const long lanes = (long)p.nbins * p.nsrc * NG;
  const int grid = (int)((lanes + OP_THREADS - 1) / OP_THREADS);
  if (p.cust.on) hipLaunchKernelGGL((k_opacity_coop<NG, true>), dim3(grid), dim3(OP_THREADS), 0, s, p);
  else hipLaunchKernelGGL((k_opacity_coop<NG, false>), dim3(grid), dim3(OP_THREADS), 0, s, p);
}

bool launch_opacity(const OpacityParams &p, hipStream_t s) {
  long total = (long)p.nbins * p.nz;
  // every g-point count but the tuned 8 goes to the group-of-lanes kernel with the next power of two of lanes per item
  // (OpacityParams::generic, CLIMA_HIP_GENERIC=1 when the handle is made: the wave-per-item generic kernel instead --
  // the same arithmetic order as the reference, kept as a cross-check)
  if (p.ng >= 1 && p.ng <= 32 && (p.ng != 8 ? !p.generic : p.coop != 0) && (long)p.nbins * p.nsrc > 0) {
    if (p.ng <= 8) launch_coop<8>(p, s);
    else if (p.ng <= 16) launch_coop<16>(p, s);
    else launch_coop<32>(p, s);
    return true;
  }
  if (p.ng != 8) {
    if (p.ng < 1 || p.ng > OPG_MAX_NG) return false;
    if (total <= 0) return true;
    int N2 = 2;
    while (N2 < p.ng * p.ng) N2 <<= 1;
    const size_t lds = sizeof(double) * ((size_t)N2 + 2 * p.ng * p.ng + 1 + 3 * p.ng) + sizeof(int) * (size_t)N2;
    hipLaunchKernelGGL(k_opacity_generic, dim3((unsigned)total), dim3(64), lds, s, p, N2);
    return true;
  }
  total = (long)p.nbins * p.nsrc;  // one lane per (bin, source layer)
  if (total <= 0) return true;
  const int grid = (int)((total + OP_THREADS - 1) / OP_THREADS);
  using Kern = void (*)(OpacityParams);
  static const Kern kern[2][3] = {{k_opacity8<0, false>, k_opacity8<1, false>, k_opacity8<2, false>},
                                  {k_opacity8<0, true>, k_opacity8<1, true>, k_opacity8<2, true>}};
  hipLaunchKernelGGL(kern[p.cust.on ? 1 : 0][p.rebin_mode], dim3(grid), dim3(OP_THREADS), 0, s, p);
  return true;
}

// ------------------------------------------------------------------------------------
// k_twostream
// ------------------------------------------------------------------------------------
//
// One workgroup per (channel, bin[, g-point group]).  Work items are (layer i, column c)
// pairs, p = i*nc + c (layer TOA-first, nc g-point columns in this block).  Unknown
// ordering of the 2nz-row tridiagonal system follows the reference (SURVEY.md Appendix A,
// twostream.f90:91-117, :249-275): rows 2i and 2i+1 (0-based) are y1_i and y2_i, so layer
// i "owns" rows 2i, 2i+1:
//     row 2i   (i>=1): couples layers i-1,i   (Fortran odd rows  l=2(i-1)+1, :97-103)
//     row 2i+1 (i<nz-1): couples layers i,i+1 (Fortran even rows l=2i,       :106-112)
//     row 0: top boundary (:93-96);  row 2nz-1: surface boundary (:113-117)
//
// LDS image: 6 arrays [nz][nc] f64, recycled through the phases
//     G  : cap_gamma          -> c' row 2i    -> beta  row 2i    -> staging fup
//     X  : fast_exp(-lambda tau)   -> c' row 2i+1  -> beta  row 2i+1  -> staging fdn
//     E0 : tau'/tauc, cp0     -> E  row 2i    -> d' -> alpha row 2i -> staging amean
//     E1 : cm0                -> E  row 2i+1  -> d' -> alpha row 2i+1
//     U  :                       l  row 2i    -> gamma row 2i
//     V  :                       l  row 2i+1  -> gamma row 2i+1
//
// The solve is a domain-decomposed Thomas factorisation executed by ONE wave: every
// column is cut into S chunks of consecutive layers, lane (chunk j, column c) solves
// its chunk as a two-stream problem of its own (see dd_solve), the chunks are joined with
// wave shuffles, and all threads finish their own rows in parallel.  Same elimination
// arithmetic as tridiag (twostream.f90:297-316) inside a chunk; the dependent chain is S
// times shorter.

constexpr int TS_MAXP = 4;  // (layer, column) pairs per thread

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

struct TsLds {
  double *G, *X, *E0, *E1, *U, *V;
  double *Bp;    // [nz+1] Planck
  double *L0;    // [3][nc] level-0 values
  double *Seg;   // [nseg][nc] scan segment totals
  double *Bnd;   // [2][S][nc] chunk parameters: Uin_j (up-flux entering at the bottom), Din_j
  double *Ch;    // [4][S][nc] chunk constants: cp0,cm0 of its first layer, cpb,cmb of its last
  int *chunk;    // [nz] chunk index of each layer
};

// chunk of layer i when nz layers are cut into S chunks [a_j,b_j), a_j = j*nz/S
__device__ __forceinline__ int chunk_of(int i, int nz, int S) {
  int j = ((i + 1) * S - 1) / nz;
  while ((j * nz) / S > i) j--;
  while (((j + 1) * nz) / S <= i) j++;
  return j;
}
// pair index -> (layer, column)
__device__ __forceinline__ void pair_split(int pr, int nc, int sh, int &i, int &c) {
  if (sh >= 0) { i = pr >> sh; c = pr & (nc - 1); }
  else { i = pr / nc; c = pr - i * nc; }
}

// Domain-decomposed two-stream solve.  Chunk j (layers [a,b)) is solved as a two-stream
// problem of its own with flux boundary conditions
//     top:    downward diffuse flux entering layer a   = Din_j   (row 2a:   -Din + e1 y1 - e2 y2 = -cm0_a)
//     bottom: upward flux entering layer b-1 from below = Uin_j  (row 2b-1:  e1 y1 + e2 y2 - Uin = -cpb_{b-1})
// (the first chunk keeps the reference's TOA row :93-96, the last its surface row :113-117),
// i.e. the interface rows 2b-1, 2b of the global system are replaced by the two flux
// conditions they encode.  Every chunk is a well-posed Toon system, so the Thomas sweeps
// (tridiag, twostream.f90:297-316) keep their pivots; one downward sweep carries the
// source column d' and the Din column l, the upward sweep yields for every row
//     y_r = alpha_r + beta_r*Uin_j + gamma_r*Din_j.
// Chunks are joined by flux continuity (adding method):
//     Din_{j+1} = fdn at the bottom of chunk j = dS + dD*Din_j + dU*Uin_j
//     Uin_j     = fup at the top of chunk j+1  = uS + uD*Din_{j+1} + uU*Uin_{j+1}
// resolved per column with wave shuffles (one lane per chunk).
__device__ void dd_solve(const TsLds &s, const int nz, const int nc, const int S,
                         const int lane, const double Rsfc) {
  const int j = lane / nc, c = lane - j * nc;
  const int a = (j * nz) / S, b = ((j + 1) * nz) / S;  // layers [a,b)
  const int len = b - a;
  E4 u = make_e(0.0, 0.0);                               // layer i-1 (unused at i == a)
  E4 v = make_e(s.G[a * nc + c], s.X[a * nc + c]);       // layer i
  const E4 ea = v;
  // ---- downward elimination: y_r + c' y_{r+1} + l*Din = d'
  double cp = 0.0, dp = 0.0, lp = -1.0;
  for (int t = 0; t < len; t++) {
    const int i = a + t;
    E4 w = v;  // layer i+1 (only needed inside the chunk)
    if (i + 1 < b) w = make_e(s.G[(i + 1) * nc + c], s.X[(i + 1) * nc + c]);
    const double Ea = s.E0[i * nc + c], Eb = s.E1[i * nc + c];
    double A, B, D;
    // row 2i
    if (t == 0) { A = (a == 0) ? 0.0 : -1.0; B = v.e1; D = -v.e2; }
    else { A = u.e2 * u.e3 - u.e4 * u.e1; B = u.e1 * v.e1 - u.e3 * v.e3; D = u.e3 * v.e4 - u.e1 * v.e2; }
    double r = rcp_nr(B - A * cp);
    double cn = D * r, dn = (Ea - A * dp) * r, ln = (-A * lp) * r;
    s.G[i * nc + c] = cn; s.E0[i * nc + c] = dn; s.U[i * nc + c] = ln;
    // row 2i+1
    if (t == len - 1) {
      if (b == nz) { A = v.e1 - Rsfc * v.e3; B = v.e2 - Rsfc * v.e4; D = 0.0; }
      else { A = v.e1; B = v.e2; D = -1.0; }
    } else {
      A = w.e2 * v.e1 - v.e3 * w.e4; B = v.e2 * w.e2 - v.e4 * w.e4; D = w.e1 * w.e4 - w.e2 * w.e3;
    }
    r = rcp_nr(B - A * cn);
    cp = D * r; dp = (Eb - A * dn) * r; lp = (-A * ln) * r;
    s.X[i * nc + c] = cp; s.E1[i * nc + c] = dp; s.V[i * nc + c] = lp;
    u = v; v = w;
  }
  const E4 eb = u;  // layer b-1
  // ---- upward: y_r = alpha_r + beta_r*Uin + gamma_r*Din
  double al = 0.0, be = 1.0, ga = 0.0;
  double aB0 = 0, bB0 = 0, gB0 = 0, aB1 = 0, bB1 = 0, gB1 = 0;  // rows 2b-2, 2b-1
  double aA1 = 0, bA1 = 0, gA1 = 0;                              // row 2a+1
  for (int t = len - 1; t >= 0; t--) {
    const int i = a + t;
    double cc = s.X[i * nc + c], dd = s.E1[i * nc + c], ll = s.V[i * nc + c];
    al = dd - cc * al; be = -cc * be; ga = -ll - cc * ga;
    s.E1[i * nc + c] = al; s.X[i * nc + c] = be; s.V[i * nc + c] = ga;
    if (t == len - 1) { aB1 = al; bB1 = be; gB1 = ga; }
    if (t == 0) { aA1 = al; bA1 = be; gA1 = ga; }
    cc = s.G[i * nc + c]; dd = s.E0[i * nc + c]; ll = s.U[i * nc + c];
    al = dd - cc * al; be = -cc * be; ga = -ll - cc * ga;
    s.E0[i * nc + c] = al; s.G[i * nc + c] = be; s.U[i * nc + c] = ga;
    if (t == len - 1) { aB0 = al; bB0 = be; gB0 = ga; }
  }
  // outgoing fluxes of the chunk as affine functions of (Din, Uin):
  //   up at its top    (fup(1)-form, :143):  y1 e3 - y2 e4 + cp0   (first layer)
  //   down at its bottom (fdn(i+1), :147):   y1 e3 + y2 e4 + cmb   (last layer)
  const double cp0a = s.Ch[(0 * S + j) * nc + c], cmbb = s.Ch[(3 * S + j) * nc + c];
  const double uS = al * ea.e3 - aA1 * ea.e4 + cp0a, uD = ga * ea.e3 - gA1 * ea.e4, uU = be * ea.e3 - bA1 * ea.e4;
  const double dS = aB0 * eb.e3 + aB1 * eb.e4 + cmbb, dD = gB0 * eb.e3 + gB1 * eb.e4, dU = bB0 * eb.e3 + bB1 * eb.e4;
  // ---- adding recursion, identical in every lane of a column.
  // bottom-up: Uin_{k-1} = rho_{k-1}*Din_k + sig_{k-1}
  double rho = 0.0, sig = 0.0;            // for interface below chunk k (none below the last chunk)
  double my_rho = 0.0, my_sig = 0.0;      // relation Uin_j = my_rho*Din_{j+1} + my_sig
  for (int k = S - 1; k >= 1; k--) {
    const int src = k * nc + c;
    const double kuS = __shfl(uS, src), kuD = __shfl(uD, src), kuU = __shfl(uU, src);
    const double kdS = __shfl(dS, src), kdD = __shfl(dD, src), kdU = __shfl(dU, src);
    if (k == j) { my_rho = rho; my_sig = sig; }
    // Uin_k = m*(rho*dS + sig + rho*dD*Din_k),  m = 1/(1 - rho*dU)
    const double m = rcp_nr(1.0 - rho * kdU);
    const double n_sig = kuS + kuU * m * (rho * kdS + sig);
    const double n_rho = kuD + kuU * m * rho * kdD;
    rho = n_rho; sig = n_sig;
  }
  if (j == 0) { my_rho = rho; my_sig = sig; }
  // top-down: Din_0 = 0
  double Din = 0.0, myDin = 0.0, myUin = 0.0;
  for (int k = 0; k < S; k++) {
    const int src = k * nc + c;
    const double krho = __shfl(my_rho, src), ksig = __shfl(my_sig, src);
    const double kdS = __shfl(dS, src), kdD = __shfl(dD, src), kdU = __shfl(dU, src);
    const double m = rcp_nr(1.0 - krho * kdU);
    const double Uin = (k == S - 1) ? 0.0 : m * (krho * kdS + ksig + krho * kdD * Din);
    if (k == j) { myDin = Din; myUin = Uin; }
    Din = kdS + kdD * Din + kdU * Uin;
  }
  s.Bnd[(0 * S + j) * nc + c] = myUin;
  s.Bnd[(1 * S + j) * nc + c] = myDin;
}

template <int MAXT, int MINW>
__global__ __launch_bounds__(MAXT, MINW) void k_twostream(TwoStreamParams p) {
  extern __shared__ __align__(16) double lds[];
  const int nz = p.nz, ng = p.ng, nc = p.ncols, S = p.nchunks;
  const int cg0 = blockIdx.y * nc;  // first g-point column of this block
  const int npairs = nz * nc;
  TsLds s;
  s.G = lds; s.X = s.G + npairs; s.E0 = s.X + npairs; s.E1 = s.E0 + npairs; s.U = s.E1 + npairs; s.V = s.U + npairs;
  s.Bp = s.V + npairs; s.L0 = s.Bp + (nz + 1); s.Seg = s.L0 + 3 * nc; s.Bnd = s.Seg + ((nz + 7) / 8) * nc;
  s.Ch = s.Bnd + 2 * S * nc;
  s.chunk = (int *)(s.Ch + 4 * S * nc);
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < nz; i += nt) s.chunk[i] = chunk_of(i, nz, S);  // read after the next barrier
  const int sh = p.nc_shift;  // log2(nc) when nc is a power of two, else -1
  const bool solar = (int)blockIdx.x < p.n_sol;
  const int ll = solar ? p.sol_lo + (int)blockIdx.x : p.ir_lo + ((int)blockIdx.x - p.n_sol);
  const int l = (solar ? p.sol_start : p.ir_start) + ll;  // opacity bin (radiate.f90:57)
  const double *tauL = p.tau + ((size_t)l * ng + cg0) * nz;
  const double *w0L = p.w0 + ((size_t)l * ng + cg0) * nz;
  const double *gL = p.g + (size_t)l * nz;

  // values each thread keeps for its pairs across the solve
  double kG[TS_MAXP], kx[TS_MAXP], kcpb[TS_MAXP], kcmb[TS_MAXP], kdir[TS_MAXP], kdiru[TS_MAXP];
  double cp0_first = 0.0;  // cp0 of layer 0 (pair k=0 of threads tid<nc)
  double Rsfc;

  if (solar) {
    // =========================== solar (two_stream_solar, twostream.f90:10-154) ======
    Rsfc = p.albedo[ll];
    double ktau[TS_MAXP], kw0[TS_MAXP], kgt[TS_MAXP], klam[TS_MAXP];
    const double sqrt3 = 1.7320508075688772;  // sqrt(3.0_dp)
#pragma unroll
    for (int k = 0; k < TS_MAXP; k++) {
      const int pr = tid + k * nt;
      if (pr < npairs) {
        int i, c;
        pair_split(pr, nc, sh, i, c);
        const double tau_in = tauL[c * nz + i], w0_in = w0L[c * nz + i], gt_in = gL[i];
        // delta-Eddington (:38-40)
        const double tau = tau_in * (1.0 - w0_in * gt_in * gt_in);
        const double w0 = w0_in * (1.0 - gt_in * gt_in) / (1.0 - w0_in * gt_in * gt_in);
        const double gt = gt_in / (1.0 + gt_in);
        // quadrature coefficients (:43-44), lambda, Gamma (:50-51), exp (:56)
        const double gam1 = sqrt3 * (2.0 - w0 * (1 + gt)) / 2.0;
        const double gam2 = sqrt3 * w0 * (1.0 - gt) / 2.0;
        const double lam = sqrt(gam1 * gam1 - gam2 * gam2);
        ktau[k] = tau; kw0[k] = w0; kgt[k] = gt; klam[k] = lam;
        kG[k] = gam2 / (gam1 + lam);
        kx[k] = fast_exp(-lam * tau);
        s.E0[pr] = tau;
      }
    }
    __syncthreads();
    // tauc (:64-67): optical depth above each layer.  Segmented scan: 8-layer segments
    // summed in order, segment offsets summed in order.
    {
      const int nseg = (nz + 7) / 8;
      int sg, c;
      pair_split(tid, nc, sh, sg, c);
      const int i0 = sg * 8, i1 = min(nz, i0 + 8);
      if (sg < nseg) {
        double ssum = 0.0;
        for (int i = i0; i < i1; i++) ssum = ssum + s.E0[i * nc + c];
        s.Seg[sg * nc + c] = ssum;
      }
      __syncthreads();
      if (sg < nseg) {
        double run = 0.0;
        for (int s2 = 0; s2 < sg; s2++) run = run + s.Seg[s2 * nc + c];
        for (int i = i0; i < i1; i++) {
          const double t = s.E0[i * nc + c];
          s.E0[i * nc + c] = run;
          run = run + t;
        }
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
        const double tauc = s.E0[pr];
        const double gam1 = sqrt3 * (2.0 - kw0[k] * (1 + kgt[k])) / 2.0;
        const double gam2 = sqrt3 * kw0[k] * (1.0 - kgt[k]) / 2.0;
        const double lam2 = klam[k] * klam[k];
        double CP0 = 0.0, CPB = 0.0, CM0 = 0.0, CMB = 0.0, DIR = 0.0, DIRU = 0.0;
        for (int z = 0; z < p.nzen; z++) {
          const double u0 = p.zen_u[z], wz = p.zen_w[z], iu = p.zen_iu[z];
          const double gam3 = (1.0 - sqrt3 * kgt[k] * u0) / 2.0;
          const double gam4 = 1.0 - gam3;
          const double facp = kw0[k] * ((gam1 - iu) * gam3 + gam4 * gam2);
          const double facm = kw0[k] * ((gam1 + iu) * gam4 + gam2 * gam3);
          const double et0 = fast_exp(-tauc * iu);
          const double etb = et0 * fast_exp(-ktau[k] * iu);
          const double rden = wz * rcp_nr(lam2 - iu * iu);  // w_z / denom (:80)
          const double fp = facp * rden, fm = facm * rden;
          CP0 = __builtin_fma(et0, fp, CP0);
          CPB = __builtin_fma(etb, fp, CPB);
          CM0 = __builtin_fma(et0, fm, CM0);
          CMB = __builtin_fma(etb, fm, CMB);
          DIR = __builtin_fma(wz * u0, etb, DIR);   // direct(i+1) = u0*etb (:82)
          DIRU = __builtin_fma(wz, etb, DIRU);      // direct(i+1)/u0
        }
        kcpb[k] = CPB; kcmb[k] = CMB; kdir[k] = DIR; kdiru[k] = DIRU;
        s.G[pr] = kG[k];
        s.X[pr] = kx[k];
        s.E0[pr] = CP0;
        s.E1[pr] = CM0;
        if (pr < nc) cp0_first = CP0;
      }
    }
    if (tid < nc) {  // level-0 direct terms: direct(1)=u0 (:73), direct(1)/u0 = 1
      s.L0[1 * nc + tid] = dir0;
      s.L0[2 * nc + tid] = wsum;
    }
  } else {
    // =========================== IR (two_stream_ir, twostream.f90:156-295) ===========
    const double emis = p.emissivity[ll];
    Rsfc = p.has_hard_surface ? 1.0 - emis : 0.0;  // :186-190
    const double avg_freq = 0.5 * (p.freq[l] + p.freq[l + 1]);  // radiate.f90:64
    for (int n = tid; n < nz + 1; n += nt)                        // radiate.f90:65-69
      s.Bp[n] = p.bplanck ? p.bplanck[n] : planck_fcn(avg_freq, n == nz ? *p.T_surface : p.T[nz - 1 - n]);
    __syncthreads();
    const double norm = 2.0 * PI * 0.5;
#pragma unroll
    for (int k = 0; k < TS_MAXP; k++) {
      const int pr = tid + k * nt;
      if (pr < npairs) {
        int i, c;
        pair_split(pr, nc, sh, i, c);
        const double tau = tauL[c * nz + i], w0 = w0L[c * nz + i], gt = gL[i];
        const double gam1 = 2.0 - w0 * (1.0 + gt);  // :195-201
        const double gam2 = w0 * (1.0 - gt);
        const double lam = sqrt(gam1 * gam1 - gam2 * gam2);
        const double G = gam2 / (gam1 + lam);
        const double x = fast_exp(-lam * tau);
        double b0n, b1n;  // :216-227
        if (tau <= p.ir_tau_min) {
          b0n = 0.5 * (s.Bp[i] + s.Bp[i + 1]);
          b1n = 0.0;
        } else {
          b0n = s.Bp[i];
          b1n = (s.Bp[i + 1] - b0n) / tau;
        }
        const double r = 1.0 / (gam1 + gam2);
        const double cp0 = norm * (b0n + b1n * (r));  // :229-232
        const double cpb = norm * (b0n + b1n * (tau + r));
        const double cm0 = norm * (b0n + b1n * (-r));
        const double cmb = norm * (b0n + b1n * (tau - r));
        kG[k] = G; kx[k] = x; kcpb[k] = cpb; kcmb[k] = cmb; kdir[k] = 0.0; kdiru[k] = 0.0;
        s.G[pr] = G;
        s.X[pr] = x;
        s.E0[pr] = cp0;
        s.E1[pr] = cm0;
        if (pr < nc) cp0_first = cp0;
      }
    }
    if (tid < nc) {
      s.L0[1 * nc + tid] = 0.0;  // fdn(1) = 0 (:289)
      s.L0[2 * nc + tid] = 0.0;
    }
  }
  __syncthreads();

  // ---- right-hand sides from neighbouring layers: layer i's thread forms E(2i+1) (:111)
  //      and E(2i+2) (:102); layer 0's also E(0) (:96); layer nz-1's E(2nz-1) (:117)
  double R1[TS_MAXP], R2[TS_MAXP], R0 = 0.0;
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    R1[k] = 0.0; R2[k] = 0.0;
    if (pr < npairs) {
      int i, c;
      pair_split(pr, nc, sh, i, c);
      const E4 e = make_e(kG[k], kx[k]);
      const int j = s.chunk[i];
      if (i == 0) { R0 = 0.0 - s.E1[pr]; s.Ch[(0 * S + 0) * nc + c] = s.E0[pr]; }  // -cm0 of the top layer (:96)
      if (i < nz - 1) {
        const E4 f = make_e(s.G[pr + nc], s.X[pr + nc]);
        const double cp0n = s.E0[pr + nc], cm0n = s.E1[pr + nc];
        if (s.chunk[i + 1] != j) {
          // chunk boundary below layer i: the two interface rows become flux conditions
          R1[k] = 0.0 - kcpb[k];   // row 2i+1: fup entering layer i from below = Uin_j
          R2[k] = 0.0 - cm0n;      // row 2i+2: fdn entering layer i+1 from above = Din_{j+1}
          s.Ch[(2 * S + j) * nc + c] = kcpb[k];
          s.Ch[(3 * S + j) * nc + c] = kcmb[k];
          s.Ch[(0 * S + j + 1) * nc + c] = cp0n;
          s.Ch[(1 * S + j + 1) * nc + c] = cm0n;
        } else {
          R1[k] = f.e2 * (cp0n - kcpb[k]) - f.e4 * (cm0n - kcmb[k]);  // row 2i+1 (:111)
          R2[k] = e.e3 * (cp0n - kcpb[k]) + e.e1 * (kcmb[k] - cm0n);  // row 2i+2 (:102)
        }
      } else {
        double Ssfc;
        if (solar) {
          Ssfc = Rsfc * kdir[k];  // :89 (zenith-weighted direct beam at the ground)
        } else if (p.has_hard_surface) {
          Ssfc = p.emissivity[ll] * PI * s.Bp[nz];  // :237
        } else {  // :241-246
          const double tau = tauL[c * nz + i];
          const double b1_bot = (tau <= p.ir_tau_min) ? 0.0 : (s.Bp[nz] - s.Bp[nz - 1]) / tau;
          Ssfc = PI * (s.Bp[nz] + 0.5 * b1_bot);
        }
        R1[k] = Ssfc - kcpb[k] + Rsfc * kcmb[k];  // row 2nz-1 (:117)
        s.Ch[(2 * S + j) * nc + c] = kcpb[k];
        s.Ch[(3 * S + j) * nc + c] = kcmb[k];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    if (pr < npairs) {
      const int i = (sh >= 0) ? (pr >> sh) : pr / nc;
      s.E1[pr] = R1[k];
      if (i < nz - 1) s.E0[pr + nc] = R2[k];
      if (i == 0) s.E0[pr] = R0;
    }
  }
  __syncthreads();

  // ---- tridiagonal solve by one wave: S chunks x nc columns lanes
  if (tid < S * nc) dd_solve(s, nz, nc, S, tid, Rsfc);
  __syncthreads();

  // ---- level fluxes (:143-148, :288-293) and mean intensity (:135-140)
  const double inv_u1 = solar ? 1.7320508075688772 : 0.0;  // 1/u1, u1 = 1/sqrt(3)
  double ofu[TS_MAXP], ofd[TS_MAXP], oam[TS_MAXP];
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    ofu[k] = ofd[k] = oam[k] = 0.0;
    if (pr < npairs) {
      int i, c;
      pair_split(pr, nc, sh, i, c);
      const E4 e = make_e(kG[k], kx[k]);
      const int j = s.chunk[i];
      const double Uin = s.Bnd[(0 * S + j) * nc + c], Din = s.Bnd[(1 * S + j) * nc + c];
      const double y1 = s.E0[pr] + s.G[pr] * Uin + s.U[pr] * Din;  // Y(2i)
      const double y2 = s.E1[pr] + s.X[pr] * Uin + s.V[pr] * Din;  // Y(2i+1)
      ofu[k] = (y1 * e.e1 + y2 * e.e2 + kcpb[k]);
      ofd[k] = (y1 * e.e3 + y2 * e.e4 + kcmb[k]) + kdir[k];
      oam[k] = inv_u1 * (y1 * (e.e1 + e.e3) + y2 * (e.e2 + e.e4) + kcpb[k] + kcmb[k]) + kdiru[k];
      if (i == 0) {
        const double top = (y1 * e.e3 - y2 * e.e4) + cp0_first;
        s.L0[c] = top;                                         // fup(1)
        s.L0[2 * nc + c] = inv_u1 * top + s.L0[2 * nc + c];     // amean(1)
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < TS_MAXP; k++) {
    const int pr = tid + k * nt;
    if (pr < npairs) { s.G[pr] = ofu[k]; s.X[pr] = ofd[k]; s.E0[pr] = oam[k]; }
  }
  __syncthreads();

  // ---- g-point weights (radiate.f90:122-126), unit factors (:167-180), reversal to
  //      ground-first (:140-154).  With several g-point groups per bin the groups add
  //      their weighted partial sums into pre-zeroed outputs.
  const bool split = gridDim.y > 1;
  double scale = 1.0;
  if (solar) scale = p.photons_sol[ll] * p.photon_scale_factor;  // clima_radtran.f90:302
  for (int n = tid; n < nz + 1; n += nt) {
    double fu = 0.0, fd = 0.0, am = 0.0;
    for (int c = 0; c < nc; c++) {
      const double w = p.wbin[cg0 + c];
      if (n == 0) {
        fu = fu + s.L0[c] * w;
        fd = fd + s.L0[nc + c] * w;
        am = am + s.L0[2 * nc + c] * w;
      } else {
        fu = fu + s.G[(n - 1) * nc + c] * w;
        fd = fd + s.X[(n - 1) * nc + c] * w;
        am = am + s.E0[(n - 1) * nc + c] * w;
      }
    }
    const size_t o = (size_t)ll * (nz + 1) + (nz - n);
    if (solar) {
      fu = fu * scale * p.diurnal_fac;
      fd = fd * scale * p.diurnal_fac;
      am = am * scale * p.diurnal_fac;
      am = am * p.am_f1[ll];
      am = am * p.am_f2[ll] * p.am_dw[ll];
      if (split) { atomicAdd(&p.sol_fup_a[o], fu); atomicAdd(&p.sol_fdn_a[o], fd); atomicAdd(&p.sol_amean[o], am); }
      else { p.sol_fup_a[o] = fu; p.sol_fdn_a[o] = fd; p.sol_amean[o] = am; }
    } else {
      if (split) { atomicAdd(&p.ir_fup_a[o], fu); atomicAdd(&p.ir_fdn_a[o], fd); }
      else { p.ir_fup_a[o] = fu; p.ir_fdn_a[o] = fd; }
    }
  }
  if (blockIdx.y == 0) {
    double *tb = solar ? p.sol_tau_band : p.ir_tau_band;
    for (int i = tid; i < nz; i += nt) tb[(size_t)ll * nz + i] = p.tau_band[(size_t)l * nz + (nz - 1 - i)];
  }
}

struct ZeroParams {
  double *ptr[6];
  size_t count[6];
  int n;
};
__global__ __launch_bounds__(256) void k_zero(ZeroParams z) {
  double *p = z.ptr[blockIdx.y];
  const size_t n = z.count[blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

static void ts_zero_outputs(const TwoStreamParams &p, hipStream_t s);

// LDS bytes for nc columns per block
static size_t ts_lds_bytes(int nz, int nc, int S) {
  return sizeof(double) * ((size_t)6 * nz * nc + (nz + 1) + 3 * nc + (size_t)((nz + 7) / 8) * nc + (size_t)6 * S * nc + (size_t)(nz + 2) / 2 + 1);
}

bool launch_twostream(TwoStreamParams &p, hipStream_t s, size_t *lds_bytes) {
  // columns per block: all g-points when the LDS image fits, else split the g-points over
  // gridDim.y groups (outputs are then accumulated with atomics into zeroed arrays)
  int nc = p.ncols;
  if (nc <= 0) {
    // prefer an LDS image small enough for 3 workgroups per CU (latency hiding), but never
    // more than two g-point groups per bin: two partial sums added into a zeroed output
    // are order-independent, so results stay bitwise reproducible
    nc = p.ng;
    if (ts_lds_bytes(p.nz, nc, 64 / nc) > 53 * 1024 && p.ng % 2 == 0) nc = p.ng / 2;
  }
  while (nc > 1 && (ts_lds_bytes(p.nz, nc, 64 / nc > 16 ? 16 : 64 / nc) > 150 * 1024 || p.ng % nc != 0 || nc > 64)) nc--;
  if (p.ng % nc != 0) return false;
  int S = 64 / nc;
  if (S > 16) S = 16;
  if (S > p.nz) S = p.nz;
  if (S < 1) S = 1;
  p.ncols = nc;
  p.nchunks = S;
  p.nc_shift = -1;
  for (int b = 0; b < 7; b++) if ((1 << b) == nc) p.nc_shift = b;
  const size_t lds = ts_lds_bytes(p.nz, nc, S);
  if (lds_bytes) *lds_bytes = lds;
  if (lds > 160 * 1024) return false;
  const int npairs = p.nz * nc;
  const int nseg = (p.nz + 7) / 8;
  int threads = (npairs + TS_MAXP - 1) / TS_MAXP;
  threads = ((threads + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  if (threads < nseg * nc) threads = ((nseg * nc + 63) / 64) * 64;
  if (threads > 1024) return false;
  const int grid = p.n_sol + p.n_ir;
  if (grid <= 0) return true;
  const dim3 g(grid, p.ng / nc);
  if (g.y > 1) ts_zero_outputs(p, s);
  if (threads <= 512) {
    if (!ensure_max_lds((const void *)k_twostream<512, 4>)) return false;
    hipLaunchKernelGGL((k_twostream<512, 4>), g, dim3(threads), lds, s, p);
  } else {
    if (!ensure_max_lds((const void *)k_twostream<1024, 4>)) return false;
    hipLaunchKernelGGL((k_twostream<1024, 4>), g, dim3(threads), lds, s, p);
  }
  return true;
}

// ------------------------------------------------------------------------------------
// k_twostream_w: wave-per-column form of the two-stream solve (no block barriers, no
// serial phase).  One wave = one (channel, bin, g-point) column; lane q owns the chunk of
// consecutive layers [q*nz/64, (q+1)*nz/64) (<= LMAX = ceil(nz/64) layers) entirely in registers.  The
// chunk is solved as a two-stream problem of its own with flux boundary conditions (same
// construction as dd_solve above), which turns it into an affine map
//     (Din, Uin) -> (Dout, Uout) = (dS + dD*Din + dU*Uin,  uS + uD*Din + uU*Uin).
// The 64 chunks of the column are joined by two wave-level scans (over DPP, below):
//   * bottom-up: the reflectance/source (rho, sigma) seen from above each interface,
//       Uin_{q-1} = rho_{q-1}*Din_q + sigma_{q-1},
//     is a Moebius recursion in rho; written projectively, (n_rho, n_sigma, den) <- M_q * (...)
//     with M_q = [[uU*dD-uD*dU, 0, uD], [uU*dS-uS*dU, uU, uS], [-dU, 0, 1]], it becomes a
//     suffix product of 3x3 matrices (7 structural non-zeros), i.e. a Kogge-Stone scan;
//   * top-down: Din_{q+1} = alpha_q + beta_q*Din_q, an affine prefix scan.
// A block is 4 waves = 4 g-point columns of one bin; their weighted level fluxes meet in LDS
// once at the end.
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// Wave-wide affine scans over DPP (data-parallel primitives) instead of ds_bpermute shuffles.
// x_i = a_i + b_i * x_{i-1} over the 64 lanes, as composition of the maps (a_i, b_i); the
// schedule is the row_shr 1,2,3 / 4 / 8 / row_bcast 15 / row_bcast 31 sequence (7 steps, each a
// DPP move of the two halves of a double: a few cycles, where a ds_bpermute round trip is ~100).
// A lane with no source in a step keeps `old`: 0 for an a-part, 1 for a b-part, which makes the
// step the identity for it -- no participation masks needed.
// ------------------------------------------------------------------------------------
constexpr int WSCAN_STEPS = 7;

// value a step brings in from the lower lane(s); `orig` is the value before the scan (steps 0-2
// read the original neighbours), `cur` the running one
template <int STEP>
__device__ __forceinline__ double wscan_fetch(double old, double orig, double cur) {
  if constexpr (STEP == 0) return dpp_mov<DPP_ROW_SHR + 1, 0xf, 0xf>(old, orig);
  else if constexpr (STEP == 1) return dpp_mov<DPP_ROW_SHR + 2, 0xf, 0xf>(old, orig);
  else if constexpr (STEP == 2) return dpp_mov<DPP_ROW_SHR + 3, 0xf, 0xf>(old, orig);
  else if constexpr (STEP == 3) return dpp_mov<DPP_ROW_SHR + 4, 0xf, 0xe>(old, cur);
  else if constexpr (STEP == 4) return dpp_mov<DPP_ROW_SHR + 8, 0xf, 0xc>(old, cur);
  else if constexpr (STEP == 5) return dpp_mov<DPP_ROW_BCAST15, 0xa, 0xf>(old, cur);
  else return dpp_mov<DPP_ROW_BCAST31, 0xc, 0xf>(old, cur);
}

// Full scan of (a, b); bstep[k] receives the multiplier lane i applies in step k, so that later
// scans with the same b's but other a's only need wscan_apply.
template <int STEP>
__device__ __forceinline__ void wscan_build_step(double &a, double &b, const double a0, const double b0, double *bstep) {
  const double ta = wscan_fetch<STEP>(0.0, a0, a);
  const double tb = wscan_fetch<STEP>(1.0, b0, b);
  bstep[STEP] = b;
  a = a + b * ta;
  b = b * tb;
}
// HALF: two independent scans, lanes 0-31 and 32-63 (the last step, which carries lane 31 into the
// upper half, is left out)
template <bool HALF = false>
__device__ __forceinline__ void wscan_build(double &a, double &b, double *bstep) {
  const double a0 = a, b0 = b;
  wscan_build_step<0>(a, b, a0, b0, bstep); wscan_build_step<1>(a, b, a0, b0, bstep);
  wscan_build_step<2>(a, b, a0, b0, bstep); wscan_build_step<3>(a, b, a0, b0, bstep);
  wscan_build_step<4>(a, b, a0, b0, bstep); wscan_build_step<5>(a, b, a0, b0, bstep);
  if constexpr (!HALF) wscan_build_step<6>(a, b, a0, b0, bstep);
}
// a-part only, with the multipliers of a previous wscan_build (read through `bs(k)`)
template <class BS>
__device__ __forceinline__ double wscan_apply(double a, BS bs) {
  const double a0 = a;
  a = a + bs(0) * wscan_fetch<0>(0.0, a0, a);
  a = a + bs(1) * wscan_fetch<1>(0.0, a0, a);
  a = a + bs(2) * wscan_fetch<2>(0.0, a0, a);
  a = a + bs(3) * wscan_fetch<3>(0.0, a0, a);
  a = a + bs(4) * wscan_fetch<4>(0.0, a0, a);
  a = a + bs(5) * wscan_fetch<5>(0.0, a0, a);
  a = a + bs(6) * wscan_fetch<6>(0.0, a0, a);
  return a;
}
// inclusive prefix sum over the 64 lanes (same schedule, a-part only with unit multipliers)
template <bool HALF = false>
__device__ __forceinline__ double wscan_sum(double a) {
  const double a0 = a;
  a = a + wscan_fetch<0>(0.0, a0, a);
  a = a + wscan_fetch<1>(0.0, a0, a);
  a = a + wscan_fetch<2>(0.0, a0, a);
  a = a + wscan_fetch<3>(0.0, a0, a);
  a = a + wscan_fetch<4>(0.0, a0, a);
  a = a + wscan_fetch<5>(0.0, a0, a);
  if constexpr (!HALF) a = a + wscan_fetch<6>(0.0, a0, a);
  return a;
}
// x[lane-1], 0 in lane 0 (HALF: and in lane 32)
template <bool HALF = false>
__device__ __forceinline__ double wave_shr1(double x) {
  const double v = dpp_mov<DPP_WAVE_SHR1, 0xf, 0xf>(0.0, x);
  if constexpr (HALF) return (threadIdx.x & 31) == 0 ? 0.0 : v;
  else return v;
}
// x[63-lane] (HALF: mirrored within each half)
template <bool HALF = false>
__device__ __forceinline__ double wave_reverse(double x) { return __shfl(x, (int)(threadIdx.x & 63) ^ (HALF ? 31 : 63)); }

struct M7 {
  double m00, m02, m10, m11, m12, m20, m22;
};
__device__ __forceinline__ M7 m7_mul(const M7 &l, const M7 &r) {
  M7 p;
  p.m00 = l.m00 * r.m00 + l.m02 * r.m20;
  p.m02 = l.m00 * r.m02 + l.m02 * r.m22;
  p.m10 = l.m10 * r.m00 + l.m11 * r.m10 + l.m12 * r.m20;
  p.m11 = l.m11 * r.m11;
  p.m12 = l.m10 * r.m02 + l.m11 * r.m12 + l.m12 * r.m22;
  p.m20 = l.m20 * r.m00 + l.m22 * r.m20;
  p.m22 = l.m20 * r.m02 + l.m22 * r.m22;
  return p;
}

// One step of the DPP prefix scan (wscan_fetch schedule) of 3x3 products P_i <- P_i * P_{i-1} * ...:
// a lane without a source in the step multiplies by the identity
template <int STEP>
__device__ __forceinline__ void m7_scan_step(M7 &P, const M7 &P0) {
  M7 R;
  R.m00 = wscan_fetch<STEP>(1.0, P0.m00, P.m00); R.m02 = wscan_fetch<STEP>(0.0, P0.m02, P.m02);
  R.m10 = wscan_fetch<STEP>(0.0, P0.m10, P.m10); R.m11 = wscan_fetch<STEP>(1.0, P0.m11, P.m11);
  R.m12 = wscan_fetch<STEP>(0.0, P0.m12, P.m12); R.m20 = wscan_fetch<STEP>(0.0, P0.m20, P.m20);
  R.m22 = wscan_fetch<STEP>(1.0, P0.m22, P.m22);
  P = m7_mul(P, R);
}
template <bool HALF = false>
__device__ __forceinline__ void m7_prefix_scan(M7 &P) {
  const M7 P0 = P;
  m7_scan_step<0>(P, P0); m7_scan_step<1>(P, P0); m7_scan_step<2>(P, P0); m7_scan_step<3>(P, P0);
  m7_scan_step<4>(P, P0); m7_scan_step<5>(P, P0);
  if constexpr (!HALF) m7_scan_step<6>(P, P0);
}

constexpr int TSW_COLS = 4;  // waves (g-point columns) per block

// ------------------------------------------------------------------------------------
// twostream_p_body: the solve of one column by one wave (see above), as straight-line code.
// A lane whose chunk holds fewer than L layers fills its L slots from the top with zero-thickness
// layers (tau = 0, w0 = 0: transparent, and with the same Planck value on both faces they emit
// nothing); its real layers sit below them.  A zero-thickness layer hands both fluxes through
// unchanged, so the column's solution is the same (to rounding: its rows have pivots of 2), and
// every lane runs the same L layers.  With no per-layer branch there are no exec-mask regions, no
// merge copies and no zero initialisation of skipped layers, the zenith angles become the outer
// loop (their constants are fetched once, the direct-beam transmission is a running product down
// the chunk) and the whole coefficient phase is one block for the scheduler.  In this code a
// wave's run time follows its instruction count (see fast_exp); an earlier form that branched on
// every layer's presence spent a third of its instructions on that bookkeeping.
// ------------------------------------------------------------------------------------
// element offsets of a batch column (all 0 for a single call)
struct TsOfs {
  size_t opr, col, res;
};

// PAIRED: every layer 2m+1 carries the optical properties of layer 2m bit for bit (AdiabatClimate's
// doubled radiative grid, src/adiabat/clima_adiabat.f90:729-773: the host sets TwoStreamParams::paired
// when pair_reuse marked every pair as exact).  The lanes' chunks are then cut at pair boundaries, a
// pair's tau / w0 / g are loaded once, and everything that depends on them alone -- the gammas, lambda,
// Gamma, exp(-lambda tau), and per zenith angle the attenuation factor and the source factors -- is
// computed once per pair; the Planck source, the running direct beam and the elimination are per layer
// as always (same operations on the same values: results are bitwise those of the unpaired form).
//
// HALF: the wave solves TWO g-point columns, one per half of 32 lanes, with L = ceil(nz/32) slots per
// lane (a block of 4 waves then covers all 8 g-points of a bin).  A column of 200 layers fills 200 of
// 4 x 64 = 256 slots in the whole-wave form and 400 of 7 x 64 = 448 here, the chunk scans have one
// step fewer and there are half as many of them -- about a fifth fewer instructions per column.
template <int L, bool SOLAR, int NZMAX, bool COHERENT, bool RESK, bool PAIRED = false, bool HALF = false>
__device__ __forceinline__ void twostream_p_body(const TwoStreamParams &p, const int bin_local, double *lds,
                                                 const int gy, const int bz, const int tslot = -1,
                                                 const TsOfs co = TsOfs{0, 0, 0}) {
  static_assert(!PAIRED || (L % 2) == 0, "paired slots come in twos");
  static_assert(!(PAIRED && HALF), "no paired half-wave form");
  constexpr int W = HALF ? 32 : 64, WSH = HALF ? 5 : 6;   // lanes per column
  constexpr int NCOL = TSW_COLS * (HALF ? 2 : 1);          // g-point columns per block
#ifdef CLIMA_STAMPS
#define TSTAMP(k)                                                                                  \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (tslot >= 0 && g_stamp_buf && threadIdx.x == 0) g_stamp_buf[tslot + (k)] = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define TSTAMP(k) do { } while (0)
  (void)tslot;
#endif
  TSTAMP(0);
  constexpr bool solar = SOLAR;
  // exp with its constants resident in VGPRs where the register budget is there anyway (RESK: the
  // fused grid, 256 per wave); the stand-alone kernels keep their higher occupancy instead
  ExpK K;
  if constexpr (RESK) K.load();
  auto fexp = [&](double x) {
    if constexpr (RESK) return fast_exp(x, K);
    else return fast_exp(x);
  };
  (void)fexp;   // (only the CLIMA_ZEN_EXP_POLY build still uses the polynomial exp here)
  auto planck = [&](double nu, double T) {
    if constexpr (RESK) return planck_fcn(nu, T, K);
    else return planck_fcn(nu, T);
  };
  const int nz = p.nz, ng = p.ng, nl = nz + 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (W - 1);   // lane within the column's group
  const int wc = HALF ? (int)(threadIdx.x >> 5) : wave;               // column within the block
  const int ll = (solar ? p.sol_lo : p.ir_lo) + bin_local;
  const int l = (solar ? p.sol_start : p.ir_start) + ll;  // opacity bin (radiate.f90:57)
  const int c_raw = p.col_base + gy * NCOL + wc;
  const bool col_on = c_raw < ng;
  const int c = col_on ? c_raw : ng - 1;
  const double wcol = col_on ? p.wbin[c] : 0.0;  // g-point weight (radiate.f90:122-126)
  const double *tauL = p.tau + co.opr + ((size_t)l * ng + c) * nz;
  // (w0_from_scat: the layer's scattering optical depth in place of the g-point's w0, see below)
  const double *w0L = p.w0_from_scat ? p.scat + co.opr + (size_t)l * nz : p.w0 + co.opr + ((size_t)l * ng + c) * nz;
  const double *gL = p.g + co.opr + (size_t)l * nz;
  // layers [a,b) TOA-first; slot t holds layer a + t - pad when t >= pad, a zero-thickness layer otherwise
  const int a = PAIRED ? 2 * ((lane * (nz >> 1)) >> 6) : (lane * nz) >> WSH;
  const int b = PAIRED ? 2 * (((lane + 1) * (nz >> 1)) >> 6) : ((lane + 1) * nz) >> WSH;
  const int pad = L - (b - a);
  const bool is_toa = lane == 0, is_sfc = lane == W - 1;  // b == nz holds for the last lane only, and its chunk is never empty
  const double sqrt3 = 1.7320508075688772;
  const double inv_u1 = solar ? sqrt3 : 0.0;  // 1/u1, u1 = 1/sqrt(3) (solar only)

  // ---- the slots' inputs: one batch of loads (a valid address also for the zero-thickness slots)
  double tau_s[L], w0_s[L], gt_s[L];
#pragma unroll
  for (int t = 0; t < L; t++) {
    if (PAIRED && (t & 1)) {  // the pair's second layer: the same values (pad is even: t-1 is its partner)
      tau_s[t] = tau_s[t - 1]; w0_s[t] = w0_s[t - 1]; gt_s[t] = gt_s[t - 1];
    } else {
      const int i = min(max(a + t - pad, 0), nz - 1);
      tau_s[t] = ld_opr<COHERENT>(&tauL[i]);
      w0_s[t] = ld_opr<COHERENT>(&w0L[i]);
      gt_s[t] = ld_opr<COHERENT>(&gL[i]);
    }
  }
  const bool copies_band = gy == 0 && p.col_base == 0 && p.b_out == 0;  // this block copies the band optical depth out
  // What the block's LDS table will hold -- the thread's entry of the exp table (solar) or the temperature /
  // Planck value of the thread's level (IR) -- is requested with the loads above, not after them: behind
  // their first use it was one more memory round trip before the block's first arithmetic.
  const double *Tcol = p.T + co.col + (size_t)bz * p.b_T;   // (batched shared-opacity IR launches: blockIdx.z selects
  const double *Tsfc = p.T_surface + co.col + (size_t)bz * p.b_Ts;   //  the temperature column; strides 0 otherwise)
  static_assert(EXP2_N == 64 * TSW_COLS, "one table entry per thread");
  const double e2_in = EXP2_TAB[threadIdx.x];
  double table_in = 0.0;
  if constexpr (!solar) {
    const int n = min((int)threadIdx.x, nz);
    table_in = p.bplanck ? p.bplanck[n] : (n == nz ? *Tsfc : Tcol[nz - 1 - min(n, nz - 1)]);
  }
  // exp table (exp_tab): the solar zenith-angle loop's attenuations and both channels' exp(-lambda tau)
  __shared__ double s_e2[EXP2_N];
  s_e2[threadIdx.x] = e2_in;   // (every path has a barrier before its first exp_tab)
  if (p.w0_from_scat) {
    // the opacity tiles of this grid left w0 unwritten: w0 = min(MAX_W0, scat / tau) (types.f90:869-875;
    // by the correctly rounded reciprocal, <= 1 ulp from the quotient the stored array would hold)
#pragma unroll
    for (int t = 0; t < L; t++)
      w0_s[t] = tau_s[t] <= TAU_MIN ? 0.0 : dmin(MAX_W0, w0_s[t] * rcp_nr(tau_s[t]));
  }
#pragma unroll
  for (int t = 0; t < L; t++) {
    const bool real = t >= pad;
    tau_s[t] = real ? tau_s[t] : 0.0;
    w0_s[t] = real ? w0_s[t] : 0.0;
    gt_s[t] = real ? gt_s[t] : 0.0;
  }

  double G[L], X[L], cp0[L], cm0[L], cpb[L], cmb[L], dir[L], diru[L];
  double Rsfc, Ssfc = 0.0, lvl0_dn = 0.0, lvl0_am = 0.0;
  // this column's rows of the block's level values (each lane writes the levels under its own layers)
  // (level n of column w: lds[(w nl + n) 3 + k], k = 0 up-flux, 1 down-flux, 2 mean intensity.  `mine`
  // is the entry of the level under the lane's slot 0 -- slot t's is mine[3 t + k], at constant offsets
  // from ONE per-lane address; it is only dereferenced for t >= pad, where it lies in the lane's range)
  double *const mine = lds + ((size_t)wc * nl + (a - pad + 1)) * 3;
  // PARK (5 and more slots per lane): what the zenith-angle loop does not touch waits in LDS instead of
  // in registers -- Gamma, A and w0/2 of a layer in the lane's own three entries of its level (a
  // zero-thickness slot has no entry and needs none: its Gamma and w0/2 are 0) -- and the direct beam
  // goes to the down-flux / mean-intensity entries as soon as it is summed.  With everything in registers the 7-slot
  // form spilled 25-120 of them to scratch, and every reload stalls the wave for an L2 round trip:
  // variants with FEWER instructions but more spills ran slower.
  // Both channels: the source terms at the layers' lower faces (C+(tau), and in the IR C-(tau): the
  // solar down-flux and mean-intensity entries hold the direct beam), which the elimination reads for the
  // last time long before the level fluxes need them again, wait there through the scans.
  constexpr bool PARK = HALF || L >= 5;

  if constexpr (solar) {
    // ---- delta-Eddington (:38-40), quadrature coefficients (:43-44), lambda, Gamma (:50-51)
    double taup[L], lam[L], zA[L], zB[L], zH[L];
    double tot = 0.0;
#pragma unroll
    for (int t = 0; t < L; t++) {
      taup[t] = tau_s[t] * (1.0 - w0_s[t] * gt_s[t] * gt_s[t]);
      tot = tot + taup[t];
    }
    // optical depth above the chunk (tauc, :64-67): exclusive wave scan of the chunk totals
    double tcum = wave_shr1<HALF>(wscan_sum<HALF>(tot));  // DPP scans: no LDS round trips (0 enters lane 0)
    double tauc0 = tcum;
#pragma unroll
    for (int t = 0; t < L; t++) {
      if (PAIRED && (t & 1)) {  // the pair's second layer: same tau', w0', g' -> same coefficients
        G[t] = G[t - 1]; lam[t] = lam[t - 1]; zA[t] = zA[t - 1]; zB[t] = zB[t - 1]; zH[t] = zH[t - 1];
        tcum = tcum + taup[t];
        cp0[t] = cm0[t] = cpb[t] = cmb[t] = dir[t] = diru[t] = 0.0;
        continue;
      }
      // quotients with denominators of ordinary size: numerator times the correctly rounded
      // reciprocal (6 instructions instead of the 11 of a full division, <= 1 ulp)
      const double w0p = w0_s[t] * (1.0 - gt_s[t] * gt_s[t]) * rcp_nr(1.0 - w0_s[t] * gt_s[t] * gt_s[t]);
      const double gtp = gt_s[t] * rcp_nr(1.0 + gt_s[t]);
      const double gam1 = sqrt3 * (2.0 - w0p * (1 + gtp)) / 2.0;
      const double gam2 = sqrt3 * w0p * (1.0 - gtp) / 2.0;
      const double lm = sqrt_nr(gam1 * gam1 - gam2 * gam2);
      G[t] = gam2 * rcp_nr(gam1 + lm);
      // a zero-thickness slot has w0p = 0, so its C+/C- vanish whatever the denominator
      // lam^2 - 1/u0^2 is -- as long as that is not 0 (its lambda is sqrt(3), u0 may be 1/sqrt(3)):
      // lambda = 0 keeps it at -1/u0^2 (and exp(-lambda tau') = 1 as before: tau' = 0)
      lam[t] = (t >= pad) ? lm : 0.0;
      // The source factors of :45-46 and :75-77 are affine in u0 and 1/u0:
      //   w0 ((gam1 - 1/u0) gam3 + gam4 gam2) = (w0/2) (A - s),  w0 ((gam1 + 1/u0) gam4 + gam2 gam3) = (w0/2) (A + s),
      //   gam3 = (1 - sqrt3 g u0)/2, gam4 = 1 - gam3,  A = gam1 + gam2 + sqrt3 g,  s = 1/u0 - sqrt3 g (gam2 - gam1) u0:
      // 6 operations per (layer, zenith angle) instead of 14
      const double a = sqrt3 * gtp;
      zA[t] = gam1 + gam2 + a;
      zB[t] = a * (gam2 - gam1);
      zH[t] = 0.5 * w0p;
      tcum = tcum + taup[t];
      cp0[t] = cm0[t] = cpb[t] = cmb[t] = dir[t] = diru[t] = 0.0;
    }
    if constexpr (PARK) {
#pragma unroll
      for (int t = 0; t < L; t++)
        if (t >= pad) { mine[3 * t] = G[t]; mine[3 * t + 1] = zA[t]; mine[3 * t + 2] = zH[t]; }
    }
    __syncthreads();  // s_e2
    TSTAMP(1);
    // ---- C+/C- and direct beam (:73-87) summed over the zenith angles with their weights (the
    //      matrix does not depend on u0: sum_z w_z*solve(E_z) == solve(sum_z w_z*E_z)).  exp(-tauc/u0)
    //      is evaluated at the chunk's top and carried down as a running product (:78-79).
    double wsum = 0.0, dir0 = 0.0;
    auto zen = [&](const int z) {
      const double u0 = p.zen_u_v[z], wz = p.zen_w_v[z], iu = p.zen_iu_v[z];
      wsum = wsum + wz;
      dir0 = dir0 + wz * u0;
      const double iu2 = iu * iu, wzu = wz * u0;
      const double kz = -iu * EXP2_PER_E;   // exp(-tau'/u0) = 2^(tau' kz / 256)
#ifdef CLIMA_ZEN_EXP_POLY
      double et = fexp(-tauc0 * iu);
#else
      double et = exp_tab_any(tauc0, kz, s_e2);
#endif
      double ex = 0.0, R = 0.0, sR = 0.0;  // attenuation and source factors of the current layer (pair)
#pragma unroll
      for (int t = 0; t < L; t++) {
        if (!(PAIRED && (t & 1))) {
#ifdef CLIMA_ZEN_EXP_POLY
          ex = fexp(-taup[t] * iu);  // :79
#else
          ex = exp_tab_any(taup[t], kz, s_e2);  // :79
#endif
          R = wz * rcp_n1(__builtin_fma(lam[t], lam[t], -iu2));   // w_z / denom (:80)
          sR = __builtin_fma(-zB[t], u0, iu) * R;                   // s w_z / denom
        }
        // C+ = (w0/2) (A - s) e / denom, C- = (w0/2) (A + s) e / denom at the layer's top (e = et) and
        // bottom (etb): the sums over the angles of e/denom and s e/denom are carried (in cp0 / cm0 and
        // cpb / cmb), A and w0/2 go on afterwards
        const double etb = et * ex;
        cp0[t] = __builtin_fma(et, R, cp0[t]);
        cpb[t] = __builtin_fma(etb, R, cpb[t]);
        cm0[t] = __builtin_fma(et, sR, cm0[t]);
        cmb[t] = __builtin_fma(etb, sR, cmb[t]);
        dir[t] = __builtin_fma(wzu, etb, dir[t]);   // direct(i+1) = u0*etb (:82)
        diru[t] = __builtin_fma(wz, etb, diru[t]);  // direct(i+1)/u0
        et = etb;
      }
    };
    if (NZMAX > 0 && p.nzen >= NZMAX) {  // the usual case: no per-angle test
#pragma unroll
      for (int z = 0; z < NZMAX; z++) zen(z);
    } else {
#pragma unroll
      for (int z = 0; z < NZMAX; z++)
        if (z < p.nzen) zen(z);
    }
    for (int z = NZMAX; z < p.nzen; z++) zen(z);
    Rsfc = p.albedo[ll];
    Ssfc = Rsfc * dir[L - 1];  // :89 (used by the surface row only)
#pragma unroll
    for (int t = 0; t < L; t++) {
      double zAt = zA[t], zHt = zH[t];
      if constexpr (PARK) {
        G[t] = 0.0; zAt = 0.0; zHt = 0.0;   // a zero-thickness slot: w0' = 0 -> gamma2 = 0, Gamma = 0 (A: any finite value)
        if (t >= pad) {
          G[t] = mine[3 * t]; zAt = mine[3 * t + 1]; zHt = mine[3 * t + 2];
          mine[3 * t + 1] = dir[t]; mine[3 * t + 2] = diru[t];
        }
      }
      const double a0 = zAt * cp0[t], s0 = cm0[t], ab = zAt * cpb[t], sb = cmb[t];
      cp0[t] = zHt * (a0 - s0); cm0[t] = zHt * (a0 + s0);
      cpb[t] = zHt * (ab - sb); cmb[t] = zHt * (ab + sb);
      if constexpr (PARK) {
        if (t >= pad) mine[3 * t] = cpb[t];
      }
      if (PAIRED && (t & 1)) X[t] = X[t - 1];
      else X[t] = exp_tab(-lam[t] * taup[t], s_e2);  // :56
    }
    lvl0_dn = dir0;   // direct(1) = u0 (:73)
    lvl0_am = wsum;   // direct(1)/u0 = 1
  } else {
    Rsfc = p.has_hard_surface ? 1.0 - p.emissivity[ll] : 0.0;  // :186-190
    const double avg_freq = 0.5 * (p.freq[l] + p.freq[l + 1]);  // radiate.f90:64
    // Planck source at the levels (radiate.f90:65-69): the same for the block's g-point columns, so
    // each of the nz+1 values is computed once per block instead of L+1 times per lane of every wave
    double *sB = lds + (size_t)3 * NCOL * nl;
    if ((int)threadIdx.x < nl) sB[threadIdx.x] = p.bplanck ? table_in : planck(avg_freq, table_in);  // TOA-first level
    for (int n = threadIdx.x + blockDim.x; n < nl; n += blockDim.x)   // (more levels than threads)
      sB[n] = p.bplanck ? p.bplanck[n] : planck(avg_freq, n == nz ? *Tsfc : Tcol[nz - 1 - min(n, nz - 1)]);
    __syncthreads();
    // the L+1 faces of the slots (level a for every face of a zero-thickness slot)
    double bpl[L + 1];
#pragma unroll
    for (int s = 0; s <= L; s++) bpl[s] = sB[a + max(s - pad, 0)];
    TSTAMP(1);
    double r_pair = 0.0;
#pragma unroll
    for (int t = 0; t < L; t++) {
      const double tau_in = tau_s[t], w0_in = w0_s[t], gt_in = gt_s[t];
      if (PAIRED && (t & 1)) {  // the pair's second layer: same tau, w0, g -> same coefficients
        G[t] = G[t - 1]; X[t] = X[t - 1];
      } else {
        const double gam1 = 2.0 - w0_in * (1.0 + gt_in);  // :195-201
        const double gam2 = w0_in * (1.0 - gt_in);
        const double lam = sqrt_nr(gam1 * gam1 - gam2 * gam2);
        G[t] = gam2 * rcp_nr(gam1 + lam);
        X[t] = exp_tab(-lam * tau_in, s_e2);
        r_pair = rcp_nr(gam1 + gam2);
      }
      const double bpl_top = bpl[t], bpl_bot = bpl[t + 1];
      double b0n, b1n;  // :216-227 (a zero-thickness slot is thin whatever ir_tau_min is set to)
      if (tau_in <= p.ir_tau_min || t < pad) {
        b0n = 0.5 * (bpl_top + bpl_bot);
        b1n = 0.0;
      } else {
        b0n = bpl_top;
        b1n = (bpl_bot - b0n) * rcp_nr(tau_in);
      }
      const double norm = 2.0 * PI * 0.5;
      const double r = r_pair;
      cp0[t] = norm * (b0n + b1n * (r));  // :229-232
      cpb[t] = norm * (b0n + b1n * (tau_in + r));
      cm0[t] = norm * (b0n + b1n * (-r));
      cmb[t] = norm * (b0n + b1n * (tau_in - r));
      dir[t] = diru[t] = 0.0;
      if constexpr (PARK) {
        if (t >= pad) { mine[3 * t] = cpb[t]; mine[3 * t + 1] = cmb[t]; }
      }
    }
    {  // surface source (:236-247), used by the surface row only
      const double tau_in = tau_s[L - 1], bpl_top = bpl[L - 1], bpl_bot = bpl[L];
      if (p.has_hard_surface) {
        Ssfc = p.emissivity[ll] * PI * bpl_bot;
      } else {
        const double b1_bot = (tau_in <= p.ir_tau_min) ? 0.0 : (bpl_bot - bpl_top) / tau_in;
        Ssfc = PI * (bpl_bot + 0.5 * b1_bot);
      }
    }
  }
  TSTAMP(2);

  // ---- the chunk's tridiagonal system, eliminated downward (y_r + c' y_{r+1} + l*Din = d').
  //      Row 2t-1 couples slots t-1,t (Fortran even rows, :106-112), row 2t likewise (odd rows,
  //      :97-103); row 0 and row 2L-1 are the flux boundary rows (TOA :93-96 / surface :113-117 at
  //      the column ends).
  // C+(0) of the lane's first real layer (the column top's source in lane 0: the last lines use it)
  double cp0_top = cp0[0];
#pragma unroll
  for (int t = 1; t < L; t++) cp0_top = (t == pad) ? cp0[t] : cp0_top;
  double rc[2 * L], rd[2 * L], rl[2 * L];
  {
    double cp, dp, lp;
    E4 u = make_e(G[0], X[0]);
    {
      // row 0: TOA row (:93-96) or the flux condition "-Din + e1 y1 - e2 y2 = -cm0"
      const double A = is_toa ? 0.0 : -1.0;
      const double r = rcp_nr(u.e1);
      cp = (-u.e2) * r; dp = (0.0 - cm0[0]) * r; lp = A * r;
      rc[0] = cp; rd[0] = dp; rl[0] = lp;
    }
#pragma unroll
    for (int t = 1; t < L; t++) {
      const E4 v = make_e(G[t], X[t]);
      // row 2t-1 (slots t-1, t)
      double A = v.e2 * u.e1 - u.e3 * v.e4, B = u.e2 * v.e2 - u.e4 * v.e4, D = v.e1 * v.e4 - v.e2 * v.e3;
      double E = v.e2 * (cp0[t] - cpb[t - 1]) - v.e4 * (cm0[t] - cmb[t - 1]);
      double r = rcp_nr(B - A * cp);
      const double cn = D * r, dn = (E - A * dp) * r, ln = (-A * lp) * r;
      rc[2 * t - 1] = cn; rd[2 * t - 1] = dn; rl[2 * t - 1] = ln;
      // row 2t
      A = u.e2 * u.e3 - u.e4 * u.e1; B = u.e1 * v.e1 - u.e3 * v.e3; D = u.e3 * v.e4 - u.e1 * v.e2;
      E = u.e3 * (cp0[t] - cpb[t - 1]) + u.e1 * (cmb[t - 1] - cm0[t]);
      r = rcp_nr(B - A * cn);
      cp = D * r; dp = (E - A * dn) * r; lp = (-A * ln) * r;
      rc[2 * t] = cp; rd[2 * t] = dp; rl[2 * t] = lp;
      u = v;
    }
    {
      // last row: surface row (:113-117) or "e1 y1 + e2 y2 - Uin = -cpb"
      const double A = is_sfc ? u.e1 - Rsfc * u.e3 : u.e1;
      const double B = is_sfc ? u.e2 - Rsfc * u.e4 : u.e2;
      const double D = is_sfc ? 0.0 : -1.0;
      const double E = is_sfc ? Ssfc - cpb[L - 1] + Rsfc * cmb[L - 1] : 0.0 - cpb[L - 1];
      const double r = rcp_nr(B - A * cp);
      rc[2 * L - 1] = D * r; rd[2 * L - 1] = (E - A * dp) * r; rl[2 * L - 1] = (-A * lp) * r;
    }
  }
  // ---- upward: y_r = alpha_r + beta_r*Uin + gamma_r*Din   (alpha -> rd, beta -> rc, gamma -> rl)
  {
    double al = 0.0, be = 1.0, ga = 0.0;
#pragma unroll
    for (int r = 2 * L - 1; r >= 0; r--) {
      const double cc = rc[r], dd = rd[r], lc = rl[r];
      al = dd - cc * al; be = -cc * be; ga = -lc - cc * ga;
      rd[r] = al; rc[r] = be; rl[r] = ga;
    }
  }
  // ---- the chunk as an affine map of (Din, Uin): up-flux leaving through its top (fup(1) form,
  //      :143), down-flux through its bottom (:147)
  const E4 ea = make_e(G[0], X[0]), eb = make_e(G[L - 1], X[L - 1]);
  double uS = rd[0] * ea.e3 - rd[1] * ea.e4 + cp0[0], uD = rl[0] * ea.e3 - rl[1] * ea.e4, uU = rc[0] * ea.e3 - rc[1] * ea.e4;
  double dS = rd[2 * L - 2] * eb.e3 + rd[2 * L - 1] * eb.e4 + cmb[L - 1], dD = rl[2 * L - 2] * eb.e3 + rl[2 * L - 1] * eb.e4,
         dU = rc[2 * L - 2] * eb.e3 + rc[2 * L - 1] * eb.e4;
  // a chunk without any real layer (columns of fewer than 64 layers) is the identity exactly, not
  // to the rounding of L zero-thickness layers: with up to 63 such chunks in a row that rounding --
  // an ulp of the Planck source each -- would show in fluxes much smaller than the source
  if (pad == L) { uS = 0.0; uD = 0.0; uU = 1.0; dS = 0.0; dD = 1.0; dU = 0.0; }
  TSTAMP(3);
  // ---- bottom-up suffix scan of the projective reflectance recursion: the chunk matrices are
  //      mirrored across the wave (one LDS permute each), which turns it into a prefix scan that
  //      runs over DPP -- 2 LDS round trips instead of 6, and no participation selects
  M7 P;
  P.m00 = wave_reverse<HALF>(uU * dD - uD * dU); P.m02 = wave_reverse<HALF>(uD); P.m10 = wave_reverse<HALF>(uU * dS - uS * dU);
  P.m11 = wave_reverse<HALF>(uU); P.m12 = wave_reverse<HALF>(uS); P.m20 = wave_reverse<HALF>(-dU); P.m22 = 1.0;
  m7_prefix_scan<HALF>(P);
  TSTAMP(4);
  // P applied to (0,0,1): the state above chunk 63-lane; the state below a chunk is that of the
  // chunk under it, one (mirrored) lane down, and 0 under the last chunk
  const double rinv = rcp_nr(P.m22);
  const double rho = wave_reverse<HALF>(wave_shr1<HALF>(P.m02 * rinv)), sig = wave_reverse<HALF>(wave_shr1<HALF>(P.m12 * rinv));
  // ---- top-down affine scan: Din_{q+1} = alpha_q + beta_q*Din_q
  const double mm = rcp_nr(1.0 - rho * dU);
  double sa = dS + dU * mm * (rho * dS + sig);
  double sb = dD * (1.0 + dU * mm * rho);
  {
    double bstep[WSCAN_STEPS];
    wscan_build<HALF>(sa, sb, bstep);
  }
  const double Din = wave_shr1<HALF>(sa);
  const double Uin = mm * (rho * dS + sig + rho * dD * Din);
  TSTAMP(5);

  // ---- level fluxes (:143-148, :288-293), mean intensity (:135-140), g-point weight
#pragma unroll
  for (int t = 0; t < L; t++) {
    const int i = a + t - pad;
    const E4 e = make_e(G[t], X[t]);
    const double y1 = rd[2 * t] + rc[2 * t] * Uin + rl[2 * t] * Din;
    const double y2 = rd[2 * t + 1] + rc[2 * t + 1] * Uin + rl[2 * t + 1] * Din;
    if (t >= pad) {
      double cpb_t = cpb[t], cmb_t = cmb[t], dir_t = dir[t], diru_t = diru[t];
      if constexpr (PARK) {
        cpb_t = mine[3 * t];
        if constexpr (solar) { dir_t = mine[3 * t + 1]; diru_t = mine[3 * t + 2]; }
        else cmb_t = mine[3 * t + 1];
      }
      mine[3 * t] = wcol * (y1 * e.e1 + y2 * e.e2 + cpb_t);
      mine[3 * t + 1] = wcol * ((y1 * e.e3 + y2 * e.e4 + cmb_t) + dir_t);
      mine[3 * t + 2] = wcol * (inv_u1 * (y1 * (e.e1 + e.e3) + y2 * (e.e2 + e.e4) + cpb_t + cmb_t) + diru_t);
      if (i == 0) {
        const double top = (y1 * e.e3 - y2 * e.e4) + cp0_top;
        double *lv0 = lds + (size_t)wc * nl * 3;
        lv0[0] = wcol * top;
        lv0[1] = wcol * lvl0_dn;
        lv0[2] = wcol * (inv_u1 * top + lvl0_am);
      }
    }
  }
  TSTAMP(6);
  // What the last lines need from memory -- the bin's unit factors and the band optical depth this block
  // copies out (a device-scope load, as slow as the first ones) -- is requested HERE, neither earlier (held
  // in registers through the solve they were spilled, and a scratch reload takes as long as the load) nor
  // later (after the barrier the wave would sit through the whole latency)
  asm volatile("" ::: "memory");
  double scale = 1.0, amf = 1.0, band_tau = 0.0;
  if (solar) {
    scale = p.photons_sol[ll] * p.photon_scale_factor;  // clima_radtran.f90:302
    amf = p.am_f1[ll];
  }
  const double amf2 = solar ? p.am_f2[ll] : 1.0, amdw = solar ? p.am_dw[ll] : 1.0;
  if (copies_band && (int)threadIdx.x < nz) band_tau = ld_opr<COHERENT>(&p.tau_band[co.opr + (size_t)l * nz + (nz - 1 - (int)threadIdx.x)]);
  __syncthreads();
  TSTAMP(7);
  // ---- sum over the block's g-points, unit factors (radiate.f90:167-180), reversal (:140-154)
  const bool split = p.accumulate != 0;
  for (int n = threadIdx.x; n < nl; n += blockDim.x) {
    double fu = 0.0, fd = 0.0, am = 0.0;
#pragma unroll
    for (int w = 0; w < NCOL; w++) {
      fu = fu + lds[((size_t)w * nl + n) * 3];
      fd = fd + lds[((size_t)w * nl + n) * 3 + 1];
      am = am + lds[((size_t)w * nl + n) * 3 + 2];
    }
    const size_t o = co.res + (size_t)ll * nl + (nz - n);
    if (solar) {
      fu = fu * scale * p.diurnal_fac;
      fd = fd * scale * p.diurnal_fac;
      am = am * scale * p.diurnal_fac;
      am = am * amf;
      am = am * amf2 * amdw;
      if (split) { atomicAdd(&p.sol_fup_a[o], fu); atomicAdd(&p.sol_fdn_a[o], fd); atomicAdd(&p.sol_amean[o], am); }
      else { p.sol_fup_a[o] = fu; p.sol_fdn_a[o] = fd; p.sol_amean[o] = am; }
    } else {
      const size_t ob = o + (size_t)bz * p.b_out;
      if (split) { atomicAdd(&p.ir_fup_a[ob], fu); atomicAdd(&p.ir_fdn_a[ob], fd); }
      else { p.ir_fup_a[ob] = fu; p.ir_fdn_a[ob] = fd; }
    }
  }
  if (copies_band) {
    double *tb = (solar ? p.sol_tau_band : p.ir_tau_band) + co.res;
    if ((int)threadIdx.x < nz) tb[(size_t)ll * nz + threadIdx.x] = band_tau;
    for (int i = threadIdx.x + blockDim.x; i < nz; i += blockDim.x) tb[(size_t)ll * nz + i] = ld_opr<COHERENT>(&p.tau_band[co.opr + (size_t)l * nz + (nz - 1 - i)]);
  }
  TSTAMP(8);
#undef TSTAMP
}

// one launch for both channels: blocks [0, n_sol) are solar bins (the heavier ones first),
// blocks [n_sol, n_sol+n_ir) IR bins
// LMAX = 8 would take 260 VGPRs (one wave per SIMD); capping it at 256 costs a few spills and buys
// the second wave
template <int LMAX>
__global__ __launch_bounds__(64 * TSW_COLS, LMAX > 4 ? 2 : 1) void k_twostream_w(TwoStreamParams p) {
  extern __shared__ __align__(16) double lds[];  // [TSW_COLS][nz+1][3] weighted level values, then the Planck table
  if ((int)blockIdx.x < p.n_sol) twostream_p_body<LMAX, true, 0, false, false>(p, (int)blockIdx.x, lds, (int)blockIdx.y, (int)blockIdx.z);
  else twostream_p_body<LMAX, false, 0, false, false>(p, (int)blockIdx.x - p.n_sol, lds, (int)blockIdx.y, (int)blockIdx.z);
}

// The half-wave form as a launch of its own (round 3): a wave solves two g-point columns, a block of 4 waves the 8
// g-point columns of group blockIdx.y -- for k-distribution settings of 16, 24, 32 g-points (their calls take one
// launch per kernel) at 65-224 layers: half as many blocks as the whole-wave kernel needs, each adding one addend per
// level instead of two.  16 g-points: 98 -> ~55 us of two-stream launches per call.
template <int L>
__global__ __launch_bounds__(64 * TSW_COLS, 2) void k_twostream_h(TwoStreamParams p) {
  extern __shared__ __align__(16) double lds[];  // [2 TSW_COLS][nz+1][3] weighted level values, then the Planck table
  if ((int)blockIdx.x < p.n_sol) twostream_p_body<L, true, 0, false, false, false, true>(p, (int)blockIdx.x, lds, (int)blockIdx.y, (int)blockIdx.z);
  else twostream_p_body<L, false, 0, false, false, false, true>(p, (int)blockIdx.x - p.n_sol, lds, (int)blockIdx.y, (int)blockIdx.z);
}

static void ts_zero_outputs(const TwoStreamParams &p, hipStream_t s) {
  // partial sums over g-point groups accumulate into zeroed outputs: one launch
  const size_t nl = (size_t)p.nz + 1;
  ZeroParams z;
  z.n = 0;
  if (p.n_ir > 0) {
    z.ptr[z.n] = p.ir_fup_a + (size_t)p.ir_lo * nl; z.count[z.n++] = nl * p.n_ir;
    z.ptr[z.n] = p.ir_fdn_a + (size_t)p.ir_lo * nl; z.count[z.n++] = nl * p.n_ir;
  }
  if (p.n_sol > 0) {
    z.ptr[z.n] = p.sol_fup_a + (size_t)p.sol_lo * nl; z.count[z.n++] = nl * p.n_sol;
    z.ptr[z.n] = p.sol_fdn_a + (size_t)p.sol_lo * nl; z.count[z.n++] = nl * p.n_sol;
    z.ptr[z.n] = p.sol_amean + (size_t)p.sol_lo * nl; z.count[z.n++] = nl * p.n_sol;
  }
  if (z.n > 0) hipLaunchKernelGGL(k_zero, dim3(128, z.n), dim3(256), 0, s, z);
}

// wave-per-column launcher; false when nz needs more than 8 layers per lane
int twostream_w_groups(int ng) { return (ng + TSW_COLS - 1) / TSW_COLS; }

// slots per lane (3..7) when launch_twostream_w() will take the half-wave kernel k_twostream_h for this call, else 0.
// With 8 g-points that kernel stores every output value itself: the caller need not clear the outputs first.
int twostream_w_half_slots(const TwoStreamParams &p) {
  static const bool off = [] { const char *e = getenv("CLIMA_HIP_NO_HALF"); return e && e[0] == '1'; }();
  const int hs = (p.nz + 31) / 32;
  if (off || p.ng < 8 || p.ng % 8 != 0 || hs < 3 || hs > 7 || p.force_slots != 0 || p.nzen > MAX_ZEN) return 0;
  if (sizeof(double) * (3 * 2 * TSW_COLS + 1) * ((size_t)p.nz + 1) > 64 * 1024) return 0;
  // 8 g-points: its blocks are half as many and nearly twice as long as the whole-wave kernel's, which pays when the
  // launch is throughput-bound or when it saves the clearing launch -- IR-only calls on stored opacities (102 / 202
  // layers: 19.3 -> 16.7, 26.1 -> 22.3 us per call) -- and loses when a few bins' blocks are the whole launch (a rank's
  // share of 8: 12.9 -> 15.2 us); profiles/r03_coop_ab.txt
  if (p.ng == 8 && p.n_sol > 0 && p.n_sol + p.n_ir < 512) return 0;
  return hs;
}

bool launch_twostream_w(TwoStreamParams &p, hipStream_t s, size_t *lds_bytes, bool zeroed) {
  const int lmax = std::max((p.nz + 63) / 64, p.force_slots);
  if (lmax > 8) return false;
  const int groups = (p.ng + TSW_COLS - 1) / TSW_COLS;
  const size_t lds = sizeof(double) * (3 * TSW_COLS + 1) * ((size_t)p.nz + 1);  // level values of the columns + the bin's Planck table
  if (lds_bytes) *lds_bytes = lds;
  if (lds > 64 * 1024) return false;
  const int grid = p.n_sol + p.n_ir;
  if (grid <= 0) return true;
  using Kern = void (*)(TwoStreamParams);
  if (const int hs = twostream_w_half_slots(p)) {
    // 8, 16, 24, 32 g-points at 65-224 layers: the half-wave kernel, 8 columns per block.  With 8 g-points a block holds
    // its bin's whole g-point sum and stores it: nothing to clear first, no adds (IR-only calls on stored opacities,
    // few-item calls: one launch and ~1/5 of the instructions fewer than the whole-wave kernel with its two groups)
    static const Kern kh[5] = {k_twostream_h<3>, k_twostream_h<4>, k_twostream_h<5>, k_twostream_h<6>, k_twostream_h<7>};
    const size_t ldsh = sizeof(double) * (3 * 2 * TSW_COLS + 1) * ((size_t)p.nz + 1);
    if (ldsh <= 48 * 1024 || ensure_max_lds((const void *)kh[hs - 3], 64 * 1024)) {
      if (lds_bytes) *lds_bytes = ldsh;
      const int g8 = p.ng / 8;
      if (g8 > 1 && !zeroed) ts_zero_outputs(p, s);
      const int per = g8 <= 2 ? g8 : 1;   // two addends onto zero are order-independent; more go one launch at a time
      for (int g0 = 0; g0 < g8; g0 += per) {
        p.col_base = g0 * 2 * TSW_COLS;
        p.accumulate = g8 > 1 ? 1 : 0;
        hipLaunchKernelGGL(kh[hs - 3], dim3(grid, per, p.b_ncol > 0 ? p.b_ncol : 1), dim3(64 * TSW_COLS), ldsh, s, p);
      }
      return true;
    }
  }
  if (groups > 1 && !zeroed) ts_zero_outputs(p, s);
  static const Kern kern[8] = {k_twostream_w<1>, k_twostream_w<2>, k_twostream_w<3>, k_twostream_w<4>,
                               k_twostream_w<5>, k_twostream_w<6>, k_twostream_w<7>, k_twostream_w<8>};
  if (lds > 48 * 1024 && !ensure_max_lds((const void *)kern[lmax - 1], 64 * 1024)) return false;  // (static LDS on top)
  // Up to two g-point groups go in one launch: two partial sums added into a zeroed output
  // are order-independent.  More groups (ng > 8) run as one launch per group on the same
  // stream, each adding a single addend, so results stay bitwise reproducible.
  const dim3 blk(64 * TSW_COLS);
  const int per_launch = groups <= 2 ? groups : 1;
  for (int g0 = 0; g0 < groups; g0 += per_launch) {
    p.col_base = g0 * TSW_COLS;
    p.accumulate = groups > 1 ? 1 : 0;
    const dim3 g(grid, per_launch, p.b_ncol > 0 ? p.b_ncol : 1);
    hipLaunchKernelGGL(kern[lmax - 1], g, blk, lds, s, p);  // layer slots per lane = ceil(nz/64)
  }
  return true;
}




__global__ void k_test_wscan(const double *a, const double *b, double *out, int nwaves) {
  // out[0]: inclusive affine scan x_i = a_i + b_i x_{i-1}; out[1]: the same through build + apply;
  // out[2]: x shifted up by one lane; out[3]: reversed
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwaves * 64) return;
  const size_t n = (size_t)nwaves * 64;
  double x = a[i], y = b[i], bstep[WSCAN_STEPS];
  wscan_build(x, y, bstep);
  out[i] = x;
  out[n + i] = wscan_apply(a[i], [&](int k) { return bstep[k]; });
  out[2 * n + i] = wave_shr1(a[i]);
  out[3 * n + i] = wave_reverse(a[i]);
}
void launch_test_wscan(const double *a, const double *b, double *out, int nwaves, hipStream_t s) {
  hipLaunchKernelGGL(k_test_wscan, dim3(nwaves), dim3(64), 0, s, a, b, out, nwaves);
}

// ------------------------------------------------------------------------------------
// k_twostream_ir_batch: many temperature columns on ONE set of opacities (the RCE Jacobian's
// radiative work, src/adiabat/clima_adiabat_solve.f90:798-812).  Everything in two_stream_ir
// that does not depend on temperature -- gamma's, lambda, exp(-lambda tau), the elimination of
// the tridiagonal system, the reflectances that join the lane chunks -- is done once per
// (bin, g-point); a temperature column then costs only its Planck values (computed once per
// bin and shared by the g-point waves through LDS), the linear source terms, the right-hand
// side sweeps and two affine wave scans: ~200 instructions per lane instead of ~1700.
//
// Same decomposition as twostream_p_body (lane q owns layers [q nz/64, (q+1) nz/64), flux
// boundary conditions per chunk); written as
//     y_r = alpha_r + beta_r Uin + gamma_r Din            inside a chunk,
//     sig_above_q = A_q + Bq sig_above_{q+1}              bottom-up (source seen from above),
//     Din_{q+1}   = sa_q + sb_q Din_q                     top-down,
// with beta, gamma, B, sb and the reflectances rho fixed by the opacities.
// One block = the (up to) 8 g-point waves of one bin; it loops over its share of the columns.
// ------------------------------------------------------------------------------------
constexpr int IRB_TILE = 8;    // columns whose Planck values sit in LDS together
// (working on two columns at once, to interleave their dependent chains, spilled ~100 more
// registers and measured slower: 13.5 vs 10.0 us per column)
//
// NW = g-point waves per block.  Up to 4 layer slots per lane (nz <= 256) a block is 8 waves -- all g-points of
// the usual k-distribution at once, two waves per SIMD, 256 registers each.  5-8 slots (257-512 layers:
// AdiabatClimate's doubled radiative grid at nz = 200 is 402) need ~300 registers per lane: those instances are
// blocks of FOUR waves, one wave per SIMD, which gives the register allocator the SIMD's whole file (256 VGPRs +
// 256 AGPRs as spill space: a register-to-register move where scratch would be an L2 round trip).  The g-points
// are taken in groups of NW, one group after the other, and a group adds its weighted level fluxes onto what the
// groups before it left in the output (the same thread wrote it): the sum over the g-points keeps the order
// ((f0 + f1) + f2) + ... whatever NW is, so both forms agree to rounding (hipcc contracts a few products differently in the two instances) -- and any
// g-point count is covered.
template <int L, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k_twostream_ir_batch(TwoStreamParams p, int ncol, int cols_per_block) {
  extern __shared__ __align__(16) double lds[];
  const int nz = p.nz, ng = p.ng, nl = nz + 1;
  double *sB = lds;                          // [IRB_TILE][nl] Planck, TOA-first levels
  double *sF0 = lds + (size_t)IRB_TILE * nl;  // [2 columns][2][NW][nl + 1] weighted level fluxes (+ a dump entry per row)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ll = p.ir_lo + (int)blockIdx.x;
  const int l = p.ir_start + ll;
  // slots as in twostream_p_body: L per lane, a short chunk padded from the top with
  // zero-thickness layers, so that nothing below branches on the chunk's length
  const int a = (lane * nz) >> 6, b = ((lane + 1) * nz) >> 6, pad = L - (b - a);
  const bool is_toa = lane == 0, is_sfc = lane == 63, empty = pad == L;
  const double avg_freq = 0.5 * (p.freq[l] + p.freq[l + 1]);
  const bool hard = p.has_hard_surface != 0;
  const double emis = hard ? p.emissivity[ll] : 0.0;
  const double Rsfc = hard ? 1.0 - emis : 0.0;  // twostream.f90:186-190
  const int c_begin = (int)blockIdx.z * cols_per_block;
  const int c_end = min(ncol, c_begin + cols_per_block);

  // (the 8-wave form covers ng <= 8 in one group: written as a loop it cost the 4-slot instance 28 scratch accesses)
  for (int g0 = 0; g0 < (NW == 8 ? 1 : ng); g0 += NW) {
  const bool col_on = g0 + wave < ng;
  const int cg = col_on ? g0 + wave : ng - 1;
  const double wcol = col_on ? p.wbin[cg] : 0.0;
  const double *tauL = p.tau + ((size_t)l * ng + cg) * nz;
  const double *w0L = p.w0 + ((size_t)l * ng + cg) * nz;
  const double *gL = p.g + (size_t)l * nz;

  // ---- temperature-independent part -------------------------------------------------
  double G[L], X[L], itau[L], rq[L], tauv[L];
  double rr[2 * L], ar[2 * L], cc[2 * L], be[2 * L], ga[2 * L];
  {
#pragma unroll
    for (int t = 0; t < L; t++) {
      const bool real = t >= pad;
      const int i = min(max(a + t - pad, 0), nz - 1);
      const double tau_in = real ? tauL[i] : 0.0, w0_in = real ? w0L[i] : 0.0, gt_in = real ? gL[i] : 0.0;
      const double gam1 = 2.0 - w0_in * (1.0 + gt_in);  // :195-201
      const double gam2 = w0_in * (1.0 - gt_in);
      const double lam = sqrt_nr(gam1 * gam1 - gam2 * gam2);
      G[t] = gam2 * rcp_nr(gam1 + lam);
      X[t] = fast_exp(-lam * tau_in);
      itau[t] = (tau_in <= p.ir_tau_min || !real) ? 0.0 : 1.0 / tau_in;  // 0 marks the thin-layer source rule (:216-227)
      rq[t] = rcp_nr(gam1 + gam2);
      tauv[t] = tau_in;
    }
    E4 u = make_e(G[0], X[0]);
    double cp, lp;
    {
      const double A = is_toa ? 0.0 : -1.0;
      const double r = rcp_nr(u.e1);
      rr[0] = r; ar[0] = A * r;
      cp = (-u.e2) * r; lp = A * r;
      cc[0] = cp; ga[0] = lp;
    }
#pragma unroll
    for (int t = 1; t < L; t++) {
      const E4 v = make_e(G[t], X[t]);
      double A = v.e2 * u.e1 - u.e3 * v.e4, B = u.e2 * v.e2 - u.e4 * v.e4, D = v.e1 * v.e4 - v.e2 * v.e3;
      double r = rcp_nr(B - A * cp);
      rr[2 * t - 1] = r; ar[2 * t - 1] = A * r;
      const double cn = D * r, ln = (-A * lp) * r;
      cc[2 * t - 1] = cn; ga[2 * t - 1] = ln;
      A = u.e2 * u.e3 - u.e4 * u.e1; B = u.e1 * v.e1 - u.e3 * v.e3; D = u.e3 * v.e4 - u.e1 * v.e2;
      r = rcp_nr(B - A * cn);
      rr[2 * t] = r; ar[2 * t] = A * r;
      cp = D * r; lp = (-A * ln) * r;
      cc[2 * t] = cp; ga[2 * t] = lp;
      u = v;
    }
    {
      const double A = is_sfc ? u.e1 - Rsfc * u.e3 : u.e1;
      const double B = is_sfc ? u.e2 - Rsfc * u.e4 : u.e2;
      const double D = is_sfc ? 0.0 : -1.0;
      const double r = rcp_nr(B - A * cp);
      rr[2 * L - 1] = r; ar[2 * L - 1] = A * r;
      cc[2 * L - 1] = D * r; ga[2 * L - 1] = (-A * lp) * r;
    }
  }
  // upward sweep of the Uin / Din coefficients (ga holds the l_r of the downward pass on entry)
  {
    double bev = 1.0, gav = 0.0;
#pragma unroll
    for (int r = 2 * L - 1; r >= 0; r--) {
      const double c_ = cc[r], lc = ga[r];
      bev = -c_ * bev; gav = -lc - c_ * gav;
      be[r] = bev; ga[r] = gav;
    }
  }
  // the chunk as a map of (Din, Uin): temperature-independent entries (a chunk without a real
  // layer is the identity exactly)
  const E4 ea = make_e(G[0], X[0]), eb = make_e(G[L - 1], X[L - 1]);
  const double ea3 = ea.e3, ea4 = ea.e4, eb3 = eb.e3, eb4 = eb.e4;
  const double uD = empty ? 0.0 : ga[0] * ea3 - ga[1] * ea4, uU = empty ? 1.0 : be[0] * ea3 - be[1] * ea4;
  const double dD = empty ? 1.0 : ga[2 * L - 2] * eb3 + ga[2 * L - 1] * eb4, dU = empty ? 0.0 : be[2 * L - 2] * eb3 + be[2 * L - 1] * eb4;
  // reflectance seen from above every interface: projective suffix scan of 2x2 matrices
  double rho;
  {
    double m00 = uU * dD - uD * dU, m02 = uD, m20 = -dU, m22 = 1.0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const double r00 = __shfl_down(m00, d), r02 = __shfl_down(m02, d), r20 = __shfl_down(m20, d), r22 = __shfl_down(m22, d);
      if (lane + d < 64) {
        const double n00 = m00 * r00 + m02 * r20, n02 = m00 * r02 + m02 * r22;
        const double n20 = m20 * r00 + m22 * r20, n22 = m20 * r02 + m22 * r22;
        m00 = n00; m02 = n02; m20 = n20; m22 = n22;
      }
    }
    const double rho_above = m02 * rcp_nr(m22);
    rho = __shfl_down(rho_above, 1);
    if (lane == 63) rho = 0.0;
  }
  const double mm = rcp_nr(1.0 - rho * dU);
  // scan coefficients: sig_above_q = A_q + Bq*sig_above_{q+1};  Din_{q+1} = sa_q + sb_q*Din_q
  // Both scans run over DPP (wscan_*); the bottom-up one is a prefix scan of the lane-reversed
  // data.  Their per-step multipliers live in LDS, one slot per thread and step: they are read
  // once per column and would otherwise cost 28 VGPRs.
  // (the four-wave form has the SIMD's whole register file: there they stay in registers)
  // (measured for the 1- and 2-slot instances of the 8-wave form as well, to fit two blocks per CU: 4.6 -> 4.8 us per
  // column at 102 layers with the table gone, 10.3 when also compiled for four waves per SIMD -- spills)
  constexpr bool STEP_REGS = NW == 4;
  double *sStep = sF0 + (size_t)4 * NW * (nl + 1) + threadIdx.x;  // [2 * WSCAN_STEPS][blockDim.x]
  double stA[WSCAN_STEPS], stB[WSCAN_STEPS];
  {
    double dummy = 0.0, bq = wave_reverse(uU * mm);
    wscan_build(dummy, bq, stA);
    double sb = dD * (1.0 + dU * mm * rho);
    dummy = 0.0;
    wscan_build(dummy, sb, stB);
    if constexpr (!STEP_REGS) {
#pragma unroll
      for (int k = 0; k < WSCAN_STEPS; k++) {
        sStep[(size_t)k * blockDim.x] = stA[k];
        sStep[(size_t)(WSCAN_STEPS + k) * blockDim.x] = stB[k];
      }
    }
  }
  const double kA = uU * mm * rho;   // A_q = uS + kA*dS
  const double kS = dU * mm;         // sa_q = dS + kS*(rho*dS + sig)
  // the L+1 faces of the slots: level a for every face of a zero-thickness slot
  int face[L + 1];
#pragma unroll
  for (int s = 0; s <= L; s++) face[s] = a + max(s - pad, 0);

  // ---- the columns ---------------------------------------------------------------------
  for (int c0 = c_begin; c0 < c_end; c0 += IRB_TILE) {
    const int nt = min(IRB_TILE, c_end - c0);
    __syncthreads();  // the previous tile's Planck values are no longer read
    for (int idx = threadIdx.x; idx < nt * nl; idx += blockDim.x) {
      const int cj = idx / nl, n = idx - cj * nl;
      const double *Tc = p.T + (size_t)(c0 + cj) * p.b_T;
      const double temp = (n == nz) ? p.T_surface[(size_t)(c0 + cj) * p.b_Ts] : Tc[nz - 1 - n];  // radiate.f90:65-69
      sB[(size_t)cj * nl + n] = p.bplanck ? p.bplanck[n] : planck_fcn(avg_freq, temp);
    }
    __syncthreads();
    for (int cj = 0; cj < nt; cj++) {
      const double *Bc = sB + (size_t)cj * nl;
      double Bf[L + 1];
#pragma unroll
      for (int s = 0; s <= L; s++) Bf[s] = Bc[face[s]];
      double dd[2 * L], cpbv[L], cmbv[L];
      double cp0_top, b1n_last = 0.0;
      {
        double dpv = 0.0;
        E4 u = make_e(G[0], X[0]);
#pragma unroll
        for (int t = 0; t < L; t++) {
          const double Bt = Bf[t], Bb = Bf[t + 1];
          const double b1n = (Bb - Bt) * itau[t];               // :216-227
          const double b0n = (itau[t] == 0.0) ? 0.5 * (Bt + Bb) : Bt;
          const double cp0 = PI * (b0n + b1n * rq[t]);          // :229-232, norm = 2 pi * 1/2
          cpbv[t] = PI * (b0n + b1n * (tauv[t] + rq[t]));
          const double cm0 = PI * (b0n - b1n * rq[t]);
          cmbv[t] = PI * (b0n + b1n * (tauv[t] - rq[t]));
          if (t == 0) {
            cp0_top = cp0;
            dpv = (0.0 - cm0) * rr[0];
            dd[0] = dpv;
          } else {
            const E4 v = make_e(G[t], X[t]);
            const double E1 = v.e2 * (cp0 - cpbv[t - 1]) - v.e4 * (cm0 - cmbv[t - 1]);
            const double dn = E1 * rr[2 * t - 1] - ar[2 * t - 1] * dpv;
            dd[2 * t - 1] = dn;
            const double E2 = u.e3 * (cp0 - cpbv[t - 1]) + u.e1 * (cmbv[t - 1] - cm0);
            dpv = E2 * rr[2 * t] - ar[2 * t] * dn;
            dd[2 * t] = dpv;
            u = v;
          }
          if (t == L - 1) b1n_last = b1n;
        }
        // last row: surface (:236-247) or the flux condition
        const double Ssfc = hard ? emis * PI * Bf[L] : PI * (Bf[L] + 0.5 * b1n_last);
        const double E = is_sfc ? Ssfc - cpbv[L - 1] + Rsfc * cmbv[L - 1] : 0.0 - cpbv[L - 1];
        dd[2 * L - 1] = E * rr[2 * L - 1] - ar[2 * L - 1] * dpv;
      }
      // upward: alpha_r (overwrites dd)
      {
        double al = 0.0;
#pragma unroll
        for (int r = 2 * L - 1; r >= 0; r--) {
          al = dd[r] - cc[r] * al;
          dd[r] = al;
        }
      }
      const double uS = empty ? 0.0 : dd[0] * ea3 - dd[1] * ea4 + cp0_top;
      const double dS = empty ? 0.0 : dd[2 * L - 2] * eb3 + dd[2 * L - 1] * eb4 + cmbv[L - 1];
      // bottom-up: source seen from above each interface (prefix scan in lane-reversed order),
      // then top-down: diffuse flux entering each chunk from above
      const double sgr = wscan_apply(wave_reverse(uS + kA * dS),
                                     [&](int k) { return STEP_REGS ? stA[k] : sStep[(size_t)k * blockDim.x]; });
      const double sig = wave_reverse(wave_shr1(sgr));   // below chunk q: what chunk q+1 shows from above; 0 under the last
      const double sa = wscan_apply(dS + kS * (rho * dS + sig),
                                    [&](int k) { return STEP_REGS ? stB[k] : sStep[(size_t)(WSCAN_STEPS + k) * blockDim.x]; });
      const double Din = wave_shr1(sa);
      const double Uin = mm * (rho * dS + sig + rho * dD * Din);
      // level fluxes (:288-293), g-point weight; consecutive columns alternate the two staging buffers
      const int buf = cj & 1;
      double *sFu = sF0 + (size_t)buf * 2 * NW * (nl + 1) + (size_t)(0 * NW + wave) * (nl + 1);
      double *sFd = sF0 + (size_t)buf * 2 * NW * (nl + 1) + (size_t)(1 * NW + wave) * (nl + 1);
      double top_up = 0.0;
#pragma unroll
      for (int t = 0; t < L; t++) {
        const int i = a + t - pad;
        const E4 e = make_e(G[t], X[t]);
        const double y1 = dd[2 * t] + be[2 * t] * Uin + ga[2 * t] * Din;
        const double y2 = dd[2 * t + 1] + be[2 * t + 1] * Uin + ga[2 * t + 1] * Din;
        // (no branch per slot: a zero-thickness slot's values go to a dump entry behind the wave's rows)
        const int lv = t >= pad ? i + 1 : nl;
        sFu[lv] = wcol * (y1 * e.e1 + y2 * e.e2 + cpbv[t]);
        sFd[lv] = wcol * (y1 * e.e3 + y2 * e.e4 + cmbv[t]);
        // up-flux through the top of the lane's slot 0: in lane 0 the column's top (a zero-thickness slot above the
        // first real layer hands the flux through)
        if (t == 0) top_up = (y1 * e.e3 - y2 * e.e4) + cp0_top;
      }
      if (is_toa) {
        sFu[0] = wcol * top_up;
        sFd[0] = 0.0;
      }
      __syncthreads();
      // sum over the g-points (in g order: a later group continues the sum the earlier ones stored), reversal to
      // ground-first (radiate.f90:140-154)
      for (int n = threadIdx.x; n < nl; n += blockDim.x) {
        const double *sF = sF0 + (size_t)buf * 2 * NW * (nl + 1);
        const size_t o = (size_t)(c0 + cj) * p.b_out + (size_t)ll * nl + (nz - n);
        double fu = 0.0, fd = 0.0;
        if (g0 > 0) { fu = p.ir_fup_a[o]; fd = p.ir_fdn_a[o]; }
#pragma unroll
        for (int w = 0; w < NW; w++) {
          fu = fu + sF[(size_t)(0 * NW + w) * (nl + 1) + n];
          fd = fd + sF[(size_t)(1 * NW + w) * (nl + 1) + n];
        }
        p.ir_fup_a[o] = fu;
        p.ir_fdn_a[o] = fd;
      }
    }
  }
  __syncthreads();   // the next g-point group reuses the staging buffers and the step tables
  }
}

// false when the configuration is outside what the kernel covers: more than 8 layer slots per lane (nz > 512)
// or an LDS image beyond 160 KiB.  force_nw (test hook): 4 selects the four-wave form for 1-4 slots too.
bool launch_twostream_ir_batch(TwoStreamParams &p, int ncol, hipStream_t s, int force_nw) {
  const int lmax = std::max((p.nz + 63) / 64, p.force_slots);
  if (lmax > 8 || p.n_ir <= 0 || ncol <= 0 || p.ng < 1) return false;
  const int nw = (lmax > 4 || p.ng > 8 || force_nw == 4) ? 4 : 8;
  const size_t lds = sizeof(double) * ((size_t)IRB_TILE * ((size_t)p.nz + 1) + (size_t)4 * nw * ((size_t)p.nz + 2) + (nw == 8 ? 2 * WSCAN_STEPS * 64 * nw : 0));
  if (lds > 160 * 1024) return false;
  using Kern = void (*)(TwoStreamParams, int, int);
  static const Kern kern8[4] = {k_twostream_ir_batch<1, 8>, k_twostream_ir_batch<2, 8>, k_twostream_ir_batch<3, 8>, k_twostream_ir_batch<4, 8>};
  static const Kern kern4[8] = {k_twostream_ir_batch<1, 4>, k_twostream_ir_batch<2, 4>, k_twostream_ir_batch<3, 4>, k_twostream_ir_batch<4, 4>,
                                k_twostream_ir_batch<5, 4>, k_twostream_ir_batch<6, 4>, k_twostream_ir_batch<7, 4>, k_twostream_ir_batch<8, 4>};
  const Kern k = nw == 8 ? kern8[lmax - 1] : kern4[lmax - 1];
  if (!ensure_max_lds((const void *)k)) return false;
  // enough blocks to fill the chip a few times over, each with a worthwhile run of columns
  int cpb = (ncol + 3) / 4;
  if (cpb < IRB_TILE) cpb = std::min(ncol, IRB_TILE);
  const dim3 grid(p.n_ir, 1, (ncol + cpb - 1) / cpb), blk(64 * nw);
  hipLaunchKernelGGL(k, grid, blk, lds, s, p, ncol, cpb);  // layer slots per lane = ceil(nz/64)
  return true;
}

// ------------------------------------------------------------------------------------
#ifndef FUSED_NZMAX
#define FUSED_NZMAX 8
#endif
// k_fused: opacity and two-stream work of one call -- or of a batch of columns -- in ONE grid, one
// workgroup per work item.  The items of column c are the blocks [c*(n_op+n_ts), (c+1)*(n_op+n_ts)):
// first its n_op opacity tiles, then its n_ts two-stream items (one per (bin, g-point group), ordered
// by readiness), each of which first waits until the opacity tiles that cover its bin have
// published.  The opacity code holds two waves per SIMD and its second residency round is half
// empty; here the two-stream blocks move into those slots as soon as they free up instead of
// waiting for the whole launch to drain -- and in a batch, one column's two-stream tail runs beside
// the next column's opacity tiles.
// Forward progress: workgroups are dispatched in index order (a property of this hardware's
// dispatcher, not of the programming model: DESIGN.md records it as a dependency), so whatever a
// two-stream block waits for is already running or done, and opacity tiles wait for nothing.  The
// wait is bounded all the same: on expiry the block stamps the timeout word and returns, and the
// host computes the call again through the separate launches.
// (A persistent form -- two blocks per CU taking items off a queue, which needs no dispatch-order
// assumption and no workgroup launch per ~10 us two-stream item -- was built and measured: with both
// bodies inlined into the item loop hipcc spilled 220-470 registers; with the bodies outlined as
// functions every access went through generic-address-space pointers (flat loads, no scalar loads)
// and the call took 178 us instead of 104.)
// LSEL = 0: the two-stream part carries the 2-, 3- and 4-slot forms (fp.slots selects); LSEL = 5..8:
// that one slot count (columns of 257-512 layers), a kernel of its own so that its register needs do
// not disturb the allocation of the others.
// HALF: the two-stream items are one block per bin, its waves solving two g-point columns each
// (twostream_p_body's half-wave form) with LSEL = ceil(nz/32) slots per lane.
template <int RM, bool CUSTOM, int LSEL, bool PAIRED = false, bool HALF = false>
__global__ __launch_bounds__(OP_THREADS, 2) void k_fused(OpacityParams op, TwoStreamParams ts, FusedParams fp) {
  extern __shared__ __align__(16) double lds[];
  const int per_col = fp.n_op + fp.n_ts;
  const int cb = (int)blockIdx.x / per_col, loc = (int)blockIdx.x - cb * per_col;
  if (loc < fp.n_op) {
    // ---- an opacity tile of column cb
    // The opacity tiles' waves go first where a SIMD holds a wave of either kind (round 4, A/B on one box, three rounds
    // each: config 2 89.4 against 90.0 us per call, config 3 82.5 against 82.9, config 5 even; priority 3 the same as 1;
    // the two-stream blocks' waves first instead: 91.2).  The tiles are the kernel's critical path -- every two-stream
    // block waits for some of them -- and a two-stream wave is latency-bound whichever slot it gets.
    __builtin_amdgcn_s_setprio(1);
    const size_t oc = (size_t)cb * fp.bs.col;
    if ((long)loc * OP_THREADS < (long)op.nbins * op.col.meta[2 * oc])  // tiles past the compacted lane range are empty
      opacity8_body<RM, CUSTOM, true>(op, loc, oc, (size_t)cb * fp.bs.prep, (size_t)cb * fp.bs.opr);
    // the opr stores above went out at device scope (write-through); wait until they are
    // acknowledged, then let every wave of the block arrive before the flag goes up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(&fp.done[(size_t)cb * fp.bs.done + loc], fp.call_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- a two-stream item of column cb
  const int b = loc - fp.n_op;
  int gy, bl;
  {
    // Dispatch order of the two-stream blocks = block index order, so the index is mapped to work by
    // readiness: first the solar bins whose opacities come out of the first residency round of
    // opacity tiles (both g-point groups), then the IR bins (all of them are in that round or
    // early in the second), last the solar bins of the second round.  A block whose opacities are
    // not out yet holds its slot while it waits; in plain (group, channel, bin) order the late solar
    // bins of group 0 sat in front of ready IR work (-0.8 us per call).
    const int nS = ts.n_sol, nI = ts.n_ir, E = fp.sol_early, Lt = nS - E;
    if constexpr (HALF) {   // one group: early solar, IR, late solar
      gy = 0;
      bl = b < E ? b : b < E + nI ? nS + (b - E) : b - nI;
    } else {
      const int seg[6] = {E, E, nI, nI, Lt, Lt};
      int r = b, k = 0;
      while (k < 5 && r >= seg[k]) { r -= seg[k]; k++; }
      gy = k & 1;
      bl = k < 2 ? r : k < 4 ? nS + r : E + r;
    }
  }
  const bool solar = bl < ts.n_sol;
  const int ll = solar ? ts.sol_lo + bl : ts.ir_lo + (bl - ts.n_sol);
  const int l = (solar ? ts.sol_start : ts.ir_start) + ll;
#ifdef CLIMA_STAMPS
  if (op.stamps && threadIdx.x == 0 && cb == 0) op.stamps[64 + 2 * 3128 + 3 * b] = __builtin_amdgcn_s_memrealtime();
#endif
  // The wait for the opacity tiles that cover this item's bin: thread 0 polls, everyone meets at the barrier.
  // (Round 3, measured on config 2: with NO wait at all -- an unsafe build, for the timing only -- the call takes
  // 90.45 instead of 91.40 us, so the whole hand-off is worth under 1 us; a form in which every wave issued its
  // loads together with the flag reads and validated them against a completion time stamp of the tile
  // (s_memrealtime), one device round trip instead of two, kept its first loads in 1200 of 1200 blocks and was
  // 1.2 us SLOWER: 92.4 us.  A two-stream block's start-up latency is hidden by the other wave of its SIMD.)
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    const int nsrc = op.col.meta[2 * (size_t)cb * fp.bs.col];
    const long t0 = (long)(l - op.bin_lo) * nsrc;
    const int d0 = max((int)(t0 / OP_THREADS), 0), d1 = min((int)((t0 + nsrc - 1) / OP_THREADS), fp.n_op - 1);
    const int *done = fp.done + (size_t)cb * fp.bs.done;
    int ok = 1;
    for (int d = d0; d <= d1 && ok; d++) {
      int spins = 0;
      while (__hip_atomic_load(&done[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != fp.call_id) {
        __builtin_amdgcn_s_sleep(16);
        if (++spins > fp.max_spins) { ok = 0; break; }
      }
    }
    if (!ok) atomicMax(fp.timeout_flag, fp.call_id);  // its own word: the host re-issues the call unfused
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) return;
#ifdef CLIMA_STAMPS
  if (op.stamps && threadIdx.x == 0 && cb == 0) op.stamps[64 + 2 * 3128 + 3 * b + 1] = __builtin_amdgcn_s_memrealtime();
  // two blocks whose phases are stamped: late ones of the whole-wave form; a solar and an IR one of the half-wave form
  const int tslot = HALF ? (b == fp.sol_early / 2 ? 32 : (b == fp.sol_early + ts.n_ir / 2 ? 48 : -1))
                         : (b == 1500) ? 32 : (b == 2200 ? 48 : -1);
#else
  const int tslot = -1;
#endif
  const TsOfs co{(size_t)cb * fp.bs.opr, (size_t)cb * fp.bs.col, (size_t)cb * fp.bs.res};
  // The number of slots per lane follows the column height (65-128 layers: 2, up to 192: 3, up to
  // 256: 4; 5-8 in the LSEL kernels).  Columns of at most 64 layers are not fused at all
  // (fused_supported): their opacity tiles do not fill the machine once, so there is no half-empty
  // second round to fill, and the stand-alone one-slot two-stream kernel runs at five waves per SIMD
  // instead of two.
  if constexpr (HALF) {
    if (solar) twostream_p_body<LSEL, true, FUSED_NZMAX, true, false, false, true>(ts, bl, lds, gy, 0, tslot, co);
    else twostream_p_body<LSEL, false, 0, true, false, false, true>(ts, bl - ts.n_sol, lds, gy, 0, tslot, co);
  } else if constexpr (LSEL == 0 && PAIRED) {
    if (fp.slots == 2) {
      if (solar) twostream_p_body<2, true, FUSED_NZMAX, true, true, true>(ts, bl, lds, gy, 0, tslot, co);
      else twostream_p_body<2, false, 0, true, true, true>(ts, bl - ts.n_sol, lds, gy, 0, tslot, co);
    } else {
      if (solar) twostream_p_body<4, true, FUSED_NZMAX, true, true, true>(ts, bl, lds, gy, 0, tslot, co);
      else twostream_p_body<4, false, 0, true, true, true>(ts, bl - ts.n_sol, lds, gy, 0, tslot, co);
    }
  } else if constexpr (LSEL == 0) {
    if (fp.slots == 2) {
      if (solar) twostream_p_body<2, true, FUSED_NZMAX, true, true>(ts, bl, lds, gy, 0, tslot, co);
      else twostream_p_body<2, false, 0, true, true>(ts, bl - ts.n_sol, lds, gy, 0, tslot, co);
    } else if (fp.slots == 3) {
      if (solar) twostream_p_body<3, true, FUSED_NZMAX, true, true>(ts, bl, lds, gy, 0, tslot, co);
      else twostream_p_body<3, false, 0, true, true>(ts, bl - ts.n_sol, lds, gy, 0, tslot, co);
    } else {
      if (solar) twostream_p_body<4, true, FUSED_NZMAX, true, true>(ts, bl, lds, gy, 0, tslot, co);
      else twostream_p_body<4, false, 0, true, true>(ts, bl - ts.n_sol, lds, gy, 0, tslot, co);
    }
  } else {
    if (solar) twostream_p_body<LSEL, true, FUSED_NZMAX, true, false, PAIRED>(ts, bl, lds, gy, 0, tslot, co);
    else twostream_p_body<LSEL, false, 0, true, false, PAIRED>(ts, bl - ts.n_sol, lds, gy, 0, tslot, co);
  }
#ifdef CLIMA_STAMPS
  __syncthreads();
  if (op.stamps && threadIdx.x == 0 && cb == 0) op.stamps[64 + 2 * 3128 + 3 * b + 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

// false when the configuration is outside what the fused form covers (the caller then uses the
// separate launches)
bool fused_supported(const OpacityParams &op, const TwoStreamParams &ts) {
  const int slots = (ts.nz + 63) / 64;
  if (op.ng != 8 || slots < 2 || slots > 8 || ts.nzen > MAX_ZEN) return false;
  if (slots > 4 && (op.rebin_mode != 0 || op.cust.on)) return false;  // the 5-8 slot kernels exist for the default form only
  return (long)op.nbins * op.nz > 0 && ts.n_sol + ts.n_ir > 0;
}

int fused_tiles(const OpacityParams &op) {
  return (int)(((long)op.nbins * op.nsrc + OP_THREADS - 1) / OP_THREADS);
}

using FusedKern = void (*)(OpacityParams, TwoStreamParams, FusedParams);
// slots of the paired form: twice the pair slots, ceil((nz/2)/64); 0 when the form does not apply
static int paired_slots(const OpacityParams &op, const TwoStreamParams &ts) {
  if (!ts.paired || (ts.nz & 1) || op.rebin_mode != 0 || op.cust.on) return 0;
  const int s = 2 * ((ts.nz / 2 + 63) / 64);
  return (s == 2 || s == 4 || s == 6 || s == 8) ? s : 0;
}
static FusedKern fused_kernel_paired(int slots) {
  return slots <= 4 ? (FusedKern)k_fused<0, false, 0, true> : slots == 6 ? (FusedKern)k_fused<0, false, 6, true> : (FusedKern)k_fused<0, false, 8, true>;
}
// slots per lane of the half-wave two-stream form, ceil(nz/32); 0 when the form does not apply
// (CLIMA_HIP_NO_HALF=1 switches it off: the A/B switch of tools/gpu_ab.sh)
static int half_slots(const OpacityParams &op, const TwoStreamParams &ts) {
  static const bool off = [] { const char *e = getenv("CLIMA_HIP_NO_HALF"); return e && e[0] == '1'; }();
  if (off || ts.ng != 8 || op.rebin_mode != 0 || op.cust.on) return 0;
  const int s = (ts.nz + 31) / 32;   // 65-224 layers: 3-7 slots (225-256 would take 8: the whole-wave form's 4 fill the lanes as well)
  return (s >= 3 && s <= 7) ? s : 0;
}
// slots of the half-wave form when launch_fused() will take it for this call, else 0.  Its blocks hold all 8
// g-points of a bin and store every output value themselves: the caller need not clear the outputs first.
int fused_half_form(const OpacityParams &op, const TwoStreamParams &ts, int ncol) {
  (void)ncol;
  if (!fused_supported(op, ts)) return 0;
  return half_slots(op, ts);   // (taken before the paired form wherever both apply: launch_fused)
}
static FusedKern fused_kernel_half(int slots) {
  static const FusedKern k[5] = {k_fused<0, false, 3, false, true>, k_fused<0, false, 4, false, true>, k_fused<0, false, 5, false, true>,
                                 k_fused<0, false, 6, false, true>, k_fused<0, false, 7, false, true>};
  return k[slots - 3];
}
static FusedKern fused_kernel(const OpacityParams &op, int slots) {
  static const FusedKern k04[2][3] = {{k_fused<0, false, 0>, k_fused<1, false, 0>, k_fused<2, false, 0>},
                                      {k_fused<0, true, 0>, k_fused<1, true, 0>, k_fused<2, true, 0>}};
  static const FusedKern k58[4] = {k_fused<0, false, 5>, k_fused<0, false, 6>, k_fused<0, false, 7>, k_fused<0, false, 8>};
  return slots <= 4 ? k04[op.cust.on ? 1 : 0][op.rebin_mode] : k58[slots - 5];
}

// fp.ncol, fp.bs, fp.call_id, fp.max_spins, fp.done, fp.timeout_flag come from the caller.
// op.nsrc = source layers per column (nz for a batch, where the tile count is an upper bound).
bool launch_fused(const OpacityParams &op, TwoStreamParams &ts, FusedParams fp, hipStream_t s) {
  if (!fused_supported(op, ts)) return false;
  const int nb = ts.n_sol + ts.n_ir;
  const int groups = (ts.ng + TSW_COLS - 1) / TSW_COLS;  // 2
  fp.n_op = fused_tiles(op);
  fp.n_ts = nb * groups;
  fp.slots = (ts.nz + 63) / 64;  // 2..8 (fused_supported)
  // 65-224 layers: the half-wave form, also on an all-pairs grid (round 3, measured on AdiabatClimate's doubled grids:
  // 102 layers 66.3 against 69.1 us per call in the paired whole-wave form, 202 layers 78.2 against 80.6); the paired
  // form takes the all-pairs grids beyond (402 layers: 131 against ~145 unpaired)
  const int hs = half_slots(op, ts);
  const int ps = (!hs && fp.ncol <= 1) ? paired_slots(op, ts) : 0;   // (a batch's columns are not all pairs)
  if (ps) fp.slots = ps;
  if (hs) { fp.slots = hs; fp.n_ts = nb; }
  if (fp.ncol < 1) fp.ncol = 1;
  {
    // solar bins (of this shard) whose opacity tiles sit in the first residency round: two blocks per CU
    const long first_round_bins = ((long)2 * device_cus() * OP_THREADS) / std::max(op.nsrc, 1);
    fp.sol_early = (int)std::min<long>(ts.n_sol, std::max<long>(0, first_round_bins - (long)(ts.sol_start + ts.sol_lo - op.bin_lo)));
  }
  ts.col_base = 0; ts.accumulate = hs ? 0 : 1;   // (a half-wave block holds all 8 g-points of its bin)
  ts.scat = op.scat; ts.w0_from_scat = op.write_w0 ? 0 : 1;
  const size_t lds = sizeof(double) * (3 * TSW_COLS * (hs ? 2 : 1) + 1) * ((size_t)ts.nz + 1);
  const long items = (long)fp.ncol * (fp.n_op + fp.n_ts);
  const FusedKern k = ps ? fused_kernel_paired(ps) : hs ? fused_kernel_half(hs) : fused_kernel(op, fp.slots);
  if (lds > 48 * 1024 && !ensure_max_lds((const void *)k, 64 * 1024)) return false;  // (the kernel has static LDS too)
  hipLaunchKernelGGL(k, dim3((unsigned)items), dim3(OP_THREADS), lds, s, op, ts, fp);
  return true;
}

bool launch_fused_twostream_only(TwoStreamParams &ts, int slots, const int *meta_nsrc, hipStream_t s, bool half, bool paired) {
  if (ts.ng != 8 || slots < 2 || slots > 8 || (ts.nz + 63) / 64 > slots || ts.nzen > MAX_ZEN) return false;
  if (half && (paired || slots < 3 || slots > 7 || (ts.nz + 31) / 32 > slots)) return false;
  // the paired form: chunks are cut at pair boundaries, so a lane holds 2 ceil((nz/2)/64) slots at most
  if (paired && ((ts.nz & 1) || (slots & 1) || 2 * ((ts.nz / 2 + 63) / 64) > slots)) return false;
  ts.paired = paired ? 1 : 0;
  OpacityParams op;
  memset(&op, 0, sizeof(op));
  op.nz = ts.nz;
  op.col.meta = meta_nsrc;   // one int: the two-stream blocks read the column's source-layer count
  FusedParams fp;
  memset(&fp, 0, sizeof(fp));
  fp.ncol = 1;
  fp.n_op = 0;        // no opacity tiles: the two-stream blocks have nothing to wait for
  fp.n_ts = (ts.n_sol + ts.n_ir) * (half ? 1 : (ts.ng + TSW_COLS - 1) / TSW_COLS);
  fp.slots = slots;
  fp.sol_early = ts.n_sol;
  ts.col_base = 0; ts.accumulate = half ? 0 : 1;
  const size_t lds = sizeof(double) * (3 * TSW_COLS * (half ? 2 : 1) + 1) * ((size_t)ts.nz + 1);
  const FusedKern k = half ? fused_kernel_half(slots) : paired ? fused_kernel_paired(slots) : fused_kernel(op, slots);
  if (lds > 48 * 1024 && !ensure_max_lds((const void *)k, 64 * 1024)) return false;
  hipLaunchKernelGGL(k, dim3(fp.n_ts), dim3(OP_THREADS), lds, s, op, ts, fp);
  return true;
}

// ------------------------------------------------------------------------------------
// spectral integration: fup_n(i) = sum_l fup_a(i,l)*(freq(l)-freq(l+1)) (radiate.f90:184-192)
// in two deterministic stages (chunks of INT_CHUNK bins in bin order, then the chunk
// sums in order), then f_total (clima_radtran.f90:316)
// ------------------------------------------------------------------------------------
#ifndef CLIMA_INT_CHUNK
#define CLIMA_INT_CHUNK 32
#endif
constexpr int INT_CHUNK = CLIMA_INT_CHUNK;

__global__ __launch_bounds__(256) void k_integrate_partial(IntegrateParams p) {
  const int nl = p.nz + 1;
  const int a = blockIdx.y;
  const bool sol = a >= 2;
  if (sol && !p.do_solar) return;
  const double *src = a == 0 ? p.ir_fup_a : a == 1 ? p.ir_fdn_a : a == 2 ? p.sol_fup_a : p.sol_fdn_a;
  const double *freq = sol ? p.sol_freq : p.ir_freq;
  const int lo = sol ? p.sol_lo : p.ir_lo, cnt = sol ? p.sol_n : p.ir_n;
  const int l0 = lo + blockIdx.x * INT_CHUNK;
  const int l1 = min(lo + cnt, l0 + INT_CHUNK);
  for (int i = threadIdx.x; i < nl; i += blockDim.x) {
    double acc = 0.0;
    if (l0 < l1) {
      double v[INT_CHUNK];
#pragma unroll
      for (int k = 0; k < INT_CHUNK; k++) v[k] = (l0 + k < l1) ? src[(size_t)(l0 + k) * nl + i] : 0.0;
#pragma unroll
      for (int k = 0; k < INT_CHUNK; k++)
        if (l0 + k < l1) acc = __builtin_fma(v[k], freq[l0 + k] - freq[l0 + k + 1], acc);
    }
    p.partial[((size_t)a * p.nchunk + blockIdx.x) * nl + i] = acc;
  }
}

__global__ __launch_bounds__(1024) void k_integrate_final(IntegrateParams p) {
  const int nl = p.nz + 1;
  if (p.timeout_out && threadIdx.x == 0) {   // see k_integrate_one
    const int t = *p.timeout_flag;
    *p.timeout_out = (t == p.id_opr ? 1.0 : 0.0) + (t == p.id_sol ? 1024.0 : 0.0);
  }
  for (int t = threadIdx.x; t < 4 * nl; t += blockDim.x) {
    const int a = t / nl, i = t - a * nl;
    if (a >= 2 && !p.do_solar) {
      if (p.flux_part) p.flux_n[a * nl + i] = p.flux_part[a * nl + i];  // see k_integrate_one
      continue;
    }
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < p.nchunk; k++) acc = acc + p.partial[((size_t)a * p.nchunk + k) * nl + i];
    p.flux_n[a * nl + i] = acc;
    if (p.flux_part) p.flux_part[a * nl + i] = acc;
  }
  __syncthreads();
  if (p.f_total)
    for (int i = threadIdx.x; i < nl; i += blockDim.x)
      p.f_total[i] = (p.flux_n[3 * nl + i] - p.flux_n[2 * nl + i]) + (p.flux_n[1 * nl + i] - p.flux_n[0 * nl + i]);
}

// Batched IR form (shared opacity, ncol temperature columns): the same two deterministic
// stages per column, then f_total with the solar level fluxes of the handle's last solar call.
__global__ __launch_bounds__(256) void k_integrate_partial_b(BatchIntegrateParams p) {
  const int nl = p.nz + 1;
  const int a = blockIdx.y, col = blockIdx.z;
  const double *src = (a == 0 ? p.fup_a : p.fdn_a) + (size_t)col * p.spec_stride;
  const int l0 = p.ir_lo + blockIdx.x * INT_CHUNK;
  const int l1 = min(p.ir_lo + p.ir_n, l0 + INT_CHUNK);
  // (the chunk's loads go out together: one after the other, each waiting for the one before, a one-column batch -- the
  // response form's base profile -- spent 18 us here)
  if (l1 <= l0) {     // a rank without IR bins (a bin shard of solar bins only): its one chunk is empty
    for (int i = threadIdx.x; i < nl; i += blockDim.x) p.partial[(((size_t)col * 2 + a) * p.nchunk + blockIdx.x) * nl + i] = 0.0;
    return;
  }
  for (int i = threadIdx.x; i < nl; i += blockDim.x) {
    double v[INT_CHUNK], df[INT_CHUNK];
#pragma unroll
    for (int k = 0; k < INT_CHUNK; k++) {
      const int l = min(l0 + k, l1 - 1);
      v[k] = src[(size_t)l * nl + i];
      df[k] = p.freq[l] - p.freq[l + 1];
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < INT_CHUNK; k++)
      if (l0 + k < l1) acc = __builtin_fma(v[k], df[k], acc);
    p.partial[(((size_t)col * 2 + a) * p.nchunk + blockIdx.x) * nl + i] = acc;
  }
}

__global__ __launch_bounds__(256) void k_integrate_final_b(BatchIntegrateParams p) {
  const int nl = p.nz + 1;
  const int col = blockIdx.x;
  for (int i = threadIdx.x; i < nl; i += blockDim.x) {
    double up = 0.0, dn = 0.0;
    for (int k0 = 0; k0 < p.nchunk; k0 += 8) {      // (eight chunk sums of each array in flight, added in chunk order)
      double u[8], d[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int kk = min(k0 + k, p.nchunk - 1);
        u[k] = p.partial[(((size_t)col * 2 + 0) * p.nchunk + kk) * nl + i];
        d[k] = p.partial[(((size_t)col * 2 + 1) * p.nchunk + kk) * nl + i];
      }
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (k0 + k < p.nchunk) { up = up + u[k]; dn = dn + d[k]; }
    }
    double *o = p.out + (size_t)(p.col0 + col) * nl;
    o[i] = up;
    o[p.out_arr + i] = dn;
    o[2 * p.out_arr + i] = (p.flux_n[3 * nl + i] - p.flux_n[2 * nl + i]) + (dn - up);  // clima_radtran.f90:287
  }
}

void launch_integrate_batch(const BatchIntegrateParams &p, int ncol, hipStream_t s) {
  hipLaunchKernelGGL(k_integrate_partial_b, dim3(p.nchunk, 2, ncol), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_integrate_final_b, dim3(ncol), dim3(256), 0, s, p);
}

int integrate_chunks(int nbins) { return nbins <= 0 ? 1 : (nbins + INT_CHUNK - 1) / INT_CHUNK; }

// Both stages in one launch: block (level group, array) owns INT_LV = 16 consecutive levels
// (one 128-byte line per bin) of ONE of the four arrays; thread (level, chunk group) forms the
// chunk sums exactly as k_integrate_partial does, with the frequency widths staged in LDS once
// per block; the chunk sums meet in LDS and are added in chunk order.  Same association as the
// two-launch form.  f_total is NOT formed here (its four operands sit in four blocks): the host
// forms it from the four rows it fetches anyway (fetch_small), with the same expression.
// The earlier form (4 levels x 4 arrays per block) spent its time in the L1's tag lookups:
// every wave load touched 16 lines for 32 bytes each, and the widths were re-read per lane.
#ifndef CLIMA_INT_LV
#define CLIMA_INT_LV 16
#endif
#ifndef CLIMA_INT_CG
#define CLIMA_INT_CG 32
#endif
constexpr int INT_LV = CLIMA_INT_LV;   // levels per block: 13 x 4 blocks at nz = 200
constexpr int INT_CG = CLIMA_INT_CG;   // chunk groups: threads = INT_LV * INT_CG = 512

__global__ __launch_bounds__(INT_LV * INT_CG) void k_integrate_one(IntegrateParams p) {
  extern __shared__ __align__(16) double s_int[];  // widths [nchunk*INT_CHUNK], then partial [nchunk][INT_LV]
  const int a = blockIdx.y;
  const bool sol = a >= 2;
  const int nl = p.nz + 1;
  const int lv = threadIdx.x % INT_LV, cg = threadIdx.x / INT_LV;
  const int i = blockIdx.x * INT_LV + lv;
  if (p.timeout_out && a == 0 && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    const int t = *p.timeout_flag;   // the fused grid of this call has drained: the word is final
    *p.timeout_out = (t == p.id_opr ? 1.0 : 0.0) + (t == p.id_sol ? 1024.0 : 0.0);
  }
  if (p.host_out && a == 0 && blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x < 2)   // the error words ride along
    reinterpret_cast<int *>(p.host_out + 5 * nl)[threadIdx.x] = p.err_words[threadIdx.x];
  if (sol && !p.do_solar) {
    // solar rows keep the last solar call's values (clima_radtran.f90:286-289).  On a bin-sharded
    // handle flux_n is the all-reduce buffer and holds REDUCED rows by now: this rank's partial
    // solar rows are put back from flux_part, or the next reduce would count them `world` times
    if (p.flux_part && cg == 0 && i < nl) p.flux_n[a * nl + i] = p.flux_part[a * nl + i];
    if (p.host_out && cg == 0 && i < nl) p.host_out[a * nl + i] = p.flux_n[a * nl + i];
    return;
  }
  const int cb = blockIdx.z;  // column of a batch
  const double *src = (a == 0 ? p.ir_fup_a : a == 1 ? p.ir_fdn_a : a == 2 ? p.sol_fup_a : p.sol_fdn_a) + (size_t)cb * p.bs.res;
  double *const flux_n = p.flux_n + (size_t)cb * p.bs.flux;
  const double *freq = sol ? p.sol_freq : p.ir_freq;
  const int lo = sol ? p.sol_lo : p.ir_lo, cnt = sol ? p.sol_n : p.ir_n;
  double *s_df = s_int;
  double *s_part = s_int + (size_t)p.nchunk * INT_CHUNK;
  // first chunk's loads go out before the widths are staged
  double v[INT_CHUNK];
  {
    const int l0 = lo + cg * INT_CHUNK;
#pragma unroll
    for (int k = 0; k < INT_CHUNK; k++)
      v[k] = (cg < p.nchunk && i < nl && l0 + k < lo + cnt) ? src[(size_t)(l0 + k) * nl + i] : 0.0;
  }
  for (int k = threadIdx.x; k < p.nchunk * INT_CHUNK; k += blockDim.x)
    s_df[k] = k < cnt ? freq[lo + k] - freq[lo + k + 1] : 0.0;
  __syncthreads();
  for (int ch = cg; ch < p.nchunk; ch += INT_CG) {
    const int l0 = lo + ch * INT_CHUNK;
    if (ch != cg) {
#pragma unroll
      for (int k = 0; k < INT_CHUNK; k++)
        v[k] = (i < nl && l0 + k < lo + cnt) ? src[(size_t)(l0 + k) * nl + i] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < INT_CHUNK; k++)
      if (l0 + k < lo + cnt) acc = __builtin_fma(v[k], s_df[ch * INT_CHUNK + k], acc);
    s_part[ch * INT_LV + lv] = acc;
  }
  __syncthreads();
  if (cg == 0 && i < nl) {
    double acc = 0.0;
    for (int k = 0; k < p.nchunk; k++) acc = acc + s_part[k * INT_LV + lv];
    flux_n[a * nl + i] = acc;
    if (p.flux_part) p.flux_part[a * nl + i] = acc;
    if (p.host_out) p.host_out[a * nl + i] = acc;
  }
}

// true when launch_integrate() takes the one-launch kernel (the form that can store into the host's block)
bool integrate_one_launch(const IntegrateParams &p) {
  return sizeof(double) * (size_t)p.nchunk * (INT_CHUNK + INT_LV) <= 64 * 1024;
}

void launch_integrate(const IntegrateParams &p, hipStream_t s) {
  const int nl = p.nz + 1;
  const size_t lds = sizeof(double) * (size_t)p.nchunk * (INT_CHUNK + INT_LV);
  if (lds <= 64 * 1024) {
    hipLaunchKernelGGL(k_integrate_one, dim3((nl + INT_LV - 1) / INT_LV, 4, p.ncol > 0 ? p.ncol : 1), dim3(INT_LV * INT_CG), lds, s, p);
    return;
  }
  hipLaunchKernelGGL(k_integrate_partial, dim3(p.nchunk, 4), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_integrate_final, dim3(1), dim3(1024), 0, s, p);
}

// w0 of every (bin, g-point, layer) from the layers' scattering optical depth, for calls whose fused grid left
// the array unwritten (OpacityParams::write_w0 = 0): the expression of the opacity tile's store
// (types.f90:869-875)
__global__ __launch_bounds__(256) void k_w0_from_scat(const double *tau, const double *scat, double *w0, int nw, int ng, int nz) {
  const size_t n = (size_t)nw * ng * nz;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t l = i / ((size_t)ng * nz), j = i % nz;
    const double t = tau[i];
    w0[i] = t <= TAU_MIN ? 0.0 : fmin(MAX_W0, scat[l * nz + j] / t);
  }
}
void launch_w0_from_scat(const double *tau, const double *scat, double *w0, int nw, int ng, int nz, hipStream_t s) {
  hipLaunchKernelGGL(k_w0_from_scat, dim3(2048), dim3(256), 0, s, tau, scat, w0, nw, ng, nz);
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
__global__ void k_test_exp(const double *x, double *y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = fast_exp(x[i]);
}
__global__ void k_test_rcp(const double *x, double *y, int n) {
  // y[0..n): raw v_rcp_f64; [n..2n): one Newton step; [2n..3n): two (rcp_nr); [3n..4n): sqrt_nr(|x|)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  y[i] = r;
  double e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  y[n + i] = r;
  y[2 * n + i] = rcp_nr(v);
  y[3 * n + i] = sqrt_nr(fabs(v));
}
void launch_test_rcp(const double *x, double *y, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_test_rcp, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n);
}
void launch_test_exp(const double *x, double *y, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_test_exp, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n);
}
__global__ __launch_bounds__(256) void k_test_exp_tab(const double *x, double *y, int n, int base10) {
  __shared__ double s_e2[EXP2_N];
  for (int i = threadIdx.x; i < EXP2_N; i += blockDim.x) s_e2[i] = EXP2_TAB[i];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = base10 ? ten2power_tab(x[i], s_e2) : exp_tab(x[i], s_e2);
}
void launch_test_exp_tab(const double *x, double *y, int n, int base10, hipStream_t s) {
  hipLaunchKernelGGL(k_test_exp_tab, dim3((n + 255) / 256), dim3(256), 0, s, x, y, n, base10);
}

__global__ void k_copy(double *dst, const double *src, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy(double *dst, const double *src, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_copy, dim3((unsigned)std::min<size_t>((n + 255) / 256, 64)), dim3(256), 0, s, dst, src, n);
}
void launch_scale(double *a, size_t n, double f, hipStream_t s) {
  if (n == 0) return;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_scale, dim3(grid), dim3(256), 0, s, a, n, f);
}

#include "ir_green.inc"

}  // namespace clima
