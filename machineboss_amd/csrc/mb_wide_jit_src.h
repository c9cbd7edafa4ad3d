// mb_wide_jit_src.h -- what every generated one-tape kernel starts with (mb_wide_jit.cpp appends the unrolled periods): the structures
// the launch passes, and the arithmetic of the ahead-of-time kernels of mb_wide.hip, statement for statement -- the one-exponential
// online log-sum-exp update, the lane-group butterflies over DPP / ds_bpermute, the fp64 correction term, the exchange's bounded waits.
#pragma once

namespace mb {
static const char *kWideJitPrelude = R"MBJIT(
#define NEG_INF (-__builtin_inf())
#define W_NEG_BIG (-1e300)
struct PairDesc { long long inBase, outBase; int inLen, outLen; long long cellBase; int launch0; int pad; long long envBase; };
struct WideDev { const void *segA, *segB; long long strideA; int nA, nB; int S, NV, NX, W; int resultIdx; int backward; int inputTape; int lastOnly; };
struct WideJitArgs { const unsigned *tab[16]; const unsigned *impIdx[16]; const unsigned *stream[16]; int nSeq, nExpTot; double *X; const long long *xOff; unsigned *err; long long timeoutTicks; };

typedef __attribute__((address_space(3))) double j_lds_f64;
typedef __attribute__((address_space(3))) int j_lds_i32;
__device__ __forceinline__ double lds_rd(unsigned a) { return *(const j_lds_f64 *)(unsigned long long)a; }
__device__ __forceinline__ void lds_wr(unsigned a, double v) { *(j_lds_f64 *)(unsigned long long)a = v; }
__device__ __forceinline__ int lds_rdi(unsigned a) { return *(const j_lds_i32 *)(unsigned long long)a; }
__device__ __forceinline__ void lds_wri(unsigned a, int v) { *(j_lds_i32 *)(unsigned long long)a = v; }
__device__ __forceinline__ double ldw(const unsigned *__restrict__ p, int stride) { return __hiloint2double((int)p[stride], (int)p[0]); }

// v_max_f64 as is: operands are sums of finite weights or -inf, never NaN
__device__ __forceinline__ double jmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// exp(x), x <= 0, through v_exp_f32: the multiplication by log2(e) in fp64 BEFORE the conversion (mb_wide.hip: wide_exp_diff)
__device__ __forceinline__ float jexp(double x) { return __builtin_amdgcn_exp2f((float)(x * 1.4426950408889634)); }
__device__ __forceinline__ void jfold(double &m, float &s, double v) {
  const float e = jexp(-fabs(v - m));
  const bool up = v > m;
  s = __fmaf_rn(up ? s : 1.0f, e, up ? 1.0f : s);
  m = jmax(m, v);
}
template <int H>
__device__ __forceinline__ int jxor(int v) {
  // (bound_ctrl with full row / bank masks: every lane is written, so the compiler needs no copy of the old value in front of the DPP move)
  if (H == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);       // quad_perm:[1,0,3,2]
  if (H == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);       // quad_perm:[2,3,0,1]
  if (H == 4) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);      // row_half_mirror
  if (H == 8) return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);      // row_mirror
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  return __builtin_amdgcn_ds_bpermute((lane ^ H) << 2, v);
}
template <int H> __device__ __forceinline__ double jxord(double m) { return __hiloint2double(jxor<H>(__double2hiint(m)), jxor<H>(__double2loint(m))); }
// every lane group of the wavefront has G lanes: no masks
template <int H, int G> __device__ __forceinline__ void jmax_all(double &m) { if (G > H) m = jmax(m, jxord<H>(m)); }
template <int H, int G> __device__ __forceinline__ void jsum_all(float &s) { if (G > H) s += __int_as_float(jxor<H>(__float_as_int(s))); }
template <int H, int G> __device__ __forceinline__ void jsum_all(double &s) { if (G > H) s += jxord<H>(s); }
template <int G> __device__ __forceinline__ void jreduce_max_all(double &m) {
  jmax_all<1, G>(m); jmax_all<2, G>(m); jmax_all<4, G>(m); jmax_all<8, G>(m); jmax_all<16, G>(m); jmax_all<32, G>(m);
}
template <int G> __device__ __forceinline__ void jreduce_sum_all(double &m, float &s) {
  const double own = m;
  jreduce_max_all<G>(m);
  s *= jexp(own - m);
  jsum_all<1, G>(s); jsum_all<2, G>(s); jsum_all<4, G>(s); jsum_all<8, G>(s); jsum_all<16, G>(s); jsum_all<32, G>(s);
}
// groups of g lanes (per lane) in a wavefront whose largest group has GW: the steps beyond a lane's own group are masked
template <int H, int GW> __device__ __forceinline__ void jmax_msk(double &m, int g) { if (GW > H) { const double mx = jmax(m, jxord<H>(m)); m = (H < g) ? mx : m; } }
template <int H, int GW> __device__ __forceinline__ void jsum_msk(float &s, int g) { if (GW > H) { const float so = __int_as_float(jxor<H>(__float_as_int(s))); if (H < g) s += so; } }
template <int H, int GW> __device__ __forceinline__ void jsum_msk(double &s, int g) { if (GW > H) { const double so = jxord<H>(s); if (H < g) s += so; } }
template <int GW> __device__ __forceinline__ void jreduce_max_msk(double &m, int g) {
  jmax_msk<1, GW>(m, g); jmax_msk<2, GW>(m, g); jmax_msk<4, GW>(m, g); jmax_msk<8, GW>(m, g); jmax_msk<16, GW>(m, g); jmax_msk<32, GW>(m, g);
}
template <int GW> __device__ __forceinline__ void jreduce_sum_msk(double &m, float &s, int g) {
  const double own = m;
  jreduce_max_msk<GW>(m, g);
  s *= jexp(own - m);
  jsum_msk<1, GW>(s, g); jsum_msk<2, GW>(s, g); jsum_msk<4, GW>(s, g); jsum_msk<8, GW>(s, g); jsum_msk<16, GW>(s, g); jsum_msk<32, GW>(s, g);
}
// the two-pass log-sum-exp of a generated round: group maximum first (the ladders above), then every lane sums exp(candidate - maximum) over
// its own candidates and the group sums the lanes -- no running rescale per candidate, no rescale between the two ladders
template <int G> __device__ __forceinline__ void jreduce_fsum_all(float &s) { jsum_all<1, G>(s); jsum_all<2, G>(s); jsum_all<4, G>(s); jsum_all<8, G>(s); jsum_all<16, G>(s); jsum_all<32, G>(s); }
template <int G> __device__ __forceinline__ void jreduce_fsum_all(double &s) { jsum_all<1, G>(s); jsum_all<2, G>(s); jsum_all<4, G>(s); jsum_all<8, G>(s); jsum_all<16, G>(s); jsum_all<32, G>(s); }
template <int GW> __device__ __forceinline__ void jreduce_fsum_msk(float &s, int g) { jsum_msk<1, GW>(s, g); jsum_msk<2, GW>(s, g); jsum_msk<4, GW>(s, g); jsum_msk<8, GW>(s, g); jsum_msk<16, GW>(s, g); jsum_msk<32, GW>(s, g); }
template <int GW> __device__ __forceinline__ void jreduce_fsum_msk(double &s, int g) { jsum_msk<1, GW>(s, g); jsum_msk<2, GW>(s, g); jsum_msk<4, GW>(s, g); jsum_msk<8, GW>(s, g); jsum_msk<16, GW>(s, g); jsum_msk<32, GW>(s, g); }
// (value, place): the maximum, among equal maxima the SMALLEST place -- std::max_element's first maximum (src/dpmatrix.defs.h:171-174)
template <int H, int GW> __device__ __forceinline__ void jkey_msk(unsigned &key, int g) { if (GW > H) { const unsigned ko = (unsigned)jxor<H>((int)key); if (H < g) key = min(key, ko); } }
template <int GW> __device__ __forceinline__ void jreduce_tb(double &m, unsigned &key, int g) {
  const double own = m;
  jreduce_max_msk<GW>(m, g);
  key = (own == m) ? key : 0xFFFFFFFFu;
  jkey_msk<1, GW>(key, g); jkey_msk<2, GW>(key, g); jkey_msk<4, GW>(key, g); jkey_msk<8, GW>(key, g); jkey_msk<16, GW>(key, g); jkey_msk<32, GW>(key, g);
}

// ---- the log-sum-exp correction term in fp64 (E-steps over long sequences; mb_wide.hip: wide_exp64 / wide_log64) ----
__device__ __forceinline__ double jexp64(double x, unsigned tab) {
  x = jmax(x, -740.0);
  const double kf = __builtin_rint(x * 92.33248261689366);
  const double r = __builtin_fma(kf, -0.010830424696249145, x);
  const int k = (int)kf;
  const double p = __builtin_fma(__builtin_fma(__builtin_fma(__builtin_fma(r, 0.041666666666666664, 0.16666666666666666), r, 0.5), r, 1.0), r, 1.0);
  return __builtin_ldexp(lds_rd(tab + ((unsigned)(k & 63) << 3)) * p, k >> 6);
}
__device__ __forceinline__ double jlog64(double s, unsigned tab) {
  const double t0 = (double)__log2f((float)s) * 0.6931471805599453;
  const double d = __builtin_fma(s, jexp64(-t0, tab), -1.0);
  return t0 + __builtin_fma(-0.5 * d, d, d);
}
__device__ __forceinline__ void jfold64(double &m, double &s, double v, unsigned tab) {
  const double e = jexp64(-fabs(v - m), tab);
  const bool up = v > m;
  s = __builtin_fma(up ? s : 1.0, e, up ? 1.0 : s);
  m = jmax(m, v);
}
template <int G> __device__ __forceinline__ void jreduce_sum64_all(double &m, double &s, unsigned tab) {
  const double own = m;
  jreduce_max_all<G>(m);
  s *= jexp64(own - m, tab);
  jsum_all<1, G>(s); jsum_all<2, G>(s); jsum_all<4, G>(s); jsum_all<8, G>(s); jsum_all<16, G>(s); jsum_all<32, G>(s);
}
template <int GW> __device__ __forceinline__ void jreduce_sum64_msk(double &m, double &s, int g, unsigned tab) {
  const double own = m;
  jreduce_max_msk<GW>(m, g);
  s *= jexp64(own - m, tab);
  jsum_msk<1, GW>(s, g); jsum_msk<2, GW>(s, g); jsum_msk<4, GW>(s, g); jsum_msk<8, GW>(s, g); jsum_msk<16, GW>(s, g); jsum_msk<32, GW>(s, g);
}

// ---- the exchange between the parts of a machine (mb_wide.h: WidePartDev): no flags, a sentinel, bounded waits ----
#define X_EMPTY (~0ull)
__device__ __forceinline__ unsigned long long x_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double x_wait(const unsigned long long *p, unsigned *err, long long timeoutTicks) {
  const long long t0 = (long long)wall_clock64();
  for (;;) {
    const unsigned long long b = x_load(p);
    if (b != X_EMPTY) return __longlong_as_double((long long)b);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return NEG_INF;
    if ((long long)wall_clock64() - t0 > timeoutTicks) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return NEG_INF; }
    __builtin_amdgcn_s_sleep(4);
  }
}
__device__ __forceinline__ void x_store(unsigned long long *p, double v) {
  const unsigned long long bits = v == v ? (unsigned long long)__double_as_longlong(v) : 0x7ff8000000000000ull;      // (a NaN would read as "not yet": none is stored as all-ones)
  __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
)MBJIT";
}  // namespace mb
