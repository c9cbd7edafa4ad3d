// mb_device_math.h -- log-space reductions used by the kernels.
//
// Reference semantics: sum_reduce = log_sum_exp(a,b) = max(a,b) + f(|a-b|), f(x) = log(1+exp(-x)), with
// (a == b) handled explicitly so that (-inf,-inf) -> -inf (src/logsumexp.h:72-90, src/dpmatrix.h:121).
// The reference evaluates f through a 100 001-entry interpolated table that returns 0 for x >= 10
// (src/logsumexp.h:20-21,48-70); the device evaluates f directly, which SURVEY.md section 6 measured to move
// log-likelihoods by <= 3e-8 relative -- four orders inside the 1e-4 parity tolerance.
// max_reduce = std::max (src/dpmatrix.h:122) is exact, so Viterbi cells are bit-identical to the CPU's.
#pragma once
#include <hip/hip_runtime.h>

namespace mb {

__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }  // std::max(a,b)

// exact double-precision variant (generic kernels, mb_fill): max + log1p(exp(-diff))
__device__ __forceinline__ double lse2_exact(double a, double b) {
  double mx, df;
  if (a == b) { mx = a; df = 0.0; }
  else if (a < b) { mx = b; df = b - a; }
  else { mx = a; df = a - b; }
  return mx + log1p(exp(-df));
}

// fast variant (tiled kernels): the correction term f(diff) in (0, log 2] is evaluated in fp32 with the
// hardware v_exp_f32 / v_log_f32 (abs. error ~1e-7) and added back in fp64; max and cells stay fp64.
__device__ __forceinline__ double lse2_fast(double a, double b) {
  const double mx = dmax(a, b);
  const double mn = (a < b) ? a : b;
  // diff = mx - mn >= 0; (-inf,-inf): mx - mn = NaN -> guard through the a == b test of the reference
  const float df = (a == b) ? 0.0f : (float)(mx - mn);
  const float e = __expf(-df);          // exp(-inf) = 0 covers one-sided -inf
  const float c = __logf(1.0f + e);
  return mx + (double)c;
}

}  // namespace mb
