// mb_logsumexp.cpp -- host-side log-space helpers of the reference's logsumexp.h that callers outside the DP fills keep
// using (src/logsumexp.h:72-172: log_sum_exp with 2..5 arguments and over vectors, log_accum_exp, logInnerProduct,
// log_subtract_exp), exported through the C-ABI so that replacing src/logsumexp.* leaves them available.
//
// Semantics are the reference's table build: log_sum_exp(a,b) = max + f(|a-b|), f(x) = log(1 + exp(-x)) read from a table of
// 100 001 entries at step 1e-4 on [0,10] by linear interpolation; f = 0 for x >= 10, NaN and infinities; a == b is handled
// first so that (-inf,-inf) -> -inf (src/logsumexp.h:20-21,48-90, src/logsumexp.cpp:8-18).  (The device kernels evaluate f
// directly -- DESIGN.md section 2 -- these helpers are for host code that wants the reference's numbers bit for bit.)
#include <cmath>
#include <limits>
#include <vector>

#include "mbhip.h"

namespace {
const double kMax = 10.0, kPrec = 1e-4;
const int kEntries = (int)(kMax / kPrec) + 1;

struct Table {
  std::vector<double> f;
  Table() : f((size_t)kEntries + 1, 0.0) { for (int n = 0; n < kEntries; ++n) f[(size_t)n] = std::log(1.0 + std::exp(-(n * kPrec))); }
};
const Table &table() { static const Table t; return t; }

inline double unary(double x) {
  if (x >= kMax || std::isnan(x) || std::isinf(x)) return 0.0;
  if (x < 0) return -x;     // (the reference warns and carries on)
  const std::vector<double> &f = table().f;
  const int n = (int)(x / kPrec);
  const double f0 = f[(size_t)n], f1 = f[(size_t)n + 1];
  return f0 + (f1 - f0) * ((x - n * kPrec) / kPrec);
}
}  // namespace

extern "C" {

double mb_log_sum_exp(double a, double b) {
  double mx, diff;
  if (a == b) { mx = a; diff = 0; }
  else if (a < b) { mx = b; diff = b - a; }
  else { mx = a; diff = a - b; }
  return mx + unary(diff);
}

// log_sum_exp(const vguard<double>&): left-to-right accumulation from -inf (src/logsumexp.h:109-114)
double mb_log_sum_exp_n(const double *v, size_t n) {
  double tot = -std::numeric_limits<double>::infinity();
  for (size_t k = 0; k < n; ++k) tot = mb_log_sum_exp(tot, v[k]);
  return tot;
}

// logInnerProduct(v1, v2) = log sum_k exp(v1[k] + v2[k]) (src/logsumexp.h:143-148); v3 may be NULL (:150-155)
double mb_log_inner_product(const double *v1, const double *v2, const double *v3, size_t n) {
  double lip = -std::numeric_limits<double>::infinity();
  for (size_t k = 0; k < n; ++k) lip = mb_log_sum_exp(lip, v1[k] + v2[k] + (v3 ? v3[k] : 0.0));
  return lip;
}

}  // extern "C"
