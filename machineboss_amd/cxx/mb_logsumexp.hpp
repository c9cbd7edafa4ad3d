// mb_logsumexp.hpp -- the host helpers of the reference's src/logsumexp.h:72-172 under their own names, over the C-ABI
// (mb_log_sum_exp*, include/mbhip.h): what a caller that included "logsumexp.h" keeps compiling against once
// src/logsumexp.{h,cpp} are replaced.  Table semantics of the reference, bit for bit.
#pragma once
#include <limits>
#include <stdexcept>
#include <vector>
#include <cmath>

#include "mbhip.h"

namespace MachineBossHIP {

typedef double LogProb;

inline double log_sum_exp(double a, double b) { return mb_log_sum_exp(a, b); }
inline double log_sum_exp(double a, double b, double c) { return log_sum_exp(log_sum_exp(a, b), c); }
inline double log_sum_exp(double a, double b, double c, double d) { return log_sum_exp(log_sum_exp(log_sum_exp(a, b), c), d); }
inline double log_sum_exp(double a, double b, double c, double d, double e) { return log_sum_exp(log_sum_exp(log_sum_exp(log_sum_exp(a, b), c), d), e); }
inline double log_accum_exp(double &a, double b) { a = log_sum_exp(a, b); return a; }
inline double log_sum_exp(const std::vector<double> &v) { return mb_log_sum_exp_n(v.data(), v.size()); }
inline double log_sum_exp(const std::vector<std::vector<double>> &v) {
  double tot = -std::numeric_limits<double>::infinity();
  for (const auto &row : v) (void)log_accum_exp(tot, log_sum_exp(row));
  return tot;
}
inline double log_subtract_exp(double a, double b) {
  if (a < b) throw std::runtime_error("Sign error in log_subtract_exp");
  return a + std::log(1. - std::exp(b - a));
}
inline LogProb logInnerProduct(const std::vector<LogProb> &v1, const std::vector<LogProb> &v2) { return mb_log_inner_product(v1.data(), v2.data(), nullptr, v1.size()); }
inline LogProb logInnerProduct(const std::vector<LogProb> &v1, const std::vector<LogProb> &v2, const std::vector<LogProb> &v3) {
  return mb_log_inner_product(v1.data(), v2.data(), v3.data(), v1.size());
}
inline LogProb logInnerProduct(const std::vector<std::vector<LogProb>> &v1, const std::vector<std::vector<LogProb>> &v2) {
  LogProb lip = -std::numeric_limits<double>::infinity();
  for (size_t k = 0; k < v1.size(); ++k) lip = log_sum_exp(lip, logInnerProduct(v1[k], v2[k]));
  return lip;
}
inline std::vector<LogProb> log_vector(const std::vector<double> &v) { std::vector<LogProb> r; for (double x : v) r.push_back(std::log(x)); return r; }

}  // namespace MachineBossHIP
