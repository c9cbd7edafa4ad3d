// mb_usage.h -- posterior counts as a third, dependency-free pass over two materialised matrices (see mb_usage.hip).
#pragma once
#include <vector>

#include "mb_internal.h"

namespace mb {

struct UsagePlan {
  bool tried = false, ok = false;
  int nSlot = 0, W = 0, S = 0;      // transitions per lane, lanes per workgroup (one workgroup per input column), states
  size_t ldsBytes = 0;
  std::vector<uint32_t> h_rec;      // [column token 0 .. nIn][slot][lane] records of 16 bytes
  void *d_rec = nullptr;
};
// false: the machine does not fit the pass (more than 4 096 transitions applicable in a column, state vectors beyond the LDS ...)
bool usage_build(const mb_machine *m, UsagePlan &U);
void usage_free(UsagePlan &U);
// counts[e] += sum over the cells of every pair of exp(F - LL + B + w); fwd / bwd: the chunk's matrices (reference layout, PairDesc::cellBase)
int usage_launch(const mb_machine *m, const UsagePlan &U, const PairDesc *d_pairs, const std::vector<PairDesc> &hp, const int *d_in, const int *d_out,
                 const double *fwd, const double *bwd, double *d_counts, hipStream_t st);

}  // namespace mb
