// mb_wide.h -- "one workgroup = one column" kernel family for large one-tape machines (see mb_wide.hip).
#pragma once
#include <vector>

#include "mb_internal.h"

namespace mb {

// One candidate of one lane (16 bytes, one global_load_dwordx4): value = vector[src] + w.
//   src = sel << 30 | index   sel 0: this column's state vector, 1: this column's extra entries (emit-only parts of
//                             states that also have silent predecessors, the dummy entry), 2: the previous column's
//                             state vector (index S = -inf sentinel, S + 1 = seed).
struct alignas(16) WideRec {
  double w;
  uint32_t src;
  uint32_t pad;
};

// A round: every lane group of one workgroup finalises one node (state or emit-only part).  Lane `l` of a group of g
// lanes folds candidates l, l + g, l + 2g, ... of its node (depth slots), the group is reduced with shuffles, its first
// lane stores.  Record of (slot j, lane) = rec[recBase + tok * tokStride + j * W + lane]; tokStride = 0 when no node of
// the round has emitting candidates.  dst[dstBase + lane] = sel << 30 | log2(g) << 27 | index.
struct WideRound {
  int recBase, tokStride, depth, dstBase;
  int maxG, sync, pad0, pad1;
};

struct WideDev {
  const WideRound *rounds;
  const WideRec *recs;
  const uint32_t *dsts;
  int nRounds, S, NV, NX, W;
  int resultIdx;     // state whose value in the last column is the log-likelihood
  int backward;
};

struct WideProgram {
  bool ok = false, dirty = true;
  bool backward = false, viterbi = false;
  int stages = 0;            // closure stages the silent levels were grouped into (0 = levelled, exact)
  int W = 1024;
  long long nPairs = 0;      // closure pairs
  long long slotsPerColumn = 0;
  int nSync = 0;
  std::vector<WideRound> rounds;
  std::vector<WideRec> recs;
  std::vector<uint32_t> dsts;
  int NV = 0, NX = 0;
  WideRound *d_rounds = nullptr;
  WideRec *d_recs = nullptr;
  uint32_t *d_dsts = nullptr;
  WideDev dev{};
  size_t vecBytes() const { return (size_t)(2 * NV + NX) * sizeof(double); }
};

// which machines this family takes: one-tape generators (no input alphabet) with enough states to fill a workgroup
bool wide_applicable(const mb_machine *m);
// (re)build the program of one direction/semiring from the machine's current weights and upload it
bool wide_build(const mb_machine *m, bool backward, bool viterbi, WideProgram &P);
void wide_free(WideProgram &P);
// sweep every pair of the chunk; pool != nullptr: materialise the matrix (reference layout); loglike != nullptr: gather
// the log-likelihood of each pair.  d_desc/hp describe the same pairs (cellBase relative to pool).
int wide_fill(const mb_machine *m, WideProgram &P, const PairDesc *d_desc, long long nPairs, const int *d_out, double *pool,
              double *loglike, hipStream_t st);

}  // namespace mb
