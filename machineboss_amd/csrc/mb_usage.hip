// mb_usage.hip -- BackwardMatrix::getCounts as a THIRD PASS over two materialised matrices (src/backward.cpp:58-87 IS a third sweep:
// Forward filled, Backward filled, then every cell's outgoing transitions weighed).
//
// The tiled family's default E-step fuses that sweep into a second Forward fill that keeps no matrix (mb_medium.hip, count programs:
// 16 B per lattice cell, nothing re-read).  For machines of a few hundred states the fused sweep is bound by its step's dependent chain,
// not by memory (the 482-state composition of BASELINE config 4: 0.19 of the HBM roofline at 16 B), while the plain fills
// of the same machine run at 0.4.  The usage sums themselves have NO dependency between cells:
//     count[e] += exp( (F(i, o, src) - LL) + (B(i + di, o + do, dst) + w) )        over the cells whose tokens the transition's labels match
// so with both matrices in HBM they are a streaming pass: one workgroup per INPUT COLUMN i walks the rows o = 0 .. outLen, the state
// vectors F(i, o, .), B(i, o .. o + 1, .), B(i + 1, o .. o + 1, .) in LDS (each S contiguous doubles in the reference's layout,
// src/dpmatrix.h:90-96), ONE LANE PER TRANSITION that can apply in this column (silent, output-only, and the input-consuming ones of the
// column's own token), a lane's sums in registers, one fp64 atomic per transition and column at the end.  Real traffic: both matrices
// written once, Forward read once, Backward read once from HBM and once more by the neighbouring column (L2 when the two columns run
// side by side on one XCD -- the unit order below sees to it): ~32-40 B per lattice cell against 16 for the fused sweep; what it buys is
// TIME on machines whose fused sweep is latency-bound.  Chosen per machine by mb_api.hip (MB_MEDIUM_COUNT_PASSES: 0 auto, 2 fused, 3 this).
#include <algorithm>
#include <cstring>
#include <vector>

#include "mb_internal.h"
#include "mb_usage.h"

namespace mb {

// word: src | dst << 15 | di << 30 | valid << 31      meta: transition (global edge id) | output token << 24
struct alignas(16) UsageRec { double w; uint32_t word; uint32_t meta; };

// (workgroups of <= 512 lanes with several transitions per lane: 31 KB of LDS for the 482-state machine, so FOUR to five workgroups share a
//  CU and their row loads overlap -- one workgroup of 832 lanes per CU waited for its own loads, 1.46 us per row where the CU's share of
//  the HBM bandwidth allows 0.66)
template <int NSLOT>
__global__ __launch_bounds__(512) void k_medium_usage(int S, int W, const UsageRec *__restrict__ rec, const PairDesc *__restrict__ pairs, const int2 *__restrict__ units,
                                                       long long nUnits, const int *__restrict__ inTok, const int *__restrict__ outTok,
                                                       const double *__restrict__ fwdPool, const double *__restrict__ bwdPool, double *__restrict__ counts, int det) {
  extern __shared__ double ul[];
  // units are dealt so that the workgroups an XCD runs together are ADJACENT columns of one pair (blockIdx -> XCD goes round-robin):
  // column i + 1's Backward rows, which this column reads as its right neighbour's, are then in that XCD's L2
  const long long perX = (nUnits + 7) / 8;
  const long long u = (long long)(blockIdx.x & 7u) * perX + (blockIdx.x >> 3);
  if (u >= nUnits) return;
  const int2 un = units[u];
  const PairDesc pd = pairs[un.x];
  const int i = un.y, inLen = pd.inLen, outLen = pd.outLen, tid = threadIdx.x;
  const long long I = inLen + 1;
  const double *F = fwdPool + pd.cellBase + (long long)i * S, *B = bwdPool + pd.cellBase + (long long)i * S;      // row o: + o * I * S
  const double ll = bwdPool[pd.cellBase];      // backward.logLike() (src/backward.cpp:66)
  if (!(ll > -INFINITY)) return;
  const int a = i < inLen ? inTok[pd.inBase + i] : 0;
  const int *out = outTok + pd.outBase;
  double w[NSLOT], acc[NSLOT];
  uint32_t word[NSLOT], meta[NSLOT];
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) {
    const UsageRec r = rec[((size_t)a * NSLOT + k) * W + tid];
    w[k] = r.w; word[k] = r.word; meta[k] = r.meta; acc[k] = 0.0;
  }
  // LDS: Forward row [2][S]; Backward rows [3][2][S] (a ring of three rows, this column and the next)
  double *Fv = ul, *Bv = ul + 2 * (size_t)S;
  const bool right = i < inLen;
  const long long rowStride = I * S;
  auto stage = [&](int o) {      // row o of F, row o + 1 of B (both columns) -- row 0 of B is staged by the prologue
    for (int k = tid; k < S; k += W) {
      Fv[(o & 1) * S + k] = F[(long long)o * rowStride + k];
      if (o + 1 <= outLen) {
        const int slot = (o + 1) % 3;
        Bv[(slot * 2) * S + k] = B[(long long)(o + 1) * rowStride + k];
        if (right) Bv[(slot * 2 + 1) * S + k] = B[(long long)(o + 1) * rowStride + S + k];
      }
    }
  };
  for (int k = tid; k < S; k += W) { Bv[k] = B[k]; if (right) Bv[S + k] = B[S + k]; }
  stage(0);
  __syncthreads();
  for (int o = 0; o <= outLen; ++o) {
    // the next row's vectors are requested before this row's terms are summed (registers), and stored behind them
    const int on = o + 1;
    double nf[4], nb0[4], nb1[4];
    const bool more = on <= outLen, moreB = on + 1 <= outLen;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = tid + q * W;
      nf[q] = (more && k < S) ? F[(long long)on * rowStride + k] : 0.0;
      nb0[q] = (moreB && k < S) ? B[(long long)(on + 1) * rowStride + k] : 0.0;
      nb1[q] = (moreB && right && k < S) ? B[(long long)(on + 1) * rowStride + S + k] : 0.0;
    }
    const int b = o < outLen ? out[o] : 0;
    const double *f = Fv + (o & 1) * S;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
      const uint32_t ot = meta[k] >> 24;
      if ((word[k] >> 31) && (ot == 0u || (int)ot == b)) {
        const int src = (int)(word[k] & 0x7fffu), dst = (int)((word[k] >> 15) & 0x7fffu), di = (int)((word[k] >> 30) & 1u);
        const int slot = (o + (ot ? 1 : 0)) % 3;
        const double logOdds = f[src] - ll, tll = Bv[(slot * 2 + di) * S + dst] + w[k];
        acc[k] += (double)__builtin_amdgcn_exp2f((float)((logOdds + tll) * 1.4426950408889634));
      }
    }
    if (more) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = tid + q * W;
        if (k < S) {
          Fv[(on & 1) * S + k] = nf[q];
          if (moreB) { const int slot = (on + 1) % 3; Bv[(slot * 2) * S + k] = nb0[q]; if (right) Bv[(slot * 2 + 1) * S + k] = nb1[q]; }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NSLOT; ++k)
    if ((word[k] >> 31) && acc[k] != 0.0) {
      const uint32_t e = meta[k] & 0xffffffu;
      if (det) atomicAdd((unsigned long long *)counts + e, (unsigned long long)fmin(fmax(acc[k] * MB_DET_GLOBAL_SCALE + 0.5, 0.0), 4611686018427387904.0));
      else atomicAdd(&counts[e], acc[k]);
    }
}

bool usage_build(const mb_machine *m, UsagePlan &U) {
  usage_free(U);
  U.tried = true;
  const int S = m->S, nIn = m->nIn;
  if (S >= (1 << 15) || m->nTrans >= (1 << 24) || m->nOut >= 255 || nIn < 1 || m->nOut < 1) return false;
  // per column token a (0: the last column, nothing is consumed): silent and input-consuming transitions first (always active), then
  // the output-consuming ones grouped by token, so that a wavefront's lanes mostly share a token and skip a row together
  std::vector<std::vector<long long>> lists(nIn + 1);
  size_t longest = 0;
  for (int a = 0; a <= nIn; ++a) {
    for (int pass = 0; pass <= m->nOut; ++pass)
      for (long long e = 0; e < m->nTrans; ++e) {
        if (m->logW[e] == -INFINITY) continue;
        if (m->inTok[e] != 0 && (int)m->inTok[e] != a) continue;
        if ((int)m->outTok[e] != pass) continue;
        lists[a].push_back(e);
      }
    longest = std::max(longest, lists[a].size());
  }
  if (longest == 0) return false;
  // lanes: about 448 with up to eight transitions each (several workgroups per CU), more lanes for machines whose state vectors need them
  int nSlot = std::min(8, std::max(1, (int)((longest + 447) / 448)));
  int W = std::max(64, (int)(((longest + nSlot - 1) / nSlot + 63) / 64 * 64));
  while ((S + W - 1) / W > 4 && W < 512) W += 64;      // (a lane stages at most four entries of a state vector per row)
  if (W > 512 || (S + W - 1) / W > 4 || (size_t)W * nSlot < longest) return false;
  U.nSlot = nSlot; U.W = W; U.S = S;
  U.ldsBytes = (size_t)8 * S * sizeof(double);
  if (U.ldsBytes > 160 * 1024) return false;
  U.h_rec.assign((size_t)(nIn + 1) * nSlot * W * 4, 0u);      // 16 bytes = 4 words per record
  for (int a = 0; a <= nIn; ++a)
    for (size_t r = 0; r < lists[a].size(); ++r) {
      const long long e = lists[a][r];
      const size_t k = r / W, l = r % W, at = (((size_t)a * nSlot + k) * W + l) * 4;
      const double w = m->logW[e];
      std::memcpy(&U.h_rec[at], &w, 8);
      U.h_rec[at + 2] = (uint32_t)m->src[e] | ((uint32_t)m->dst[e] << 15) | (m->inTok[e] ? 1u << 30 : 0u) | (1u << 31);
      U.h_rec[at + 3] = (uint32_t)e | ((uint32_t)m->outTok[e] << 24);
    }
  if (hipMalloc(&U.d_rec, U.h_rec.size() * 4) != hipSuccess || hipMemcpy(U.d_rec, U.h_rec.data(), U.h_rec.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    usage_free(U); U.tried = true;
    return false;
  }
  U.ok = true;
  return true;
}

void usage_free(UsagePlan &U) {
  if (U.d_rec) (void)hipFree(U.d_rec);
  U = UsagePlan();
}

int usage_launch(const mb_machine *m, const UsagePlan &U, const PairDesc *d_pairs, const std::vector<PairDesc> &hp, const int *d_in, const int *d_out,
                 const double *fwd, const double *bwd, double *d_counts, hipStream_t st) {
  if (!U.ok) { set_error("usage pass: no plan"); return 1; }
  std::vector<int2> units;
  for (size_t p = 0; p < hp.size(); ++p) for (int i = 0; i <= hp[p].inLen; ++i) units.push_back(make_int2((int)p, i));
  if (units.empty()) return 0;
  int2 *d_units = nullptr;
  MB_HIP(sm_alloc((void **)&d_units, units.size() * sizeof(int2)));
  if (h2d_large(d_units, units.data(), units.size() * sizeof(int2)) || !hip_ok(hipStreamSynchronize(st), "usage units")) { sm_free(d_units); return 1; }
  const long long nUnits = (long long)units.size();
  const unsigned grid = (unsigned)(((nUnits + 7) / 8) * 8);
  const void *rec = U.d_rec;
#define MB_USAGE_GO(N) do { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute((const void *)k_medium_usage<N>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL((k_medium_usage<N>), dim3(grid), dim3((unsigned)U.W), U.ldsBytes, st, U.S, U.W, (const UsageRec *)rec, d_pairs, (const int2 *)d_units, nUnits, d_in, d_out, fwd, bwd, d_counts, g_deterministic ? 1 : 0); } while (0)
  switch (U.nSlot) { case 1: MB_USAGE_GO(1); break; case 2: MB_USAGE_GO(2); break; case 3: MB_USAGE_GO(3); break; case 4: MB_USAGE_GO(4); break; case 5: MB_USAGE_GO(5); break; case 6: MB_USAGE_GO(6); break; case 7: MB_USAGE_GO(7); break; default: MB_USAGE_GO(8); break; }
#undef MB_USAGE_GO
  const bool ok = hip_ok(hipGetLastError(), "usage pass launch");
  // (the unit list must outlive the kernel: released once the stream has passed it)
  if (!hip_ok(hipStreamSynchronize(st), "usage pass")) { sm_free(d_units); return 1; }
  sm_free(d_units);
  return ok ? 0 : 1;
}

}  // namespace mb
