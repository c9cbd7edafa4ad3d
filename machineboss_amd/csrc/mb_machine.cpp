// mb_machine.cpp -- host-side "machine compiler": flattened EvaluatedMachine -> CSR views + silent levels.
//
// Replaces the nested map<InputToken, map<OutputToken, multimap<StateIndex,Trans>>> of the reference
// (src/eval.h:66-68, filled by EvaluatedMachine::init, src/eval.cpp:47-69) by two CSR tables whose row order
// reproduces the maps' iteration order, and derives the level schedule that lets a GPU finalise the states of
// one supercell level by level instead of strictly in state order (the reference's `for d` loop,
// src/forward.defs.h:32, relies on silent edges pointing to higher states, src/machine.cpp:758-764).
#include <algorithm>
#include <numeric>

#include "mb_internal.h"

namespace mb {

static void build_csr(const mb_machine *m, bool incoming, std::vector<int> &off, std::vector<uint32_t> &perm) {
  const long long nKeys = (long long)m->S * m->K;
  off.assign(nKeys + 1, 0);
  perm.resize(m->nTrans);
  auto rowOf = [&](long long e) {
    const long long st = incoming ? m->dst[e] : m->src[e];
    return (st * (m->nIn + 1) + m->inTok[e]) * (m->nOut + 1) + m->outTok[e];
  };
  std::vector<uint32_t> ids(m->nTrans);
  std::iota(ids.begin(), ids.end(), 0u);
  // (row, other endpoint, global id): global id ascending == (src, transIndex) ascending == insertion order
  std::stable_sort(ids.begin(), ids.end(), [&](uint32_t a, uint32_t b) {
    const long long ra = rowOf(a), rb = rowOf(b);
    if (ra != rb) return ra < rb;
    const uint32_t oa = incoming ? m->src[a] : m->dst[a], ob = incoming ? m->src[b] : m->dst[b];
    return oa < ob;
  });
  for (long long e = 0; e < m->nTrans; ++e) off[rowOf(e) + 1]++;
  for (long long k = 0; k < nKeys; ++k) off[k + 1] += off[k];
  perm = ids;
}

bool compile_machine(mb_machine *m, std::string *err) {
  const int S = m->S;
  m->K = (m->nIn + 1) * (m->nOut + 1);
  if ((long long)S * m->K + 1 > 0x7fffffffLL || m->nTrans > 0x7fffffffLL) {
    *err = "machine too large for 32-bit CSR offsets";
    return false;
  }
  for (long long e = 0; e < m->nTrans; ++e) {
    if (m->src[e] >= (uint32_t)S || m->dst[e] >= (uint32_t)S) { *err = "State does not exist"; return false; }
    if (m->inTok[e] > m->nIn || m->outTok[e] > m->nOut) { *err = "edge token outside alphabet"; return false; }
    if (e > 0 && m->src[e] < m->src[e - 1]) { *err = "edges must be given in global-id order (ascending source state)"; return false; }
    const bool silent = m->inTok[e] == 0 && m->outTok[e] == 0;
    // Machine::isAdvancingMachine (src/machine.cpp:758-764): checked for source states >= 1 only
    if (silent && m->src[e] >= 1 && m->dst[e] <= m->src[e]) { *err = "Machine is not topologically sorted"; return false; }
    if (m->inTok[e] && m->outTok[e]) m->hasMatch = true;
    else if (m->inTok[e]) m->hasIns = true;
    else if (m->outTok[e]) m->hasDel = true;
  }
  build_csr(m, true, m->inOff, m->inPerm);
  build_csr(m, false, m->outOff, m->outPerm);
  // silent levels.  A silent self-loop on state 0 (the one case the reference's check lets through) never
  // contributes to a fill because it reads a cell that is still -inf; it is skipped here and in the kernels.
  m->levF.assign(S, 0);
  m->levB.assign(S, 0);
  std::vector<std::vector<uint32_t>> silIn(S), silOut(S);
  for (long long e = 0; e < m->nTrans; ++e)
    if (m->inTok[e] == 0 && m->outTok[e] == 0 && m->src[e] < m->dst[e]) {
      silIn[m->dst[e]].push_back(m->src[e]);
      silOut[m->src[e]].push_back(m->dst[e]);
    }
  for (int d = 0; d < S; ++d)
    for (uint32_t s : silIn[d]) m->levF[d] = std::max(m->levF[d], m->levF[s] + 1);
  for (int s = S - 1; s >= 0; --s)
    for (uint32_t d : silOut[s]) m->levB[s] = std::max(m->levB[s], m->levB[d] + 1);
  auto group = [&](const std::vector<int> &lev, std::vector<int> &off, std::vector<int> &states, int &nLev) {
    nLev = S ? *std::max_element(lev.begin(), lev.end()) + 1 : 0;
    off.assign(nLev + 1, 0);
    for (int s = 0; s < S; ++s) off[lev[s] + 1]++;
    for (int l = 0; l < nLev; ++l) off[l + 1] += off[l];
    states.resize(S);
    std::vector<int> fill(off.begin(), off.end() - 1);
    for (int s = 0; s < S; ++s) states[fill[lev[s]]++] = s;
  };
  group(m->levF, m->levFOff, m->levFState, m->nLevF);
  group(m->levB, m->levBOff, m->levBState, m->nLevB);
  m->maxInDeg = 0;
  for (int d = 0; d < S; ++d) {
    int deg = 0;
    for (int k = 0; k < m->K; ++k) deg += m->inOff[(long long)d * m->K + k + 1] - m->inOff[(long long)d * m->K + k];
    m->maxInDeg = std::max(m->maxInDeg, deg);
  }
  return true;
}

template <class T>
static bool to_device(T *&d, const std::vector<T> &h) {
  const size_t n = std::max<size_t>(h.size(), 1);
  if (!d && !hip_ok(hipMalloc((void **)&d, n * sizeof(T)), "hipMalloc(machine table)")) return false;
  if (!h.empty() && !hip_ok(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(machine table)")) return false;
  return true;
}

bool upload_weights(mb_machine *m) {
  std::vector<double> inW(m->nTrans), outW(m->nTrans);
  for (long long a = 0; a < m->nTrans; ++a) { inW[a] = m->logW[m->inPerm[a]]; outW[a] = m->logW[m->outPerm[a]]; }
  return to_device(m->d_inW, inW) && to_device(m->d_outW, outW);
}

bool upload_machine(mb_machine *m) {
  std::vector<uint32_t> inSrc(m->nTrans), outDst(m->nTrans);
  for (long long a = 0; a < m->nTrans; ++a) { inSrc[a] = m->src[m->inPerm[a]]; outDst[a] = m->dst[m->outPerm[a]]; }
  bool ok = to_device(m->d_inOff, m->inOff) && to_device(m->d_outOff, m->outOff) && to_device(m->d_inSrc, inSrc) &&
            to_device(m->d_inEid, m->inPerm) && to_device(m->d_outDst, outDst) && to_device(m->d_outEid, m->outPerm) &&
            to_device(m->d_eInTok, m->inTok) && to_device(m->d_eOutTok, m->outTok) &&
            to_device(m->d_levFOff, m->levFOff) && to_device(m->d_levFState, m->levFState) &&
            to_device(m->d_levBOff, m->levBOff) && to_device(m->d_levBState, m->levBState) && upload_weights(m);
  if (!ok) return false;
  DevMachine &d = m->dev;
  d.S = m->S; d.nIn = m->nIn; d.nOut = m->nOut; d.K = m->K; d.nLevF = m->nLevF; d.nLevB = m->nLevB;
  d.inOff = m->d_inOff; d.inSrc = m->d_inSrc; d.inW = m->d_inW; d.inEid = m->d_inEid;
  d.outOff = m->d_outOff; d.outDst = m->d_outDst; d.outW = m->d_outW; d.outEid = m->d_outEid;
  d.eInTok = m->d_eInTok; d.eOutTok = m->d_eOutTok;
  d.levFOff = m->d_levFOff; d.levFState = m->d_levFState; d.levBOff = m->d_levBOff; d.levBState = m->d_levBState;
  return true;
}

void free_machine_device(mb_machine *m) {
  void *ptrs[] = {m->d_inOff, m->d_outOff, m->d_inSrc, m->d_inEid, m->d_outDst, m->d_outEid, m->d_inW, m->d_outW,
                  m->d_eInTok, m->d_eOutTok, m->d_levFOff, m->d_levFState, m->d_levBOff, m->d_levBState};
  for (void *p : ptrs) if (p) (void)hipFree(p);
}

}  // namespace mb
