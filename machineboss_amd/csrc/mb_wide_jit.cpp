// mb_wide_jit.cpp -- generator of the per-machine retimed one-tape kernel (see mb_wide_jit.h).
//
// Input: the record streams of a retimed program exactly as k_wide_retimed / k_wide_retimed_parts read them (wide_ret_build; the
// format is documented at WideRetDev / WidePartDev in mb_wide.h and replayed without a device by tests/test_retimed_plan.py).  Output:
// HIP source in which a period of that program is straight-line code, and the per-lane constant table it loads once.
// Replaces, for one-tape machines, the interpreter's rendering of MappedForwardMatrix::fill (src/forward.defs.h:23-49),
// ViterbiMatrix::fill (src/viterbi.cpp:18-43) and BackwardMatrix::fill (src/backward.cpp:18-50) with inLen = 0.
#include "mb_wide_jit.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>

#include "mb_jit.h"
#include "mb_wide_jit_src.h"

namespace mb {

static int jenv(const char *name, int dflt) {
  const char *v = opt_env(name);
  return v && *v ? atoi(v) : dflt;
}
bool wide_jit_enabled() { return jenv("MB_WIDE_JIT", 1) != 0; }
// timing experiments with WRONG results (library built with -DMB_EXPERIMENTS only): bits of MB_WIDE_JIT_KNOCKOUT -- 1 no lane-group
// reduction, 2 no barriers, 4 no per-period bookkeeping (token window, penalty table, imports), 8 no exports / matrix stores
static int knockout() {
#ifdef MB_EXPERIMENTS
  return jenv("MB_WIDE_JIT_KNOCKOUT", 0);
#else
  return 0;
#endif
}

static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// ---- the streams -> what is unrolled --------------------------------------------------------------------------------------------
bool wide_jit_describe(const WideJitIn &in, bool acc, int level, WideJitDesc &D, std::string *why, int maxRing) {
  auto fail = [&](const char *msg) { if (why) *why = msg; return false; };
  D = WideJitDesc();
  D.in = in;
  const int W = in.W, NB = in.ret.NB, NVs = in.ret.NVs, nS = in.ret.nSlots, rowLen = in.ret.rowLen, nPen = in.ret.nPen;
  if (W <= 0 || W % 64 || W > 1024 || NB < 2 || NB > 4 || nS <= 0) return fail("geometry");
  const WideRec *st = in.stream;
  // rounds: a slot whose first lane carries END closes one (the flag is uniform over the lanes)
  int first = 0, lastEnd = -1;
  for (int j = 0; j < nS; ++j) {
    const uint32_t pad0 = st[(size_t)j * W].pad;
    if (!(pad0 & 0x80000000u)) continue;
    WideJitRound R;
    R.firstSlot = first; R.depth = j - first + 1; R.sync = (pad0 & 0x40000000u) != 0;
    first = j + 1; lastEnd = j;
    D.rounds.push_back(R);
  }
  if (D.rounds.empty()) return fail("no rounds");
  D.nSlots = lastEnd + 1;
  for (int j = D.nSlots; j < nS; ++j)      // (behind the last round: the ring's padding, nothing but -inf candidates)
    for (int l = 0; l < W; ++l) if (st[(size_t)j * W + l].w != -INFINITY) return fail("a candidate behind the period's last round");
  if (!D.rounds.back().sync) return fail("a period must end with a barrier");
  D.slots.assign(D.nSlots, WideJitSlot());
  for (int j = 0; j < D.nSlots; ++j)
    for (int l = 0; l < W; ++l) {
      const WideRec &rc = st[(size_t)j * W + l];
      const uint32_t penIdx = rc.src & 0x1fffu;
      if (rc.w != -INFINITY && (penIdx >= (uint32_t)nPen || penIdx % (uint32_t)rowLen != 0)) D.slots[j].anyPen = true;      // (column 0 of a lag's row: a silent candidate, penalty 0.0 always)
      if (in.w2 && in.w2[(size_t)j * W + l] != 0.0) D.slots[j].anyW2 = true;
      const uint32_t byteAddr = rc.src >> 14;
      if ((byteAddr & 7u) || byteAddr / 8u >= (uint32_t)(NB * NVs)) return fail("source address");
    }
  const int nWaves = W / 64;
  for (WideJitRound &R : D.rounds) {
    const WideRec *lastRec = st + (size_t)(R.firstSlot + R.depth - 1) * W;
    std::vector<int> gW;
    bool allSame = true; int gAll = 0;
    for (int w = 0; w < nWaves; ++w) {
      const uint32_t lg0 = (lastRec[w * 64].pad >> 26) & 7u;
      bool heads = false, mixed = false;
      for (int l = w * 64; l < w * 64 + 64; ++l) {
        const uint32_t pad = lastRec[l].pad, x = pad & WIDE_RET_NO_DST;
        if (x == WIDE_RET_NO_DST) continue;
        heads = true;
        if (((pad >> 26) & 7u) != lg0) mixed = true;
        if (x < (uint32_t)in.S) R.anyCell = true;
        if (in.part && x >= (uint32_t)in.expBase && x < (uint32_t)(in.expBase + in.nExp)) R.anyExport = true;
        if (in.resultEntry >= 0 && x == (uint32_t)in.resultEntry) R.resultLane = l;
      }
      if (mixed != (((lastRec[w * 64].pad >> 29) & 1u) != 0)) return fail("MIXED flag");
      if (!heads) continue;      // (an idle wavefront runs whatever ladder the others run: its lanes hold -inf)
      const int g = 1 << lg0;
      if (mixed) { R.anyMixed = true; allSame = false; }
      if (!gAll) gAll = g; else if (gAll != g) allSame = false;
      if (std::find(gW.begin(), gW.end(), g) == gW.end()) gW.push_back(g);
    }
    if (!gAll) gAll = 1;
    R.uniform = allSame && !R.anyMixed;
    R.gAll = gAll;
    std::sort(gW.begin(), gW.end());
    R.gWaves = gW;
  }
  // LDS: [penalty tables][token window][2^(j/64) table][ring][dummy]
  const int nPenAll = nPen + in.nImp;
  const size_t fixed = (size_t)WIDE_RET_TOKWIN * 4 + (acc ? 512 : 0) + (size_t)NB * NVs * 8 + 8;
  D.NPT = NB;
  if ((size_t)D.NPT * nPenAll * 8 + fixed > 160 * 1024) D.NPT = 2;
  if ((size_t)D.NPT * nPenAll * 8 + fixed > 160 * 1024) return fail("LDS");
  D.U = NB / gcd_i(NB, D.NPT) * D.NPT;
  D.penBase = 0;
  D.tokBase = (unsigned)((size_t)D.NPT * nPenAll * 8);
  D.expBase = D.tokBase + WIDE_RET_TOKWIN * 4;
  D.ringBase = D.expBase + (acc ? 512u : 0u);
  D.dummyAddr = D.ringBase + (unsigned)NB * NVs * 8u;
  D.ldsBytes = (size_t)D.dummyAddr + 8;
  // the per-lane constants
  D.level = level;
  if (level >= 1 && nPenAll * 8 >= (1 << 14)) return fail("penalty table beyond the packed words of a streamed program");
  auto add = [&](int kind, int index, int cm, int words) { D.fields.push_back(WideJitField{kind, index, cm, words}); D.nWords += words; };
  for (int j = 0; j < D.nSlots; ++j) {
    add(WJ_W, j, 0, 2);
    if (D.slots[j].anyW2) add(WJ_W2, j, 0, 2);
    if (level == 0) {
      for (int cm = 0; cm < NB; ++cm) add(WJ_ADDR, j, cm, 1);
      if (D.slots[j].anyPen) add(WJ_PEN, j, 0, 1);
    }
  }
  for (int r = 0; r < (int)D.rounds.size(); ++r) {
    const bool needKq = D.rounds[r].anyCell || D.rounds[r].anyExport || D.rounds[r].resultLane >= 0;
    if (level == 0) {
      for (int cm = 0; cm < NB; ++cm) add(WJ_DST, r, cm, 1);
      if (needKq) add(WJ_KQ, r, 0, 1);
    }
    if (D.rounds[r].anyCell) add(WJ_GX, r, 0, 1);
    if (D.rounds[r].anyExport) add(WJ_XO, r, 0, 1);
    if (!D.rounds[r].uniform) add(WJ_GL, r, 0, 1);
  }
  D.regEstimate = D.nWords + 40;
  if (level >= 1) {
    for (const WideJitRound &R : D.rounds) {
      for (int q = 0; q < R.depth; ++q) D.items.push_back(WideJitItem{WJ_ADDR, R.firstSlot + q});
      D.items.push_back(WideJitItem{WJ_DST, (int)(&R - D.rounds.data())});
    }
    const int n = (int)D.items.size();
    D.ring = 0;
    for (int d : {8, 9, 10, 11, 12, 7, 13, 6, 14, 15, 16, 17}) if (d <= n && d <= maxRing && n % d == 0) { D.ring = d; break; }      // a prefetch depth that divides the period, else padding
    if (D.ring) D.IP = n; else { D.ring = std::max(2, std::min(8, maxRing)); D.IP = (n + D.ring - 1) / D.ring * D.ring; }
    if (n < 6 && n <= maxRing) { D.ring = n; D.IP = n; }
    D.regEstimate += D.ring + 6;
  }
  return true;
}

void wide_jit_table(const WideJitDesc &D, const WideJitFlags &F, std::vector<uint32_t> &tab, std::vector<uint32_t> *stream) {
  const WideJitIn &in = D.in;
  const int W = in.W, NB = in.ret.NB, NVs = in.ret.NVs;
  tab.assign((size_t)D.nWords * W, 0u);
  size_t k = 0;
  auto put64 = [&](int l, double v) { uint64_t b; std::memcpy(&b, &v, 8); tab[k * W + l] = (uint32_t)b; tab[(k + 1) * W + l] = (uint32_t)(b >> 32); };
  for (const WideJitField &f : D.fields) {
    for (int l = 0; l < W; ++l) {
      if (f.kind == WJ_W || f.kind == WJ_W2 || f.kind == WJ_ADDR || f.kind == WJ_PEN) {
        const WideRec &rc = in.stream[(size_t)f.index * W + l];
        if (f.kind == WJ_W) put64(l, rc.w);
        else if (f.kind == WJ_W2) put64(l, in.w2[(size_t)f.index * W + l]);
        else if (f.kind == WJ_ADDR) {
          const uint32_t e0 = (rc.src >> 14) / 8u, vec = (e0 / (uint32_t)NVs + (uint32_t)f.cm) % (uint32_t)NB;
          tab[k * W + l] = D.ringBase + (vec * (uint32_t)NVs + e0 % (uint32_t)NVs) * 8u;
        } else tab[k * W + l] = D.penBase + (rc.src & 0x1fffu) * 8u;      // (entry of penalty table 0; the period's table is a literal offset)
      } else {
        const WideJitRound &R = D.rounds[f.index];
        const uint32_t pad = in.stream[(size_t)(R.firstSlot + R.depth - 1) * W + l].pad, x = pad & WIDE_RET_NO_DST;
        const bool none = x == WIDE_RET_NO_DST;
        const uint32_t kq = (pad >> 20) & 63u, vec0 = (pad >> 18) & 3u, lg = (pad >> 26) & 7u;
        if (f.kind == WJ_DST) tab[k * W + l] = none ? D.dummyAddr : D.ringBase + (((vec0 + (uint32_t)f.cm) % (uint32_t)NB) * (uint32_t)NVs + x) * 8u;
        else if (f.kind == WJ_KQ) tab[k * W + l] = none ? 0u : kq;
        else if (f.kind == WJ_GX) {
          const bool cell = !none && x < (uint32_t)in.S;
          const uint32_t g = cell ? (in.gmap ? in.gmap[x] : x) : 0u;
          tab[k * W + l] = cell ? (F.tb ? g : g * 8u) : 0xFFFFFFFFu;
        } else if (f.kind == WJ_XO) {
          const bool ex = !none && in.part && x >= (uint32_t)in.expBase && x < (uint32_t)(in.expBase + in.nExp);
          tab[k * W + l] = ex ? (uint32_t)(in.expIdx0 + (int)(x - (uint32_t)in.expBase)) * 8u : 0xFFFFFFFFu;
        } else {      // WJ_GL: log2 of the lane's group | 8 when the wavefront holds groups of different sizes (its first lane says so)
          const uint32_t p0 = in.stream[(size_t)(R.firstSlot + R.depth - 1) * W + (l & ~63)].pad;
          tab[k * W + l] = lg | (((p0 >> 29) & 1u) ? 8u : 0u);
        }
      }
    }
    k += (size_t)f.words;
  }
  if (D.level >= 1 && stream) {
    // [rotation][item][lane]: a slot's source (byte address << 14 | penalty byte offset), a round's destination (lag << 18 | byte address)
    stream->assign((size_t)NB * D.IP * W, 0u);
    for (int cm = 0; cm < NB; ++cm)
      for (int i = 0; i < (int)D.items.size(); ++i)
        for (int l = 0; l < W; ++l) {
          uint32_t word;
          if (D.items[i].kind == WJ_ADDR) {
            const WideRec &rc = in.stream[(size_t)D.items[i].index * W + l];
            const uint32_t e0 = (rc.src >> 14) / 8u, vec = (e0 / (uint32_t)NVs + (uint32_t)cm) % (uint32_t)NB;
            word = ((D.ringBase + (vec * (uint32_t)NVs + e0 % (uint32_t)NVs) * 8u) << 14) | ((rc.src & 0x1fffu) * 8u);
          } else {
            const WideJitRound &R = D.rounds[D.items[i].index];
            const uint32_t pad = in.stream[(size_t)(R.firstSlot + R.depth - 1) * W + l].pad, x = pad & WIDE_RET_NO_DST;
            const bool none = x == WIDE_RET_NO_DST;
            const uint32_t kq = (pad >> 20) & 63u, vec0 = (pad >> 18) & 3u;
            word = none ? D.dummyAddr : ((kq << 18) | (D.ringBase + (((vec0 + (uint32_t)cm) % (uint32_t)NB) * (uint32_t)NVs + x) * 8u));
          }
          (*stream)[((size_t)cm * D.IP + i) * W + l] = word;
        }
  }
}

// ---- the source -----------------------------------------------------------------------------------------------------------------
namespace {
struct Gen {
  std::ostringstream o;
  const WideJitDesc &D;
  const WideJitFlags &F;
  int partNo;
  int itemPos = 0;      // level 1: items taken so far in the unrolled loop body
  Gen(const WideJitDesc &d, const WideJitFlags &f, int p) : D(d), F(f), partNo(p) {}
  // byte offset of item g of the loop body (period g / IP, rotation (g / IP) mod NB) in the part's stream
  size_t itemOffset(int g) const { const int u = g / D.IP, i = g % D.IP, cm = u % D.in.ret.NB; return ((size_t)cm * D.IP + i) * D.in.W * 4; }
  // the next item's word: taken from the prefetch ring, and the ring entry reloaded `ring` items ahead (wrapping into the next iteration)
  std::string take(const char *name) {
    const int g = itemPos++, N = D.U * D.IP, slot = g % D.ring;
    std::ostringstream t;
    t << "const unsigned " << name << " = q" << slot << "; q" << slot << " = SLD(" << itemOffset((g + D.ring) % N) << "u);";
    return t.str();
  }
  static std::string fname(const WideJitField &f) {
    static const char *nm[] = {"w", "v", "a", "p", "d", "kq", "gx", "xo", "gl"};
    std::string s = std::string(nm[f.kind]) + std::to_string(f.index);
    if (f.kind == WJ_ADDR || f.kind == WJ_DST) s += "_" + std::to_string(f.cm);
    return s;
  }
  // what: 0 the mode's own reduction; 1 / 2 (two-pass sum rounds): the maximum over the group / the sum over the group
  std::string sumExpr;      // what == 3: the statement that sums a lane's exponentials once the group's maximum is known
  void reduce(const WideJitRound &R, int r, int what = 0) {
    if (knockout() & 1) { if (what == 3) o << "      " << sumExpr << "\n"; return; }
    const char *sfx = F.acc ? "64" : "";
    const std::string tabArg = F.acc ? ", EXPTAB_" : "";
    auto ladder = [&](int g, bool masked, const char *ind) {
      if (what == 3) {      // two-pass sum round: the group's maximum, every lane's exponentials relative to it, the group's sum -- one switch
        if (g > 1) { if (masked) o << ind << "jreduce_max_msk<" << g << ">(m, 1 << (int)(gl" << r << " & 7u));\n"; else o << ind << "jreduce_max_all<" << g << ">(m);\n"; }
        o << ind << sumExpr << "\n";
        if (g > 1) { if (masked) o << ind << "jreduce_fsum_msk<" << g << ">(s, 1 << (int)(gl" << r << " & 7u));\n"; else o << ind << "jreduce_fsum_all<" << g << ">(s);\n"; }
        return;
      }
      if (g <= 1 && !F.tb) return;
      if (F.tb) { o << ind << "jreduce_tb<" << g << ">(m, key, 1 << (int)(gl" << r << " & 7u));\n"; return; }
      if (what == 2) {
        if (masked) o << ind << "jreduce_fsum_msk<" << g << ">(s, 1 << (int)(gl" << r << " & 7u));\n";
        else o << ind << "jreduce_fsum_all<" << g << ">(s);\n";
        return;
      }
      if (F.viterbi || what == 1) {
        if (masked) o << ind << "jreduce_max_msk<" << g << ">(m, 1 << (int)(gl" << r << " & 7u));\n";
        else o << ind << "jreduce_max_all<" << g << ">(m);\n";
      } else {
        if (masked) o << ind << "jreduce_sum" << sfx << "_msk<" << g << ">(m, s, 1 << (int)(gl" << r << " & 7u)" << tabArg << ");\n";
        else o << ind << "jreduce_sum" << sfx << "_all<" << g << ">(m, s" << tabArg << ");\n";
      }
    };
    if (R.uniform) {
      if (F.tb) {      // (the key carries the lane's place within its group: the group size is this literal)
        if (R.gAll > 1) o << "      { const double own = m; jreduce_max_all<" << R.gAll << ">(m); key = (own == m) ? key : 0xFFFFFFFFu;"
                          << " jkey_msk<1, " << R.gAll << ">(key, 64); jkey_msk<2, " << R.gAll << ">(key, 64); jkey_msk<4, " << R.gAll << ">(key, 64); jkey_msk<8, " << R.gAll
                          << ">(key, 64); jkey_msk<16, " << R.gAll << ">(key, 64); jkey_msk<32, " << R.gAll << ">(key, 64); }\n";
      } else ladder(R.gAll, false, "      ");
      return;
    }
    // wavefronts differ: the first lane of a wavefront carries its largest group (the planner lays groups out by decreasing size)
    o << "      { const unsigned gw_ = (unsigned)__builtin_amdgcn_readfirstlane((int)gl" << r << ");\n";
    o << "        switch (gw_ & 7u) {\n";
    for (int g : R.gWaves) {
      int lg = 0; while ((1 << lg) < g) ++lg;
      o << "          case " << lg << ":\n";
      if (F.tb) ladder(g, true, "            ");
      else if (R.anyMixed) { o << "            if (gw_ & 8u) {\n"; ladder(g, true, "              "); o << "            } else {\n"; ladder(g, false, "              "); o << "            }\n"; }
      else ladder(g, false, "            ");
      o << "            break;\n";
    }
    o << "          default: " << (what == 3 ? sumExpr + " " : std::string()) << "break;\n        }\n      }\n";
  }
  void round(int r, int cm, int pt) {
    const WideJitRound &R = D.rounds[r];
    const WideJitIn &in = D.in;
    const unsigned penOff = (unsigned)pt * (unsigned)(in.ret.nPen + in.nImp) * 8u;
    const bool twoPass = !F.viterbi && jenv("MB_WIDE_JIT_TWOPASS", 1) != 0 && R.depth <= 8;
    o << "    { // round " << r << ": slots " << R.firstSlot << ".." << R.firstSlot + R.depth - 1 << "\n";
    // the LDS reads of every slot first (a streamed program: its packed words taken from the prefetch ring), then -- in the period's first
    // round -- the bookkeeping of the NEXT period (token window, penalty table, imports: a few lanes' work whose own LDS round trip then
    // overlaps with the round's reads instead of standing in front of them: it was 10 % of a period), then the arithmetic
    for (int q = 0; q < R.depth; ++q) {
      const int j = R.firstSlot + q;
      std::string addr = "a" + std::to_string(j) + "_" + std::to_string(cm), pen = "p" + std::to_string(j) + " + " + std::to_string(penOff) + "u";
      if (D.level >= 1) {
        const std::string xn = "x" + std::to_string(q) + "_";
        o << "      " << take(xn.c_str()) << "\n";
        addr = xn + " >> 14"; pen = "(" + xn + " & 0x3FFFu) + " + std::to_string(penOff) + "u";
      }
      o << "      const double r" << q << "_ = lds_rd(" << addr << ");";
      if (D.slots[j].anyPen) o << " const double p" << q << "_ = lds_rd(" << pen << ");";
      o << "\n";
    }
    if (r == 0) top(topU);
    for (int q = 0; q < R.depth; ++q) {
      const int j = R.firstSlot + q;
      o << "      const double c" << q << "_ = r" << q << "_ + " << (D.slots[j].anyPen ? "(w" + std::to_string(j) + " + p" + std::to_string(q) + "_)" : "w" + std::to_string(j));
      if (D.slots[j].anyW2) o << " + v" << j;
      o << ";\n";
    }
    if (twoPass) o << "      double m;\n";
    else {
      o << "      double m" << (F.viterbi ? " = c0_;" : " = W_NEG_BIG;") << (F.viterbi ? "" : (F.acc ? " double s = 0.0;" : " float s = 0.0f;")) << (F.tb ? " unsigned best = 0u;" : "") << "\n";
      for (int q = F.viterbi ? 1 : 0; q < R.depth; ++q) {
        if (F.viterbi) { o << "     "; if (F.tb) o << " best = c" << q << "_ > m ? " << q << "u : best;"; o << " m = jmax(m, c" << q << "_);\n"; }
        else o << "      " << (F.acc ? "jfold64(m, s, c" : "jfold(m, s, c") << q << (F.acc ? "_, EXPTAB_);\n" : "_);\n");
      }
    }
    if (F.tb) {
      if (R.uniform) { int lg = 0; while ((1 << lg) < R.gAll) ++lg; o << "      unsigned key = (best << " << lg << ") | ((unsigned)tid & " << (R.gAll - 1) << "u);\n"; }
      else o << "      unsigned key = (best << (gl" << r << " & 7u)) | ((unsigned)tid & ((1u << (gl" << r << " & 7u)) - 1u));\n";
    }
    if (twoPass) {
      o << "      m = c0_;";
      for (int q = 1; q < R.depth; ++q) o << " m = jmax(m, c" << q << "_);";
      o << " m = jmax(m, W_NEG_BIG);\n";      // (a finite stand-in: -inf - -inf below would be NaN)
      std::ostringstream se;
      if (F.acc) { se << "s = jexp64(c0_ - m, EXPTAB_);"; for (int q = 1; q < R.depth; ++q) se << " s += jexp64(c" << q << "_ - m, EXPTAB_);"; }
      else { se << "s = jexp(c0_ - m);"; for (int q = 1; q < R.depth; ++q) se << " s += jexp(c" << q << "_ - m);"; }
      sumExpr = se.str();
      o << "      " << (F.acc ? "double" : "float") << " s;\n";
      reduce(R, r, 3);
    } else
    reduce(R, r);
    if (F.viterbi) o << "      const double res = m;\n";
    else if (F.acc) o << "      const double res = s >= 0.5 ? m + (s == 1.0 ? 0.0 : jlog64(s, EXPTAB_)) : NEG_INF;\n";
    else o << "      const double res = s > 0.0f ? m + (double)__log2f(s) * 0.6931471805599453 : NEG_INF;\n";
    std::string kq = "kq" + std::to_string(r);
    if (D.level >= 1) {
      o << "      " << take("dw_") << " lds_wr(dw_ & 0x3FFFFu, res);\n";
      kq = "(dw_ >> 18)";
    } else o << "      lds_wr(d" << r << "_" << cm << ", res);\n";
    if ((R.anyCell || R.anyExport || R.resultLane >= 0) && !(knockout() & 8)) {
      o << "      { const int c = " << (F.backward ? "t - (int)" : "(t - KMAX_) + (int)") << kq << ";\n";
      if (R.anyExport) o << "        if (xo" << r << " != 0xFFFFFFFFu && (unsigned)c <= (unsigned)L) x_store((unsigned long long *)(xRowB + ((size_t)(unsigned)c * XS8_ + xo" << r << ")), res);\n";
      if (R.anyCell) {
        if (F.tb) o << "        if (codes && gx" << r << " != 0xFFFFFFFFu && (unsigned)c <= (unsigned)L) codeRow[" << kq << " * SB_ + gx" << r << "] = (unsigned char)key;\n";
        else {
          o << "        if (storeAll) { if (gx" << r << " != 0xFFFFFFFFu && (unsigned)c <= (unsigned)L) *(double *)(rowPtr + (" << kq << " * (unsigned)(SG_ * 8) + gx" << r << ")) = res; }\n";
          o << "        else if (storeLast) { if (gx" << r << " != 0xFFFFFFFFu && c == L) *(double *)((char *)cells + gx" << r << ") = res; }\n";
        }
      }
      if (R.resultLane >= 0) o << "        if (loglike && tid == " << R.resultLane << " && c == L) loglike[seq] = res;\n";
      o << "      }\n";
    }
    if (R.sync && !((knockout() & 2) && r + 1 < (int)D.rounds.size())) o << "      __syncthreads();\n";
    o << "    }\n";
  }
  int topU = 0;
  // the bookkeeping of period u: the token of column t + 2 into the window, the NEXT period's penalty table (and the imports of its newest column)
  void top(int u) {
    const WideJitIn &in = D.in;
    const int pn = (u + 1) % D.NPT, nPen = in.ret.nPen, nPenAll = nPen + in.nImp, W = in.W;
    const unsigned PN = (unsigned)pn * (unsigned)nPenAll * 8u;
    if (knockout() & 4) return;
    o << "      if (tid == 0) { lds_wri(TOK_ + (unsigned)((t + 2) & 63) * 4u, tokNext); tokNext = tokAt(t + 3); }\n";
    for (int e0 = 0; e0 < nPen; e0 += W) {
      if (e0 == 0) o << "      if (tid < " << std::min(nPen, W) << ") lds_wr(" << PN << "u + (unsigned)tid * 8u, penalty(myKt, myCol, t + 1));\n";
      else o << "      if (tid + " << e0 << " < " << nPen << ") { const int e_ = tid + " << e0 << ", kt_ = e_ / ROWLEN_; lds_wr(" << PN << "u + (unsigned)e_ * 8u, penalty(kt_, e_ - kt_ * ROWLEN_, t + 1)); }\n";
    }
    if (in.nImp > 0) {
      o << "      if (impLane) {\n"
        << "        double v_ = NEG_INF;\n"
        << "        if (t + 1 <= L) v_ = impAhead != X_EMPTY ? __longlong_as_double((long long)impAhead) : x_wait(impPtr + (size_t)(t + 1) * XS_, A.err, A.timeoutTicks);\n"
        << "        lds_wr(" << PN + (unsigned)nPen * 8u << "u + (unsigned)impI * 8u, v_);\n"
        << "        if (t + 2 <= L) impAhead = x_load(impPtr + (size_t)(t + 2) * XS_);\n"
        << "      }\n";
    }
  }
  void period(int u) {
    const WideJitIn &in = D.in;
    const int NB = in.ret.NB, cm = u % NB, pt = u % D.NPT;
    topU = u;
    o << "    // ---- period t, t mod " << D.U << " == " << u << ": newest column in ring vector " << cm << ", penalties in table " << pt << " ----\n";
    o << "    if (t >= nPer) break;\n";
    if (F.tb) o << "    codeRow = codes + (long long)(t - KMAX_) * SB_;\n";
    else o << "    rowPtr = (char *)cells + (long long)(" << (F.backward ? "L - t" : "t - KMAX_") << ") * (SG_ * 8);\n";
    for (int r = 0; r < (int)D.rounds.size(); ++r) round(r, cm, pt);
    if (D.level >= 1)      // (the padding behind a period's items: the ring moves on)
      for (int i = (int)D.items.size(); i < D.IP; ++i) { const int g = itemPos++, N = D.U * D.IP; o << "    q" << g % D.ring << " = SLD(" << itemOffset((g + D.ring) % N) << "u);\n"; }
    o << "    ++t;\n";
  }
  std::string function() {
    const WideJitIn &in = D.in;
    const int NB = in.ret.NB, W = in.W, nPen = in.ret.nPen, nPenAll = nPen + in.nImp;
    o << "// ---- part " << partNo << ": " << W << " lanes, " << D.rounds.size() << " rounds and " << D.nSlots << " slots per period, ring " << NB << " x " << in.ret.NVs
      << ", " << in.ret.kMax + 1 << " columns in flight, " << in.nImp << " imports, " << in.nExp << " exports, " << D.nWords << " constant words per lane" << (D.level ? " + a stream of packed address words" : "") << ", LDS " << D.ldsBytes << " bytes ----\n";
    o << "static __device__ __forceinline__ void part_" << partNo << "(const WideJitArgs &A, const PairDesc pd, const unsigned seq, const int *__restrict__ outTok,\n"
      << "    double *__restrict__ pool, double *__restrict__ loglike, const int lastOnly) {\n";
    o << "  constexpr int W_ = " << W << ", NB_ = " << NB << ", NVS_ = " << in.ret.NVs << ", KMAX_ = " << in.ret.kMax << ", ROWLEN_ = " << in.ret.rowLen << ", NPEN_ = " << nPen
      << ", NIMP_ = " << in.nImp << ", S_ = " << in.S << ", SG_ = " << in.Sg << ", SB_ = " << ((in.Sg + 3) & ~3) << ";\n";
    o << "  constexpr unsigned TOK_ = " << D.tokBase << "u, EXPTAB_ = " << D.expBase << "u, RING_ = " << D.ringBase << "u;\n";
    o << "  constexpr size_t XS_ = " << F.nExpTot << ", XS8_ = " << (size_t)F.nExpTot * 8 << ";\n";
    o << "  (void)EXPTAB_; (void)XS_; (void)XS8_; (void)S_; (void)SB_; (void)NIMP_;\n";
    o << "  const int tid = threadIdx.x;\n";
    o << "  const int L = " << (F.inputTape ? "pd.inLen" : "pd.outLen") << ";\n";
    o << "  const int *out = outTok + " << (F.inputTape ? "pd.inBase" : "pd.outBase") << ";\n";
    o << "  const unsigned *__restrict__ T = A.tab[" << partNo << "] + tid;\n";
    size_t k = 0;
    for (const WideJitField &f : D.fields) {
      if (f.words == 2) o << "  const double " << fname(f) << " = ldw(T + " << k * W << ", " << W << ");\n";
      else o << "  const unsigned " << fname(f) << " = T[" << k * W << "];\n";
      k += (size_t)f.words;
    }
    // the ring, the constants, the token window, the first penalty table
    o << "  for (int k = tid; k < NB_ * NVS_; k += W_) lds_wr(RING_ + (unsigned)k * 8u, NEG_INF);\n";
    o << "  if (tid < 64) lds_wri(TOK_ + (unsigned)tid * 4u, 0);\n";
    if (F.acc) o << "  if (tid < 64) lds_wr(EXPTAB_ + (unsigned)tid * 8u, exp2((double)tid * 0.015625));\n";
    o << "  __syncthreads();\n";
    o << "  if (tid < NB_) lds_wr(RING_ + (unsigned)(tid * NVS_ + S_ + 1) * 8u, 0.0);\n";
    o << "  double *cells = " << (F.tb ? "nullptr" : "pool ? pool + pd.cellBase : nullptr") << ";\n";
    o << "  unsigned char *codes = " << (F.tb ? "pool ? (unsigned char *)pool + pd.cellBase : nullptr" : "nullptr") << ";\n";
    o << "  const bool storeAll = cells && !lastOnly, storeLast = cells && lastOnly; (void)storeAll; (void)storeLast; (void)codes;\n";
    o << "  auto tokAt = [&](int c) -> int { return (c >= 1 && c <= L) ? " << (F.backward ? "out[L - c]" : "out[c - 1]") << " : 0; };\n";
    o << "  if (tid == 0) lds_wri(TOK_ + 4u, tokAt(1));\n";
    o << "  int tokNext = tid == 0 ? tokAt(2) : 0;\n";
    o << "  const int myKt = tid / ROWLEN_, myCol = tid - myKt * ROWLEN_; (void)myKt; (void)myCol;\n";
    o << "  auto penalty = [&](int kt, int col, int newest) -> double {\n"
      << "    const int c = newest - kt;\n"
      << "    const bool ok = col == 0 || (col == ROWLEN_ - 1 ? c == 0 : (c >= 1 && lds_rdi(TOK_ + (unsigned)(c & 63) * 4u) == col));\n"
      << "    return ok ? 0.0 : NEG_INF;\n  };\n";
    o << "  __syncthreads();\n";
    o << "  for (int e = tid; e < NPEN_; e += W_) { const int kt = e / ROWLEN_; lds_wr((unsigned)e * 8u, penalty(kt, e - kt * ROWLEN_, 0)); }\n";
    if (in.nImp > 0) {
      o << "  unsigned long long *xRow = (unsigned long long *)A.X + (size_t)A.xOff[seq] * XS_;\n";
      o << "  const bool impLane = tid >= W_ - NIMP_;\n  const int impI = tid - (W_ - NIMP_);\n";
      o << "  const unsigned long long *impPtr = impLane ? xRow + A.impIdx[" << partNo << "][impI] : nullptr;\n";
      o << "  unsigned long long impAhead = X_EMPTY;\n";
      o << "  if (impLane) { lds_wr((unsigned)(NPEN_ + impI) * 8u, x_wait(impPtr, A.err, A.timeoutTicks)); if (L >= 1) impAhead = x_load(impPtr + XS_); }\n";
    } else if (in.nExp > 0) o << "  unsigned long long *xRow = (unsigned long long *)A.X + (size_t)A.xOff[seq] * XS_;\n";
    if (in.nExp > 0) o << "  char *xRowB = (char *)xRow;\n";
    o << "  __syncthreads();\n";
    if (D.level >= 1) {
      o << "  const __amdgpu_buffer_rsrc_t srs_ = __builtin_amdgcn_make_buffer_rsrc((void *)A.stream[" << partNo << "], 0, 0x7fffffff, 0x00020000);\n";
      o << "  const int laneOff4_ = tid * 4;\n";
      o << "#define SLD(off) ((unsigned)__builtin_amdgcn_raw_buffer_load_b32(srs_, laneOff4_, (int)(off), 0))\n";
      for (int k = 0; k < D.ring; ++k) o << "  unsigned q" << k << " = SLD(" << itemOffset(k) << "u);\n";
    }
    o << "  const int nPer = L + 1 + KMAX_;\n";
    o << "  char *rowPtr = nullptr; unsigned char *codeRow = nullptr; (void)rowPtr; (void)codeRow;\n";
    o << "  int t = 0;\n";
    o << "  for (;;) {\n";
    for (int u = 0; u < D.U; ++u) period(u);
    o << "  }\n";
    if (D.level >= 1) o << "#undef SLD\n";
    o << "}\n\n";
    (void)nPenAll;
    return o.str();
  }
};
}  // namespace

std::string wide_jit_source(const std::vector<WideJitDesc> &parts, const WideJitFlags &F) {
  std::ostringstream s;
  s << "// generated by mb_wide_jit.cpp: retimed one-tape sweep, " << (F.viterbi ? (F.tb ? "max semiring keeping traceback codes" : "max semiring") : (F.acc ? "log-sum-exp (fp64 correction term)" : "log-sum-exp"))
    << ", " << (F.backward ? "backward" : "forward") << ", " << parts.size() << (parts.size() == 1 ? " workgroup" : " workgroups") << " per sequence\n";
  s << kWideJitPrelude;
  int W = 0;
  for (size_t p = 0; p < parts.size(); ++p) { Gen g(parts[p], F, (int)p); s << g.function(); W = parts[p].in.W; }
  s << "extern \"C\" __global__ __launch_bounds__(" << W << ") void k_wide_jit(WideDev P, WideJitArgs A, const PairDesc *__restrict__ pairs, const int *__restrict__ outTok,\n"
    << "                                                            double *__restrict__ pool, double *__restrict__ loglike) {\n";
  if (parts.size() == 1) s << "  part_0(A, pairs[blockIdx.x], blockIdx.x, outTok, pool, loglike, P.lastOnly);\n";
  else {
    // workgroup = part * nSeq + sequence: a part waits for lower parts only, and those are dispatched first
    s << "  const unsigned part = blockIdx.x / (unsigned)A.nSeq, seq = blockIdx.x - part * (unsigned)A.nSeq;\n";
    s << "  switch (__builtin_amdgcn_readfirstlane((int)part)) {\n";
    for (size_t p = 0; p < parts.size(); ++p) s << "    case " << p << ": part_" << p << "(A, pairs[seq], seq, outTok, pool, loglike, P.lastOnly); break;\n";
    s << "    default: break;\n  }\n";
  }
  s << "}\n";
  return s.str();
}

// ---- modules ----------------------------------------------------------------------------------------------------------------------
WideJitModule::~WideJitModule() { if (module) (void)hipModuleUnload((hipModule_t)module); }
void WideJitKernel::release() {
  for (uint32_t *p : d_tab) if (p) (void)hipFree(p);
  for (uint32_t *p : d_stream) if (p) (void)hipFree(p);
  d_tab.clear(); d_stream.clear();
  mod.reset();
  tried = false;
}

// modules by source text: a program rebuilt after a weight update has the structure it had (the planner keeps lanes, merges and periods
// across updates) and finds its kernel here instead of going through hiprtc's cache and the loader again
static std::map<std::string, std::shared_ptr<WideJitModule>> g_wideJitModules;

static bool wide_jit_build_one(const std::vector<WideJitDesc> &descs, const std::string &src, const std::vector<WideJitIn> &ins, const WideJitFlags &F, WideJitKernel &K, std::string *why, bool *spilled);

// Where every part keeps its constants.  The compiler must hold them in registers: level 0 -- everything -- when a workgroup of this many
// wavefronts has the registers by the estimate, else level 1 -- the rotation-dependent words streamed through a prefetch ring.  The
// estimate is not the register allocator: a build that spills (config 5's Backward parts with the fp64 correction term at 2 workgroups
// per sequence: 28 bytes of scratch) is planned again with less in registers -- attempt 1: every part streams; 2, 3: a prefetch ring of
// 4, 2 words.  MB_WIDE_JIT_LEVEL forces the level of attempt 0.
bool wide_jit_plan(const std::vector<WideJitIn> &ins, bool acc, int attempt, std::vector<WideJitDesc> &descs, std::string *why) {
  auto fail = [&](const std::string &msg) { if (why) *why = msg; return false; };
  descs.assign(ins.size(), WideJitDesc());
  const int wavesPerSimd = (ins[0].W / 64 + 3) / 4, regs = 512 / wavesPerSimd;
  const int forceLevel = attempt == 0 ? jenv("MB_WIDE_JIT_LEVEL", -1) : 1;
  const int maxRing = attempt <= 1 ? 17 : (attempt == 2 ? 4 : 2);
  for (size_t p = 0; p < ins.size(); ++p) {
    std::string w;
    bool ok = false;
    for (int level = forceLevel >= 0 ? forceLevel : 0; level <= (forceLevel >= 0 ? forceLevel : 1) && !ok; ++level) {
      if (!wide_jit_describe(ins[p], acc, level, descs[p], &w, maxRing)) return fail("part " + std::to_string(p) + ": " + w);
      ok = descs[p].regEstimate <= regs || forceLevel >= 0;
    }
    if (!ok) return fail("constants per lane (" + std::to_string(descs[p].nWords) + " words even with the address words streamed) beyond the register file of " + std::to_string(wavesPerSimd) + " wavefronts per SIMD");
    if (descs[p].in.W != descs[0].in.W) return fail("lanes differ between parts");
  }
  return true;
}

bool wide_jit_build(const std::vector<WideJitIn> &ins, const WideJitFlags &F, WideJitKernel &K, std::string *why) {
  K.release();
  K.tried = true;
  auto fail = [&](const std::string &msg) { if (why) *why = msg; return false; };
  if (!wide_jit_enabled()) return fail("MB_WIDE_JIT=0");
  if (ins.empty() || ins.size() > (size_t)WIDE_JIT_MAX_PARTS) return fail("parts");
  std::string lastSrc, lastWhy = "no plan";
  for (int attempt = 0; attempt < WIDE_JIT_ATTEMPTS; ++attempt) {
    std::vector<WideJitDesc> descs;
    std::string w;
    if (!wide_jit_plan(ins, F.acc, attempt, descs, &w)) { if (attempt == 0) return fail(w); break; }
    const std::string src = wide_jit_source(descs, F);
    if (src == lastSrc) continue;      // (nothing left to move out of the registers at this step)
    lastSrc = src;
    bool spilled = false;
    if (wide_jit_build_one(descs, src, ins, F, K, &lastWhy, &spilled)) return true;
    if (!spilled) break;
    if (opt_env("MB_WIDE_VERBOSE")) fprintf(stderr, "[mbhip] wide jit: attempt %d spills (%s): planned again with fewer constants in registers\n", attempt, lastWhy.c_str());
  }
  K.release(); K.tried = true;
  return fail(lastWhy);
}

static bool wide_jit_build_one(const std::vector<WideJitDesc> &descs, const std::string &src, const std::vector<WideJitIn> &ins, const WideJitFlags &F, WideJitKernel &K, std::string *why, bool *spilled) {
  auto fail = [&](const std::string &msg) { if (why) *why = msg; return false; };
  size_t lds = 0;
  for (const WideJitDesc &D : descs) lds = std::max(lds, D.ldsBytes);
  if (const char *dump = opt_env("MB_WIDE_JIT_DUMP")) {
    const std::string path = std::string(dump) + (F.viterbi ? (F.tb ? ".tb" : ".max") : (F.acc ? ".sum64" : ".sum")) + (F.backward ? ".bwd" : ".fwd") + ".k" + std::to_string(ins.size()) + ".hip";
    if (FILE *f = fopen(path.c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
  }
  auto it = g_wideJitModules.find(src);
  if (it != g_wideJitModules.end()) K.mod = it->second;
  else {
    std::string code, log;
    bool fromCache = false;
    if (!jit_compile(src, "mb_wide_jit.hip", code, &log, &fromCache)) {
      if (opt_env("MB_WIDE_VERBOSE")) fprintf(stderr, "[mbhip] wide jit: hiprtc failed:\n%s\n", log.c_str());
      return fail("hiprtc: " + log.substr(0, 400));
    }
    const long long scratch = jit_kernel_meta(code, ".private_segment_fixed_size");
    if (opt_env("MB_WIDE_VERBOSE")) fprintf(stderr, "[mbhip] wide jit: %zu bytes of source, code object %zu bytes%s, scratch %lld bytes, %lld spilled VGPRs\n", src.size(), code.size(), fromCache ? " (cache)" : "", scratch, jit_kernel_meta(code, ".vgpr_spill_count"));
    if (scratch > 0 && !jenv("MB_WIDE_JIT_ALLOW_SCRATCH", 0)) { *spilled = true; return fail("the kernel spills to scratch memory (" + std::to_string(scratch) + " bytes)"); }
    hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
      (void)hipGetLastError();
      mod = nullptr;
      if (fromCache) { jit_evict(src); if (!jit_compile(src, "mb_wide_jit.hip", code, &log, nullptr) || hipModuleLoadData(&mod, code.data()) != hipSuccess) mod = nullptr; }
      if (!mod) return fail("hipModuleLoadData");
    }
    if (hipModuleGetFunction(&fn, mod, "k_wide_jit") != hipSuccess) { (void)hipModuleUnload(mod); return fail("hipModuleGetFunction"); }
    (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    K.mod = std::make_shared<WideJitModule>();
    K.mod->module = mod; K.mod->func = fn;
    if (g_wideJitModules.size() >= 48) g_wideJitModules.clear();      // (programs in use keep their modules alive through their own references)
    g_wideJitModules[src] = K.mod;
  }
  K.ldsBytes = lds; K.W = descs[0].in.W; K.k = (int)ins.size();
  K.args = WideJitArgs{};
  K.d_tab.assign(ins.size(), nullptr); K.d_stream.assign(ins.size(), nullptr);
  for (size_t p = 0; p < ins.size(); ++p) {
    std::vector<uint32_t> tab, stream;
    wide_jit_table(descs[p], F, tab, &stream);
    if (hipMalloc((void **)&K.d_tab[p], std::max<size_t>(tab.size(), 1) * 4) != hipSuccess ||
        hipMemcpy(K.d_tab[p], tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); K.release(); K.tried = true; return fail("table upload"); }
    K.args.tab[p] = K.d_tab[p];
    if (!stream.empty()) {
      if (hipMalloc((void **)&K.d_stream[p], stream.size() * 4) != hipSuccess ||
          hipMemcpy(K.d_stream[p], stream.data(), stream.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); K.release(); K.tried = true; return fail("stream upload"); }
      K.args.stream[p] = K.d_stream[p];
    }
  }
  return true;
}

int wide_jit_launch(const WideJitKernel &K, const WideDev &dev, unsigned grid, size_t ldsBytes, const PairDesc *d_desc, const int *d_tape, double *pool, double *loglike, hipStream_t st) {
  if (!K.mod || !K.mod->func) { set_error("one-tape generated kernel: not built"); return 1; }
  WideDev d = dev; WideJitArgs a = K.args;
  void *args[] = {&d, &a, (void *)&d_desc, (void *)&d_tape, &pool, &loglike};
  if (!hip_ok(hipModuleLaunchKernel((hipFunction_t)K.mod->func, grid, 1, 1, (unsigned)K.W, 1, 1, (unsigned)std::max(ldsBytes, K.ldsBytes), st, args, nullptr), "launch of the generated one-tape kernel")) return 1;
  return 0;
}

}  // namespace mb
