// mb_medium_jit.cpp -- run-time specialisation of the "lanes = states" tile kernel with hiprtc (see
// mb_medium_jit_src.h for the rationale).  The generator unrolls the compiled program of ONE machine into
// straight-line HIP; everything structural becomes a literal.
#include <hip/hiprtc.h>

#include <algorithm>
#include <cstdlib>
#include <sstream>
#include <string>

#include "mb_medium.h"
#include "mb_medium_jit_src.h"

namespace mb {

static constexpr int JIT_MAX_CANDS = 12;   // candidates evaluated in one straight-line round body

static std::string generate_source(const mb_machine *m, const MedProgram &P, const MedGeom &geo, int mode, bool recsInLds) {
  std::ostringstream defs, body;
  const int S = m->S;
  defs << "#define JS " << S << "\n#define JSPAD " << P.Spad << "\n#define JNS " << P.NS << "\n#define JG " << P.G
       << "\n#define JC " << geo.C << "\n#define JWAVES " << geo.waves << "\n#define JMODE " << (mode == MB_VITERBI ? 1 : 0)
       << "\n#define JNOUT " << m->nOut << "\n#define JENDNODE " << P.dev.endNode
       << "\n#define JLDSRECS " << (recsInLds ? (long long)P.ldsImageIdx.size() : 0ll)
       << "\n#define JDUMMYOFF " << P.dummyOff << "\n#define JHALO " << (S + geo.waves * 64 - 1) / (geo.waves * 64) << "\n";
  static const char *vec[4] = {"aDiag", "aLeft", "aDown", "aCur"};
  static const char *tok[4] = {"tokM16", "itOff16", "otOff16", "q16"};
  long long ldsOff = 0;   // running record offset inside the LDS image (same order as MedProgram::ldsImageIdx)
  for (size_t r = 0; r < P.roundInfo.size(); ++r) {
    const MedRoundInfo &ri = P.roundInfo[r];
    const int n = (int)ri.slots.size();
    body << "      {  // round " << r << ": " << n << " candidate slot(s)\n";
    auto loadRec = [&](int k, const std::string &name) {
      const MedSlotInfo &sl = ri.slots[k];
      if (sl.T == 3 && recsInLds) body << "        const Rec " << name << " = ld_l(ldsRec, " << ldsOff * 16 << "u + q16);\n";
      else body << "        const Rec " << name << " = ld_g(grb + " << sl.recBase * 16 << "ull, " << tok[sl.T] << ");\n";
      if (sl.T == 3) ldsOff += P.LPG;
    };
    if (n <= JIT_MAX_CANDS) {
      for (int k = 0; k < n; ++k) loadRec(k, "r" + std::to_string(k));
      for (int k = 0; k < n; ++k)
        body << "        const double v" << k << " = med_lds(ldsb, " << vec[ri.slots[k].T] << " + (int)r" << k << ".srcOff) + r" << k << ".w;\n";
      if (n == 1) {
        body << "        const double res = v0;\n";
      } else {
        body << "        double mx = dmax(v0, v1);\n";
        for (int k = 2; k < n; ++k) body << "        mx = dmax(mx, v" << k << ");\n";
        if (mode == MB_VITERBI) body << "        const double res = mx;\n";
        else {
          body << "        const double gM = (mx == NEG_INF) ? 0.0 : mx;\n        const float sm = ex2(v0 - gM)";
          for (int k = 1; k < n; ++k) body << " + ex2(v" << k << " - gM)";
          body << ";\n        const double res = gM + (double)(__builtin_amdgcn_logf(sm) * MED_LN2);\n";
        }
      }
      body << "        *(double *)(ldsb + (aCur + (int)(active ? r0.dstOff : (unsigned)JDUMMYOFF))) = res;\n";
    } else {
      // many candidates: groups of JIT_MAX_CANDS folded into a running (max, scaled sum)
      body << "        double accM = NEG_INF; float accS = 0.0f; unsigned dstOff = 0xFFFFFFFFu;\n";
      for (int k0 = 0; k0 < n; k0 += JIT_MAX_CANDS) {
        const int k1 = std::min(n, k0 + JIT_MAX_CANDS);
        body << "        {\n";
        for (int k = k0; k < k1; ++k) loadRec(k, "r" + std::to_string(k));
        if (k0 == 0) body << "        dstOff = r0.dstOff;\n";
        for (int k = k0; k < k1; ++k)
          body << "        const double v" << k << " = med_lds(ldsb, " << vec[ri.slots[k].T] << " + (int)r" << k << ".srcOff) + r" << k << ".w;\n";
        body << "        double mx = v" << k0 << ";\n";
        for (int k = k0 + 1; k < k1; ++k) body << "        mx = dmax(mx, v" << k << ");\n";
        if (mode == MB_VITERBI) body << "        accM = dmax(accM, mx);\n";
        else {
          body << "        const double nm = dmax(accM, mx), gM = (nm == NEG_INF) ? 0.0 : nm;\n        accS = accS * ex2(accM - gM)";
          for (int k = k0; k < k1; ++k) body << " + ex2(v" << k << " - gM)";
          body << ";\n        accM = nm;\n";
        }
        body << "        }\n";
      }
      if (mode == MB_VITERBI) body << "        const double res = accM;\n";
      else body << "        const double res = ((accM == NEG_INF) ? 0.0 : accM) + (double)(__builtin_amdgcn_logf(accS) * MED_LN2);\n";
      body << "        *(double *)(ldsb + (aCur + (int)(active ? dstOff : (unsigned)JDUMMYOFF))) = res;\n";
    }
    body << "      }\n";
    if (ri.sync) body << "      med_wave_sync();\n";
  }
  std::string src = kMedJitSkeleton;
  auto replace = [&](const std::string &mark, const std::string &with) {
    const size_t p = src.find(mark);
    if (p != std::string::npos) src.replace(p, mark.size(), with);
  };
  replace("/*@DEFS@*/", defs.str());
  replace("/*@BODY@*/", body.str());
  return src;
}

bool medium_jit_get(const mb_machine *m, MedProgram &P, const MedGeom &geo, int mode) {
  MedJit &J = P.jit[mode == MB_VITERBI ? 1 : 0];
  if (J.tried) return J.func != nullptr;
  J.tried = true;
  const char *e = getenv("MB_MEDIUM_JIT");
  if (e && *e == '0') return false;
  long long totalSlots = 0;
  for (const MedRoundInfo &ri : P.roundInfo) {
    if (ri.slots.empty()) return false;
    totalSlots += (long long)ri.slots.size();
  }
  if (totalSlots > 20000) return false;
  if (P.roundInfo.size() > 4096) return false;   // keep the generated code within reach of the instruction cache
  // token-independent records go to LDS when they fit next to the ring
  const size_t ring = (size_t)P.NS * (geo.C + 1) * P.Spad * sizeof(double);
  const size_t recBytes = P.ldsImageIdx.size() * sizeof(MedRec);
  J.recsInLds = recBytes > 0 && ring + recBytes + 64 <= 160 * 1024;
  J.ldsBytes = ring + (J.recsInLds ? recBytes : 0);
  const std::string src = generate_source(m, P, geo, mode, J.recsInLds);
  if (const char *dump = getenv("MB_MEDIUM_JIT_DUMP")) {
    if (FILE *f = fopen((std::string(dump) + (mode == MB_VITERBI ? ".vit" : ".sum") + (P.backward ? ".bwd" : ".fwd") + (P.closure ? ".clos" : ".exact") + ".hip").c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
  }
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "mb_medium_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return false;
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
  const hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, 0);
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    if (getenv("MB_MEDIUM_JIT_VERBOSE")) fprintf(stderr, "[mbhip] hiprtc failed:\n%s\n", log.c_str());
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  std::string code(cs, 0);
  hiprtcGetCode(prog, &code[0]);
  hiprtcDestroyProgram(&prog);
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) return false;
  if (hipModuleGetFunction(&fn, mod, "k_medium_jit") != hipSuccess) { (void)hipModuleUnload(mod); return false; }
  (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  J.module = mod; J.func = fn;
  return true;
}

void medium_jit_free(MedProgram &P) {
  for (MedJit &J : P.jit) {
    if (J.module) (void)hipModuleUnload((hipModule_t)J.module);
    J = MedJit();
  }
}

}  // namespace mb
