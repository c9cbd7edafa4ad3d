// Exercises the C++ shim (machineboss_amd/cxx/mb_dp.hpp) on the reference's bitnoise golden case
// (t/machine/bitnoise.json, t/io/params.json p=0.99 q=0.01, t/io/tiny.json: input 001, output 101).
#include <cstdio>
#include <cmath>
#include "mb_dp.hpp"
using namespace MachineBossHIP;
int main() {
  FlatMachine m; m.nStates = 1; m.nInTok = 2; m.nOutTok = 2;
  const double p = std::log(0.99), q = std::log(0.01);
  m.addTransition(0, 0, 1, 1, p); m.addTransition(0, 0, 1, 2, q); m.addTransition(0, 0, 2, 2, p); m.addTransition(0, 0, 2, 1, q);
  m.finish();
  TokSeqPair sp{{1, 1, 2}, {2, 1, 2}};
  ForwardMatrix fwd(m, sp); BackwardMatrix bwd(m, sp); ViterbiMatrix vit(m, sp); RollingOutputForwardMatrix roll(m, sp);
  MachineCounts mc(m, {sp});
  std::printf("fwd %.5g back %.5g vit %.5g roll %.5g path %zu counts %g %g %g %g ll %.5g\n", fwd.logLike(), bwd.logLike(), vit.logLike(),
              roll.logLike(), vit.path().size(), mc.count[0][0], mc.count[0][1], mc.count[0][2], mc.count[0][3], mc.loglike);
  const bool ok = std::fabs(fwd.logLike() + 4.6253) < 1e-4 && std::fabs(bwd.logLike() + 4.6253) < 1e-4 && vit.path().size() == 3 &&
                  std::fabs(mc.count[0][0] - 1) < 1e-9 && std::fabs(mc.count[0][3]) < 1e-12 && std::fabs(fwd.cell(1, 1, 0) + 4.6052) < 1e-4 &&
                  std::isinf(fwd.cell(0, 1, 0));
  // path envelope of the alignment 0/1 0/0 1/1 (t/io/tinypath.json -> t/expect/tinypath_path_env.json [[0,1],[1,2],[2,3],[3,4]])
  Envelope env; env.initPath({{true, true}, {true, true}, {true, true}});
  const bool envOk = env.inStart == std::vector<int32_t>({0, 1, 2, 3}) && env.inEnd == std::vector<int32_t>({1, 2, 3, 4}) && env.fits(sp) && !env.isFull();
  ForwardMatrix fenv(m, sp, env);
  MachineCounts mce(m); mce.add(m, {sp}, {env});
  const bool envDp = std::fabs(fenv.logLike() + 4.6253) < 1e-4 && std::isinf(fenv.cell(1, 0, 0)) && std::fabs(mce.count[0][0] + mce.count[0][1] + mce.count[0][2] + mce.count[0][3] - 3) < 1e-9;
  Envelope area; area.initPathArea({{true, true}, {true, true}, {true, true}}, 1);
  const bool areaOk = area.inStart == std::vector<int32_t>({0, 0, 1, 2}) && area.inEnd == std::vector<int32_t>({2, 3, 4, 4});
  std::printf("env %d %d %d\n", (int)envOk, (int)envDp, (int)areaOk);
  // the collective of --train: a one-rank communicator (all a single-GPU box can form) and the no-communicator no-op
  char id[128];
  bool commOk = mb_comm_unique_id(id) == 0;
  mb_comm *comm = commOk ? mb_comm_init(id, 1, 0) : nullptr;
  commOk = commOk && comm != nullptr;
  const double before = mc.count[0][0], llBefore = mc.loglike;
  mc.allReduce(nullptr); mc.allReduce(comm);
  commOk = commOk && mc.count[0][0] == before && mc.loglike == llBefore;
  mb_comm_destroy(comm);
  std::printf("comm %d\n", (int)commOk);
  const bool all = ok && envOk && envDp && areaOk && commOk;
  std::printf(all ? "FACADE OK\n" : "FACADE MISMATCH\n");
  return all ? 0 : 1;
}
