// mb_wide_jit.h -- the retimed one-tape sweep (mb_wide.hip: k_wide_retimed, k_wide_retimed_parts) GENERATED PER MACHINE.
//
// The ahead-of-time kernels interpret a retimed program: every slot decodes a 16-byte record streamed from L2 (source address, penalty
// entry, end-of-round word through readfirstlane), every round's end walks a reduction ladder with a scalar branch per step, decodes the
// destination word and does the address arithmetic for ring, matrix and exchange.  DESIGN 4.2d's knock-out runs showed what that costs:
// a stage of a part's period is ~100 instructions per wavefront of which a dozen are the DP's own.  Here the same program -- the same
// record streams, as the device-free replay of tests/test_retimed_plan.py reads them -- is unrolled into straight-line HIP for ONE machine
// (and one cut of it) and compiled with hiprtc through mb_jit.cpp's cache, as the tiled family does (mb_medium_jit.cpp; the reference's
// own precedent: src/compiler.cpp:536, Compiler::compileForward):
//   * the rounds of a period, their slot counts, barriers, lane-group sizes, the ring geometry and the LDS layout are LITERALS; a period
//     is unrolled once per rotation of the ring, so a source's LDS address is a per-lane constant per rotation;
//   * a lane's records -- weights, second weights, source addresses, penalty entries, destination addresses, lags, matrix and exchange
//     offsets -- are loaded ONCE into VGPRs from a per-lane constant table ([word][lane], coalesced); the slot path issues no global load;
//   * slots whose candidates are all silent skip the penalty look-up, slots without two-transition candidates the second add;
//   * the structure depends on the machine's GRAPH only: a weight update rebuilds the table, not the kernel (same source text, same module).
// Arithmetic and candidate order are the interpreter's: max programs give the same bits (cells, traceback codes, paths), sum programs the
// same sums in the same order.  The interpreter stays the checked fallback (MB_WIDE_JIT=0, hiprtc missing, rings in L2, register spills).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "mb_wide.h"

namespace mb {

// one retimed program as the ahead-of-time kernel reads it (host copies)
struct WideJitIn {
  WideRetDev ret{};                  // geometry (ret.rec unused)
  int W = 0;
  const WideRec *stream = nullptr;   // [NB][nSlots][W] (+ the ring's slack)
  const double *w2 = nullptr;        // [NB][nSlots][W] second weights, or nullptr
  bool part = false;
  int S = 0, Sg = 0;                 // ring entries below S are cells of the matrix (a part: its own states); states of the machine
  int nImp = 0, expBase = 0, nExp = 0, expIdx0 = 0, resultEntry = -1;
  const uint32_t *gmap = nullptr;    // [S] machine state of own state x (nullptr: identity)
};

// what the generator unrolls: one per part (or one for the whole machine).  `fields` is the per-lane constant table's layout -- the
// source declares one variable per field in this order, wide_jit_table fills the words in the same order
struct WideJitSlot { bool anyPen = false, anyW2 = false; };
struct WideJitRound {
  int firstSlot = 0, depth = 0;
  bool sync = false, uniform = true, anyCell = false, anyExport = false;
  int gAll = 1;                      // uniform: every wavefront's groups have this size
  std::vector<int> gWaves;           // distinct largest-group sizes of the wavefronts (not uniform: a switch over them)
  bool anyMixed = false;             // some wavefront holds groups of different sizes (masked ladder)
  int resultLane = -1;               // the lane whose node is the state the log-likelihood is read from
};
enum WideJitFieldKind { WJ_W = 0, WJ_W2, WJ_ADDR, WJ_PEN, WJ_DST, WJ_KQ, WJ_GX, WJ_XO, WJ_GL };
struct WideJitField { int kind, index, cm, words; };      // index: slot or round
// STREAMED programs (level 1: the constants of a lane do not fit its registers -- the one-workgroup program of a large machine, 24 slots and
// 1024 lanes): what depends on the ring's rotation -- a slot's source address (with its penalty entry), a round's destination address (with
// the node's lag) -- comes from a stream of packed 32-bit words, [rotation][item][lane] in execution order, read `ring` items ahead through
// buffer loads; weights, second weights and the rotation-independent words of a round stay in registers.
//   slot item:  LDS byte address << 14 | byte offset of the penalty entry in a table      round item:  lag << 18 | LDS byte address
struct WideJitItem { int kind, index; };                  // WJ_ADDR (slot) or WJ_DST (round), in execution order
struct WideJitDesc {
  WideJitIn in;
  int nSlots = 0;                    // slots of a period without the ring's padding
  int NPT = 2, U = 2;                // penalty tables, periods per unrolled loop body (lcm(NB, NPT))
  unsigned penBase = 0, tokBase = 0, expBase = 0, ringBase = 0, dummyAddr = 0;
  size_t ldsBytes = 0;
  std::vector<WideJitSlot> slots;
  std::vector<WideJitRound> rounds;
  std::vector<WideJitField> fields;
  int nWords = 0;
  int level = 0;                     // 0: every constant in registers; 1: rotation-dependent words streamed
  std::vector<WideJitItem> items;    // level 1: the items of a period
  int IP = 0, ring = 0;              // items per period with the padding, prefetch depth (IP % ring == 0)
  int regEstimate = 0;
};

struct WideJitFlags { bool viterbi = false, tb = false, acc = false, backward = false, inputTape = false; int nExpTot = 0; };

bool wide_jit_describe(const WideJitIn &in, bool acc, int level, WideJitDesc &D, std::string *why, int maxRing = 17);
// the plan of one build attempt (see mb_wide_jit.cpp): 0 the register estimate's choice ... WIDE_JIT_ATTEMPTS - 1 the least in registers
const int WIDE_JIT_ATTEMPTS = 4;
bool wide_jit_plan(const std::vector<WideJitIn> &ins, bool acc, int attempt, std::vector<WideJitDesc> &descs, std::string *why);
std::string wide_jit_source(const std::vector<WideJitDesc> &parts, const WideJitFlags &F);
void wide_jit_table(const WideJitDesc &D, const WideJitFlags &F, std::vector<uint32_t> &tab, std::vector<uint32_t> *stream = nullptr);

// compile (or find) the kernel of these parts and upload their tables; false: the interpreter keeps the program (why: for the log)
bool wide_jit_build(const std::vector<WideJitIn> &ins, const WideJitFlags &F, WideJitKernel &K, std::string *why);
int wide_jit_launch(const WideJitKernel &K, const WideDev &dev, unsigned grid, size_t ldsBytes, const PairDesc *d_desc, const int *d_tape, double *pool, double *loglike, hipStream_t st);
bool wide_jit_enabled();

}  // namespace mb
