/* mbhip.h -- C-ABI of the MI355X (gfx950) Forward / Backward / Viterbi DP engine for Machine Boss.
 *
 * This is the drop-in boundary for the reference's DP hot path.  The reference has no FFI layer: its
 * boundary is the C++ class interface of src/{dpmatrix,forward,backward,viterbi,counts,logsumexp}.h
 * ("construction is computation").  Each entry point below names the reference interface it replaces
 * (paths relative to the reference repository root).  INTEGRATION.md shows the C++ shim that re-implements
 * those classes on top of this header.
 *
 * Conventions
 *  - plain pointers and sizes only; all buffers are caller-owned host memory unless stated otherwise;
 *    the library owns device memory.  Calls are synchronous.
 *  - threading: the reference has no threads on this path.  Every entry point below takes one
 *    process-wide lock (ApiGuard, mb_api.hip), so calls from several host threads are safe and are serialised -- the
 *    device workspaces and compiled programs are process-wide; scale out with one process per GPU, not with threads.
 *    mb_last_error() is thread-local.  A filled matrix handed back to the caller is plain host memory.
 *  - every function returning int returns 0 on success, non-zero on error; mb_last_error() then gives the
 *    message the reference would have put into its runtime_error (src/util.cpp:39-48).
 *  - tokens are int32, token 0 = epsilon, tokens 1..N index the sorted alphabet (src/eval.h:13-22).
 *  - transitions ("edges") are identified by their GLOBAL id e = transOffset[src] + transIndex, the order in
 *    which EvaluatedMachine::init visits them (src/eval.cpp:47-69).  counts[] and Viterbi paths use these ids.
 *  - matrices use the reference's IdentityIndexMapper layout with a full envelope (src/dpmatrix.h:34-44,90-96):
 *        cell(inPos,outPos,state) = cells[((outPos*(inLen+1)) + inPos)*nStates + state]      (double)
 *  - envelopes (src/seqpair.h:75-97) restrict the cells that exist: mb_batch_set_envelopes / mb_fill_env.  The matrix
 *    handed back is still the full rectangle; cells outside the envelope hold -inf, which is what the reference's
 *    const cell() accessor returns for them (src/dpmatrix.h:142-144).  Note the reference's 3-argument DPMatrix
 *    constructor ignores its Envelope argument and uses Envelope(seqPair): the alignment's path envelope if the pair
 *    carries an alignment, else the full one (src/dpmatrix.defs.h:16-17, src/seqpair.cpp:104-110) -- the host shims
 *    (mb_dp.hpp, dp.py) reproduce that choice.
 */
#ifndef MBHIP_H_INCLUDED
#define MBHIP_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mb_machine mb_machine; /* device-resident flattened EvaluatedMachine (src/eval.h:59-98) */
typedef struct mb_batch mb_batch;     /* device-resident tokenised SeqPairList (src/seqpair.h:56-121)   */

/* fill modes for mb_fill / mb_batch_fill */
enum { MB_FORWARD = 0, MB_VITERBI = 1, MB_BACKWARD = 2 };

/* mb_batch_forward flags */
enum {
  MB_MATERIALISE = 0, /* ForwardMatrix: full (inLen+1)(outLen+1)nStates matrix kept in HBM (src/forward.h:19-27)  */
  MB_ROLLING = 1      /* RollingOutputForwardMatrix: log-likelihood only (src/dpmatrix.h:46-58, target/boss.cpp:799) */
};

/* ---- process / device ------------------------------------------------------------------------------------ */
int mb_device_count(void);        /* number of visible HIP devices (0 if none)                                 */
int mb_set_device(int device);    /* one process drives one GPU (rank-local device)                            */
int mb_synchronize(void);         /* wait for everything the library queued on its device (every compute entry point returns with its
                                     results on the host already; this is the timing bracket of a multi-rank host, bench.py)   */
const char *mb_last_error(void);  /* thread-local message of the last failing call                             */
double mb_last_device_ms(void);   /* device time (HIP events on the library's stream) of the last batch call   */
const char *mb_last_kernel_name(void); /* name of the dominant kernel of the last batch call (for profiles)     */
int64_t mb_last_launch_count(void);    /* launches of that kernel in the last batch call                            */

/* ---- machine ----------------------------------------------------------------------------------------------
 * Replaces the per-state incoming/outgoing maps built by EvaluatedMachine::init (src/eval.cpp:40-70).
 * Edges are passed in global-id order (ascending src, then transIndex).  The library derives
 *   - the `incoming` iteration order (dst, inTok, outTok, src, insertion order)   src/eval.h:66-68, eval.cpp:60-61
 *   - the `outgoing` iteration order (src, inTok, outTok, dst, insertion order)
 *   - silent-transition levels (the reference relies on state order; src/eval.cpp:44, machine.cpp:758-764)
 * Returns NULL on error (e.g. "Machine is not topologically sorted").  nInTok/nOutTok exclude epsilon. */
mb_machine *mb_machine_create(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans,
                              const uint32_t *src, const uint32_t *dst, const uint16_t *inTok,
                              const uint16_t *outTok, const double *logWeight);
/* New log-weights for the same topology: one call per EM iteration (src/fitter.cpp:28-29). */
int mb_machine_set_weights(mb_machine *m, const double *logWeight);
void mb_machine_destroy(mb_machine *m);
int32_t mb_machine_n_states(const mb_machine *m);
int64_t mb_machine_n_trans(const mb_machine *m);
int32_t mb_machine_n_levels(const mb_machine *m); /* depth of the silent-transition DAG + 1 */
/* Edge ids in the reference's iteration order, for tests: which = 0 incoming, 1 outgoing. out[nTrans]. */
int mb_machine_edge_order(const mb_machine *m, int which, uint32_t *out);

/* ---- batch of sequence pairs ------------------------------------------------------------------------------
 * Replaces the `for (const auto& seqPair: data.seqPairs)` loops of target/boss.cpp:796,826 and
 * src/counts.cpp:40-42: the whole list is tokenised by the caller (Tokenizer::tokenize, src/eval.h:29-41),
 * handed over once, and kept resident in HBM.  Pair p has input tokens inTok[inOff[p]..inOff[p+1]) and output
 * tokens outTok[outOff[p]..outOff[p+1]).  Tokens outside [1,nInTok] / [1,nOutTok] are rejected with the
 * reference's "Can't tokenize symbol" condition (src/eval.h:33-37). */
mb_batch *mb_batch_create(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff,
                          const int32_t *outTok, const int64_t *outOff);
void mb_batch_destroy(mb_batch *b);
int64_t mb_batch_cells(const mb_batch *b); /* sum over pairs of (inLen+1)(outLen+1)nStates */
/* ONE-TAPE machines (generators / recognisers of >= 256 states), every call below: a batch of fewer sequences than the device has CUs,
 * whose longest sequence has >= MB_ONETAPE_PARTS_MIN_LEN (4 096) symbols, is swept with k workgroups per sequence (the machine cut along
 * its strongly connected components).  Viterbi matrices, scores and paths are the same bits as with one workgroup; sums
 * move in the last digits.  A workgroup waits for another part's value at most MB_ONETAPE_PART_TIMEOUT_S (20) seconds: past that the
 * CALL FAILS (non-zero return, mb_last_error names the wait) -- it neither hangs nor returns partial results.  MB_ONETAPE_PARTS=1: off. */
/* Envelopes of the pairs (Envelope::inStart / inEnd, src/seqpair.h:75-97): pair p owns rows envOff[p]..envOff[p+1] of
 * inStart[] / inEnd[] -- either outLen+1 rows (cell (x,y) exists <=> inStart[y] <= x < inEnd[y]) or none (full).
 * Rejected like DPMatrix::alloc does (src/dpmatrix.defs.h:31-32): "Envelope/sequence mismatch", "Envelope is not
 * connected".  Restricted envelopes run on the fast families too (the JENV variants of the small-machine and tiled
 * kernels clip every cell outside its row's [inStart, inEnd) to -inf and do not launch tiles that hold no cell of any
 * envelope); one-tape machines and the ahead-of-time fallback take the generic family. */
int mb_batch_set_envelopes(mb_batch *b, const int64_t *envOff, const int32_t *inStart, const int32_t *inEnd);

/* Forward log-likelihoods, loglike[nPairs] (-inf allowed).
 * MB_MATERIALISE = ForwardMatrix(eval, sp).logLike()             src/forward.defs.h:23-55, src/api.cpp:32-35
 * MB_ROLLING     = RollingOutputForwardMatrix(eval, sp).logLike() target/boss.cpp:799-800               */
int mb_batch_forward(mb_batch *b, int flags, double *loglike);

/* ViterbiMatrix(eval, sp): logLike() and path()                   src/viterbi.cpp:18-51, dpmatrix.defs.h:61-110
 * loglike[nPairs]; pathOff[nPairs+1] (output) delimits each pair's start->end list of global edge ids in
 * pathEdges[pathCap].  A pair whose end cell is -inf gets an empty path (the reference refuses to trace it,
 * src/dpmatrix.defs.h:84, target/boss.cpp:831).  pathEdges may be NULL to skip tracebacks.
 * mb_viterbi_path_bound gives a sufficient pathCap contribution for one pair.
 * With paths the fill keeps ONE traceback byte per cell instead of the fp64 cell wherever the machine's family can (small and
 * tiled families always; one-tape family when the fp64 matrices would take a quarter of the memory budget, option MB_ONETAPE_TB);
 * scores and paths are the reference's either way (first maximum in its enumeration order, src/dpmatrix.defs.h:93-103,171-174). */
int64_t mb_viterbi_path_bound(const mb_machine *m, int64_t inLen, int64_t outLen);
int mb_batch_viterbi(mb_batch *b, double *loglike, int64_t *pathOff, uint32_t *pathEdges, int64_t pathCap);

/* MachineCounts(eval, seqPairList): E-step                        src/counts.cpp:37-64, src/backward.cpp:58-87
 * counts[nTrans] += posterior expected usage of every transition, summed over the batch;
 * *loglikeSum += sum of forward.logLike() (MachineCounts::loglike); loglike[nPairs] optional (may be NULL).
 * Pairs with a -inf likelihood contribute nothing to counts (the reference would produce NaN there).
 * The reference's loop is serial and reproduces bit for bit; here the order of some additions follows the scheduling (counts agree
 * to ~1e-10 from run to run).  Option MB_DETERMINISTIC=1 (mb_set_option or the environment, read when the call begins) puts every
 * such accumulator into 64-bit fixed point: repeated calls return identical counts, below 6.7e7 per transition and call.  A count in [6.7e7, 2.7e8) makes the call FAIL (every conversion saturates at 2^62, the host
 * checks the accumulators); the check is NOT airtight beyond that -- a 64-bit accumulator that several saturated terms push past 2^64
 * wraps, and a true count of 2.7e8 or more per transition and call may come back small with rc = 0: keep a call's batch below 2.7e8
 * expected uses of any one transition (2.7e8 emitted symbols), or use the floating-point mode. */
int mb_batch_counts(mb_batch *b, double *counts, double *loglikeSum, double *loglike);

/* One full matrix back to the host, for DPMatrix::cell()/writeJson() (src/dpmatrix.defs.h:39-53), the golden
 * matrix tests (t/src/testforward.cpp, testbackward.cpp) and Machine::downsample (src/machine.cpp:2053-2076).
 * mode MB_FORWARD uses startState (ForwardMatrix 4-argument ctor, src/forward.h:24); Viterbi seeds state 0
 * (src/viterbi.cpp:30); Backward seeds the last state (src/backward.cpp:33).
 * cellsOut[(inLen+1)*(outLen+1)*nStates]. */
int mb_fill(mb_machine *m, int mode, const int32_t *in, int64_t inLen, const int32_t *out, int64_t outLen,
            int32_t startState, double *cellsOut);

/* mb_fill with an envelope (envStart/envEnd: outLen+1 entries each, or both NULL for the full envelope). */
int mb_fill_env(mb_machine *m, int mode, const int32_t *in, int64_t inLen, const int32_t *out, int64_t outLen,
                int32_t startState, const int32_t *envStart, const int32_t *envEnd, double *cellsOut);

/* ---- convenience wrappers over host buffers (create batch, run, destroy) ----------------------------------
 * forwardLogLike / viterbiLogLike+viterbiAlign / forwardBackwardCounts of src/api.h:20-34.                   */
int mb_forward_batch(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff,
                     const int32_t *outTok, const int64_t *outOff, int flags, double *loglike);
int mb_viterbi_batch(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff,
                     const int32_t *outTok, const int64_t *outOff, double *loglike, int64_t *pathOff,
                     uint32_t *pathEdges, int64_t pathCap);
int mb_counts_batch(mb_machine *m, int64_t nPairs, const int32_t *inTok, const int64_t *inOff,
                    const int32_t *outTok, const int64_t *outOff, double *counts, double *loglikeSum,
                    double *loglike);

/* ---- multi-GPU: the one exchange step of the path ---------------------------------------------------------------
 * Pairs are independent, so a sharded pair list needs no collective for --loglike / --viterbi / --align.  For
 * --counts / --train the per-rank MachineCounts are summed (MachineCounts::operator+=, src/counts.cpp:66-71): ONE
 * all-reduce of nTransitions + 1 doubles per EM iteration over RCCL (xGMI).  The communicator is bootstrapped by the
 * host: rank 0 calls mb_comm_unique_id and ships the 128 bytes to the other ranks by its own means, every rank calls
 * mb_comm_init (after mb_set_device).  comm == NULL (single process) makes mb_allreduce_counts a no-op.  RCCL is
 * opened on first use; a caller that never shards never needs it. */
typedef struct mb_comm mb_comm;
int mb_comm_unique_id(char id[128]);
mb_comm *mb_comm_init(const char id[128], int nRanks, int rank);
void mb_comm_destroy(mb_comm *comm);
int mb_allreduce_counts(mb_comm *comm, double *counts, size_t n, double *loglike);   /* in place; loglike may be NULL */

/* ---- host-side log-space helpers (src/logsumexp.h:72-172) -----------------------------------------------------------------
 * The helpers of logsumexp.h that code outside the DP fills uses (fitting, prefix/beam decoders, tests).  Pure host
 * arithmetic with the reference's table semantics (100 001-entry table of log(1+exp(-x)), step 1e-4, linear interpolation,
 * 0 beyond 10 nats), bit for bit; machineboss_amd/cxx/mb_logsumexp.hpp wraps them under the reference's names. */
double mb_log_sum_exp(double a, double b);                                          /* log_sum_exp(a,b), :72-90        */
double mb_log_sum_exp_n(const double *v, size_t n);                                 /* log_sum_exp(vguard<double>), :109 */
double mb_log_inner_product(const double *v1, const double *v2, const double *v3 /* or NULL */, size_t n);   /* :143-155 */

/* ---- tuning / introspection (not part of the reference surface) ------------------------------------------ */
/* Select the kernel family: 0 = auto, 1 = generic (any machine), 2 = small-S lanes=cells, 3 = medium-S
 * lanes=states.  Used by tests to cross-check kernels against each other and by bench.py. */
int mb_set_kernel(int which);
/* Bytes of device memory the library may use for DP matrices (default: 80 % of free HBM). */
int mb_set_memory_budget(size_t bytes);
/* The library keeps its matrix pools allocated between calls (grow-only); this frees them. */
int mb_release_workspace(void);
/* What the matrix pools cost this process so far: device allocations and releases of pool slots, slots evicted to make room,
 * bytes allocated in all, and the wall-clock milliseconds spent inside hipMalloc / hipFree for them.  Steady-state calls do none
 * (the pools are grow-only and the budget that sizes chunks is sticky); a caller that sees these move between two like calls is
 * paying seconds per call for memory, not for kernels.  Any pointer may be NULL. */
int mb_alloc_stats(int64_t *poolAllocs, int64_t *poolFrees, int64_t *evictions, uint64_t *bytesAllocated, double *ms);
/* Run-time compilation (hiprtc) done by this process so far: wall-clock milliseconds, compiles, and code objects taken
 * from the on-disk cache instead ($MB_JIT_CACHE_DIR, default ~/.cache/mbhip; MB_JIT_CACHE=0 disables it). */
int mb_jit_stats(double *compileMs, int64_t *compiles, int64_t *cacheHits);
/* Transcendental instructions the log-sum-exp Forward sweep of this machine issues per lattice CELL (v_exp_f32, v_log_f32,
 * padding lanes of the kernel family included) -- what the rolling (log-likelihood-only) mode is priced against, since it
 * moves no matrix through HBM.  family: name of the kernel family that would run the sweep. */
int mb_machine_sweep_ops(mb_machine *m, double *expPerCell, double *logPerCell, const char **family);
/* Tuning knobs (kernel family thresholds, strip geometry, closure stages ...: DESIGN.md section 4.4).  They are read when
 * a machine's programs and kernels are built, from the process environment; these calls are the same switchboard for a
 * host that prefers calls.  Names start with "MB_"; value NULL or "" restores the default. */
int mb_set_option(const char *name, const char *value);
const char *mb_get_option(const char *name);

/* Writes the HIP source the run-time code generator produces for this machine (mode MB_FORWARD = sum semiring,
 * MB_VITERBI = max, 3 = Forward fused with posterior counts, 4 = max keeping one traceback byte per cell; + 16 = the tile
 * kernel that keeps no matrix (implied by 4); backward and closure (0 = levelled, K >= 1 = silent closure in
 * K stages) select the program; G = columns per
 * wavefront, 1 ... 64) to `path`.  mode + 32 writes the PROGRAM instead of the source, as the kernels read it: 16 int32 (magic
 * 0x4D454431, S, Spad, LPG, G, chunks, nIn, nOut, seedOff, dummyOff, records, usage slots, backward, closure, counting, flat),
 * chunks x 8 int32 descriptors, records of 16 bytes (fp64 weight, srcOff, dstOff), usage slots (int32 table, int32 first
 * record), one int32 per record (the transition it stands for, -1 padding) -- tests/test_tiled_plan.py replays it against the oracle.  Host only: works without a GPU, so the generated kernel
 * can be inspected / cross-compiled offline. */
int mb_debug_jit_source(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src,
                        const uint32_t *dst, const uint16_t *inTok, const uint16_t *outTok, const double *logWeight,
                        int mode, int backward, int closure, int G, const char *path);

/* The same for the small-machine family (machines of <= 16 states: lane = column, states in registers); mode 0 = sum
 * semiring, 1 = max with fp64 cells, 2 = max with one traceback byte per cell, 3 = Forward fused with posterior counts;
 * mode + 32 writes the PROGRAM the generator unrolls instead: 20 int32 (magic 0x534D5031, S, nIn, nOut, backward, seed and end
 * state, table entries, off[4], nTab[4], candidates, 3 x 0), the evaluation order, decOff [S + 1], the candidates (T, src, dup,
 * tab) in the reference's enumeration order, w[] (fp64), eid[] (int32) -- tests/test_small_plan.py replays it. */
int mb_debug_small_source(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src,
                          const uint32_t *dst, const uint16_t *inTok, const uint16_t *outTok, const double *logWeight,
                          int mode, int backward, int materialise, const char *path);

/* The retimed program of a ONE-TAPE machine (the one-tape family's sweep, mb_wide.hip) exactly as the kernel reads it,
 * written to `path`: 12 int32 -- magic 0x52455431, lanes, slots per period, ring depth NB, doubles per ring vector, largest
 * lag, penalty row length, penalty entries, period, states, 1 = ring in L2 (records name entries, not LDS addresses),
 * streams -- then (NB * slots + 8) * lanes records of 16 bytes (fp64 weight, source word, end-of-round word).
 * mode MB_FORWARD = sum semiring, MB_VITERBI = max; MB_VITERBI + 16 (forward only) = the max program that keeps one traceback
 * code per cell (the default route of --align on one-tape machines), followed by its decode tables: int32 count + tbOff
 * [states + 1], int32 count + tbEntry (position in the incoming view << 16 | emitting << 15 | source state; 0xFFFFFFFF the
 * seed), int32 count + the incoming view's global edge ids.  Host only: the planner can be checked without a device. */
int mb_debug_wide_retimed(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src,
                          const uint32_t *dst, const uint16_t *inTok, const uint16_t *outTok, const double *logWeight,
                          int mode, int backward, const char *path);

/* The same program cut for k WORKGROUPS PER SEQUENCE (batches of fewer sequences than the device has CUs): int32
 * magic 0x52455432, parts, exchange columns, states; then per part 16 int32 (lanes, slots per period, NB, doubles per ring vector,
 * largest lag, penalty row length, penalty entries, period, own states, imports, first export entry, first exchange column,
 * exports, result entry or -1, table words, byte offset of the second weights or 0), the table (machine state of every own state,
 * then the exchange column of every import), (NB * slots + 8) * lanes records and -- parts with two-transition candidates (DESIGN
 * 4.2d) -- as many doubles, the candidates' second weights.  lanes = 0: the library's own choice.  Fails when the machine's
 * transition graph has no cut. */
int mb_debug_wide_parts(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src,
                        const uint32_t *dst, const uint16_t *inTok, const uint16_t *outTok, const double *logWeight,
                        int mode, int backward, int k, int lanes, const char *path);
/* The one-tape sweep GENERATED for this machine (run-time specialised retimed kernel), rendered on the host only: HIP source to `path`,
 * the unrolled program (rounds, slots, per-lane constant table) to `path`.prog for a device-free replay; k >= 2: the machine cut for k
 * workgroups per sequence, k <= 1: the one-workgroup program; mode: MB_FORWARD / MB_VITERBI (+ 16: traceback codes, + 64: fp64 correction
 * term); compile != 0: also compiled with hiprtc (fails when the kernel would spill to scratch memory). */
int mb_debug_wide_jit(int32_t nStates, int32_t nInTok, int32_t nOutTok, int64_t nTrans, const uint32_t *src, const uint32_t *dst,
                      const uint16_t *inTok, const uint16_t *outTok, const double *logWeight, int mode, int backward, int k, int lanes, int compile, const char *path);

#ifdef __cplusplus
}
#endif
#endif /* MBHIP_H_INCLUDED */
