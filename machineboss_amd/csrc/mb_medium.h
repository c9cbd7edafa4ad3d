// mb_medium.h -- host/device structures of the "lanes = states" tiled kernel family (see mb_medium.hip).
#pragma once
#include <vector>

#include "mb_internal.h"

namespace mb {

constexpr int MED_MAXSLOT = 8;   // candidate slots evaluated together (two-pass max / sum-exp in registers)

// The compiled "program" of a machine for one sweep direction.
//   round  = up to LPG states of one silent level, one state per lane of a lane group;
//   slot   = one candidate (source state, log-weight) per lane; slots of a round are grouped by table
//            (0 match -> (i-1,o-1), 1 input-only -> (i-1,o), 2 output-only -> (i,o-1), 3 silent -> (i,o));
//   chunk  = up to MED_MAXSLOT slots of one round, the unit the kernel software-pipelines.
// meta[chunk*(1+MED_MAXSLOT)]: word 0 = ns | first<<4 | last<<5 | sync<<6 | round<<8;
//                              word 1+k = slot offset (28 bits) | table<<28.
// slot arrays: src/w[offset + token*LPG + laneInGroup], token = table-specific (pair index, inTok, outTok, 0).
struct MedProgDev {
  int S, Spad, R, LPG, G, NS, nChunks;
  int nIn, nOut, startNode, endNode;
  const int *meta;
  const short *dest;        // [R*LPG] state finalised by (round, laneInGroup), -1 = idle
  const uint16_t *src;      // candidate source state (padding: S, the -inf sentinel)
  const double *w;          // candidate log-weight   (padding: -inf)
};

struct MedProgram {
  int G = 0, LPG = 0, R = 0, NS = 0, Spad = 0, nChunks = 0;
  bool backward = false;
  std::vector<int> meta;
  std::vector<short> dest;
  std::vector<uint16_t> src;
  std::vector<uint32_t> eid;   // slot entry -> global edge id (0xFFFFFFFF = padding), to refresh weights per EM iteration
  int *d_meta = nullptr;
  short *d_dest = nullptr;
  uint16_t *d_src = nullptr;
  double *d_w = nullptr;
  MedProgDev dev{};
};

struct MedGeom { int waves = 0, C = 0; size_t ldsBytes = 0; };

bool medium_build(const mb_machine *m, bool backward, int G, MedProgram &P);
bool medium_refresh_weights(const mb_machine *m, MedProgram &P);
void medium_free(MedProgram &P);
bool medium_geometry(const mb_machine *m, const MedProgram &P, MedGeom &geo);
int medium_fill_materialised(const mb_machine *m, const MedProgram &P, const MedGeom &geo, int mode, const PairDesc *d_pairs,
                             const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out, double *d_pool,
                             hipStream_t st);
int medium_forward_rolling(const mb_machine *m, const MedProgram &P, const MedGeom &geo, const PairDesc *d_pairs,
                           const std::vector<PairDesc> &pairs, const int *d_in, const int *d_out, double *d_colHalo,
                           const long long *d_haloBase, double *d_loglike, hipStream_t st);

}  // namespace mb
