// mb_medium.h -- host/device structures of the "lanes = states" tiled kernel family (see mb_medium.hip).
#pragma once
#include <vector>

#include "mb_internal.h"

namespace mb {

struct MedTabDev {
  const uint16_t *src;   // [slots] candidate source state (padding: 0)
  const double *w;       // [slots] candidate log-weight   (padding: -inf)
  const int *base;       // [R]  first slot of round r
  const int *nslots;     // [R]  candidate slots per (token, lane) in round r (wave-uniform loop bound)
};

// tables: 0 = match (token pair), 1 = input-only, 2 = output-only, 3 = silent;
// slot index = base[r] + (token*nslots[r] + k)*LPG + laneInGroup
struct MedProgDev {
  int S, Spad, R, LPG, G, NS;
  int nIn, nOut, startNode, endNode;
  const short *dest;            // [R*LPG] state finalised by (round, laneInGroup), -1 = idle
  const unsigned char *sync;    // [R] 1: a silent-level boundary follows round r
  MedTabDev tab[4];
};

struct MedProgram {
  int G = 0, LPG = 0, R = 0, NS = 0, Spad = 0;
  bool backward = false;
  std::vector<short> dest;
  std::vector<unsigned char> sync;
  std::vector<int> base[4], nslots[4];
  std::vector<uint16_t> src[4];
  std::vector<uint32_t> eid[4];   // slot -> global edge id (0xFFFFFFFF = padding), to refresh weights per EM iteration
  short *d_dest = nullptr;
  unsigned char *d_sync = nullptr;
  uint16_t *d_src[4] = {nullptr, nullptr, nullptr, nullptr};
  double *d_w[4] = {nullptr, nullptr, nullptr, nullptr};
  int *d_base[4] = {nullptr, nullptr, nullptr, nullptr};
  int *d_nslots[4] = {nullptr, nullptr, nullptr, nullptr};
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
