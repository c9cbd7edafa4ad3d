// Probe for the small-S "lane = column" kernel family (round 2 design check, run once on the GPU box):
//  (1) does v_mov_b32_dpp wave_shr:1 shift across the whole wavefront on gfx950, lane 0 keeping `old`?
//  (2) issue rate of a wave_shr DPP move against a plain v_mov
//  (3) HBM write rate of the anti-diagonal store pattern: lane c writes 64 B at ((t-c)*I + i0+c)*64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_shr(int *out) {
  int v = threadIdx.x * 10 + 1;
  int old = -7;
  out[threadIdx.x] = __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
}

template <int DPP>
__global__ void k_rate(int *out, int n) {
  int v = threadIdx.x, acc = 0;
  for (int k = 0; k < n; ++k) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (DPP) v = __builtin_amdgcn_update_dpp(acc, v, 0x138, 0xf, 0xf, false);
      else v = v ^ acc;
      acc += v;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// one wavefront per tile: 64 columns x TS steps of 64-byte supercells, 4 dwordx4 stores per lane and step
__global__ __launch_bounds__(256) void k_store(double *cells, int I, int O, int TS, int nTilesPerRow) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int a = wave % nTilesPerRow, b = wave / nTilesPerRow;   // strip a, block b
  const int i = a * 64 + lane;
  const int t0 = b * TS;
  typedef double d2 __attribute__((ext_vector_type(2)));
  for (int t = t0; t < t0 + TS; ++t) {
    const int o = t - lane;
    if (o >= 0 && o < O && i < I) {
      d2 *p = (d2 *)(cells + ((long long)o * I + i) * 8);
      const d2 v = {(double)t, (double)lane};
      p[0] = v; p[1] = v; p[2] = v; p[3] = v;
    }
  }
}

int main() {
  int *d; CK(hipMalloc(&d, 1 << 22));
  hipLaunchKernelGGL(k_shr, dim3(1), dim3(64), 0, 0, d);
  std::vector<int> h(64);
  CK(hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost));
  bool ok = h[0] == -7;
  for (int l = 1; l < 64; ++l) ok = ok && h[l] == (l - 1) * 10 + 1;
  printf("wave_shr:1 %s (lane0=%d lane1=%d lane16=%d lane32=%d lane63=%d)\n", ok ? "OK" : "BROKEN", h[0], h[1], h[16], h[32], h[63]);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int dpp = 0; dpp < 2; ++dpp) {
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0));
      if (dpp) hipLaunchKernelGGL(k_rate<1>, dim3(1024), dim3(256), 0, 0, d, 4096);
      else hipLaunchKernelGGL(k_rate<0>, dim3(1024), dim3(256), 0, 0, d, 4096);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("rate %s: %.3f ms for 4096*16 (mov+add) pairs per wave, 4096 waves\n", dpp ? "wave_shr dpp" : "plain", best);
  }
  const int I = 1024, O = 65536, TS = 128;   // 1024 x 65536 supercells x 64 B = 4 GiB
  double *cells; CK(hipMalloc(&cells, (size_t)I * O * 64));
  const int nA = I / 64, nB = (O + 64) / TS;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_store, dim3(nA * nB / 4), dim3(256), 0, 0, cells, I, O, TS, nA);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("anti-diagonal 64-B stores: %.3f ms, %.1f GB/s\n", ms, (double)I * O * 64 / ms / 1e6);
  }
  CK(hipMemset(cells, 0, (size_t)I * O * 64));
  CK(hipEventRecord(e0)); CK(hipMemsetAsync(cells, 1, (size_t)I * O * 64)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("hipMemset of the same bytes: %.3f ms, %.1f GB/s\n", ms, (double)I * O * 64 / ms / 1e6); }
  return ok ? 0 : 2;
}
