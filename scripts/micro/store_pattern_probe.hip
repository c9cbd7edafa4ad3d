// Which store shape reaches the HBM write ceiling for the anti-diagonal sweep of 64-byte supercells (8 states x fp64)?
//   mode 0: every lane stores its own supercell with 4 x 16 B (64 requests of 16 B per wave instruction)
//   mode 1: the 4 lanes of a quad store ONE supercell per instruction (16 requests of 64 B): needs a 4 x 4 transpose
//           of the 16-byte pieces inside each quad (LDS) in the real kernel; here only the address pattern is timed
//   mode 2: mode 1 + two steps paired so that 8 lanes cover (i,o),(i+1,o): 128-byte runs
//   mode 3: loads in the shape of mode 0;  mode 4: loads in the shape of mode 1
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k_pat(double *cells, int I, int O, int TS, int nA, double *sink) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int a = wave % nA, b = wave / nA;
  const int t0 = b * TS;
  d2 acc = {0.0, 0.0};
  for (int t = t0; t < t0 + TS; t += (MODE == 2 ? 2 : 1)) {
    if (MODE == 0 || MODE == 3) {
      const int i = a * 64 + lane, o = t - lane;
      if (o >= 0 && o < O) {
        d2 *p = (d2 *)(cells + ((long long)o * I + i) * 8);
        if (MODE == 0) { const d2 v = {(double)t, (double)lane}; p[0] = v; p[1] = v; p[2] = v; p[3] = v; }
        else { acc += p[0]; acc += p[1]; acc += p[2]; acc += p[3]; }
      }
    } else if (MODE == 1 || MODE == 4) {
      const int q = lane & 3, g = lane & ~3;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int col = g + m, i = a * 64 + col, o = t - col;
        if (o >= 0 && o < O) {
          d2 *p = (d2 *)(cells + ((long long)o * I + i) * 8) + q;
          if (MODE == 1) { const d2 v = {(double)t, (double)lane}; *p = v; } else acc += *p;
        }
      }
    } else {
      // two steps t, t+1: lanes 8r..8r+7 write supercells (col, o = t - col) [step t] and (col + 1, o) [step t+1]: 128 contiguous bytes
      const int q = lane & 7, r = lane >> 3;
#pragma unroll
      for (int m = 0; m < 16; ++m) {          // 128 supercells of the two steps = 64 runs of 2; 8 runs per instruction
        const int run = m * 8 + r;            // run k: even columns pair (2k', ...) -- pattern only
        const int col = run & 63, step = run >> 6;   // 0..63, 0..1
        const int o = t + step - col - (step ? 0 : 0);
        const int i = a * 64 + (col & ~1) + (q >> 2);
        const int oo = t - (col & ~1);
        (void)o;
        if (oo >= 0 && oo < O && step == 0) {
          d2 *p = (d2 *)(cells + ((long long)oo * I + i) * 8) + (q & 3);
          const d2 v = {(double)t, (double)lane}; *p = v;
        } else if (oo + 1 >= 0 && oo + 1 < O && step == 1) {
          d2 *p = (d2 *)(cells + ((long long)(oo + 1) * I + i) * 8) + (q & 3);
          const d2 v = {(double)t, (double)lane}; *p = v;
        }
      }
    }
  }
  if (MODE >= 3 && acc.x == 12345.678) sink[0] = acc.y;
}

int main() {
  const int I = 1024, O = 65536, TS = 128;
  double *cells, *sink; CK(hipMalloc(&cells, (size_t)I * O * 64)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(cells, 0, (size_t)I * O * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nA = I / 64, nB = (O + 64) / TS;
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0));
      const dim3 grid(nA * nB / 4), block(256);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_pat<0>, grid, block, 0, 0, cells, I, O, TS, nA, sink); break;
        case 1: hipLaunchKernelGGL(k_pat<1>, grid, block, 0, 0, cells, I, O, TS, nA, sink); break;
        case 2: hipLaunchKernelGGL(k_pat<2>, grid, block, 0, 0, cells, I, O, TS, nA, sink); break;
        case 3: hipLaunchKernelGGL(k_pat<3>, grid, block, 0, 0, cells, I, O, TS, nA, sink); break;
        default: hipLaunchKernelGGL(k_pat<4>, grid, block, 0, 0, cells, I, O, TS, nA, sink); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("mode %d: %.3f ms  %.1f GB/s\n", mode, best, (double)I * O * 64 / best / 1e6);
  }
  return 0;
}
