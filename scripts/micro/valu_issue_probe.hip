// Issue cost of the vector instructions the DP tile kernels are made of, per wave64 instruction on one SIMD of gfx950, with 1, 2 and 3
// wavefronts resident on the SIMD (workgroups of 256 / 512 / 768 threads, one per CU): cycles per instruction of ONE wavefront's stream of
// independent instructions (8 accumulators), and the SIMD's aggregate.  hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 4096
template <int OP>
__global__ void k(double *out, long long *cyc, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  const double b = seed * 1e-9; const float fb = (float)b;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < REP; ++i) {
    if (OP == 0) { asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
    if (OP == 1) { asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
    if (OP == 2) { asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb)); }
    if (OP == 3) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
    if (OP == 4) { asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
    if (OP == 5) { asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
    if (OP == 6) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb) : "vcc"); }
    if (OP == 7) { asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int OP> void run(const char *name) {
  double *out; long long *cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(double)); hipMalloc(&cyc, 256 * 16 * sizeof(long long));
  printf("%-14s", name);
  for (int waves : {4, 8, 12, 16}) {
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, 1.0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * waves);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (long long c : h) s += (double)c;
    const double perWave = s / h.size() / (REP * 8.0);      // cycle-counter ticks per instruction seen by one wavefront
    printf("  %d/SIMD: %6.2f per instr per wave, %5.2f per instr per SIMD", waves / 4, perWave, perWave / (waves / 4));
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}
int main() {
  printf("ticks of s_memrealtime/readcyclecounter per wave64 instruction (REP %d x 8 independent instructions); 1, 2, 3, 4 wavefronts per SIMD\n", REP);
  run<2>("v_add_f32"); run<0>("v_add_f64"); run<1>("v_max_f64"); run<5>("v_fma_f64"); run<4>("v_cvt_f32_f64"); run<6>("v_cndmask_b32"); run<3>("v_exp_f32"); run<7>("v_log_f32");
  return 0;
}
