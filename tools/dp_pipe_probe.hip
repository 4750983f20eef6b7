// Probe (gfx950): do fp64 VALU instructions of one wave run UNDER the fp64 MFMAs of another wave on the same SIMD?
// One workgroup of 512 threads on one CU: waves 0-3 (one per SIMD) stream independent v_mfma_f64_16x16x4_f64, waves 4-7 (their
// SIMD partners) stream v_fma_f64 / v_fma_f32 / v_rsq_f32.  Timed alone and together (HIP events around the launch).
// Build: hipcc --offload-arch=gfx950 -O3 tools/dp_pipe_probe.hip -o tools/dp_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <bool MFMA, int VALU>   // VALU: 0 none, 1 v_fma_f64, 2 v_fma_f32, 3 v_rsq_f32
__global__ __launch_bounds__(512) void probe(double* out, int iters_m, int iters_v) {
  const int wave = threadIdx.x >> 6;
  double r = 0.0;
  if (wave < 4) {
    if (MFMA) {
      d4 acc[8];
      for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
      const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
      for (int it = 0; it < iters_m; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    }
  } else if (VALU) {
    double d[8]; float f[8];
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = 1.5f + i; }
    const double c = 0.9999999;
    for (int it = 0; it < iters_v; ++it) {
      if (VALU == 1)
        asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\n"
                     "v_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8\n"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(c));
      if (VALU == 2)
        asm volatile("v_fma_f32 %0, %0, %0, %0\nv_fma_f32 %1, %1, %1, %1\nv_fma_f32 %2, %2, %2, %2\nv_fma_f32 %3, %3, %3, %3\n"
                     "v_fma_f32 %4, %4, %4, %4\nv_fma_f32 %5, %5, %5, %5\nv_fma_f32 %6, %6, %6, %6\nv_fma_f32 %7, %7, %7, %7\n"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
      if (VALU == 3)
        asm volatile("v_rsq_f32 %0, %0\nv_rsq_f32 %1, %1\nv_rsq_f32 %2, %2\nv_rsq_f32 %3, %3\n"
                     "v_rsq_f32 %4, %4\nv_rsq_f32 %5, %5\nv_rsq_f32 %6, %6\nv_rsq_f32 %7, %7\n"
                     : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
    }
    for (int i = 0; i < 8; ++i) r += d[i] + f[i];
  }
  out[threadIdx.x] = r;
}

template <bool MFMA, int VALU>
float run(double* out, int im, int iv) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MFMA, VALU>), dim3(1), dim3(512), 0, 0, out, im, iv);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  double* out; CK(hipMalloc(&out, 512 * sizeof(double)));
  const int im = 20000;              // 160,000 MFMAs per wave: 10.24 M cycles of a 64-cycle instruction
  const int iv16 = 16 * im;          // v_fma_f64: 4 cycles each -> 8 x 4 x 16 x im = the same 10.24 M cycles
  printf("us: MFMA f64 alone %.0f\n", run<true, 0>(out, im, 0));
  printf("us: v_fma_f64 alone %.0f, together with the MFMAs %.0f\n", run<false, 1>(out, 0, iv16), run<true, 1>(out, im, iv16));
  printf("us: v_fma_f32 alone %.0f, together with the MFMAs %.0f\n", run<false, 2>(out, 0, iv16), run<true, 2>(out, im, iv16));
  printf("us: v_rsq_f32 alone %.0f, together with the MFMAs %.0f\n", run<false, 3>(out, 0, iv16 / 4), run<true, 3>(out, im, iv16 / 4));
  return 0;
}
