// Probe: do the VALU issue cycles of two (four) waves on one SIMD ADD or OVERLAP?  (the busy budget of the fused fit.)
// One workgroup on one CU, every wave runs the same stream of independent fp64 VALU instructions (8 register sets); the KERNEL's
// duration (HIP events) at 1, 2, 4 waves per SIMD answers it -- s_memtime of one wave does not: the oldest wave wins the issue
// arbitration and never notices its SIMD mates (tools/valu_rate_probe.hip measures exactly that).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_overlap_probe.hip -o tools/valu_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template <int OP>
__global__ __launch_bounds__(1024) void probe(double* out, int iters) {
  double d[8];
  for (int i = 0; i < 8; ++i) d[i] = 1.0 + threadIdx.x * 1e-3 + i;
  double c = 1.0000001;
  for (int it = 0; it < iters; ++it) {
    if (OP == 0)
      asm volatile("v_fma_f64 %0, %0, %8, %8\nv_fma_f64 %1, %1, %8, %8\nv_fma_f64 %2, %2, %8, %8\nv_fma_f64 %3, %3, %8, %8\n"
                   "v_fma_f64 %4, %4, %8, %8\nv_fma_f64 %5, %5, %8, %8\nv_fma_f64 %6, %6, %8, %8\nv_fma_f64 %7, %7, %8, %8\n"
                   : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(c));
    if (OP == 1)
      asm volatile("v_mov_b64 %0, %8\nv_mov_b64 %1, %8\nv_mov_b64 %2, %8\nv_mov_b64 %3, %8\nv_mov_b64 %4, %8\nv_mov_b64 %5, %8\nv_mov_b64 %6, %8\nv_mov_b64 %7, %8\n"
                   : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(c));
    if (OP == 2) {
      int* q = reinterpret_cast<int*>(d);
      asm volatile("v_and_b32 %0, 63, %0\nv_and_b32 %1, 63, %1\nv_and_b32 %2, 63, %2\nv_and_b32 %3, 63, %3\nv_and_b32 %4, 63, %4\nv_and_b32 %5, 63, %5\nv_and_b32 %6, 63, %6\nv_and_b32 %7, 63, %7\n"
                   : "+v"(q[0]), "+v"(q[2]), "+v"(q[4]), "+v"(q[6]), "+v"(q[8]), "+v"(q[10]), "+v"(q[12]), "+v"(q[14]));
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += d[i];
  out[threadIdx.x] = s;
}

template <int OP>
void run(const char* name) {
  double* out;
  CK(hipMalloc(&out, 1024 * sizeof(double)));
  const int iters = 200000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double base = 0;
  for (int threads : {256, 512, 1024}) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, out, iters);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    if (threads == 256) base = best;
    printf("%-12s %d wave(s) per SIMD: kernel %8.3f ms = %.2f x the one-wave time; %5.2f ns per instruction and SIMD\n", name, threads / 256, best,
           best / base, best * 1e6 / (iters * 8.0 * (threads / 256)));
  }
  CK(hipFree(out));
}
int main() {
  run<0>("v_fma_f64");
  run<1>("v_mov_b64");
  run<2>("v_and_b32");
  return 0;
}
