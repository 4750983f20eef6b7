// Probe: issue cost of the VALU instructions the kernel-matrix build is made of (gfx950), one wave per SIMD
// and two waves per SIMD.  s_memtime ticks per instruction and wave, 8 independent register sets.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

#define OP8(STR)                                                                                     \
  asm volatile(STR(0) STR(1) STR(2) STR(3) STR(4) STR(5) STR(6) STR(7)                               \
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), \
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])  \
               : "v"(c), "s"(sc))
// operand numbering: %0-7 doubles d, %8-15 32-bit f, %16 double c (VGPR), %17 double sc (SGPR)
#define S_FMA(i)    "v_fma_f64 %" #i ", %" #i ", %16, %16\n"
#define S_MUL(i)    "v_mul_f64 %" #i ", %" #i ", %16\n"
#define S_ADD(i)    "v_add_f64 %" #i ", %" #i ", %16\n"
#define S_MAXS(i)   "v_max_f64 %" #i ", %" #i ", %17\n"
#define S_RNDNE(i)  "v_rndne_f64 %" #i ", %" #i "\n"
#define S_LDEXP(i)  "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define S_MOV64(i)  "v_mov_b64 %" #i ", %16\n"
#define S_CVTF32(i) "v_cvt_f32_f64 %1" #i ", %" #i "\n"
template <int OP>
__global__ __launch_bounds__(1024) void probe(double* out, long long* cyc, int iters) {
  double d[8]; float f[8];
  for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = 1.5f + i; }
  double c = 1.0000001, sc = 0.5;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) OP8(S_FMA);
    if (OP == 1) OP8(S_MUL);
    if (OP == 2) OP8(S_ADD);
    if (OP == 3) OP8(S_MAXS);
    if (OP == 4) OP8(S_RNDNE);
    if (OP == 5) OP8(S_LDEXP);
    if (OP == 6) OP8(S_MOV64);
    if (OP == 7) asm volatile("v_cvt_f32_f64 %8, %0\nv_cvt_f32_f64 %9, %1\nv_cvt_f32_f64 %10, %2\nv_cvt_f32_f64 %11, %3\nv_cvt_f32_f64 %12, %4\nv_cvt_f32_f64 %13, %5\nv_cvt_f32_f64 %14, %6\nv_cvt_f32_f64 %15, %7\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc));
    if (OP == 8) asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\nv_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc));
    if (OP == 9) asm volatile("v_rsq_f32 %8, %8\nv_rsq_f32 %9, %9\nv_rsq_f32 %10, %10\nv_rsq_f32 %11, %11\nv_rsq_f32 %12, %12\nv_rsq_f32 %13, %13\nv_rsq_f32 %14, %14\nv_rsq_f32 %15, %15\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc));
    if (OP == 10) asm volatile("v_cvt_i32_f64 %8, %0\nv_cvt_i32_f64 %9, %1\nv_cvt_i32_f64 %10, %2\nv_cvt_i32_f64 %11, %3\nv_cvt_i32_f64 %12, %4\nv_cvt_i32_f64 %13, %5\nv_cvt_i32_f64 %14, %6\nv_cvt_i32_f64 %15, %7\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc));
    if (OP == 11) asm volatile("v_and_b32 %8, 63, %8\nv_and_b32 %9, 63, %9\nv_and_b32 %10, 63, %10\nv_and_b32 %11, 63, %11\nv_and_b32 %12, 63, %12\nv_and_b32 %13, 63, %13\nv_and_b32 %14, 63, %14\nv_and_b32 %15, 63, %15\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc));
    if (OP == 12) asm volatile("v_lshl_add_u32 %8, %8, 3, %9\nv_lshl_add_u32 %9, %9, 3, %10\nv_lshl_add_u32 %10, %10, 3, %11\nv_lshl_add_u32 %11, %11, 3, %12\nv_lshl_add_u32 %12, %12, 3, %13\nv_lshl_add_u32 %13, %13, 3, %14\nv_lshl_add_u32 %14, %14, 3, %15\nv_lshl_add_u32 %15, %15, 3, %8\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc));
    if (OP == 13) asm volatile("v_fma_f32 %8, %8, %8, %8\nv_fma_f32 %9, %9, %9, %9\nv_fma_f32 %10, %10, %10, %10\nv_fma_f32 %11, %11, %11, %11\nv_fma_f32 %12, %12, %12, %12\nv_fma_f32 %13, %13, %13, %13\nv_fma_f32 %14, %14, %14, %14\nv_fma_f32 %15, %15, %15, %15\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc));
    if (OP == 14) asm volatile("v_accvgpr_write_b32 a0, %8\nv_accvgpr_write_b32 a1, %9\nv_accvgpr_write_b32 a2, %10\nv_accvgpr_write_b32 a3, %11\nv_accvgpr_write_b32 a4, %12\nv_accvgpr_write_b32 a5, %13\nv_accvgpr_write_b32 a6, %14\nv_accvgpr_write_b32 a7, %15\n"
               : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                 "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(c), "s"(sc) : "a0","a1","a2","a3","a4","a5","a6","a7");
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += d[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 1024 * sizeof(double))); CK(hipMalloc(&cyc, sizeof(long long)));
  const int iters = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    long long best = 1LL << 60;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
      CK(hipDeviceSynchronize());
      long long h; CK(hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost));
      if (h < best) best = h;
    }
    printf("%-16s %3d threads (%d wave/SIMD): %6.2f ticks per instruction and wave\n", name, threads, threads <= 256 ? 1 : threads / 256, (double)best / (iters * 8.0));
  }
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  run<0>("v_fma_f64"); run<1>("v_mul_f64"); run<2>("v_add_f64"); run<3>("v_max_f64 (sgpr)"); run<4>("v_rndne_f64"); run<5>("v_ldexp_f64");
  run<6>("v_mov_b64"); run<7>("v_cvt_f32_f64"); run<8>("v_cvt_f64_f32"); run<9>("v_rsq_f32"); run<10>("v_cvt_i32_f64"); run<11>("v_and_b32");
  run<12>("v_lshl_add_u32"); run<13>("v_fma_f32"); run<14>("v_accvgpr_write");
  return 0;
}
