// Probe: v_mfma_f64_16x16x4_f64 operand/result lane maps and issue rate on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__global__ void layout_kernel(const double* A, const double* B, double* C) {
  // A: 16x4 row-major, B: 4x16 row-major, C: 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ void rate_kernel(double* out, long long* cyc, int iters) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-3, b = 0.5 - l * 1e-3;
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// fp64 VALU fma throughput and dependent latency
__global__ void fma_kernel(double* out, long long* cyc, int iters, int dep) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double m = 1.0000001, c = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  if (dep) {
    for (int it = 0; it < iters; ++it) {
      x0 = fma(x0, m, c); x0 = fma(x0, m, c); x0 = fma(x0, m, c); x0 = fma(x0, m, c);
      x0 = fma(x0, m, c); x0 = fma(x0, m, c); x0 = fma(x0, m, c); x0 = fma(x0, m, c);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      x0 = fma(x0, m, c); x1 = fma(x1, m, c); x2 = fma(x2, m, c); x3 = fma(x3, m, c);
      x4 = fma(x4, m, c); x5 = fma(x5, m, c); x6 = fma(x6, m, c); x7 = fma(x7, m, c);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// latency of dependent sqrt / div / exp chains in fp64
__global__ void lat_kernel(double* out, long long* cyc, int iters, int which) {
  double x = 2.0 + threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (which == 0) x = sqrt(x) + 1.5;
    else if (which == 1) x = 1.0 / x + 1.5;
    else if (which == 2) x = exp(-x) + 1.5;
    else if (which == 3) x = rsqrt(x) + 1.5;
    else x = __builtin_amdgcn_rsq(x) + 1.5;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// throughput of independent exp / sqrt (8 independent per thread)
__global__ void thr_kernel(double* out, long long* cyc, int iters, int which) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = 0.1 * i + threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (which == 0) x[i] = sqrt(x[i] + 1.0);
      else if (which == 1) x[i] = 1.0 / (x[i] + 1.0);
      else x[i] = exp(-x[i]);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  // ---- layout check with asymmetric integer data
  std::vector<double> A(64), B(64), C(256), R(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 5 + j * 11 + (j * j) % 7;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
  double *dA, *dB, *dC; CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) if (C[i] != R[i]) ++bad;
  printf("layout check: %d mismatches of 256 (A[l&15][l>>4], B[l>>4][l&15], C row=(l>>4)+4r col=l&15)\n", bad);

  double* dout; long long* dcyc; CK(hipMalloc(&dout, 8 * 1024 * 1024)); CK(hipMalloc(&dcyc, 8));
  long long cyc; int iters = 2000;
  auto run = [&](auto kern, int blocks, int threads, const char* name, double per_iter, auto... args) {
    kern<<<blocks, threads>>>(dout, dcyc, iters, args...); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0)); kern<<<blocks, threads>>>(dout, dcyc, iters, args...); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s blocks=%4d thr=%4d  ticks/op=%8.2f  wall_ms=%.3f  ns/op=%.2f\n", name, blocks, threads, (double)cyc / (iters * per_iter), ms, ms * 1e6 / (iters * per_iter));
  };
  // s_memtime ticks at 100MHz constant clock on gfx9 (not shader cycles) -> report wall too
  run(rate_kernel<1>, 1, 64, "mfma f64 16x16x4, 1 wave, 1 acc (dep)", 1);
  run(rate_kernel<2>, 1, 64, "mfma f64 16x16x4, 1 wave, 2 acc", 2);
  run(rate_kernel<4>, 1, 64, "mfma f64 16x16x4, 1 wave, 4 acc", 4);
  run(rate_kernel<8>, 1, 64, "mfma f64 16x16x4, 1 wave, 8 acc", 8);
  run(rate_kernel<4>, 1, 256, "mfma f64, 4 waves (1/SIMD), 4 acc", 4);
  run(rate_kernel<4>, 1, 512, "mfma f64, 8 waves (2/SIMD), 4 acc", 4);
  run(rate_kernel<4>, 256, 256, "mfma f64, 256 blk x 4 waves, 4 acc", 4);
  run(rate_kernel<4>, 256, 512, "mfma f64, 256 blk x 8 waves, 4 acc", 4);
  run(fma_kernel, 1, 64, "v_fma_f64 indep x8, 1 wave", 8, 0);
  run(fma_kernel, 1, 64, "v_fma_f64 dependent, 1 wave", 8, 1);
  run(fma_kernel, 1, 256, "v_fma_f64 indep x8, 4 waves", 8, 0);
  run(fma_kernel, 1, 512, "v_fma_f64 indep x8, 8 waves", 8, 0);
  run(fma_kernel, 256, 1024, "v_fma_f64 indep x8, 256 blk x 16 waves", 8, 0);
  run(lat_kernel, 1, 64, "sqrt f64 dependent (+add)", 1, 0);
  run(lat_kernel, 1, 64, "div f64 dependent (+add)", 1, 1);
  run(lat_kernel, 1, 64, "exp f64 dependent (+add)", 1, 2);
  run(lat_kernel, 1, 64, "rsqrt() f64 dependent (+add)", 1, 3);
  run(lat_kernel, 1, 64, "v_rsq_f64 raw dependent (+add)", 1, 4);
  run(thr_kernel, 1, 256, "sqrt f64 indep x8, 4 waves", 8, 0);
  run(thr_kernel, 1, 256, "div f64 indep x8, 4 waves", 8, 1);
  run(thr_kernel, 1, 256, "exp f64 indep x8, 4 waves", 8, 2);
  run(thr_kernel, 1, 1024, "exp f64 indep x8, 16 waves", 8, 2);
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s CUs=%d clock=%d kHz wallclock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate, 0);
  return bad != 0;
}
