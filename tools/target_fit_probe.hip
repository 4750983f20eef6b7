// Stand-alone probe of csrc/gp_target_fit.hip: one objective + gradient evaluation with phase stamps (build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSCAML_TF_STAMPS -I scalable-meta-learning-with-gaussian-processes_amd/csrc tools/target_fit_probe.hip -o tools/target_fit_probe
// run on the GPU box: tools/target_fit_probe [n T D kind threads]).  Synthetic SPD inputs; prints microseconds per phase.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "gp_target_fit.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 80, T = argc > 2 ? atoi(argv[2]) : 32, D = argc > 3 ? atoi(argv[3]) : 6, kind = argc > 4 ? atoi(argv[4]) : 1;
  const int threads = argc > 5 ? atoi(argv[5]) : 512;
  const int mfma = argc > 6 ? atoi(argv[6]) : 1;
  const int P = D + 2 + T, E = n * (n + 1) / 2, B = 1;
  std::vector<double> means((size_t)T * n), covs((size_t)T * E), X((size_t)n * D), y(n), z(P);
  srand(1);
  auto rnd = [] { return rand() / (double)RAND_MAX; };
  for (auto& v : X) v = rnd();
  for (auto& v : y) v = rnd() - 0.5;
  for (auto& v : means) v = rnd() - 0.5;
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < n; ++a)
      for (int b = 0; b <= a; ++b) covs[(size_t)t * E + a * (a + 1) / 2 + b] = a == b ? 0.5 + rnd() : 0.01 * (rnd() - 0.5);
  for (int i = 0; i < D + 2; ++i) z[i] = i < D ? -4.0 : (i == D ? -6.0 : 0.0);
  for (int i = 0; i < T; ++i) z[D + 2 + i] = 0.1;
  double *dm, *dc, *dX, *dy, *dz, *dval, *dgrad, *djit;
  int32_t* dinfo;
  long long* dst;
  CK(hipMalloc(&dm, means.size() * 8)); CK(hipMalloc(&dc, covs.size() * 8)); CK(hipMalloc(&dX, X.size() * 8)); CK(hipMalloc(&dy, n * 8));
  CK(hipMalloc(&dz, P * 8)); CK(hipMalloc(&dval, 8)); CK(hipMalloc(&dgrad, P * 8)); CK(hipMalloc(&djit, 8)); CK(hipMalloc(&dinfo, 4));
  CK(hipMalloc(&dst, 16 * 8));
  CK(hipMemcpy(dm, means.data(), means.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, covs.data(), covs.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dz, z.data(), P * 8, hipMemcpyHostToDevice));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tf_stamps), &dst, sizeof(dst)));
  scaml::TargetFitParams p{};
  p.means_t = dm; p.covs_p = dc; p.X = dX; p.y = dy; p.m_all = 0.1; p.s_all = 1.3;
  p.spec.ls_lo = 1e-4; p.spec.ls_hi = 1e2; p.spec.os_lo = 1e-4; p.spec.os_hi = 1e2; p.spec.nz_lo = 1e-8; p.spec.nz_hi = 1e-2;
  p.spec.ls_prior = {2, 0, 0.5, 1.5, 0.0}; p.spec.os_prior = {2, 0, -2.0, 3.0, 0.0}; p.spec.nz_prior = {2, 0, -8.0, 2.0, 0.0}; p.spec.w_prior = {1, 0, 1.0, 1.0, 0.0};
  p.spec.w_lower = 1e-10;
  p.z = dz; p.value = dval; p.grad = dgrad; p.info = dinfo; p.jitter = djit; p.B = B; p.n = n; p.T = T; p.D = D; p.kind = kind; p.mode = 0; p.history = 1;
  const size_t nw = threads / 64;
  const size_t nbk = (size_t)(n + 15) / 16;
  const size_t mats = mfma ? 2 * (nbk * (nbk + 1) / 2) * 16 * 17 : (size_t)(n + 1) * (n + 2) / 2 + (size_t)n * (n + 1) / 2;
  const size_t lds = (mats + (size_t)n * D + 2 * (n + 1) + 4 * n + 16 + 2 * T + 2 * (D + 2) + D + nw * (scaml::TARGET_FIT_DMAX + 2) + nw + 8 +
                      2 * scaml::TARGET_FIT_HMAX) * 8;
  p.use_mfma = mfma;
  CK(hipFuncSetAttribute((const void*)scaml::scaml_target_fit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(scaml::scaml_target_fit_kernel, dim3(B), dim3(threads), lds, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long st[16];
    double val;
    int32_t info;
    CK(hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost)); CK(hipMemcpy(&val, dval, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost));
    const char* names[] = {"build", "eliminate", "scalars", "scale U + alpha", "K^-1 -> G", "grad w", "grad theta"};
    printf("n=%d T=%d D=%d threads=%d mfma=%d: launch %.1f us, value %.6f info %d |", n, T, D, threads, mfma, ms * 1e3, val, info);
    for (int i = 0; i < 7; ++i) printf(" %s %.1f", names[i], (st[i + 1] - st[i]) * 0.01);
    printf("\n");
  }
  return 0;
}
