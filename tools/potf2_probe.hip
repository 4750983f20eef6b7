// Standalone timing of potf2_inv_block (panel-wave diagonal-block factorisation) on one wave,
// alone on a CU and next to a wave streaming MFMAs.  Build: hipcc --offload-arch=gfx950 -O3
// -mllvm -amdgpu-mfma-vgpr-form=1 tools/potf2_probe.hip -o tools/potf2_probe
#include "../scalable-meta-learning-with-gaussian-processes_amd/csrc/gp_fit_fused.hip"
#include <cstdio>
#include <vector>
using namespace scaml;
__global__ void probe(double* out, long long* cyc, int reps, int noisy) {
  __shared__ double panel[64 * PP + 2 * 16 * PP + 256];
  double* Wk = panel + 64 * PP; double* LT = Wk + 16 * PP; double* trash = LT + 16 * PP;   // trash: [80] dump slots + [16] pivot log
  int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * PP; i += blockDim.x) { int r = i / PP, c = i % PP; panel[i] = (r == c ? 20.0 : 0.0) + 1.0 / (1 + r + c); }
  __syncthreads();
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    long long t0 = __builtin_amdgcn_s_memtime();
    int bad = 0;
    for (int r = 0; r < reps; ++r) {
      // refill the block so every repetition factors the same SPD matrix
      for (int i = lane; i < 16 * PP; i += 64) { int rr = i / PP, c = i % PP; panel[i] = (rr == c ? 20.0 : 0.0) + 1.0 / (1 + rr + c); }
      d4_t a;
      for (int g = 0; g < 4; ++g) a[g] = panel[((lane >> 4) + 4 * g) * PP + (lane & 15)];
      bad |= potf2_inv_block(a, LT, Wk, trash, 0, lane);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cyc[0] = (t1 - t0) / reps; out[0] = LT[3 * PP + 1] + Wk[2] + bad; }
  } else if (noisy) {
    // a co-resident wave streaming dependent MFMAs (like an update wave in U2)
    d4_t acc = {0, 0, 0, 0}; double a = 1.0 + lane * 1e-3, b = 0.5;
    for (int r = 0; r < reps * 120; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    out[tid] = acc[0];
  }
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
  for (int noisy = 0; noisy < 2; ++noisy) for (int threads : {64, 512}) {
    probe<<<1, threads>>>(out, cyc, 20, noisy); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("threads=%d noisy=%d: %lld cycles per 16x16 potf2+inverse (incl. %d-double refill)\n", threads, noisy, c, 16 * PP);
  }
  return 0;
}
