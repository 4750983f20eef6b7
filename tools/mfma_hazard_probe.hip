// Probe: how many wait states a VALU read needs behind v_mfma_f64_16x16x4_f64 on gfx950 -- is the RAW hazard
// interlocked by the hardware or software-managed?  For K = 0 .. 20 wait states between the (last of CHAIN chained)
// MFMA and a v_mov_b64 of result register pair G, report whether the copy saw the new value.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_hazard_probe.hip -o tools/mfma_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

#define NOPS_0  ""
#define NOPS_1  "s_nop 0\n"
#define NOPS_2  "s_nop 1\n"
#define NOPS_3  "s_nop 2\n"
#define NOPS_4  "s_nop 3\n"
#define NOPS_5  "s_nop 4\n"
#define NOPS_6  "s_nop 5\n"
#define NOPS_7  "s_nop 6\n"
#define NOPS_8  "s_nop 7\n"
#define NOPS_9  "s_nop 8\n"
#define NOPS_10 "s_nop 9\n"
#define NOPS_11 "s_nop 10\n"
#define NOPS_12 "s_nop 11\n"
#define NOPS_14 "s_nop 13\n"
#define NOPS_16 "s_nop 15\n"
#define NOPS_18 "s_nop 15\ns_nop 1\n"
#define NOPS_20 "s_nop 15\ns_nop 3\n"
#define MF "v_mfma_f64_16x16x4_f64 v[20:27], %1, %2, v[20:27]\n"
#define BODY(CHAINSTR, NOPSTR, SRC)                                                               \
  asm volatile("v_mov_b64 v[20:21], 0\nv_mov_b64 v[22:23], 0\nv_mov_b64 v[24:25], 0\nv_mov_b64 v[26:27], 0\n" \
               "s_nop 15\ns_nop 7\n" CHAINSTR NOPSTR "v_mov_b64 %0, " SRC "\ns_nop 15\ns_nop 7\n"      \
               : "=v"(r) : "v"(a), "v"(b) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27")

template <int K, int G, int CHAIN>
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  double a = 1.0 + (l & 15), b = 2.0 + (l >> 4), r;
#define DO(KK)                                                                                     \
  if (K == KK) {                                                                                   \
    if (CHAIN == 1) { if (G == 0) BODY(MF, NOPS_##KK, "v[20:21]"); else if (G == 1) BODY(MF, NOPS_##KK, "v[22:23]"); else if (G == 2) BODY(MF, NOPS_##KK, "v[24:25]"); else BODY(MF, NOPS_##KK, "v[26:27]"); }           \
    else            { if (G == 0) BODY(MF MF MF MF, NOPS_##KK, "v[20:21]"); else if (G == 1) BODY(MF MF MF MF, NOPS_##KK, "v[22:23]"); else if (G == 2) BODY(MF MF MF MF, NOPS_##KK, "v[24:25]"); else BODY(MF MF MF MF, NOPS_##KK, "v[26:27]"); } \
  }
  DO(0) DO(1) DO(2) DO(3) DO(4) DO(5) DO(6) DO(7) DO(8) DO(9) DO(10) DO(11) DO(12) DO(14) DO(16) DO(18) DO(20)
  out[l] = r;
}


// the pattern hipcc produced behind the TRSM macro of the N <= 32 kernel: B operand in AGPRs, first MFMA starts from
// the inline constant 0, then s_nop 0 and three register-pair copies (the last one reads the last result pair)
__global__ void probe_site(double* out) {
  const int l = threadIdx.x;
  double a = 1.0 + (l & 15), b = 2.0 + (l >> 4), r1, r2, r3;
  asm volatile("v_accvgpr_write_b32 a8, %3\nv_accvgpr_write_b32 a9, %4\nv_accvgpr_write_b32 a10, %3\nv_accvgpr_write_b32 a11, %4\n"
               "v_accvgpr_write_b32 a12, %3\nv_accvgpr_write_b32 a13, %4\nv_accvgpr_write_b32 a14, %3\nv_accvgpr_write_b32 a15, %4\n"
               "v_mov_b64 v[20:21], 0\nv_mov_b64 v[22:23], 0\nv_mov_b64 v[24:25], 0\nv_mov_b64 v[26:27], 0\ns_nop 15\ns_nop 7\n"
               "v_mfma_f64_16x16x4_f64 v[20:27], %5, a[8:9], 0\nv_mfma_f64_16x16x4_f64 v[20:27], %5, a[10:11], v[20:27]\n"
               "v_mfma_f64_16x16x4_f64 v[20:27], %5, a[12:13], v[20:27]\nv_mfma_f64_16x16x4_f64 v[20:27], %5, a[14:15], v[20:27]\n"
               "s_nop 0\nv_mov_b64 %0, v[22:23]\nv_mov_b64 %1, v[24:25]\nv_mov_b64 %2, v[26:27]\ns_nop 15\ns_nop 7\n"
               : "=&v"(r1), "=&v"(r2), "=&v"(r3)
               : "v"(__double2loint(b)), "v"(__double2hiint(b)), "v"(a)
               : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
  out[l] = r1; out[64 + l] = r2; out[128 + l] = r3;
}

template <int K, int G, int CHAIN>
bool run() {
  double* out; CK(hipMalloc(&out, 64 * sizeof(double)));
  CK(hipMemset(out, 0, 64 * sizeof(double)));
  hipLaunchKernelGGL((probe<K, G, CHAIN>), dim3(1), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  double h[64]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  bool ok = true;
  for (int l = 0; l < 64; ++l) {
    // D[row = (l >> 4) + 4 g][col = l & 15] = sum_k A[row][k] B[k][col]; A lane (r, k) = 1 + r, B lane (k, c) = 2 + k
    const int row = (l >> 4) + 4 * G;
    double ref = 0; for (int k = 0; k < 4; ++k) ref += (1.0 + row) * (2.0 + k);
    ref *= CHAIN;
    if (h[l] != ref) ok = false;
  }
  CK(hipFree(out));
  return ok;
}
template <int G, int CHAIN>
void sweep() {
  printf("chain of %d MFMA(s), reading result pair %d after K wait states:\n  K:", CHAIN, G);
  bool r[] = {run<0, G, CHAIN>(), run<1, G, CHAIN>(), run<2, G, CHAIN>(), run<3, G, CHAIN>(), run<4, G, CHAIN>(), run<5, G, CHAIN>(), run<6, G, CHAIN>(),
              run<7, G, CHAIN>(), run<8, G, CHAIN>(), run<9, G, CHAIN>(), run<10, G, CHAIN>(), run<11, G, CHAIN>(), run<12, G, CHAIN>(), run<14, G, CHAIN>(),
              run<16, G, CHAIN>(), run<18, G, CHAIN>(), run<20, G, CHAIN>()};
  const int ks[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 18, 20};
  for (int i = 0; i < 17; ++i) printf(" %d:%s", ks[i], r[i] ? "ok" : "STALE");
  printf("\n");
}
int main() {
  sweep<0, 1>(); sweep<1, 1>(); sweep<2, 1>(); sweep<3, 1>(); sweep<0, 4>(); sweep<3, 4>();
  {
    double* out; CK(hipMalloc(&out, 192 * sizeof(double)));
    hipLaunchKernelGGL(probe_site, dim3(1), dim3(64), 0, 0, out);
    CK(hipDeviceSynchronize());
    double h[192]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("compiler pattern (4 MFMAs, B in AGPRs; s_nop 0; copies of pairs 1, 2, 3):");
    for (int g = 1; g < 4; ++g) {
      bool ok = true;
      for (int l = 0; l < 64; ++l) ok &= h[64 * (g - 1) + l] == 4 * 14.0 * (1.0 + (l >> 4) + 4 * g);
      printf(" pair %d %s", g, ok ? "ok" : "STALE");
    }
    printf("\n");
  }
  return 0;
}
