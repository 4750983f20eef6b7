// scaml_host.cpp — host side of libscaml_hip.so: the C ABI of include/scaml_gp.h on top of the
// gfx950 code object built from csrc/*.hip.
//
// The device code is compiled separately (see __graft_entry__.build): hipcc emits device LLVM IR,
// csrc/patch_ir.py adds the function attributes clang cannot express ("amdgpu-agpr-alloc"="0":
// the compiler may not touch the AGPR half of the register file, which the kernels manage by
// hand; per-kernel "amdgpu-num-vgpr"), clang lowers it to a code object that is embedded below
// and loaded through the HIP module API on first use.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

#include "../../include/scaml_gp.h"
#include "gp_fit_params.h"
#include "gp_posterior_params.h"
#include "gp_target_params.h"
#include <math.h>

extern "C" const unsigned char scaml_hsaco_blob[];   // generated: lib/hsaco_blob.c
extern "C" const unsigned long scaml_hsaco_blob_len;

namespace {

thread_local char g_last_error[256] = "";
bool g_no_grad_split = getenv("SCAML_GRAD_NO_SPLIT") != nullptr;              // developer A/B switch
int g_grad_path = getenv("SCAML_GRAD_LEGACY") ? 1 : (getenv("SCAML_GRAD_FUSED") ? 2 : 0);   // 0 by shape, 1 two launches, 2 single launch (scaml_debug_force_two_launch_grad)

void set_error(const char* what, hipError_t e) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}

constexpr int PP = 17;  // LDS panel pitch, must match csrc/gp_fit_fused.hip

struct FitVariant {
  int nb, wu;
  hipFunction_t fn[2][2];  // [kind][dense | blocked addressing]
};

struct Module {
  std::mutex mu;
  bool loaded = false;
  hipModule_t mod = nullptr;
  FitVariant fit[5] = {{2, 1, {}}, {4, 3, {}}, {8, 3, {}}, {16, 7, {}}, {8, 7, {}}};   // [4]: wide variant for 64 < N <= 128
  int num_cus = 0;
  hipFunction_t post[2] = {nullptr, nullptr};
  hipFunction_t post_cov[2] = {nullptr, nullptr};
  hipFunction_t post_linv[2] = {nullptr, nullptr};
  hipFunction_t post_linv_cov[2] = {nullptr, nullptr};
  hipFunction_t post_linv_grad[2] = {nullptr, nullptr};
  hipFunction_t wsum = nullptr;
  hipFunction_t linv = nullptr;
  hipFunction_t kmat[2] = {nullptr, nullptr};
  hipFunction_t chosolve = nullptr;
  hipFunction_t mllgrad[2] = {nullptr, nullptr};
  hipFunction_t tgt_assemble[2] = {nullptr, nullptr};
  hipFunction_t tgt_finish = nullptr;
  hipFunction_t tgt_fit = nullptr;
  hipFunction_t tgt_grad[2] = {nullptr, nullptr};
  hipFunction_t blk_round = nullptr, blk_finish = nullptr;
  hipFunction_t coop[2] = {nullptr, nullptr};
  hipFunction_t blk_solve[2][2] = {}, blk_syrk[2] = {nullptr, nullptr};   // solve: [kind][D <= 8]
  hipFunction_t mllgrad_fused[4][2][2] = {};   // [size class NBT = 2, 4, 8, 16][kind][LDS-DMA staging]
  hipFunction_t mllgrad_split[2][2][2] = {};   // LDS-DMA staging, [N <= 128 | N <= 256 class][2 | 4 workgroups per task][kind]
  hipError_t load() {
    std::lock_guard<std::mutex> lk(mu);
    if (loaded) return hipSuccess;
    hipError_t e = hipModuleLoadData(&mod, scaml_hsaco_blob);
    if (e != hipSuccess) return e;
    for (auto& v : fit) {
      for (int kb = 0; kb < 4; ++kb) {
        const int kind = kb >> 1, blk = kb & 1;
        char name[128];
        if (blk) snprintf(name, sizeof(name), "_ZN5scaml21gp_fit_blocked_kernelILi%dELi%dELi%dEEEvNS_9FitParamsENS_14FitBlockParamsE", v.nb, v.wu, kind);
        else snprintf(name, sizeof(name), "_ZN5scaml19gp_fit_fused_kernelILi%dELi%dELi%dEEEvNS_9FitParamsE", v.nb, v.wu, kind);
        e = hipModuleGetFunction(&v.fn[kind][blk], mod, name);
        if (e != hipSuccess) return e;
        // the kernels use up to the full 160 KiB of LDS
        e = hipFuncSetAttribute((const void*)v.fn[kind][blk], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
      }
    }
    for (int kind = 0; kind < 2; ++kind) {
      char name[128];
      snprintf(name, sizeof(name), "_ZN5scaml19gp_posterior_kernelILi%dEEEvNS_15PosteriorParamsE", kind);
      if ((e = hipModuleGetFunction(&post[kind], mod, name)) != hipSuccess) return e;
      if ((e = hipFuncSetAttribute((const void*)post[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
      snprintf(name, sizeof(name), "_ZN5scaml23gp_posterior_cov_kernelILi%dEEEvNS_18PosteriorCovParamsE", kind);
      if ((e = hipModuleGetFunction(&post_cov[kind], mod, name)) != hipSuccess) return e;
      snprintf(name, sizeof(name), "_ZN5scaml24gp_posterior_linv_kernelILi%dELb0ELb0EEEvNS_15PosteriorParamsE", kind);
      if ((e = hipModuleGetFunction(&post_linv[kind], mod, name)) != hipSuccess) return e;
      if ((e = hipFuncSetAttribute((const void*)post_linv[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
      snprintf(name, sizeof(name), "_ZN5scaml24gp_posterior_linv_kernelILi%dELb1ELb0EEEvNS_15PosteriorParamsE", kind);
      if ((e = hipModuleGetFunction(&post_linv_cov[kind], mod, name)) != hipSuccess) return e;
      if ((e = hipFuncSetAttribute((const void*)post_linv_cov[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
      snprintf(name, sizeof(name), "_ZN5scaml24gp_posterior_linv_kernelILi%dELb1ELb1EEEvNS_15PosteriorParamsE", kind);
      if ((e = hipModuleGetFunction(&post_linv_grad[kind], mod, name)) != hipSuccess) return e;
      if ((e = hipFuncSetAttribute((const void*)post_linv_grad[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    }
    if ((e = hipModuleGetFunction(&wsum, mod, "scaml_weighted_task_sum_kernel")) != hipSuccess) return e;
    if ((e = hipModuleGetFunction(&linv, mod, "_ZN5scaml14gp_linv_kernelENS_10LinvParamsE")) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)linv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipModuleGetFunction(&chosolve, mod, "_ZN5scaml19gp_cho_solve_kernelENS_14ChoSolveParamsE")) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)chosolve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    for (int kind = 0; kind < 2; ++kind) {
      char name[128];
      snprintf(name, sizeof(name), "_ZN5scaml18gp_mll_grad_kernelILi%dEEEvNS_13MllGradParamsE", kind);
      if ((e = hipModuleGetFunction(&mllgrad[kind], mod, name)) != hipSuccess) return e;
      // (the kernel also has 1 KB of static LDS)
      if ((e = hipFuncSetAttribute((const void*)mllgrad[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048)) != hipSuccess) return e;
      snprintf(name, sizeof(name), "_Z23gp_kernel_matrix_kernelILi%dEEvN5scaml18KernelMatrixParamsE", kind);
      if ((e = hipModuleGetFunction(&kmat[kind], mod, name)) != hipSuccess) return e;
    }
    for (int sc = 0; sc < 4; ++sc) {
      for (int kind = 0; kind < 2; ++kind) {
        for (int dma = 0; dma < 2; ++dma) {
          char name[128];
          snprintf(name, sizeof(name), "_ZN5scaml24gp_mll_grad_fused_kernelILi%dELi%dELb%dELi1EEEvNS_18MllGradFusedParamsE", 2 << sc, kind, dma);
          if ((e = hipModuleGetFunction(&mllgrad_fused[sc][kind][dma], mod, name)) != hipSuccess) return e;
          if ((e = hipFuncSetAttribute((const void*)mllgrad_fused[sc][kind][dma], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        }
      }
    }
    for (int kind = 0; kind < 2; ++kind) {
      char name[128];
      snprintf(name, sizeof(name), "_Z28scaml_target_assemble_kernelILi%dEEvN5scaml20TargetAssembleParamsE", kind);
      if ((e = hipModuleGetFunction(&tgt_assemble[kind], mod, name)) != hipSuccess) return e;
    }
    for (int kind = 0; kind < 2; ++kind) {
      char name[128];
      snprintf(name, sizeof(name), "_Z24scaml_target_grad_kernelILi%dEEvPKdS1_S1_S1_S1_S1_S1_S1_dPKiiiiPdS4_", kind);
      if ((e = hipModuleGetFunction(&tgt_grad[kind], mod, name)) != hipSuccess) return e;
    }
    if ((e = hipModuleGetFunction(&tgt_finish, mod, "scaml_target_finish_kernel")) != hipSuccess) return e;
    if ((e = hipModuleGetFunction(&tgt_fit, mod, "scaml_target_fit_kernel")) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)tgt_fit, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    for (int kind = 0; kind < 2; ++kind) {
      char name[128];
      snprintf(name, sizeof(name), "_ZN5scaml18gp_fit_coop_kernelILi%dEEEvNS_13CoopFitParamsE", kind);
      if ((e = hipModuleGetFunction(&coop[kind], mod, name)) != hipSuccess) return e;
      if ((e = hipFuncSetAttribute((const void*)coop[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    }
    if ((e = hipModuleGetFunction(&blk_round, mod, "scaml_blocked_round_kernel")) != hipSuccess) return e;
    if ((e = hipModuleGetFunction(&blk_finish, mod, "scaml_blocked_finish_kernel")) != hipSuccess) return e;
    for (int kind = 0; kind < 2; ++kind) {
      char name[128];
      for (int sd = 0; sd < 2; ++sd) {
        snprintf(name, sizeof(name), "_ZN5scaml23gp_blocked_solve_kernelILi%dELb%dEEEvNS_16BlockedFitParamsE", kind, sd);
        if ((e = hipModuleGetFunction(&blk_solve[kind][sd], mod, name)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute((const void*)blk_solve[kind][sd], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
      }
      snprintf(name, sizeof(name), "_ZN5scaml22gp_blocked_syrk_kernelILi%dEEEvNS_16BlockedFitParamsE", kind);
      if ((e = hipModuleGetFunction(&blk_syrk[kind], mod, name)) != hipSuccess) return e;
      if ((e = hipFuncSetAttribute((const void*)blk_syrk[kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    }
    for (int cls = 0; cls < 2; ++cls) {
      for (int sp = 0; sp < 2; ++sp) {
        for (int kind = 0; kind < 2; ++kind) {
          char name[128];
          snprintf(name, sizeof(name), "_ZN5scaml24gp_mll_grad_fused_kernelILi%dELi%dELb1ELi%dEEEvNS_18MllGradFusedParamsE", 8 << cls, kind, 2 << sp);
          if ((e = hipModuleGetFunction(&mllgrad_split[cls][sp][kind], mod, name)) != hipSuccess) return e;
          if ((e = hipFuncSetAttribute((const void*)mllgrad_split[cls][sp][kind], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        }
      }
    }
    loaded = true;
    return hipSuccess;
  }
};

// One Module per device ordinal: hipModuleLoadData loads the code object into the CURRENT device's context and the
// hipFunction_t handles are only valid there, so a process that works on several GPUs (cuda:0, then cuda:1) gets one
// lazily loaded copy per device.  Every entry point runs on the current device (hipGetDevice), which therefore has
// to be the device the pointers and the stream belong to.
constexpr int kMaxDevices = 64;
Module& module() {
  static std::mutex mu;
  static Module* mods[kMaxDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (!mods[dev]) mods[dev] = new Module();
  return *mods[dev];
}

size_t fit_lds_bytes(int nb, int wu, int D) {
  const int np = nb * 16;
  size_t regionA = (size_t)3 * np * PP + (size_t)(nb + 2) * 16 * PP + 4 * 256;   // PT[3], WAll[nb], LT[2], DG[2], CR[2]
  const size_t buildA = (size_t)(((D + 3) & ~3) + 4) * np + (nb == 16 ? 24 * 256 : 0);   // staged points (+ tail rows) + the panel wave's tile images (N > 128 only)
  if (buildA > regionA) regionA = buildA;
  // + vectors, trash/exp table, row lists, 1/l, fail flag + 6 nb hand-off counters (ints)
  return (regionA + 3 * np + 160 + (size_t)wu * nb * 4 + D + (D & 1) + 2 + 3 * (size_t)nb) * sizeof(double);
}

}  // namespace

extern "C" {

int scaml_version(void) { return 400; }  // 0.4.0 = 10000 * 0 + 100 * 4 + 0
const char* scaml_last_error(void) { return g_last_error; }
int scaml_fit_max_n(void) { return 256; }

int scaml_fit_max_d(int N) {
  // largest D whose staged point stack fits the 160 KiB LDS next to the vectors
  int np = N <= 32 ? 32 : (N <= 64 ? 64 : (N <= 128 ? 128 : 256));
  int budget = 160 * 1024 / 8 - 3 * np - 160 - 7 * 16 * 4 - 3 * 16 - 4 - (np == 256 ? 24 * 256 : 0);
  int d = ((budget / (np + 1)) & ~3) - 4;   // rows: D rounded up to 4, + 4 tail rows; + 1/l per dimension
  return d > 1024 ? 1024 : d;
}

static int fit_common(scaml::FitParams p, int kind, void* stream, const scaml::FitBlockParams* blk = nullptr) {
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) {
    set_error("loading the gfx950 code object", e);
    return SCAML_E_LAUNCH;
  }
  const int N = p.N;
  int vi = N <= 32 ? 0 : (N <= 64 ? 1 : (N <= 128 ? 2 : 3));
  if (vi == 2) {
    // 64 < N <= 128: four waves per task let two workgroups share a CU; when the stack does not fill the CUs even
    // once that buys nothing, and eight waves on the task's kernel matrix and trailing update are faster
    if (m.num_cus == 0) {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) m.num_cus = cus;
      if (m.num_cus <= 0) m.num_cus = 256;
    }
    if (p.T <= m.num_cus) vi = 4;
  }
  const FitVariant& v = m.fit[vi];
  const size_t lds = fit_lds_bytes(v.nb, v.wu, p.D);
  if (lds > 160 * 1024) return SCAML_E_TOOLARGE;
  struct { scaml::FitParams p; scaml::FitBlockParams b; } args{p, blk ? *blk : scaml::FitBlockParams{}};   // (kernarg layout: both 8-byte aligned)
  size_t psize = blk ? sizeof(args) : sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  e = hipModuleLaunchKernel(v.fn[kind][blk ? 1 : 0], (unsigned)p.T, 1, 1, (unsigned)(v.wu + 1) * 64, 1, 1, (unsigned)lds,
                            (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) {
    set_error("hipModuleLaunchKernel(gp_fit_fused)", e);
    return SCAML_E_LAUNCH;
  }
  return SCAML_OK;
}

int scaml_gp_fit_fused_f64(const double* X, const double* y, const double* theta,
                           const int32_t* n_points, const double* jitter_in,
                           int T, int N, int D, int kind,
                           double* L, double* alpha, double* quad, double* logdet, double* mll,
                           int32_t* info, double* jitter_used, double* Linv_diag, unsigned flags, void* stream) {
  if (T < 0 || N < 1 || D < 1) return SCAML_E_BADARG;
  if (!X || !y || !theta || !info) return SCAML_E_BADARG;
  if ((flags & SCAML_FIT_STORE_L) && !L) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (N > scaml_fit_max_n()) return SCAML_E_TOOLARGE;
  if (D > scaml_fit_max_d(N)) return SCAML_E_TOOLARGE;
  if (T == 0) return SCAML_OK;
  scaml::FitParams p{X, y, theta, n_points, jitter_in, nullptr, L, alpha, quad, logdet, mll, info, jitter_used, Linv_diag, T, N, D, flags};
  return fit_common(p, kind, stream);
}

// ---- (3b) blocked fit, 256 < N <= 512 ---------------------------------------------------------------------
int scaml_fit_blocked_max_n(void) { return 512; }

static size_t blocked_solve_lds_bytes(int D) {
  const size_t dp = D <= 8 ? 9 : (size_t)(D | 1);   // (D <= 8: staged zero-padded to 8 dimensions)
  return (3 * 16 * 258 + 2 * 4 * 256 + 256 * dp + 64 * dp + 256 + 64 + (D <= 8 ? 8 : (size_t)D) + 1) * sizeof(double);
}

int scaml_fit_blocked_max_d(void) {
  static int dmax = 0;
  if (!dmax) {
    int d = 1;
    while (blocked_solve_lds_bytes(d + 1) <= 160 * 1024) ++d;
    const int dfit = scaml_fit_max_d(256);
    dmax = d < dfit ? d : dfit;
  }
  return dmax;
}

namespace {
struct BlockedLayout {
  size_t S, Vimg, r2, q12, jit_cur, jit_ladder, n1, n2, active, info1, info2, total;
};
BlockedLayout blocked_layout(int T, int N) {
  const size_t n2 = (size_t)(N - 256), t = (size_t)T, tp = (t + 1) & ~(size_t)1;   // (int arrays: 8-byte multiples)
  BlockedLayout l{};
  size_t o = 0;
  l.S = o; o += t * n2 * n2 * 8;
  l.Vimg = o; o += t * 256 * 256 * 8;
  l.r2 = o; o += t * (size_t)N * 8;
  l.q12 = o; o += 4 * t * 8;
  l.jit_cur = o; o += t * 8;
  l.jit_ladder = o; o += t * 8;
  l.n1 = o; o += tp * 4;
  l.n2 = o; o += tp * 4;
  l.active = o; o += tp * 4;
  l.info1 = o; o += tp * 4;
  l.info2 = o; o += tp * 4;
  l.total = o;
  return l;
}
}  // namespace

// developer A/B switch: 0 by shape, 1 the 2 x 2 sequence of launches only, 2 the several-CUs-per-task kernel whenever it is launchable
static int g_blocked_fit_path = getenv("SCAML_BLOCKED_FIT_PATH") ? atoi(getenv("SCAML_BLOCKED_FIT_PATH")) : 0;
static int g_blocked_fit_last = 0;   // which one the last call took (1 / 2)
static bool g_coop_far = getenv("SCAML_COOP_FAR") != nullptr;   // developer A/B / tests: write-through payload stores even when a task's workgroups share an XCD
int scaml_debug_coop_far(int on) {
  const int was = g_coop_far ? 1 : 0;
  if (on == 0 || on == 1) g_coop_far = on == 1;
  return was;
}
int scaml_debug_blocked_fit_path(int mode) {
  const int was = g_blocked_fit_path;
  if (mode >= 0 && mode <= 2) g_blocked_fit_path = mode;
  return mode == -1 ? g_blocked_fit_last : was;
}

long long scaml_gp_fit_blocked_workspace_bytes(int T, int N) {
  if (T < 0 || N <= 256 || N > 512) return 0;
  return (long long)blocked_layout(T, N).total;
}

int scaml_gp_fit_blocked_f64(const double* X, const double* y, const double* theta,
                             const int32_t* n_points, const double* jitter_in,
                             int T, int N, int D, int kind,
                             double* L, double* alpha, double* quad, double* logdet, double* mll,
                             int32_t* info, double* jitter_used, double* Linv_diag, unsigned flags,
                             void* workspace, long long workspace_bytes, void* stream) {
  if (T < 0 || N < 1 || D < 1) return SCAML_E_BADARG;
  if (!X || !y || !theta || !info || !L || !alpha || !Linv_diag) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (N <= scaml_fit_max_n() || N > scaml_fit_blocked_max_n() || (N & 15)) return SCAML_E_TOOLARGE;
  if (D > scaml_fit_blocked_max_d()) return SCAML_E_TOOLARGE;
  if (T == 0) return SCAML_OK;
  const BlockedLayout lay = blocked_layout(T, N);
  if (!workspace || workspace_bytes < (long long)lay.total || ((uintptr_t)workspace & 15)) return SCAML_E_BADARG;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  char* ws = (char*)workspace;
  {
    // Several CUs per task (csrc/gp_fit_coop.hip) while the stack leaves CUs idle: P workgroups per task, ALL resident at once
    // (one per CU: the dynamic LDS request is kept above half a CU's), so the launch is only taken when T P <= #CUs.
    if (m.num_cus == 0) {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) m.num_cus = cus;
      if (m.num_cus <= 0) m.num_cus = 256;
    }
    const int nbc = (N + 31) / 32;
    int parts = m.num_cus / T;
    parts = parts > 8 ? 8 : parts;
    parts = parts > nbc ? nbc : parts;
    const size_t lds_coop = (size_t)(64 + 16 + 32 + 32 + 16 + 8 + 6 * 32 * 33 + (size_t)N * (D | 1)) * sizeof(double);
    const size_t flag_bytes = (((size_t)T * 44 * 4) + 15) & ~(size_t)15;
    const size_t need = flag_bytes + (size_t)T * N * 8 + (size_t)T * 64 * 8;
    // by shape (dev_coop_time.py, profiles/r03_notes.md): three or more workgroups per task always pay; two only while a workgroup's eight
    // or fewer block columns leave it time to keep up with the diagonal chain (N <= 320)
    const bool take = g_blocked_fit_path == 2 ? parts >= 1 : (g_blocked_fit_path == 0 && (parts >= 3 || (parts == 2 && nbc <= 10)));
    if (take && D <= 16 && lds_coop <= 160 * 1024 && need <= (size_t)workspace_bytes) {
      hipStream_t st = (hipStream_t)stream;
      if ((e = hipMemsetAsync(ws, 0, flag_bytes, st)) != hipSuccess) { set_error("hipMemsetAsync(coop flags)", e); return SCAML_E_LAUNCH; }
      scaml::CoopFitParams c{X, y, theta, n_points, jitter_in, L, alpha, quad, logdet, mll, info, jitter_used, Linv_diag,
                             (unsigned*)ws, (unsigned*)ws + (size_t)T * 32, (unsigned*)ws + (size_t)T * 36, (double*)(ws + flag_bytes), (double*)(ws + flag_bytes) + (size_t)T * N,
                             T, N, D, flags | (g_coop_far ? 0x80000000u : 0u), parts};
      size_t csize = sizeof(c);
      void* cconfig[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &c, HIP_LAUNCH_PARAM_BUFFER_SIZE, &csize, HIP_LAUNCH_PARAM_END};
      const size_t lds_req = lds_coop > 82 * 1024 ? lds_coop : 82 * 1024;
      const int t8 = (T + 7) / 8 * 8;
      if ((e = hipModuleLaunchKernel(m.coop[kind], (unsigned)(t8 * parts), 1, 1, 512, 1, 1, (unsigned)lds_req, st, nullptr, cconfig)) != hipSuccess) {
        set_error("hipModuleLaunchKernel(gp_fit_coop)", e); return SCAML_E_LAUNCH;
      }
      g_blocked_fit_last = 2;
      return SCAML_OK;
    }
    g_blocked_fit_last = 1;
  }
  const int N1 = 256, N2 = N - N1, NBT = N / 16;
  scaml::BlockedFitParams p{X, y, theta, n_points, jitter_in, L, alpha, quad, logdet, mll, info, jitter_used, Linv_diag,
                            (double*)(ws + lay.S), (double*)(ws + lay.Vimg), (double*)(ws + lay.r2), (double*)(ws + lay.q12), (double*)(ws + lay.jit_cur),
                            (double*)(ws + lay.jit_ladder), (int32_t*)(ws + lay.n1), (int32_t*)(ws + lay.n2), (int32_t*)(ws + lay.active),
                            (int32_t*)(ws + lay.info1), (int32_t*)(ws + lay.info2), T, N, D, flags, 0};
  const unsigned fl = SCAML_FIT_STORE_L | SCAML_FIT_NO_RETRY | (flags & SCAML_FIT_ZERO_UPPER);
  scaml::FitParams f1{X, y, theta, p.n1, p.jit_cur, nullptr, L, alpha, p.q12, p.q12 + T, nullptr, p.info1, nullptr, Linv_diag, T, N1, D, fl | scaml::FIT_FORWARD_ONLY};
  const scaml::FitBlockParams b1{(long long)N * D, N, (long long)N * N, (long long)NBT * 256, p.active, N};
  scaml::FitParams f2{nullptr, p.r2 + N1, nullptr, p.n2, nullptr, p.S, L + (size_t)N1 * N + N1, alpha + N1, p.q12 + 2 * (size_t)T, p.q12 + 3 * (size_t)T,
                      nullptr, p.info2, nullptr, Linv_diag + (size_t)(N1 / 16) * 256, T, N2, 1, fl};
  const scaml::FitBlockParams b2{0, N, (long long)N * N, (long long)NBT * 256, p.active, N};
  const size_t lds_solve = blocked_solve_lds_bytes(D);
  const size_t lds_syrk = (size_t)(8 * 512 + 128 * (D | 1) + 64 + D + 1) * sizeof(double);
  const int nt = (N2 + 63) / 64;
  const int t8 = (T + 7) / 8 * 8;   // tasks ride on grid.x, dealt to the XCDs (bk_task_part)
  const int rounds = (flags & SCAML_FIT_NO_RETRY) ? 1 : 4;
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  hipStream_t st = (hipStream_t)stream;
  for (int r = 0; r < rounds; ++r) {
    p.round = r;
    if ((e = hipModuleLaunchKernel(m.blk_round, (unsigned)((T + 255) / 256), 1, 1, 256, 1, 1, 0, st, nullptr, config)) != hipSuccess) {
      set_error("hipModuleLaunchKernel(blocked_round)", e); return SCAML_E_LAUNCH;
    }
    int rc = fit_common(f1, kind, stream, &b1);
    if (rc != SCAML_OK) return rc;
    if ((e = hipModuleLaunchKernel(m.blk_solve[kind][D <= 8 ? 1 : 0], (unsigned)(t8 * nt), 1, 1, 512, 1, 1, (unsigned)lds_solve, st, nullptr, config)) != hipSuccess) {
      set_error("hipModuleLaunchKernel(gp_blocked_solve)", e); return SCAML_E_LAUNCH;
    }
    if ((e = hipModuleLaunchKernel(m.blk_syrk[kind], (unsigned)(t8 * (nt * (nt + 1) / 2)), 1, 1, 512, 1, 1, (unsigned)lds_syrk, st, nullptr, config)) != hipSuccess) {
      set_error("hipModuleLaunchKernel(gp_blocked_syrk)", e); return SCAML_E_LAUNCH;
    }
    rc = fit_common(f2, SCAML_KIND_RBF, stream, &b2);
    if (rc != SCAML_OK) return rc;
  }
  if ((e = hipModuleLaunchKernel(m.blk_finish, (unsigned)T, 1, 1, 1024, 1, 1, 0, st, nullptr, config)) != hipSuccess) {
    set_error("hipModuleLaunchKernel(blocked_finish)", e); return SCAML_E_LAUNCH;
  }
  return SCAML_OK;
}

// ---- (2) batched jittered Cholesky of given matrices ------------------------------------------------
int scaml_potrf_batched_f64(const double* A, const double* y, const int32_t* n_points, const double* jitter_in,
                            int T, int N, double* L, double* alpha, double* quad, double* logdet,
                            int32_t* info, double* jitter_used, double* Linv_diag, unsigned flags, void* stream) {
  if (T < 0 || N < 1) return SCAML_E_BADARG;
  if (!A || !info) return SCAML_E_BADARG;
  if ((flags & SCAML_FIT_STORE_L) && !L) return SCAML_E_BADARG;
  if (alpha && !y) return SCAML_E_BADARG;
  if (N > scaml_fit_max_n()) return SCAML_E_TOOLARGE;
  if (T == 0) return SCAML_OK;
  scaml::FitParams p{nullptr, y, nullptr, n_points, jitter_in, A, L, alpha, quad, logdet, nullptr, info, jitter_used, Linv_diag, T, N, 1, flags};
  return fit_common(p, SCAML_KIND_RBF, stream);
}

// ---- (5) batched source posteriors ------------------------------------------------------------
static size_t posterior_lds_bytes(int N, int D, int waves, bool x_in_lds) {
  const size_t np = (size_t)((N + 15) / 16) * 16;
  return (64 + np + D + (D & 1) + (x_in_lds ? (size_t)D * np : 0) + (size_t)waves * (16 * D + np * 16)) * sizeof(double);
}

int scaml_posterior_max_n(void) { return 512; }

int scaml_posterior_batched_f64(const double* Xq, const double* X, const double* theta, const double* L,
                                const double* Linv_diag, const double* alpha, const double* y_mean,
                                const double* y_std, const int32_t* n_points, int T, int N, int M, int D, int kind,
                                double* mu, double* var, double* V, unsigned flags, void* stream) {
  if (T < 0 || N < 1 || M < 0 || D < 1) return SCAML_E_BADARG;
  if (!Xq || !X || !theta || !alpha) return SCAML_E_BADARG;
  if (!(flags & SCAML_POST_MEAN_ONLY) && (!L || !Linv_diag)) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (N > scaml_posterior_max_n()) return SCAML_E_TOOLARGE;
  if (T == 0 || M == 0) return SCAML_OK;
  // waves per workgroup / X staging: the largest configuration whose LDS fits 160 KiB
  int waves = 0;
  bool xl = true;
  for (int cand : {4, 2, 1}) {
    if (posterior_lds_bytes(N, D, cand, true) <= 160 * 1024) { waves = cand; xl = true; break; }
    if (posterior_lds_bytes(N, D, cand, false) <= 160 * 1024) { waves = cand; xl = false; break; }
  }
  if (!waves) return SCAML_E_TOOLARGE;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  const int per_task = (flags & SCAML_POST_XQ_PER_TASK) ? 1 : 0, mean_only = (flags & SCAML_POST_MEAN_ONLY) ? 1 : 0;
  if (mean_only && (var || V)) return SCAML_E_BADARG;
  scaml::PosteriorParams p{Xq, X, theta, L, Linv_diag, alpha, y_mean, y_std, n_points, mu, var, V, T, N, M, D, xl ? 1 : 0, per_task, mean_only};
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  const int strips = (M + 15) / 16;
  e = hipModuleLaunchKernel(m.post[kind], (unsigned)((strips + waves - 1) / waves), (unsigned)T, 1, (unsigned)waves * 64, 1, 1,
                            (unsigned)posterior_lds_bytes(N, D, waves, xl), (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_posterior)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

int scaml_posterior_cov_f64(const double* Xq, const double* theta, const double* V, const double* y_std,
                            int T, int N, int M, int Ma, int D, int kind, double* cov, unsigned flags, void* stream) {
  if (T < 0 || N < 1 || M < 0 || Ma < 0 || Ma > M || D < 1) return SCAML_E_BADARG;
  if (!Xq || !theta || !V || !cov) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (T == 0 || M == 0 || Ma == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  scaml::PosteriorCovParams p{Xq, theta, V, y_std, cov, T, N, M, Ma, D, (flags & SCAML_POST_XQ_PER_TASK) ? 1 : 0};
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  const int tiles_c = (M + 15) / 16, tiles_a = (Ma + 15) / 16;
  e = hipModuleLaunchKernel(m.post_cov[kind], (unsigned)((tiles_c + 3) / 4), (unsigned)tiles_a, (unsigned)T, 256, 1, 1, 0,
                            (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_posterior_cov)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

// ---- (6) weighted sum over tasks ---------------------------------------------------------------
int scaml_weighted_task_sum_f64(const double* in, const double* w, const uint8_t* active, int T, long long len,
                                int power, double* out, void* stream) {
  if (T < 0 || len < 0 || (power != 1 && power != 2)) return SCAML_E_BADARG;
  if (!in || !w || !out) return SCAML_E_BADARG;
  if (len == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  void* args[] = {(void*)&in, (void*)&w, (void*)&active, (void*)&T, (void*)&len, (void*)&power, (void*)&out};
  e = hipModuleLaunchKernel(m.wsum, (unsigned)((len + 63) / 64), 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(weighted_task_sum)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

// ---- (1) stand-alone kernel matrix ---------------------------------------------------------------
int scaml_kernel_matrix_f64(const double* X1, const double* X2, const double* theta, int T, int N1, int N2, int D,
                            int kind, int x2_shared, int add_noise, double* K, void* stream) {
  if (T < 0 || N1 < 1 || N2 < 1 || D < 1) return SCAML_E_BADARG;
  if (!X1 || !theta || !K) return SCAML_E_BADARG;
  if (!X2 && N1 != N2) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (T == 0) return SCAML_OK;
  if (N1 > 65535 || T > 65535) return SCAML_E_TOOLARGE;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  scaml::KernelMatrixParams p{X1, X2, theta, K, T, N1, N2, D, x2_shared, add_noise};
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  e = hipModuleLaunchKernel(m.kmat[kind], (unsigned)((N2 + 127) / 128), (unsigned)N1, (unsigned)T, 128, 1, 1, 0, (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_kernel_matrix)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

// ---- batched Cholesky solve ----------------------------------------------------------------------
static int cho_solve_common(const double* L, const double* Linv_diag, const double* B, const int32_t* n_points,
                           int T, int N, int R, double* Xout, int mode, void* stream) {
  if (T < 0 || N < 1 || R < 0) return SCAML_E_BADARG;
  if (!L || !Linv_diag || !B || !Xout) return SCAML_E_BADARG;
  if (N > scaml_posterior_max_n()) return SCAML_E_TOOLARGE;
  if (T == 0 || R == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  const int nb = (N + 15) / 16, np = nb * 16, strips = (R + 15) / 16;
  int waves = (np * 16 * 8 * 4 <= 160 * 1024) ? 4 : ((np * 16 * 8 * 2 <= 160 * 1024) ? 2 : 1);
  if (waves > strips) waves = strips;
  scaml::ChoSolveParams p{L, Linv_diag, B, n_points, Xout, T, N, R, mode};
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  e = hipModuleLaunchKernel(m.chosolve, (unsigned)((strips + waves - 1) / waves), (unsigned)T, 1, (unsigned)waves * 64, 1, 1,
                            (unsigned)((size_t)waves * np * 16 * 8), (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_cho_solve)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

int scaml_cho_solve_batched_f64(const double* L, const double* Linv_diag, const double* B, const int32_t* n_points,
                                int T, int N, int R, double* Xout, void* stream) {
  return cho_solve_common(L, Linv_diag, B, n_points, T, N, R, Xout, 0, stream);
}

int scaml_solve_lt_batched_f64(const double* L, const double* Linv_diag, const double* B, const int32_t* n_points,
                               int T, int N, int R, double* Xout, void* stream) {
  return cho_solve_common(L, Linv_diag, B, n_points, T, N, R, Xout, 1, stream);
}

// the weighted target prior in one call: mean with w, covariance with w^2 (two launches of the sum kernel)
int scaml_weighted_prior_reduce_f64(const double* mu, const double* cov, const double* w, const uint8_t* active, int T, int M,
                                    int Ma, double* mu_s, double* cov_s, void* stream) {
  if (T < 0 || M < 0 || Ma < 0) return SCAML_E_BADARG;
  if (!w || (mu && !mu_s) || (cov && !cov_s) || (!mu && !cov)) return SCAML_E_BADARG;
  int rc = SCAML_OK;
  if (mu) rc = scaml_weighted_task_sum_f64(mu, w, active, T, (long long)M, 1, mu_s, stream);
  if (rc == SCAML_OK && cov) rc = scaml_weighted_task_sum_f64(cov, w, active, T, (long long)Ma * M, 2, cov_s, stream);
  return rc;
}

// ---- explicit inverse factor + posteriors from it -------------------------------------------------
static int launch_linv(Module& m, const double* L, const double* Linv_diag, const int32_t* n_points, int T, int N,
                       double* Linv, void* stream, int lower_only = 0) {
  const int nb = (N + 15) / 16, np = nb * 16;
  const int waves = (np * 16 * 8 * 4 <= 160 * 1024) ? 4 : ((np * 16 * 8 * 2 <= 160 * 1024) ? 2 : 1);
  scaml::LinvParams p{L, Linv_diag, n_points, Linv, T, N, lower_only};
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  // one workgroup per task takes all strips (balanced over its waves); small stacks are split over more
  // workgroups so that the 256 CUs stay busy
  int groups = 1;
  while (groups * 2 * waves <= nb && (long long)T * groups * 2 <= 256) groups *= 2;
  hipError_t e = hipModuleLaunchKernel(m.linv, (unsigned)groups, (unsigned)T, 1, (unsigned)waves * 64, 1, 1,
                                       (unsigned)((size_t)waves * np * 16 * 8), (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_linv)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

int scaml_linv_batched_f64(const double* L, const double* Linv_diag, const int32_t* n_points, int T, int N, double* Linv,
                           void* stream) {
  if (T < 0 || N < 1) return SCAML_E_BADARG;
  if (!L || !Linv_diag || !Linv) return SCAML_E_BADARG;
  if (N > scaml_posterior_max_n()) return SCAML_E_TOOLARGE;
  if (T == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  return launch_linv(m, L, Linv_diag, n_points, T, N, Linv, stream);
}

int scaml_linv_batched_lower_f64(const double* L, const double* Linv_diag, const int32_t* n_points, int T, int N, double* Linv,
                                 void* stream) {
  if (T < 0 || N < 1) return SCAML_E_BADARG;
  if (!L || !Linv_diag || !Linv) return SCAML_E_BADARG;
  if (N > scaml_posterior_max_n()) return SCAML_E_TOOLARGE;
  if (T == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  return launch_linv(m, L, Linv_diag, n_points, T, N, Linv, stream, 1);
}

static int posterior_linv_common(const double* Xq, const double* X, const double* theta, const double* Linv, const double* alpha,
                                 const double* y_mean, const double* y_std, const int32_t* n_points, const double* VA, int T, int N,
                                 int M, int Ma, int D, int kind, double* mu, double* var, double* V, double* cov, unsigned flags,
                                 void* stream, bool grad = false, const double* Xa = nullptr) {
  if (T < 0 || N < 1 || M < 0 || D < 1) return SCAML_E_BADARG;
  if (!Xq || !X || !theta || !Linv || !alpha) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (flags & SCAML_POST_MEAN_ONLY) return SCAML_E_BADARG;   // (use scaml_posterior_batched_f64 for that)
  if (N > scaml_posterior_max_n()) return SCAML_E_TOOLARGE;
  if (VA && (!cov || Ma < 1 || (!grad && Ma > M))) return SCAML_E_BADARG;
  if (VA && (Ma > 96 || Ma > N)) return SCAML_E_TOOLARGE;      // six 16-point strips of leading query points at most
  if (T == 0 || M == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  const int np = ((N + 15) / 16) * 16;
  const int d4 = (D + 3) & ~3;   // query points, 1 / lengthscale zero-padded to the MFMA k-step
  const size_t base = (size_t)(64 + np + d4 + 16 * d4 + 32 + 16 + np + (size_t)np * 16) * sizeof(double);
  const bool xl = false;         // (the task's points are MFMA operands read from memory: nothing to stage)
  const size_t with_x = base;
  if (base > 160 * 1024) return SCAML_E_TOOLARGE;
  scaml::PosteriorParams p{Xq, X, theta, Linv, nullptr, alpha, y_mean, y_std, n_points, mu, var, V, T, N, M, D, xl ? 1 : 0,
                           (flags & SCAML_POST_XQ_PER_TASK) ? 1 : 0, 0, VA, cov, VA ? Ma : 0, 0, Xa};
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  const unsigned strips = (unsigned)((M + 15) / 16);
  const unsigned blocks = (unsigned)(((T + 7) / 8) * 8) * strips;   // XCD-aware (task, strip) map inside the kernel
  e = hipModuleLaunchKernel(grad ? m.post_linv_grad[kind] : (VA ? m.post_linv_cov[kind] : m.post_linv[kind]), blocks, 1, 1, 512, 1, 1,
                            (unsigned)(xl ? with_x : base), (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_posterior_linv)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

int scaml_posterior_linv_f64(const double* Xq, const double* X, const double* theta, const double* Linv, const double* alpha,
                             const double* y_mean, const double* y_std, const int32_t* n_points, int T, int N, int M, int D,
                             int kind, double* mu, double* var, double* V, unsigned flags, void* stream) {
  return posterior_linv_common(Xq, X, theta, Linv, alpha, y_mean, y_std, n_points, nullptr, T, N, M, 0, D, kind, mu, var, V, nullptr,
                               flags, stream);
}

int scaml_posterior_linv_cov_f64(const double* Xq, const double* X, const double* theta, const double* Linv, const double* alpha,
                                 const double* y_mean, const double* y_std, const int32_t* n_points, const double* VA, int T, int N,
                                 int M, int Ma, int D, int kind, double* mu, double* var, double* cov, unsigned flags, void* stream) {
  if (!VA || !cov) return SCAML_E_BADARG;
  return posterior_linv_common(Xq, X, theta, Linv, alpha, y_mean, y_std, n_points, VA, T, N, M, Ma, D, kind, mu, var, nullptr, cov,
                               flags, stream);
}

// (5d) the posterior pass with input gradients: 16 columns per query point = [value, d/dx_0 .. d/dx_{D-1}, zeros]
int scaml_posterior_linv_grad_f64(const double* Xq, const double* Xa, const double* X, const double* theta, const double* Linv,
                                  const double* alpha, const double* y_mean, const double* y_std, const int32_t* n_points,
                                  const double* VA, int T, int N, int Mq, int Ma, int D, int kind, double* mu, double* var, double* cov,
                                  unsigned flags, void* stream) {
  if (Mq < 0 || Ma < 0 || D > 15) return D > 15 ? SCAML_E_TOOLARGE : SCAML_E_BADARG;
  if (Ma > 0 && (!VA || !cov || !Xa)) return SCAML_E_BADARG;
  if (Mq > (1 << 26)) return SCAML_E_TOOLARGE;
  return posterior_linv_common(Xq, X, theta, Linv, alpha, y_mean, y_std, n_points, Ma > 0 ? VA : nullptr, T, N, 16 * Mq, Ma > 0 ? Ma : 0, D, kind,
                               mu, var, nullptr, Ma > 0 ? cov : nullptr, flags, stream, true, Xa);
}

// ---- (4) gradient of the marginal log-likelihood ------------------------------------------------
long long scaml_mll_backward_workspace_doubles(int T, int N, int D) {
  const long long nb = (N + 15) / 16;
  return (long long)T * N * N + (long long)T * (nb * (nb + 1) / 2) * (D + 2);
}

int scaml_mll_backward_f64(const double* X, const double* theta, const double* L, const double* Linv_diag,
                           const double* alpha, const int32_t* n_points, int T, int N, int D, int kind,
                           double* workspace, double* partials_out, void* stream) {
  if (T < 0 || N < 1 || D < 1) return SCAML_E_BADARG;
  if (!X || !theta || !L || !Linv_diag || !alpha || !workspace) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (N > scaml_posterior_max_n()) return SCAML_E_TOOLARGE;
  if ((size_t)4 * 64 * ((D + 1) | 1) * sizeof(double) > 160 * 1024 - 2048) return SCAML_E_TOOLARGE;   // staged points of four waves (D <= 76)
  if (T == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  const int nb = (N + 15) / 16, nt = nb * (nb + 1) / 2;
  double* Linv = workspace;
  double* partials = partials_out ? partials_out : workspace + (size_t)T * N * N;
  // N <= 256, D <= 8: ONE launch, one workgroup per task, K^-1 by column strips held in registers; neither L^-1 nor
  // K^-1 touches memory (csrc/gp_mll_grad_fused.hip).  Larger problems take the two-launch path below.
  // (measured, tools/dev_grad_select.py: a stack of at most 64 tasks of more than 64 points leaves most CUs idle with one
  //  workgroup per task, and there the two launches below -- L^-1 strips and K^-1 tiles spread over the chip -- win: 96 vs 120 us at
  //  T = 32, N = 256; 44 vs 49 us at T = 64, N = 128)
  const bool small_stack = N > 64 && T <= 64;
  if (N <= 256 && D <= 8 && g_grad_path != 1 && !(small_stack && g_grad_path != 2 && !g_no_grad_split)) {
    const int sc = N <= 32 ? 0 : (N <= 64 ? 1 : (N <= 128 ? 2 : 3));
    const int nbt = 2 << sc, np = 16 * nbt, nw = nbt / 2;
    const size_t lds = ((size_t)2 * 16 * (np + 2) + (size_t)2 * np * 9 + 16 + np + (size_t)nw * 512 + 64 + 8 + (size_t)nw * 10) * sizeof(double);
    scaml::MllGradFusedParams p{X, theta, L, Linv_diag, alpha, n_points, partials, T, N, D};
    size_t psize = sizeof(p);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
    // direct-to-LDS staging moves raw 16-byte pieces: only where no element needs masking and every row is 16-byte aligned
    const int dma = (n_points == nullptr && N % 16 == 0 && ((uintptr_t)L % 16) == 0 && ((uintptr_t)Linv_diag % 16) == 0) ? 1 : 0;
    // a stack that leaves CUs idle with one workgroup per task is split over 2 or 4 workgroups per task (strips are
    // independent): BASELINE configs[3] runs 128 tasks per GPU on 256 CUs
    if (m.num_cus == 0) {
      int dev = 0, cus = 0;
      if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) m.num_cus = cus;
      if (m.num_cus <= 0) m.num_cus = 256;
    }
    // (split tasks pay for themselves in the N <= 256 class between 65 and 128 tasks only: 100 vs 105 us at T = 128; in the N <= 128
    //  class every split measured slower than one workgroup per task: 57 vs 49 us at T = 64)
    int split = 1;
    if (sc == 3 && dma && !g_no_grad_split && 2 * T <= m.num_cus) split = 2;
    hipFunction_t fn = split == 1 ? m.mllgrad_fused[sc][kind][dma] : m.mllgrad_split[sc - 2][split == 2 ? 0 : 1][kind];
    e = hipModuleLaunchKernel(fn, (unsigned)T, (unsigned)split, 1, (unsigned)(nbt * 32 / split), 1, 1, (unsigned)lds, (hipStream_t)stream,
                              nullptr, config);
    if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_mll_grad_fused)", e); return SCAML_E_LAUNCH; }
    return SCAML_OK;
  }
  {
    const int rc = launch_linv(m, L, Linv_diag, n_points, T, N, Linv, stream);   // (dense: the 2 x 2 super-tiles of the tile kernel read zero blocks above the diagonal)
    if (rc != SCAML_OK) return rc;
  }
  {
    scaml::MllGradParams p{X, theta, alpha, Linv, n_points, partials, T, N, D};
    size_t psize = sizeof(p);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
    // 1-D grid, XCD-aware (task, tile group) map inside the kernel
    const int nbs = (nb + 1) / 2, ns = nbs * (nbs + 1) / 2;   // 2 x 2 super-tiles, one per wave
    const unsigned blocks = (unsigned)(((T + 7) / 8) * 8) * (unsigned)((ns + 3) / 4);
    e = hipModuleLaunchKernel(m.mllgrad[kind], blocks, 1, 1, 256, 1, 1, (unsigned)(4 * 64 * ((D + 1) | 1) * sizeof(double)), (hipStream_t)stream,
                              nullptr, config);
    if (e != hipSuccess) { set_error("hipModuleLaunchKernel(gp_mll_grad)", e); return SCAML_E_LAUNCH; }
  }
  return SCAML_OK;
}

// ---- target GP: assemble the joint prior block / finish the posterior (a10) -----------------------------------------
int scaml_target_assemble_f64(const double* cov_s, const double* mean_s, const double* var_s, const double* Xall,
                              const double* theta, const double* train_targets, double m_all, double s_all, int n, int M, int D,
                              int kind, double* Knn, double* resid, double* Knq, double* mean_q, double* var_q, void* stream) {
  if (n < 1 || M < 0 || D < 1) return SCAML_E_BADARG;
  if (!cov_s || !mean_s || !var_s || !Xall || !theta || !train_targets || !Knn || !resid) return SCAML_E_BADARG;
  if (M > 0 && (!Knq || !mean_q || !var_q)) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (!(s_all > 0.0)) return SCAML_E_BADARG;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  scaml::TargetAssembleParams p{cov_s, mean_s, var_s, Xall, theta, train_targets, m_all, s_all, Knn, resid, Knq, mean_q, var_q, n, M, D};
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  long long elems = (long long)n * (n + M);
  if (elems < M) elems = M;
  e = hipModuleLaunchKernel(m.tgt_assemble[kind], (unsigned)((elems + 255) / 256), 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(target_assemble)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

int scaml_target_finish_f64(const double* Knq, const double* Z, const double* alpha, const double* mean_q, const double* var_q,
                            double m_all, double s_all, double noise_add, const int32_t* info, int n, int M, double* mu, double* var,
                            void* stream) {
  if (n < 1 || M < 0) return SCAML_E_BADARG;
  if (M == 0) return SCAML_OK;
  if (!Knq || !Z || !alpha || !mean_q || !var_q || !mu || !var) return SCAML_E_BADARG;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  void* args[] = {(void*)&Knq, (void*)&Z, (void*)&alpha, (void*)&mean_q, (void*)&var_q, (void*)&m_all, (void*)&s_all, (void*)&noise_add,
                  (void*)&info, (void*)&n, (void*)&M, (void*)&mu, (void*)&var};
  e = hipModuleLaunchKernel(m.tgt_finish, (unsigned)((M + 127) / 128), 1, 1, 128, 1, 1, 0, (hipStream_t)stream, args, nullptr);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(target_finish)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

int scaml_target_posterior_grad_f64(const double* cov_g, const double* mu_g, const double* var_g, const double* Xt, const double* Xq,
                                    const double* theta, const double* alpha, const double* Z, double s_all, const int32_t* info, int n,
                                    int Mq, int D, int kind, double* dmu, double* dvar, void* stream) {
  if (n < 0 || Mq < 0 || D < 1) return SCAML_E_BADARG;
  if (!mu_g || !var_g || !Xq || !theta || !dmu || !dvar) return SCAML_E_BADARG;
  if (n > 0 && (!cov_g || !Xt || !alpha || !Z)) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (!(s_all > 0.0)) return SCAML_E_BADARG;
  if (Mq == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  void* args[] = {(void*)&cov_g, (void*)&mu_g, (void*)&var_g, (void*)&Xt, (void*)&Xq, (void*)&theta, (void*)&alpha, (void*)&Z, (void*)&s_all,
                  (void*)&info, (void*)&n, (void*)&Mq, (void*)&D, (void*)&dmu, (void*)&dvar};
  if (D > 15) return SCAML_E_TOOLARGE;
  e = hipModuleLaunchKernel(m.tgt_grad[kind], (unsigned)Mq, 1, 1, 64, 1, 1, 0, (hipStream_t)stream, args, nullptr);   // one wave per query point
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(target_grad)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}

// ---- (8) target GP: objective + gradient, and the whole L-BFGS refit, in one launch (csrc/gp_target_fit.hip) -------------
namespace {
constexpr int kTargetFitThreads = 512;
int g_target_fit_path = getenv("SCAML_TARGET_FIT_NO_MFMA") ? 1 : 0;   // developer A/B switch: 0 by shape, 1 column-by-column elimination only
size_t target_fit_lds_doubles(int n, int T, int D, bool mfma) {
  const size_t nw = kTargetFitThreads / 64, nb = (size_t)(n + 15) / 16;
  const size_t mats = mfma ? 2 * (nb * (nb + 1) / 2) * 16 * 17 : (size_t)(n + 1) * (n + 2) / 2 + (size_t)n * (n + 1) / 2;
  return mats + (size_t)n * D + 2 * (size_t)(n + 1) + 4 * (size_t)n + 16 + 2 * (size_t)T +
         2 * (size_t)(D + 2) + D + nw * (scaml::TARGET_FIT_DMAX + 2) + nw + 8 + 2 * scaml::TARGET_FIT_HMAX;
}
// the matrix-core factorisation takes n <= 112 (two block triangles of 16 x 17 tiles in LDS)
bool target_fit_use_mfma(int n, int T, int D) {
  return g_target_fit_path == 0 && n <= 112 && target_fit_lds_doubles(n, T, D, true) * sizeof(double) <= 160 * 1024;
}
int target_spec_from_host(const double* spec, scaml::TargetSpec& sp) {
  sp.ls_lo = spec[0]; sp.ls_hi = spec[1]; sp.os_lo = spec[2]; sp.os_hi = spec[3]; sp.nz_lo = spec[4]; sp.nz_hi = spec[5];
  if (!(sp.ls_hi > sp.ls_lo) || !(sp.os_hi > sp.os_lo) || !(sp.nz_hi > sp.nz_lo)) return SCAML_E_BADARG;
  scaml::TargetPrior* pr[4] = {&sp.ls_prior, &sp.os_prior, &sp.nz_prior, &sp.w_prior};
  for (int q = 0; q < 4; ++q) {
    const int kind = (int)spec[6 + 3 * q];
    const double p1 = spec[7 + 3 * q], p2 = spec[8 + 3 * q];
    if (kind < 0 || kind > 2) return SCAML_E_BADARG;
    if (kind == 1 && !(p1 > 0.0 && p2 > 0.0)) return SCAML_E_BADARG;
    if (kind == 2 && !(p2 > 0.0)) return SCAML_E_BADARG;
    pr[q]->kind = kind; pr[q]->pad_ = 0; pr[q]->p1 = p1; pr[q]->p2 = p2;
    pr[q]->c0 = kind == 1 ? p1 * log(p2) - lgamma(p1) : (kind == 2 ? -log(p2) - 0.9189385332046727 : 0.0);
  }
  sp.w_lower = spec[18];
  return SCAML_OK;
}
int target_fit_launch(scaml::TargetFitParams& p, void* stream) {
  if (p.B < 0 || p.n < 1 || p.T < 1 || p.D < 1) return SCAML_E_BADARG;
  if (!p.means_t || !p.covs_p || !p.X || !p.y || !p.z || !p.value || !p.info) return SCAML_E_BADARG;
  if (p.kind != SCAML_KIND_RBF && p.kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (!(p.s_all > 0.0)) return SCAML_E_BADARG;
  if (p.D > scaml::TARGET_FIT_DMAX) return SCAML_E_TOOLARGE;
  p.use_mfma = target_fit_use_mfma(p.n, p.T, p.D) ? 1 : 0;
  const size_t lds = target_fit_lds_doubles(p.n, p.T, p.D, p.use_mfma != 0) * sizeof(double);
  if (lds > 160 * 1024) return SCAML_E_TOOLARGE;
  if (p.B == 0) return SCAML_OK;
  Module& m = module();
  hipError_t e = m.load();
  if (e != hipSuccess) { set_error("loading the gfx950 code object", e); return SCAML_E_LAUNCH; }
  size_t psize = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
  e = hipModuleLaunchKernel(m.tgt_fit, (unsigned)p.B, 1, 1, kTargetFitThreads, 1, 1, (unsigned)lds, (hipStream_t)stream, nullptr, config);
  if (e != hipSuccess) { set_error("hipModuleLaunchKernel(target_fit)", e); return SCAML_E_LAUNCH; }
  return SCAML_OK;
}
}  // namespace

int scaml_target_fit_max_d(void) { return scaml::TARGET_FIT_DMAX; }

int scaml_target_fit_max_n(int T, int D) {
  if (T < 1 || D < 1 || D > scaml::TARGET_FIT_DMAX) return 0;
  int n = 0;
  while (n < 4096 && target_fit_lds_doubles(n + 1, T, D, false) * sizeof(double) <= 160 * 1024) ++n;
  return n;
}

long long scaml_target_fit_workspace_doubles(int B, int T, int D, int history) {
  if (B < 0 || T < 1 || D < 1 || history < 1) return 0;
  return (long long)B * (6 + 2 * (long long)history) * (D + 2 + T);
}

int scaml_target_mll_f64(const double* means_t, const double* covs_packed, const double* X, const double* y, double m_all, double s_all,
                         const double* spec_host, const double* z, int B, int n, int T, int D, int kind, double* value, double* grad,
                         int32_t* info, double* jitter_used, void* stream) {
  if (!spec_host || !grad) return SCAML_E_BADARG;
  scaml::TargetFitParams p{};
  p.means_t = means_t; p.covs_p = covs_packed; p.X = X; p.y = y; p.m_all = m_all; p.s_all = s_all;
  const int rc = target_spec_from_host(spec_host, p.spec);
  if (rc != SCAML_OK) return rc;
  p.z = const_cast<double*>(z); p.value = value; p.grad = grad; p.info = info; p.jitter = jitter_used;
  p.B = B; p.n = n; p.T = T; p.D = D; p.kind = kind; p.mode = 0; p.history = 1; p.max_ls = 0;
  return target_fit_launch(p, stream);
}

int scaml_target_fit_f64(const double* means_t, const double* covs_packed, const double* X, const double* y, double m_all, double s_all,
                         const double* spec_host, double* z, int B, int n, int T, int D, int kind, int max_iter, int history, double gtol,
                         double ftol, double* value, int32_t* info, double* jitter_used, int32_t* stats, double* workspace,
                         long long workspace_doubles, void* stream) {
  if (!spec_host || !workspace) return SCAML_E_BADARG;
  if (max_iter < 0 || history < 1 || history > scaml::TARGET_FIT_HMAX) return SCAML_E_BADARG;
  if (workspace_doubles < scaml_target_fit_workspace_doubles(B, T, D, history)) return SCAML_E_BADARG;
  scaml::TargetFitParams p{};
  p.means_t = means_t; p.covs_p = covs_packed; p.X = X; p.y = y; p.m_all = m_all; p.s_all = s_all;
  const int rc = target_spec_from_host(spec_host, p.spec);
  if (rc != SCAML_OK) return rc;
  p.z = z; p.value = value; p.grad = nullptr; p.info = info; p.jitter = jitter_used; p.workspace = workspace; p.stats = stats;
  p.B = B; p.n = n; p.T = T; p.D = D; p.kind = kind; p.mode = 1; p.max_iter = max_iter; p.history = history; p.max_ls = 20;
  p.gtol = gtol; p.ftol = ftol;
  return target_fit_launch(p, stream);
}

// Developer switch: 1 keeps scaml_target_mll_f64 / scaml_target_fit_f64 on the column-by-column elimination where the matrix-core
// factorisation would apply (A/B timing, testing one path against the other), 0 restores the choice by shape.  Returns the previous mode.
int scaml_debug_target_fit_path(int mode) {
  const int was = g_target_fit_path;
  g_target_fit_path = mode == 1 ? 1 : 0;
  return was;
}

// Developer switch: 1 routes scaml_mll_backward_f64 through the two-launch path (L^-1 in the workspace, then the K^-1 tile kernel)
// even where the single-launch kernel applies, 2 through the single-launch kernel even for the small stacks the two launches serve
// by default, 0 restores the choice by shape -- for A/B timing and for testing one path against the other.  Returns the previous mode.
int scaml_debug_force_two_launch_grad(int mode) {
  const int was = g_grad_path;
  g_grad_path = mode == 1 ? 1 : (mode == 2 ? 2 : 0);
  return was;
}

// Diagnostic builds only (SCAML_STAMPS): point the device-side stamp buffer at caller memory.
int scaml_debug_set_stamp_buffer(long long* buf) {
  Module& m = module();
  if (m.load() != hipSuccess) return SCAML_E_LAUNCH;
  hipDeviceptr_t sym = nullptr;
  size_t bytes = 0;
  if (hipModuleGetGlobal(&sym, &bytes, m.mod, "g_stamp_buf") != hipSuccess) return SCAML_E_BADARG;
  return hipMemcpyHtoD(sym, &buf, sizeof(buf)) == hipSuccess ? 0 : SCAML_E_LAUNCH;
}

}  // extern "C"
