// gp_fit_fused.hip — fused "task-posterior" kernel for gfx950 (MI355X):
//   K = os * k(X/l, X/l) + (noise + jitter) I  ->  L L^T = K  ->  v = L^-1 y, alpha = L^-T v
//   quad = v.v, logdet = 2 sum log L_ii, mll = -(quad + logdet + n log 2pi) / (2 n)
// for a stack of T independent tasks, one workgroup per task.
//
// Replaces (for the whole stack at once) the reference's per-task chain
//   scamlgp/model.py:176-188 -> scamlgp/utils.py:171-177 -> gpytorch ExactMarginalLogLikelihood
//   -> linear_operator psd_safe_cholesky -> torch.linalg.cholesky_ex / solve_triangular.
//
// Design: the kernel matrix never exists in memory.  The trailing matrix of the
// right-looking Cholesky lives in MFMA accumulator registers: the lower triangle is cut into
// 16x16 tiles, tile t (column-major over the triangle) belongs to wave t % W, slot t / W, and is
// held in the C/D layout of v_mfma_f64_16x16x4_f64 (col = lane & 15, row = (lane >> 4) + 4 * reg).
// Each lane evaluates the kernel function for the four elements it owns straight into those
// registers.  Per 16-column panel: (P1) owners spill the panel column to LDS, (P2) wave 0
// factors the 16x16 diagonal block and forward-substitutes y, (P3) one thread per row solves
// the sub-diagonal rows against the block, (P4) every wave applies the rank-16 update to its
// tiles with 4 MFMAs per tile, operands read from the LDS panel; finished L tiles are written
// to HBM from registers as 128-byte row segments.  alpha comes from a blocked back-substitution
// over the L tiles still held in registers.  A failed pivot restarts the task in-kernel with the
// next jitter (1e-8, 1e-7, 1e-6), as linear_operator's psd_safe_cholesky does on the host.
#include "scaml_common.hpp"
#include "../../include/scaml_gp.h"

namespace scaml {

struct FitParams {
  const double* X;
  const double* y;
  const double* theta;
  const int32_t* n_points;
  const double* jitter_in;
  double* L;
  double* alpha;
  double* quad;
  double* logdet;
  double* mll;
  int32_t* info;
  double* jitter_used;
  int T, N, D;
  unsigned flags;
};

template <int NB, int W, int KIND>
__global__ __launch_bounds__(W * 64) void gp_fit_fused_kernel(FitParams p) {
  constexpr int NP = NB * 16;               // padded matrix order
  constexpr int NT = NB * (NB + 1) / 2;     // lower-triangular tiles
  constexpr int SLOTS = (NT + W - 1) / W;   // tiles per wave
  constexpr int PITCH = NP + 16;            // panel row pitch (doubles): conflict-free operand reads
  constexpr int NTHREADS = W * 64;
  static_assert(NTHREADS >= NP, "one thread per matrix row is required");

  extern __shared__ double lds[];
  // region A (overlaid): xsT [D][NP] during the kernel-matrix build; PT[2][16][PITCH] + LkkAll[NB][16][16] afterwards
  double* xsT = lds;
  double* PT = lds;
  double* LkkAll = lds + 2 * 16 * PITCH;
  const int regionA = (p.D * NP > 2 * 16 * PITCH + NB * 256) ? p.D * NP : 2 * 16 * PITCH + NB * 256;
  double* ytil = lds + regionA;   // [NP] running right-hand side
  double* vv = ytil + NP;         // [NP] v = L^-1 y
  double* ww = vv + NP;           // [NP] back-substitution workspace -> alpha
  double* dl = ww + NP;           // [NP] diag(L)
  double* rinv = dl + NP;         // [NP] 1 / diag(L)
  double* invl = rinv + NP;       // [D]  1 / lengthscale
  int* flagp = (int*)(invl + p.D + (p.D & 1));  // [2] fail index

  const int task = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15;   // tile column owned by this lane
  const int lq = lane >> 4;   // tile row group: rows lq + 4 * reg
  const int N = p.N, D = p.D;
  int n = p.n_points ? p.n_points[task] : N;
  n = n < 0 ? 0 : (n > N ? N : n);
  const double* Xg = p.X + (size_t)task * N * D;
  const double* yg = p.y + (size_t)task * N;
  const double* th = p.theta + (size_t)task * (D + 2);
  const double os = th[D];
  const double noise = th[D + 1];
  const double jit_in = p.jitter_in ? p.jitter_in[task] : 0.0;
  double* Lg = (p.flags & SCAML_FIT_STORE_L) ? p.L + (size_t)task * N * N : nullptr;

  // tile coordinates of this wave's slots (wave-uniform)
  int ti[SLOTS], tj[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    int t = s * W + wave;
    int j = 0, off = 0;
    while (j < NB - 1 && off + (NB - j) <= t) { off += NB - j; ++j; }
    bool valid = t < NT;
    tj[s] = valid ? j : -1;
    ti[s] = valid ? j + (t - off) : -1;
  }

  // strict upper triangle of L := 0 (optional), fire-and-forget
  if (Lg && (p.flags & SCAML_FIT_ZERO_UPPER)) {
    for (int e = tid; e < n * n; e += NTHREADS) {
      int r = e / n, c = e - r * n;
      if (c > r) Lg[(size_t)r * N + c] = 0.0;
    }
  }

  d4_t acc[SLOTS];
  int fail = 0;
  double jitter = 0.0;
  const int max_attempts = (p.flags & SCAML_FIT_NO_RETRY) ? 1 : 4;

  for (int attempt = 0; attempt < max_attempts; ++attempt) {
    jitter = attempt == 0 ? 0.0 : (attempt == 1 ? 1e-8 : (attempt == 2 ? 1e-7 : 1e-6));
    const double diag_add = noise + jitter + jit_in;
    __syncthreads();  // previous attempt done with region A
    if (tid < D) invl[tid] = 1.0 / th[tid];
    if (tid == 0) flagp[0] = 0;
    __syncthreads();
    // ---- stage X / l transposed into LDS: xsT[d][row]; y into ytil
    for (int e = tid; e < NP * D; e += NTHREADS) {
      int r = e / D, d = e - r * D;
      double v = r < n ? Xg[(size_t)r * D + d] * invl[d] : 0.0;
      xsT[d * NP + r] = v;
    }
    for (int r = tid; r < NP; r += NTHREADS) ytil[r] = r < n ? yg[r] : 0.0;
    __syncthreads();

    // ---- kernel matrix straight into the accumulator tiles
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (ti[s] >= 0) {
        const int col = 16 * tj[s] + lc;
        const int row0 = 16 * ti[s] + lq;
        double d2[4] = {0.0, 0.0, 0.0, 0.0};
        for (int d = 0; d < D; ++d) {
          const double* xr = xsT + d * NP;
          const double xc = xr[col];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            double df = xr[row0 + 4 * g] - xc;
            d2[g] = __builtin_fma(df, df, d2[g]);
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = row0 + 4 * g;
          double kv = os * kernel_from_sqdist<KIND>(d2[g]);
          if (row == col) kv += diag_add;
          if (row >= n || col >= n) kv = row == col ? 1.0 : 0.0;
          acc[s][g] = kv;
        }
      }
    }
    __syncthreads();  // xsT dead from here: region A becomes PT / LkkAll

    // ---- right-looking blocked Cholesky, panel width 16
    fail = 0;
    for (int k = 0; k < NB; ++k) {
      double* buf = PT + (k & 1) * 16 * PITCH;
      // P1: spill panel column k (raw trailing values) to LDS, buf[c][row]
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (tj[s] == k) {
          const int rb = 16 * ti[s] + lq;
#pragma unroll
          for (int g = 0; g < 4; ++g) buf[lc * PITCH + rb + 4 * g] = acc[s][g];
        }
      }
      __syncthreads();
      // P2: wave 0 factors the 16x16 diagonal block (row lc per lane, replicated over lq)
      if (wave == 0) {
        double a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) a[c] = buf[c * PITCH + 16 * k + lc];
        double yv = ytil[16 * k + lc];
        double my_rinv = 0.0, my_diag = 1.0, my_v = 0.0;
        int bad = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          double dpiv = readlane_f64(a[c], c);
          if (!(dpiv > 0.0)) {  // also catches NaN
            if (!bad) bad = 16 * k + c + 1;
            dpiv = 1.0;
          }
          const double ri = rsqrt_pos(dpiv);
          const double lcol = a[c] * ri;  // l_{r,c} for r > c
          const double ldiag = sqrt_from_rinv(dpiv, ri);
          a[c] = lc == c ? ldiag : (lc > c ? lcol : 0.0);
          const double vc = readlane_f64(yv, c) * ri;
          if (lc == c) { my_rinv = ri; my_diag = ldiag; my_v = vc; }
          yv = __builtin_fma(-a[c], vc, yv);
#pragma unroll
          for (int j = c + 1; j < 16; ++j) {
            const double ljc = readlane_f64(a[c], j);
            a[j] = __builtin_fma(-a[c], ljc, a[j]);
          }
        }
        if (lq == 0) {
          double* Lk = LkkAll + k * 256;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            Lk[lc * 16 + c] = a[c];
            buf[c * PITCH + 16 * k + lc] = a[c];
          }
          rinv[16 * k + lc] = my_rinv;
          dl[16 * k + lc] = my_diag;
          vv[16 * k + lc] = my_v;
        }
        if (bad && lane == 0) flagp[0] = bad;
      }
      __syncthreads();
      fail = flagp[0];
      if (fail) break;
      // P3: one thread per sub-diagonal row: x L_kk^T = a, then y_r -= x . v_k
      {
        const int r = 16 * (k + 1) + tid;
        if (r < NP) {
          double x[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) x[c] = buf[c * PITCH + r];
          const double* Lk = LkkAll + k * 256;
          const double* ri = rinv + 16 * k;
          const double* vk = vv + 16 * k;
          double yr = ytil[r];
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            x[c] *= ri[c];
#pragma unroll
            for (int j = c + 1; j < 16; ++j) x[j] = __builtin_fma(-x[c], Lk[j * 16 + c], x[j]);
            yr = __builtin_fma(-x[c], vk[c], yr);
          }
#pragma unroll
          for (int c = 0; c < 16; ++c) buf[c * PITCH + r] = x[c];
          ytil[r] = yr;
        }
      }
      __syncthreads();
      // P4: rank-16 trailing update on the matrix cores; finished tiles return to registers
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (tj[s] > k) {
          const double* pa = buf + lq * PITCH + 16 * ti[s] + lc;
          const double* pb = buf + lq * PITCH + 16 * tj[s] + lc;
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const double av = pa[4 * m * PITCH];
            const double bv = pb[4 * m * PITCH];
            acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, bv, acc[s], 0, 0, 0);
          }
        } else if (tj[s] == k) {
          const int rb = 16 * ti[s] + lq;
          const int col = 16 * k + lc;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = rb + 4 * g;
            double lv = buf[lc * PITCH + row];
            if (col > row) lv = 0.0;
            acc[s][g] = lv;
            if (Lg && row < n && col < n && col <= row) Lg[(size_t)row * N + col] = lv;
          }
        }
      }
    }
    if (!fail) break;
  }

  // ---- scalars: quad, logdet (wave 0), then alpha by blocked back-substitution
  __syncthreads();
  if (!fail) {
    if (wave == 0) {
      double q = 0.0, ld = 0.0;
      for (int r = lane; r < NP; r += 64) {
        const double v = vv[r];
        q = __builtin_fma(v, v, q);
        ld += log(dl[r]);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        q += __shfl_xor(q, o);
        ld += __shfl_xor(ld, o);
      }
      if (lane == 0) {
        ld *= 2.0;
        if (p.quad) p.quad[task] = q;
        if (p.logdet) p.logdet[task] = ld;
        if (p.mll) p.mll[task] = n > 0 ? -0.5 * (q + ld + n * 1.8378770664093454836) / n : 0.0;
      }
    }
    if (p.alpha) {
      for (int r = tid; r < NP; r += NTHREADS) ww[r] = vv[r];
      __syncthreads();
      for (int k = NB - 1; k >= 0; --k) {
        if (wave == 0) {
          // lane m (< 16) holds column m of L_kk and w_m; solve L_kk^T a = w
          const double* Lk = LkkAll + k * 256;
          double colm[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) colm[c] = Lk[c * 16 + lc];
          double wreg = ww[16 * k + lc];
          const double myri = rinv[16 * k + lc];
          double res = 0.0;
#pragma unroll
          for (int c = 15; c >= 0; --c) {
            const double ac = readlane_f64(wreg * myri, c);
            if (lc == c) res = ac;
            wreg = __builtin_fma(-colm[c], ac, wreg);  // only lanes m < c are used later
          }
          if (lq == 0) ww[16 * k + lc] = res;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          if (ti[s] == k && tj[s] < k) {
            double part = 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) part = __builtin_fma(acc[s][g], ww[16 * k + lq + 4 * g], part);
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (lq == 0) ww[16 * tj[s] + lc] -= part;
          }
        }
        __syncthreads();
      }
      for (int r = tid; r < n; r += NTHREADS) p.alpha[(size_t)task * N + r] = ww[r];
    }
  } else if (tid == 0) {
    const double nan = __builtin_nan("");
    if (p.quad) p.quad[task] = nan;
    if (p.logdet) p.logdet[task] = nan;
    if (p.mll) p.mll[task] = nan;
  }
  if (tid == 0) {
    p.info[task] = fail;
    if (p.jitter_used) p.jitter_used[task] = jitter;
  }
}

template <int NB, int W>
static size_t fit_lds_bytes(int D) {
  const int NP = NB * 16, PITCH = NP + 16;
  size_t regionA = (size_t)2 * 16 * PITCH + NB * 256;
  if ((size_t)D * NP > regionA) regionA = (size_t)D * NP;
  return (regionA + 5 * NP + D + (D & 1) + 2) * sizeof(double);
}

template <int NB, int W>
static int launch_fit(const FitParams& p, int kind, hipStream_t stream) {
  const size_t lds = fit_lds_bytes<NB, W>(p.D);
  if (lds > 160 * 1024) return SCAML_E_TOOLARGE;
  auto kern = kind == SCAML_KIND_RBF ? gp_fit_fused_kernel<NB, W, 0> : gp_fit_fused_kernel<NB, W, 1>;
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SCAML_E_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(p.T), dim3(W * 64), lds, stream, p);
  return hipGetLastError() == hipSuccess ? SCAML_OK : SCAML_E_LAUNCH;
}

}  // namespace scaml

static thread_local char g_last_error[256] = "";

extern "C" {

int scaml_version(void) { return 100; }  // 0.1.0
const char* scaml_last_error(void) { return g_last_error; }
int scaml_fit_max_n(void) { return 256; }

int scaml_fit_max_d(int N) {
  // largest D whose staged point stack fits the 160 KiB LDS next to the vectors
  int np = N <= 32 ? 32 : (N <= 64 ? 64 : (N <= 128 ? 128 : 256));
  int budget = 160 * 1024 / 8 - 5 * np - 4;
  int d = budget / (np + 1);
  return d > 1024 ? 1024 : d;
}

int scaml_gp_fit_fused_f64(const double* X, const double* y, const double* theta,
                           const int32_t* n_points, const double* jitter_in,
                           int T, int N, int D, int kind,
                           double* L, double* alpha, double* quad, double* logdet, double* mll,
                           int32_t* info, double* jitter_used, unsigned flags, void* stream) {
  if (T < 0 || N < 1 || D < 1) return SCAML_E_BADARG;
  if (!X || !y || !theta || !info) return SCAML_E_BADARG;
  if ((flags & SCAML_FIT_STORE_L) && !L) return SCAML_E_BADARG;
  if (kind != SCAML_KIND_RBF && kind != SCAML_KIND_MATERN52) return SCAML_E_BADARG;
  if (N > scaml_fit_max_n()) return SCAML_E_TOOLARGE;
  if (D > scaml_fit_max_d(N)) return SCAML_E_TOOLARGE;
  if (T == 0) return SCAML_OK;
  scaml::FitParams p{X, y, theta, n_points, jitter_in, L, alpha, quad, logdet, mll, info, jitter_used, T, N, D, flags};
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (N <= 32) rc = scaml::launch_fit<2, 1>(p, kind, s);
  else if (N <= 64) rc = scaml::launch_fit<4, 2>(p, kind, s);
  else if (N <= 128) rc = scaml::launch_fit<8, 4>(p, kind, s);
  else rc = scaml::launch_fit<16, 8>(p, kind, s);
  if (rc == SCAML_E_LAUNCH) {
    hipError_t e = hipGetLastError();
    snprintf(g_last_error, sizeof(g_last_error), "%s", hipGetErrorString(e));
  }
  return rc;
}

}  // extern "C"
