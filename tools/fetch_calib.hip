// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the GP kernels use
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of a 16-B/lane streaming read; "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Each kernel streams BYTES once: read8 / read16 = 8 / 16 bytes per lane per load instruction; write8 / write16 likewise.
// hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/fetch_calib ; rocprofv3 --pmc WRITE_SIZE ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr size_t BYTES = 512ull << 20;   // > the 256 MB Infinity Cache: every pass comes from HBM

__global__ void read8(const double* __restrict__ p, size_t n, double* sink) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678) *sink = s;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* sink) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
  if (s == 12345.678) *sink = s;
}
__global__ void write8(double* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
__global__ void write16(double2* __restrict__ p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}
// 128-byte row segments, 8 bytes per lane, rows 2 KB apart: the fused fit's store pattern for L (and the L loads of the solve kernels)
__global__ void write8_rowseg(double* __restrict__ p, size_t nrows) {
  const int lane = threadIdx.x & 63, lq = lane >> 4, lc = lane & 15;
  for (size_t r0 = (blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6)) * 4; r0 < nrows; r0 += (size_t)gridDim.x * (blockDim.x >> 6) * 4)
    for (int seg = 0; seg < 16; ++seg) p[(r0 + lq) * 256 + 16 * seg + lc] = 1.0;
}
int main() {
  double *a, *sink;
  CHECK(hipMalloc(&a, BYTES)); CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(a, 0, BYTES));
  for (int rep = 0; rep < 3; ++rep) {
    read8<<<2048, 256>>>(a, BYTES / 8, sink);
    read16<<<2048, 256>>>((const double2*)a, BYTES / 16, sink);
    write8<<<2048, 256>>>(a, BYTES / 8);
    write16<<<2048, 256>>>((double2*)a, BYTES / 16);
    write8_rowseg<<<2048, 256>>>(a, BYTES / 2048);
  }
  CHECK(hipDeviceSynchronize());
  printf("fetch_calib: each kernel moved %zu bytes (%.1f MB = %.0f KB)\n", BYTES, BYTES / 1e6, BYTES / 1024.0);
  return 0;
}
