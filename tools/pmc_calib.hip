// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS path's access shapes (the guide's x2 correction is measured
// for 16 B/lane streaming loads only): known byte counts, 8 B per lane.
//   k_stream8 : coalesced 8 B/lane read of `bytes`               (L-BFGS history / vector traffic)
//   k_gather8 : 8 B/lane reads at pseudo-random cells of a table  (ESDF corner gathers), `bytes` requested in total
//   k_store8  : coalesced 8 B/lane write of `bytes`
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void k_stream8(const double* a, size_t n, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345) out[0] = s;
}
__global__ void k_gather8(const double* a, size_t n, size_t per_thread, double* out) {
  double s = 0;
  unsigned long long h = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1;
  for (size_t k = 0; k < per_thread; k++) {
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    s += a[h % n];
  }
  if (s == 1.2345) out[0] = s;
}
__global__ void k_store8(double* a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}
int main() {
  const size_t n = (size_t)1 << 29;  // 4 GiB of doubles: far past the 256 MiB Infinity Cache
  double *a, *out;
  CK(hipMalloc(&a, n * 8));
  CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, n * 8));
  hipLaunchKernelGGL(k_store8, dim3(4096), dim3(256), 0, 0, a, n);
  hipLaunchKernelGGL(k_stream8, dim3(4096), dim3(256), 0, 0, a, n, out);
  const size_t threads = 4096 * 256, per = 256;  // 2^28 gathers of 8 B = 2 GiB requested
  hipLaunchKernelGGL(k_gather8, dim3(4096), dim3(256), 0, 0, a, n, per, out);
  CK(hipDeviceSynchronize());
  printf("k_store8 bytes %zu\nk_stream8 bytes %zu\nk_gather8 requested bytes %zu (each 8-B read touches one 64-B line: %zu line bytes)\n",
         n * 8, n * 8, threads * per * 8, threads * per * 64);
  return 0;
}
