// Which HIP streams actually run concurrently?  Each stream gets one kernel of 64 single-wave blocks spinning ~100 ms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ void spin(unsigned long long* out, int k, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[2 * k] = t0; out[2 * k + 1] = wall_clock64(); }
}
int main(int argc, char** argv) {
  const int ns = argc > 1 ? atoi(argv[1]) : 4;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;  // 0 plain, 1 nonblocking, 2 priorities cycling
  hipStream_t main_s;
  hipStreamCreate(&main_s);
  std::vector<hipStream_t> st(ns);
  int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  printf("priority range least=%d greatest=%d\n", lo, hi);
  for (int i = 0; i < ns; i++) {
    if (mode == 0) hipStreamCreate(&st[i]);
    else if (mode == 1) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    else hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi + (i % (lo - hi + 1)));
  }
  unsigned long long* d;
  hipMalloc(&d, 16 * ns);
  hipMemset(d, 0, 16 * ns);
  hipEvent_t ev; hipEventCreate(&ev);
  hipEventRecord(ev, main_s);
  for (int i = 0; i < ns; i++) {
    hipStreamWaitEvent(st[i], ev, 0);
    hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st[i], d, i, 10000000ull);  // 100 ms at 100 MHz
  }
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * ns);
  hipMemcpy(h.data(), d, 16 * ns, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull;
  for (int i = 0; i < ns; i++) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
  for (int i = 0; i < ns; i++) printf("  stream %d: start %6.1f ms end %6.1f ms\n", i, (h[2 * i] - t0) * 1e-5, (h[2 * i + 1] - t0) * 1e-5);
  return 0;
}
