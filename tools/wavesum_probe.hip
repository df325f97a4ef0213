// Latency probe for the wave-wide f64 sum that every dot product of the solver ends in (tools/, not product code).
//   A: 4 DPP steps inside each 16-lane row + 8 v_readlane + 3 adds            (the round-2 tree)
//   B: 4 DPP steps + v_permlane16_swap + v_permlane32_swap (gfx950), 2 adds   (same tree, same bits, no SGPR round trip)
// Prints cycles (s_memtime, 100 MHz constant clock -> also wall-clock ns) per dependent reduction and checks B == A bitwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double rows(double v) {
  v += dpp_f64<0xB1>(v); v += dpp_f64<0x4E>(v); v += dpp_f64<0x141>(v); v += dpp_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ double sumA(double v) {
  v = rows(v);
  const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ double sumB(double v) {
  v = rows(v);
  {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);   // even row + odd row
  }
  {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);   // lower half + upper half
  }
  return v;
}
template <int V> __global__ void k(const double* in, double* out, long long* cyc, int iters) {
  double x = in[threadIdx.x], acc = 0.0;
  const double y = in[64 + threadIdx.x];
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    const double s = V == 0 ? sumA(x) : sumB(x);
    x = fma(-s * 1e-3, y, x);   // the axpy that depends on the reduction, as in the two-loop recursion
    acc += s;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x + acc;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double h[128]; for (int i = 0; i < 128; i++) h[i] = 1.0 / (1 + i) + (i % 7) * 1e-9;
  double *d, *o; long long* c;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 8); hipMalloc(&c, 8);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 100000;
  double res[2][64];
  for (int v = 0; v < 2; v++) {
    for (int rep = 0; rep < 2; rep++) {
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, o, c, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, o, c, iters);
      hipDeviceSynchronize();
    }
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost); hipMemcpy(res[v], o, 512, hipMemcpyDeviceToHost);
    printf("variant %c: %.1f ticks(10ns) per reduction+axpy = %.0f ns\n", 'A' + v, (double)cy / iters, 10.0 * cy / iters);
  }
  printf("bitwise equal: %s\n", memcmp(res[0], res[1], 512) == 0 ? "yes" : "NO");
  return 0;
}
