// Signed Euclidean distance fields of a map on the GPU — GridMap::updateESDF (src/map/src/grid_map.cpp:125-521) with the
// 1-D lower-envelope pass fillESDF (grid_map.cpp:89-123, Felzenszwalb & Huttenlocher): the step that precedes
// optimizeTraj in every episode of the benchmark loop.
//
// One thread per grid line and pass.  Lines are numbered so that consecutive threads own lines that are adjacent in
// memory (z fastest), which makes every step q of the sweep a coalesced access across the wave for the two strided
// passes; the envelope stacks v[] / z[] live in HBM workspace laid out [k][line] for the same reason.  Arithmetic and
// its order are the reference's (integer q*q, one division per envelope test, res * sqrt at the end), so the result is
// bit-identical to the CPU construction.
#pragma once
#include <hip/hip_runtime.h>

namespace topay {

struct EdtPass {
  long long nlines;        // lines of this pass (per map)
  int n;                   // cells per line
  long long inner;         // line -> base = (line / inner) * outer_stride + (line % inner) * inner_stride
  long long outer_stride, inner_stride;
  long long step;          // element stride along the line
  long long map_stride;    // elements between consecutive maps of a batch (blockIdx.y = map), same for source and target
  long long ws_stride;     // workspace elements per map
};

#define TOPAY_EDT_DMAX 1.79769313486231570815e+308

// SRC 0: occupancy bytes, value = ((occ == 1) == (pass == 0)) ? 0 : DMAX ; SRC 1: doubles
// FIN 0: store the squared distance ; FIN 1: dd = res * sqrt(val); pass 0: out = dd ; pass 1: if (dd > 0) out += res - dd
template <int SRC, int FIN>
__global__ void k_edt_pass(EdtPass P, const signed char* occ, const double* src, double* dst, int* vws, double* zws, int pass,
                           double res) {
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (line >= P.nlines) return;
  const long long mo = (long long)blockIdx.y * P.map_stride;
  if (SRC == 0) occ += mo; else src += mo;
  dst += mo;
  vws += (long long)blockIdx.y * P.ws_stride;
  zws += (long long)blockIdx.y * P.ws_stride;
  const long long base = (line / P.inner) * P.outer_stride + (line % P.inner) * P.inner_stride;
  const long long L = P.nlines;
  auto f = [&](int q) -> double {
    const long long a = base + (long long)q * P.step;
    if (SRC == 0) return ((occ[a] == 1) == (pass == 0)) ? 0.0 : TOPAY_EDT_DMAX;
    return src[a];
  };
  int* v = vws + line;      // v[k] at v[k * L]
  double* z = zws + line;   // z[k] at z[k * L]
  const int n = P.n;
  int k = 0;
  v[0] = 0;
  z[0] = -TOPAY_EDT_DMAX;
  z[L] = TOPAY_EDT_DMAX;
  for (int q = 1; q <= n - 1; q++) {
    k++;
    double s;
    const double fq = f(q) + (double)(q * q);
    do {
      k--;
      const int vk = v[(long long)k * L];
      s = (fq - (f(vk) + (double)(vk * vk))) / (double)(2 * q - 2 * vk);
    } while (s <= z[(long long)k * L]);
    k++;
    v[(long long)k * L] = q;
    z[(long long)k * L] = s;
    z[(long long)(k + 1) * L] = TOPAY_EDT_DMAX;
  }
  k = 0;
  for (int q = 0; q <= n - 1; q++) {
    while (z[(long long)(k + 1) * L] < (double)q) k++;
    const int vk = v[(long long)k * L];
    const double val = (double)((q - vk) * (q - vk)) + f(vk);
    const long long a = base + (long long)q * P.step;
    if (FIN == 0) {
      dst[a] = val;
    } else {
      const double dd = res * sqrt(val);
      if (pass == 0) dst[a] = dd;
      else if (dd > 0.0) dst[a] += (-dd + res);
    }
  }
}

}  // namespace topay
