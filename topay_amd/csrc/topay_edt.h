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

#ifndef TOPAY_CPU_EMU
extern __shared__ double topay_edt_smem[];
#define TOPAY_EDT_LDS topay_edt_smem
#else
#define TOPAY_EDT_LDS ((double*)hip_emu::S().dyn_smem)
#endif

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
// WS  0: envelope stacks in the HBM workspace ; WS 1: in LDS (dynamic shared memory: z [n+2][64] doubles followed by
//        v [n+2][64] 16-bit indices, entry k of lane l at [k][l]: conflict-free), for lines of up to ~250 cells
template <int SRC, int FIN, int WS>
__global__ void k_edt_pass(EdtPass P, const signed char* occ, const double* src, double* dst, int* vws, double* zws, int pass,
                           double res) {
  double* edt_lds = TOPAY_EDT_LDS;  // dynamic shared memory of the block
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = line < P.nlines;
  const long long mo = (long long)blockIdx.y * P.map_stride;
  if (SRC == 0) occ += mo; else src += mo;
  dst += mo;
  const long long lc = live ? line : P.nlines - 1;   // idle lanes of the last block shadow the last line (no stores)
  const long long base = (lc / P.inner) * P.outer_stride + (lc % P.inner) * P.inner_stride;
  const int n = P.n;
  // stack accessors
  long long L;
  int* vg = nullptr;
  double* zg = nullptr;
  double* zl = nullptr;
  unsigned short* vl = nullptr;
  if (WS == 0) {
    L = P.nlines;
    vg = vws + (long long)blockIdx.y * P.ws_stride + lc;
    zg = zws + (long long)blockIdx.y * P.ws_stride + lc;
  } else {
    L = 64;
    zl = edt_lds + threadIdx.x;
    vl = (unsigned short*)(edt_lds + (size_t)(n + 2) * 64) + threadIdx.x;
  }
  auto V = [&](int k) -> int { return WS == 0 ? vg[(long long)k * L] : (int)vl[k * 64]; };
  auto Z = [&](int k) -> double { return WS == 0 ? zg[(long long)k * L] : zl[k * 64]; };
  auto setV = [&](int k, int q) { if (WS == 0) vg[(long long)k * L] = q; else vl[k * 64] = (unsigned short)q; };
  auto setZ = [&](int k, double z) { if (WS == 0) zg[(long long)k * L] = z; else zl[k * 64] = z; };
  auto f = [&](int q) -> double {
    const long long a = base + (long long)q * P.step;
    if (SRC == 0) return ((occ[a] == 1) == (pass == 0)) ? 0.0 : TOPAY_EDT_DMAX;
    return src[a];
  };
  int k = 0;
  setV(0, 0);
  setZ(0, -TOPAY_EDT_DMAX);
  setZ(1, TOPAY_EDT_DMAX);
  for (int q = 1; q <= n - 1; q++) {
    k++;
    double s;
    const double fq = f(q) + (double)(q * q);
    do {
      k--;
      const int vk = V(k);
      s = (fq - (f(vk) + (double)(vk * vk))) / (double)(2 * q - 2 * vk);
    } while (s <= Z(k));
    k++;
    setV(k, q);
    setZ(k, s);
    setZ(k + 1, TOPAY_EDT_DMAX);
  }
  k = 0;
  for (int q = 0; q <= n - 1; q++) {
    while (Z(k + 1) < (double)q) k++;
    const int vk = V(k);
    const double val = (double)((q - vk) * (q - vk)) + f(vk);
    const long long a = base + (long long)q * P.step;
    if (live) {
      if (FIN == 0) {
        dst[a] = val;
      } else {
        const double dd = res * sqrt(val);
        if (pass == 0) dst[a] = dd;
        else if (dd > 0.0) dst[a] += (-dd + res);
      }
    }
  }
}

// Seeds of the "inflate" fields (grid_map.cpp:283-300, 355-372): a cell is occupied when the field it is derived from
// is below the chassis radius there.
__global__ void k_edt_threshold(const double* field, double thr, signed char* occ, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) occ[i] = field[i] < thr ? 1 : 0;
}
// occ_buffer_2d_critical (grid_map.cpp:733-747): any point of the cloud above the cell, i.e. the 3-D occupancy
// projected onto the plane (used when the caller does not hand the buffer over).
__global__ void k_edt_project(const signed char* occ3, signed char* occ2, long long n2, int nz, long long maps) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2 * maps) return;
  const signed char* col = occ3 + i * nz;   // cell (map, x, y): nz consecutive bytes
  signed char any = 0;
  for (int z = 0; z < nz; z++) any |= (col[z] == 1);
  occ2[i] = any;
}

}  // namespace topay
