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

// ---------------------------------------------------------------------------------------------------------------
// The same 1-D pass for lines of up to 512 cells, by exhaustive search instead of the serial envelope construction.
// Every squared distance of the construction is an integer (or "no seed"), so the lower envelope's value at q is
// exactly  min over v of (q - v)^2 + f(v),  and that minimum can be taken directly: one thread per OUTPUT cell walks
// outwards from q and stops as soon as r^2 alone is no better than what it has (r^2 >= best).  Near obstacles that is
// a handful of steps; the work is 10-30 integer operations per cell instead of ~10 dependent memory round trips per
// cell, it is the same for every lane of a wave (neighbouring cells have neighbouring distances), and every access is
// coalesced.  32-bit integers, "no seed" = 2^30 (r^2 <= 2^18 cannot overflow it); results are the reference's bit for bit
// (tests/test_edt.py).  k_edt_direct: lines along the contiguous axis, neighbours straight from global memory (the reads
// of a wave overlap: L1).  k_edt_tile: strided lines; a tile of W memory-adjacent lines x n cells is staged in LDS
// ([n][W], conflict-free), so HBM sees one read and one write per cell and pass.
// ---------------------------------------------------------------------------------------------------------------
#define TOPAY_EDT_INF (1 << 30)

// SRC 0: occupancy bytes ; SRC 1: int32 squared distances.  FIN as in k_edt_pass (FIN 0 stores int32).
template <int SRC, int FIN>
__device__ __forceinline__ void edt_store(int best, long long a, int* dst_i, double* dst_d, int pass, double res) {
  if (FIN == 0) {
    dst_i[a] = best;
  } else {
    const double val = best >= TOPAY_EDT_INF ? TOPAY_EDT_DMAX : (double)best;
    const double dd = res * sqrt(val);
    if (pass == 0) dst_d[a] = dd;
    else if (dd > 0.0) dst_d[a] += (-dd + res);
  }
}

template <int SRC, int FIN>
__global__ void k_edt_direct(long long n_elems, int n, long long map_stride, const signed char* occ, const int* src, int* dst_i,
                             double* dst_d, int pass, double res) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // cell within the map, lines are [k n, (k + 1) n)
  if (e >= n_elems) return;
  const long long mo = (long long)blockIdx.y * map_stride;
  const int q = (int)((unsigned)e % (unsigned)n);   // (a map of lines <= 512 cells has fewer than 2^31 cells: 32-bit division)
  auto f = [&](long long a) -> int {
    if (SRC == 0) return ((occ[mo + a] == 1) == (pass == 0)) ? 0 : TOPAY_EDT_INF;
    return src[mo + a];
  };
  int best = f(e);
  for (int r = 1; r < n; r++) {
    const int rr = r * r;
    if (rr >= best) break;
    const bool lo = q - r >= 0, hi = q + r < n;
    if (!lo && !hi) break;
    if (lo) { const int t = rr + f(e - r); best = t < best ? t : best; }
    if (hi) { const int t = rr + f(e + r); best = t < best ? t : best; }
  }
  edt_store<SRC, FIN>(best, mo + e, dst_i, dst_d, pass, res);
}

// First pass along the contiguous axis when the lines are 16, 32 or 64 cells long (the benchmark map's z axis: 16): the
// input is binary, so the squared distance along the line is the square of the distance to the nearest seed, and a
// line is a bit field of the wave's ballot -- the nearest seed below / above cell q is a count-leading / trailing-zeros
// on the line's bits.  One thread per cell, no loop, loads and stores in memory order.
template <int N>
__global__ void k_edt_first_ballot(long long n_elems, long long map_stride, const signed char* occ, int* dst_i, int pass) {
  static_assert(N == 16 || N == 32 || N == 64, "line length must divide the wave");
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long mo = (long long)blockIdx.y * map_stride;
  const bool live = e < n_elems;
  const bool seed = live && ((occ[mo + (live ? e : 0)] == 1) == (pass == 0));
  const unsigned long long all = __ballot(seed);
  const int lane = threadIdx.x & 63, q = lane & (N - 1);
  const unsigned long long line = N == 64 ? all : ((all >> (lane - q)) & ((1ull << N) - 1ull));
  const unsigned long long below = line & ((q == 63) ? ~0ull : ((2ull << q) - 1ull));   // seeds at cells <= q
  const unsigned long long above = line >> q;                                             // seeds at cells >= q, shifted to bit 0
  int d = TOPAY_EDT_INF;
  if (below) d = q - (63 - __clzll((long long)below));
  if (above) { const int u = __ffsll((long long)above) - 1; d = u < d ? u : d; }
  if (live) dst_i[mo + e] = d >= TOPAY_EDT_INF ? TOPAY_EDT_INF : d * d;
}

// Tile t of a map: W lines whose cells q sit at  tile_base(t) + q * step + l,  l < W  (W contiguous cells), n cells each.
// tiles are numbered so that tile_base = (t / inner_tiles) * outer_stride + (t % inner_tiles) * W.
template <int SRC, int FIN>
__global__ void k_edt_tile(int n, int W, long long step, long long inner_tiles, long long outer_stride, long long map_stride,
                           const signed char* occ, const int* src, int* dst_i, double* dst_d, int pass, double res) {
  int* tile = (int*)TOPAY_EDT_LDS;   // [n][W]
  const long long t = blockIdx.x;
  const long long base = (long long)blockIdx.y * map_stride + (t / inner_tiles) * outer_stride + (t % inner_tiles) * W;
  const int cells = n * W;
  // (q, l) of cell c = threadIdx.x + k blockDim.x, advanced without a division per cell
  const int q0 = (int)threadIdx.x / W, l0 = (int)threadIdx.x - q0 * W, dq = (int)blockDim.x / W, dl = (int)blockDim.x - dq * W;
  int q = q0, l = l0;
  for (int c = threadIdx.x; c < cells; c += blockDim.x) {
    const long long a = base + (long long)q * step + l;
    tile[c] = SRC == 0 ? ((((occ[a] == 1) == (pass == 0)) ? 0 : TOPAY_EDT_INF)) : src[a];
    q += dq; l += dl;
    if (l >= W) { l -= W; q++; }
  }
  __syncthreads();
  q = q0; l = l0;
  for (int c = threadIdx.x; c < cells; c += blockDim.x, q += dq, l += dl) {
    if (l >= W) { l -= W; q++; }
    int best = tile[c];
    for (int r = 1; r < n; r++) {
      const int rr = r * r;
      if (rr >= best) break;
      const bool lo = q - r >= 0, hi = q + r < n;
      if (!lo && !hi) break;
      if (lo) { const int v = rr + tile[c - r * W]; best = v < best ? v : best; }
      if (hi) { const int v = rr + tile[c + r * W]; best = v < best ? v : best; }
    }
    edt_store<SRC, FIN>(best, base + (long long)q * step + l, dst_i, dst_d, pass, res);
  }
}

// Final pass of a signed field in one go: the squared distances of the positive and of the negative part (two int32
// volumes from the passes before) are both staged, both scanned, and the field  res sqrt(d+) - (res sqrt(d-) - res)  is
// written once -- the same operations in the same order as the two separate final passes (edt_store), without the
// read-modify-write of the double field in between.  LDS: two tiles [n][W].
__global__ void k_edt_tile_signed(int n, int W, long long step, long long inner_tiles, long long outer_stride, long long map_stride,
                                  const int* src_pos, const int* src_neg, double* dst_d, double res) {
  int* tp = (int*)TOPAY_EDT_LDS;   // [n][W]
  int* tn = tp + n * W;
  const long long t = blockIdx.x;
  const long long base = (long long)blockIdx.y * map_stride + (t / inner_tiles) * outer_stride + (t % inner_tiles) * W;
  const int cells = n * W;
  const int q0 = (int)threadIdx.x / W, l0 = (int)threadIdx.x - q0 * W, dq = (int)blockDim.x / W, dl = (int)blockDim.x - dq * W;
  int q = q0, l = l0;
  for (int c = threadIdx.x; c < cells; c += blockDim.x) {
    const long long a = base + (long long)q * step + l;
    tp[c] = src_pos[a];
    tn[c] = src_neg[a];
    q += dq; l += dl;
    if (l >= W) { l -= W; q++; }
  }
  __syncthreads();
  q = q0; l = l0;
  for (int c = threadIdx.x; c < cells; c += blockDim.x, q += dq, l += dl) {
    if (l >= W) { l -= W; q++; }
    int bp = tp[c], bn = tn[c];
    for (int r = 1; r < n; r++) {
      const int rr = r * r;
      if (rr >= bp && rr >= bn) break;
      const bool lo = q - r >= 0, hi = q + r < n;
      if (!lo && !hi) break;
      if (lo) {
        const int vp = rr + tp[c - r * W], vn = rr + tn[c - r * W];
        bp = vp < bp ? vp : bp;
        bn = vn < bn ? vn : bn;
      }
      if (hi) {
        const int vp = rr + tp[c + r * W], vn = rr + tn[c + r * W];
        bp = vp < bp ? vp : bp;
        bn = vn < bn ? vn : bn;
      }
    }
    const double vp = bp >= TOPAY_EDT_INF ? TOPAY_EDT_DMAX : (double)bp;
    const double vn = bn >= TOPAY_EDT_INF ? TOPAY_EDT_DMAX : (double)bn;
    double out = res * sqrt(vp);
    const double dn = res * sqrt(vn);
    if (dn > 0.0) out += (-dn + res);
    dst_d[base + (long long)q * step + l] = out;
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
