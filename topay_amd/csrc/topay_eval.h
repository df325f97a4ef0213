// One-wavefront-per-trajectory cost + analytic gradient of TopAY's (s,theta) NLP.
//
// Follows (all under /root/reference/src):
//   planner/src/moma_traj_opt.cpp:817-955   first/secondStageCostCallback
//   planner/src/moma_traj_opt.cpp:957-1198  calFirstStagePenalGrad
//   planner/src/moma_traj_opt.cpp:1200-1829 calSecondStagePenalGrad
//   planner/include/utils/minco.hpp:824-1069, banded_system.hpp:66-145
//   simulator/fake_moma/include/fake_moma/moma_param.h:203-337, map/include/map/grid_map.h:364-509
//
// Work decomposition inside the wave (64 lanes, one workgroup = one wave):
//   * "row lanes":    lane <-> row of the 6N x 6N MINCO system (2 rows per lane when 6N > 64).
//                     They own the gradient w.r.t. the coefficient rows (gdC) in registers.
//   * "sample lanes": lane <-> even ("full") Simpson sample e = 13*piece + m, 64 per pass.
//                     Each also handles the odd sample that follows it.
//   * banded LU / substitutions: the 6x7 (resp. 6x9) update block of one pivot across lanes, in LDS.
// XY positions are a wave prefix scan of Simpson panel integrals; the XY-gradient "chain"
// (moma_traj_opt.cpp:1313-1314,1667-1668,1812-1822) is the matching suffix scan, done in a second
// sweep.  Per-sample gradient rows travel sample-lane -> row-lane through a small LDS pass buffer.
#pragma once

#include <hip/hip_runtime.h>

#include "topay_math.h"
#include "topay_types.h"

#ifndef TOPAY_CPU_EMU
#define HIP_DYN_SHARED_DECL extern __shared__ double topay_lds[];
#endif

// Optimizer/robot parameters live in constant memory: every access is a scalar load the compiler can re-issue at
// the point of use instead of keeping hundreds of SGPRs of kernel arguments alive across the whole solve.
__constant__ DevParams g_P;
// The parameter block through ONE base address per function, held in a scalar register pair: the compiler otherwise forms the
// address of every field it reads from the program counter anew (s_getpc_b64 + 64-bit add: three scalar instructions ahead of
// each of the 133 parameter loads of the manipulator block -- a wave issues one instruction per four cycles whatever its kind).
// The empty asm hides where the pointer comes from, so the fields become immediate offsets from it.
typedef const TOPAY_CST DevParams& dev_params_ref;
__device__ __forceinline__ dev_params_ref dev_params() {
#ifndef TOPAY_CPU_EMU
  const TOPAY_CST DevParams* p = (const TOPAY_CST DevParams*)&g_P;
  asm("" : "+s"(p));
  return *p;
#else
  return g_P;
#endif
}

// Optional scheduling fences between the independent sub-blocks of the manipulator block (off: a leftover of rounds 1-2;
// every kernel is built for 256 registers since round 4 and the block's sphere loop carries its own fence, TOPAY_OCC2_FENCE).
#ifdef TOPAY_USE_SCHED_FENCE
#define TOPAY_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define TOPAY_SCHED_FENCE() do { } while (0)
#endif

// The compiler must treat the value as changed (no instruction is emitted): stops common-subexpression reuse across a
// rarely taken path, which would otherwise be paid for with registers on the common one.
#ifndef TOPAY_CPU_EMU
#define TOPAY_OPAQUE(x) asm volatile("" : "+v"(x))
#define TOPAY_OPAQUE_I(x) asm volatile("" : "+v"(x))
#else
#define TOPAY_OPAQUE(x) do { } while (0)
#define TOPAY_OPAQUE_I(x) do { } while (0)
#endif

// On the functions that call the non-inlined device functions.  Those callees use the whole register file, and the
// compiler lets such a function skip the saving of callee-saved registers (the caller then saves exactly what it keeps across
// the call) only if no call of it carries LLVM's `tail` marker -- which the optimiser adds to every call that is handed no
// pointer into the caller's stack frame.  Round 5 took the stack out of the manipulator block's interface, the marker
// appeared, and the block began to save and restore all 112 callee-saved VGPRs on every call (388 scratch instructions).
// With this attribute the marker is not added.
#ifndef TOPAY_CPU_EMU
#define TOPAY_CALLS_BIG_FUNCTIONS __attribute__((disable_tail_calls))
#else
#define TOPAY_CALLS_BIG_FUNCTIONS
#endif

// A wave-uniform `true` the compiler cannot see through (one s_cmp + s_cbranch): starts a new basic block on purpose.
__device__ __forceinline__ bool topay_opaque_true() {
#ifndef TOPAY_CPU_EMU
  int one = 1;
  asm volatile("" : "+s"(one));
  return one != 0;
#else
  return true;
#endif
}

#ifndef TOPAY_ESDF_LOOKAHEAD
#define TOPAY_ESDF_LOOKAHEAD 2
#endif
#ifndef TOPAY_OCC2_FENCE
#define TOPAY_OCC2_FENCE 1
#endif
#ifndef TOPAY_ESDF_LOOKAHEAD_OCC2
#define TOPAY_ESDF_LOOKAHEAD_OCC2 1
#endif

namespace topay {

// ---------------------------------------------------------------------------------------------
// wave helpers (collectives: call only from wave-uniform control flow)
// ---------------------------------------------------------------------------------------------
// One fixed summation tree for every wave reduction: xor-butterfly with offsets 1,2,4,8 inside each 16-lane row
// (DPP quad_perm / row_half_mirror / row_mirror; additions commute, so all lanes of a row end up with identical
// bits), then (r0+r1)+(r2+r3) over the four rows via readlane.  Every lane receives the same bits, which keeps
// wave-uniform control flow uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
#ifndef TOPAY_CPU_EMU
  // (mov_dpp: no "old" operand -- with update_dpp(lo, lo, ...) the compiler copies the source into the destination first,
  // two more 32-bit moves per level of every reduction: 8 of the 39 vector instructions per history pair of the two-loop recursion)
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
#else
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
#endif
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// rows of 16 lanes outside ROWMASK receive 0.0 (the caller adds the result)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64_rows(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
// 64-lane sum: a butterfly inside every row of 16 (every lane of row k then holds r_k), then across the rows:
// (r0 + r1) + (r2 + r3), in lane 63.  The sum is 20 instructions of VALU issue -- f64 has no DPP operand form, every
// level is two 32-bit DPP moves and an add -- and that, not the latency of the chain, is what a reduction costs
// (docs/EXPERIMENTS.md); the cross-row levels replace four lane reads and three adds of rounds 1-2, same bits (the
// operands of every addition are the same, in the other order).
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]   : lane ^ 1
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]   : lane ^ 2
  v += dpp_f64<0x141>(v);  // row_half_mirror       : quad q <-> quad q^1
  v += dpp_f64<0x140>(v);  // row_mirror            : half h <-> half h^1
#ifndef TOPAY_CPU_EMU
  // Only lane 63 is read below: the rows a broadcast does not reach may hold anything, so the moves need no zeroed
  // destination (two more 32-bit moves per level with the row masks of the emulator's form; lane 63's operands are the same).
  v += dpp_f64<0x142>(v);             // row_bcast15: row k += lane 15 of row k - 1               -> lane 63: r3 + r2, lane 31: r1 + r0
  v += dpp_f64<0x143>(v);             // row_bcast31: rows 2, 3 += lane 31                        -> lane 63: (r3 + r2) + (r1 + r0)
#else
  v += dpp_f64_rows<0x142, 0xA>(v);   // row_bcast15: rows 1 and 3 += lane 15 of the row before  -> r1 + r0, r3 + r2
  v += dpp_f64_rows<0x143, 0xC>(v);   // row_bcast31: rows 2 and 3 += lane 31                    -> (r3 + r2) + (r1 + r0)
#endif
  return readlane_f64(v, 63);
}
__device__ __forceinline__ double wave_max(double v) {
  double o;
  o = dpp_f64<0xB1>(v); v = o > v ? o : v;
  o = dpp_f64<0x4E>(v); v = o > v ? o : v;
  o = dpp_f64<0x141>(v); v = o > v ? o : v;
  o = dpp_f64<0x140>(v); v = o > v ? o : v;
  const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
  const double a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
  return a > b ? a : b;
}
// LDS hand-off between lanes of the one wave of this workgroup: LDS operations of a wave complete in issue
// order, so only compiler reordering has to be prevented (no s_barrier, no vmcnt drain).
__device__ __forceinline__ void lds_sync() {
#ifdef TOPAY_FULL_SYNC
  __syncthreads();
  return;
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
// Hand-off through GLOBAL memory between the lanes of one wave (stores by some lanes, loads by others): the stores are
// complete (vmcnt) before any lane goes on, without a workgroup barrier -- usable by one wave of a several-waves workgroup.
__device__ __forceinline__ void wave_global_sync() {
#ifndef TOPAY_CPU_EMU
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
  __builtin_amdgcn_wave_barrier();
#endif
}
// The lane's number formed anew (two instructions, no operand): used after a call of the manipulator block, so that the
// number -- and everything derived from it -- need not be carried across the call in a register the callee clobbers.
__device__ __forceinline__ int fresh_lane_id(int known) {
#ifndef TOPAY_CPU_EMU
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  (void)known;
  return l;
#else
  return known;
#endif
}
// inclusive prefix sum over lanes
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    double o = __shfl_up(v, off);
    if (lane >= off) v += o;
  }
  return v;
}
// inclusive suffix sum over lanes
__device__ __forceinline__ double wave_incl_rscan(double v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    double o = __shfl_down(v, off);
    if (lane + off < 64) v += o;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------
// scalar pieces — moma_traj_opt.h:745-830
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double expC2(double tau) {
  return tau > 0.0 ? ((0.5 * tau + 1.0) * tau + 1.0) : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0);
}
__device__ __forceinline__ double logC2(double T) {
  return T > 1.0 ? (sqrt(2.0 * T - 1.0) - 1.0) : (1.0 - sqrt(2.0 / T - 1.0));
}
__device__ __forceinline__ double dTdTau(double tau) {
  if (tau > 0) return tau + 1.0;
  double den = (0.5 * tau - 1.0) * tau + 1.0;
  return (1.0 - tau) / (den * den);
}
__device__ __forceinline__ double sigmoidC2(double vq, double max_q) {
  double e = expC2(vq);
  return 2.0 * max_q * e / (1.0 + e) - max_q;
}
__device__ __forceinline__ double invSigmoidC2(double q, double max_q) {
  double b = 0.5 * (max_q + q) / max_q;
  return logC2(b / (1 - b));
}
__device__ __forceinline__ double dQdVq(double vq, double max_q) {
  double e1 = expC2(vq) + 1.0;
  return 2.0 * max_q * dTdTau(vq) / (e1 * e1);
}
// smoothL1Penalty, only meaningful for x > 0 — moma_traj_opt.h:810-830 (constants precomputed in DevParams)
__device__ __forceinline__ void smoothL1(dev_params_ref P, double x, double mu, double& f, double& df) {
  if (x < mu) {
    f = (P.sl_f4c * x + P.sl_f3c) * x * x * x;
    df = (P.sl_d3c * x + P.sl_d2c) * x * x;
  } else {
    f = x - P.sl_half;
    df = 1.0;
  }
}
// 1/K as a constant factor: the reference divides by int_K in every penalty term (e.g. moma_traj_opt.cpp:1315); a
// multiplication by the rounded reciprocal differs by at most one ulp and saves an IEEE division per term.
#define TOPAY_INV_K (1.0 / TOPAY_K)

// ---------------------------------------------------------------------------------------------
// ESDF interpolation — grid_map.h:364-441 (2-D), 443-509 (3-D); out of map => d = 0, grad = 0
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
// Clamped cell index pair (i, i+1) -> (lo, hi) as grid_map.h:727-733 does, without ever forming i + 1 on an
// unclamped i: a saturated float->int conversion (points far outside the map) would overflow, and the compiler
// may assume it does not.
__device__ __forceinline__ void clamp_pair(int i, int top, int& lo, int& hi) {
  const int ic = i < -1 ? -1 : (i > top ? top : i);
  lo = ic < 0 ? 0 : ic;
  hi = ic + 1 > top ? top : ic + 1;
}

// Wave-uniform copy of a map descriptor, forced into scalar registers.
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// Wave-uniform pointers forced into scalar registers: a value the compiler cannot prove uniform (loaded from the context
// block in private memory) lives in a vector register, and everything in vector registers that is live across the call of
// the manipulator block is saved to and restored from scratch memory around it, once per sample pass.
template <typename T>
__device__ __forceinline__ T* uniform_gptr(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
#ifndef TOPAY_CPU_EMU
template <typename T>
__device__ __forceinline__ TOPAY_GLB T* uniform_ptr(TOPAY_GLB T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (TOPAY_GLB T*)(((unsigned long long)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ TOPAY_LDS T* uniform_ptr(TOPAY_LDS T* p) {
  return (TOPAY_LDS T*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)p);
}
#else
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) { return p; }
#endif

__device__ __forceinline__ DevMap load_map(const TOPAY_GLB DevMap* mp) {
  DevMap m;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    m.origin[a] = uniform_f64(mp->origin[a]);
    m.min_b[a] = uniform_f64(mp->min_b[a]);
    m.max_b[a] = uniform_f64(mp->max_b[a]);
    m.dims[a] = __builtin_amdgcn_readfirstlane(mp->dims[a]);
  }
  m.res = uniform_f64(mp->res);
  m.res_inv = uniform_f64(mp->res_inv);
  m.pad = 0;
  {
    const unsigned long long p2 = (unsigned long long)mp->esdf2d, p3 = (unsigned long long)mp->esdf3d;
    const unsigned lo2 = __builtin_amdgcn_readfirstlane((int)(p2 & 0xffffffffu)), hi2 = __builtin_amdgcn_readfirstlane((int)(p2 >> 32));
    const unsigned lo3 = __builtin_amdgcn_readfirstlane((int)(p3 & 0xffffffffu)), hi3 = __builtin_amdgcn_readfirstlane((int)(p3 >> 32));
    m.esdf2d = (glb_cdp)(((unsigned long long)hi2 << 32) | lo2);
    m.esdf3d = (glb_cdp)(((unsigned long long)hi3 << 32) | lo3);
  }
  return m;
}

__device__ __forceinline__ void esdf2d_query(const DevMap& M, double px, double py, double& dist, double& gx, double& gy) {
  bool in = !(px < M.min_b[0] + 1e-4 || py < M.min_b[1] + 1e-4 || px > M.max_b[0] - 1e-4 || py > M.max_b[1] - 1e-4);
  dist = 0.0; gx = 0.0; gy = 0.0;
  if (in) {
    const double r = M.res, ri = M.res_inv;
    int ix = (int)floor((px - 0.5 * r - M.origin[0]) * ri);
    int iy = (int)floor((py - 0.5 * r - M.origin[1]) * ri);
    double dx = (px - ((ix + 0.5) * r + M.origin[0])) * ri;
    double dy = (py - ((iy + 0.5) * r + M.origin[1])) * ri;
    const int ny = M.dims[1];
    int x0, x1, y0, y1;
    clamp_pair(ix, M.dims[0] - 1, x0, x1);
    clamp_pair(iy, ny - 1, y0, y1);
    glb_cdp e = M.esdf2d;
    double v00 = e[(size_t)x0 * ny + y0], v01 = e[(size_t)x0 * ny + y1];
    double v10 = e[(size_t)x1 * ny + y0], v11 = e[(size_t)x1 * ny + y1];
    double v0 = v00 * (1 - dx) + v10 * dx;
    double v1 = v01 * (1 - dx) + v11 * dx;
    dist = v0 * (1 - dy) + v1 * dy;
    gy = (v1 - v0) * ri;
    double g0 = (1 - dy) * (v10 - v00);
    g0 += dy * (v11 - v01);
    gx = g0 * ri;
  }
}

__device__ __forceinline__ void esdf3d_query(const DevMap& M, double px, double py, double pz, double& dist, double& gx,
                                             double& gy, double& gz) {
  bool in = !(px < M.min_b[0] + 1e-4 || py < M.min_b[1] + 1e-4 || pz < M.min_b[2] + 1e-4 ||
              px > M.max_b[0] - 1e-4 || py > M.max_b[1] - 1e-4 || pz > M.max_b[2] - 1e-4);
  // branch-free: the gathers are issued unconditionally at clamped indices (so that the scheduler can start them
  // early and overlap several spheres) and the result is discarded for points outside the map (d = 0, grad = 0)
  {
    const double r = M.res, ri = M.res_inv;
    int ix = (int)floor((px - 0.5 * r - M.origin[0]) * ri);
    int iy = (int)floor((py - 0.5 * r - M.origin[1]) * ri);
    int iz = (int)floor((pz - 0.5 * r - M.origin[2]) * ri);
    double dx = (px - ((ix + 0.5) * r + M.origin[0])) * ri;
    double dy = (py - ((iy + 0.5) * r + M.origin[1])) * ri;
    double dz = (pz - ((iz + 0.5) * r + M.origin[2])) * ri;
    const int ny = M.dims[1], nz = M.dims[2];
    int x0, x1, y0, y1, z0, z1;
    clamp_pair(ix, M.dims[0] - 1, x0, x1);
    clamp_pair(iy, ny - 1, y0, y1);
    clamp_pair(iz, nz - 1, z0, z1);
    glb_cdp e = M.esdf3d;
    size_t b00 = ((size_t)x0 * ny + y0) * nz, b01 = ((size_t)x0 * ny + y1) * nz;
    size_t b10 = ((size_t)x1 * ny + y0) * nz, b11 = ((size_t)x1 * ny + y1) * nz;
    double v000 = e[b00 + z0], v001 = e[b00 + z1], v010 = e[b01 + z0], v011 = e[b01 + z1];
    double v100 = e[b10 + z0], v101 = e[b10 + z1], v110 = e[b11 + z0], v111 = e[b11 + z1];
    const double ex = 1 - dx, ey = 1 - dy, ez = 1.0 - dz;
    double v00 = fma(v100, dx, v000 * ex);
    double v01 = fma(v101, dx, v001 * ex);
    double v10 = fma(v110, dx, v010 * ex);
    double v11 = fma(v111, dx, v011 * ex);
    double v0 = fma(v10, dy, v00 * ey);
    double v1 = fma(v11, dy, v01 * ey);
    dist = fma(v1, dz, v0 * ez);
    gz = (v1 - v0) * ri;
    gy = fma(v11 - v01, dz, (v10 - v00) * ez) * ri;
    double g0 = ez * ey * (v100 - v000);
    g0 = fma(ez * dy, v110 - v010, g0);
    g0 = fma(dz * ey, v101 - v001, g0);
    g0 = fma(dz * dy, v111 - v011, g0);
    gx = g0 * ri;
  }
  dist = in ? dist : 0.0; gx = in ? gx : 0.0; gy = in ? gy : 0.0; gz = in ? gz : 0.0;
}

// The same lookup split in two so that the gathers of the next sphere can be in flight while the penalties of the
// current one (divergent branches the scheduler will not move loads across) are evaluated.
//
// Four 16-byte gathers instead of eight 8-byte ones (round 5).  The two z-neighbours of a corner pair are adjacent doubles of
// the field (x-major, z fastest), so one load fetches both: the pair starts at zb = min(z0, nz - 2) and the clamped indices z0,
// z1 (equal at either face of the map) pick their element of it -- the same eight values into the same arithmetic, half the
// vector-memory instructions.  A gather costs the compute unit's L1 one tag lookup per lane whatever its width, and eight waves
// of a compute unit share that L1: 96 -> 48 such instructions per call of the manipulator block.  (The loads are 8-byte
// aligned; a pair that straddles a cache line costs two lookups, one case in eight or sixteen.  nz >= 2 is checked when a map
// is set: with a single layer the pair would reach past the field.)
typedef double esdf_pair __attribute__((vector_size(16), aligned(8)));
struct Esdf3dReq {
  esdf_pair p00, p01, p10, p11;   // (z pair) of the rows (x0, y0), (x0, y1), (x1, y0), (x1, y1)
  double dx, dy, dz;
  bool in, z0hi, z1hi;            // z0 / z1 is the pair's second element
};
__device__ __forceinline__ void esdf3d_issue(const DevMap& M, double px, double py, double pz, Esdf3dReq& q) {
  q.in = !(px < M.min_b[0] + 1e-4 || py < M.min_b[1] + 1e-4 || pz < M.min_b[2] + 1e-4 ||
           px > M.max_b[0] - 1e-4 || py > M.max_b[1] - 1e-4 || pz > M.max_b[2] - 1e-4);
  const double r = M.res, ri = M.res_inv;
  int ix = (int)floor((px - 0.5 * r - M.origin[0]) * ri);
  int iy = (int)floor((py - 0.5 * r - M.origin[1]) * ri);
  int iz = (int)floor((pz - 0.5 * r - M.origin[2]) * ri);
  q.dx = (px - ((ix + 0.5) * r + M.origin[0])) * ri;
  q.dy = (py - ((iy + 0.5) * r + M.origin[1])) * ri;
  q.dz = (pz - ((iz + 0.5) * r + M.origin[2])) * ri;
  const int ny = M.dims[1], nz = M.dims[2];
  int x0, x1, y0, y1, z0, z1;
  clamp_pair(ix, M.dims[0] - 1, x0, x1);
  clamp_pair(iy, ny - 1, y0, y1);
  clamp_pair(iz, nz - 1, z0, z1);
  glb_cdp e = M.esdf3d;
  // One linear index with a multiply (integer multiplies run at a quarter of the vector rate), the other rows by adding the
  // strides of the axes along which the clamped neighbour differs (x1 - x0, y1 - y0 are 0 or 1).  A field has fewer than
  // 2^32 cells (checked when the map is set), so the indices are 32-bit.
  const int zb = z0 < nz - 2 ? z0 : nz - 2;
  q.z0hi = z0 != zb;
  q.z1hi = z1 != zb;
  const unsigned i00 = ((unsigned)x0 * (unsigned)ny + (unsigned)y0) * (unsigned)nz + (unsigned)zb;
  const unsigned sx = x1 != x0 ? (unsigned)(ny * nz) : 0u, sy = y1 != y0 ? (unsigned)nz : 0u;
  const unsigned i01 = i00 + sy, i10 = i00 + sx;
  const unsigned i11 = i10 + sy;
#ifndef TOPAY_CPU_EMU
  typedef const TOPAY_GLB esdf_pair* pair_ptr;
  q.p00 = *(pair_ptr)(e + (size_t)i00);
  q.p01 = *(pair_ptr)(e + (size_t)i01);
  q.p10 = *(pair_ptr)(e + (size_t)i10);
  q.p11 = *(pair_ptr)(e + (size_t)i11);
#else
  q.p00[0] = e[(size_t)i00]; q.p00[1] = e[(size_t)i00 + 1];
  q.p01[0] = e[(size_t)i01]; q.p01[1] = e[(size_t)i01 + 1];
  q.p10[0] = e[(size_t)i10]; q.p10[1] = e[(size_t)i10 + 1];
  q.p11[0] = e[(size_t)i11]; q.p11[1] = e[(size_t)i11 + 1];
#endif
}
__device__ __forceinline__ void esdf3d_finish(const DevMap& M, const Esdf3dReq& q, double& dist, double& gx, double& gy,
                                              double& gz) {
  const double ri = M.res_inv;
  const double dx = q.dx, dy = q.dy, dz = q.dz;
  const double ex = 1 - dx, ey = 1 - dy, ez = 1.0 - dz;
  const double v000 = q.z0hi ? q.p00[1] : q.p00[0], v001 = q.z1hi ? q.p00[1] : q.p00[0];
  const double v010 = q.z0hi ? q.p01[1] : q.p01[0], v011 = q.z1hi ? q.p01[1] : q.p01[0];
  const double v100 = q.z0hi ? q.p10[1] : q.p10[0], v101 = q.z1hi ? q.p10[1] : q.p10[0];
  const double v110 = q.z0hi ? q.p11[1] : q.p11[0], v111 = q.z1hi ? q.p11[1] : q.p11[0];
  double v00 = fma(v100, dx, v000 * ex);
  double v01 = fma(v101, dx, v001 * ex);
  double v10 = fma(v110, dx, v010 * ex);
  double v11 = fma(v111, dx, v011 * ex);
  double v0 = fma(v10, dy, v00 * ey);
  double v1 = fma(v11, dy, v01 * ey);
  dist = fma(v1, dz, v0 * ez);
  gz = (v1 - v0) * ri;
  gy = fma(v11 - v01, dz, (v10 - v00) * ez) * ri;
  double g0 = ez * ey * (v100 - v000);
  g0 = fma(ez * dy, v110 - v010, g0);
  g0 = fma(dz * ey, v101 - v001, g0);
  g0 = fma(dz * dy, v111 - v011, g0);
  gx = g0 * ri;
  dist = q.in ? dist : 0.0; gx = q.in ? gx : 0.0; gy = q.in ? gy : 0.0; gz = q.in ? gz : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Wave-level evaluation context: LDS carve-up + per-trajectory global pointers
// ---------------------------------------------------------------------------------------------
struct EvalCtx {
  int lane, N, rows, n;
  // several waves per trajectory (topay_eval_mw.h): thread index in the workgroup, wave index, small cross-wave scratch
  int tid, wave;
  lds_dp red;    // [8] partial sums of a workgroup reduction (two phases) | [64] pass totals | [2][NW][64] per-round costs | [NW] masks
  lds_dp adj;    // [9][rows] right-hand sides / solution of the adjoint solve (the coefficients' block: == cL)
  glb_dp coefg;  // HBM copy of the coefficients (the candidate's result block), read by the dJ/dT correction
  int cl_in_lds; // the coefficients of the last evaluation are still in C.cL (0 after a gradient phase -- they are in coefg)
  // LDS
  lds_dp cL;     // [9][rows]  MINCO coefficients, column d contiguous (the reference's col-major c)
  lds_dp Tp;     // [5][N]     T, T^2..T^5
  glb_cdp hd, tl; // HBM [27] each: head / tail PVA, 9x3 col-major (read once per evaluation by the right-hand side)
  int npass_lds;  // passes the pass-total block of the LDS plan is sized for
  lds_dp gdT;    // [N]        penalty dJ/dT accumulator
  lds_dp pcs;    // [4*(N+1)]  per-piece scratch: stage-1 tracking gradient (2N) | piece-end XY (2(N+1))
  lds_dp gC;     // [9][rows]  penalty dJ/dC accumulator, element (row, d) owned by the row lane of `row`
  glb_dp sbuf;   // HBM [14][sb_stride]: per-sample gradient rows parked between the cost and the gradient phase
  int sb_stride;
  glb_dp mstash; // HBM [sb_stride][36]: forces of self-colliding sphere pairs of a sample (manipulator_block; rarely touched)
  lds_dp pw;     // [26][6]    integer powers jj^k of the Simpson sample index (constant for the whole solve)
  lds_dp X;      // union region: band + reciprocal diagonal (14*rows) | sample buffers (26N + 960)
  // global
  glb_cdp x;
  glb_dp g;
  glb_dp lu;          // [14*rows] stash
  glb_cdp init_xy;
  double sx, sy, ex, ey;           // start xy, goal xy
  double lam0, lam1, rho0, rho1;   // ALM state
  double fxe0, fxe1;               // final_xy_error of this evaluation (stage 2)
  // diagnostic build only (TOPAY_STAMPS): per-phase shader-clock accumulators, [16] per trajectory
  TOPAY_GLB long long* stamps;
  long long t_last;
};

// jj^k for jj = 0..25 (sample index within a piece; 25 is read but always multiplied by zero), k = 0..5: the local time of sample jj is jj * hs, so the
// monomial basis of coefficient row k factors as (jj^k) * hs^k and the row lanes only need the three hs-powers
// of their row (basis_k(k, hs)) once per pass instead of a power chain per sample.
__device__ __forceinline__ void fill_power_table(lds_dp pw, int lane) {
  for (int t = lane; t < 156; t += 64) {
    const int jj = t / 6, k = t - 6 * jj;
    double v = 1.0;
    for (int u = 0; u < k; u++) v *= (double)jj;
    pw[t] = v;
  }
}

#define BAND(i, j) band[((i) - (j) + 6) * rows + (j)]

// Phase stamps for the diagnostic build (-DTOPAY_STAMPS): never compiled into the product library.
#ifdef TOPAY_STAMPS
#define STAMP(C, k)                                                    \
  do {                                                                 \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();    \
    if ((C).stamps && (C).lane == 0) (C).stamps[k] += now_ - (C).t_last; \
    (C).t_last = (long long)__builtin_amdgcn_s_memtime();              \
  } while (0)
#else
#define STAMP(C, k) do { } while (0)
#endif
// sub-interval stamp that does not reset the phase clock
#ifdef TOPAY_STAMPS
#define SUBSTAMP_BEGIN(C) const long long sub_t0_ = (long long)__builtin_amdgcn_s_memtime()
#define SUBSTAMP_END(C, k)                                                                                   \
  do {                                                                                                       \
    if ((C).stamps && (C).lane == 0) (C).stamps[k] += (long long)__builtin_amdgcn_s_memtime() - sub_t0_;      \
  } while (0)
#else
#define SUBSTAMP_BEGIN(C) do { } while (0)
#define SUBSTAMP_END(C, k) do { } while (0)
#endif

// Decides, once the cost of an evaluation is known, whether its gradient will be used.  In the reference's line search
// (lbfgs.hpp:318-340) a trial that fails the sufficient-decrease test is discarded without its gradient ever being
// read (the next trial overwrites it, an error exit restores the previous one), and that is 37 % of all evaluations:
// for those the gradient phase -- row accumulation, sweep 2, adjoint solve, assembly -- is skipped.  The decision
// uses the same expressions as the line search itself, so the iteration is unchanged.
struct GradGate {
  bool always;      // first evaluation of a run, test hooks
  bool has_early;   // past > 0
  double finit, thr /* finit + stp * dgtest */, early /* delta / past */;
  // Early rejection (stage 2): every term of the cost is non-negative, so once the cost accumulated so far exceeds
  // skip_thr the trial is certain to fail the sufficient-decrease test and certain not to be early-accepted; when the
  // line search is also certain to continue after such a failure (early_ok, decided by the solver), nothing of this
  // trial is read except that verdict, and the sample bodies of the remaining passes are skipped.
  bool early_ok;
  double skip_thr;
  __device__ __forceinline__ bool needs(double f) const {
    if (always) return true;
    if (isinf(f) || isnan(f)) return false;                                        // INVALID_FUNCVAL: reverted
    if (has_early && fabs(finit - f) / (fabs(finit) + 1.0) < early) return true;   // early accept
    return !(f > thr);                                                             // else: needs g . d
  }
};

// Banded triangular sweeps with one lane per right-hand side (banded_system.hpp:96-118 and 123-145).
// The reference's substitutions are column sweeps: step j finalises x(j) and updates the six following (or
// preceding) entries of the same right-hand side.  The nine right-hand sides are independent, so lane d < 9 owns
// column d of the 6N x 9 block and carries the six pending entries in registers: a step is six independent
// multiply-subtracts with no LDS round trip and no barrier (the cross-lane version needed both, ~300 cycles per
// step).  Arithmetic and its order per entry are unchanged (mul, then sub, j ascending / descending).
//   MODE 0  L   x = b   (generate, forward):  b(i) -= A(i,j) b(j),            i = j+1..j+6
//   MODE 1  U   x = b   (generate, backward): b(i) -= A(i,j) (b(j)/A(j,j)),   i = j-1..j-6 ; stores b(j)/A(j,j)
//   MODE 2  U^T x = b   (adjoint, forward):   b(i) -= A(j,i) (b(j)/A(j,j)),   i = j+1..j+6 ; stores b(j)/A(j,j)
//   MODE 3  L^T x = b   (adjoint, backward):  b(i) -= A(j,i) b(j),            i = j-1..j-6
// rows = 6N is a multiple of 6: blocks of six steps with compile-time register indices.
//
// The factors are NOT resident in LDS (round 4): the band (84 N doubles) beside the right-hand sides (54 N) was the
// peak of the LDS plan and decided how many trajectories share a compute unit.  They stream from the candidate's LU
// block in HBM ([14][rows]: 13 diagonals, then the reciprocal diagonal; written once per evaluation by the
// factorisation) through two windows of 7 rows x 30 columns in LDS: a chunk is 24 steps (four blocks), all 64 lanes
// of the wave request the next chunk's window, the nine owner lanes sweep the current one, the requested values are
// written to the other window.  Window row r holds diagonal D0 + r (D0 = 7: the lower factor, modes 0 and 3; D0 = 0:
// the upper factor, modes 1 and 2), row 6 the reciprocal diagonal; window column = matrix column - clo.  Which value
// feeds which multiply-subtract is unchanged.
#define TOPAY_SWEEP_CHUNK 24
#define TOPAY_SWEEP_WCOLS 30
#define TOPAY_SWEEP_WIN (7 * TOPAY_SWEEP_WCOLS)   // doubles per window; the sweeps use two
template <int MODE>
__device__ __forceinline__ void band_sweep(lds_dp v, bool owner, glb_cdp lu, lds_dp win, int rows, int lane) {
  constexpr bool FWD = (MODE == 0 || MODE == 2);
  constexpr bool SCALE = (MODE == 1 || MODE == 2);
  constexpr int D0 = (MODE == 0 || MODE == 3) ? 7 : 0;
  constexpr int CH = TOPAY_SWEEP_CHUNK, WC = TOPAY_SWEEP_WCOLS, WIN = TOPAY_SWEEP_WIN;
  constexpr int NEL = (SCALE ? 7 : 6) * WC;      // window elements in use
  constexpr int NQ = (NEL + 63) / 64;            // per lane
  double w[6];
  double x = 0.0;
#pragma unroll
  for (int t = 0; t < 6; t++) w[t] = 0.0;
  if (owner) {
    if (FWD) {
      x = v[0];
#pragma unroll
      for (int t = 0; t < 6; t++) w[t] = v[1 + t];
    } else {
      x = v[rows - 1];
#pragma unroll
      for (int t = 0; t < 6; t++) w[t] = v[rows - 2 - t];
    }
  }
  const int nchunk = (rows + CH - 1) / CH;
  // first matrix column of chunk k's window
  auto chunk_clo = [&](int k) { return FWD ? CH * k : rows - 1 - CH * k - (WC - 1); };
  auto request = [&](int k, double (&q)[NQ]) {
    const int clo = chunk_clo(k);
#pragma unroll
    for (int u = 0; u < NQ; u++) {
      const int e = lane + 64 * u;
      const int r = e / WC, cc = e - r * WC;
      int c = clo + cc;
      c = c < 0 ? 0 : (c > rows - 1 ? rows - 1 : c);   // columns outside the matrix only ever feed rows that do not exist
      const int d = (r < 6 ? D0 + r : 13);
      q[u] = lu[(d < 14 ? d : 13) * rows + c];         // (e >= NEL: a valid address, the value is dropped)
    }
  };
  auto deposit = [&](int k, const double (&q)[NQ]) {
    lds_dp wb = win + (k & 1) * WIN;
#pragma unroll
    for (int u = 0; u < NQ; u++) {
      const int e = lane + 64 * u;
      if (e < NEL) wb[e] = q[u];
    }
  };
  double q[NQ];
  request(0, q);
  deposit(0, q);
  lds_sync();
  for (int k = 0; k < nchunk; k++) {
    const bool more = k + 1 < nchunk;
    if (more) request(k + 1, q);
    if (owner) {
      lds_cdp wb = win + (k & 1) * WIN;
      const int clo = chunk_clo(k);
#pragma unroll 1
      for (int b0 = CH * k; b0 < CH * (k + 1) && b0 < rows; b0 += 6) {
        // everything the block reads from LDS, issued up front: 36 coefficients, 6 scales, 6 incoming entries
        double cf[6][6], sc[6], nw[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const int j = FWD ? b0 + u : rows - 1 - (b0 + u);
#pragma unroll
          for (int t = 1; t <= 6; t++) {
            int idx;
            if (MODE == 0) idx = (t - 1) * WC + (j - clo);            // A(j+t, j)
            else if (MODE == 1) idx = (6 - t) * WC + (j - clo);       // A(j-t, j)
            else if (MODE == 2) idx = (6 - t) * WC + (j + t - clo);   // A(j, j+t)
            else idx = (t - 1) * WC + (j - t - clo);                  // A(j, j-t)
            // Unconditional reads.  Where row i = j +- t falls outside the matrix the value is a never-written zero of
            // the band (modes 0, 1) or an unrelated entry (modes 2, 3) and only ever feeds window slots of rows that
            // do not exist and are never stored.
            cf[u][t - 1] = wb[idx];
          }
          if (SCALE) sc[u] = wb[6 * WC + (j - clo)];
          int in = FWD ? j + 7 : j - 7;
          in = in < 0 ? 0 : (in > rows - 1 ? rows - 1 : in);
          nw[u] = v[in];
        }
        double xo[6];
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const double xs = SCALE ? x * sc[u] : x;
          xo[u] = xs;
#pragma unroll
          for (int t = 0; t < 6; t++) w[(u + t) % 6] -= cf[u][t] * xs;
          x = w[u % 6];
          w[u % 6] = nw[u];
        }
#pragma unroll
        for (int u = 0; u < 6; u++) {
          const int j = FWD ? b0 + u : rows - 1 - (b0 + u);
          v[j] = xo[u];
        }
      }
    }
    if (more) deposit(k + 1, q);
    lds_sync();
  }
}

// polynomial basis of local time s: b0 = s^k, b1, b2, b3 derivatives — moma_traj_opt.cpp:1263-1270
struct Basis {
  double b0[6], b1[6], b2[6], b3[6];
};
__device__ __forceinline__ void make_basis(double s1, Basis& B) {
  const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s3 * s2;
  B.b0[0] = 1.0; B.b0[1] = s1; B.b0[2] = s2; B.b0[3] = s3; B.b0[4] = s4; B.b0[5] = s5;
  B.b1[0] = 0.0; B.b1[1] = 1.0; B.b1[2] = 2.0 * s1; B.b1[3] = 3.0 * s2; B.b1[4] = 4.0 * s3; B.b1[5] = 5.0 * s4;
  B.b2[0] = 0.0; B.b2[1] = 0.0; B.b2[2] = 2.0; B.b2[3] = 6.0 * s1; B.b2[4] = 12.0 * s2; B.b2[5] = 20.0 * s3;
  B.b3[0] = 0.0; B.b3[1] = 0.0; B.b3[2] = 0.0; B.b3[3] = 6.0; B.b3[4] = 24.0 * s1; B.b3[5] = 60.0 * s2;
}
// value and derivatives of dimension d of piece i at the basis point (reads 6 coefficients from LDS)
__device__ __forceinline__ void poly4(lds_cdp cL, int rows, int i, int d, const Basis& B, double& p0, double& p1,
                                      double& p2, double& p3) {
  lds_cdp c = cL + d * rows + 6 * i;
  const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
  p0 = fma(c5, B.b0[5], fma(c4, B.b0[4], fma(c3, B.b0[3], fma(c2, B.b0[2], fma(c1, B.b0[1], c0)))));
  p1 = fma(c5, B.b1[5], fma(c4, B.b1[4], fma(c3, B.b1[3], fma(c2, B.b1[2], c1))));
  p2 = fma(c5, B.b2[5], fma(c4, B.b2[4], fma(c3, B.b2[3], c2 * B.b2[2])));
  p3 = fma(c5, B.b3[5], fma(c4, B.b3[4], c3 * B.b3[3]));
}
__device__ __forceinline__ void poly3(lds_cdp cL, int rows, int i, int d, const Basis& B, double& p0, double& p1,
                                      double& p2) {
  lds_cdp c = cL + d * rows + 6 * i;
  const double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
  p0 = fma(c5, B.b0[5], fma(c4, B.b0[4], fma(c3, B.b0[3], fma(c2, B.b0[2], fma(c1, B.b0[1], c0)))));
  p1 = fma(c5, B.b1[5], fma(c4, B.b1[4], fma(c3, B.b1[3], fma(c2, B.b1[2], c1))));
  p2 = fma(c5, B.b2[5], fma(c4, B.b2[4], fma(c3, B.b2[3], c2 * B.b2[2])));
}

// integrand of the Simpson XY integral at local time s of piece i: sdot*(cos th, sin th)
__device__ __forceinline__ void xy_integrand(lds_cdp cL, int rows, int i, double s1, double& fx, double& fy) {
  const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s3 * s2;
  lds_cdp ct = cL + 0 * rows + 6 * i;
  lds_cdp cs = cL + 1 * rows + 6 * i;
  const double th = fma(ct[5], s5, fma(ct[4], s4, fma(ct[3], s3, fma(ct[2], s2, fma(ct[1], s1, ct[0])))));
  const double sd = fma(cs[5], 5.0 * s4, fma(cs[4], 4.0 * s3, fma(cs[3], 3.0 * s2, fma(cs[2], 2.0 * s1, cs[1]))));
  double sn, cn;
  det_sincos(th, &sn, &cn);
  fx = sd * cn;
  fy = sd * sn;
}

// ---------------------------------------------------------------------------------------------
// Stage-2 manipulator block of one even sample: FK (moma_param.h:203-247), 12 ESDF lookups
// (moma_traj_opt.cpp:1477-1520), self collision (1521-1612), Jacobian-transpose (moma_param.h:249-337),
// joint position limits (1616-1666).
//
// World sphere centre  P_k = p0 + A * rho_k,  A = Rz(theta) * relative_R,  p0 = (x, y, h) + Rz(theta) * relative_t,
// where rho_k comes from the joint chain run in the arm-local frame (pure rotations).  relative_R is the
// reference's 0.7071068 literal matrix, i.e. not exactly orthonormal, so the joint torques are formed in the local
// frame from g' = A^T g:  tau_i = u_i . sum (rho - o_{i+1}) x g'  — the exact derivative of the reference's matrix
// products for any A, unlike the world-frame axis x r form.  Yaw and x, y are taken in the world frame (Rz exact).
// pos = (x, y, theta, q1..q7).  Returns cost and the "/K" gdT part; moma_grad[10] = d/d(x, y, theta, q).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void joint_rotate(double* R, int i, double c_, double s_) {
  if (i % 2 == 0) {  // R <- R * Rz(q): mixes columns 0,1
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double r0 = R[a * 3 + 0], r1 = R[a * 3 + 1];
      R[a * 3 + 0] = fma(r0, c_, r1 * s_);
      R[a * 3 + 1] = fma(r1, c_, -(r0 * s_));
    }
  } else {  // R <- R * Ry(q): mixes columns 0,2
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double r0 = R[a * 3 + 0], r2 = R[a * 3 + 2];
      R[a * 3 + 0] = fma(r0, c_, -(r2 * s_));
      R[a * 3 + 2] = fma(r0, s_, r2 * c_);
    }
  }
}

// Interface (round 5).  Nothing of the block's inputs or outputs travels through the stack any more:
//   * in: the sample (piece i, even sample index j, local half step, step) and its XY position -- the pose (theta, q1..q7)
//     is evaluated HERE from the coefficients in LDS (round 4 passed the ten pose values by value: 16 of the 32 argument
//     dwords went through scratch memory because the hidden return-value pointer took the 33rd register);
//   * out: five doubles in registers (an aggregate of at most 16 dwords is returned in VGPRs): d/dx, d/dy, d/dtheta, cost and
//     the "/K" dJ/dT part; the seven joint entries of moma_grad go to the lane's column of seven rows of the wave's LDS pass
//     buffer (mg_lds[q * 64], q = 0..6: the buffer is idle during the sample passes), each as soon as its torque is known --
//     what the 12-double return value in scratch memory did for the register pressure of the block's last part, without the
//     scratch memory.
// e = the sample's index in the candidate's self-collision block, -1 for a padding lane (results dropped, no HBM write).
struct ManiOut {
  double gx, gy, gth;
  double cost, gdT;
};
#ifdef TOPAY_ASM_MARKS   // (probe builds: comment lines in the assembly that delimit the sections of the block)
#define MMARK(k) asm volatile("; TOPAY_MARK " #k)
#else
#define MMARK(k) do { } while (0)
#endif
#ifdef TOPAY_STAMPS
__device__ long long g_mani_stamps[8];
#define MSTAMP(k)                                                                                   \
  do {                                                                                              \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();                                 \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_mani_stamps[k] += now_ - mt_;                        \
    mt_ = now_;                                                                                     \
  } while (0)
#else
#define MSTAMP(k) do { } while (0)
#endif
// Register plan (round 4).  OCC = waves per SIMD the caller's kernel is built for: 1 -> 512 registers per lane, 2 -> 256.
// The block used to hold the 12 sphere centres AND 12 force accumulators (144 VGPRs) because the rare self-collision
// pairs add to the forces of two spheres at once, ahead of the per-sphere terms.  Now a sphere's force is born in the
// iteration of the sphere loop that consumes its centre (the arm-local force takes the centre's registers), and the pair
// contributions -- needed by fewer than one sample in a thousand -- are accumulated in an HBM block by the lanes that
// have any, in the pair order of before, and read back at the top of the sphere's iteration: same operands, same order,
// same bits as the 144-register version.  LA = spheres whose ESDF gathers are issued ahead (2 with one wave per SIMD;
// 1 with two, where the other wave covers the latency and the request registers are what is scarce).
template <int OCC>
__device__ __noinline__ ManiOut manipulator_block(const TOPAY_GLB DevMap* mp, lds_cdp cL, int rows, int pi, int pj, double half, double step,
                                                  double posx, double posy, int e, glb_dp mstash, lds_dp mg_lds) {
  dev_params_ref P = dev_params();
  const DevMap M = load_map(mp);
  const bool in_act = e >= 0;
  const double invK = topay_hold_f64(TOPAY_INV_K), ten = topay_hold_f64(10.0);
  const glb_dp in_stash = mstash + 36 * (in_act ? e : 0);
  // pose of the sample: order-0 polynomials of theta and the seven joints (the arc length is not part of the pose), in the
  // arithmetic of poly4 / make_basis
  double pos[10];
  pos[0] = posx; pos[1] = posy;
  double sth, cth;
  {
    const double s1 = pj * half;
    const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s3 * s2;
#pragma unroll
    for (int d = 0; d < 9; d++) {
      if (d == 1) continue;
      lds_cdp c = cL + d * rows + 6 * pi;
      pos[d == 0 ? 2 : d + 1] = fma(c[5], s5, fma(c[4], s4, fma(c[3], s3, fma(c[2], s2, fma(c[1], s1, c[0])))));
    }
  }
  const double omg = (pj == 0 || pj == 2 * TOPAY_K) ? 0.5 : 1.0;
  ManiOut out;
  double cost, gdTk;
  const double mu = P.relu_mu;
  const double w = omg * step;
#ifdef TOPAY_STAMPS
  long long mt_ = (long long)__builtin_amdgcn_s_memtime();
#endif
  double sq[7], cq[7];
  {
    // yaw and the seven joints, step by step across the eight angles (det_sincos_n: the constants of a step are formed once)
    double ang[8], sn8[8], cs8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ang[i] = pos[2 + i];
    det_sincos_n<8>(ang, sn8, cs8);
    sth = sn8[0]; cth = cs8[0];
#pragma unroll
    for (int i = 0; i < 7; i++) { sq[i] = sn8[1 + i]; cq[i] = cs8[1 + i]; }
  }
  MSTAMP(0);  // 8 sincos
  MMARK(0);
  double A[9];
  {
    const double Rz[9] = {cth, -sth, 0.0, sth, cth, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++)
        A[a * 3 + b] = Rz[a * 3 + 0] * P.relR[0 * 3 + b] + Rz[a * 3 + 1] * P.relR[1 * 3 + b] + Rz[a * 3 + 2] * P.relR[2 * 3 + b];
  }
  const double p0x = pos[0] + (cth * P.relT[0] - sth * P.relT[1]);
  const double p0y = pos[1] + (sth * P.relT[0] + cth * P.relT[1]);
  const double p0z = P.p0z;
  // walk 1: world sphere centres (the arm-local rho_k are not kept; the torque walks below regenerate them)
  // spheres per link: link0:{0,1} 1:{2} 2:{3,4} 3:{5} 4:{6,7} 5:{8} 6:{9,10} 7:{11}
  double Px[TOPAY_NSPH], Py[TOPAY_NSPH], Pz[TOPAY_NSPH];
  {
    double R[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
    double q0 = 0.0, q1 = 0.0, q2 = 0.0;
    int sidx = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int cnt = (i % 2 == 0) ? 2 : 1;
#pragma unroll
      for (int c = 0; c < cnt; c++) {
        const double lx = fma(R[2], P.sph_off[sidx], q0), ly = fma(R[5], P.sph_off[sidx], q1), lz = fma(R[8], P.sph_off[sidx], q2);
        Px[sidx] = p0x + fma(A[2], lz, fma(A[1], ly, A[0] * lx));
        Py[sidx] = p0y + fma(A[5], lz, fma(A[4], ly, A[3] * lx));
        Pz[sidx] = p0z + fma(A[8], lz, fma(A[7], ly, A[6] * lx));
        sidx++;
      }
      q0 = fma(R[2], P.colli_length[i], q0);
      q1 = fma(R[5], P.colli_length[i], q1);
      q2 = fma(R[8], P.colli_length[i], q2);
      if (i == 7) break;
      joint_rotate(R, i, cq[i], sq[i]);
      TOPAY_SCHED_FENCE();
    }
  }
  MSTAMP(1);  // walk 1
  MMARK(1);
  // The joints' cosines wait in the lane's LDS column (the seven words that take the torques at the end) while the sphere
  // loop needs the registers: walk 2a reads them from there, walk 2b reads word i before torque i is written to it.  (They
  // and the sines used to be spilled to scratch memory across the loop by the compiler: 14 of the block's 25 spilled values.)
#pragma unroll
  for (int i = 0; i < 7; i++) mg_lds[i * 64] = cq[i];
  cost = 0.0;
  gdTk = 0.0;
  const double wMC = P.s2_mani_colli_weight, wSC = P.s2_self_colli_weight;
  // sphere pairs: collision_matrix == -1 <=> non-adjacent spheres (moma_param.h:128-143: at the zero pose
  // only self and neighbouring spheres overlap) — moma_traj_opt.cpp:1566-1611
  // The clearances of all 55 pairs are independent straight-line arithmetic; only a lane that sees a positive one walks
  // the penalty path, which recomputes the same expressions and keeps the forces of the two spheres in its column of the
  // HBM block (read-modify-write, pair order).
  bool anypair;
  {
    double worst[TOPAY_NSPH - 2];
    double wall = -1.0;
#pragma unroll
    for (int a = 0; a < TOPAY_NSPH - 2; a++) {
      worst[a] = -1.0;
#pragma unroll
      for (int b = a + 2; b < TOPAY_NSPH; b++) {
        const double dx = Px[a] - Px[b], dy = Py[a] - Py[b], dz = Pz[a] - Pz[b];
        const double dist = P.pair_rr2[a * TOPAY_NSPH + b] - fma(dz, dz, fma(dy, dy, dx * dx));
        worst[a] = fmax(worst[a], dist);
      }
      wall = fmax(wall, worst[a]);
    }
    anypair = in_act && wall > 0;
    if (anypair) {
      // (the centres are made opaque here: otherwise the compiler keeps the 165 coordinate differences of the screening
      // above alive for this path -- in scratch memory -- instead of recomputing the few it needs)
#pragma unroll
      for (int k = 0; k < TOPAY_NSPH; k++) { TOPAY_OPAQUE(Px[k]); TOPAY_OPAQUE(Py[k]); TOPAY_OPAQUE(Pz[k]); }
      const glb_dp sg = in_stash;
#pragma unroll
      for (int v = 0; v < 3 * TOPAY_NSPH; v++) sg[v] = 0.0;
#pragma unroll
      for (int a = 0; a < TOPAY_NSPH - 2; a++) {
        if (worst[a] > 0) {
#pragma unroll
          for (int b = a + 2; b < TOPAY_NSPH; b++) {
            const double dx = Px[a] - Px[b], dy = Py[a] - Py[b], dz = Pz[a] - Pz[b];
            const double dist = P.pair_rr2[a * TOPAY_NSPH + b] - fma(dz, dz, fma(dy, dy, dx * dx));
            if (dist > 0) {
              double pe, pd;
              smoothL1(P, dist, mu, pe, pd);
              const double sc = -w * wSC * pd * 2.0;
              sg[3 * a + 0] = fma(sc, dx, sg[3 * a + 0]);
              sg[3 * a + 1] = fma(sc, dy, sg[3 * a + 1]);
              sg[3 * a + 2] = fma(sc, dz, sg[3 * a + 2]);
              sg[3 * b + 0] = fma(-sc, dx, sg[3 * b + 0]);
              sg[3 * b + 1] = fma(-sc, dy, sg[3 * b + 1]);
              sg[3 * b + 2] = fma(-sc, dz, sg[3 * b + 2]);
              gdTk += omg * wSC * (pe * invK);
              cost += w * wSC * pe;
            }
          }
        }
      }
    }
  }
  MSTAMP(2);  // sphere pairs
  MMARK(2);
  // chassis top (spheres with index > 2, 1525-1539) and environment collision (1477-1520)
  double bFx = 0.0, bFy = 0.0, bMz = 0.0;  // base: x, y, yaw (everything rotates about the vertical axis through (x, y))
  constexpr int LA = OCC >= 2 ? TOPAY_ESDF_LOOKAHEAD_OCC2 : TOPAY_ESDF_LOOKAHEAD;  // spheres whose gathers are issued ahead
  Esdf3dReq rq[LA + 1];
  double Lx[TOPAY_NSPH], Ly[TOPAY_NSPH], Lz[TOPAY_NSPH];   // arm-local forces g' = A^T g
#pragma unroll
  for (int k = 0; k < LA; k++) esdf3d_issue(M, Px[k], Py[k], Pz[k], rq[k]);
#pragma unroll
  for (int k = 0; k < TOPAY_NSPH; k++) {
    if (k + LA < TOPAY_NSPH) esdf3d_issue(M, Px[k + LA], Py[k + LA], Pz[k + LA], rq[(k + LA) % (LA + 1)]);
    double Gx = 0.0, Gy = 0.0, Gz = 0.0;
    if (anypair) {
      const glb_cdp sg = in_stash;
      Gx = sg[3 * k + 0]; Gy = sg[3 * k + 1]; Gz = sg[3 * k + 2];
    }
    if (k >= 3) {
      const double height = P.sph_top[k] - Pz[k];
      if (height > 0) {
        double pe, pd;
        smoothL1(P, height, mu, pe, pd);
        Gz += -w * wSC * pd;
        gdTk += omg * wSC * (pe * invK);
        cost += w * wSC * pe;
      }
    }
    double d, gx, gy, gz;
#ifdef TOPAY_STAMPS
    {
      // exposed latency of this sphere's four pair gathers: cycles until they have returned (the 4 min(LA, spheres left)
      // issued after them may stay in flight), measured where the first of them is needed
      const long long w0_ = (long long)__builtin_amdgcn_s_memtime();
      constexpr int FULL = 4 * LA;   // (four pair gathers per sphere) s_waitcnt vmcnt(n): expcnt / lgkmcnt fields left at their maxima
      const int left = TOPAY_NSPH - 1 - k;
      if (left >= LA) __builtin_amdgcn_s_waitcnt(0x0f70 | (FULL & 15) | ((FULL >> 4) << 14));
      else if (left == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | 4);
      else __builtin_amdgcn_s_waitcnt(0x0f70);
      const long long w1_ = (long long)__builtin_amdgcn_s_memtime();
      if (blockIdx.x == 0 && threadIdx.x == 0) { g_mani_stamps[6] += w1_ - w0_; g_mani_stamps[7] += 1; }
    }
#endif
    esdf3d_finish(M, rq[k % (LA + 1)], d, gx, gy, gz);
    const double viola = P.sph_viol[k] - d * ten;
    if (viola > 0) {
      double pe, pd;
      smoothL1(P, viola, mu, pe, pd);
      const double sc = -w * wMC * pd;
      Gx += sc * gx * ten; Gy += sc * gy * ten; Gz += sc * gz * ten;
      gdTk += omg * wMC * (pe * invK);
      cost += w * wMC * pe;
    }
    bFx += Gx;
    bFy += Gy;
    bMz = fma(Px[k] - pos[0], Gy, fma(-(Py[k] - pos[1]), Gx, bMz));
    // g' = A^T g (arm-local frame); the world position is no longer needed
    Lx[k] = fma(A[6], Gz, fma(A[3], Gy, A[0] * Gx));
    Ly[k] = fma(A[7], Gz, fma(A[4], Gy, A[1] * Gx));
    Lz[k] = fma(A[8], Gz, fma(A[5], Gy, A[2] * Gx));
    if (OCC >= 2 && TOPAY_OCC2_FENCE) __builtin_amdgcn_sched_barrier(0);
    TOPAY_SCHED_FENCE();
  }
  MSTAMP(3);  // ESDF loop
  MMARK(3);
  // joints: tau_i = u_i . (Mo_beyond - o_{i+1} x F_beyond) with F, Mo = sums of g' and rho x g'.
  // walk 2a accumulates the totals, walk 2b peels off the links at or below each joint.
  double Fx = 0, Fy = 0, Fz = 0, Mx = 0, My = 0, Mz = 0;
  {
    double R[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
    double q0 = 0.0, q1 = 0.0, q2 = 0.0;
    int sidx = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int cnt = (i % 2 == 0) ? 2 : 1;
#pragma unroll
      for (int c = 0; c < cnt; c++) {
        const double lx = fma(R[2], P.sph_off[sidx], q0), ly = fma(R[5], P.sph_off[sidx], q1), lz = fma(R[8], P.sph_off[sidx], q2);
        Fx += Lx[sidx]; Fy += Ly[sidx]; Fz += Lz[sidx];
        Mx = fma(ly, Lz[sidx], fma(-lz, Ly[sidx], Mx));
        My = fma(lz, Lx[sidx], fma(-lx, Lz[sidx], My));
        Mz = fma(lx, Ly[sidx], fma(-ly, Lx[sidx], Mz));
        sidx++;
      }
      q0 = fma(R[2], P.colli_length[i], q0);
      q1 = fma(R[5], P.colli_length[i], q1);
      q2 = fma(R[8], P.colli_length[i], q2);
      if (i == 7) break;
      joint_rotate(R, i, mg_lds[i * 64], sq[i]);
      TOPAY_SCHED_FENCE();
    }
  }
  {
    double R[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
    double o0 = 0.0, o1 = 0.0, o2 = 0.0;
    int sidx = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
      const int cnt = (i % 2 == 0) ? 2 : 1;
#pragma unroll
      for (int c = 0; c < cnt; c++) {  // remove link i's spheres from the "beyond" sums
        const double lx = fma(R[2], P.sph_off[sidx], o0), ly = fma(R[5], P.sph_off[sidx], o1), lz = fma(R[8], P.sph_off[sidx], o2);
        Fx -= Lx[sidx]; Fy -= Ly[sidx]; Fz -= Lz[sidx];
        Mx = fma(-ly, Lz[sidx], fma(lz, Ly[sidx], Mx));
        My = fma(-lz, Lx[sidx], fma(lx, Lz[sidx], My));
        Mz = fma(-lx, Ly[sidx], fma(ly, Lx[sidx], Mz));
        sidx++;
      }
      o0 = fma(R[2], P.colli_length[i], o0);
      o1 = fma(R[5], P.colli_length[i], o1);
      o2 = fma(R[8], P.colli_length[i], o2);
      const double cqi_ = mg_lds[i * 64];   // (read before the torque takes the word)
      // joint i turns frame i about its local z (even i) or y (odd i) axis through o_{i+1}
      const int ac = (i % 2 == 0) ? 2 : 1;
      const double ax = R[0 * 3 + ac], ay = R[1 * 3 + ac], az = R[2 * 3 + ac];
      const double tx = Mx - fma(o1, Fz, -(o2 * Fy));
      const double ty = My - fma(o2, Fx, -(o0 * Fz));
      const double tz = Mz - fma(o0, Fy, -(o1 * Fx));
      mg_lds[i * 64] = fma(az, tz, fma(ay, ty, ax * tx));
      joint_rotate(R, i, cqi_, sq[i]);
      TOPAY_SCHED_FENCE();
    }
  }
  MSTAMP(4);  // walks 2a/2b
  MMARK(4);
  // joint position limits — moma_traj_opt.cpp:1616-1666 (symmetric joint_pos_limit_max, reference quirk); the rare
  // contribution to a joint's entry is a read-modify-write of the lane's own LDS word
  const double wJP = P.s2_mani_pos_weight;
#pragma unroll
  for (int ji = 0; ji < 7; ji++) {
    double v = pos[ji + 3] - P.joint_pos_limit_max[ji];
    if (v > 0) {
      double pe, pd;
      smoothL1(P, v, mu, pe, pd);
      mg_lds[ji * 64] += w * wJP * pd;
      gdTk += omg * wJP * (pe * invK);
      cost += w * wJP * pe;
    }
    v = -P.joint_pos_limit_max[ji] - pos[ji + 3];
    if (v > 0) {
      double pe, pd;
      smoothL1(P, v, mu, pe, pd);
      mg_lds[ji * 64] -= w * wJP * pd;
      gdTk += omg * wJP * (pe * invK);
      cost += w * wJP * pe;
    }
  }
  MSTAMP(5);  // joint limits
  MMARK(5);
  out.gx = bFx;
  out.gy = bFy;
  out.gth = bMz;
  out.cost = cost;
  out.gdT = gdTk;
  return out;
}

// kinodynamic penalties shared by both stages — moma_traj_opt.cpp:1059-1115 / 1334-1462.
// th1,th2,th3 = theta', theta'', theta'''; s1..s3 likewise.  Adds to cost, gdT and the gradBeta entries
// (gth1 = d/d theta', gth2 = d/d theta'', gs1, gs2).
__device__ __forceinline__ void kinodynamic_block(dev_params_ref P, double wM, double wA, double wD, double omg,
                                                  double step, double real_alpha, double th1, double th2, double th3,
                                                  double sd1, double sd2, double sd3, double& cost, double& gdT,
                                                  double& gth1, double& gth2, double& gs1, double& gs2) {
  const double mu = P.relu_mu;
  const double max_v = P.max_v, max_w = P.max_w, max_a = P.max_a, max_dw = P.max_dw;
  const double w = omg * step;
#pragma unroll
  for (int sgn = -1; sgn <= 1; sgn += 2) {
    const double v = sgn * max_v * th1 + max_w * sd1 - P.max_vw;
    if (v > 0) {
      double pe, pd;
      smoothL1(P, v, mu, pe, pd);
      const double gt = real_alpha * (sgn * max_v * th2 + max_w * sd2);
      gth1 += w * wM * pd * sgn * max_v;
      gs1 += w * wM * pd * max_w;
      gdT += omg * wM * (pd * gt * step + pe * TOPAY_INV_K);
      cost += w * wM * pe;
    }
  }
#pragma unroll
  for (int sgn = -1; sgn <= 1; sgn += 2) {
    const double v = sgn * max_v * th1 - max_w * sd1 - P.max_vw;
    if (v > 0) {
      double pe, pd;
      smoothL1(P, v, mu, pe, pd);
      const double gt = real_alpha * (sgn * max_v * th2 - max_w * sd2);
      gth1 += w * wM * pd * sgn * max_v;
      gs1 -= w * wM * pd * max_w;
      gdT += omg * wM * (pd * gt * step + pe * TOPAY_INV_K);
      cost += w * wM * pe;
    }
  }
  const double vAcc = sd2 * sd2 - P.max_a2;
  const double vAlp = th2 * th2 - P.max_dw2;
  if (vAcc > 0) {
    double pe, pd;
    smoothL1(P, vAcc, mu, pe, pd);
    const double gt = 2.0 * real_alpha * sd2 * sd3;
    gs2 += w * wA * pd * 2.0 * sd2;
    gdT += omg * wA * (pd * gt * step + pe * TOPAY_INV_K);
    cost += w * wA * pe;
  }
  if (vAlp > 0) {
    double pe, pd;
    smoothL1(P, vAlp, mu, pe, pd);
    const double gt = 2.0 * real_alpha * th2 * th3;
    gth2 += w * wD * pd * 2.0 * th2;
    gdT += omg * wD * (pd * gt * step + pe * TOPAY_INV_K);
    cost += w * wD * pe;
  }
}

// k-th basis entries of order 0,1,2 at local time s (k = coefficient index of a row lane)
__device__ __forceinline__ void basis_k(int k, double s, double& b0, double& b1, double& b2) {
  // powers s^(k-2), s^(k-1), s^k with the convention s^negative -> unused (multiplied by 0)
  double pm2 = 1.0, pm1 = 1.0, p = 1.0;
#pragma unroll
  for (int t = 1; t <= 5; t++) {
    if (t <= k) p *= s;
    if (t <= k - 1) pm1 *= s;
    if (t <= k - 2) pm2 *= s;
  }
  b0 = p;
  b1 = (k >= 1) ? (double)k * pm1 : 0.0;
  b2 = (k >= 2) ? (double)(k * (k - 1)) * pm2 : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Body of one even Simpson sample (i, j) of sweep 1: kinodynamic penalties, and in stage 2 the chassis ESDF query, the
// manipulator block and the joint velocity / acceleration limits.  Outputs the per-sample gradient rows gB[12]
// (theta orders 0-2, s orders 1-2, joints order 0), the sample's dJ/dT part, its positional gradient and its cost.
// Shared by the one-wave and the several-waves evaluation: same arithmetic, same bits.
//
// Stage 2 is two functions around the call of the block, sample_mani() and sample_rest(): the caller forms the sample's
// geometry (piece, sample index, step, position) from the lane number before the call and AGAIN after it -- everything a
// lane holds in registers across the call is saved to and restored from scratch memory, so nothing but the loop's own
// state is kept (round 4: ~20 dwords of loop state and the 12-double return value per pass; rounds 1-3 parked 14 doubles per
// lane in an LDS pass buffer).  Every sum is formed in the order of before.
// ---------------------------------------------------------------------------------------------
template <int OCC>
__device__ __forceinline__ ManiOut sample_mani(lds_cdp cL, int rows, int i, int j, int e, bool act, double step, double half, double posx,
                                               double posy, const TOPAY_GLB DevMap* mp, glb_dp mstash, lds_dp mg_lds) {
  ManiOut mo_;
  // The call sits in a block of its own, entered by a SCALAR branch the compiler cannot fold: the register allocator
  // parks what is live across the call at the top of the call's block, and when that block is the join of a divergent
  // `if`, this image's compiler puts those copies ahead of the EXEC restore of the join (docs/EXPERIMENTS.md, "The hardware-only failures";
  // tools/isa_lint.py found it again in round 4 when the callee's smaller register need changed the caller's allocation).
  if (topay_opaque_true()) mo_ = manipulator_block<OCC>(mp, cL, rows, i, j, half, step, posx, posy, act ? e : -1, mstash, mg_lds);
  return mo_;
}

template <int STAGE>
__device__ __forceinline__ void sample_rest(dev_params_ref P, lds_cdp cL, int rows, int i, int j, double step, double half, double posx, double posy,
                                            const TOPAY_GLB DevMap* mp, double wM, double wA, double wD, const ManiOut& mo_, lds_cdp mg_lds,
                                            double (&gB)[12], double& gdTs, double& gpx, double& gpy, bool& jva, double& cst) {
#pragma unroll
  for (int v = 0; v < 12; v++) gB[v] = 0.0;
  gdTs = 0.0; gpx = 0.0; gpy = 0.0;
  jva = false;
  Basis B;
  make_basis(j * half, B);
  const double omg = (j == 0 || j == 2 * TOPAY_K) ? 0.5 : 1.0;
  const double real_alpha = 1.0 / TOPAY_K * ((double)j / 2.0);
  double th0, th1, th2, th3, s0, sd1, sd2, sd3;
  poly4(cL, rows, i, 0, B, th0, th1, th2, th3);
  poly4(cL, rows, i, 1, B, s0, sd1, sd2, sd3);
  cst = 0.0;
  kinodynamic_block(P, wM, wA, wD, omg, step, real_alpha, th1, th2, th3, sd1, sd2, sd3, cst, gdTs, gB[1], gB[2],
                    gB[3], gB[4]);
  if (STAGE == 2) {
    // chassis collision — moma_traj_opt.cpp:1304-1332
    double d2, g2x, g2y;
    {
      const DevMap M = load_map(mp);
      esdf2d_query(M, posx, posy, d2, g2x, g2y);
    }
    const double viola = P.chassis_r105 - d2;
    if (viola > 0) {
      double pe, pd;
      smoothL1(P, viola, P.relu_mu, pe, pd);
      const double sc = -omg * step * P.s2_collision_weight * pd;
      gpx += sc * g2x;
      gpy += sc * g2y;
      gdTs += omg * P.s2_collision_weight * (pe * TOPAY_INV_K);
      cst += omg * step * P.s2_collision_weight * pe;
    }
    // joint velocity / acceleration limits — moma_traj_opt.cpp:1674-1710 (cost and gdT here; the rare gradBeta rows are
    // produced in the follow-up round of the gradient phase when any lane is active), and moma_grad.tail(7) . dq
    double qacc = 0.0;
#pragma unroll
    for (int q = 0; q < 7; q++) {
      double a0, a1, a2, a3;
      poly4(cL, rows, i, 2 + q, B, a0, a1, a2, a3);
      const double vDq = a1 * a1 - P.joint_vel_limit2[q];
      const double vD2q = a2 * a2 - P.joint_acc_limit2[q];
      if (vDq > 0) {
        double pe, pd;
        smoothL1(P, vDq, P.relu_mu, pe, pd);
        gdTs += omg * P.s2_mani_vel_weight * (pd * (2.0 * real_alpha * a1 * a2) * step + pe * TOPAY_INV_K);
        cst += omg * step * P.s2_mani_vel_weight * pe;
        jva = true;
      }
      if (vD2q > 0) {
        double pe, pd;
        smoothL1(P, vD2q, P.relu_mu, pe, pd);
        gdTs += omg * P.s2_mani_acc_weight * (pd * (2.0 * real_alpha * a2 * a3) * step + pe * TOPAY_INV_K);
        cst += omg * step * P.s2_mani_acc_weight * pe;
        jva = true;
      }
      const double mgq = mg_lds[q * 64];   // the block's d/dq_q, from the lane's column of the pass buffer
      gB[5 + q] = mgq;                     // gradBeta row 0 of the joints (1671)
      qacc += mgq * a1;
      TOPAY_SCHED_FENCE();
    }
    cst = cst + mo_.cost;
    gdTs = gdTs + mo_.gdT;
    gpx += mo_.gx;
    gpy += mo_.gy;
    gB[0] = mo_.gth;                      // gdC(:, theta) += beta0 * moma_grad(2)   (1669)
    gdTs += mo_.gth * th1 * real_alpha;   // (1670)
    gdTs += qacc * real_alpha;            // (1672)
  }
}

}  // namespace topay
