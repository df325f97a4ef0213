// ============================================================================
// TEST INFRASTRUCTURE — CPU wave emulator for the HIP kernel sources.
//
// Compiling topay_amd/csrc/*.hip with `g++ -x c++ -I tests/emu/include` picks up THIS file
// instead of the real <hip/hip_runtime.h>.  It runs each workgroup's lanes as ucontext fibers
// on one OS thread so the unmodified kernel source (wave-uniform collectives only) can be
// checked against the oracle in the `-m "not gpu"` suite and under ASan/UBSan.
//
// It is NOT a fallback: the product (topay_amd/api.py) loads only libtopay_hip.so built by hipcc
// and fails loudly when that is missing.  Nothing under topay_amd/ references this directory.
//
// Rules the kernels follow so that emulation is faithful: __syncthreads / __shfl* / __ballot /
// readlane are only executed in wave-uniform control flow (also required for defined behaviour
// on the GPU for barriers).
// ============================================================================
#pragma once
#include <ucontext.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>

#define TOPAY_CPU_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__
#define HIP_SYMBOL(x) (&(x))
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

namespace hip_emu {
struct Lane {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  int wait_kind = 0;   // 0 runnable, 1 waiting for its wave's barrier generation to pass wait_gen, 2 for the block's
  int wait_gen = 0;
};
struct State {
  uint3_emu tid{0, 0, 0}, bid{0, 0, 0};
  dim3 bdim, gdim;
  std::vector<Lane> lanes;
  int cur = 0;
  int nl = 0;          // fibres of the running block
  ucontext_t sched;
  // exchange buffers for collectives: 2 phases x lanes x 16 bytes
  std::vector<unsigned char> xbuf;
  std::vector<int> phase;  // per lane
  // barriers: a wave-level one per wavefront of 64 fibres (collectives, wave_barrier) and the block's (__syncthreads,
  // s_barrier) -- a workgroup of several waves really runs its waves out of step between two block barriers
  std::vector<int> wave_cnt, wave_gen;
  int blk_cnt = 0, blk_gen = 0;
  unsigned char* dyn_smem = nullptr;
  std::function<void()> body;
  long barriers = 0;
};
inline State& S() {
  static State s;
  return s;
}
inline void lane_entry() {
  State& s = S();
  s.body();
  s.lanes[s.cur].done = true;
  swapcontext(&s.lanes[s.cur].ctx, &s.sched);
}
// Barrier of the 64 fibres of the calling fibre's wavefront.  The last fibre to arrive goes on; the others are parked
// until the wave's generation counter has moved (the scheduler does not switch to a parked fibre).
inline void wave_barrier() {
  State& s = S();
  s.barriers++;
  const int me = s.cur, w = me >> 6;
  const int wsize = std::min(64, s.nl - 64 * w);
  if (++s.wave_cnt[w] == wsize) {
    s.wave_cnt[w] = 0;
    s.wave_gen[w]++;
    return;
  }
  s.lanes[me].wait_kind = 1;
  s.lanes[me].wait_gen = s.wave_gen[w];
  swapcontext(&s.lanes[me].ctx, &s.sched);
}
// Barrier of the whole block.
inline void barrier() {
  State& s = S();
  s.barriers++;
  const int me = s.cur;
  if (++s.blk_cnt == s.nl) {
    s.blk_cnt = 0;
    s.blk_gen++;
    return;
  }
  s.lanes[me].wait_kind = 2;
  s.lanes[me].wait_gen = s.blk_gen;
  swapcontext(&s.lanes[me].ctx, &s.sched);
}
inline void run_block(unsigned bx, size_t shmem_bytes, const std::function<void()>& body) {
  State& s = S();
  const int nl = (int)(s.bdim.x * s.bdim.y * s.bdim.z);
  s.nl = nl;
  s.bid = {bx, 0, 0};
  s.body = body;
  static const size_t STK = 1 << 20;
  if ((int)s.lanes.size() < nl) s.lanes.resize(nl);
  s.xbuf.assign((size_t)2 * nl * 16, 0);
  s.phase.assign(nl, 0);
  s.wave_cnt.assign((nl + 63) / 64, 0);
  s.wave_gen.assign((nl + 63) / 64, 0);
  s.blk_cnt = 0;
  s.blk_gen = 0;
  std::vector<unsigned char> smem(shmem_bytes + 64, 0);
  s.dyn_smem = smem.data();
  for (int l = 0; l < nl; l++) {
    Lane& L = s.lanes[l];
    if (!L.stack) L.stack = (char*)malloc(STK);
    L.done = false;
    L.wait_kind = 0;
    getcontext(&L.ctx);
    L.ctx.uc_stack.ss_sp = L.stack;
    L.ctx.uc_stack.ss_size = STK;
    L.ctx.uc_link = &s.sched;
    makecontext(&L.ctx, (void (*)())lane_entry, 0);
  }
  int remaining = nl;
  while (remaining > 0) {
    int ran = 0;
    for (int l = 0; l < nl; l++) {
      Lane& L = s.lanes[l];
      if (L.done) continue;
      if (L.wait_kind == 1 && s.wave_gen[l >> 6] == L.wait_gen) continue;
      if (L.wait_kind == 2 && s.blk_gen == L.wait_gen) continue;
      L.wait_kind = 0;
      s.cur = l;
      s.tid = {(unsigned)l % s.bdim.x, ((unsigned)l / s.bdim.x) % s.bdim.y, (unsigned)l / (s.bdim.x * s.bdim.y)};
      swapcontext(&s.sched, &L.ctx);
      ran++;
      if (L.done) remaining--;
    }
    if (ran == 0) {   // every live fibre waits for a barrier that cannot complete: some fibres skipped it or exited early
      fprintf(stderr, "[hip_emu] deadlock in block %u: %d fibres wait at a barrier the others never reach\n", bx, remaining);
      abort();
    }
  }
}
template <typename T>
inline T exchange(T v, int src, bool valid_src = true) {
  static_assert(sizeof(T) <= 16, "exchange payload");
  State& s = S();
  const int nl = (int)s.lanes.size();
  (void)nl;
  int me = s.cur;
  int ph = s.phase[me];
  s.phase[me] ^= 1;
  const int lanes = (int)(s.bdim.x * s.bdim.y * s.bdim.z);
  unsigned char* base = s.xbuf.data() + (size_t)ph * lanes * 16;
  memcpy(base + (size_t)me * 16, &v, sizeof(T));
  wave_barrier();
  T r = v;
  if (valid_src && src >= 0 && src < lanes) memcpy(&r, base + (size_t)src * 16, sizeof(T));
  return r;
}
}  // namespace hip_emu

#define threadIdx (hip_emu::S().tid)
#define blockIdx (hip_emu::S().bid)
#define blockDim (hip_emu::S().bdim)
#define gridDim (hip_emu::S().gdim)
#define warpSize 64
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hip_emu::S().dyn_smem;

inline void __syncthreads() { hip_emu::barrier(); }
inline int __lane_id_emu() { return (int)(hip_emu::S().cur % 64); }

template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  int me = hip_emu::S().cur;
  int base = me & ~(width - 1);
  return hip_emu::exchange(v, base + (src & (width - 1)));
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  int me = hip_emu::S().cur;
  return hip_emu::exchange(v, me ^ mask);
}
template <typename T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {
  int me = hip_emu::S().cur;
  int lane = me & (width - 1);
  return hip_emu::exchange(v, me - (int)delta, lane >= (int)delta);
}
template <typename T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
  int me = hip_emu::S().cur;
  int lane = me & (width - 1);
  return hip_emu::exchange(v, me + (int)delta, lane + (int)delta < width);
}
inline unsigned long long __ballot(int pred) {
  // gather predicates of the 64 lanes of this wave
  int me = hip_emu::S().cur;
  int wbase = me & ~63;
  unsigned long long bits = 0;
  // one exchange round: everybody publishes, then reads all 64
  hip_emu::State& s = hip_emu::S();
  int ph = s.phase[me];
  s.phase[me] ^= 1;
  const int lanes = (int)(s.bdim.x * s.bdim.y * s.bdim.z);
  unsigned char* base = s.xbuf.data() + (size_t)ph * lanes * 16;
  int p = pred ? 1 : 0;
  memcpy(base + (size_t)me * 16, &p, sizeof(int));
  hip_emu::wave_barrier();
  for (int l = 0; l < 64 && wbase + l < lanes; l++) {
    int q;
    memcpy(&q, base + (size_t)(wbase + l) * 16, sizeof(int));
    if (q) bits |= (1ull << l);
  }
  return bits;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }  // lanes are fibers of one thread
inline int atomicOr(int* p, int v) { int o = *p; *p |= v; return o; }
inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
inline int __all(int pred) {
  int lanes = (int)(hip_emu::S().bdim.x);
  unsigned long long full = lanes >= 64 ? ~0ull : ((1ull << lanes) - 1);
  return (__ballot(pred) & full) == full;
}
// The kernels apply readfirstlane only to values that are already wave-uniform (to move them to scalar registers), also
// under divergent control flow where a lane-exchange would not be collective: identity is the faithful emulation.
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
inline int __builtin_amdgcn_readlane(int v, int lane) { return hip_emu::exchange(v, (hip_emu::S().cur & ~63) + lane); }
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  (void)bank_mask; (void)bound_ctrl;
  const int me = hip_emu::S().cur, l = me & 15, base = me & ~15;
  int from;
  if (ctrl == 0x142 || ctrl == 0x143) {   // row_bcast15 / row_bcast31: lane 15 of the previous row / lane 31 into rows 2-3
    const int row = (me >> 4) & 3;
    const bool take = ((row_mask >> row) & 1) && (ctrl == 0x142 ? row >= 1 : row >= 2);
    const int src_lane = ctrl == 0x142 ? (row >= 1 ? base - 1 : me) : (me & ~63) + 31;
    const int got = hip_emu::exchange(src, src_lane);
    return take ? got : old;
  }
  (void)old; (void)row_mask;
  switch (ctrl) {
    case 0xB1: from = base + (l ^ 1); break;                     // quad_perm [1,0,3,2]
    case 0x4E: from = base + (l ^ 2); break;                     // quad_perm [2,3,0,1]
    case 0x141: from = base + ((l & 8) | (7 - (l & 7))); break;  // row_half_mirror
    case 0x140: from = base + (15 - l); break;                   // row_mirror
    default: fprintf(stderr, "[hip_emu] unsupported dpp ctrl 0x%x\n", ctrl); abort();
  }
  return hip_emu::exchange(src, from);
}
inline void __builtin_amdgcn_wave_barrier() { hip_emu::wave_barrier(); }
inline void __builtin_amdgcn_s_barrier() { hip_emu::barrier(); }
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
inline unsigned long long wall_clock64() { return 0; }
inline unsigned long long __builtin_amdgcn_s_memtime() { return (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count(); }
inline int __double2loint(double v) { long long b; memcpy(&b, &v, 8); return (int)(b & 0xffffffffLL); }
inline double __longlong_as_double(long long b) { double v; memcpy(&v, &b, 8); return v; }
inline int __double2hiint(double v) { long long b; memcpy(&b, &v, 8); return (int)((b >> 32) & 0xffffffffLL); }
inline double __hiloint2double(int hi, int lo) {
  unsigned long long b = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
  double v; memcpy(&v, &b, 8); return v;
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
using std::isinf;
using std::isnan;
using std::max;
using std::min;

// ---- host runtime shims --------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct hipEvent_emu { std::chrono::steady_clock::time_point t; };
typedef hipEvent_emu* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
inline const char* hipGetErrorString(hipError_t) { return "hip_emu"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)calloc(n ? n : 1, 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipMemcpyToSymbolAsync(void* sym, const void* src, size_t n, size_t off, hipMemcpyKind, hipStream_t) {
  memcpy((char*)sym + off, src, n);
  return 0;
}
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
#define hipStreamNonBlocking 1
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return 0; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
enum { hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return 0; }
enum { hipErrorNotReady = 600 };
inline hipError_t hipStreamQuery(hipStream_t) { return 0; }
struct hipDeviceProp_t { int multiProcessorCount = 2; };   // two 'CUs': the persistent launches of the tests really loop
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { *p = hipDeviceProp_t(); return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEvent_emu(); return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return 0;
}
template <typename F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return 0; }

template <typename K, typename... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
  hip_emu::State& s = hip_emu::S();
  s.bdim = block;
  s.gdim = grid;
  for (unsigned b = 0; b < grid.x; b++) hip_emu::run_block(b, shmem, [&]() { kernel(args...); });
}
