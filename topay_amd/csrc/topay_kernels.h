// Device side of libtopay_hip.so: the solve / evaluation kernels (one workgroup per trajectory, persistent queues) and the
// small kernels around them.  Included by topay_hip.hip (the host side of the C-ABI) and by the CPU lane-emulator build.
#pragma once
#include <hip/hip_runtime.h>

#include "topay_solve.h"
#include "topay_feas.h"
#include "topay_edt.h"
#include "topay_front.h"
#include "topay_mcrrt.h"
#include "topay_jps.h"

// Waves per SIMD a kernel is built for (OCC): the register allocator leaves room for that many -- 512 / OCC registers
// (VGPR + AGPR) per lane.  Since round 4 every solve / evaluation kernel is built for OCC = 2: 256 registers, which the
// allocator takes as 256 VGPRs and no AGPRs.  The serial chains of a solve (banded LU, substitutions, hand-offs) overlap
// with the second resident wave; the sample body and the two-loop recursion are bound by VALU issue and do not
// (DESIGN.md section 9).  Every device function of the call graph has to fit, so the non-inlined ones (manipulator_block,
// minco_generate_mw, eval_cost_grad_mw, the in-solve gate) are templated on OCC and the kernel's attribute reaches them per
// instantiation.
using namespace topay;

#ifndef TOPAY_CPU_EMU
extern __shared__ double topay_lds[];
#define TOPAY_LDS_PTR ((lds_dp)topay_lds)
#else
#define TOPAY_LDS_PTR ((lds_dp)hip_emu::S().dyn_smem)
#endif

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_init(DevBatch Bt, const double* paths, const long long* path_off, const int* path_len,
                       const double* bvel, const double* bacc, double* scratch, int scratch_stride, int maxN, int stride_n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bt.B) return;
  init_one(g_P, paths + path_off[b] * 10, path_len[b], bvel + (size_t)b * 20, bacc + (size_t)b * 20,
           scratch + (size_t)b * scratch_stride, maxN, Bt.N + b, Bt.s1_past + b, Bt.head + (size_t)b * 27,
           Bt.tail + (size_t)b * 27, Bt.start_xy + 2 * b, Bt.goal_xy + 2 * b, Bt.init_xy + (size_t)b * 2 * maxN,
           Bt.x0 + (size_t)b * stride_n);
}

// Where candidate b's variable-length blocks start: every per-candidate array is packed by the candidate's own size
// (pieces before it: poff, decision-vector elements before it: noff), not strided by the longest member of the batch.
__device__ __forceinline__ long long uniform_i64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// LDS of one trajectory's workgroup: the evaluation's blocks, then [8] past costs and [40] solver state parked across an evaluation
template <int NW>
__host__ __device__ __forceinline__ int eval_lds_doubles(int Nmax_lds) {
  return lds_doubles_mw(Nmax_lds, NW);
}

template <int RMAX, int NW>
__device__ __forceinline__ void load_ctx(EvalCtx& C, const DevBatch& Bt, int b, int Nmax_lds) {
  constexpr int NT = 64 * NW;
  C.tid = threadIdx.x;
  C.lane = threadIdx.x & 63;
  C.wave = threadIdx.x >> 6;
  C.N = __builtin_amdgcn_readfirstlane(Bt.N[b]);  // wave-uniform: keep it (and what derives from it) in scalar registers
  C.rows = 6 * C.N;
  C.n = 10 * C.N - 8;
  C.red = nullptr; C.adj = nullptr; C.cl_in_lds = 1;
  carve_mw(C, TOPAY_LDS_PTR, Nmax_lds, NW);
  fill_power_table(C.pw, C.lane);
  C.hd = (glb_cdp)(Bt.head + (size_t)b * 27);
  C.tl = (glb_cdp)(Bt.tail + (size_t)b * 27);
  const long long po = uniform_i64(Bt.poff[b]);
  C.lu = (glb_dp)(Bt.lu + 84 * po);
  C.sb_stride = TOPAY_EP * C.N;
  C.sbuf = (glb_dp)(Bt.sbuf + 14 * TOPAY_EP * po);
  C.mstash = (glb_dp)(Bt.mstash + 36 * TOPAY_EP * po);
  C.coefg = (glb_dp)(Bt.coef + 54 * po);
  C.init_xy = (glb_cdp)(Bt.init_xy + (size_t)b * 2 * TOPAY_MAX_N);
  C.sx = Bt.start_xy[2 * b]; C.sy = Bt.start_xy[2 * b + 1];
  C.ex = Bt.goal_xy[2 * b];  C.ey = Bt.goal_xy[2 * b + 1];
  C.fxe0 = 0.0; C.fxe1 = 0.0;
  C.stamps = nullptr;
  C.t_last = 0;
#ifdef TOPAY_STAMPS
  // diagnostic build: the trace buffer (topay_set_trace with cap >= 32) doubles as the stamp accumulator
  if (Bt.trace && Bt.trace_cap >= 32) C.stamps = (TOPAY_GLB long long*)(Bt.trace + (size_t)b * Bt.trace_cap) + 8;
  C.t_last = (long long)__builtin_amdgcn_s_memtime();
#endif
}

// getTraj() state of the last evaluation (moma_traj_opt.h:943-946) into the candidate's result blocks
template <int NW>
__device__ __forceinline__ void store_result(const EvalCtx& C, const DevBatch& Bt, int b) {
  constexpr int NT = 64 * NW;
  const int N = C.N, rows = C.rows;
  const long long po = uniform_i64(Bt.poff[b]);
  // (after a gradient phase the coefficients already sit in the result block, C.cL holds the adjoint)
  if (C.cl_in_lds) {
    double* coef = Bt.coef + 54 * po;
    for (int t = C.tid; t < 9 * rows; t += NT) coef[t] = C.cL[t];
  }
  for (int t = C.tid; t < N; t += NT) Bt.T[po + t] = C.Tp[t];
  double* kn = Bt.knots + 2 * (po + b);
  if (C.tid == 0) { kn[0] = C.sx; kn[1] = C.sy; }
  for (int t = C.tid; t < 2 * N; t += NT) kn[2 + t] = C.pcs[2 * N + 2 + t];
}

// test hook: one cost/gradient evaluation of trajectory order[blockIdx] at Bt.x with ALM state Bt.alm
template <int RMAX, int NW, int OCC>
__device__ __forceinline__ void eval_body(const DevBatch& Bt, const DevMap* maps, int stage, int Nmax_lds, int repeats) {
  const int b = Bt.order[blockIdx.x];
  const bool commit = (stage & 16) != 0;
  stage &= 15;
  EvalCtx C;
  load_ctx<RMAX, NW>(C, Bt, b, Nmax_lds);
  const TOPAY_GLB DevMap* mp = (const TOPAY_GLB DevMap*)(maps + __builtin_amdgcn_readfirstlane(Bt.map_id[b]));
  const long long no = uniform_i64(Bt.noff[b]);
  C.x = (glb_cdp)(Bt.x + no);
  C.g = (glb_dp)(Bt.work + 4 * no);
  C.lam0 = Bt.alm[4 * b]; C.lam1 = Bt.alm[4 * b + 1]; C.rho0 = Bt.alm[4 * b + 2]; C.rho1 = Bt.alm[4 * b + 3];
  __syncthreads();
  double f = 0.0;
  // (negative repeats: cost only -- the gate then answers "gradient not needed", as for a rejected line-search trial)
  const bool cost_only = repeats < 0;
  if (cost_only) repeats = -repeats;
  for (int r = 0; r < repeats; r++) {
    GradGate gate;
    gate.always = !cost_only; gate.has_early = false; gate.finit = 0.0; gate.thr = -1.0e300; gate.early = 0.0;
    gate.early_ok = false; gate.skip_thr = 0.0;
    __syncthreads();
    if (stage == 1) f = eval_cost_grad_mw<1, RMAX, NW, OCC>(C, mp, gate);
    else f = eval_cost_grad_mw<2, RMAX, NW, OCC>(C, mp, gate);
  }
  if (C.tid == 0) {
    Bt.fout[b] = f;
    Bt.xyerr[2 * b] = C.fxe0;
    Bt.xyerr[2 * b + 1] = C.fxe1;
  }
  if (commit) {   // topay_load_solution: the spline of this x becomes the candidate's result, as after a solve that ended here
    __syncthreads();
    store_result<NW>(C, Bt, b);
    if (C.tid == 0) { Bt.cost[b] = f; Bt.success[b] = 1; }
  }
}

// The gate inside the solve kernel is a call: inlined, its 9 000 instructions and their live ranges became part of the
// solver's register allocation (161 instead of 33 spilled VGPRs in k_solve1).
template <int OCC>
__device__ __noinline__ void feasibility_gate_in_solve(const FeasIO F, const TOPAY_GLB DevMap* mp) { feasibility_gate(F, mp); }

// NWE = waves of the workgroup (= waves of an evaluation), NW = waves the solver runs on: NWE > NW = 1 is the helper-wave
// scheme of the k_lat* kernels (topay_solve.h).
template <int RMAX, int NW, int OCC, int NWE = NW, int RMAX_E = RMAX>
__device__ __forceinline__ void solve_one(const DevBatch& Bt, const DevMap* maps, int Nmax_lds, int b) {
  constexpr int NT = 64 * NWE;
  const unsigned long long t_begin = wall_clock64();
  // scheduling only (never read by the solve): lets the host issue the next batch once every candidate of this one
  // is resident, see topay_optimize_async
  if (threadIdx.x == 0 && Bt.started && Bt.N[b] <= Bt.gate_maxN) {
#ifndef TOPAY_CPU_EMU
    __hip_atomic_fetch_add(Bt.started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    Bt.started[0] += 1;
#endif
  }
  EvalCtx C;
  load_ctx<RMAX_E, NWE>(C, Bt, b, Nmax_lds);
  const TOPAY_GLB DevMap* mp = (const TOPAY_GLB DevMap*)(maps + __builtin_amdgcn_readfirstlane(Bt.map_id[b]));
  lds_dp pf = TOPAY_LDS_PTR + eval_lds_doubles<NWE>(Nmax_lds);  // [8] past costs, then [40] solver state parked across an evaluation (+ the command block of the helper-wave kernels)
  const long long no = uniform_i64(Bt.noff[b]);
  const int n = C.n;
  SolveIO S;
  S.x = (glb_dp)(Bt.x + no);
  S.g = (glb_dp)(Bt.work + 4 * no);
  S.xp = (glb_dp)(Bt.work + 4 * no + n);
  S.gp = (glb_dp)(Bt.work + 4 * no + 2 * (long long)n);
  S.d = (glb_dp)(Bt.work + 4 * no + 3 * (long long)n);
  S.hist_s = (glb_dp)(Bt.hist_s + (long long)Bt.hist_m * no);
  S.hist_y = (glb_dp)(Bt.hist_y + (long long)Bt.hist_m * no);
  S.hist_ys = (glb_dp)(Bt.hist_ys + (size_t)b * Bt.hist_m);
  S.hist_al = (glb_dp)(Bt.hist_alpha + (size_t)b * Bt.hist_m);
  S.nstride = n;
  S.stats = (glb_ip)(Bt.stats + (size_t)b * 8);
  S.trace = Bt.trace ? (glb_dp)(Bt.trace + (size_t)b * Bt.trace_cap) : (glb_dp)nullptr;
  S.trace_cap = Bt.trace_cap;
  const int grp = Bt.group_id ? __builtin_amdgcn_readfirstlane(Bt.group_id[b]) : -1;
  S.grp_tau = (grp >= 0 && Bt.cancel_budget > 0) ? (TOPAY_GLB int*)(Bt.group_tau + grp) : (TOPAY_GLB int*)nullptr;
  S.cancel_flag = (const TOPAY_GLB int*)Bt.cancel_flag;
  S.cancel_budget = Bt.cancel_budget;
  // x <- x0
  {
    const double* x0 = Bt.x0 + (size_t)b * (10 * TOPAY_MAX_N - 8);
    for (int e = C.tid; e < C.n; e += NT) S.x[e] = x0[e];
  }
  int success = 0, interrupted = 0;
  double cost = 0.0;
  if constexpr (NWE != NW) {
    C.x = S.x;
    C.g = S.g;
    __syncthreads();   // x0 is in place for wave 0
    if (C.wave == 0) solve_trajectory<RMAX, NW, OCC, NWE, RMAX_E>(C, mp, S, Bt.s1_past[b], pf, success, cost, interrupted);
    else eval_helper_loop<RMAX_E, NWE, OCC>(C, mp, pf + 48);
  } else {
    solve_trajectory<RMAX, NW, OCC>(C, mp, S, Bt.s1_past[b], pf, success, cost, interrupted);
  }
  // results: state of the last evaluation (getTraj(), moma_traj_opt.h:943-946) + traj_cost
  __syncthreads();
  store_result<NWE>(C, Bt, b);
  if (Bt.gate_in_solve) {
    // printConstraintsSituations of the returned trajectory (planner.cpp:878-880) by wave 0, from the result blocks just
    // written; panels and sample times go to the candidate's L-BFGS history blocks, which are dead now
    __syncthreads();
    int* fl = Bt.feas_flags + 2 * b;
    if (interrupted) {
      if (C.tid == 0) { fl[0] = 0; fl[1] = 0; }
    } else if (threadIdx.x < 64) {
      const long long po = uniform_i64(Bt.poff[b]);
      const long long hist_doubles = (long long)Bt.hist_m * C.n;
      FeasIO F;
      F.coef = Bt.coef + 54 * po;
      F.T = Bt.T + po;
      F.N = C.N;
      F.x0 = C.sx; F.y0 = C.sy;
      F.th0 = Bt.head[(size_t)b * 27];
      F.cseq = Bt.hist_s + (long long)Bt.hist_m * no;
      F.tk = Bt.hist_y + (long long)Bt.hist_m * no;
      F.cap_panels = hist_doubles / 2 - 1;
      F.cap_samples = hist_doubles;
      F.report = Bt.feas_report + (size_t)b * 38;
      F.feasible = fl;
      F.truncated = Bt.gate_truncated;
      feasibility_gate_in_solve<OCC>(F, mp);
      // first feasible success of its planning call: its work clock opens the 100 ms (cancel_budget) window of the others
      if (S.grp_tau && success) {
        wave_global_sync();
        if (C.tid == 0 && fl[0]) {
          const int clock = (S.stats[2] + S.stats[5]) * C.N;
          atomicMin((int*)S.grp_tau, clock);
        }
      }
    }
    __syncthreads();
  }
  if (C.tid == 0) {
    if (Bt.interrupted) Bt.interrupted[b] = interrupted;
    Bt.success[b] = success;
    Bt.cost[b] = cost;
    Bt.xyerr[2 * b] = C.fxe0;
    Bt.xyerr[2 * b + 1] = C.fxe1;
    Bt.alm[4 * b] = C.lam0; Bt.alm[4 * b + 1] = C.lam1; Bt.alm[4 * b + 2] = C.rho0; Bt.alm[4 * b + 3] = C.rho1;
    Bt.elapsed_us[b] = (double)(wall_clock64() - t_begin) * 0.01;
    Bt.start_us[b] = (double)t_begin * 0.01;
#ifndef TOPAY_CPU_EMU
    {
      const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
      const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11));
      Bt.hw_id[b] = (int)(((xcc & 0xF) << 16) | (((hw >> 13) & 0x7) << 12) | (((hw >> 8) & 0xF) << 4) | ((hw >> 4) & 0x3));
    }
#else
    Bt.hw_id[b] = 0;
#endif
  }
}

// The queues of one batch, own class first, then the smaller ones (see DevBatch::queue_next).  Without queues
// (queue_next null: one workgroup per position of `order`) the loop body runs once, for order[blockIdx.x]: one call site
// of the solve for both launch schemes, i.e. one copy of the solver in the kernel.
template <int RMAX, int NW, int OCC, int NWE = NW, int RMAX_E = RMAX>
__device__ __forceinline__ void drain_queues(const DevBatch& B, const DevMap* maps, int Nmax_lds, int my_class) {
  const bool queued = B.queue_next != nullptr;
  const int lowest = queued ? B.queue_lowest : my_class;
  for (int cls = my_class; cls >= lowest; cls--) {
    const int count = queued ? B.queue_count[cls] : 1, off = queued ? B.queue_off[cls] : (int)blockIdx.x;
    for (int once = 0;; once++) {
      // thread 0 takes the next position and looks at topay_cancel's flag (threads.interrupt_all(): a candidate that has not
      // begun when the flag is up is not begun -- a batch larger than the resident grid would otherwise run the whole of
      // stage 1 of every queued candidate after the deadline of topay_optimize_within); bit 30 carries the verdict
      int pos = 0;
      if (threadIdx.x == 0) {
        pos = queued ? atomicAdd(B.queue_next + cls, 1) : once;
        int stop = 0;
#ifndef TOPAY_CPU_EMU
        if (B.cancel_flag) stop = __hip_atomic_load(B.cancel_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
#else
        if (B.cancel_flag) stop = *B.cancel_flag != 0;
#endif
        pos = (pos < 0x40000000 ? pos : 0x3fffffff) | (stop << 30);
      }
      if (NWE == 1) {
        pos = __shfl(pos, 0);
      } else {   // the position travels to the other waves through the first LDS word (nothing of a solve is live here)
        TOPAY_LDS int* w0 = (TOPAY_LDS int*)TOPAY_LDS_PTR;
        if (threadIdx.x == 0) w0[0] = pos;
        __syncthreads();
        pos = w0[0];
        __syncthreads();
      }
      const bool cancelled = (pos & 0x40000000) != 0;
      pos &= 0x3fffffff;
      if (cancelled && pos < count) {   // interrupted before its first evaluation: no trajectory, success 0, verdicts 0 / 0
        if (threadIdx.x == 0) {
          const int b = B.order[off + pos];
          int* st = B.stats + (size_t)b * 8;
          for (int q = 0; q < 8; q++) st[q] = 0;
          st[3] = TOPAY_INTERRUPTED;
          if (B.interrupted) B.interrupted[b] = 1;
          B.success[b] = 0;
          if (B.gate_in_solve) { B.feas_flags[2 * b] = 0; B.feas_flags[2 * b + 1] = 0; }
        }
        continue;
      }
      if (pos >= count) break;
      solve_one<RMAX, NW, OCC, NWE, RMAX_E>(B, maps, Nmax_lds, B.order[off + pos]);
      __syncthreads();
    }
  }
}

// Persistent launch: the grid is one workgroup per SIMD slot (or fewer; a workgroup of NW waves takes NW slots), and
// every workgroup takes candidates from the launch's queue -- positions of `order`, longest first -- until it is empty.
// The hardware dispatcher places workgroups in order on a fixed round-robin of XCDs / shader engines and stalls on a
// full one while others have room (about 10 % of the slots stay empty when it has to place 8000 workgroups of unequal
// length); a resident workgroup that fetches its next candidate itself leaves no slot idle and starts candidates
// strictly in queue order.  Which workgroup solves which candidate is timing-dependent, the result of a candidate is
// not (nothing is shared between candidates).
template <int RMAX, int NW, int OCC, int NWE = NW, int RMAX_E = RMAX>
__device__ __forceinline__ void solve_body(const DevBatch& Bt, const DevMap* maps, int Nmax_lds) {
  drain_queues<RMAX, NW, OCC, NWE, RMAX_E>(Bt, maps, Nmax_lds, Bt.queue_class);   // the batch is the kernel argument (scalar loads, no copy)
}

// One wave per trajectory: k_solve<rows per lane> for N <= 10 / 21 / 32, built for two waves per SIMD (256 registers, no
// AGPRs: besides the occupancy, this keeps the register allocator from parking values in AGPRs across the calls, the
// copies this image's compiler misplaces -- tools/isa_lint.py).  Several waves per trajectory (topay_eval_mw.h):
// k_solve<rows per thread>w<waves>, rows <= 64 x waves x rows per thread, one wave per SIMD.
#define TOPAY_SOLVE_KERNEL(NAME, R, W, OCC)                                                                         \
  __global__ void __launch_bounds__(64 * W, OCC) NAME(DevBatch Bt, const DevMap* maps, int Nmax_lds) { \
    solve_body<R, W, OCC>(Bt, maps, Nmax_lds);                                                             \
  }
// Helper-wave kernels for a handful of candidates (a planning call on an otherwise idle device): the one-wave solver of
// rows-per-lane RS on wave 0, evaluations on WE waves with RE rows per thread.  Results are those of k_solve<RS>, bit for bit.
#define TOPAY_LATENCY_KERNEL(NAME, RS, RE, WE, OCC)                                                        \
  __global__ void __launch_bounds__(64 * WE, OCC) NAME(DevBatch Bt, const DevMap* maps, int Nmax_lds) {    \
    solve_body<RS, 1, OCC, WE, RE>(Bt, maps, Nmax_lds);                                                    \
  }
#define TOPAY_EVAL_KERNEL(NAME, R, W, OCC)                                                                          \
  __global__ void __launch_bounds__(64 * W, OCC) NAME(DevBatch Bt, const DevMap* maps, int stage, int repeats, int Nmax_lds) { \
    eval_body<R, W, OCC>(Bt, maps, stage, Nmax_lds, repeats);                                              \
  }
#ifndef TOPAY_NO_KERNEL_TABLE   // (tools: a probe that instantiates one kernel of its own)
TOPAY_SOLVE_KERNEL(k_solve1, 1, 1, 2)
TOPAY_SOLVE_KERNEL(k_solve2, 2, 1, 2)
TOPAY_SOLVE_KERNEL(k_solve3, 3, 1, 2)
// the long classes (N = 33..64 / 65..170): one-wave solver with 10 / 28 vector elements per lane on wave 0, evaluations on
// four waves with 2 / 4 system rows per thread
TOPAY_LATENCY_KERNEL(k_long5, 5, 2, 4, 2)
TOPAY_LATENCY_KERNEL(k_long14, 14, 4, 4, 2)
TOPAY_LATENCY_KERNEL(k_lat1, 1, 1, 4, 2)
TOPAY_LATENCY_KERNEL(k_lat2, 2, 1, 4, 2)
TOPAY_LATENCY_KERNEL(k_lat3, 3, 1, 4, 2)
TOPAY_EVAL_KERNEL(k_eval1, 1, 1, 2)
TOPAY_EVAL_KERNEL(k_eval2, 2, 1, 2)
TOPAY_EVAL_KERNEL(k_eval3, 3, 1, 2)
TOPAY_EVAL_KERNEL(k_eval2w2, 2, 2, 2)
TOPAY_EVAL_KERNEL(k_eval3w2, 3, 2, 2)
TOPAY_EVAL_KERNEL(k_eval3w4, 3, 4, 2)
TOPAY_EVAL_KERNEL(k_eval4w4, 4, 4, 2)
// evaluation only (test hook topay_eval_waves: one wave for N = 33..64, four waves for N <= 85 -- the references of the
// order-identity tests of the several-waves evaluation)
TOPAY_EVAL_KERNEL(k_eval4, 4, 1, 2)
TOPAY_EVAL_KERNEL(k_eval6, 6, 1, 2)
TOPAY_EVAL_KERNEL(k_eval2w4, 2, 4, 2)
#ifdef TOPAY_EXPERIMENTS
// A/B variants (tools/ab_lib.sh builds with -DTOPAY_EXPERIMENTS): the one-wave kernels of the long classes and four waves
// for N <= 64
TOPAY_SOLVE_KERNEL(k_solve2w4, 2, 4, 2)
TOPAY_SOLVE_KERNEL(k_solve3w4, 3, 4, 2)
TOPAY_SOLVE_KERNEL(k_solve4, 4, 1, 2)
TOPAY_SOLVE_KERNEL(k_solve6, 6, 1, 2)
TOPAY_SOLVE_KERNEL(k_solve2w2, 2, 2, 2)
TOPAY_SOLVE_KERNEL(k_solve3w2, 3, 2, 2)
#endif
#endif  // TOPAY_NO_KERNEL_TABLE

// feasibility gate (printConstraintsSituations / checkFeasible) of every candidate's returned trajectory
// (two waves per SIMD like the solve kernels: 256 registers, no AGPRs -- with 512 the allocator parked values in AGPRs, whose
// copies are what this image's compiler can misplace at a control-flow join, tools/isa_lint.py)
__global__ void __launch_bounds__(64, 2) k_feasible(DevBatch Bt, const DevMap* maps, double* cseq, double* tk, long long cap_panels,
                                                 long long cap_samples, double* report, int* flags) {
  const int b = blockIdx.x;
  const int N = Bt.N[b];
  if (N <= 0) {
    if (threadIdx.x == 0) { flags[2 * b] = 0; flags[2 * b + 1] = 0; }
    return;
  }
  FeasIO F;
  F.coef = Bt.coef + 54 * Bt.poff[b];
  F.T = Bt.T + Bt.poff[b];
  F.N = N;
  F.x0 = Bt.start_xy[2 * b]; F.y0 = Bt.start_xy[2 * b + 1];
  F.th0 = Bt.head[(size_t)b * 27];
  F.cseq = cseq + (size_t)b * 2 * (cap_panels + 1);
  F.tk = tk + (size_t)b * cap_samples;
  F.cap_panels = cap_panels; F.cap_samples = cap_samples;
  F.report = report + (size_t)b * 38;
  F.feasible = flags + 2 * b;
  F.truncated = nullptr;
  const TOPAY_GLB DevMap* mp = (const TOPAY_GLB DevMap*)(maps + __builtin_amdgcn_readfirstlane(Bt.map_id[b]));
  feasibility_gate(F, mp);
}

// MomaParam::getMeshPose of n states, one thread per state
__global__ void __launch_bounds__(64) k_mesh_pose(topay_mesh_params_t K, int n, const double* states, double* parts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double st[10];
  for (int k = 0; k < 10; k++) st[k] = states[(size_t)i * 10 + k];
  mesh_pose(K, st, parts + (size_t)i * 77);
}

// MomaTraj playback of one candidate (car_seq + getState at given times)
__global__ void __launch_bounds__(64) k_playback(DevBatch Bt, int b, double* cseq, long long cap_panels, int nq, const double* times,
                                                 double* states, double* seq_out, int* nseq_out) {
  const int N = Bt.N[b];
  if (N <= 0) {
    if (threadIdx.x == 0) *nseq_out = 0;
    return;
  }
  FeasIO F;
  F.coef = Bt.coef + 54 * Bt.poff[b];
  F.T = Bt.T + Bt.poff[b];
  F.N = N;
  F.x0 = Bt.start_xy[2 * b]; F.y0 = Bt.start_xy[2 * b + 1];
  F.th0 = Bt.head[(size_t)b * 27];
  F.cseq = cseq;
  F.tk = nullptr;
  F.cap_panels = cap_panels; F.cap_samples = 0;
  F.report = nullptr;
  F.feasible = nullptr;
  F.truncated = nullptr;
  playback(F, nq, times, states, seq_out, nseq_out);
}

// getTraj() of a selection of candidates, packed by pieces (topay_get_results): one workgroup per selected candidate
__global__ void k_gather_results(DevBatch Bt, int n, const int* idx, const int* piece_off, double* durations, double* coeffs,
                                 double* knots) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int b = idx[k];
  const int N = piece_off[k + 1] - piece_off[k];
  if (N <= 0) return;
  const int rows = 6 * N, p0 = piece_off[k];
  const double* cm = Bt.coef + 54 * Bt.poff[b];   // [9][rows], element d * rows + 6 p + k = coefficient of t^k
  for (int t = threadIdx.x; t < N * 54; t += blockDim.x) {
    const int p = t / 54, r = t - 54 * p, d = r / 6, kk = r - 6 * d;
    coeffs[(size_t)p0 * 54 + t] = cm[(size_t)d * rows + 6 * p + 5 - kk];   // per piece 9 x 6, highest order first
  }
  for (int t = threadIdx.x; t < N; t += blockDim.x) durations[p0 + t] = Bt.T[Bt.poff[b] + t];
  for (int t = threadIdx.x; t < 2 * (N + 1); t += blockDim.x) knots[2 * (size_t)(p0 + k) + t] = Bt.knots[2 * (Bt.poff[b] + b) + t];
}

// GridMap::isWholeBodyCollision for a batch of states
__global__ void k_whole_body(const DevMap* maps, int map_id, int n, const double* states, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const DevMap M = maps[map_id];
  out[i] = whole_body_collision(M, states + (size_t)i * 10) ? 1 : 0;
}

// test hook for the deterministic elementary functions: out[4i..4i+3] = sin(a_i), cos(a_i), atan2(a_i, b_i), -
__global__ void k_math(const double* a, const double* b, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  det_sincos(a[i], &s, &c);
  out[4 * i] = s;
  out[4 * i + 1] = c;
  out[4 * i + 2] = det_atan2(a[i], b[i]);
  out[4 * i + 3] = sqrt(fabs(a[i])) / (1.0 + fabs(b[i]));
}

