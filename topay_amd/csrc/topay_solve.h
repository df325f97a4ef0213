// Persistent one-wavefront-per-trajectory solver: stage-1 L-BFGS, stage-2 ALM loop of L-BFGS runs.
//
// Follows /root/reference/src/planner/src/moma_traj_opt.cpp:359-497 (optimize part of optimizeTraj) and
// planner/include/utils/lbfgs.hpp:276-389 (line_search_lewisoverton), 439-722 (lbfgs_optimize), including the
// reference's non-standard early accept (lbfgs.hpp:327-330), cautious update (675-677) and full-length
// history (mem_size 256).  The 1.0 s wall-clock cap of the ALM loop (moma_traj_opt.cpp:403-407) is replaced
// by the deterministic cap DevParams::alm_max_outer.
//
// The L-BFGS vectors x, g, xp, gp, d and the (s, y) history live in HBM, element e owned by lane e % 64;
// dot products are wave reductions; the whole solve is one launch with a single evaluation call site.
#pragma once
#include "topay_eval.h"
#include "topay_eval_mw.h"

#ifndef TOPAY_PF_ELEMS
#define TOPAY_PF_ELEMS 24
#endif
#ifndef TOPAY_PF_MAX
#define TOPAY_PF_MAX 12
#endif

namespace topay {

struct SolveIO {
  glb_dp x;        // [n] decision vector (in: x0, out: final)
  glb_dp g;        // [n]
  glb_dp xp;       // [n]
  glb_dp gp;       // [n]
  glb_dp d;        // [n]
  glb_dp hist_s;   // [m][nstride]
  glb_dp hist_y;   // [m][nstride]
  glb_dp hist_ys;  // [m]  1 / (y.s)
  glb_dp hist_al;  // [m]
  int nstride;
  glb_ip stats;    // [8]
  glb_dp trace;    // optional f-per-evaluation trace
  int trace_cap;
  // cancellation (DevBatch::group_tau / cancel_flag): null = never
  TOPAY_GLB int* grp_tau;
  const TOPAY_GLB int* cancel_flag;
  int cancel_budget;
};

// Vector operations on the n-element L-BFGS vectors.  A lane owns EPL / 2 pairs of adjacent elements: register t holds
// element 128 (t / 2) + 2 lane + (t % 2), n <= 64 EPL, n even (10 N - 8), every vector and history row starts on a
// 16-byte boundary (n_max is even) -- so a pair moves with one 16-byte load or store: half the memory instructions per
// byte, which is what bounds a single wave (the counter of outstanding loads has 63 steps, whatever their width).
// Every load of an operation is issued before its arithmetic (clamped index + mask instead of a trip-count loop, whose
// iterations the compiler serialises): a single wave has nothing else to hide the HBM / L2 latency behind.
typedef double dpair __attribute__((vector_size(16)));   // plain vector type: loads through address-space pointers need no constructor
typedef const TOPAY_GLB dpair* glb_cpp;
typedef TOPAY_GLB dpair* glb_pp;
// (NT = threads per trajectory: 64 for the one-wave kernels; an NW-wave workgroup divides the pairs over 64 NW threads,
// thread tid owning the pairs NT p + tid)
template <int NT = 64>
__device__ __forceinline__ bool vec_in(int tid, int t, int n) { return 2 * (NT * (t >> 1) + tid) < n; }
// Pair i (elements 2 i, 2 i + 1) of an n-element row (n even, 16-byte aligned): a raw BUFFER load whose descriptor ends at
// the row's n-th element, so pairs beyond the row come back as zeros from the hardware's range check -- no index clamp
// and no select per loaded value (the two-loop recursion is bound by instruction issue: 8 of its 48 vector instructions
// per history pair were such selects).  The descriptor lives in scalar registers (uniform base).
// STREAM: the load may carry a cache-policy hint (TOPAY_HIST_AUX, bits of a gfx940+ buffer instruction: 1 = sc0, 2 = nt,
// 16 = sc1).  The (s, y) history of the two-loop recursion is read once per iteration -- 1.2 TB per step that no cache holds --
// and pushes reused lines (scratch, parked rows, LU factors) out of L2.  Measured with nt = 1 (round 5, docs/EXPERIMENTS.md):
// HBM-side writes - 20 %, reads + 6 %, strictly serial step + 4 % (the second loop re-reads the oldest pairs of the first
// right away): off.
#ifndef TOPAY_HIST_AUX
#define TOPAY_HIST_AUX 0
#endif
template <bool STREAM = false>
__device__ __forceinline__ dpair row_pair_or_zero(glb_cdp row, int n, int i) {
#ifndef TOPAY_CPU_EMU
  typedef unsigned int u32x4 __attribute__((vector_size(16)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, (short)0, n * 8, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16, 0, STREAM ? TOPAY_HIST_AUX : 0);
  return __builtin_bit_cast(dpair, v);
#else
  dpair q;
  q[0] = 2 * i < n ? row[2 * i] : 0.0;
  q[1] = 2 * i < n ? row[2 * i + 1] : 0.0;
  return q;
#endif
}
template <int EPL, int NT = 64>
__device__ __forceinline__ void vec_load(glb_cdp a, int n, int tid, double (&v)[EPL]) {
  static_assert(EPL % 2 == 0, "pairs");
#pragma unroll
  for (int p = 0; p < EPL / 2; p++) {
    const dpair q = row_pair_or_zero(a, n, NT * p + tid);   // (pairs beyond n: zeros; every user masks them anyway)
    v[2 * p] = q[0];
    v[2 * p + 1] = q[1];
  }
}
template <int EPL, int NT = 64>
__device__ __forceinline__ void vec_store(glb_dp a, int n, int tid, const double (&v)[EPL]) {
  const glb_pp ap = (glb_pp)a;
#pragma unroll
  for (int p = 0; p < EPL / 2; p++) {
    const int i = NT * p + tid;
    dpair q;
    q[0] = v[2 * p];
    q[1] = v[2 * p + 1];
    if (2 * i < n) ap[i] = q;
  }
}
// a thread's share of sum a[e] b[e]: its elements in ascending t (the caller finishes with the fixed wave / workgroup tree)
template <int EPL, int NT = 64>
__device__ __forceinline__ double vec_dot_part(glb_cdp a, glb_cdp b, int n, int tid) {
  double av[EPL], bv[EPL];
  vec_load<EPL, NT>(a, n, tid, av);
  vec_load<EPL, NT>(b, n, tid, bv);
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < EPL; t++) s += vec_in<NT>(tid, t, n) ? av[t] * bv[t] : 0.0;
  return s;
}

// Helper waves (the kernels for a handful of candidates, topay_kernels.h: k_lat*).  With NWE > NW = 1 the workgroup has NWE
// waves but the solver runs on wave 0 alone -- its vectors, reductions and two-loop recursion are those of the one-wave
// kernels, so the bits of a solve are the one-wave bits -- and the other waves only join the evaluations (which are
// order-identical for any number of waves, topay_eval_mw.h): wave 0 posts the evaluation's inputs in a command block in
// LDS, every wave meets at a workgroup barrier and evaluates.  What that shortens is the sample passes of the two sweeps (a
// 10-piece candidate has three; four waves run them side by side); the solver's serial chains stay.
//   command block, 16 doubles: finit, thr, early, skip_thr, lam0, lam1, rho0, rho1 | ints: op (1 evaluate, 0 leave), stage,
//   gate.always, gate.has_early, gate.early_ok
#define TOPAY_CMD_DOUBLES 16
template <int RMAX_E, int NWE, int OCC>
__device__ __forceinline__ void eval_helper_loop(EvalCtx& C, const TOPAY_GLB DevMap* mp, lds_dp cmd) {
  for (;;) {
    __syncthreads();
    const TOPAY_LDS int* ic = (const TOPAY_LDS int*)(cmd + 8);
    if (__builtin_amdgcn_readfirstlane(ic[0]) == 0) break;
    GradGate gate;
    gate.finit = cmd[0]; gate.thr = cmd[1]; gate.early = cmd[2]; gate.skip_thr = cmd[3];
    C.lam0 = cmd[4]; C.lam1 = cmd[5]; C.rho0 = cmd[6]; C.rho1 = cmd[7];
    const int stage = __builtin_amdgcn_readfirstlane(ic[1]);
    gate.always = ic[2] != 0; gate.has_early = ic[3] != 0; gate.early_ok = ic[4] != 0;
    if (stage == 1) (void)eval_cost_grad_mw<1, RMAX_E, NWE, OCC>(C, mp, gate);
    else (void)eval_cost_grad_mw<2, RMAX_E, NWE, OCC>(C, mp, gate);
    wg_lds_barrier();
  }
}

// diagnostics build: where the solver's time between two evaluations goes (slots 11..15 of the stamp block; the phase clock of
// STAMP is left alone).  This is how the host-memory poll below was found.
#ifdef TOPAY_STAMPS
#define LSTAMP(k)                                                                              \
  do {                                                                                         \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime();                            \
    if (C.stamps && C.lane == 0) C.stamps[k] += now_ - lt_;                                    \
    lt_ = (long long)__builtin_amdgcn_s_memtime();                                             \
  } while (0)
#else
#define LSTAMP(k) do { } while (0)
#endif
template <int RMAX, int NW = 1, int OCC = 2, int NWE = NW, int RMAX_E = RMAX>
__device__ __forceinline__ void solve_trajectory(EvalCtx& C, const TOPAY_GLB DevMap* mp, SolveIO& S, int s1_past,
                                                 lds_dp pf /* LDS [8 + 40], then the command block if NWE != NW */, int& success_out,
                                                 double& cost_out, int& interrupted_out) {
  static_assert(NWE == NW || NW == 1, "helper waves join a one-wave solver");
  constexpr bool HELPERS = NWE != NW;
  dev_params_ref P = dev_params();
  constexpr int NT = 64 * NW;    // threads that run the solver
  // barrier among the solver's threads (with helper waves: wave 0 alone -- no workgroup barrier outside the hand-shake)
  auto ssync = [&]() {
    if constexpr (HELPERS) { wave_global_sync(); lds_sync(); }
    else __syncthreads();
  };
  const int tid = C.tid, n = __builtin_amdgcn_readfirstlane(C.n);
  constexpr int EPL = 2 * RMAX;  // decision-vector elements per thread: n <= NT * EPL
  // sums / maxima over the trajectory's threads in a fixed order: the wave tree, then (NW > 1) the waves' partial results
  // through LDS, (w0 + w1) + (w2 + w3)
  const int wave = __builtin_amdgcn_readfirstlane(C.wave);
  const lds_dp c_red = C.red;
  int rp = 0;
  auto wsum = [&](double v) -> double {
    if constexpr (NW == 1) return wave_sum(v);
    else return wg_sum<NW>(c_red, rp, wave, v);
  };
  auto wmax = [&](double v) -> double {
    if constexpr (NW == 1) return wave_max(v);
    else return wg_max<NW>(c_red, rp, wave, v);
  };
  int stage = 1;
  int alm_iter = 0;
  bool success = false, interrupted = false;
  int st_s1_ret = 0, st_s1_it = 0, st_s1_ev = 0, st_s2_ret = 0, st_s2_it = 0, st_s2_ev = 0, st_sumb = 0;
  C.lam0 = P.alm_init_lambda[0]; C.lam1 = P.alm_init_lambda[1];
  C.rho0 = P.alm_init_rho[0];    C.rho1 = P.alm_init_rho[1];
  C.x = S.x;
  C.g = S.g;

  // per-run L-BFGS state (lbfgs.hpp:447-449)
  int k = 0, end = 0, bound = 0, count = 0, ret = 0, evals = 0;
  double fx = 0.0, step = 0.0, stp = 0.0, finit = 0.0, dginit = 0.0, dgtest = 0.0, dstest = 0.0, mu = 0.0, nu = 0.0;
  bool brackt = false, touched = false;
  enum { MODE_INIT = 0, MODE_LS = 1 };
  int mode = MODE_INIT;
  double cost = 0.0;
  int ntrace = 0;

  // topay_cancel's flag lives in pinned HOST memory: reading it is a round trip over the bus (2-3 us, and a wave's loads return
  // in order, so the next vector load waits for it whatever lies between).  Read at every evaluation it cost 8-12k cycles of each,
  // 3 % of a solve; it is read at every eighth (every fourth / every one for candidates of more than 16 / 32 pieces, whose
  // evaluations are long): a cancelled call comes back within a millisecond or two all the same.
#ifdef TOPAY_STAMPS
  long long lt_ = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    const DevLbfgs& lp = stage == 1 ? P.s1_lbfgs : P.s2_lbfgs;
    const int past = stage == 1 ? s1_past : lp.past;
    const int mem = lp.mem_size;
    // ------------------------------------------------------------------ interruption point (moma_traj_opt.cpp:402, 887)
    if (stage == 2 && (S.grp_tau || S.cancel_flag)) {
      // one thread reads (the values may change between two reads), everybody gets its verdict: lane 0's through a
      // broadcast inside the wave, through the parked-state block across waves
      int stop = 0;
      if (tid == 0) {
        const int clock = (st_s1_ev + st_s2_ev + evals) * C.N;   // piece-evaluations done so far
        const int poll_mask = C.N <= 16 ? 7 : (C.N <= 32 ? 3 : 0);
#ifndef TOPAY_CPU_EMU
        if (S.grp_tau) stop = (long long)clock > (long long)__hip_atomic_load(S.grp_tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + S.cancel_budget;
        if (S.cancel_flag && (evals & poll_mask) == 0) stop |= __hip_atomic_load(S.cancel_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
#else
        if (S.grp_tau) stop = (long long)clock > (long long)*S.grp_tau + S.cancel_budget;
        if (S.cancel_flag && (evals & poll_mask) == 0) stop |= *S.cancel_flag != 0;
#endif
      }
      if constexpr (NW == 1) {
        stop = __shfl(stop, 0);
      } else {
        TOPAY_LDS int* sw = (TOPAY_LDS int*)(pf + 46);
        __syncthreads();
        if (tid == 0) sw[0] = stop;
        __syncthreads();
        stop = sw[0];
      }
      if (stop) {
        st_s2_ret = TOPAY_INTERRUPTED; st_s2_it += k; st_s2_ev += evals;
        interrupted = true;
        success = false;
        break;
      }
    }
    // ------------------------------------------------------------------ evaluate at x
    ssync();
    STAMP(C, 9);  // L-BFGS bookkeeping between evaluations
    double f;
    GradGate gate;
    gate.always = (mode == MODE_INIT);
    gate.has_early = past > 0;
    gate.finit = finit;
    gate.thr = finit + stp * dgtest;      // the line search's own expressions (below)
    gate.early = past > 0 ? lp.delta / past : 0.0;
    // would the line search go on to another trial if this one failed the sufficient-decrease test?  (the checks that
    // follow such a failure below: trial budget, bracket width, step bounds)
    {
      const double stp_next = 0.5 * (mu + stp);
      gate.early_ok = mode == MODE_LS && (count + 1 < lp.max_linesearch) && !((stp - mu) < lp.machine_prec * stp) &&
                      !(stp_next < lp.min_step) && !(stp_next > lp.max_step);
      // twice the early-accept tolerance: the verdict must not hinge on how the solver rounds its own test
      const double no_early = past > 0 ? finit + 2.0 * gate.early * (fabs(finit) + 1.0) : gate.thr;
      gate.skip_thr = gate.thr > no_early ? gate.thr : no_early;
    }
    // The solver's scalar state is parked in LDS across the evaluation (the callees take the whole register file, and
    // what is live across a call would otherwise be spilled to scratch memory): every lane writes the same values to
    // the same words and reads them back afterwards.
    {
      lds_dp pk = pf + 8;
      pk[0] = fx; pk[1] = step; pk[2] = stp; pk[3] = finit; pk[4] = dginit; pk[5] = dgtest; pk[6] = dstest; pk[7] = mu;
      pk[8] = nu; pk[9] = cost;
      TOPAY_LDS int* ik = (TOPAY_LDS int*)(pk + 10);
      ik[0] = stage; ik[1] = alm_iter; ik[2] = st_s1_ret; ik[3] = st_s1_it; ik[4] = st_s1_ev; ik[5] = st_s2_ret; ik[6] = st_s2_it;
      ik[7] = st_s2_ev; ik[8] = st_sumb; ik[9] = k; ik[10] = end; ik[11] = bound; ik[12] = count; ik[13] = ret; ik[14] = evals;
      ik[15] = mode; ik[16] = ntrace; ik[17] = (brackt ? 1 : 0) | (touched ? 2 : 0);
      ik[18] = S.nstride; ik[19] = S.trace_cap; ik[20] = s1_past; ik[21] = S.cancel_budget;
      TOPAY_LDS unsigned long long* qk = (TOPAY_LDS unsigned long long*)(pk + 21);
      qk[0] = (unsigned long long)S.x; qk[1] = (unsigned long long)S.g; qk[2] = (unsigned long long)S.xp;
      qk[3] = (unsigned long long)S.gp; qk[4] = (unsigned long long)S.d; qk[5] = (unsigned long long)S.hist_s;
      qk[6] = (unsigned long long)S.hist_y; qk[7] = (unsigned long long)S.hist_ys; qk[8] = (unsigned long long)S.hist_al;
      qk[9] = (unsigned long long)S.stats; qk[10] = (unsigned long long)S.trace; qk[11] = (unsigned long long)mp;
      qk[12] = (unsigned long long)S.grp_tau; qk[13] = (unsigned long long)S.cancel_flag;
    }
    if constexpr (HELPERS) {
      lds_dp cmd = pf + 48;
      cmd[0] = gate.finit; cmd[1] = gate.thr; cmd[2] = gate.early; cmd[3] = gate.skip_thr;
      cmd[4] = C.lam0; cmd[5] = C.lam1; cmd[6] = C.rho0; cmd[7] = C.rho1;
      TOPAY_LDS int* ic = (TOPAY_LDS int*)(cmd + 8);
      ic[0] = 1; ic[1] = stage; ic[2] = gate.always ? 1 : 0; ic[3] = gate.has_early ? 1 : 0; ic[4] = gate.early_ok ? 1 : 0;
      __syncthreads();   // the helper waves wait here (eval_helper_loop); x written above is visible to them
    }
    LSTAMP(15);   // interruption poll, barrier, state parked
    if (stage == 1) f = eval_cost_grad_mw<1, RMAX_E, NWE, OCC>(C, mp, gate);
    else f = eval_cost_grad_mw<2, RMAX_E, NWE, OCC>(C, mp, gate);
#ifdef TOPAY_STAMPS
    lt_ = (long long)__builtin_amdgcn_s_memtime();
#endif
    if (NWE > 1) wg_lds_barrier();   // every wave is out of the evaluation's last reduction before the scratch is used again
    rp = 0;
    {
      // (read back as wave-uniform values: scalar registers, scalar branches, scalar base addresses for the vector loads)
      lds_cdp pk = pf + 8;
      fx = uniform_f64(pk[0]); step = uniform_f64(pk[1]); stp = uniform_f64(pk[2]); finit = uniform_f64(pk[3]);
      dginit = uniform_f64(pk[4]); dgtest = uniform_f64(pk[5]); dstest = uniform_f64(pk[6]); mu = uniform_f64(pk[7]);
      nu = uniform_f64(pk[8]); cost = uniform_f64(pk[9]);
      const TOPAY_LDS int* ik = (const TOPAY_LDS int*)(pk + 10);
      auto ui = [&](int q) { return __builtin_amdgcn_readfirstlane(ik[q]); };
      stage = ui(0); alm_iter = ui(1); st_s1_ret = ui(2); st_s1_it = ui(3); st_s1_ev = ui(4); st_s2_ret = ui(5); st_s2_it = ui(6);
      st_s2_ev = ui(7); st_sumb = ui(8); k = ui(9); end = ui(10); bound = ui(11); count = ui(12); ret = ui(13); evals = ui(14);
      mode = ui(15); ntrace = ui(16);
      const int fl = ui(17);
      brackt = (fl & 1) != 0; touched = (fl & 2) != 0;
      S.nstride = ui(18); S.trace_cap = ui(19); s1_past = ui(20); S.cancel_budget = ui(21);
      const TOPAY_LDS unsigned long long* qk = (const TOPAY_LDS unsigned long long*)(pk + 21);
      auto up = [&](int q) {
        const unsigned long long v = qk[q];
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
      };
      S.x = (glb_dp)up(0); S.g = (glb_dp)up(1); S.xp = (glb_dp)up(2); S.gp = (glb_dp)up(3); S.d = (glb_dp)up(4);
      S.hist_s = (glb_dp)up(5); S.hist_y = (glb_dp)up(6); S.hist_ys = (glb_dp)up(7); S.hist_al = (glb_dp)up(8);
      S.stats = (glb_ip)up(9); S.trace = (glb_dp)up(10); mp = (const TOPAY_GLB DevMap*)up(11);
      S.grp_tau = (TOPAY_GLB int*)up(12); S.cancel_flag = (const TOPAY_GLB int*)up(13);
      f = uniform_f64(f);
    }
    evals++;
    LSTAMP(11);   // state read back
#ifndef TOPAY_STAMPS
    if (S.trace && tid == 0 && ntrace < S.trace_cap) S.trace[ntrace] = f;
#endif
    ntrace++;

    enum { GO_EVAL = 0, GO_LS_BEGIN = 1, GO_RUN_END = 2 };
    int go = GO_EVAL;
    if (mode == MODE_INIT) {
      // lbfgs.hpp:523-554
      fx = f;
      if (tid == 0) pf[0] = fx;
      double gmax = 0.0, xmax = 0.0;
      {
        double gv[EPL], xv[EPL], dv[EPL];
        vec_load<EPL, NT>(S.g, n, tid, gv);
        vec_load<EPL, NT>(S.x, n, tid, xv);
#pragma unroll
        for (int t = 0; t < EPL; t++) {
          const bool in = vec_in<NT>(tid, t, n);
          dv[t] = -gv[t];
          gmax = in ? fmax(gmax, fabs(gv[t])) : gmax;
          xmax = in ? fmax(xmax, fabs(xv[t])) : xmax;
        }
        vec_store<EPL, NT>(S.d, n, tid, dv);
      }
      gmax = wmax(gmax);
      xmax = wmax(xmax);
      k = 0;
      if (gmax / fmax(1.0, xmax) < lp.g_epsilon) {
        ret = TOPAY_LBFGS_CONVERGENCE;
        go = GO_RUN_END;
      } else {
        step = 1.0 / sqrt(wsum(vec_dot_part<EPL, NT>(S.d, S.d, n, tid)));
        k = 1;
        end = 0;
        bound = 0;
        go = GO_LS_BEGIN;
      }
    } else {
      // ---- one line-search trial evaluated — lbfgs.hpp:318-387
      ++count;
      int ls = 0;  // 0: continue, >0: accepted with `count` evaluations, <0: error
      if (isinf(f) || isnan(f)) {
        ls = TOPAY_LBFGSERR_INVALID_FUNCVAL;
      } else if (past > 0 && fabs(finit - f) / (fabs(finit) + 1.0) < lp.delta / past) {
        ls = count;  // reference-specific early accept (lbfgs.hpp:327-330)
      } else {
        bool accepted = false;
        if (f > finit + stp * dgtest) {
          nu = stp;
          brackt = true;
        } else {
          const double gs = wsum(vec_dot_part<EPL, NT>(S.g, S.d, n, tid));
          if (gs < dstest) mu = stp;
          else accepted = true;
        }
        if (accepted) ls = count;
        else if (lp.max_linesearch <= count) ls = TOPAY_LBFGSERR_MAXIMUMLINESEARCH;
        else if (brackt && (nu - mu) < lp.machine_prec * nu) ls = TOPAY_LBFGSERR_WIDTHTOOSMALL;
        else {
          if (brackt) stp = 0.5 * (mu + nu);
          else stp *= 2.0;
          if (stp < lp.min_step) ls = TOPAY_LBFGSERR_MINIMUMSTEP;
          else if (stp > lp.max_step) {
            if (touched) ls = TOPAY_LBFGSERR_MAXIMUMSTEP;
            else { touched = true; stp = lp.max_step; }
          }
        }
      }
      if (ls == 0) {
        {
          double pv[EPL], dv[EPL], xv[EPL];
          vec_load<EPL, NT>(S.xp, n, tid, pv);
          vec_load<EPL, NT>(S.d, n, tid, dv);
#pragma unroll
          for (int t = 0; t < EPL; t++) xv[t] = pv[t] + stp * dv[t];
          vec_store<EPL, NT>(S.x, n, tid, xv);
        }
        go = GO_EVAL;
      } else if (ls < 0) {
        // revert to the previous point — lbfgs.hpp:575-582 (fx keeps the last trial value)
        fx = f;
        {
          double pv[EPL], qv[EPL];
          vec_load<EPL, NT>(S.xp, n, tid, pv);
          vec_load<EPL, NT>(S.gp, n, tid, qv);
          vec_store<EPL, NT>(S.x, n, tid, pv);
          vec_store<EPL, NT>(S.g, n, tid, qv);
        }
        ret = ls;
        go = GO_RUN_END;
      } else {
        // ---- iteration accepted — lbfgs.hpp:584-714
        fx = f;
        step = stp;
        bool fin = false;
        // progress callback earlyExit (moma_traj_opt.cpp:1873) cancels when k > max_iterations: never before 632
        if (stage == 2 && k > lp.max_iterations) { ret = TOPAY_LBFGS_CANCELED; fin = true; }
        if (!fin) {
          double gmax = 0.0, xmax = 0.0;
          {
            double gv[EPL], xv[EPL];
            vec_load<EPL, NT>(S.g, n, tid, gv);
            vec_load<EPL, NT>(S.x, n, tid, xv);
#pragma unroll
            for (int t = 0; t < EPL; t++) {
              const bool in = vec_in<NT>(tid, t, n);
              gmax = in ? fmax(gmax, fabs(gv[t])) : gmax;
              xmax = in ? fmax(xmax, fabs(xv[t])) : xmax;
            }
          }
          gmax = wmax(gmax);
          xmax = wmax(xmax);
          if (gmax / fmax(1.0, xmax) < lp.g_epsilon) { ret = TOPAY_LBFGS_CONVERGENCE; fin = true; }
        }
        if (!fin && past > 0) {
          if (past <= k) {
            const double rate = fabs(pf[k % past] - fx) / fmax(1.0, fabs(fx));
            if (rate < lp.delta) { ret = TOPAY_LBFGS_STOP; fin = true; }
          }
          if (!fin) {
            ssync();
            if (tid == 0) pf[k % past] = fx;
            ssync();
          }
        }
        if (!fin && lp.max_iterations != 0 && lp.max_iterations <= k) { ret = TOPAY_LBFGSERR_MAXIMUMITERATION; fin = true; }
        if (fin) {
          go = GO_RUN_END;
        } else {
          ++k;
          // s = x - xp, y = g - gp; ys, yy, |s|^2, |gp|^2 — lbfgs.hpp:647-677
          glb_dp sE = S.hist_s + (size_t)end * S.nstride;
          glb_dp yE = S.hist_y + (size_t)end * S.nstride;
          double ys = 0.0, yy = 0.0, ss = 0.0, gg = 0.0;
          {
            double xv[EPL], pv[EPL], gv[EPL], qv[EPL], sv[EPL], yv[EPL], dv[EPL];
            vec_load<EPL, NT>(S.x, n, tid, xv);
            vec_load<EPL, NT>(S.xp, n, tid, pv);
            vec_load<EPL, NT>(S.g, n, tid, gv);
            vec_load<EPL, NT>(S.gp, n, tid, qv);
#pragma unroll
            for (int t = 0; t < EPL; t++) {
              const bool in = vec_in<NT>(tid, t, n);
              const double se = xv[t] - pv[t], ye = gv[t] - qv[t], gpe = qv[t];
              sv[t] = se;
              yv[t] = ye;
              ys += in ? ye * se : 0.0; yy += in ? ye * ye : 0.0; ss += in ? se * se : 0.0; gg += in ? gpe * gpe : 0.0;
              dv[t] = -gv[t];
            }
            vec_store<EPL, NT>(sE, n, tid, sv);
            vec_store<EPL, NT>(yE, n, tid, yv);
            vec_store<EPL, NT>(S.d, n, tid, dv);
          }
          ys = wsum(ys); yy = wsum(yy); ss = wsum(ss); gg = wsum(gg);
          // 1/ys is stored instead of ys: one division per iteration instead of two per history pair.
          // Every lane stores the same value; each lane later reads back its own store.
          S.hist_ys[end] = 1.0 / ys;
          const double cau = ss * sqrt(gg) * lp.cautious_factor;
          if (ys > cau) {
            ++bound;
            bound = mem < bound ? mem : bound;
            end = (end + 1) % mem;
            if (stage == 2) st_sumb += bound;
            // two-loop recursion (lbfgs.hpp:691-710) with the direction held in registers (pairs of adjacent elements per lane, see vec_load).
            // History pairs stream from HBM through a PF-deep ring of register slots so that PF loads are always in flight
            // while the dependent dot-product / axpy chain runs.
            // Only loads may be in flight inside the loops (one store would make the memory counter unordered and
            // every wait a full drain), and every load is issued unconditionally so that the number outstanding is
            // static: the alpha values live in LDS (256 doubles) and the prefetch keeps running past the end (it re-reads
            // valid, unused rows).
            constexpr int PF0 = TOPAY_PF_ELEMS / EPL < TOPAY_PF_MAX ? TOPAY_PF_ELEMS / EPL : TOPAY_PF_MAX;
            constexpr int PF = PF0 < 1 ? 1 : PF0;  // pairs in flight (at least one: 28 elements per lane in the longest class)
            SUBSTAMP_BEGIN(C);
            double dr[EPL];
            {
              double gv[EPL];
              vec_load<EPL, NT>(S.g, n, tid, gv);
#pragma unroll
              for (int t = 0; t < EPL; t++) dr[t] = vec_in<NT>(tid, t, n) ? -gv[t] : 0.0;
            }
            double sb[PF][EPL], yb[PF][EPL], rb[PF];
            // [mem <= 256]: every lane writes / reads the same entry (LDS broadcast).  The ring lives in the evaluation's
            // scratch region (band / pass buffers, >= 960 doubles): nothing of an evaluation stays there between two calls,
            // and the recursion runs between them -- 2 KB less LDS per workgroup.
            lds_dp alpha = C.X;
            // wave-uniform loop state in scalar registers (the values are uniform by construction; the compiler only
            // sees that they were derived from vector compares)
            const int endu = __builtin_amdgcn_readfirstlane(end), boundu = __builtin_amdgcn_readfirstlane(bound);
            const int memu = __builtin_amdgcn_readfirstlane(mem), nstr = __builtin_amdgcn_readfirstlane(S.nstride);
            // Row addresses as 32-bit byte offsets from the two history blocks (a candidate's history is < 4 GB): one scalar
            // multiply per pair instead of a 64-bit product and shift, and the alpha ring is indexed by the STEP of the first
            // loop (the second loop walks the same pairs backwards, so it reads alpha[bound - 1 - i]) -- a lone wave issues
            // one instruction per four cycles whatever its kind, and 27 of the 60 per history pair were this bookkeeping.
            const unsigned rbytes = (unsigned)nstr * 8u;
            typedef const TOPAY_GLB char* glb_ccp;
            auto load_pair = [&](int slot, int jj) {
              const unsigned ro = (unsigned)jj * rbytes;
              const glb_cdp sj = (glb_cdp)((glb_ccp)S.hist_s + ro);
              const glb_cdp yj = (glb_cdp)((glb_ccp)S.hist_y + ro);
#pragma unroll
              for (int p = 0; p < EPL / 2; p++) {
                const int i = NT * p + tid;
                const dpair sv = row_pair_or_zero<true>(sj, n, i), yv = row_pair_or_zero<true>(yj, n, i);   // zeros beyond the row; streaming
                sb[slot][2 * p] = sv[0];
                sb[slot][2 * p + 1] = sv[1];
                yb[slot][2 * p] = yv[0];
                yb[slot][2 * p + 1] = yv[1];
              }
              rb[slot] = *(glb_cdp)((glb_ccp)S.hist_ys + (unsigned)jj * 8u);
            };
            // Each loop runs whole groups of PF steps without a branch inside (the scalar bookkeeping of the next load then
            // fills the wait states the DPP moves of a reduction need anyway), then the < PF steps left over, which find
            // their pairs already in the ring.
            // ---- first loop: newest pair first.  Pair index of step i: (end - 1 - i) mod mem
            auto step1 = [&](int u, int i) {
              double part = 0.0;
#pragma unroll
              for (int t = 0; t < EPL; t++) part = fma(sb[u][t], dr[t], part);
              const double al = wsum(part) * rb[u];
              alpha[i] = al;
#pragma unroll
              for (int t = 0; t < EPL; t++) dr[t] = fma(-al, yb[u][t], dr[t]);
            };
            int jl = endu;  // next pair to load (walks down, wrapping)
#pragma unroll
            for (int u = 0; u < PF; u++) { jl = jl == 0 ? memu - 1 : jl - 1; load_pair(u, jl); }
            int i0 = 0;
            for (; i0 + PF <= boundu; i0 += PF) {
#pragma unroll
              for (int u = 0; u < PF; u++) {
                step1(u, i0 + u);
                jl = jl == 0 ? memu - 1 : jl - 1;
                load_pair(u, jl);
              }
            }
#pragma unroll
            for (int u = 0; u < PF - 1; u++)
              if (i0 + u < boundu) step1(u, i0 + u);
            const double scl = ys / yy;
#pragma unroll
            for (int t = 0; t < EPL; t++) dr[t] *= scl;
            lds_sync();  // alpha entries written above are read below
            // ---- second loop: oldest pair first.  Pair index of step i: (end - bound + i) mod mem
            auto step2 = [&](int u, int i) {
              const double av = alpha[boundu - 1 - i];
              double part = 0.0;
#pragma unroll
              for (int t = 0; t < EPL; t++) part = fma(yb[u][t], dr[t], part);
              const double beta = wsum(part) * rb[u];
              const double co = av - beta;
#pragma unroll
              for (int t = 0; t < EPL; t++) dr[t] = fma(co, sb[u][t], dr[t]);
            };
            jl = endu - boundu;  // next pair to load (walks up, wrapping)
            jl = jl < 0 ? jl + memu : jl;
#pragma unroll
            for (int u = 0; u < PF; u++) { load_pair(u, jl); jl = jl + 1 == memu ? 0 : jl + 1; }
            for (i0 = 0; i0 + PF <= boundu; i0 += PF) {
#pragma unroll
              for (int u = 0; u < PF; u++) {
                step2(u, i0 + u);
                load_pair(u, jl);
                jl = jl + 1 == memu ? 0 : jl + 1;
              }
            }
#pragma unroll
            for (int u = 0; u < PF - 1; u++)
              if (i0 + u < boundu) step2(u, i0 + u);
            vec_store<EPL, NT>(S.d, n, tid, dr);
            SUBSTAMP_END(C, 10);  // two-loop recursion
          }
          step = 1.0;
          go = GO_LS_BEGIN;
        }
      }
    }

    LSTAMP(12);   // first point / line-search trial / accepted iteration (with the two-loop recursion, slot 10)
    if (go == GO_LS_BEGIN) {
      // lbfgs.hpp:559-573 + line-search prologue 288-311
      {
        double xv[EPL], gv[EPL];
        vec_load<EPL, NT>(S.x, n, tid, xv);
        vec_load<EPL, NT>(S.g, n, tid, gv);
        vec_store<EPL, NT>(S.xp, n, tid, xv);
        vec_store<EPL, NT>(S.gp, n, tid, gv);
      }
      stp = step;
      count = 0;
      brackt = false;
      touched = false;
      mu = 0.0;
      nu = lp.max_step;
      int err = 0;
      if (!(stp > 0.0)) err = TOPAY_LBFGSERR_INVALIDPARAMETERS;
      else {
        dginit = wsum(vec_dot_part<EPL, NT>(S.gp, S.d, n, tid));
        if (0.0 < dginit) err = TOPAY_LBFGSERR_INCREASEGRADIENT;
      }
      if (err) {
        ret = err;  // x == xp, g == gp already
        go = GO_RUN_END;
      } else {
        finit = fx;
        dgtest = lp.f_dec_coeff * dginit;
        dstest = lp.s_curv_coeff * dginit;
        {
          double pv[EPL], dv[EPL], xv[EPL];
          vec_load<EPL, NT>(S.xp, n, tid, pv);
          vec_load<EPL, NT>(S.d, n, tid, dv);
#pragma unroll
          for (int t = 0; t < EPL; t++) xv[t] = pv[t] + stp * dv[t];
          vec_store<EPL, NT>(S.x, n, tid, xv);
        }
        mode = MODE_LS;
        go = GO_EVAL;
      }
    }

    LSTAMP(13);   // line-search prologue
    if (go == GO_RUN_END) {
      cost = fx;
      const bool okret = (ret == TOPAY_LBFGS_CONVERGENCE || ret == TOPAY_LBFGS_CANCELED || ret == TOPAY_LBFGS_STOP ||
                          ret == TOPAY_LBFGSERR_MAXIMUMITERATION);
      if (stage == 1) {
        st_s1_ret = ret; st_s1_it = k; st_s1_ev = evals;
        if (!okret) break;  // moma_traj_opt.cpp:370-374
        stage = 2;
        alm_iter = 0;
      } else {
        st_s2_ret = ret; st_s2_it += k; st_s2_ev += evals;
        if (!(okret || ret == TOPAY_LBFGSERR_MAXIMUMLINESEARCH)) { success = false; break; }  // 429-441
        const double en = sqrt(C.fxe0 * C.fxe0 + C.fxe1 * C.fxe1);
        if (en < P.alm_tolerance) { success = true; break; }  // 451-455
        C.lam0 += C.rho0 * C.fxe0;  // 456-459
        C.lam1 += C.rho1 * C.fxe1;
        C.rho0 = fmin((1 + P.alm_gamma[0]) * C.rho0, P.alm_rho_max[0]);
        C.rho1 = fmin((1 + P.alm_gamma[1]) * C.rho1, P.alm_rho_max[1]);
      }
      // deterministic stand-in for the 1.0 s wall-clock cap checked at the top of the ALM loop (403-407)
      if (alm_iter >= P.alm_max_outer) break;
      if (P.alm_work_budget > 0 && st_s2_ev * C.N >= P.alm_work_budget) break;
      alm_iter++;
      evals = 0;
      mode = MODE_INIT;
    }
  }
  if constexpr (HELPERS) {   // release the helper waves
    TOPAY_LDS int* ic = (TOPAY_LDS int*)(pf + 48 + 8);
    ic[0] = 0;
    __syncthreads();
  }
  if (tid == 0) {
    S.stats[0] = st_s1_ret; S.stats[1] = st_s1_it; S.stats[2] = st_s1_ev; S.stats[3] = st_s2_ret;
    S.stats[4] = st_s2_it; S.stats[5] = st_s2_ev; S.stats[6] = alm_iter; S.stats[7] = st_sumb;
  }
  success_out = success ? 1 : 0;
  cost_out = cost;
  interrupted_out = interrupted ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// set_init_traj == optimizeTraj lines 146-357 (moma_traj_opt.cpp).  One thread per trajectory: the
// resampling is short, sequential scalar work.  scratch: per trajectory (3*P+1) x 14 doubles.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void normalize_angle(double ref, double& a) {
  const double PI = 3.14159265358979323846;
  while (ref - a > PI) a += 2 * PI;
  while (ref - a < -PI) a -= 2 * PI;
}
__device__ __forceinline__ double duration_trapezoid(double length, double startV, double endV, double maxV, double maxA) {
  double startv2 = startV * startV, endv2 = endV * endV, maxv2 = maxV * maxV;
  if (startV > maxV) startv2 = maxv2;
  if (endV > maxV) endv2 = maxv2;
  const double critical_len = (maxv2 - startv2) / (2 * maxA) + (maxv2 - endv2) / (2 * maxA);
  if (length >= critical_len) return (maxV - startV) / maxA + (maxV - endV) / maxA + (length - critical_len) / maxV;
  const double tmpv = sqrt(0.5 * (startv2 + endv2 + 2 * maxA * length));
  return (tmpv - startV) / maxA + (tmpv - endV) / maxA;
}
__device__ __forceinline__ double arc_trapezoid(double curt, double locallength, double startV, double endV, double maxV,
                                                double maxA) {
  double startv2 = startV * startV, endv2 = endV * endV, maxv2 = maxV * maxV;
  if (startV > maxV) startv2 = maxv2;
  if (endV > maxV) endv2 = maxv2;
  const double critical_len = (maxv2 - startv2) / (2 * maxA) + (maxv2 - endv2) / (2 * maxA);
  if (locallength >= critical_len) {
    const double t1 = (maxV - startV) / maxA;
    const double t2 = t1 + (locallength - critical_len) / maxV;
    if (curt <= t1) return startV * curt + 0.5 * maxA * (curt * curt);
    else if (curt <= t2) return startV * t1 + 0.5 * maxA * (t1 * t1) + (curt - t1) * maxV;
    else return startV * t1 + 0.5 * maxA * (t1 * t1) + (t2 - t1) * maxV + maxV * (curt - t2) -
                0.5 * maxA * (curt - t2) * (curt - t2);
  } else {
    const double tmpv = sqrt(0.5 * (startv2 + endv2 + 2 * maxA * locallength));
    const double tmpt = (tmpv - startV) / maxA;
    if (curt <= tmpt) return startV * curt + 0.5 * maxA * (curt * curt);
    else return startV * tmpt + 0.5 * maxA * (tmpt * tmpt) + tmpv * (curt - tmpt) - 0.5 * maxA * (curt - tmpt) * (curt - tmpt);
  }
}

// node layout in scratch: [0..11] = x y theta dtheta darc q7 ; [12] path_arc ; [13] weighted_path_arc
#define ND 14
__device__ __forceinline__ void init_one(const DevParams& P, const double* path, int Plen, const double* bvel,
                                         const double* bacc, double* nodes, int maxN, int* N_out, int* past_out,
                                         double* head, double* tail, double* start_xy, double* goal_xy, double* init_xy,
                                         double* x0) {
  auto BV = [&](int r, int c) { return bvel[c * 10 + r]; };
  auto BA = [&](int r, int c) { return bacc[c * 10 + r]; };
  start_xy[0] = path[0]; start_xy[1] = path[1];
  goal_xy[0] = path[(size_t)(Plen - 1) * 10]; goal_xy[1] = path[(size_t)(Plen - 1) * 10 + 1];
  int cnt = 0;
  auto push = [&](const double* s12) {
    for (int q = 0; q < 12; q++) nodes[(size_t)cnt * ND + q] = s12[q];
    cnt++;
  };
  double s12[12];
  for (int q = 0; q < 12; q++) s12[q] = 0.0;
  s12[0] = path[0]; s12[1] = path[1]; s12[2] = path[2];
  for (int q = 0; q < 7; q++) s12[5 + q] = path[3 + q];
  push(s12);
  for (int i = 1; i < Plen; i++) {
    const double* cur = path + (size_t)i * 10;
    const double* prev = path + (size_t)(i - 1) * 10;
    const double* back = nodes + (size_t)(cnt - 1) * ND;
    for (int q = 0; q < 12; q++) s12[q] = 0.0;
    const double dx = cur[0] - prev[0], dy = cur[1] - prev[1];
    const double arc_len = sqrt(dx * dx + dy * dy);
    double now_theta = cur[2];
    normalize_angle(back[2], now_theta);
    double theta_diff = now_theta - back[2];
    if (fabs(theta_diff) > 1e-2) {
      if (arc_len < 1e-2) {
        s12[0] = cur[0]; s12[1] = cur[1]; s12[2] = now_theta; s12[3] = theta_diff; s12[4] = 0.0;
        for (int q = 0; q < 7; q++) s12[5 + q] = cur[3 + q];
        push(s12);
      } else {
        for (int q = 0; q < 12; q++) s12[q] = back[q];
        double direct_theta = det_atan2(cur[1] - back[1], cur[0] - back[0]);
        normalize_angle(back[2], direct_theta);
        theta_diff = direct_theta - back[2];
        s12[2] = direct_theta; s12[3] = theta_diff; s12[4] = 0.0;
        push(s12);
        s12[0] = cur[0]; s12[1] = cur[1]; s12[2] = direct_theta; s12[3] = 0.0; s12[4] = arc_len;
        for (int q = 0; q < 7; q++) s12[5 + q] = cur[3 + q];
        push(s12);
        const double* back2 = nodes + (size_t)(cnt - 1) * ND;
        normalize_angle(back2[2], now_theta);
        theta_diff = now_theta - back2[2];
        s12[2] = now_theta; s12[3] = theta_diff; s12[4] = 0.0;
        push(s12);
      }
    } else if (arc_len > 1e-2) {
      s12[0] = cur[0]; s12[1] = cur[1]; s12[2] = now_theta; s12[3] = 0.0; s12[4] = arc_len;
      for (int q = 0; q < 7; q++) s12[5 + q] = cur[3 + q];
      push(s12);
    }
  }
  const int path_num = cnt;
  double total_len = 0, wtotal = 0;
  nodes[12] = 0.0; nodes[13] = 0.0;
  for (int idx = 1; idx < path_num; idx++) {
    double* nd = nodes + (size_t)idx * ND;
    total_len += nd[4];
    nd[12] = total_len;
    wtotal += 0.2 * fabs(nd[3]) + 1.4 * fabs(nd[4]);
    nd[13] = wtotal;
  }
  const double total_time = duration_trapezoid(wtotal, BV(0, 0), 0.0, P.max_v, P.max_a);
  int np = (int)(total_time / P.sample_interval + 0.5);
  if (np < P.min_piece_num) np = P.min_piece_num;
  const double sample_interval = total_time / np;
  // inner points — moma_traj_opt.cpp:247-277.  They are written straight into x0 (theta, s, Vq).
  int now_idx = 1, ninner = 0;
  // temporary inner storage: reuse x0 region?  store inner (9) + xy (2) per point in `nodes` tail (beyond path_num)
  double* inner = nodes + (size_t)path_num * ND;  // [k][11]
  for (double t = sample_interval; t < total_time - 1e-3; t += sample_interval) {
    const double arc = arc_trapezoid(t, wtotal, BV(0, 0), 0.0, P.max_v, P.max_a);
    for (int kk = now_idx; kk < path_num; kk++) {
      const double* pn = nodes + (size_t)kk * ND;
      const double* pp = nodes + (size_t)(kk - 1) * ND;
      const double tmp_arc = pn[13];
      if (tmp_arc >= arc) {
        now_idx = kk;
        const double l1 = tmp_arc - arc;
        const double l = pn[13] - pp[13];
        if (ninner < maxN - 1) {
          double* o = inner + (size_t)ninner * 11;
          o[0] = pp[2] + (l - l1) / l * (pn[3]);
          o[1] = pp[12] + (l - l1) / l * (pn[4]);
          for (int q = 0; q < 7; q++) o[2 + q] = pp[5 + q] + (l - l1) / l * (pn[5 + q] - pp[5 + q]);
          o[9] = l1 / l * pp[0] + (l - l1) / l * (pn[0]);
          o[10] = l1 / l * pp[1] + (l - l1) / l * (pn[1]);
        }
        ninner++;
        break;
      }
    }
  }
  const int N = ninner + 1;
  if (N > maxN) {  // not representable in this build: flagged, the solver skips it
    *N_out = -N;
    *past_out = 0;
    return;
  }
  *N_out = N;
  for (int i = 0; i < ninner; i++) {
    init_xy[2 * i] = inner[(size_t)i * 11 + 9];
    init_xy[2 * i + 1] = inner[(size_t)i * 11 + 10];
  }
  init_xy[2 * ninner] = goal_xy[0];
  init_xy[2 * ninner + 1] = goal_xy[1];
  // boundary PVA — moma_traj_opt.cpp:281-297
  for (int q = 0; q < 27; q++) { head[q] = 0.0; tail[q] = 0.0; }
  const double* n0 = nodes;
  const double* nl = nodes + (size_t)(path_num - 1) * ND;
  head[0 * 9 + 0] = n0[2];
  head[1 * 9 + 0] = BV(1, 0);
  head[2 * 9 + 0] = BA(1, 0);
  head[1 * 9 + 1] = BV(0, 0);
  head[2 * 9 + 1] = BA(0, 0);
  for (int q = 0; q < 7; q++) {
    head[0 * 9 + 2 + q] = n0[5 + q];
    head[1 * 9 + 2 + q] = BV(3 + q, 0);
    head[2 * 9 + 2 + q] = BA(3 + q, 0);
  }
  tail[0 * 9 + 0] = nl[2];
  tail[0 * 9 + 1] = nl[12];
  for (int q = 0; q < 7; q++) {
    tail[0 * 9 + 2 + q] = nl[5 + q];
    tail[1 * 9 + 2 + q] = BV(3 + q, 1);
    tail[2 * 9 + 2 + q] = BA(3 + q, 1);
  }
  // packed decision vector — moma_traj_opt.cpp:324-344
  double* Tau = x0;
  double* Theta = x0 + N;
  double* Arc = x0 + 2 * N - 1;
  double* Vq = x0 + 3 * N - 1;
  const double tau0 = logC2(sample_interval);
  for (int i = 0; i < N - 1; i++) {
    Tau[i] = tau0;
    Theta[i] = inner[(size_t)i * 11 + 0];
    Arc[i] = inner[(size_t)i * 11 + 1];
    for (int q = 0; q < 7; q++) Vq[i * 7 + q] = invSigmoidC2(inner[(size_t)i * 11 + 2 + q], P.joint_pos_limit_max[q]);
  }
  Tau[N - 1] = tau0;
  Arc[N - 1] = tail[0 * 9 + 1];
  // short-path handling — moma_traj_opt.cpp:354-357
  *past_out = (fabs(tail[0 * 9 + 1]) < P.s1_shot_path_horizon) ? P.s1_shot_path_past : P.s1_normal_past;
}

}  // namespace topay
