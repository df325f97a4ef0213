// Several wavefronts per trajectory: the cost + gradient evaluation of topay_eval.h with the sample passes, the rows of
// the 6N x 6N system and the L-BFGS vectors divided over the NW waves of one workgroup (NW = 2 or 4: the SIMDs of one
// compute unit), LDS shared.  Used for the long candidates -- the ones whose single-wave solve is the tail of a batch,
// whose LDS idles the other SIMDs of their compute unit, and (N > 64) the ones a single wave cannot hold at all.
//
// Same reference lines as topay_eval.h.  What is serial in the algorithm stays serial (the 6N pivots of the banded LU on
// wave 0, the substitutions with one lane per right-hand side); everything per sample, per row and per vector element is
// divided.  The arithmetic of an evaluation is ORDER-IDENTICAL to the one-wave evaluation:
//   * XY prefix / chain suffix: scan inside a pass by the wave that owns it, pass totals added sequentially in pass
//     order (the one-wave code's running carry);
//   * per-lane penalty cost: per-round hand-over of the pass costs through LDS, added in pass order;
//   * row accumulation: every row thread walks the samples of its piece in ascending order, pass by pass;
//   * sums over pieces: the pieces sit on wave 0's lanes for N <= 64, the other waves add exact zeros.
// So for N <= 64 an NW-wave evaluation returns bit for bit what the one-wave kernel returns (asserted on the GPU:
// tests/test_multiwave.py); the L-BFGS vector arithmetic around it is divided differently (topay_solve.h) and is
// restated per (EPL, NW) in the oracle's device-order mode.
//
// The dJ/dC accumulator lives in registers of the row threads (RMAX x 9 doubles), not in LDS.  Compact layout (N > 80:
// 202 N doubles no longer fit the 160 KB of a compute unit): the adjoint solve then runs in the LDS block of the
// coefficients, which move to the candidate's result block in HBM first (the dJ/dT correction reads them from there).
#pragma once
#include "topay_eval.h"

namespace topay {

// Workgroup barrier that orders LDS traffic only: outstanding global loads / stores (the prefetch ring of the two-loop
// recursion, the parked gradient rows) are not drained.  Cross-wave hand-offs through global memory use __syncthreads().
__device__ __forceinline__ void wg_lds_barrier() {
#ifndef TOPAY_CPU_EMU
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#else
  __syncthreads();
#endif
}

// (one wave: the wave-level hand-off of topay_eval.h, no s_barrier)
template <int NW>
__device__ __forceinline__ void wg_barrier() {
  if (NW == 1) lds_sync();
  else wg_lds_barrier();
}

// Hand-off through GLOBAL memory (stores by some threads, loads by others).  One wave: the memory instructions of a wave are
// performed in issue order, lanes or no lanes, so only the compiler has to be kept from reordering -- no wait for the stores to
// come back (a round trip to L2 / HBM each time: the LU stash, the coefficients' move to the result block and the gradient at
// the end of every evaluation).  Several waves: the full barrier.
template <int NW>
__device__ __forceinline__ void wg_global_barrier() {
  if (NW == 1) lds_sync();
  else __syncthreads();
}

// Sum over the workgroup in a fixed order: wave tree (wave_sum), then (w0 + w1) + (w2 + w3).  `red` holds two sets of
// four partial sums used alternately, so one barrier per reduction suffices (a wave can only write a set again after
// every wave has passed the barrier that follows the reads of its previous use); `phase` is a workgroup-uniform local
// counter, and the caller separates reductions that do not share one by a barrier.
template <int NW>
__device__ __forceinline__ double wg_combine(lds_dp red, int& phase, int wave, double wsum) {
  static_assert(NW == 1 || NW == 2 || NW == 4, "waves per trajectory");
  if (NW == 1) return wsum;
  lds_dp r = red + (phase & 1) * 4;
  phase++;
  r[wave] = wsum;
  wg_lds_barrier();
  if (NW == 2) return r[0] + r[1];
  return (r[0] + r[1]) + (r[2] + r[3]);
}
template <int NW>
__device__ __forceinline__ double wg_sum(lds_dp red, int& phase, int wave, double v) {
  return wg_combine<NW>(red, phase, wave, wave_sum(v));
}
template <int NW>
__device__ __forceinline__ double wg_max(lds_dp red, int& phase, int wave, double v) {
  const double wm = wave_max(v);
  if (NW == 1) return wm;
  lds_dp r = red + (phase & 1) * 4;
  phase++;
  r[wave] = wm;
  wg_lds_barrier();
  double m = r[0] > r[1] ? r[0] : r[1];
  if (NW == 4) {
    const double m2 = r[2] > r[3] ? r[2] : r[3];
    m = m > m2 ? m : m2;
  }
  return m;
}

// LDS of an NW-wave workgroup: coefficients | T powers | dJ/dT | per-piece scratch | power table | cross-wave scratch |
// union (band + reciprocal diagonal | positional gradients + one pass buffer per wave).  LDS is what decides how many
// trajectories share a compute unit once the kernels fit two waves per SIMD (round 4), so the adjoint solve runs in the
// coefficients' block for every class (they go to the candidate's result block in HBM first; a block of its own for the
// adjoint cost the four-wave classes 18 / 27 KB at N = 42 / 63 and the bench 5.7 %, docs/EXPERIMENTS.md).
// One wave: the plan holds nothing it can do without -- a pass buffer of 7 rows (the gradient rows of a pass are handed to the row lanes in two halves),
// no per-round cost exchange (the cost of a pass stays in its lane's register), pass totals sized by the passes there are,
// and the boundary conditions read from HBM where they are used (once per evaluation): 148 N + 264 doubles for N >= 8,
// 14 / 20 / 27 / 40 KB at N = 10 / 15 / 21 / 32 (rounds 1-3: 21 / 27 / 36 / 54 KB).
__host__ __device__ __forceinline__ int mw_pb_rows(int NW) { return NW == 1 ? 7 : 14; }
__host__ __device__ __forceinline__ int mw_npass(int Nmax) { return (TOPAY_EP * Nmax + 63) / 64; }
// [8] partial sums of a workgroup reduction (two phases; NW > 1) | [2 npass] pass totals | [2][NW][64] per-round costs
// (NW > 1) | [NW] masks
__host__ __device__ __forceinline__ int mw_red_doubles(int NW) { return NW > 1 ? 8 : 0; }
__host__ __device__ __forceinline__ int mw_misc_doubles(int Nmax, int NW) {
  return mw_red_doubles(NW) + 2 * mw_npass(Nmax) + (NW > 1 ? 2 * NW * 64 : 0) + NW;
}
__host__ __device__ __forceinline__ int lds_doubles_mw(int Nmax, int NW) {
  const int rows = 6 * Nmax;
  // the big block: band + reciprocal diagonal while the system is factorised | afterwards the coefficients (also the adjoint's
  // right-hand sides / solution) and, behind them, the positional gradients + pass buffers (the substitutions' windows and the
  // solver's alpha array use the same space)
  const int lu = 14 * rows, sw = 9 * rows + 26 * Nmax + mw_pb_rows(NW) * 64 * NW;
  return 5 * Nmax + Nmax + 4 * (Nmax + 1) + 156 + mw_misc_doubles(Nmax, NW) + (lu > sw ? lu : sw);
}
__device__ __forceinline__ void carve_mw(EvalCtx& C, lds_dp base, int Nmax, int NW) {
  const int rows = 6 * Nmax;
  lds_dp p = base;
  C.Tp = p; p += 5 * Nmax;
  C.gdT = p; p += Nmax;
  C.pcs = p; p += 4 * (Nmax + 1);
  C.pw = p; p += 156;
  C.red = p; p += mw_misc_doubles(Nmax, NW);
  C.cL = p;                // (the band of the factorisation starts here too)
  C.adj = C.cL;
  C.gC = C.adj;
  C.X = p + 9 * rows;
  C.npass_lds = mw_npass(Nmax);
}

// MINCO generate for an NW-wave workgroup: fills divided over all threads, LU on wave 0 (the pivots are a serial chain),
// substitutions on lanes 0..8 of wave 0 (topay_eval.h: band_sweep).  The band lives in the block the coefficients take
// afterwards: factorise, stash the factors in the candidate's LU block in HBM (the adjoint needs them again anyway), fill the
// right-hand sides over the band, substitute with the factors streamed back through two small windows.
template <int NW, int OCC>
__device__ __noinline__ void minco_generate_mw(EvalCtx& C) {
  constexpr int NT = 64 * NW;
  const lds_dp c_Tp = C.Tp;
  const lds_dp c_X = C.X;
  const lds_dp c_gdT = C.gdT;
  const glb_cdp c_hd = C.hd, c_tl = C.tl;   // head / tail PVA, 9 x 3 col-major each (HBM: read once per evaluation)
  const glb_dp c_lu = C.lu;
  const glb_cdp c_x = C.x;
  dev_params_ref P = dev_params();
  const int tid = C.tid, lane = C.lane, wave = __builtin_amdgcn_readfirstlane(C.wave);
  const int N = __builtin_amdgcn_readfirstlane(C.N), rows = __builtin_amdgcn_readfirstlane(C.rows);
  lds_dp cL = C.cL;
  lds_dp band = cL;                 // [13][rows] while the system is factorised, then the coefficients' block
  lds_dp rdiag = cL + 13 * rows;
  glb_cdp Tau = c_x;
  glb_cdp Theta = c_x + N;
  glb_cdp Arc = c_x + 2 * N - 1;
  glb_cdp Vq = c_x + 3 * N - 1;

  // Everything this function reads from HBM is requested first -- the decision vector (written by the solver a moment
  // ago) and the boundary conditions -- so that the zero fills below run under the loads' latency instead of ahead of it
  // (the wave-level fences of the LDS hand-offs keep the compiler from moving loads up by itself).
  const double tau_v = tid < N ? Tau[tid] : 0.0;
  constexpr int NI = 9;   // inner-point values per thread: 9 (N - 1) <= 9 NT
  double inner_v[NI];
#pragma unroll
  for (int u = 0; u < NI; u++) {
    const int t = tid + NT * u;
    inner_v[u] = 0.0;
    if (t < 9 * (N - 1)) {
      const int i = t / 9, d = t - 9 * i;
      const int dq = d >= 2 ? d - 2 : 0;   // (clamped: the compiler may issue the loads of all three branches)
      inner_v[u] = d == 0 ? Theta[i] : (d == 1 ? Arc[i] : Vq[i * 7 + dq]);
    }
  }
  double bc_v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (tid < 9) {
    const int d = tid;
    bc_v[0] = c_hd[0 * 9 + d]; bc_v[1] = c_hd[1 * 9 + d]; bc_v[2] = c_hd[2 * 9 + d];
    bc_v[3] = (d == 1) ? Arc[N - 1] : c_tl[0 * 9 + d];   // minco_end_state(1,0) = Arc[N-1]
    bc_v[4] = c_tl[1 * 9 + d]; bc_v[5] = c_tl[2 * 9 + d];
  }
  for (int t = tid; t < 13 * rows; t += NT) band[t] = 0.0;
  if (tid < N) {
    double T1 = expC2(tau_v);  // calTfromTau, moma_traj_opt.h:778-786
    double T2 = T1 * T1, T3 = T2 * T1, T4 = T2 * T2, T5 = T4 * T1;
    c_Tp[0 * N + tid] = T1; c_Tp[1 * N + tid] = T2; c_Tp[2 * N + tid] = T3;
    c_Tp[3 * N + tid] = T4; c_Tp[4 * N + tid] = T5;
    c_gdT[tid] = 0.0;
  }
  wg_barrier<NW>();
  if (tid == 0) {
    BAND(0, 0) = 1.0; BAND(1, 1) = 1.0; BAND(2, 2) = 2.0;
  }
  if (tid < N - 1) {
    const int i = tid;
    const double T1 = c_Tp[i], T2 = c_Tp[N + i], T3 = c_Tp[2 * N + i], T4 = c_Tp[3 * N + i], T5 = c_Tp[4 * N + i];
    const int r = 6 * i;
    BAND(r + 3, r + 3) = 6.0;  BAND(r + 3, r + 4) = 24.0 * T1; BAND(r + 3, r + 5) = 60.0 * T2; BAND(r + 3, r + 9) = -6.0;
    BAND(r + 4, r + 4) = 24.0; BAND(r + 4, r + 5) = 120.0 * T1; BAND(r + 4, r + 10) = -24.0;
    BAND(r + 5, r) = 1.0; BAND(r + 5, r + 1) = T1; BAND(r + 5, r + 2) = T2; BAND(r + 5, r + 3) = T3;
    BAND(r + 5, r + 4) = T4; BAND(r + 5, r + 5) = T5;
    BAND(r + 6, r) = 1.0; BAND(r + 6, r + 1) = T1; BAND(r + 6, r + 2) = T2; BAND(r + 6, r + 3) = T3;
    BAND(r + 6, r + 4) = T4; BAND(r + 6, r + 5) = T5; BAND(r + 6, r + 6) = -1.0;
    BAND(r + 7, r + 1) = 1.0; BAND(r + 7, r + 2) = 2 * T1; BAND(r + 7, r + 3) = 3 * T2; BAND(r + 7, r + 4) = 4 * T3;
    BAND(r + 7, r + 5) = 5 * T4; BAND(r + 7, r + 7) = -1.0;
    BAND(r + 8, r + 2) = 2.0; BAND(r + 8, r + 3) = 6 * T1; BAND(r + 8, r + 4) = 12 * T2; BAND(r + 8, r + 5) = 20 * T3;
    BAND(r + 8, r + 8) = -2.0;
  }
  if (tid == NT - 1) {
    const int i = N - 1, R0 = 6 * N;
    const double T1 = c_Tp[i], T2 = c_Tp[N + i], T3 = c_Tp[2 * N + i], T4 = c_Tp[3 * N + i], T5 = c_Tp[4 * N + i];
    BAND(R0 - 3, R0 - 6) = 1.0; BAND(R0 - 3, R0 - 5) = T1; BAND(R0 - 3, R0 - 4) = T2; BAND(R0 - 3, R0 - 3) = T3;
    BAND(R0 - 3, R0 - 2) = T4; BAND(R0 - 3, R0 - 1) = T5;
    BAND(R0 - 2, R0 - 5) = 1.0; BAND(R0 - 2, R0 - 4) = 2 * T1; BAND(R0 - 2, R0 - 3) = 3 * T2; BAND(R0 - 2, R0 - 2) = 4 * T3;
    BAND(R0 - 2, R0 - 1) = 5 * T4;
    BAND(R0 - 1, R0 - 4) = 2; BAND(R0 - 1, R0 - 3) = 6 * T1; BAND(R0 - 1, R0 - 2) = 12 * T2; BAND(R0 - 1, R0 - 1) = 20 * T3;
  }
  wg_barrier<NW>();
  STAMP(C, 0);  // fills
  // LU without pivoting on wave 0 (banded_system.hpp:66-91); the other waves wait at the barrier below
  if (wave == 0) {
    const int t = lane / 7 + 1, u = lane - (lane / 7) * 7;
    // BAND(a, b) = band[(a - b + 6) rows + b]: the four entries a lane reads at pivot k are lane constants + k.  Lanes 42..63
    // read the pivot itself; a lane whose row or column lies beyond the matrix (the last six pivots) reads some other entry
    // of the band and writes nothing.
    const bool lane_in = lane < 42;
    const int o_kk = 6 * rows;
    const int o_ik = lane_in ? (t + 6) * rows : o_kk;
    const int o_ij = lane_in ? (t - u + 6) * rows + u : o_kk;
    const int o_kj = lane_in ? (6 - u) * rows + u : o_kk;
    for (int k = 0; k <= rows - 2; k++) {
      const int i = k + t, j = k + u;
      const bool act = lane_in && (i < rows) && (j < rows);
      // all four reads are issued ahead of the division: one LDS round trip per pivot instead of two, and no branch around
      // the multiply-subtract
      double akk = band[o_kk + k], aik = band[o_ik + k], aij = band[o_ij + k], akj = band[o_kj + k];
      TOPAY_OPAQUE(aij); TOPAY_OPAQUE(akj);   // (or the compiler moves these two reads back behind the division)
      const double m = aik / akk;
      const double nv = (u == 0) ? m : (aij - m * akj);
      lds_sync();
      if (act) band[o_ij + k] = nv;
      lds_sync();
    }
  }
  wg_barrier<NW>();
  STAMP(C, 1);  // LU
  for (int t = tid; t < rows; t += NT) rdiag[t] = 1.0 / BAND(t, t);
  wg_barrier<NW>();
  // factors to the candidate's LU block; nothing of the band is read from LDS after this
  for (int t = tid; t < 14 * rows; t += NT) c_lu[t] = band[t];
  if (NW == 1) lds_sync();   // (wg_global_barrier: the sweeps below read the factors back on this wave)
  else { wave_global_sync(); wg_barrier<NW>(); }
  // right-hand sides over the band: boundary conditions and inner points, zero elsewhere
  for (int t = tid; t < 9 * rows; t += NT) cL[t] = 0.0;
  wg_barrier<NW>();
  if (tid < 9) {
    const int d = tid;
    cL[d * rows + 0] = bc_v[0];
    cL[d * rows + 1] = bc_v[1];
    cL[d * rows + 2] = bc_v[2];
    cL[d * rows + rows - 3] = bc_v[3];
    cL[d * rows + rows - 2] = bc_v[4];
    cL[d * rows + rows - 1] = bc_v[5];
  }
#pragma unroll
  for (int u = 0; u < NI; u++) {
    const int t = tid + NT * u;
    if (t < 9 * (N - 1)) {
      const int i = t / 9, d = t - 9 * i;
      const int dq = d >= 2 ? d - 2 : 0;
      cL[d * rows + 6 * i + 5] = d >= 2 ? sigmoidC2(inner_v[u], P.joint_pos_limit_max[dq]) : inner_v[u];
    }
  }
  wg_barrier<NW>();
  if (wave == 0) {
    const bool owner = lane < 9;
    const lds_dp mine = cL + (owner ? lane : 8) * rows;
    band_sweep<0>(mine, owner, (glb_cdp)c_lu, c_X, rows, lane);
    band_sweep<1>(mine, owner, (glb_cdp)c_lu, c_X, rows, lane);
  }
  C.cl_in_lds = 1;
  wg_barrier<NW>();
  STAMP(C, 2);  // substitutions, LU stash
}

// ---------------------------------------------------------------------------------------------
// The evaluation, NW waves.  RMAX = system rows per thread (rows <= 64 NW RMAX), N <= 64 NW.
// ---------------------------------------------------------------------------------------------
template <int STAGE, int RMAX, int NW, int OCC = 1>
__device__ __noinline__ TOPAY_CALLS_BIG_FUNCTIONS double eval_cost_grad_mw(EvalCtx& C, const TOPAY_GLB DevMap* mp, const GradGate gate) {
  constexpr int NT = 64 * NW;
  // (wave-uniform context fields into scalar registers: what stays in vector registers is saved and restored around every
  // call of the manipulator block)
  const lds_dp c_Tp = uniform_ptr(C.Tp);
  const lds_dp c_X = uniform_ptr(C.X);
  const double c_ex = uniform_f64(C.ex);
  const double c_ey = uniform_f64(C.ey);
  const glb_dp c_g = uniform_ptr(C.g);
  const lds_dp c_gdT = uniform_ptr(C.gdT);
  const glb_cdp c_init_xy = uniform_ptr(C.init_xy);
  const double c_lam0 = uniform_f64(C.lam0);
  const double c_lam1 = uniform_f64(C.lam1);
  const glb_dp c_lu = uniform_ptr(C.lu);
  const lds_dp c_pcs = uniform_ptr(C.pcs);
  const lds_dp c_pw = uniform_ptr(C.pw);
  const double c_rho0 = uniform_f64(C.rho0);
  const double c_rho1 = uniform_f64(C.rho1);
  const double c_sx = uniform_f64(C.sx);
  const double c_sy = uniform_f64(C.sy);
  const glb_cdp c_x = uniform_ptr(C.x);
  const lds_dp c_red = uniform_ptr(C.red);
  const lds_dp c_adj = uniform_ptr(C.adj);
  const glb_dp c_coefg = uniform_ptr(C.coefg);

  dev_params_ref P = dev_params();
  // (lane and tid are formed again after every call of the manipulator block -- fresh_lane_id -- instead of being kept
  // across it: what a lane holds in vector registers across the call goes through scratch memory)
  int lane = C.lane, tid = C.tid;
  const int wave = __builtin_amdgcn_readfirstlane(C.wave);
  const int N = __builtin_amdgcn_readfirstlane(C.N), rows = __builtin_amdgcn_readfirstlane(C.rows);
  lds_cdp cL = uniform_ptr(C.cL);
  const glb_dp c_sbuf = uniform_ptr(C.sbuf);
  const glb_dp c_mstash = uniform_ptr(C.mstash);
  const int c_sbs = __builtin_amdgcn_readfirstlane(C.sb_stride);
  const double g_skip_thr = uniform_f64(gate.skip_thr);
  const bool g_early_ok = __builtin_amdgcn_readfirstlane(gate.early_ok ? 1 : 0) != 0;
  GradGate ugate;   // the line search's gate in scalar registers (it is consulted after the sample passes)
  ugate.always = __builtin_amdgcn_readfirstlane(gate.always ? 1 : 0) != 0;
  ugate.has_early = __builtin_amdgcn_readfirstlane(gate.has_early ? 1 : 0) != 0;
  ugate.finit = uniform_f64(gate.finit); ugate.thr = uniform_f64(gate.thr); ugate.early = uniform_f64(gate.early);
  ugate.early_ok = g_early_ok; ugate.skip_thr = g_skip_thr;
  int rp = 0;                                  // phase of the workgroup-reduction scratch
  const int npl = __builtin_amdgcn_readfirstlane(C.npass_lds);
  constexpr int RD = NW > 1 ? 8 : 0;           // (one wave: the reductions need no LDS)
  const lds_dp ptot = c_red + RD;              // [npass][2] pass totals of the XY prefix / chain suffix
  const lds_dp csr = c_red + RD + 2 * npl;     // [2][NW][64] pass costs of one round (two rounds in flight; NW > 1 only)
  TOPAY_LDS unsigned long long* jmask = (TOPAY_LDS unsigned long long*)(c_red + RD + 2 * npl + (NW > 1 ? 2 * NW * 64 : 0));   // [NW]
  constexpr int PBR = NW == 1 ? 7 : 14;        // rows of a wave's pass buffer
  minco_generate_mw<NW, OCC>(C);

  // ---- jerk energy per piece (thread <-> piece; N <= NT); its dJ/dT part is formed in the gradient phase, from the same
  // expressions (it would otherwise be carried in a register across every call of the manipulator block)
  auto jerk_terms = [&](int i, double& e_out, double& gdT_out) __attribute__((always_inline)) {
    double w33 = 0, w43 = 0, w44 = 0, w53 = 0, w54 = 0, w55 = 0;
#pragma unroll
    for (int d = 0; d < 9; d++) {
      const double c3 = cL[d * rows + 6 * i + 3], c4 = cL[d * rows + 6 * i + 4], c5 = cL[d * rows + 6 * i + 5];
      const double e = P.energy_weights[d];
      w33 += (c3 * e) * c3; w43 += (c4 * e) * c3; w44 += (c4 * e) * c4;
      w53 += (c5 * e) * c3; w54 += (c5 * e) * c4; w55 += (c5 * e) * c5;
    }
    const double T1 = c_Tp[i], T2 = c_Tp[N + i], T3 = c_Tp[2 * N + i], T4 = c_Tp[3 * N + i], T5 = c_Tp[4 * N + i];
    e_out = 36.0 * w33 * T1 + 144.0 * w43 * T2 + 192.0 * w44 * T3 + 240.0 * w53 * T3 + 720.0 * w54 * T4 + 720.0 * w55 * T5;
    gdT_out = 36.0 * w33 + 288.0 * w43 * T1 + 576.0 * w44 * T2 + 720.0 * w53 * T2 + 2880.0 * w54 * T3 + 3600.0 * w55 * T4;
  };
  double jerk_e = 0.0;
  if (tid < N) {
    double unused_;
    jerk_terms(tid, jerk_e, unused_);
  }
  const double jerk_cost = wg_sum<NW>(c_red, rp, wave, jerk_e);

  lds_dp gxy = c_X;                             // [13N][2] XY prefix of each even sample, then its positional gradient
  lds_dp pball = c_X + 26 * N;                  // [NW][PBR][64] one pass buffer per wave
  lds_dp pbuf = pball + wave * (PBR * 64);
  const int NE = TOPAY_EP * N;
  const int npass = (NE + 63) / 64;
  const int nround = (npass + NW - 1) / NW;
  const double wT = STAGE == 1 ? P.s1_time_weight : P.s2_time_weight;
  const double time_cost = wT * wg_sum<NW>(c_red, rp, wave, tid < N ? c_Tp[tid] : 0.0);
  bool skip_body = false;
  double f_skip = 0.0;
  if (STAGE == 2 && g_early_ok) {
    const double partial = jerk_cost + time_cost;
    if (partial > g_skip_thr && partial <= 1.79769313486231570e308) {
      skip_body = true;
      f_skip = partial;
    }
  }
  const double wM = STAGE == 1 ? P.s1_moment_weight : P.s2_moment_weight;
  const double wA = STAGE == 1 ? P.s1_acc_weight : P.s2_acc_weight;
  const double wD = STAGE == 1 ? P.s1_domega_weight : P.s2_domega_weight;

  STAMP(C, 3);  // jerk, row bookkeeping
  // =========================== sweep 1, phase A: Simpson panels, prefix inside each pass, pass totals
  for (int k = 0; k < nround; k++) {
    const int pass = k * NW + wave;
    if (pass < npass) {
      const int e = pass * 64 + lane;
      const bool act = e < NE;
      const int i = act ? e / TOPAY_EP : N - 1;
      const int m = act ? e - TOPAY_EP * i : 0;
      const int j = 2 * m;
      const double T1 = c_Tp[i];
      const double step = T1 / TOPAY_K, half = step / 2.0, coeff = step / 6.0;
      double f0x, f0y, Ix = 0.0, Iy = 0.0;
      xy_integrand(cL, rows, i, j * half, f0x, f0y);
      if (act && m < TOPAY_K) {
        double f1x, f1y, f2x, f2y;
        xy_integrand(cL, rows, i, (j + 1) * half, f1x, f1y);
        xy_integrand(cL, rows, i, (j + 2) * half, f2x, f2y);
        Ix = coeff * f0x + 4 * coeff * f1x + coeff * f2x;
        Iy = coeff * f0y + 4 * coeff * f1y + coeff * f2y;
      }
      const double incx = wave_incl_scan(Ix, lane), incy = wave_incl_scan(Iy, lane);
      if (act) {
        gxy[2 * e] = incx - Ix;
        gxy[2 * e + 1] = incy - Iy;
      }
      if (lane == 63) {
        ptot[2 * pass] = incx;
        ptot[2 * pass + 1] = incy;
      }
    }
  }
  wg_barrier<NW>();

  // =========================== sweep 1, phase B: sample bodies, one pass per wave and round
  double carryx = 0.0, carryy = 0.0;   // XY prefix carried across passes: the totals of the passes before `pc`, added in order (scalar registers)
  int pc = 0;
  double cost_pen = 0.0;               // per-lane penalty cost in the one-wave order: lane l adds its sample of pass 0, 1, 2, ...
  for (int k = 0; k < nround; k++) {
    const int pass = k * NW + wave;
    double cst_out = 0.0;
    if (pass < npass) {
      while (pc < pass) {
        carryx = uniform_f64(carryx + ptot[2 * pc]);
        carryy = uniform_f64(carryy + ptot[2 * pc + 1]);
        pc++;
      }
      // the sample of lane ln in this pass: index, piece, step, XY position (prefix inside the pass + carry)
      int e, i, m, j;
      bool act;
      double step, half, posx, posy;
      auto geometry = [&](int ln) __attribute__((always_inline)) {
        e = pass * 64 + ln;
        act = e < NE;
        i = act ? e / TOPAY_EP : N - 1;
        m = act ? e - TOPAY_EP * i : 0;
        j = 2 * m;
        const double T1 = c_Tp[i];
        step = T1 / TOPAY_K;
        half = step / 2.0;
        const double px0 = act ? gxy[2 * e] : 0.0, py0 = act ? gxy[2 * e + 1] : 0.0;
        posx = c_sx + (carryx + px0);
        posy = c_sy + (carryy + py0);
      };
      geometry(lane);
      if (act && m == TOPAY_K) {
        c_pcs[2 * N + 2 * (i + 1)] = posx;
        c_pcs[2 * N + 2 * (i + 1) + 1] = posy;
      }
      if (!skip_body) {   // (wave-uniform: no call of the non-inlined manipulator block under a partial EXEC mask, see topay_eval.h)
        ManiOut mo;
        mo.gx = mo.gy = mo.gth = mo.cost = mo.gdT = 0.0;
        if (STAGE == 2) {
          mo = sample_mani<OCC>(cL, rows, i, j, e, act, step, half, posx, posy, mp, c_mstash, pbuf + lane);
          lane = fresh_lane_id(lane);
          tid = wave * 64 + lane;
          geometry(lane);
        }
        double gB[12], gdTs, gpx, gpy, cst;
        bool jva;
        sample_rest<STAGE>(P, cL, rows, i, j, step, half, posx, posy, mp, wM, wA, wD, mo, pbuf + lane, gB, gdTs, gpx, gpy, jva, cst);
        if (act) {
          cst_out = cst;
          gxy[2 * e] = gpx;
          gxy[2 * e + 1] = gpy;
          glb_dp sb = c_sbuf + e;
          const int ss = c_sbs;
#pragma unroll
          for (int v = 0; v < 5; v++) sb[v * ss] = gB[v];
          sb[5 * ss] = gdTs;
          if (STAGE == 2) {
#pragma unroll
            for (int v = 0; v < 7; v++) sb[(6 + v) * ss] = gB[5 + v];
            sb[13 * ss] = jva ? 1.0 : 0.0;
          }
        }
      }
    }
    // the costs of this round's passes, added lane by lane in pass order; early rejection tested after every pass
    // (one wave: the cost of the round's only pass is this lane's own)
    lds_dp cs = csr + (k & 1) * (NW * 64);
    if (NW > 1) {
      cs[wave * 64 + lane] = cst_out;
      wg_barrier<NW>();
    }
#pragma unroll
    for (int q = 0; q < NW; q++) {
      const int p2 = k * NW + q;
      if (p2 < npass) {
        cost_pen += NW > 1 ? cs[q * 64 + lane] : cst_out;
        if (STAGE == 2 && g_early_ok && !skip_body && p2 + 1 < npass) {
          const double partial = jerk_cost + wave_sum(cost_pen) + time_cost;
          if (partial > g_skip_thr && partial <= 1.79769313486231570e308) {
            skip_body = true;
            f_skip = partial;
          }
        }
      }
    }
  }
  while (pc < npass) {
    carryx = uniform_f64(carryx + ptot[2 * pc]);
    carryy = uniform_f64(carryy + ptot[2 * pc + 1]);
    pc++;
  }
  if (NW == 1) lds_sync();   // (several waves: the barrier of the last round's cost exchange) piece-end positions are read below
  STAMP(C, 4);  // sweep 1

  // ---- per-piece terms between the sweeps
  double cost_piece = 0.0;
  double chain0x = 0.0, chain0y = 0.0;
  double mt_add_all = 0.0, mt_add_own = 0.0;
  if (STAGE == 1) {
    if (tid < N) {
      const double ex = c_pcs[2 * N + 2 * (tid + 1)] - c_init_xy[2 * tid];
      const double ey = c_pcs[2 * N + 2 * (tid + 1) + 1] - c_init_xy[2 * tid + 1];
      cost_piece = P.s1_path_pos_weight * (ex * ex + ey * ey);
      c_pcs[2 * tid] = P.s1_path_pos_weight * 2.0 * ex;
      c_pcs[2 * tid + 1] = P.s1_path_pos_weight * 2.0 * ey;
    }
    wg_barrier<NW>();
  } else {
    const double Tm = tid < N ? c_Tp[tid] : 0.0;
    const double avg = wg_sum<NW>(c_red, rp, wave, Tm) / N;
    double add_all = 0.0, add_own = 0.0;
    if (tid < N) {
      const double wMT = P.s2_mean_time_weight;
      if (Tm < avg * 0.5) {
        const double dd = Tm - avg * 0.5;
        cost_piece += wMT * dd * dd;
        add_all += wMT * 2.0 * dd * (-0.5 / N);
        add_own += wMT * 2.0 * dd;
      }
      if (Tm > avg * 2.0) {
        const double dd = Tm - avg * 2.0;
        cost_piece += wMT * dd * dd;
        add_all += wMT * 2.0 * dd * (-2.0 / N);
        add_own += wMT * 2.0 * dd;
      }
    }
    mt_add_all = add_all;
    mt_add_own = add_own;
    C.fxe0 = (c_sx + carryx) - c_ex;
    C.fxe1 = (c_sy + carryy) - c_ey;
    const double ea = C.fxe0 + c_lam0 / c_rho0, eb = C.fxe1 + c_lam1 / c_rho1;
    if (tid == 0) cost_piece += 0.5 * (c_rho0 * (ea * ea) + c_rho1 * (eb * eb));
    chain0x = c_rho0 * ea;
    chain0y = c_rho1 * eb;
  }
  // (the lanes of wave 0 carry the per-lane sample costs: every wave holds the same cost_pen)
  double penalty_cost = wg_sum<NW>(c_red, rp, wave, (wave == 0 ? cost_pen : 0.0) + cost_piece);
  const bool bad = (STAGE == 2) && !(fabs(penalty_cost) <= 1.79769313486231570e308);
  const double f_total = jerk_cost + (bad ? 1.0e+22 : penalty_cost) + time_cost;

  STAMP(C, 5);  // per-piece terms, cost
  if (skip_body) return f_skip;
  if (!ugate.needs(f_total)) return f_total;

  // =========================== gradient phase ===========================
  // ---- row bookkeeping: thread tid owns the system rows tid + NT r (formed here, after the sample passes)
  int rrow[RMAX], rpiece[RMAX], rk[RMAX];
  bool ract[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; r++) {
    rrow[r] = tid + NT * r;
    ract[r] = rrow[r] < rows;
    rpiece[r] = rrow[r] / 6;
    rk[r] = rrow[r] - 6 * rpiece[r];
  }
  double jerk_gdT = 0.0;
  if (tid < N) {
    double unused_;
    jerk_terms(tid, unused_, jerk_gdT);
  }
  // dJ/dC accumulator of this thread's rows (registers): theta, s, seven joints
  double a0[RMAX], a1[RMAX], aq[RMAX][7];
  double rbh[RMAX][3];   // hs-powers of each row (basis_k(k, hs))
#pragma unroll
  for (int r = 0; r < RMAX; r++) {
    a0[r] = 0.0;
    a1[r] = 0.0;
#pragma unroll
    for (int q = 0; q < 7; q++) aq[r][q] = 0.0;
    rbh[r][0] = rbh[r][1] = rbh[r][2] = 0.0;
    if (ract[r]) {
      const double hs = c_Tp[rpiece[r]] / TOPAY_K / 2.0;
      basis_k(rk[r], hs, rbh[r][0], rbh[r][1], rbh[r][2]);
    }
  }
  constexpr int NV = (STAGE == 2) ? 13 : 6;
  if constexpr (NW == 1) {
    // One wave: a pass's gradient rows reach the row lanes in two halves through the 7-row pass buffer -- theta / s rows
    // and dJ/dT first, then the seven joint rows.  Every accumulator still sees its samples in ascending order, so the sums
    // are the sums of the 13-row round; the rare joint velocity / acceleration rows are broadcast from the flagged lanes'
    // registers (no LDS), sample by sample in ascending order.
    for (int pass = 0; pass < npass; pass++) {
      const int e = pass * 64 + lane;
      const bool act = e < NE;
      const int i = act ? e / TOPAY_EP : N - 1;
      const int m = act ? e - TOPAY_EP * i : 0;
      const int j = 2 * m;
      const double step = c_Tp[i] / TOPAY_K, half = step / 2.0;
      bool jva = false;
      double rawq[7];
      {
        glb_cdp sb = c_sbuf + (act ? e : NE - 1);
        const int ss = c_sbs;
        double raw[14];
#pragma unroll
        for (int v = 0; v < ((STAGE == 2) ? 14 : 6); v++) raw[v] = sb[v * ss];
#pragma unroll
        for (int v = 0; v < 6; v++) pbuf[v * 64 + lane] = act ? raw[v] : 0.0;
#pragma unroll
        for (int v = 0; v < 7; v++) rawq[v] = (STAGE == 2 && act) ? raw[6 + v] : 0.0;
        if (STAGE == 2) jva = act && raw[13] != 0.0;
      }
      lds_sync();
      // half 1: theta / s rows (orders 0-2 / 1-2) and dJ/dT
#pragma unroll
      for (int r = 0; r < RMAX; r++) {
        if (ract[r]) {
          const int pi = rpiece[r];
          const double h0 = rbh[r][0], h1 = rbh[r][1], h2 = rbh[r][2];
          const int k0 = rk[r], k1 = rk[r] >= 1 ? rk[r] - 1 : 0, k2 = rk[r] >= 2 ? rk[r] - 2 : 0;
          const int e_lo = max(TOPAY_EP * pi, pass * 64), e_hi = min(min(TOPAY_EP * pi + TOPAY_EP, pass * 64 + 64), NE);
          double gt = 0.0;
          constexpr int CH = 3;
          for (int c0 = e_lo; c0 < e_hi; c0 += CH) {
            double pb[CH][6], t0[CH], t1[CH], t2[CH];
#pragma unroll
            for (int u = 0; u < CH; u++) {
              const int ee = (c0 + u < e_hi) ? c0 + u : e_hi - 1;
              const int l = ee - pass * 64, mm = ee - TOPAY_EP * pi;
              lds_cdp pj = c_pw + 12 * mm;
              t0[u] = pj[k0]; t1[u] = pj[k1]; t2[u] = pj[k2];
#pragma unroll
              for (int v = 0; v < 6; v++) pb[u][v] = pbuf[v * 64 + l];
            }
#pragma unroll
            for (int u = 0; u < CH; u++) {
              const bool ok = c0 + u < e_hi;
              const double b0 = h0 * t0[u], b1 = h1 * t1[u], b2 = h2 * t2[u];
              const double i0 = fma(b2, pb[u][2], fma(b1, pb[u][1], b0 * pb[u][0]));
              const double i1 = fma(b2, pb[u][4], b1 * pb[u][3]);
              a0[r] += ok ? i0 : 0.0;
              a1[r] += ok ? i1 : 0.0;
              gt += ok ? pb[u][5] : 0.0;
            }
          }
          if (e_lo < e_hi && rk[r] == 0) c_gdT[pi] += gt;
        }
      }
      if (STAGE == 2) {
        lds_sync();
#pragma unroll
        for (int v = 0; v < 7; v++) pbuf[v * 64 + lane] = rawq[v];
        lds_sync();
        // half 2: the order-0 joint rows
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
          if (ract[r]) {
            const int pi = rpiece[r];
            const double h0 = rbh[r][0];
            const int k0 = rk[r];
            const int e_lo = max(TOPAY_EP * pi, pass * 64), e_hi = min(min(TOPAY_EP * pi + TOPAY_EP, pass * 64 + 64), NE);
            constexpr int CH = 3;
            for (int c0 = e_lo; c0 < e_hi; c0 += CH) {
              double pb[CH][7], t0[CH];
#pragma unroll
              for (int u = 0; u < CH; u++) {
                const int ee = (c0 + u < e_hi) ? c0 + u : e_hi - 1;
                const int l = ee - pass * 64, mm = ee - TOPAY_EP * pi;
                t0[u] = c_pw[12 * mm + k0];
#pragma unroll
                for (int v = 0; v < 7; v++) pb[u][v] = pbuf[v * 64 + l];
              }
#pragma unroll
              for (int u = 0; u < CH; u++) {
                const bool ok = c0 + u < e_hi;
                const double b0 = h0 * t0[u];
#pragma unroll
                for (int qq = 0; qq < 7; qq++) {
                  const double nq = fma(b0, pb[u][qq], aq[r][qq]);
                  aq[r][qq] = ok ? nq : aq[r][qq];
                }
              }
            }
          }
        }
        // rare: joint velocity / acceleration gradBeta rows 1 and 2 (moma_traj_opt.cpp:1689, 1703) of the flagged samples
        unsigned long long todo = __ballot(jva);
        if (todo != 0) {
          double g1[7], g2[7];
#pragma unroll
          for (int q = 0; q < 7; q++) { g1[q] = 0.0; g2[q] = 0.0; }
          if (jva) {
            Basis B;
            make_basis(j * half, B);
            const double omg = (j == 0 || j == 2 * TOPAY_K) ? 0.5 : 1.0;
#pragma unroll
            for (int q = 0; q < 7; q++) {
              double p0, p1, p2;
              poly3(cL, rows, i, 2 + q, B, p0, p1, p2);
              const double vDq = p1 * p1 - P.joint_vel_limit2[q];
              const double vD2q = p2 * p2 - P.joint_acc_limit2[q];
              if (vDq > 0) {
                double pe, pd;
                smoothL1(P, vDq, P.relu_mu, pe, pd);
                g1[q] = omg * step * P.s2_mani_vel_weight * pd * 2.0 * p1;
              }
              if (vD2q > 0) {
                double pe, pd;
                smoothL1(P, vD2q, P.relu_mu, pe, pd);
                g2[q] = omg * step * P.s2_mani_acc_weight * pd * 2.0 * p2;
              }
            }
          }
          while (todo) {
            const int src = __ffsll(todo) - 1;
            todo &= todo - 1;
            const int se = pass * 64 + src;
            const int spi = se / TOPAY_EP, smm = se - TOPAY_EP * spi;
            double b1v[7], b2v[7];
#pragma unroll
            for (int q = 0; q < 7; q++) { b1v[q] = readlane_f64(g1[q], src); b2v[q] = readlane_f64(g2[q], src); }
#pragma unroll
            for (int r = 0; r < RMAX; r++) {
              if (ract[r] && rpiece[r] == spi) {
                const int k1 = rk[r] >= 1 ? rk[r] - 1 : 0, k2 = rk[r] >= 2 ? rk[r] - 2 : 0;
                const double b1 = rbh[r][1] * c_pw[12 * smm + k1], b2 = rbh[r][2] * c_pw[12 * smm + k2];
#pragma unroll
                for (int q = 0; q < 7; q++) {
                  double a = aq[r][q];
                  a = fma(b1, b1v[q], a);
                  a = fma(b2, b2v[q], a);
                  aq[r][q] = a;
                }
              }
            }
          }
        }
      }
      lds_sync();   // end of the pass: the pass buffer is free again
    }
  } else {
  for (int k = 0; k < nround; k++) {
    const int pass = k * NW + wave;
    const int e = pass * 64 + lane;
    const bool act = pass < npass && e < NE;
    const int i = act ? e / TOPAY_EP : N - 1;
    const int m = act ? e - TOPAY_EP * i : 0;
    const int j = 2 * m;
    const double step = c_Tp[i] / TOPAY_K, half = step / 2.0;
    bool jva = false;
    if (pass < npass) {
      glb_cdp sb = c_sbuf + (act ? e : NE - 1);
      const int ss = c_sbs;
      double raw[14];
#pragma unroll
      for (int v = 0; v < ((STAGE == 2) ? 14 : 6); v++) raw[v] = sb[v * ss];
#pragma unroll
      for (int v = 0; v < 6; v++) pbuf[v * 64 + lane] = act ? raw[v] : 0.0;
      if (STAGE == 2) {
#pragma unroll
        for (int v = 0; v < 7; v++) pbuf[(6 + v) * 64 + lane] = act ? raw[6 + v] : 0.0;
        jva = act && raw[13] != 0.0;
      }
    }
    if (STAGE == 2) {
      const unsigned long long mk = __ballot(jva);
      if (lane == 0) jmask[wave] = mk;
    }
    wg_barrier<NW>();
    // row accumulation of one pass of the round into the accumulators of row slot r
    auto accumulate = [&](int r, int q) __attribute__((always_inline)) {
      const int p2 = k * NW + q;
      const int pi = rpiece[r];
      const double h0 = rbh[r][0], h1 = rbh[r][1], h2 = rbh[r][2];
      const int k0 = rk[r], k1 = rk[r] >= 1 ? rk[r] - 1 : 0, k2 = rk[r] >= 2 ? rk[r] - 2 : 0;
      const int e_lo = max(TOPAY_EP * pi, p2 * 64), e_hi = min(min(TOPAY_EP * pi + TOPAY_EP, p2 * 64 + 64), NE);
      if (e_lo >= e_hi) return;
      lds_cdp pq = pball + q * (PBR * 64);
      double gt = 0.0;
      constexpr int CH = 3;
      for (int c0 = e_lo; c0 < e_hi; c0 += CH) {
        double pb[CH][NV], t0[CH], t1[CH], t2[CH];
#pragma unroll
        for (int u = 0; u < CH; u++) {
          const int ee = (c0 + u < e_hi) ? c0 + u : e_hi - 1;
          const int l = ee - p2 * 64, mm = ee - TOPAY_EP * pi;
          lds_cdp pj = c_pw + 12 * mm;
          t0[u] = pj[k0]; t1[u] = pj[k1]; t2[u] = pj[k2];
#pragma unroll
          for (int v = 0; v < NV; v++) pb[u][v] = pq[v * 64 + l];
        }
#pragma unroll
        for (int u = 0; u < CH; u++) {
          const bool ok = c0 + u < e_hi;
          const double b0 = h0 * t0[u], b1 = h1 * t1[u], b2 = h2 * t2[u];
          const double i0 = fma(b2, pb[u][2], fma(b1, pb[u][1], b0 * pb[u][0]));
          const double i1 = fma(b2, pb[u][4], b1 * pb[u][3]);
          a0[r] += ok ? i0 : 0.0;
          a1[r] += ok ? i1 : 0.0;
          gt += ok ? pb[u][5] : 0.0;
          if (STAGE == 2) {
#pragma unroll
            for (int qq = 0; qq < 7; qq++) {
              const double nq = fma(b0, pb[u][6 + qq], aq[r][qq]);
              aq[r][qq] = ok ? nq : aq[r][qq];
            }
          }
        }
      }
      if (rk[r] == 0) c_gdT[pi] += gt;
    };
    unsigned long long any = 0;
    if (STAGE == 2) {
#pragma unroll
      for (int q = 0; q < NW; q++) any |= jmask[q];
    }
    if (any == 0) {
      // the samples of the row's piece in ascending order, pass by pass (the one-wave order)
#pragma unroll
      for (int r = 0; r < RMAX; r++) {
        if (ract[r]) {
          for (int q = 0; q < NW; q++) accumulate(r, q);
        }
      }
    } else {
      // rare: joint velocity / acceleration gradBeta rows 1 and 2 (moma_traj_opt.cpp:1689, 1703) of flagged samples.  Within a
      // pass the order-0 rows of all its samples first, then the rare rows of its flagged samples, pass by pass -- the
      // one-wave order exactly, also for a piece whose samples straddle two passes of this round.
      for (int q = 0; q < NW; q++) {
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
          if (ract[r]) accumulate(r, q);
        }
        unsigned long long todo = jmask[q];   // (the same for every thread: the barriers below are uniform)
        if (todo == 0) continue;
        wg_barrier<NW>();   // every row thread is done with this pass's buffer
        if (wave == q && jva) {
          Basis B;
          make_basis(j * half, B);
          const double omg = (j == 0 || j == 2 * TOPAY_K) ? 0.5 : 1.0;
#pragma unroll
          for (int qq = 0; qq < 7; qq++) {
            double p0, p1, p2, g1 = 0.0, g2 = 0.0;
            poly3(cL, rows, i, 2 + qq, B, p0, p1, p2);
            const double vDq = p1 * p1 - P.joint_vel_limit2[qq];
            const double vD2q = p2 * p2 - P.joint_acc_limit2[qq];
            if (vDq > 0) {
              double pe, pd;
              smoothL1(P, vDq, P.relu_mu, pe, pd);
              g1 = omg * step * P.s2_mani_vel_weight * pd * 2.0 * p1;
            }
            if (vD2q > 0) {
              double pe, pd;
              smoothL1(P, vD2q, P.relu_mu, pe, pd);
              g2 = omg * step * P.s2_mani_acc_weight * pd * 2.0 * p2;
            }
            pbuf[qq * 64 + lane] = g1;
            pbuf[(7 + qq) * 64 + lane] = g2;
          }
        }
        wg_barrier<NW>();
        lds_cdp pq = pball + q * (PBR * 64);
        while (todo) {
          const int src = __ffsll(todo) - 1;
          todo &= todo - 1;
          const int se = (k * NW + q) * 64 + src;
          const int spi = se / TOPAY_EP, smm = se - TOPAY_EP * spi;
          double b1v[7], b2v[7];
#pragma unroll
          for (int qq = 0; qq < 7; qq++) { b1v[qq] = pq[qq * 64 + src]; b2v[qq] = pq[(7 + qq) * 64 + src]; }
#pragma unroll
          for (int r = 0; r < RMAX; r++) {
            if (ract[r] && rpiece[r] == spi) {
              const int k1 = rk[r] >= 1 ? rk[r] - 1 : 0, k2 = rk[r] >= 2 ? rk[r] - 2 : 0;
              const double b1 = rbh[r][1] * c_pw[12 * smm + k1], b2 = rbh[r][2] * c_pw[12 * smm + k2];
#pragma unroll
              for (int qq = 0; qq < 7; qq++) {
                double a = aq[r][qq];
                a = fma(b1, b1v[qq], a);
                a = fma(b2, b2v[qq], a);
                aq[r][qq] = a;
              }
            }
          }
        }
      }
    }
    wg_barrier<NW>();   // end of the round: the pass buffers and the masks are free again
  }

  }
  if (STAGE == 2) {
    const double all = wg_sum<NW>(c_red, rp, wave, mt_add_all);
    if (tid < N) c_gdT[tid] += all + mt_add_own;
  }
  wg_barrier<NW>();
  // =========================== sweep 2: backward, XY-gradient chain ===========================
  if (!bad) {
    if (STAGE == 2) {   // phase A: pass totals of the positional gradients in the suffix scan's own order
      for (int k = 0; k < nround; k++) {
        const int pass = k * NW + wave;
        if (pass < npass) {
          const int e = pass * 64 + lane;
          const bool act = e < NE;
          const double gx = act ? gxy[2 * e] : 0.0, gy = act ? gxy[2 * e + 1] : 0.0;
          const double sx_ = wave_incl_rscan(gx, lane), sy_ = wave_incl_rscan(gy, lane);
          if (lane == 0) {
            ptot[2 * pass] = sx_;
            ptot[2 * pass + 1] = sy_;
          }
        }
      }
      wg_barrier<NW>();
    }
    double rcx = chain0x, rcy = chain0y;   // chain carried across passes: the totals of the passes after `pc - 1`, added in descending order
    pc = npass;
    for (int k = nround - 1; k >= 0; k--) {
      const int pass = k * NW + wave;
      double v0 = 0, v1 = 0, v2 = 0, v3 = 0, vT = 0;
      if (pass < npass) {
        const int e = pass * 64 + lane;
        const bool act = e < NE;
        const int i = act ? e / TOPAY_EP : N - 1;
        const int m = act ? e - TOPAY_EP * i : 0;
        const int j = 2 * m;
        double chx_in, chy_in, chx_ex, chy_ex;
        if (STAGE == 2) {
          while (pc > pass + 1) {
            pc--;
            rcx += ptot[2 * pc];
            rcy += ptot[2 * pc + 1];
          }
          const double gx = act ? gxy[2 * e] : 0.0, gy = act ? gxy[2 * e + 1] : 0.0;
          const double sx_ = wave_incl_rscan(gx, lane), sy_ = wave_incl_rscan(gy, lane);
          chx_in = sx_ + rcx; chy_in = sy_ + rcy;
          chx_ex = chx_in - gx;   chy_ex = chy_in - gy;
        } else {
          double sx_ = 0.0, sy_ = 0.0;
          for (int ii = i + 1; ii < N; ii++) { sx_ += c_pcs[2 * ii]; sy_ += c_pcs[2 * ii + 1]; }
          chx_in = chx_ex = sx_;
          chy_in = chy_ex = sy_;
        }
        if (act) {
          const double T1 = c_Tp[i];
          const double step = T1 / TOPAY_K, half = step / 2.0, coeff = step / 6.0;
          const int int_6K = TOPAY_K * 6;
#pragma unroll
          for (int odd = 0; odd < 2; odd++) {
            if (odd == 1 && m == TOPAY_K) break;
            const int jj = j + odd;
            Basis B;
            make_basis(jj * half, B);
            double th0, th1, th2, s0, sd1, sd2;
            poly3(cL, rows, i, 0, B, th0, th1, th2);
            poly3(cL, rows, i, 1, B, s0, sd1, sd2);
            double sn, cn;
            det_sincos(th0, &sn, &cn);
            const double alpha = 1.0 / (2 * TOPAY_K) * jj;
            const double W = odd ? 4.0 : ((jj == 0 || jj == 2 * TOPAY_K) ? 1.0 : 2.0);
            const double Cx = (odd ? chx_ex : chx_in) * W, Cy = (odd ? chy_ex : chy_in) * W;
            const double aTh = (-sd1 * sn * coeff) * Cx + (sd1 * cn * coeff) * Cy;
            const double aS = (cn * coeff) * Cx + (sn * coeff) * Cy;
            const double gTx = (sd2 * cn - sd1 * th1 * sn) * alpha * coeff + sd1 * cn / int_6K;
            const double gTy = (sd2 * sn + sd1 * th1 * cn) * alpha * coeff + sd1 * sn / int_6K;
            vT += gTx * Cx + gTy * Cy;
            if (odd) { v2 = aTh; v3 = aS; } else { v0 = aTh; v1 = aS; }
          }
        }
      }
      pbuf[0 * 64 + lane] = v0; pbuf[1 * 64 + lane] = v1; pbuf[2 * 64 + lane] = v2; pbuf[3 * 64 + lane] = v3;
      pbuf[4 * 64 + lane] = vT;
      wg_barrier<NW>();
#pragma unroll
      for (int r = 0; r < RMAX; r++) {
        if (ract[r]) {
          const int pi = rpiece[r];
          const double h0 = rbh[r][0], h1 = rbh[r][1];
          const int k1 = rk[r] >= 1 ? rk[r] - 1 : 0;
          for (int q = NW - 1; q >= 0; q--) {   // passes in descending order, samples inside a pass ascending
            const int p2 = k * NW + q;
            const int e_lo = max(TOPAY_EP * pi, p2 * 64), e_hi = min(min(TOPAY_EP * pi + TOPAY_EP, p2 * 64 + 64), NE);
            if (e_lo >= e_hi) continue;
            lds_cdp pq = pball + q * (PBR * 64);
            double gt = 0.0;
            constexpr int CH = 3;
            for (int c0 = e_lo; c0 < e_hi; c0 += CH) {
              double pb[CH][5], tb[CH][4];
#pragma unroll
              for (int u = 0; u < CH; u++) {
                const int ee = (c0 + u < e_hi) ? c0 + u : e_hi - 1;
                const int l = ee - p2 * 64, mm = ee - TOPAY_EP * pi;
                lds_cdp pj = c_pw + 12 * mm;
                tb[u][0] = pj[rk[r]]; tb[u][1] = pj[k1]; tb[u][2] = pj[6 + rk[r]]; tb[u][3] = pj[6 + k1];
#pragma unroll
                for (int v = 0; v < 5; v++) pb[u][v] = pq[v * 64 + l];
              }
#pragma unroll
              for (int u = 0; u < CH; u++) {
                const bool ok = c0 + u < e_hi;
                const double b0 = h0 * tb[u][0], b1 = h1 * tb[u][1];
                const double o0 = h0 * tb[u][2], o1 = h1 * tb[u][3];
                const double i0 = fma(o0, pb[u][2], b0 * pb[u][0]);
                const double i1 = fma(o1, pb[u][3], b1 * pb[u][1]);
                a0[r] += ok ? i0 : 0.0;
                a1[r] += ok ? i1 : 0.0;
                gt += ok ? pb[u][4] : 0.0;
              }
            }
            if (rk[r] == 0) c_gdT[pi] += gt;
          }
        }
      }
      wg_barrier<NW>();
    }
  } else {
    penalty_cost = 1.0e+22;
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
      a0[r] = 0.0;
      a1[r] = 0.0;
#pragma unroll
      for (int q = 0; q < 7; q++) aq[r][q] = 0.0;
    }
    if (tid < N) c_gdT[tid] = 0.0;
    wg_barrier<NW>();
  }

  STAMP(C, 6);  // gradient rows to the row threads, sweep 2
  // ---- total dJ/dC = jerk part (minco.hpp:951-976) + penalty part; adjoint solve (banded_system.hpp:123-145)
  double tot[RMAX][9];
#pragma unroll
  for (int r = 0; r < RMAX; r++) {
    if (ract[r]) {
      const int pi = rpiece[r], kk = rk[r];
      const double T1 = c_Tp[pi], T2 = c_Tp[N + pi], T3 = c_Tp[2 * N + pi], T4 = c_Tp[3 * N + pi], T5 = c_Tp[4 * N + pi];
#pragma unroll
      for (int d = 0; d < 9; d++) {
        double jg = 0.0;
        if (kk >= 3) {
          const double c3 = cL[d * rows + 6 * pi + 3], c4 = cL[d * rows + 6 * pi + 4], c5 = cL[d * rows + 6 * pi + 5];
          const double e = P.energy_weights[d];
          if (kk == 5) jg = 240.0 * c3 * e * T3 + 720.0 * c4 * e * T4 + 1440.0 * c5 * e * T5;
          else if (kk == 4) jg = 144.0 * c3 * e * T2 + 384.0 * c4 * e * T3 + 720.0 * c5 * e * T4;
          else jg = 72.0 * c3 * e * T1 + 144.0 * c4 * e * T2 + 240.0 * c5 * e * T3;
        }
        tot[r][d] = jg + (d == 0 ? a0[r] : (d == 1 ? a1[r] : aq[r][d >= 2 ? d - 2 : 0]));
      }
    }
  }
  // the adjoint solve takes the coefficients' LDS block: they go to the candidate's result block in HBM first (which
  // is where a solve that ends here leaves them anyway); the dJ/dT correction below reads them back from there
  for (int t = tid; t < 9 * rows; t += NT) c_coefg[t] = cL[t];
  C.cl_in_lds = 0;
  wg_global_barrier<NW>();
  // (the LU factors stream from the candidate's LU block through the windows of band_sweep: nothing to reload)
#pragma unroll
  for (int r = 0; r < RMAX; r++) {
    if (ract[r]) {
#pragma unroll
      for (int d = 0; d < 9; d++) c_adj[d * rows + rrow[r]] = tot[r][d];
    }
  }
  wg_barrier<NW>();
  if (wave == 0) {
    const bool owner = lane < 9;
    const lds_dp mine = c_adj + (owner ? lane : 8) * rows;
    band_sweep<2>(mine, owner, (glb_cdp)c_lu, c_X, rows, lane);
    band_sweep<3>(mine, owner, (glb_cdp)c_lu, c_X, rows, lane);
  }
  wg_barrier<NW>();
  STAMP(C, 7);  // adjoint solve
  // ---- dJ/dT correction  gdT(i) += sum(B1 .* adj rows 6i+3..6i+8) — minco.hpp:1016-1067
#pragma unroll
  for (int r = 0; r < RMAX; r++) {
    if (ract[r]) {
      const int row = rrow[r];
      int pi, br;
      bool use = true;
      if (row >= rows - 3) { pi = N - 1; br = 10 + (row - (rows - 3)); }
      else if (row < 3) { use = false; pi = 0; br = 0; }
      else { pi = (row - 3) / 6; br = (row - 3) - 6 * pi; }
      double part = 0.0;
      if (use) {
        const double T1 = c_Tp[pi], T2 = c_Tp[N + pi], T3 = c_Tp[2 * N + pi], T4 = c_Tp[3 * N + pi];
#pragma unroll
        for (int d = 0; d < 9; d++) {
          glb_cdp cg = c_coefg + d * rows + 6 * pi;   // (the LDS block holds the adjoint now)
          const double c1 = cg[1], c2 = cg[2], c3 = cg[3], c4 = cg[4], c5 = cg[5];
          double b;
          if (br == 0) b = -(24.0 * c4 + 120.0 * T1 * c5);
          else if (br == 1) b = -120.0 * c5;
          else if (br == 2 || br == 3 || br == 10) b = -(c1 + 2.0 * T1 * c2 + 3.0 * T2 * c3 + 4.0 * T3 * c4 + 5.0 * T4 * c5);
          else if (br == 4 || br == 11) b = -(2.0 * c2 + 6.0 * T1 * c3 + 12.0 * T2 * c4 + 20.0 * T3 * c5);
          else b = -(6.0 * c3 + 24.0 * T1 * c4 + 60.0 * T2 * c5);
          part += b * c_adj[d * rows + row];
        }
      }
      c_X[row] = part;   // (the windows of the sweeps are dead)
    }
  }
  wg_barrier<NW>();
  double gdT_tot = 0.0;
  if (tid < N) {
    const int i = tid;
    double s = 0.0;
    if (i < N - 1) { for (int r = 0; r < 6; r++) s += c_X[6 * i + 3 + r]; }
    else { for (int r = 0; r < 3; r++) s += c_X[rows - 3 + r]; }
    gdT_tot = jerk_gdT + c_gdT[i] + s;
  }
  // ---- chain rule to the decision variables — moma_traj_opt.cpp:936-948
  glb_cdp Tau = c_x;
  glb_cdp Vq = c_x + 3 * N - 1;
  if (tid < N) c_g[tid] = (gdT_tot + wT) * dTdTau(Tau[tid]);
  for (int t = tid; t < 9 * (N - 1); t += NT) {
    const int i = t / 9, d = t - 9 * i;
    const double gp = c_adj[d * rows + 6 * i + 5];
    const int dq = d >= 2 ? d - 2 : 0;
    if (d == 0) c_g[N + i] = gp;
    else if (d == 1) c_g[2 * N - 1 + i] = gp;
    else c_g[3 * N - 1 + 7 * i + dq] = gp * dQdVq(Vq[7 * i + dq], P.joint_pos_limit_max[dq]);
  }
  if (tid == 0) c_g[3 * N - 2] = c_adj[1 * rows + rows - 3];
  wg_global_barrier<NW>();
  STAMP(C, 8);  // dJ/dT correction, chain rule
  return f_total;
}

}  // namespace topay
