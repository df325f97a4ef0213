// Feasibility gate of a finished trajectory — printConstraintsSituations (planner/include/planner/moma_traj_opt.h:
// 1052-1204, the check the planner applies to every optimised candidate, planner.cpp:878-880) and checkFeasible
// (948-1050), with the MomaTraj playback they sample through (moma_traj_opt.h:26-137: car_seq every 0.1 s from
// Simpson panels of 0.025 s, getState = car_seq entry + one partial Simpson panel).
//
// One wavefront per trajectory:
//   1. car_seq: lanes <-> Simpson panels, wave inclusive scan with a carry across passes, prefix after every panel
//      kept in HBM scratch (the reference keeps every 4th; the others are free).
//   2. the sample times t = 0, 0.01, 0.02, ... are the reference's running sum (t += res), produced serially by one
//      lane so that the number of samples and their values are the reference's.
//   3. lanes <-> samples: state, velocity, acceleration, the 12 sphere centres, 1 + 12 ESDF values; lane-local
//      extrema, wave max / min at the end.
// report[38] = { |v|, |a|, |omega|, |domega| maxima, |q| max[7], |dq| max[7], |d2q| max[7], chassis min distance,
// sphere min distance[12] }.
#pragma once
#include "topay_eval.h"

namespace topay {

struct FeasIO {
  const double* coef;   // [9][rows], rows = 6N, element d*rows + 6i + k = coefficient of t^k
  const double* T;      // [N]
  int N;
  double x0, y0, th0;   // start_state.head(3)
  double* cseq;         // scratch [(panels + 1)][2]
  double* tk;           // scratch [samples]
  long long cap_panels, cap_samples;
  double* report;       // [38]
  int* feasible;        // [2]: printConstraintsSituations, checkFeasible
  // Counter bumped (verdicts left 0 / 0, nothing sampled) when the scratch cannot hold the trajectory's panels and sample
  // times: the in-solve gate borrows the candidate's L-BFGS history block, which a small mem_size makes short -- the host
  // then gates the batch with the separate kernel and scratch of the right size.  Null: scratch sized by the host.
  int* truncated;
};

// PolyTrajectory::locatePieceIdx (minco.hpp:356-374): t becomes the local time
__device__ __forceinline__ int feas_locate(const double* T, int N, double& t) {
  int idx;
  double dur = 0.0;
  for (idx = 0; idx < N && t > (dur = T[idx]); idx++) t -= dur;
  if (idx == N) { idx--; t += T[idx]; }
  return idx;
}
// Piece::getPos / getVel / getAcc (minco.hpp:103-150) of dimension d
__device__ __forceinline__ double feas_pos(const double* c, double t) {
  double v = 0.0, tn = 1.0;
#pragma unroll
  for (int k = 0; k <= 5; k++) { v += tn * c[k]; tn *= t; }
  return v;
}
__device__ __forceinline__ double feas_vel(const double* c, double t) {
  double v = 0.0, tn = 1.0;
#pragma unroll
  for (int k = 1; k <= 5; k++) { v += (double)k * tn * c[k]; tn *= t; }
  return v;
}
__device__ __forceinline__ double feas_acc(const double* c, double t) {
  double v = 0.0, tn = 1.0;
#pragma unroll
  for (int k = 2; k <= 5; k++) { v += (double)((k - 1) * k) * tn * c[k]; tn *= t; }
  return v;
}
// theta and arc-length rate at global time t: (p.x, v.y) of the reference's 2-vectors
__device__ __forceinline__ void feas_theta_sdot(const FeasIO& F, double t, double& th, double& sd) {
  const int i = feas_locate(F.T, F.N, t);
  const int rows = 6 * F.N;
  th = feas_pos(F.coef + 0 * rows + 6 * i, t);
  sd = feas_vel(F.coef + 1 * rows + 6 * i, t);
}
__device__ __forceinline__ void feas_simpson(const FeasIO& F, double ta, double tb, double tc, double w6, double& ix,
                                             double& iy) {
  double th1, sd1, th2, sd2, th3, sd3, s1, c1, s2, c2, s3, c3;
  feas_theta_sdot(F, ta, th1, sd1);
  feas_theta_sdot(F, tb, th2, sd2);
  feas_theta_sdot(F, tc, th3, sd3);
  det_sincos(th1, &s1, &c1);
  det_sincos(th2, &s2, &c2);
  det_sincos(th3, &s3, &c3);
  ix = w6 * (sd1 * c1 + 4.0 * sd2 * c2 + sd3 * c3);
  iy = w6 * (sd1 * s1 + 4.0 * sd2 * s2 + sd3 * s3);
}
// value-only lookups: out of the map -> 1e10 (grid_map.h:256-362)
__device__ __forceinline__ double feas_dist2d(const DevMap& M, double px, double py) {
  const bool in = !(px < M.min_b[0] + 1e-4 || py < M.min_b[1] + 1e-4 || px > M.max_b[0] - 1e-4 || py > M.max_b[1] - 1e-4);
  double d = 0.0, gx, gy;
  if (in) esdf2d_query(M, px, py, d, gx, gy);
  return in ? d : 1.0e+10;
}
__device__ __forceinline__ double feas_dist3d(const DevMap& M, double px, double py, double pz) {
  const bool in = !(px < M.min_b[0] + 1e-4 || py < M.min_b[1] + 1e-4 || pz < M.min_b[2] + 1e-4 ||
                    px > M.max_b[0] - 1e-4 || py > M.max_b[1] - 1e-4 || pz > M.max_b[2] - 1e-4);
  double d, gx, gy, gz;
  esdf3d_query(M, px, py, pz, d, gx, gy, gz);
  return in ? d : 1.0e+10;
}
__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }

__device__ __forceinline__ void feasibility_gate(const FeasIO& F, const TOPAY_GLB DevMap* mp) {
  dev_params_ref P = dev_params();
  const int lane = threadIdx.x & 63;
  const int N = F.N, rows = 6 * N;
  const DevMap M = load_map(mp);
  double Ttot = 0.0;
  for (int i = 0; i < N; i++) Ttot += F.T[i];  // getTotalDuration (minco.hpp:304-313)
  if (!(Ttot > 0.0 && Ttot < 1.0e4)) {         // no trajectory (failed solve left NaNs): infeasible, nothing to sample
    if (lane == 0) {
      for (int k = 0; k < 38; k++) F.report[k] = 0.0 / 0.0;
      F.feasible[0] = 0;
      F.feasible[1] = 0;
    }
    return;
  }

  // ---- 1. car_seq
  const double seq_res = 0.1;
  const int approx_res = 4;
  const double h = seq_res / approx_res, hh = h / 2.0, h6 = h / 6.0;
  long long num = (long long)floor(Ttot / h);
  if (num > F.cap_panels || (long long)(Ttot / 0.01) + 2 > F.cap_samples) {
    // the scratch cannot hold this trajectory: nothing is sampled (a truncated sweep would miss the tail's violations)
    if (lane == 0) {
      for (int k = 0; k < 38; k++) F.report[k] = 0.0 / 0.0;
      F.feasible[0] = 0;
      F.feasible[1] = 0;
      if (F.truncated) {
#ifndef TOPAY_CPU_EMU
        __hip_atomic_fetch_add(F.truncated, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
        F.truncated[0] += 1;
#endif
      }
    }
    return;
  }
  if (lane == 0) { F.cseq[0] = F.x0; F.cseq[1] = F.y0; }
  double carryx = 0.0, carryy = 0.0;
  for (long long p0 = 0; p0 < num; p0 += 64) {
    const long long i = p0 + lane;
    double ix = 0.0, iy = 0.0;
    if (i < num) {
      // p1 of panel i is the reference's p3 of panel i-1, i.e. evaluated at (i-1)*h + h
      const double ta = i == 0 ? 0.0 : (double)(i - 1) * h + h;
      feas_simpson(F, ta, (double)i * h + hh, (double)i * h + h, h6, ix, iy);
    }
    const double sx = wave_incl_scan(ix, lane), sy = wave_incl_scan(iy, lane);
    if (i < num) {
      F.cseq[2 * (i + 1)] = F.x0 + (carryx + sx);
      F.cseq[2 * (i + 1) + 1] = F.y0 + (carryy + sy);
    }
    carryx += __shfl(sx, 63);
    carryy += __shfl(sy, 63);
  }
  // ---- 2. sample times (t += res from 0 while t < T)
  long long nsamp = 0;
  if (lane == 0) {
    const double res = 0.01;
    long long k = 0;
    for (double t = 0.0; t < Ttot && k < F.cap_samples; t += res) F.tk[k++] = t;
    nsamp = k;
  }
  wave_global_sync();  // cseq / tk stores become visible to the other lanes
  nsamp = (long long)__shfl((int)nsamp, 0);

  // ---- 3. samples
  double mv = 0, ma = 0, mw = 0, mdw = 0, mq[7], mdq[7], md2q[7], mind = 1.0e+10, minm[TOPAY_NSPH];
#pragma unroll
  for (int q = 0; q < 7; q++) { mq[q] = 0.0; mdq[q] = 0.0; md2q[q] = 0.0; }
#pragma unroll
  for (int k = 0; k < TOPAY_NSPH; k++) minm[k] = 1.0e+10;
  for (long long s0 = 0; s0 < nsamp; s0 += 64) {
    const long long s = s0 + lane;
    if (s < nsamp) {
      const double tg = F.tk[s];
      // getState (moma_traj_opt.h:113-137); t is already inside [0, T)
      const int index = (int)floor(tg / seq_res);
      const double floor_t = index * seq_res, diff_t = tg - floor_t;
      long long pidx = (long long)index * approx_res;
      if (pidx > num) pidx = num;
      double ix, iy;
      feas_simpson(F, floor_t, floor_t + diff_t / 2.0, tg, diff_t / 6.0, ix, iy);
      const double px = F.cseq[2 * pidx] + ix, py = F.cseq[2 * pidx + 1] + iy;
      double tl = tg;
      const int i = feas_locate(F.T, N, tl);
      double pos[10], sq[7], cq[7];
      pos[0] = px; pos[1] = py;
      pos[2] = feas_pos(F.coef + 0 * rows + 6 * i, tl);
      const double v_th = feas_vel(F.coef + 0 * rows + 6 * i, tl), v_s = feas_vel(F.coef + 1 * rows + 6 * i, tl);
      const double a_th = feas_acc(F.coef + 0 * rows + 6 * i, tl), a_s = feas_acc(F.coef + 1 * rows + 6 * i, tl);
      mv = fmax(mv, fabs(v_s)); ma = fmax(ma, fabs(a_s)); mw = fmax(mw, fabs(v_th)); mdw = fmax(mdw, fabs(a_th));
#pragma unroll
      for (int q = 0; q < 7; q++) {
        const double* c = F.coef + (2 + q) * rows + 6 * i;
        pos[3 + q] = feas_pos(c, tl);
        mq[q] = fmax(mq[q], fabs(pos[3 + q]));
        mdq[q] = fmax(mdq[q], fabs(feas_vel(c, tl)));
        md2q[q] = fmax(md2q[q], fabs(feas_acc(c, tl)));
        det_sincos(pos[3 + q], &sq[q], &cq[q]);
      }
      mind = fmin(mind, feas_dist2d(M, px, py));
      // sphere centres — moma_param.h:203-247 (same walk as manipulator_block)
      double sth, cth;
      det_sincos(pos[2], &sth, &cth);
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
      const double p0z = P.chassis_height + P.relT[2];
      double R[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
      double q0 = 0.0, q1 = 0.0, q2 = 0.0;
      int sidx = 0;
#pragma unroll
      for (int li = 0; li < 8; li++) {
        const int cnt = (li % 2 == 0) ? 2 : 1;
#pragma unroll
        for (int c = 0; c < cnt; c++) {
          const double lx = fma(R[2], P.sph_off[sidx], q0), ly = fma(R[5], P.sph_off[sidx], q1), lz = fma(R[8], P.sph_off[sidx], q2);
          const double wx = p0x + fma(A[2], lz, fma(A[1], ly, A[0] * lx));
          const double wy = p0y + fma(A[5], lz, fma(A[4], ly, A[3] * lx));
          const double wz = p0z + fma(A[8], lz, fma(A[7], ly, A[6] * lx));
          minm[sidx] = fmin(minm[sidx], feas_dist3d(M, wx, wy, wz));
          sidx++;
        }
        q0 = fma(R[2], P.colli_length[li], q0);
        q1 = fma(R[5], P.colli_length[li], q1);
        q2 = fma(R[8], P.colli_length[li], q2);
        if (li == 7) break;
        joint_rotate(R, li, cq[li], sq[li]);
      }
    }
  }
  // ---- verdicts
  mv = wave_max(mv); ma = wave_max(ma); mw = wave_max(mw); mdw = wave_max(mdw);
  mind = wave_min(mind);
  bool feasible = true;
  if (mv > 1.01 * P.max_v) feasible = false;
  if (ma > 1.01 * P.max_a) feasible = false;
  if (mw > 1.01 * P.max_w) feasible = false;
  if (mdw > 1.01 * P.max_dw) feasible = false;
#pragma unroll
  for (int q = 0; q < 7; q++) {
    mq[q] = wave_max(mq[q]); mdq[q] = wave_max(mdq[q]); md2q[q] = wave_max(md2q[q]);
    if (mq[q] > 1.01 * P.joint_pos_limit_max[q]) feasible = false;
    if (mdq[q] > 1.01 * P.joint_vel_limit[q]) feasible = false;
    if (md2q[q] > 1.01 * P.joint_acc_limit[q]) feasible = false;
  }
  if (mind < 0.99 * P.chassis_colli_radius) feasible = false;
  bool strict = feasible;
#pragma unroll
  for (int k = 0; k < TOPAY_NSPH; k++) {
    minm[k] = wave_min(minm[k]);
    if (minm[k] < 0.99 * P.sph_r[k]) strict = false;
  }
  if (lane == 0) {
    F.report[0] = mv; F.report[1] = ma; F.report[2] = mw; F.report[3] = mdw;
#pragma unroll
    for (int q = 0; q < 7; q++) { F.report[4 + q] = mq[q]; F.report[11 + q] = mdq[q]; F.report[18 + q] = md2q[q]; }
    F.report[25] = mind;
#pragma unroll
    for (int k = 0; k < TOPAY_NSPH; k++) F.report[26 + k] = minm[k];
    F.feasible[0] = feasible ? 1 : 0;
    F.feasible[1] = strict ? 1 : 0;
  }
}

// MomaTraj playback of one candidate — car_seq (moma_traj_opt.h:40-69) and getState (113-137) at caller-given times.
// One wavefront: car_seq panels by a wave scan exactly as in the gate, then lanes <-> query times.
//   seq_out  [(nseq)][4] = (x, y, theta, t) every 0.1 s, nseq = floor(floor(T / 0.025) / 4) + 1
//   states   [nq][10]    = getState(times[q])
__device__ __forceinline__ void playback(const FeasIO& F, int nq, const double* times, double* states, double* seq_out,
                                         int* nseq_out) {
  const int lane = threadIdx.x & 63;
  const int N = F.N, rows = 6 * N;
  double Ttot = 0.0;
  for (int i = 0; i < N; i++) Ttot += F.T[i];
  if (!(Ttot > 0.0 && Ttot < 1.0e4)) {
    if (lane == 0 && nseq_out) *nseq_out = 0;
    return;
  }
  const double seq_res = 0.1;
  const int approx_res = 4;
  const double h = seq_res / approx_res, hh = h / 2.0, h6 = h / 6.0;
  long long num = (long long)floor(Ttot / h);
  if (num > F.cap_panels) num = F.cap_panels;
  if (lane == 0) { F.cseq[0] = F.x0; F.cseq[1] = F.y0; }
  double carryx = 0.0, carryy = 0.0;
  for (long long p0 = 0; p0 < num; p0 += 64) {
    const long long i = p0 + lane;
    double ix = 0.0, iy = 0.0;
    if (i < num) {
      const double ta = i == 0 ? 0.0 : (double)(i - 1) * h + h;
      feas_simpson(F, ta, (double)i * h + hh, (double)i * h + h, h6, ix, iy);
    }
    const double sx = wave_incl_scan(ix, lane), sy = wave_incl_scan(iy, lane);
    if (i < num) {
      F.cseq[2 * (i + 1)] = F.x0 + (carryx + sx);
      F.cseq[2 * (i + 1) + 1] = F.y0 + (carryy + sy);
    }
    carryx += __shfl(sx, 63);
    carryy += __shfl(sy, 63);
  }
  wave_global_sync();
  const long long nseq = num / approx_res + 1;
  if (seq_out) {
    for (long long k = lane; k < nseq; k += 64) {
      double th = F.th0, tt = 0.0;
      if (k > 0) {
        const long long i = k * approx_res - 1;        // the panel after which the entry is pushed
        tt = (double)(i + 1) * h;
        double tl = (double)i * h + h;                  // p3 of that panel
        const int pi = feas_locate(F.T, N, tl);
        th = feas_pos(F.coef + 0 * rows + 6 * pi, tl);
      }
      seq_out[4 * k] = F.cseq[2 * k * approx_res];
      seq_out[4 * k + 1] = F.cseq[2 * k * approx_res + 1];
      seq_out[4 * k + 2] = th;
      seq_out[4 * k + 3] = tt;
    }
  }
  if (lane == 0 && nseq_out) *nseq_out = (int)nseq;
  for (int q0 = 0; q0 < nq; q0 += 64) {
    const int q = q0 + lane;
    if (q < nq) {
      double tg = times[q];
      tg = fmin(fmax(tg, 0.0), Ttot);
      const int index = (int)floor(tg / seq_res);
      const double floor_t = index * seq_res, diff_t = tg - floor_t;
      long long pidx = (long long)index * approx_res;
      if (pidx > num) pidx = num;
      double ix, iy;
      feas_simpson(F, floor_t, floor_t + diff_t / 2.0, tg, diff_t / 6.0, ix, iy);
      double tl = tg;
      const int i = feas_locate(F.T, N, tl);
      double* o = states + (size_t)q * 10;
      o[0] = F.cseq[2 * pidx] + ix;
      o[1] = F.cseq[2 * pidx + 1] + iy;
      o[2] = feas_pos(F.coef + 0 * rows + 6 * i, tl);
#pragma unroll
      for (int d = 0; d < 7; d++) o[3 + d] = feas_pos(F.coef + (2 + d) * rows + 6 * i, tl);
    }
  }
}

// GridMap::isWholeBodyCollision (src/map/include/map/grid_map.h:613-650): the check the front-end applies to every
// sampled or interpolated state (joint limits, chassis against the 2-D field, every sphere against the 3-D field,
// spheres against the chassis top and against each other).  A point outside the map counts as a collision
// (isCollision2d / isCollision3d, grid_map.h:511-536, 699-725).  One thread per state.
// MomaParam::getColliPts (moma_param.h:203-247): centres of the 12 collision spheres of state st = (x, y, theta, q1..q7)
__device__ __forceinline__ void sphere_centres(const double* st, double (&Px)[TOPAY_NSPH], double (&Py)[TOPAY_NSPH], double (&Pz)[TOPAY_NSPH]) {
  dev_params_ref P = dev_params();
  double sq[7], cq[7], sth, cth;
#pragma unroll
  for (int q = 0; q < 7; q++) det_sincos(st[3 + q], &sq[q], &cq[q]);
  det_sincos(st[2], &sth, &cth);
  double A[9];
  {
    const double Rz[9] = {cth, -sth, 0.0, sth, cth, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++)
        A[a * 3 + b] = Rz[a * 3 + 0] * P.relR[0 * 3 + b] + Rz[a * 3 + 1] * P.relR[1 * 3 + b] + Rz[a * 3 + 2] * P.relR[2 * 3 + b];
  }
  const double p0x = st[0] + (cth * P.relT[0] - sth * P.relT[1]);
  const double p0y = st[1] + (sth * P.relT[0] + cth * P.relT[1]);
  const double p0z = P.chassis_height + P.relT[2];
  double R[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
  double q0 = 0.0, q1 = 0.0, q2 = 0.0;
  int sidx = 0;
#pragma unroll
  for (int li = 0; li < 8; li++) {
    const int cnt = (li % 2 == 0) ? 2 : 1;
#pragma unroll
    for (int c = 0; c < cnt; c++) {
      const double lx = fma(R[2], P.sph_off[sidx], q0), ly = fma(R[5], P.sph_off[sidx], q1), lz = fma(R[8], P.sph_off[sidx], q2);
      Px[sidx] = p0x + fma(A[2], lz, fma(A[1], ly, A[0] * lx));
      Py[sidx] = p0y + fma(A[5], lz, fma(A[4], ly, A[3] * lx));
      Pz[sidx] = p0z + fma(A[8], lz, fma(A[7], ly, A[6] * lx));
      sidx++;
    }
    q0 = fma(R[2], P.colli_length[li], q0);
    q1 = fma(R[5], P.colli_length[li], q1);
    q2 = fma(R[8], P.colli_length[li], q2);
    if (li == 7) break;
    joint_rotate(R, li, cq[li], sq[li]);
  }
}

__device__ __forceinline__ bool whole_body_collision(const DevMap& M, const double* st) {
  dev_params_ref P = dev_params();
  bool hit = false;
#pragma unroll
  for (int q = 0; q < 7; q++) hit = hit || st[3 + q] > P.joint_pos_limit_max[q] || st[3 + q] < -P.joint_pos_limit_max[q];
  {
    const double d = feas_dist2d(M, st[0], st[1]);   // 1e10 outside the map
    const bool in = d < 1.0e+9;
    hit = hit || !in || d < P.chassis_colli_radius;
  }
  double Px[TOPAY_NSPH], Py[TOPAY_NSPH], Pz[TOPAY_NSPH];
  sphere_centres(st, Px, Py, Pz);
#pragma unroll
  for (int i = 0; i < TOPAY_NSPH; i++) {
    const double d = feas_dist3d(M, Px[i], Py[i], Pz[i]);
    hit = hit || !(d < 1.0e+9) || d < P.sph_r[i];
    if (i > 2) {
      const double dx = Px[i] - st[0], dy = Py[i] - st[1];
      hit = hit || (Pz[i] < P.chassis_height + P.sph_r[i] && sqrt(dx * dx + dy * dy) < P.chassis_colli_radius + P.sph_r[i]);
    }
#pragma unroll
    for (int j = i + 2; j < TOPAY_NSPH; j++) {  // collision_matrix == -1: non-adjacent spheres (moma_param.h:128-143)
      const double d0 = Px[i] - Px[j], d1 = Py[i] - Py[j], d2 = Pz[i] - Pz[j];
      hit = hit || sqrt(d0 * d0 + d1 * d1 + d2 * d2) < P.sph_r[i] + P.sph_r[j];
    }
  }
  return hit;
}

// ---------------------------------------------------------------------------------------------------------------
// MomaParam::getMeshPose (moma_param.h:724-790), the per-state work of Planner::toMeshMsg (planner.cpp:2003-2056): poses
// (x, y, z, qw, qx, qy, qz) of chassis, stump, 7 links, end effector (copy of the last link) and the end effector's
// collision point (position only).  The reference composes them with Eigen's rotation types; these are Eigen's generic
// algorithms (AngleAxis -> quaternion, quaternion product, toRotationMatrix, rotation matrix -> quaternion), written
// out so that every state is one thread's straight-line arithmetic.
// ---------------------------------------------------------------------------------------------------------------
struct MeshQuat { double w, x, y, z; };
__device__ __forceinline__ MeshQuat mq_axis(double angle, int axis) {
  double s, c;
  det_sincos(0.5 * angle, &s, &c);
  MeshQuat q{c, 0.0, 0.0, 0.0};
  if (axis == 0) q.x = s * 1.0; else if (axis == 1) q.y = s * 1.0; else q.z = s * 1.0;
  return q;
}
__device__ __forceinline__ MeshQuat mq_mul(const MeshQuat& a, const MeshQuat& b) {
  return MeshQuat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                  a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ MeshQuat mq_euler(double r, double p, double y) { return mq_mul(mq_mul(mq_axis(r, 0), mq_axis(p, 1)), mq_axis(y, 2)); }
__device__ __forceinline__ void mq_matrix(const MeshQuat& q, double (&m)[9]) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  m[0] = 1.0 - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy;
  m[3] = txy + twz; m[4] = 1.0 - (txx + tzz); m[5] = tyz - twx;
  m[6] = txz - twy; m[7] = tyz + twx; m[8] = 1.0 - (txx + tyy);
}
__device__ __forceinline__ MeshQuat mq_from_matrix(const double (&a)[9]) {
  double t = a[0] + a[4] + a[8];
  double c[4];   // x, y, z, w
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    c[3] = 0.5 * t;
    t = 0.5 / t;
    c[0] = (a[7] - a[5]) * t;
    c[1] = (a[2] - a[6]) * t;
    c[2] = (a[3] - a[1]) * t;
  } else {
    int i = 0;
    if (a[4] > a[0]) i = 1;
    if (a[8] > a[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(a[4 * i] - a[4 * j] - a[4 * k] + 1.0);
    c[i] = 0.5 * t;
    t = 0.5 / t;
    c[3] = (a[3 * k + j] - a[3 * j + k]) * t;
    c[j] = (a[3 * j + i] + a[3 * i + j]) * t;
    c[k] = (a[3 * k + i] + a[3 * i + k]) * t;
  }
  return MeshQuat{c[3], c[0], c[1], c[2]};
}
__device__ __forceinline__ void mq_matmul(const double (&a)[9], const double (&b)[9], double (&r)[9]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r[3 * i + j] = (a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j]) + a[3 * i + 2] * b[6 + j];
}
__device__ __forceinline__ void mesh_pose(const topay_mesh_params_t& K, const double* st, double* out /* 11 x 7 */) {
  dev_params_ref P = dev_params();
  auto put = [&](int r, double px, double py, double pz, const MeshQuat& q) {
    double* o = out + 7 * r;
    o[0] = px; o[1] = py; o[2] = pz; o[3] = q.w; o[4] = q.x; o[5] = q.y; o[6] = q.z;
  };
  double ax = st[0], ay = st[1], az = P.chassis_height / 2.0;
  put(0, ax, ay, az, mq_euler(1.5707963267948966, st[2], 0.0));
  MeshQuat aq;
  {
    double s, c;
    det_sincos(st[2] / 2.0, &s, &c);
    aq = MeshQuat{c, 0.0, 0.0, s};
  }
  double A[9], B[9];
  mq_matrix(aq, A);
  ax += (A[0] * P.relT[0] + A[1] * P.relT[1]) + A[2] * P.relT[2];
  ay += (A[3] * P.relT[0] + A[4] * P.relT[1]) + A[5] * P.relT[2];
  az += (A[6] * P.relT[0] + A[7] * P.relT[1]) + A[8] * P.relT[2];
  {
    double RR[9];
#pragma unroll
    for (int t = 0; t < 9; t++) RR[t] = P.relR[t];
    mq_matmul(A, RR, B);
  }
  aq = mq_from_matrix(B);
  put(1, ax, ay, az, aq);
  for (int i = 0; i < 7; i++) {
    double q = st[3 + i];
    q = fmax(K.joint_pos_limit_min[i], fmin(P.joint_pos_limit_max[i], q));
    mq_matrix(aq, A);
    const double l = K.link_length[i];
    ax += (A[0] * 0.0 + A[1] * 0.0) + A[2] * l;
    ay += (A[3] * 0.0 + A[4] * 0.0) + A[5] * l;
    az += (A[6] * 0.0 + A[7] * 0.0) + A[8] * l;
    double O[9], Rq[9], AO[9];
    mq_matrix(mq_euler(K.joint_offset[3 * i], K.joint_offset[3 * i + 1], K.joint_offset[3 * i + 2]), O);
    mq_matrix(mq_euler(K.joint_dof_axis[3 * i] * q, K.joint_dof_axis[3 * i + 1] * q, K.joint_dof_axis[3 * i + 2] * q), Rq);
    mq_matmul(A, O, AO);
    mq_matmul(AO, Rq, B);
    aq = mq_from_matrix(B);
    put(2 + i, ax, ay, az, aq);
  }
  for (int c = 0; c < 7; c++) out[7 * 9 + c] = out[7 * 8 + c];
  double Px[TOPAY_NSPH], Py[TOPAY_NSPH], Pz[TOPAY_NSPH];
  sphere_centres(st, Px, Py, Pz);
  for (int c = 0; c < 7; c++) out[7 * 10 + c] = 0.0;
  out[70] = Px[TOPAY_NSPH - 1]; out[71] = Py[TOPAY_NSPH - 1]; out[72] = Pz[TOPAY_NSPH - 1];
}

}  // namespace topay
