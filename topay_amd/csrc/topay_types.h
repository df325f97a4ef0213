// Device-side parameter/descriptor PODs shared by the kernels and the host C-ABI layer.
#pragma once
#include "../../include/topay.h"

#define TOPAY_K 12          // int_K: Simpson panels per piece (optimizer.yaml:3)
#define TOPAY_SP 25         // samples per piece = 2K+1
#define TOPAY_EP 13         // even ("full") samples per piece = K+1
#define TOPAY_NSPH 12       // collision spheres (moma_param.h:94-109)
// Pieces per trajectory this build solves: 170 (a 255 s trajectory at the reference's 1.5 s sample_interval) -- what the LDS of a
// compute unit holds: 1020 system rows = 4 per thread of a four-wave workgroup and 158 of the 160 KB (topay_eval_mw.h); the
// one-wave solver of that class keeps 28 vector elements per lane (1792 >= 10 N - 8).  Launch classes: N <= 10 / 15 / 21 / 32
// one wave per trajectory (1 / 2 / 2 / 3 system rows per lane), N <= 42 / 64 / 170 four waves in the evaluations (2 / 2 / 4 rows
// per thread) with the solver on the first.  The reference itself has no cap (moma_traj_opt.cpp:245, 300-321); longer
// candidates are reported failed without a solve (success 0, cost NaN, n_pieces 0; bench.py counts them as n_not_launched).
#define TOPAY_MAX_N 170
#define TOPAY_NBUCKET 7
#define TOPAY_WAVE 64

// Address-space qualified pointers.  LDS and HBM pointers travel through structs and (non-inlined) device
// functions; as plain generic pointers every access becomes a FLAT instruction with a full vmcnt/lgkmcnt drain.
// With the qualifiers the compiler emits ds_* for LDS and global_* for HBM.  The CPU lane-emulator build of
// these sources defines both away on its command line.
#ifndef TOPAY_LDS
#define TOPAY_LDS __attribute__((address_space(3)))
#endif
#ifndef TOPAY_GLB
#define TOPAY_GLB __attribute__((address_space(1)))
#endif
#ifndef TOPAY_CST
#if defined(__HIP_DEVICE_COMPILE__)
#define TOPAY_CST __attribute__((address_space(4)))   // constant memory (the parameter block): scalar loads
#else
#define TOPAY_CST                                     // (host pass of the same sources: the qualifier means nothing there)
#endif
#endif
typedef TOPAY_LDS double* lds_dp;
typedef const TOPAY_LDS double* lds_cdp;
typedef TOPAY_GLB double* glb_dp;
typedef const TOPAY_GLB double* glb_cdp;
typedef TOPAY_GLB int* glb_ip;

struct DevMap {
  double origin[3];
  double res, res_inv;
  double min_b[3], max_b[3];
  int dims[3];
  int pad;
  glb_cdp esdf2d;
  glb_cdp esdf3d;
  glb_cdp esdf2d_inflate;    // GridMap::esdf_buffer_2d_inflate / esdf_buffer_2d_critical (front-end fields; null when the
  glb_cdp esdf2d_critical;   // map came through topay_set_map without them)
};

struct DevLbfgs {
  int mem_size, past, max_iterations, max_linesearch;
  double g_epsilon, delta, min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec;
};

struct DevParams {
  double relu_mu;
  // smoothL1Penalty constants (moma_traj_opt.h:812-817), evaluated once on the host exactly as the reference does per call
  double sl_half, sl_f3c, sl_f4c, sl_d2c, sl_d3c;
  double energy_weights[9];
  double s1_time_weight, s1_moment_weight, s1_acc_weight, s1_domega_weight, s1_path_pos_weight;
  double s2_time_weight, s2_moment_weight, s2_acc_weight, s2_domega_weight;
  double s2_collision_weight, s2_mani_colli_weight, s2_self_colli_weight;
  double s2_mani_pos_weight, s2_mani_vel_weight, s2_mani_acc_weight, s2_mean_time_weight;
  double alm_init_lambda[2], alm_init_rho[2], alm_rho_max[2], alm_gamma[2];
  double alm_tolerance;
  int alm_max_outer;
  int alm_work_budget;
  int min_piece_num;
  int pad0_;
  double sample_interval;
  int s1_normal_past, s1_shot_path_past;
  double s1_shot_path_horizon;
  DevLbfgs s1_lbfgs, s2_lbfgs;
  // robot
  double chassis_height, chassis_colli_radius;
  double max_v, max_a, max_w, max_dw;
  double colli_length[8];
  double sph_off[TOPAY_NSPH];   // offset along the link z axis of each sphere
  double sph_r[TOPAY_NSPH];     // radius of each sphere
  double joint_pos_limit_max[7];
  double joint_vel_limit[7];
  double joint_acc_limit[7];
  double relR[9];
  double relT[3];
  // Products and sums of the constants above that the evaluation needs per sample, formed once on the host with the
  // same IEEE operations in the same order (make_dev_params): a wave-uniform f64 product is a VALU instruction on this
  // hardware (there is no scalar f64 unit), and the compiler hoists such products out of the sample loop into registers
  // it then spills.  As constants they are scalar loads at the point of use.
  double pair_rr2[TOPAY_NSPH * TOPAY_NSPH];   // (r_a + r_b)^2
  double sph_viol[TOPAY_NSPH];                // r_k * 10.0 * 1.1
  double sph_top[TOPAY_NSPH];                 // chassis_height + relT[2] + r_k
  double p0z;                                 // chassis_height + relT[2]
  double max_vw, max_a2, max_dw2;             // max_v * max_w, max_a^2, max_dw^2
  double chassis_r105;                        // chassis_colli_radius * 1.05
  double joint_vel_limit2[7], joint_acc_limit2[7];
};

// Per-batch device arrays.  Fixed-size blocks are indexed by the candidate; the variable-length ones (everything sized
// by the candidate's pieces N or its decision vector n = 10 N - 8) are packed by the candidate's own size, at the
// offsets poff[b] = pieces of the candidates before b and noff[b] = decision-vector elements before b -- not strided by
// the longest member of the batch (one 64-piece candidate among 8192 would otherwise double the history of all).
struct DevBatch {
  int B, hist_m;
  const long long* poff;  // [B + 1]
  const long long* noff;  // [B + 1]
  // inputs produced by the init kernel
  int* N;             // [B] pieces
  int* s1_past;       // [B]
  int* map_id;        // [B]
  double* head;       // [B][27]  9x3 col-major start PVA
  double* tail;       // [B][27]
  double* start_xy;   // [B][2]
  double* goal_xy;    // [B][2]
  double* init_xy;    // [B][TOPAY_MAX_N][2]
  double* x0;         // [B][10 TOPAY_MAX_N - 8]  packed initial decision vector (written before N is known)
  // solver state
  double* x;          // [noff]  n per candidate
  double* work;       // [4 noff]  g, xp, gp, d (n each) per candidate
  double* hist_s;     // [m noff]  [m][n] per candidate
  double* hist_y;     // [m noff]
  double* hist_ys;    // [B][m]
  double* hist_alpha; // [B][m]
  double* lu;         // [84 poff]  LU stash (band + reciprocal diagonal), 14 x 6N per candidate
  // outputs
  int* success;       // [B]
  double* cost;       // [B]
  int* stats;         // [B][8]
  double* xyerr;      // [B][2]
  double* coef;       // [54 poff]  6N x 9 col-major per candidate
  double* T;          // [poff]
  double* knots;      // [2 (poff + b)]  N + 1 knots per candidate
  double* alm;        // [B][4] lambda0,1 rho0,1 (eval hook input / solver output)
  double* fout;       // [B] eval hook output
  double* sbuf;       // [182 poff]  [14][13 N] per-sample gradient rows parked between the cost and the gradient phase
  double* mstash;     // [468 poff]  [13 N][36] forces of self-colliding sphere pairs of a sample (rarely touched, topay_eval.h)
  double* start_us;   // [B] start of the solve on the device's constant clock (scheduling diagnostics)
  // persistent launches: one queue per N-class = positions [queue_off[k], queue_off[k] + queue_count[k]) of `order`, handed
  // out through the device counters queue_next[k]; a workgroup of class queue_class drains its own queue, then the
  // queues of the smaller classes (its kernel template and LDS cover them, and results do not depend on the template).
  // queue_next == null: one workgroup per position of `order`.
  int* queue_next;
  int queue_count[TOPAY_NBUCKET], queue_off[TOPAY_NBUCKET], queue_class;
  int queue_lowest;   // 0: drain the smaller classes' queues too; = queue_class: own queue only (TOPAY_STEAL=0, profiling)
  // feasibility gate inside the solve (printConstraintsSituations of the returned trajectory by the wave that solved it;
  // its scratch is the candidate's own, by then dead, L-BFGS history block): verdicts and extremes per candidate
  int gate_in_solve;
  int* gate_truncated; // one counter in pinned host memory: candidates whose history block was too short for the gate's scratch
  int* feas_flags;    // [B][2]
  double* feas_report;// [B][38]
  // Cancellation (planner.cpp:943-952: the candidates of one planning call that are still running 100 ms after the first
  // success are interrupted).  group_id[b] = planning call of candidate b (-1: none); group_tau[g] = smallest work clock
  // (piece-evaluations, the unit of alm_work_budget) at which a candidate of g finished successfully AND passed the gate;
  // a candidate whose own clock exceeds group_tau + cancel_budget stops at its next stage-2 evaluation (the reference's
  // interruption points: moma_traj_opt.cpp:402, 887).  cancel_flag (pinned host memory): topay_cancel, everything stops.
  const int* group_id;   // [B] or null
  int* group_tau;        // [groups]
  int cancel_budget;
  const int* cancel_flag;
  int* interrupted;      // [B]
  int* started;       // one counter in pinned host memory: candidates of this launch that have begun (dispatch gate; may be null)
  int gate_maxN;      // ... counting only candidates with at most this many pieces (the common classes, see topay_optimize_async)
  int* hw_id;         // [B] hardware slot the solve ran on: xcc << 16 | se << 12 | cu << 4 | simd  (scheduling diagnostics)
  double* elapsed_us; // [B] wall time of the solve of this trajectory (constant 100 MHz counter)
  const int* order;   // [B] block -> trajectory map
  double* trace;      // optional [B][trace_cap] f of every evaluation (debug / parity tooling), may be null
  int trace_cap;
};
