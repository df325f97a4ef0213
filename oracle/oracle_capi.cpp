// TEST INFRASTRUCTURE — C entry points of the CPU oracle (see topay_oracle.hpp header comment).
// Loaded with ctypes by oracle/oracle.py.  Never linked into the product.
#include <atomic>
#include <chrono>
#include <string>
#include <limits>
#include <thread>

#include "topay_oracle.hpp"

using namespace topay_oracle;

struct OracleHandle {
  TrajOpt opt;
  Map map;
};

extern "C" {

void* orc_create() {
  OracleHandle* h = new OracleHandle();
  h->opt.map = &h->map;
  return h;
}
void orc_destroy(void* hh) { delete (OracleHandle*)hh; }

// generic parameter setters for tests
int orc_set_param(void* hh, const char* name, double v) {
  Params& p = ((OracleHandle*)hh)->opt.prm;
  std::string n(name);
#define SETD(field) if (n == #field) { p.field = v; return 0; }
#define SETI(field) if (n == #field) { p.field = (int)v; return 0; }
  SETI(int_K) SETI(min_piece_num) SETD(relu_mu) SETD(sample_interval)
  SETD(s1_time_weight) SETD(s1_moment_weight) SETD(s1_acc_weight) SETD(s1_domega_weight) SETD(s1_path_pos_weight)
  SETI(s1_normal_past) SETI(s1_shot_path_past) SETD(s1_shot_path_horizon)
  SETD(s2_time_weight) SETD(s2_moment_weight) SETD(s2_acc_weight) SETD(s2_domega_weight) SETD(s2_collision_weight)
  SETD(s2_mani_colli_weight) SETD(s2_self_colli_weight) SETD(s2_mani_pos_weight) SETD(s2_mani_vel_weight)
  SETD(s2_mani_acc_weight) SETD(s2_mean_time_weight) SETD(alm_tolerance) SETI(alm_max_outer) SETI(alm_work_budget) SETI(exact_chain)
#undef SETD
#undef SETI
  if (n == "s1_max_iterations") { p.s1_lbfgs.max_iterations = (int)v; return 0; }
  if (n == "s2_max_iterations") { p.s2_lbfgs.max_iterations = (int)v; return 0; }
  if (n == "s1_mem_size") { p.s1_lbfgs.mem_size = (int)v; return 0; }
  if (n == "s2_mem_size") { p.s2_lbfgs.mem_size = (int)v; return 0; }
  if (n == "s1_delta") { p.s1_lbfgs.delta = v; return 0; }
  if (n == "s2_delta") { p.s2_lbfgs.delta = v; return 0; }
  if (n == "s2_past") { p.s2_lbfgs.past = (int)v; return 0; }
  return -1;
}

// map buffers are NOT copied: caller keeps them alive
void orc_set_map(void* hh, const double origin[3], double res, const int dims[3], const double min_b[3],
                 const double max_b[3], const double* esdf2d, const double* esdf3d) {
  OracleHandle* h = (OracleHandle*)hh;
  h->map.set(origin, res, dims, esdf2d, esdf3d);
  h->map.setBounds(min_b, max_b);
}

int orc_set_init_traj(void* hh, const double* init_path, int P, const double* bvel, const double* bacc) {
  return ((OracleHandle*)hh)->opt.setInitTraj(init_path, P, bvel, bacc);
}
int orc_piece_num(void* hh) { return ((OracleHandle*)hh)->opt.piece_num; }
int orc_num_vars(void* hh) { return (int)((OracleHandle*)hh)->opt.x.size(); }
void orc_get_x(void* hh, double* x) {
  auto& v = ((OracleHandle*)hh)->opt.x;
  std::memcpy(x, v.data(), v.size() * sizeof(double));
}
void orc_set_x(void* hh, const double* x) {
  auto& v = ((OracleHandle*)hh)->opt.x;
  std::memcpy(v.data(), x, v.size() * sizeof(double));
}
void orc_get_init_state(void* hh, double* start_pva27, double* end_pva27, double* init_inner_xy, double* start_state10,
                        double* end_state10, int* s1_past) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  std::memcpy(start_pva27, o.minco_start_state, 27 * sizeof(double));
  std::memcpy(end_pva27, o.minco_end_state, 27 * sizeof(double));
  std::memcpy(init_inner_xy, o.init_inner_xy.data(), o.init_inner_xy.size() * sizeof(double));
  std::memcpy(start_state10, o.start_state, 10 * sizeof(double));
  std::memcpy(end_state10, o.end_state, 10 * sizeof(double));
  *s1_past = o.s1_past;
}
void orc_set_alm(void* hh, const double lambda[2], const double rho[2]) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  o.alm_lambda[0] = lambda[0]; o.alm_lambda[1] = lambda[1];
  o.alm_rho[0] = rho[0]; o.alm_rho[1] = rho[1];
}
void orc_get_alm(void* hh, double out4[4]) {  // lambda0, lambda1, rho0, rho1 as the last solve left them
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  out4[0] = o.alm_lambda[0]; out4[1] = o.alm_lambda[1]; out4[2] = o.alm_rho[0]; out4[3] = o.alm_rho[1];
}
// one cost/gradient evaluation (stage 1 or 2) at x — the unit of parity
double orc_eval(void* hh, int stage, const double* x, double* g) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  std::vector<double> xx(x, x + o.x.size()), gg;
  double f = stage == 1 ? o.firstStageCost(xx, gg) : o.secondStageCost(xx, gg);
  std::memcpy(g, gg.data(), gg.size() * sizeof(double));
  return f;
}
void orc_get_final_xy_error(void* hh, double e[2]) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  e[0] = o.final_xy_error[0];
  e[1] = o.final_xy_error[1];
}
void orc_get_debug_terms(void* hh, double terms[13]) { std::memcpy(terms, ((OracleHandle*)hh)->opt.dbg, 13 * sizeof(double)); }
// penalty-only pieces for unit tests: returns cost; gdC (6N x 9 col-major) and gdT (N)
double orc_penalty(void* hh, int stage, const double* x, double* gdC, double* gdT) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  std::vector<double> xx(x, x + o.x.size());
  for (double& v : o.dbg) v = 0.0;
  o.unpack(xx);
  double cost;
  std::vector<double> c, t;
  if (stage == 1) o.calFirstStagePenalGrad(cost, c, t);
  else o.calSecondStagePenalGrad(cost, c, t);
  std::memcpy(gdC, c.data(), c.size() * sizeof(double));
  std::memcpy(gdT, t.data(), t.size() * sizeof(double));
  return cost;
}
// coefficients (6N x 9 col-major) and T(N) after the last generate
void orc_get_coeffs(void* hh, double* c, double* T) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  std::memcpy(c, o.minco.c.data(), o.minco.c.size() * sizeof(double));
  std::memcpy(T, o.minco.T1.data(), o.minco.T1.size() * sizeof(double));
}

int orc_optimize(void* hh) { return ((OracleHandle*)hh)->opt.optimize() ? 1 : 0; }
// Test tooling: this restatement's solver logic (L-BFGS, line search, ALM loop) with its vector arithmetic in the device's
// order (epl = elements per lane, 12 covers every class) and the cost / gradient supplied by `fn` -- the device's
// evaluation hook.  A device solve must come out bit for bit (tests/test_gpu_parity.py).
int orc_optimize_device_order(void* hh, int epl, int nw, TrajOpt::ExternalEval fn, void* user) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  o.ext_eval = fn;
  o.ext_user = user;
  g_device_epl = epl;
  g_device_nw = (nw == 2 || nw == 4) ? nw : 1;
  const bool ok = o.optimize();
  g_device_epl = 0;
  g_device_nw = 1;
  o.ext_eval = nullptr;
  o.ext_user = nullptr;
  return ok ? 1 : 0;
}
// stage-2 ALM loop only, from the current x and (lambda, rho) (orc_set_x / orc_set_alm)
int orc_optimize_warm(void* hh) { return ((OracleHandle*)hh)->opt.optimize(true) ? 1 : 0; }
// optimize with a trace of f at every evaluation; returns number of evaluations recorded (<= cap)
int orc_optimize_trace(void* hh, double* trace, int cap, int* success) {
  TrajOpt& o = ((OracleHandle*)hh)->opt;
  o.trace.clear();
  o.trace_on = true;
  *success = o.optimize() ? 1 : 0;
  o.trace_on = false;
  int n = (int)std::min<size_t>(o.trace.size(), (size_t)cap);
  std::memcpy(trace, o.trace.data(), n * sizeof(double));
  return (int)o.trace.size();
}

void orc_get_stats(void* hh, int s[8]) {
  const SolveStats& st = ((OracleHandle*)hh)->opt.stats;
  s[0] = st.stage1_ret; s[1] = st.stage1_iters; s[2] = st.stage1_evals; s[3] = st.stage2_last_ret;
  s[4] = st.stage2_iters; s[5] = st.stage2_evals; s[6] = st.alm_outer; s[7] = st.sum_bound;
}
void orc_debug_counters(long long* out2) { out2[0] = g_dbg_ls_evals; out2[1] = g_dbg_ftest_fail; }
double orc_traj_cost(void* hh) { return ((OracleHandle*)hh)->opt.traj_cost; }
void orc_get_traj(void* hh, double* durations, double* coeffs, double* knots_xy) {
  ((OracleHandle*)hh)->opt.getTraj(durations, coeffs, knots_xy);
}
// feasibility gate on the trajectory currently held (printConstraintsSituations; *strict = checkFeasible)
int orc_check_feasible(void* hh, double* report38, int* strict) {
  bool st = false;
  const bool f = ((OracleHandle*)hh)->opt.checkSituations(report38, &st);
  if (strict) *strict = st ? 1 : 0;
  return f ? 1 : 0;
}
// car_seq of the trajectory currently held: returns the number of entries, fills up to cap rows of (x, y, theta, t)
int orc_car_seq(void* hh, double* seq, int cap) {
  const auto s = ((OracleHandle*)hh)->opt.carSeq();
  for (size_t k = 0; k < s.size() && (int)k < cap; k++)
    for (int q = 0; q < 4; q++) seq[4 * k + q] = s[k][q];
  return (int)s.size();
}
// MomaParam::getMeshPose of one state (11 x 7) and Planner::toMeshMsg of the trajectory currently held
void orc_mesh_pose(void* hh, const double* state10, double* parts77) {
  double mp[11][7];
  ((OracleHandle*)hh)->opt.robot.getMeshPose(state10, mp);
  std::memcpy(parts77, mp, sizeof(mp));
}
int orc_mesh_traj(void* hh, int res, int cap, double* parts, double* yaws, double* arcs) {
  std::vector<double> p, y, a;
  const int n = ((OracleHandle*)hh)->opt.meshTraj(res, p, y, a);
  for (int k = 0; k < n && k < cap; k++) {
    std::memcpy(parts + (size_t)k * 77, p.data() + (size_t)k * 77, 77 * sizeof(double));
    yaws[k] = y[k];
    arcs[k] = a[k];
  }
  return n;
}
// MomaTraj::getState(t) of the trajectory currently held (10 values)
void orc_traj_state(void* hh, double t, double* state10) {
  auto& o = ((OracleHandle*)hh)->opt;
  const auto seq = o.carSeq();
  o.trajState(seq, t, state10);
}

// ---------------------------------------------------------------------------------------------
// Batch solve with a thread pool: the CPU baseline leg (one trajectory per task, the analogue of
// the reference's thread-per-candidate, planner.cpp:921-925).
//   paths: ragged, sum(path_len) x 10; bvel/bacc: batch x 20 (10x2 col-major each)
//   out: success[b], cost[b], n_pieces[b], stats[b*8], seconds (wall)
//   optional per-trajectory results: durations (batch x maxN), coeffs (batch x maxN x 54), knots (batch x (maxN+1) x 2), xfinal (batch x maxn)
// ---------------------------------------------------------------------------------------------
double orc_optimize_batch(const double origin[3], double res, const int dims[3], const double min_b[3],
                          const double max_b[3], const double* esdf2d, const double* esdf3d, int batch,
                          const int* path_len, const double* paths, const double* bvel, const double* bacc,
                          int nthreads, int alm_max_outer, int* success, double* cost, int* n_pieces, int* stats,
                          int maxN, double* durations, double* coeffs, double* knots) {
  std::vector<size_t> offs(batch + 1, 0);
  for (int b = 0; b < batch; b++) offs[b + 1] = offs[b] + (size_t)path_len[b] * 10;
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    OracleHandle h;
    h.opt.map = &h.map;
    h.map.set(origin, res, dims, esdf2d, esdf3d);
    h.map.setBounds(min_b, max_b);
    if (alm_max_outer > 0) h.opt.prm.alm_max_outer = alm_max_outer;
    while (true) {
      int b = next.fetch_add(1);
      if (b >= batch) break;
      h.opt.setInitTraj(paths + offs[b], path_len[b], bvel + (size_t)b * 20, bacc + (size_t)b * 20);
      bool ok = h.opt.optimize();
      success[b] = ok ? 1 : 0;
      cost[b] = h.opt.traj_cost;
      n_pieces[b] = h.opt.piece_num;
      const SolveStats& st = h.opt.stats;
      int* s = stats + (size_t)b * 8;
      s[0] = st.stage1_ret; s[1] = st.stage1_iters; s[2] = st.stage1_evals; s[3] = st.stage2_last_ret;
      s[4] = st.stage2_iters; s[5] = st.stage2_evals; s[6] = st.alm_outer; s[7] = st.sum_bound;
      if (durations && h.opt.piece_num <= maxN)
        h.opt.getTraj(durations + (size_t)b * maxN, coeffs + (size_t)b * maxN * 54, knots + (size_t)b * (maxN + 1) * 2);
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < std::max(1, nthreads); t++) th.emplace_back(worker);
  for (auto& t : th) t.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Same, but every trajectory names its own map (benchmark_tables batch: one map per scenario).  One pool of
// nthreads workers over all trajectories.  Map descriptors are arrays of length n_maps.
// Stage 1 alone (optimizeTraj:359-374: the first L-BFGS run; the ALM loop is not entered) for a batch against per-candidate
// maps: converged decision vectors x (row b: 10 N_b - 8 entries of a row of `xstride`), stage-1 cost and counters.  Stage 1
// is short and not chaotic, so this is where a converged result can be compared value by value with the device's.
double orc_stage1_batch_maps(int n_maps, const double* origin, const double* res, const int* dims, const double* min_b,
                             const double* max_b, const double* const* esdf2d, const double* const* esdf3d, const int* map_id,
                             int batch, const int* path_len, const double* paths, int nthreads, int xstride, double* xout,
                             double* cost, int* n_pieces, int* stats /*[batch][3]: return code, iterations, evaluations*/) {
  std::vector<size_t> offs(batch + 1, 0);
  for (int b = 0; b < batch; b++) offs[b + 1] = offs[b] + (size_t)path_len[b] * 10;
  std::vector<double> zeros(20, 0.0);
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    OracleHandle h;
    h.opt.map = &h.map;
    h.opt.prm.alm_max_outer = 0;
    while (true) {
      int b = next.fetch_add(1);
      if (b >= batch) break;
      const int m = map_id[b];
      if (m < 0 || m >= n_maps) { n_pieces[b] = 0; continue; }
      h.map.set(origin + 3 * m, res[m], dims + 3 * m, esdf2d[m], esdf3d[m]);
      h.map.setBounds(min_b + 3 * m, max_b + 3 * m);
      h.opt.setInitTraj(paths + offs[b], path_len[b], zeros.data(), zeros.data());
      (void)h.opt.optimize();
      n_pieces[b] = h.opt.piece_num;
      cost[b] = h.opt.traj_cost;
      stats[3 * b] = h.opt.stats.stage1_ret; stats[3 * b + 1] = h.opt.stats.stage1_iters; stats[3 * b + 2] = h.opt.stats.stage1_evals;
      const int n = (int)h.opt.x.size();
      for (int k = 0; k < n && k < xstride; k++) xout[(size_t)b * xstride + k] = h.opt.x[k];
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < std::max(1, nthreads); t++) th.emplace_back(worker);
  for (auto& t : th) t.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

double orc_optimize_batch_maps(int n_maps, const double* origin /*[M][3]*/, const double* res /*[M]*/,
                               const int* dims /*[M][3]*/, const double* min_b, const double* max_b,
                               const double* const* esdf2d, const double* const* esdf3d, const int* map_id, int batch,
                               const int* path_len, const double* paths, int nthreads, int* success, double* cost,
                               int* n_pieces, int* stats, double* seconds_each, double* total_duration /* may be null */,
                               int* gate /* may be null: printConstraintsSituations of the returned trajectory */) {
  std::vector<size_t> offs(batch + 1, 0);
  for (int b = 0; b < batch; b++) offs[b + 1] = offs[b] + (size_t)path_len[b] * 10;
  std::vector<double> zeros(20, 0.0);
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    OracleHandle h;
    h.opt.map = &h.map;
    while (true) {
      int b = next.fetch_add(1);
      if (b >= batch) break;
      const int m = map_id[b];
      if (m < 0 || m >= n_maps) { success[b] = 0; continue; }
      h.map.set(origin + 3 * m, res[m], dims + 3 * m, esdf2d[m], esdf3d[m]);
      h.map.setBounds(min_b + 3 * m, max_b + 3 * m);
      auto t1 = std::chrono::steady_clock::now();
      h.opt.setInitTraj(paths + offs[b], path_len[b], zeros.data(), zeros.data());
      bool ok = h.opt.optimize();
      if (seconds_each) seconds_each[b] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
      success[b] = ok ? 1 : 0;
      cost[b] = h.opt.traj_cost;
      n_pieces[b] = h.opt.piece_num;
      const SolveStats& st = h.opt.stats;
      int* s = stats + (size_t)b * 8;
      s[0] = st.stage1_ret; s[1] = st.stage1_iters; s[2] = st.stage1_evals; s[3] = st.stage2_last_ret;
      s[4] = st.stage2_iters; s[5] = st.stage2_evals; s[6] = st.alm_outer; s[7] = st.sum_bound;
      if (total_duration || gate) {   // what the planner ranks and gates the candidate by (planner.cpp:878-880, 999-1010)
        const int N = h.opt.piece_num;
        std::vector<double> d(N), c((size_t)N * 54), k((size_t)(N + 1) * 2);
        h.opt.getTraj(d.data(), c.data(), k.data());
        double t = 0.0;
        for (int i = 0; i < N; i++) t += d[i];
        if (total_duration) total_duration[b] = t;
        if (gate) {
          double rep[38];
          bool strict = false;
          gate[b] = (std::isfinite(t) && t > 0.0 && t < 1.0e4 && h.opt.checkSituations(rep, &strict)) ? 1 : 0;
        }
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < std::max(1, nthreads); t++) th.emplace_back(worker);
  for (auto& t : th) t.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// The planner's cancellation of a planning call's thread group (planner.cpp:829-952) on the deterministic work clock:
// threads start together, every candidate advances one piece-evaluation per unit of time; `tau` of a call = the clock at
// which its first candidate finished with optimizeTraj true AND the gate passed (future_succ), and whoever is still
// running `budget` units later (the 100 ms of cv_all.wait_for) is interrupted at its next interruption point
// (moma_traj_opt.cpp:402, 887) and has no result.  clock[b] = (stage-1 + stage-2 evaluations) x pieces of candidate b run
// to its own end; accepted[b] = success && gate.  Writes interrupted[b]; returns the number of interrupted candidates.
int orc_group_cancel(int batch, const int* group_id, const long long* clock, const int* accepted, long long budget, int* interrupted) {
  int ng = 0;
  for (int b = 0; b < batch; b++) ng = std::max(ng, group_id[b] + 1);
  std::vector<long long> tau(ng, std::numeric_limits<long long>::max() / 4);
  for (int b = 0; b < batch; b++)
    if (group_id[b] >= 0 && accepted[b]) tau[group_id[b]] = std::min(tau[group_id[b]], clock[b]);
  int n = 0;
  for (int b = 0; b < batch; b++) {
    interrupted[b] = (group_id[b] >= 0 && clock[b] > tau[group_id[b]] + budget) ? 1 : 0;
    n += interrupted[b];
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// Unit-level hooks for the known-answer tests (SURVEY.md §8c list)
// ---------------------------------------------------------------------------------------------
// banded: A given dense (n x n row-major), bandwidth p=q=bw; solve A x = b (m columns, col-major) and A^T x = b
void orc_banded_solve(const double* dense, int n, int bw, double* b, int m, int adjoint) {
  BandedSystem A;
  A.create(n, bw, bw);
  for (int i = 0; i < n; i++)
    for (int j = std::max(0, i - bw); j <= std::min(n - 1, i + bw); j++) A(i, j) = dense[(size_t)i * n + j];
  A.factorizeLU();
  if (adjoint) A.solveAdj(b, m);
  else A.solve(b, m);
}
// MINCO: generate from (head, tail, inner, T); returns coefficients, jerk cost, jerk grads, and (optionally) the
// adjoint back-prop of a supplied (gdC, gdT)
void orc_minco(int N, const double* ew9, const double* head27, const double* tail27, const double* inPs, const double* T,
               double* c_out, double* jerk, double* gdC_jerk, double* gdT_jerk, const double* gdC_in, double* gdT_io,
               double* gdP_out, double* gdTail27) {
  MinJerkOpt9 m;
  m.reset(N, ew9);
  m.generate(head27, tail27, inPs, T);
  std::memcpy(c_out, m.c.data(), m.c.size() * sizeof(double));
  *jerk = m.getTrajJerkCost();
  std::vector<double> gc, gt;
  m.calJerkGradCT(gc, gt);
  std::memcpy(gdC_jerk, gc.data(), gc.size() * sizeof(double));
  std::memcpy(gdT_jerk, gt.data(), gt.size() * sizeof(double));
  if (gdC_in) {
    std::vector<double> gC(gdC_in, gdC_in + (size_t)6 * N * 9), gT(gdT_io, gdT_io + N), gP;
    m.calGradCTtoQT(gC, gT, gP, gdTail27);
    std::memcpy(gdT_io, gT.data(), N * sizeof(double));
    std::memcpy(gdP_out, gP.data(), gP.size() * sizeof(double));
  }
}
void orc_reparam(double v, double maxq, double out[6]) {
  out[0] = TrajOpt::expC2(v);
  out[1] = v > 0 ? TrajOpt::logC2(v) : 0.0;
  out[2] = TrajOpt::getTtoTauGrad(v);
  out[3] = TrajOpt::sigmoidC2(v, maxq);
  out[4] = (std::fabs(v) < maxq) ? TrajOpt::invSigmoidC2(v, maxq) : 0.0;
  out[5] = TrajOpt::getQtoVqGrad(v, maxq);
}
void orc_smooth_l1(double x, double mu, double out[2]) {
  TrajOpt t;
  t.prm.relu_mu = mu;
  t.smoothL1Penalty(x, out[0], out[1]);
}
int orc_colli_pts(const double* moma_pos10, double* out48) {
  Robot r;
  auto pts = r.getColliPts(moma_pos10);
  for (size_t i = 0; i < pts.size(); i++) {
    out48[i * 4 + 0] = pts[i].p.x; out48[i * 4 + 1] = pts[i].p.y; out48[i * 4 + 2] = pts[i].p.z; out48[i * 4 + 3] = pts[i].r;
  }
  return (int)pts.size();
}
void orc_colli_grads(const double* moma_pos10, const double* pos_grads36, double* out10) {
  Robot r;
  std::vector<V3> pg(12);
  for (int i = 0; i < 12; i++) pg[i] = V3{pos_grads36[i * 3], pos_grads36[i * 3 + 1], pos_grads36[i * 3 + 2]};
  r.getColliGrads(moma_pos10, pg, out10);
}
void orc_collision_matrix(int* out144) {
  Robot r;
  for (int i = 0; i < 12; i++)
    for (int j = 0; j < 12; j++) out144[i * 12 + j] = r.collision_matrix[i][j];
}
void orc_esdf_query(void* hh, int dim, const double* pos, double* dist, double* grad) {
  OracleHandle* h = (OracleHandle*)hh;
  if (dim == 2) h->map.getDisWithGradI2d(pos, *dist, grad);
  else h->map.getDisWithGradI3d(pos, *dist, grad);
}
// L-BFGS on the n-d Rosenbrock function / a diagonal quadratic (kind 0/1) with the stage-`stage` parameter set
int orc_lbfgs_test(int kind, int n, int stage, double* x, double* f, int* iters, int* evals) {
  Params p;
  LbfgsParam lp = stage == 1 ? p.s1_lbfgs : p.s2_lbfgs;
  std::vector<double> xx(x, x + n);
  LbfgsStats st;
  EvalFn fn;
  if (kind == 0)
    fn = [n](const std::vector<double>& v, std::vector<double>& g) {
      double fx = 0.0;
      g.assign(n, 0.0);
      for (int i = 0; i < n; i += 2) {
        double t1 = 1.0 - v[i], t2 = 10.0 * (v[i + 1] - v[i] * v[i]);
        g[i + 1] = 20.0 * t2;
        g[i] = -2.0 * (v[i] * g[i + 1] + t1);
        fx += t1 * t1 + t2 * t2;
      }
      return fx;
    };
  else
    fn = [n](const std::vector<double>& v, std::vector<double>& g) {
      double fx = 0.0;
      g.assign(n, 0.0);
      for (int i = 0; i < n; i++) { double w = 1.0 + i; fx += 0.5 * w * (v[i] - 1.0) * (v[i] - 1.0); g[i] = w * (v[i] - 1.0); }
      return fx;
    };
  int ret = lbfgs_optimize(xx, *f, fn, nullptr, lp, &st);
  std::memcpy(x, xx.data(), n * sizeof(double));
  *iters = st.iterations;
  *evals = st.evaluations;
  return ret;
}

}  // extern "C"
