// libtopay_hip.so — kernels + host side of the C-ABI declared in include/topay.h.
// gfx950 (MI355X) only; built by `hipcc --offload-arch=gfx950 -shared -fPIC`.  No torch types, no CPU fallback:
// every entry point fails with TOPAY_ERR_NO_DEVICE when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include <dlfcn.h>

#include "topay_solve.h"
#include "topay_feas.h"
#include "topay_edt.h"
#include "topay_front.h"
#include "topay_mcrrt.h"
#include "topay_jps.h"
#include "topay_yaml.h"

#include "topay_kernels.h"

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static void set_err(const std::string& s) { g_err = s; }
#define HIPCHK(call)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      set_err(std::string(#call) + ": " + hipGetErrorString(e_));                                \
      return TOPAY_ERR_NO_DEVICE;                                                                \
    }                                                                                            \
  } while (0)

// Tuning switches read from the environment exist only in builds with -DTOPAY_EXPERIMENTS (tools/ab_lib.sh); the product
// library reads TOPAY_PERSISTENT / TOPAY_STEAL (launch scheme, used by the profiling scripts and the parity tests),
// TOPAY_RCCL_LIB, and sets GPU_MAX_HW_QUEUES when it is loaded.
#ifdef TOPAY_EXPERIMENTS
static const char* exp_env(const char* name) { return getenv(name); }
#else
static const char* exp_env(const char*) { return nullptr; }
// A tuning switch in the environment of the product library would be silently ignored (an A/B script then measures the
// same build twice): say so, once per process, when the library is loaded.
__attribute__((constructor)) static void topay_warn_ignored_switches() {
  static const char* names[] = {"TOPAY_DISPATCH_GATE", "TOPAY_EDT_ENVELOPE", "TOPAY_FORCE_CLASS", "TOPAY_GATE_IN_SOLVE", "TOPAY_LDS_PAD",
                                "TOPAY_OCC2_GAIN", "TOPAY_OVERSUBSCRIBE", "TOPAY_POISON", "TOPAY_RESERVE_SLOTS", "TOPAY_SHARE_BIAS0",
                                "TOPAY_MW_C4", "TOPAY_MW_C5"};
  for (const char* n : names)
    if (getenv(n)) fprintf(stderr, "libtopay_hip: %s is set but ignored -- tuning switches exist only in -DTOPAY_EXPERIMENTS builds (tools/ab_lib.sh)\n", n);
}
#endif

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  topay_status ensure(size_t n) {
    if (n <= bytes) return TOPAY_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    {
      hipError_t e_ = hipMalloc(&p, n);
      if (e_ != hipSuccess) {
        p = nullptr;
        set_err("device allocation of " + std::to_string(n >> 20) + " MiB failed: " + hipGetErrorString(e_));
        return TOPAY_ERR_NO_DEVICE;
      }
    }
    bytes = n;
    // debugging aid (TOPAY_POISON=<byte>): fill every fresh allocation, so that a read of memory nobody has written
    // shows up on every run instead of only when the allocator hands out dirty pages
    static const int poison = [] { const char* e = exp_env("TOPAY_POISON"); return e ? atoi(e) : -1; }();
    if (poison >= 0) {
      HIPCHK(hipMemset(p, poison, n));
      HIPCHK(hipDeviceSynchronize());
    }
    return TOPAY_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename T> T* as() { return (T*)p; }
};

// A function-local device buffer: released on every way out (the early returns of the HIPCHK macro included).
struct ScopedDevBuf : DevBuf {
  ~ScopedDevBuf() { release(); }
};

// Launch buckets by number of pieces: upper bounds (inclusive) = one bucket per kernel template (rows per lane 1 / 2 / 3 / 4 / 6).
// Each bucket is one launch on its own stream so that they run concurrently.  All streams have the SAME priority:
// mixed priorities made the hardware preempt (context-save) the low-priority waves whenever high-priority work
// arrived, and twice in ~80 runs one low-priority launch was starved for tens of seconds.  HIP maps the streams of one
// priority onto a pool of GPU_MAX_HW_QUEUES (default 4) hardware queues shared by every stream of the process, and
// streams that share a queue serialise (tools/queue_probe.hip); the library asks for 24 queues at load time (below: three contexts in
// flight use 18, and another library's stream -- RCCL's -- that lands on the queue of a persistent solve launch waits a
// whole solve; 32 and more are time-sliced by the scheduler firmware)
// when the environment does not say otherwise.  With one wave per SIMD, four workgroups of the common classes (<= 21 /
// 27 / 36 / 54 KB per wave) share a CU's 160 KB; the two rare classes of long candidates (<= 70 / 104 KB) cost their CU a
// slot or two, which is why they are kept apart from each other.
static const int kBucketMaxN[TOPAY_NBUCKET] = {10, 15, 21, 32, 42, 64, TOPAY_MAX_N};
// Launch classes are finer than kernel templates where that saves LDS: the two-rows-per-lane kernel serves N <= 15 with
// 27 KB and N <= 21 with 36 KB per workgroup (most candidates of the benchmark have 11..15 pieces).  LDS is what
// limits how many workgroups a CU hosts beside a long candidate's: giving every class-1 workgroup the 38 KB of class 2
// cost 8 % of the throughput, taking 9 KB from most of the two-rows workgroups pays the other way.
static const int kBigFirst = 4;   // the classes of long candidates (N > 32) start here

// Kernel of a launch class: rows per thread and waves per trajectory select the template; the LDS is sized by the
// longest candidate actually in the class.  The classes of long candidates have a one-wave and a several-waves variant
// (TOPAY_MW_C4 / TOPAY_MW_C5 = waves per trajectory for N <= 42 / N <= 64; N <= 170 always evaluates on four waves).
typedef void (*solve_kernel_t)(DevBatch, const DevMap*, int);
typedef void (*eval_kernel_t)(DevBatch, const DevMap*, int, int, int);
struct ClassDef {
  int max_n, rmax, nw;   // rows per thread and waves of an EVALUATION of the class (= threads of its workgroups / 64)
  solve_kernel_t solve;
  eval_kernel_t eval;
  int occ = 2;   // waves per SIMD the kernel is built for (512 / occ registers per lane; every kernel: 256, no AGPRs)
  solve_kernel_t lat = nullptr;   // helper-wave kernel of a one-wave class (topay_set_latency_mode): 4 waves per workgroup
  // The waves the SOLVER's vectors are divided over and its rows per lane (elements per thread / 2): what the bits of a solve
  // depend on (topay_class_of).  0 = as the evaluation.  The long classes run a one-wave solver on wave 0 of a four-wave
  // workgroup whose other waves join the evaluations only (helper waves, topay_solve.h): solver_nw 1, helpers true.
  int solver_nw = 0, solver_rmax = 0;
  bool helpers = false;           // `solve` is a helper-wave kernel: its LDS carries the command block
  int snw() const { return solver_nw ? solver_nw : nw; }
  int srmax() const { return solver_rmax ? solver_rmax : rmax; }
};
static const int kLatWaves = 4;
static const ClassDef* class_table() {
  static const ClassDef* tab = [] {
    static ClassDef t[TOPAY_NBUCKET] = {
        {10, 1, 1, k_solve1, k_eval1, 2, k_lat1}, {15, 2, 1, k_solve2, k_eval2, 2, k_lat2}, {21, 2, 1, k_solve2, k_eval2, 2, k_lat2},
        {32, 3, 1, k_solve3, k_eval3, 2, k_lat3},
        {42, 2, 4, k_long5, k_eval2w4, 2, nullptr, 1, 5, true}, {64, 2, 4, k_long5, k_eval2w4, 2, nullptr, 1, 5, true},
        {TOPAY_MAX_N, 4, 4, k_long14, k_eval4w4, 2, nullptr, 1, 14, true}};
    // Four waves per trajectory for N = 33..64 since round 4.  Round 3 (one wave per SIMD), one / two / four waves for both
    // classes: 9.8-10.1k / 10.1k / 9.1k trajectories/s, strictly serial steps 1.15 / 1.00 / 1.03 s -- four waves halve a long
    // candidate's solve but held four SIMDs for it.  With two waves per SIMD a wave holds half a SIMD, the common classes
    // got faster and the long candidates set the length of a batch again (their launch was the longest of a serial step,
    // 1.06-1.21 s): two / four waves for N = 43..64 now give 11.5-11.6k / 11.8-11.9k and serial steps of 1.07 / 0.95 s, four
    // for N = 33..42 as well 0.94 s (tools/experiments/r4_mw.sh).
    // Round 5: the SOLVER of the long classes runs on one wave (k_long5 / k_long14: 10 / 28 vector elements per lane) and only
    // the evaluations use the four waves -- every reduction of the four-wave solver (two per history pair of the two-loop
    // recursion) was a workgroup reduction through LDS and a barrier (k_solve2w4: VALU active 18.7 % of its wave cycles).
#ifdef TOPAY_EXPERIMENTS
    auto env_nw = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    const int w4 = env_nw("TOPAY_MW_C4", 4), w5 = env_nw("TOPAY_MW_C5", 4);
    if (w4 == 1) t[4] = {42, 4, 1, k_solve4, k_eval4, 2};
    else if (w4 == 2) t[4] = {42, 2, 2, k_solve2w2, k_eval2w2, 2};
    if (w5 == 1) t[5] = {64, 6, 1, k_solve6, k_eval6, 2};
    else if (w5 == 2) t[5] = {64, 3, 2, k_solve3w2, k_eval3w2, 2};
    // TOPAY_LONG_SOLVER=4: the four-wave solver of round 4 for the long classes (A/B)
    if (const char* e = getenv("TOPAY_LONG_SOLVER")) {
      if (atoi(e) == 4) {
        if (w4 == 4) t[4] = {42, 2, 4, k_solve2w4, k_eval2w4, 2};
        if (w5 == 4) t[5] = {64, 2, 4, k_solve2w4, k_eval2w4, 2};
        t[6] = {128, 3, 4, k_solve3w4, k_eval3w4, 2};   // (the four-wave solver holds 128 pieces: longer candidates are not launched in this A/B mode)
      }
    }
#endif
    return t;
  }();
  return tab;
}
static const int kLdsDoublesPerCU = 160 * 1024 / 8;
static size_t class_lds_bytes(const ClassDef& cd, int nm) {
  int d = lds_doubles_mw(nm, cd.nw);
  // experiments build: TOPAY_LDS_PAD="<max_n of a class>:<doubles>[,...]" asks for more LDS than the class needs (what a
  // workgroup less per compute unit costs)
  if (const char* e = exp_env("TOPAY_LDS_PAD")) {
    for (const char* q = e; q && *q;) {
      int mn = 0, add = 0;
      if (sscanf(q, "%d:%d", &mn, &add) == 2 && mn == cd.max_n) d += add;
      q = strchr(q, ',');
      if (q) q++;
    }
  }
  // + past-cost ring [8] + the solver state parked across an evaluation [40] (+ the command block of a helper-wave kernel)
  return (size_t)(d + 8 + 40 + (cd.helpers ? TOPAY_CMD_DOUBLES : 0)) * sizeof(double);
}

// Runs when the library is loaded: effective if the HIP runtime has not been initialised yet in this process
// (the runtime reads the variable once, at its first call).  A caller that initialises HIP first should export
// GPU_MAX_HW_QUEUES=24 itself.
__attribute__((constructor)) static void topay_request_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "24", 0); }
// Dispatch gate (topay_optimize_async): the context whose solve was issued last in this process.
struct topay_ctx;
static std::mutex g_issue_mutex;
// Every live context of the process (topay_create / topay_destroy), for push_params.
static std::mutex g_registry_mutex;
static std::vector<topay_ctx*> g_contexts;

static topay_ctx* g_last_issued = nullptr;

struct topay_ctx {
  int device = 0;
  topay_params_t hp;
  DevParams dp;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // maps
  std::vector<DevMap> hmaps = std::vector<DevMap>(TOPAY_MAX_MAPS);
  std::vector<DevBuf> map2d = std::vector<DevBuf>(TOPAY_MAX_MAPS), map3d = std::vector<DevBuf>(TOPAY_MAX_MAPS);
  std::vector<DevBuf> map2d_inf = std::vector<DevBuf>(TOPAY_MAX_MAPS), map2d_crit = std::vector<DevBuf>(TOPAY_MAX_MAPS);
  // Maps built on the device as a batch live in one arena per build call (the construction writes the fields where
  // they stay; the slots' descriptors point into it), kept until the context is destroyed.
  struct MapArena { DevBuf buf; int first = 0, n = 0; };
  std::vector<MapArena> map_arenas;
  DevBuf dmaps;
  std::vector<char> have_map = std::vector<char>(TOPAY_MAX_MAPS, 0);
  // topay_share_maps: slot m of this context refers to the fields of map_owner[m] (null: its own); map_sharers = the
  // contexts that refer to slots of this one.  When the owner refills or frees a slot, the sharers' slots are invalidated
  // (have_map 0 -> TOPAY_ERR_NO_MAP) after their pending solves have finished: no context keeps a dangling descriptor.
  std::vector<topay_ctx*> map_owner = std::vector<topay_ctx*>(TOPAY_MAX_MAPS, nullptr);
  std::vector<topay_ctx*> map_sharers;
  std::vector<int> h_map_id;   // map slot of every candidate of the resident batch
  std::vector<int> h_path_len; // init-path states of every candidate (launch order inside a class)
  // batch
  int B = 0, Nmax = 0, total_states = 0, Pmax = 0;
  // pieces / decision-vector elements of the candidates before b (packed per-candidate blocks, DevBatch::poff / noff)
  std::vector<long long> h_poff, h_noff;
  DevBuf poff, noff;
  size_t workspace_bytes = 0;   // device memory of the resident batch (topay_workspace_bytes)
  // per-trajectory N (0 = not representable, skipped); launch buckets by N (LDS is sized per bucket)
  std::vector<int> hN;
  static constexpr int NBUCKET = TOPAY_NBUCKET;
  std::vector<int> cls[NBUCKET];
  hipStream_t bstream[NBUCKET] = {nullptr};
  hipEvent_t bevent[NBUCKET] = {nullptr};
  hipEvent_t bstart = nullptr;
  bool pending = false;  // a topay_optimize_async has been issued and not yet waited for
  int* h_started = nullptr;  // pinned host counter the solve kernels bump once per candidate (dispatch gate)
  int n_launched = 0;        // candidates the pending solve launched
  int n_gate = 0;            // ... of which the dispatch gate waits for (the classes of up to 32 pieces)
  bool gate = true;
  bool gate_in_solve = true;   // feasibility gate by the solving wave (TOPAY_GATE_IN_SOLVE=0: the separate kernel only)
  int latency_mode = 0;        // topay_set_latency_mode: 0 never, 1 batches of at most one candidate per SIMD, 2 always
  bool gate_done = false;      // the resident flags / report are those of the last solve
  // cancellation: planning call of every candidate, the window after a call's first feasible success (piece-evaluations)
  std::vector<int> h_group;
  int n_groups = 0, cancel_budget = 0;
  DevBuf group_id, group_tau, interrupted;
  int* h_cancel = nullptr;     // pinned: topay_cancel
  // the one exchange of the multi-GPU path: all-gather of per-scenario records over RCCL (topay_comm_init)
  void* comm = nullptr;        // ncclComm_t
  int comm_world = 0, comm_rank = 0;
  hipStream_t comm_stream = nullptr;
  DevBuf comm_send, comm_recv;
  int gate_timeouts = 0;     // times the dispatch gate gave up waiting (topay_gate_timeouts)
  bool persistent = true;    // solve launches: one workgroup per SIMD slot pulling candidates from a queue
  bool steal = true;         // ... and draining the smaller classes' queues once its own is empty (TOPAY_STEAL=0: profiling)
  int simd_slots = 1024;
  double occ2_gain = 1.7;    // work per SIMD-second of the two-waves-per-SIMD classes relative to one wave per SIMD (sizes the launches only)
  DevBuf qnext;
  DevBuf mc_i, mc_d, mc_k, mc_rs, mc_in;   // node tables, Reeds-Shepp words and inputs of the last topay_mcrrt_plan
  int mc_n = 0, mc_node_cap = 0;
  DevBuf paths, path_off, path_len, bvel, bacc, scratch;
  DevBuf N, s1_past, map_id, head, tail, start_xy, goal_xy, init_xy, x0;
  DevBuf x, work, hist_s, hist_y, hist_ys, hist_alpha, lu;
  DevBuf success, cost, stats, xyerr, coef, T, knots, alm, fout, order, trace, elapsed, startus, hwid, sbuf, mstash, feas_cseq, feas_tk, feas_report, feas_flags, edt_occ, edt_tmp1, edt_tmp2, edt_v, edt_z, edt_out2, edt_out3, pb_io;
  float last_edt_ms = 0.f;
  int trace_cap = 0;
  DevBatch db;
  bool have_traj = false, solved = false;
  double last_ms = 0.0;
  int last_launches = 0, last_helper_launches = 0;
};

static void make_dev_params(const topay_params_t& p, DevParams& d) {
  memset(&d, 0, sizeof(d));
  d.relu_mu = p.relu_mu;
  {
    const double pe = p.relu_mu;
    d.sl_half = 0.5 * pe;
    d.sl_f3c = 1.0 / (pe * pe);
    d.sl_f4c = -0.5 * d.sl_f3c / pe;
    d.sl_d2c = 3.0 * d.sl_f3c;
    d.sl_d3c = 4.0 * d.sl_f4c;
  }
  for (int i = 0; i < 9; i++) d.energy_weights[i] = p.energy_weights[i];
  d.s1_time_weight = p.s1_time_weight; d.s1_moment_weight = p.s1_moment_weight; d.s1_acc_weight = p.s1_acc_weight;
  d.s1_domega_weight = p.s1_domega_weight; d.s1_path_pos_weight = p.s1_path_pos_weight;
  d.s2_time_weight = p.s2_time_weight; d.s2_moment_weight = p.s2_moment_weight; d.s2_acc_weight = p.s2_acc_weight;
  d.s2_domega_weight = p.s2_domega_weight; d.s2_collision_weight = p.s2_collision_weight;
  d.s2_mani_colli_weight = p.s2_mani_colli_weight; d.s2_self_colli_weight = p.s2_self_colli_weight;
  d.s2_mani_pos_weight = p.s2_mani_pos_weight; d.s2_mani_vel_weight = p.s2_mani_vel_weight;
  d.s2_mani_acc_weight = p.s2_mani_acc_weight; d.s2_mean_time_weight = p.s2_mean_time_weight;
  for (int i = 0; i < 2; i++) {
    d.alm_init_lambda[i] = p.alm_init_lambda[i]; d.alm_init_rho[i] = p.alm_init_rho[i];
    d.alm_rho_max[i] = p.alm_rho_max[i]; d.alm_gamma[i] = p.alm_gamma[i];
  }
  d.alm_tolerance = p.alm_tolerance;
  d.alm_max_outer = p.alm_max_outer;
  d.alm_work_budget = p.alm_work_budget;
  d.min_piece_num = p.min_piece_num;
  d.sample_interval = p.sample_interval;
  d.s1_normal_past = p.s1_normal_past; d.s1_shot_path_past = p.s1_shot_path_past;
  d.s1_shot_path_horizon = p.s1_shot_path_horizon;
  auto cp = [](const topay_lbfgs_params_t& a, DevLbfgs& b) {
    b.mem_size = a.mem_size; b.past = a.past; b.max_iterations = a.max_iterations; b.max_linesearch = a.max_linesearch;
    b.g_epsilon = a.g_epsilon; b.delta = a.delta; b.min_step = a.min_step; b.max_step = a.max_step;
    b.f_dec_coeff = a.f_dec_coeff; b.s_curv_coeff = a.s_curv_coeff; b.cautious_factor = a.cautious_factor;
    b.machine_prec = a.machine_prec;
  };
  cp(p.s1_lbfgs, d.s1_lbfgs);
  cp(p.s2_lbfgs, d.s2_lbfgs);
  d.chassis_height = p.chassis_height; d.chassis_colli_radius = p.chassis_colli_radius;
  d.max_v = p.max_v; d.max_a = p.max_a; d.max_w = p.max_w; d.max_dw = p.max_dw;
  for (int i = 0; i < 8; i++) d.colli_length[i] = p.colli_length[i];
  int s = 0;
  for (int i = 0; i < 16 && s < TOPAY_NSPH; i++)
    if (p.colli_points[i] != 0.0) {  // moma_param.h:217-218
      d.sph_off[s] = p.colli_points[i];
      d.sph_r[s] = p.colli_point_radius[i];
      s++;
    }
  for (int i = 0; i < 7; i++) {
    d.joint_pos_limit_max[i] = p.joint_pos_limit_max[i];
    d.joint_vel_limit[i] = p.joint_vel_limit[i];
    d.joint_acc_limit[i] = p.joint_acc_limit[i];
  }
  for (int i = 0; i < 9; i++) d.relR[i] = p.relative_R[i];
  for (int i = 0; i < 3; i++) d.relT[i] = p.relative_t[i];
  // derived constants (DevParams): the expressions of the evaluation, operation by operation (this file is compiled with
  // -ffp-contract=off like the device code, so the host's products and sums are the device's)
  for (int a = 0; a < TOPAY_NSPH; a++)
    for (int b = 0; b < TOPAY_NSPH; b++) {
      const double rr = d.sph_r[a] + d.sph_r[b];
      d.pair_rr2[a * TOPAY_NSPH + b] = rr * rr;
    }
  for (int k = 0; k < TOPAY_NSPH; k++) {
    d.sph_viol[k] = d.sph_r[k] * 10.0 * 1.1;
    d.sph_top[k] = d.chassis_height + d.relT[2] + d.sph_r[k];
  }
  d.p0z = d.chassis_height + d.relT[2];
  d.max_vw = d.max_v * d.max_w;
  d.max_a2 = d.max_a * d.max_a;
  d.max_dw2 = d.max_dw * d.max_dw;
  d.chassis_r105 = d.chassis_colli_radius * 1.05;
  for (int i = 0; i < 7; i++) {
    d.joint_vel_limit2[i] = d.joint_vel_limit[i] * d.joint_vel_limit[i];
    d.joint_acc_limit2[i] = d.joint_acc_limit[i] * d.joint_acc_limit[i];
  }
}

static int sphere_layout_ok(const topay_params_t& p) {
  // the kernels hard-wire MomaParam's sphere-per-link layout {2,1,2,1,2,1,2,1}
  const int want[8] = {2, 1, 2, 1, 2, 1, 2, 1};
  for (int i = 0; i < 8; i++) {
    int c = (p.colli_points[2 * i] != 0.0) + (p.colli_points[2 * i + 1] != 0.0);
    if (c != want[i]) return 0;
    if (want[i] == 1 && p.colli_points[2 * i] != 0.0) return 0;
  }
  return 1;
}

// Every copy goes through the context's own (non-blocking) stream: null-stream operations would wait for the solves of
// every other context of the process (and they for it), which serialises batches that are meant to overlap.
static hipError_t memcpy_sync(topay_ctx* c, void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, n, kind, c->stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(c->stream);
}

extern "C" {

const char* topay_last_error(void) { return g_err.c_str(); }

topay_status topay_default_params(topay_params_t* p) {
  if (!p) return TOPAY_ERR_INVALID_ARG;
  memset(p, 0, sizeof(*p));
  // src/planner/params/optimizer.yaml
  p->int_K = 12; p->min_piece_num = 3; p->relu_mu = 1.0e-3; p->sample_interval = 1.5;
  const double ew[9] = {0.33, 1, 1, 1, 1, 1, 1, 1, 1};
  for (int i = 0; i < 9; i++) p->energy_weights[i] = ew[i];
  p->s1_time_weight = 20.0; p->s1_moment_weight = 1000.0; p->s1_acc_weight = 1000.0; p->s1_domega_weight = 1000.0;
  p->s1_path_pos_weight = 200000.0; p->s1_normal_past = 2; p->s1_shot_path_past = 8; p->s1_shot_path_horizon = 0.5;
  auto lb = [](topay_lbfgs_params_t& l) {  // lbfgs.hpp:15-129 defaults
    l.mem_size = 8; l.g_epsilon = 1.0e-5; l.past = 3; l.delta = 1.0e-6; l.max_iterations = 0; l.max_linesearch = 64;
    l.min_step = 1.0e-20; l.max_step = 1.0e+20; l.f_dec_coeff = 1.0e-4; l.s_curv_coeff = 0.9; l.cautious_factor = 1.0e-6;
    l.machine_prec = 1.0e-16;
  };
  lb(p->s1_lbfgs); lb(p->s2_lbfgs);
  p->s1_lbfgs.mem_size = 256; p->s1_lbfgs.g_epsilon = 0.0; p->s1_lbfgs.min_step = 0.0; p->s1_lbfgs.delta = 1.0e-2;
  p->s1_lbfgs.max_iterations = 8000; p->s1_lbfgs.past = 2;
  p->s2_lbfgs.mem_size = 256; p->s2_lbfgs.past = 3; p->s2_lbfgs.g_epsilon = 0.0; p->s2_lbfgs.min_step = 1.0e-32;
  p->s2_lbfgs.delta = 1.0e-4; p->s2_lbfgs.max_iterations = 8000;
  p->s2_time_weight = 50.0; p->s2_moment_weight = 300.0; p->s2_acc_weight = 3000.0; p->s2_domega_weight = 3000.0;
  p->s2_collision_weight = 500000.0; p->s2_mani_colli_weight = 500000.0; p->s2_self_colli_weight = 500000.0;
  p->s2_mani_pos_weight = 500.0; p->s2_mani_vel_weight = 500.0; p->s2_mani_acc_weight = 500.0;
  p->s2_mean_time_weight = 5000.0;
  for (int i = 0; i < 2; i++) {
    p->alm_init_lambda[i] = 0.0; p->alm_init_rho[i] = 1.0e4; p->alm_rho_max[i] = 1.0e10; p->alm_gamma[i] = 9.0;
  }
  p->alm_tolerance = 0.01;
  p->alm_max_outer = 30;
  p->alm_work_budget = 24000;
  // src/simulator/fake_moma/include/fake_moma/moma_param.h:36-126
  p->chassis_height = 0.155; p->chassis_colli_radius = 0.4;
  p->max_v = 1.0; p->max_a = 0.8; p->max_w = 1.25; p->max_dw = 1.0;
  const double cl[8] = {0.139, 0.1015, 0.1525, 0.1035, 0.1285, 0.0815, 0.144, 0.05};
  const double cp[16] = {0.139 - 0.09, 0.139, 0.0, 0.1015, 0.1525 - 0.08, 0.1525, 0.0, 0.1035,
                         0.1285 - 0.07, 0.1285, 0.0, 0.0815, 0.144 - 0.07, 0.144, 0.0, 0.1};
  const double cr[16] = {0.06, 0.06, 0.0, 0.08, 0.04, 0.04, 0.0, 0.07, 0.035, 0.035, 0.0, 0.06, 0.035, 0.035, 0.0, 0.08};
  for (int i = 0; i < 8; i++) p->colli_length[i] = cl[i];
  for (int i = 0; i < 16; i++) {
    p->colli_points[i] = cp[i];
    p->colli_point_radius[i] = (cr[i] > 1e-4 && cr[i] < 0.055) ? 0.055 : cr[i];  // moma_param.h:110-112
  }
  const double qm[7] = {3.1, 2.26, 3.1, 2.355, 3.1, 2.23, 6.28};
  for (int i = 0; i < 7; i++) { p->joint_pos_limit_max[i] = qm[i]; p->joint_vel_limit[i] = 2.35; p->joint_acc_limit[i] = 6.28; }
  const double rr[9] = {0.7071068, 0.7071068, 0.0, -0.7071068, 0.7071068, 0.0, 0.0, 0.0, 1.0};
  for (int i = 0; i < 9; i++) p->relative_R[i] = rr[i];
  p->relative_t[0] = 0.0; p->relative_t[1] = 0.115; p->relative_t[2] = 0.016;
  return TOPAY_OK;
}

static topay_status validate_params(const topay_params_t* params) {
  if (params->int_K != TOPAY_K) { set_err("int_K must be 12 in this build"); return TOPAY_ERR_UNSUPPORTED; }
  if (!sphere_layout_ok(*params)) { set_err("unsupported collision sphere layout"); return TOPAY_ERR_UNSUPPORTED; }
  if (params->s1_lbfgs.mem_size <= 0 || params->s2_lbfgs.mem_size <= 0 || params->s1_lbfgs.mem_size > 256 ||
      params->s2_lbfgs.mem_size > 256 || params->s1_lbfgs.past > 8 ||
      params->s2_lbfgs.past > 8 || params->s1_shot_path_past > 8 || params->s1_normal_past > 8) {
    set_err("lbfgs mem_size must be in 1..256 (the reference uses 256) and past <= 8");
    return TOPAY_ERR_INVALID_ARG;
  }
  return TOPAY_OK;
}

topay_status topay_set_params(topay_ctx* c, const topay_params_t* params) {
  if (!c || !params) return TOPAY_ERR_INVALID_ARG;
  topay_status vs = validate_params(params);
  if (vs != TOPAY_OK) return vs;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }
  // what the resident batch was laid out with: number of pieces (init step) and history depth (workspace)
  const bool relayout = params->sample_interval != c->hp.sample_interval || params->min_piece_num != c->hp.min_piece_num ||
                        std::max(params->s1_lbfgs.mem_size, params->s2_lbfgs.mem_size) != std::max(c->hp.s1_lbfgs.mem_size, c->hp.s2_lbfgs.mem_size) ||
                        params->max_v != c->hp.max_v || params->max_a != c->hp.max_a || params->max_w != c->hp.max_w || params->max_dw != c->hp.max_dw;
  c->hp = *params;
  make_dev_params(*params, c->dp);
  if (relayout) { c->have_traj = false; c->solved = false; }
  return TOPAY_OK;
}

static topay_status create_device_state(topay_ctx* c, int device);
topay_status topay_create(const topay_params_t* params, int device, topay_ctx** out) {
  if (!params || !out) return TOPAY_ERR_INVALID_ARG;
  { topay_status vs = validate_params(params); if (vs != TOPAY_OK) return vs; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err("no HIP device available: the MI355X HIP path is required (there is no CPU fallback)");
    return TOPAY_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { set_err("device index out of range"); return TOPAY_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(device));
  topay_ctx* c = new topay_ctx();
  c->device = device;
  c->hp = *params;
  make_dev_params(*params, c->dp);
  // a failure below releases what has been created so far (topay_destroy copes with a partly built context)
  const topay_status st = create_device_state(c, device);
  if (st != TOPAY_OK) { topay_destroy(c); return st; }
  {
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    g_contexts.push_back(c);
  }
  *out = c;
  return TOPAY_OK;
}

static topay_status create_device_state(topay_ctx* c, int device) {
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&c->ev0));
  HIPCHK(hipEventCreate(&c->ev1));
  HIPCHK(hipEventCreate(&c->bstart));
  // One stream per launch class (the last class runs on the main stream), all non-blocking and never the null stream:
  // two contexts then use ten of the sixteen hardware queues the library asks for, and no operation of one context
  // waits for another context's solve.
  for (int k = 0; k < topay_ctx::NBUCKET; k++) {
    // (the two longest classes start on the main stream: streams are hardware queues, and 3 contexts x 7 streams beside
    // torch's and RCCL's exceed the 24 the library asks for -- a gather that shares a queue with a persistent solve
    // launch waits a whole solve, 9.1k instead of 10.0k trajectories/s through the RCCL path.  The N <= 64 class gets a
    // stream of its own the first time a batch also holds candidates of more than 64 pieces, launch_classes.)
    if (k >= topay_ctx::NBUCKET - 2) c->bstream[k] = c->stream;
    else HIPCHK(hipStreamCreateWithFlags(&c->bstream[k], hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&c->bevent[k]));
  }
  {
    { const char* se = getenv("TOPAY_STEAL"); c->steal = !(se && se[0] == '0'); }
    const char* pe = getenv("TOPAY_PERSISTENT");
    c->persistent = !(pe && pe[0] == '0');
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->simd_slots = 4 * prop.multiProcessorCount;   // four SIMDs per CU; a class built for two waves per SIMD launches two workgroups per slot
    if (const char* og = exp_env("TOPAY_OCC2_GAIN")) c->occ2_gain = std::max(0.5, atof(og));
    // A few slots are left to everything that is not a solve: the init kernel, the feasibility gate and the result
    // gather of the OTHER batches in flight, the runtime's copy kernels, a collective.  Resident solver waves own their
    // SIMD's whole register file, so on a device they fill completely such a kernel waits until workgroups exit.
    // TOPAY_RESERVE_SLOTS=<n> keeps n slots free (default 0: use every slot).
    {
      const char* rs = exp_env("TOPAY_RESERVE_SLOTS");
      const int reserve = rs ? atoi(rs) : 0;   // (measured: no gain from a standing reserve with two batches in flight)
      if (reserve > 0 && reserve < c->simd_slots / 2) c->simd_slots -= reserve;
    }
  }
  {
    const char* g = exp_env("TOPAY_DISPATCH_GATE");
    c->gate = !(g && g[0] == '0');
    void* hp = nullptr;
    HIPCHK(hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent));
    c->h_started = (int*)hp;
    c->h_started[0] = 0;
    c->h_cancel = c->h_started + 8;   // same pinned block: topay_cancel's flag
    c->h_cancel[0] = 0;
    { const char* ge = exp_env("TOPAY_GATE_IN_SOLVE"); c->gate_in_solve = !(ge && ge[0] == '0'); }
  }
  if (c->dmaps.ensure(sizeof(DevMap) * TOPAY_MAX_MAPS) != TOPAY_OK) return TOPAY_ERR_NO_DEVICE;
  memset(c->hmaps.data(), 0, sizeof(DevMap) * TOPAY_MAX_MAPS);
  return TOPAY_OK;
}

// The owner is about to refill (or free) slots [first, first + n): every context that shares one of them finishes its
// pending solve and loses the slot.
static void invalidate_sharers(topay_ctx* owner, int first, int n) {
  std::vector<topay_ctx*> sharers;
  {
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    sharers = owner->map_sharers;
  }
  for (topay_ctx* s : sharers) {
    bool hit = false;
    for (int m = first; m < first + n; m++) hit = hit || s->map_owner[m] == owner;
    if (!hit) continue;
    if (s->pending) (void)topay_synchronize(s);
    for (int m = first; m < first + n; m++)
      if (s->map_owner[m] == owner) {
        s->map_owner[m] = nullptr;
        s->have_map[m] = 0;
        memset(&s->hmaps[m], 0, sizeof(DevMap));
        // a resident batch that uses the slot cannot be solved, evaluated or gated any more: it has to be set again
        if (s->have_traj && std::find(s->h_map_id.begin(), s->h_map_id.end(), m) != s->h_map_id.end()) { s->have_traj = false; s->solved = false; }
      }
    bool any = false;
    for (int m = 0; m < TOPAY_MAX_MAPS; m++) any = any || s->map_owner[m] == owner;
    if (!any) {
      std::lock_guard<std::mutex> lk(g_registry_mutex);
      owner->map_sharers.erase(std::remove(owner->map_sharers.begin(), owner->map_sharers.end(), s), owner->map_sharers.end());
    }
  }
}
// slots [first, first + n) of c stop referring to another context's fields (c fills them itself, or goes away)
static void drop_shared_slots(topay_ctx* c, int first, int n) {
  std::lock_guard<std::mutex> lk(g_registry_mutex);
  for (int m = first; m < first + n; m++) {
    topay_ctx* o = c->map_owner[m];
    if (!o) continue;
    c->map_owner[m] = nullptr;
    bool any = false;
    for (int q = 0; q < TOPAY_MAX_MAPS; q++) any = any || c->map_owner[q] == o;
    if (!any) o->map_sharers.erase(std::remove(o->map_sharers.begin(), o->map_sharers.end(), c), o->map_sharers.end());
  }
}

void topay_destroy(topay_ctx* c) {
  if (!c) return;
  if (c->pending) (void)topay_synchronize(c);
  invalidate_sharers(c, 0, TOPAY_MAX_MAPS);
  drop_shared_slots(c, 0, TOPAY_MAX_MAPS);
  {
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    g_contexts.erase(std::remove(g_contexts.begin(), g_contexts.end(), c), g_contexts.end());
  }
  (void)hipSetDevice(c->device);
  DevBuf* bufs[] = {&c->dmaps, &c->paths, &c->path_off, &c->path_len, &c->bvel, &c->bacc, &c->scratch, &c->N, &c->s1_past,
                    &c->map_id, &c->head, &c->tail, &c->start_xy, &c->goal_xy, &c->init_xy, &c->x0, &c->x, &c->work,
                    &c->hist_s, &c->hist_y, &c->hist_ys, &c->hist_alpha, &c->lu, &c->poff, &c->noff, &c->group_id, &c->group_tau, &c->interrupted, &c->success, &c->cost, &c->stats,
                    &c->xyerr, &c->coef, &c->T, &c->knots, &c->alm, &c->fout, &c->order, &c->trace, &c->elapsed, &c->startus, &c->hwid, &c->sbuf, &c->mstash, &c->feas_cseq, &c->feas_tk, &c->feas_report, &c->feas_flags, &c->edt_occ, &c->edt_tmp1,
                    &c->edt_tmp2, &c->edt_v, &c->edt_z, &c->edt_out2, &c->edt_out3, &c->pb_io, &c->qnext, &c->mc_i, &c->mc_d, &c->mc_k, &c->mc_rs, &c->mc_in};
  for (DevBuf* b : bufs) b->release();
  for (int i = 0; i < TOPAY_MAX_MAPS; i++) { c->map2d[i].release(); c->map3d[i].release(); c->map2d_inf[i].release(); c->map2d_crit[i].release(); }
  for (auto& a : c->map_arenas) a.buf.release();
  for (int k = 0; k < topay_ctx::NBUCKET; k++) {
    if (c->bevent[k]) (void)hipEventDestroy(c->bevent[k]);
    if (c->bstream[k] && c->bstream[k] != c->stream) (void)hipStreamDestroy(c->bstream[k]);
  }
  {
    std::lock_guard<std::mutex> lk(g_issue_mutex);
    if (g_last_issued == c) g_last_issued = nullptr;
  }
  (void)topay_comm_destroy(c);
  c->comm_send.release(); c->comm_recv.release();
  if (c->h_started) (void)hipHostFree(c->h_started);
  if (c->bstart) (void)hipEventDestroy(c->bstart);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

topay_status topay_set_map(topay_ctx* c, int map_id, const topay_map_desc_t* desc, const double* esdf2d, const double* esdf3d) {
  if (!c || !desc || !esdf2d || !esdf3d || map_id < 0 || map_id >= TOPAY_MAX_MAPS) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }   // inputs of a solve in flight stay untouched
  const size_t n2 = (size_t)desc->dims[0] * desc->dims[1], n3 = n2 * desc->dims[2];
  if (n2 == 0 || n3 == 0) return TOPAY_ERR_INVALID_ARG;
  if (n3 >= (1ull << 32)) { set_err("map of 2^32 cells or more (the lookups index a field with 32 bits)"); return TOPAY_ERR_UNSUPPORTED; }
  if (desc->dims[2] < 2) { set_err("3-D field with a single layer (the lookups fetch z-neighbours in pairs)"); return TOPAY_ERR_UNSUPPORTED; }
  invalidate_sharers(c, map_id, 1);
  drop_shared_slots(c, map_id, 1);
  topay_status s;
  if ((s = c->map2d[map_id].ensure(n2 * 8)) != TOPAY_OK) return s;
  if ((s = c->map3d[map_id].ensure(n3 * 8)) != TOPAY_OK) return s;
  HIPCHK(memcpy_sync(c, c->map2d[map_id].p, esdf2d, n2 * 8, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, c->map3d[map_id].p, esdf3d, n3 * 8, hipMemcpyHostToDevice));
  DevMap& m = c->hmaps[map_id];
  for (int i = 0; i < 3; i++) {
    m.origin[i] = desc->origin[i]; m.dims[i] = desc->dims[i];
    m.min_b[i] = desc->min_boundary[i]; m.max_b[i] = desc->max_boundary[i];
  }
  m.res = desc->resolution;
  m.res_inv = 1.0 / desc->resolution;  // grid_map.cpp:41
  m.esdf2d = (glb_cdp)c->map2d[map_id].as<double>();
  m.esdf3d = (glb_cdp)c->map3d[map_id].as<double>();
  m.esdf2d_inflate = nullptr;
  m.esdf2d_critical = nullptr;
  c->map2d_inf[map_id].release();
  c->map2d_crit[map_id].release();
  c->have_map[map_id] = 1;
  HIPCHK(memcpy_sync(c, (char*)c->dmaps.p + sizeof(DevMap) * map_id, &c->hmaps[map_id], sizeof(DevMap), hipMemcpyHostToDevice));
  return TOPAY_OK;
}

// Read-only map slots shared between the contexts of a device: `c` takes over the descriptors (device pointers) of the
// slots `owner` holds, without a copy of the fields.  The batches in flight of a pipelined planner (one context each)
// then keep one copy of the maps instead of one per context.
extern "C" topay_status topay_share_maps(topay_ctx* c, topay_ctx* owner, int first_map_id, int n_maps) {
  if (!c || !owner || c == owner || first_map_id < 0 || n_maps <= 0 || first_map_id + n_maps > TOPAY_MAX_MAPS) return TOPAY_ERR_INVALID_ARG;
  if (c->device != owner->device) { set_err("topay_share_maps: the contexts are on different devices"); return TOPAY_ERR_INVALID_ARG; }
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }
  for (int m = first_map_id; m < first_map_id + n_maps; m++)
    if (!owner->have_map[m]) return TOPAY_ERR_NO_MAP;
  invalidate_sharers(c, first_map_id, n_maps);   // (contexts that shared c's own copies of these slots)
  drop_shared_slots(c, first_map_id, n_maps);
  for (int m = first_map_id; m < first_map_id + n_maps; m++) {
    c->map2d[m].release(); c->map3d[m].release(); c->map2d_inf[m].release(); c->map2d_crit[m].release();   // own copies of these slots, if any
    c->hmaps[m] = owner->hmaps[m];
    c->have_map[m] = 1;
    // A slot that `owner` itself only shares is registered with the context that holds the fields (the root): it is the
    // root's refill / destroy that frees them, and its list of sharers is the one invalidate_sharers walks.
    topay_ctx* root = owner->map_owner[m] ? owner->map_owner[m] : owner;
    c->map_owner[m] = root;
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    if (std::find(root->map_sharers.begin(), root->map_sharers.end(), c) == root->map_sharers.end()) root->map_sharers.push_back(c);
  }
  for (size_t i = 0; i < c->map_arenas.size();) {   // arenas of own builds that only held these slots
    topay_ctx::MapArena& a = c->map_arenas[i];
    if (a.first >= first_map_id && a.first + a.n <= first_map_id + n_maps) {
      a.buf.release();
      c->map_arenas.erase(c->map_arenas.begin() + (long)i);
    } else {
      i++;
    }
  }
  HIPCHK(memcpy_sync(c, (char*)c->dmaps.p + sizeof(DevMap) * first_map_id, &c->hmaps[first_map_id], sizeof(DevMap) * n_maps, hipMemcpyHostToDevice));
  return TOPAY_OK;
}

static int bucket_of(int N) {
  // diagnostic: TOPAY_FORCE_CLASS=k sends every candidate that fits through launch class k (1-based) or a later one
  static const int force = [] { const char* f = exp_env("TOPAY_FORCE_CLASS"); return f ? atoi(f) : 0; }();
  int k0 = 0;
  for (int k = 0; k < topay_ctx::NBUCKET; k++)
    if (N <= kBucketMaxN[k]) { k0 = k; break; }
  if (N > kBucketMaxN[topay_ctx::NBUCKET - 1]) k0 = topay_ctx::NBUCKET - 1;
  if (force >= 2 && force <= topay_ctx::NBUCKET) k0 = std::max(k0, force - 1);
  return k0;
}

static bool batch_done(topay_ctx* p);
// what the __constant__ parameter block of each device holds (last push)
static DevParams g_pushed_dp[16];
static bool g_pushed_valid[16] = {false};
static topay_status push_params(topay_ctx* c) {
  // Contexts of one process may carry different parameters, and the kernels read them from one __constant__ block for
  // as long as they run: refresh it before every launch, and if a solve of another context with *different*
  // parameters is still in flight on this device, let it finish first (contexts with equal parameters overlap freely).
  {
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    for (topay_ctx* q : g_contexts)
      if (q != c && q->pending && q->device == c->device && memcmp(&q->dp, &c->dp, sizeof(DevParams)) != 0) {
        HIPCHK(hipStreamSynchronize(q->stream));
      }
  }
  HIPCHK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_P), &c->dp, sizeof(DevParams), 0, hipMemcpyHostToDevice, c->stream));
  g_pushed_dp[c->device % 16] = c->dp;
  g_pushed_valid[c->device % 16] = true;
  return TOPAY_OK;
}

static topay_status run_init(topay_ctx* c) {
  const int B = c->B;
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  const int scratch_stride = (3 * c->Pmax + 1 + TOPAY_MAX_N) * ND;
  hipLaunchKernelGGL(k_init, dim3((B + 63) / 64), dim3(64), 0, c->stream, c->db, c->paths.as<double>(),
                     c->path_off.as<long long>(), c->path_len.as<int>(), c->bvel.as<double>(), c->bacc.as<double>(),
                     c->scratch.as<double>(), scratch_stride, TOPAY_MAX_N, 10 * TOPAY_MAX_N - 8);
  HIPCHK(hipGetLastError());
  return TOPAY_OK;
}

// ESDF construction on the device (GridMap::updateESDF, grid_map.cpp:125-521) from the occupancy grids the
// reference fills from its point cloud (grid_map.cpp:733-747): occ2d[x*ny + y] (points below the chassis height),
// occ3d[x*ny*nz + y*nz + z].  The map slots then hold the result exactly as topay_set_map would.  A batch of maps
// of equal dimensions (the benchmark loop: one map per scenario) is built by the same launches, blockIdx.y = map:
// a single 200 x 200 x 16 map has too few lines to fill the device.
topay_status topay_build_esdf_fields(topay_ctx* c, int n_maps, int first_map_id, const topay_map_desc_t* desc,
                                     const signed char* occ2d, const signed char* occ2d_critical, const signed char* occ3d) {
  if (!c || !desc || !occ2d || !occ3d || n_maps <= 0 || first_map_id < 0 || first_map_id + n_maps > TOPAY_MAX_MAPS)
    return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }   // inputs of a solve in flight stay untouched
  const int nx = desc->dims[0], ny = desc->dims[1], nz = desc->dims[2];
  const size_t n2 = (size_t)nx * ny, n3 = n2 * nz, M = (size_t)n_maps;
  if (n2 == 0 || n3 == 0) return TOPAY_ERR_INVALID_ARG;
  if (n3 >= (1ull << 32)) { set_err("map of 2^32 cells or more (the lookups index a field with 32 bits)"); return TOPAY_ERR_UNSUPPORTED; }
  if (desc->dims[2] < 2) { set_err("3-D field with a single layer (the lookups fetch z-neighbours in pairs)"); return TOPAY_ERR_UNSUPPORTED; }
  invalidate_sharers(c, first_map_id, n_maps);
  drop_shared_slots(c, first_map_id, n_maps);
  topay_status s;
  if ((s = c->edt_occ.ensure(M * (n3 + 3 * n2))) != TOPAY_OK) return s;   // 3-D, 2-D, 2-D critical, 2-D scratch
  if ((s = c->edt_tmp1.ensure(M * n3 * 8)) != TOPAY_OK) return s;
  if ((s = c->edt_tmp2.ensure(M * n3 * 8)) != TOPAY_OK) return s;
  // results: e3 | e2 | e2 inflate | e2 critical in a new arena (they stay there); the plain critical field is scratch
  // (a rebuild of the same range of slots -- a new episode's maps -- takes the arena of the previous build over)
  // arenas of earlier builds whose slots this build overwrites completely are released (a caller that varies the slot
  // ranges would otherwise accumulate full-size arenas until topay_destroy)
  for (size_t i = 0; i < c->map_arenas.size();) {
    topay_ctx::MapArena& a = c->map_arenas[i];
    const bool same = a.first == first_map_id && a.n == n_maps;
    if (!same && a.first >= first_map_id && a.first + a.n <= first_map_id + n_maps) {
      a.buf.release();
      c->map_arenas.erase(c->map_arenas.begin() + (long)i);
    } else {
      i++;
    }
  }
  topay_ctx::MapArena* ar = nullptr;
  for (auto& a : c->map_arenas)
    if (a.first == first_map_id && a.n == n_maps) ar = &a;
  if (!ar) {
    c->map_arenas.emplace_back();
    ar = &c->map_arenas.back();
    ar->first = first_map_id;
    ar->n = n_maps;
  }
  DevBuf& arena = ar->buf;
  if ((s = arena.ensure(M * (n3 + 3 * n2) * 8)) != TOPAY_OK) return s;
  if ((s = c->edt_out2.ensure(M * n2 * 8)) != TOPAY_OK) return s;
  // workspace for the envelope stacks of the pass with the most (lines x cells), per map
  const size_t ws_elems = std::max(std::max((size_t)nx * ny * (nz + 2), (size_t)nx * nz * (ny + 2)), (size_t)ny * nz * (nx + 2));
  if ((s = c->edt_v.ensure(M * ws_elems * 4)) != TOPAY_OK) return s;
  if ((s = c->edt_z.ensure(M * ws_elems * 8)) != TOPAY_OK) return s;
  signed char* d_occ3 = c->edt_occ.as<signed char>();
  signed char* d_occ2 = d_occ3 + M * n3;
  signed char* d_occ2c = d_occ2 + M * n2;
  signed char* d_occ2t = d_occ2c + M * n2;
  HIPCHK(hipMemcpyAsync(d_occ3, occ3d, M * n3, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_occ2, occ2d, M * n2, hipMemcpyHostToDevice, c->stream));
  if (occ2d_critical) HIPCHK(hipMemcpyAsync(d_occ2c, occ2d_critical, M * n2, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->ev0, c->stream));
  double* t1 = c->edt_tmp1.as<double>();
  double* t2 = c->edt_tmp2.as<double>();
  double* e3 = arena.as<double>();
  double* e2 = e3 + M * n3;
  double* e2i = e2 + M * n2;       // inflate
  double* e2c = e2i + M * n2;      // critical (holds the critical-inflate field at the end, as the reference's buffer does)
  double* e2s = c->edt_out2.as<double>();   // scratch: the plain critical field
  int* vws = c->edt_v.as<int>();
  double* zws = c->edt_z.as<double>();
  const double res = desc->resolution;
  // envelope stacks in LDS when a 64-line block fits ((n + 2) x 64 x (8 + 2) B <= 150 KB, i.e. lines up to ~238 cells)
  // and the launch is small, else in the HBM workspace
  auto lds_bytes = [](int n) { return (size_t)(n + 2) * 64 * 10; };
  auto launch = [&](auto kern_g, auto kern_l, EdtPass P, long long map_stride, const signed char* occ, const double* src,
                    double* dst, int pass) -> topay_status {
    const int bs = 64;
    P.map_stride = map_stride;
    P.ws_stride = (long long)ws_elems;
    const dim3 grid((unsigned)((P.nlines + bs - 1) / bs), (unsigned)n_maps);
    const size_t lb = lds_bytes(P.n);
    // LDS stacks cut the latency of every envelope step but leave one wave per CU resident (129 KB per 64-line block
    // at n = 200): they win while the launch cannot fill the device anyway (a single benchmark-size map: 2.7 vs 3.8 ms)
    // and lose when there are lines enough to hide the HBM latency instead (1024 maps: 191 vs 142 ms).
    if (lb <= 56 * 1024 || (lb <= 150 * 1024 && (long long)n_maps * P.nlines <= 65536)) {  // short lines: several blocks per CU still fit
      HIPCHK(hipFuncSetAttribute((const void*)kern_l, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
      hipLaunchKernelGGL(kern_l, grid, dim3(bs), lb, c->stream, P, occ, src, dst, vws, zws, pass, res);
    } else {
      hipLaunchKernelGGL(kern_g, grid, dim3(bs), 0, c->stream, P, occ, src, dst, vws, zws, pass, res);
    }
    return TOPAY_OK;
  };
  // Lines of up to 512 cells (every benchmark map: 200 x 200 x 16) take the exhaustive-search passes (topay_edt.h:
  // k_edt_direct / k_edt_tile, 32-bit squared distances between the passes); longer lines the serial envelope passes.
  const bool small_lines = std::max(nx, std::max(ny, nz)) <= 512 && exp_env("TOPAY_EDT_ENVELOPE") == nullptr;
  auto pick_w = [](long long lines) { int w = 1; for (int d = 1; d <= 64; d++) if (lines % d == 0) w = d; return w; };
  int* i1 = (int*)t1;
  int* i2 = (int*)t2;
  auto direct = [&](auto kern, long long n_elems, int n, const signed char* occ, const int* src, int* dst_i, double* dst_d, int pass) {
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_elems + 255) / 256), (unsigned)n_maps), dim3(256), 0, c->stream, n_elems, n, n_elems, occ, src,
                       dst_i, dst_d, pass, res);
  };
  auto tile = [&](auto kern, long long n_elems, int n, int W, long long step, long long inner_tiles, long long outer_stride, long long tiles,
                  const int* src, int* dst_i, double* dst_d, int pass) -> topay_status {
    const size_t lb = (size_t)n * W * sizeof(int);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)n_maps), dim3(256), lb, c->stream, n, W, step, inner_tiles, outer_stride, n_elems,
                       (const signed char*)nullptr, src, dst_i, dst_d, pass, res);
    return TOPAY_OK;
  };
  // final pass of a signed field, both signs at once (k_edt_tile_signed: two tiles of W <= 32 lines in LDS)
  auto pick_w32 = [](long long lines) { int w = 1; for (int d = 1; d <= 32; d++) if (lines % d == 0) w = d; return w; };
  auto signed_x = [&](long long n_elems, int n, int W, long long step, long long inner_tiles, long long outer_stride, long long tiles,
                      const int* sp, const int* sn, double* dst_d) {
    const size_t lb = (size_t)n * W * sizeof(int) * 2;
    hipLaunchKernelGGL(k_edt_tile_signed, dim3((unsigned)tiles, (unsigned)n_maps), dim3(256), lb, c->stream, n, W, step, inner_tiles, outer_stride,
                       n_elems, sp, sn, dst_d, res);
  };
  if (small_lines) {
    // (tiles of up to 512 x 64 cells x 4 B = 128 KB of LDS: above the 64 KB default; set once, not per launch)
    static std::once_flag edt_attr_once[16];
    std::call_once(edt_attr_once[c->device % 16], [] {
      (void)hipFuncSetAttribute((const void*)k_edt_tile<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 64 * 4);
      (void)hipFuncSetAttribute((const void*)k_edt_tile<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 64 * 4);
      (void)hipFuncSetAttribute((const void*)k_edt_tile_signed, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 32 * 4 * 2);
    });
    const int wy = pick_w(nz), wx = pick_w32((long long)ny * nz);
    int* i2n = i2 + M * n3;   // second half of the workspace volume: the negative part's squared distances after the y pass
    for (int pass = 0; pass < 2; pass++) {   // 3-D: along z and y per sign, then along x for both — grid_map.cpp:425-521
      const dim3 g1((unsigned)((n3 + 255) / 256), (unsigned)n_maps);
      if (nz == 16) hipLaunchKernelGGL(k_edt_first_ballot<16>, g1, dim3(256), 0, c->stream, (long long)n3, (long long)n3, (const signed char*)d_occ3, i1, pass);
      else if (nz == 32) hipLaunchKernelGGL(k_edt_first_ballot<32>, g1, dim3(256), 0, c->stream, (long long)n3, (long long)n3, (const signed char*)d_occ3, i1, pass);
      else if (nz == 64) hipLaunchKernelGGL(k_edt_first_ballot<64>, g1, dim3(256), 0, c->stream, (long long)n3, (long long)n3, (const signed char*)d_occ3, i1, pass);
      else direct(k_edt_direct<0, 0>, (long long)n3, nz, d_occ3, nullptr, i1, nullptr, pass);
      if ((s = tile(k_edt_tile<1, 0>, (long long)n3, ny, wy, nz, nz / wy, (long long)ny * nz, (long long)nx * (nz / wy), i1, pass == 0 ? i2 : i2n, nullptr, pass)) != TOPAY_OK) return s;
    }
    signed_x((long long)n3, nx, wx, (long long)ny * nz, ((long long)ny * nz) / wx, 0, ((long long)ny * nz) / wx, i2, i2n, e3);
  }
  for (int pass = 0; pass < 2 && !small_lines; pass++) {
    // 3-D: along z (lines (x, y)), along y (lines (x, z)), along x (lines (y, z)) — grid_map.cpp:425-521
    EdtPass pz{(long long)nx * ny, nz, (long long)nx * ny, 0, (long long)nz, 1, 0, 0};
    EdtPass py{(long long)nx * nz, ny, (long long)nz, (long long)ny * nz, 1, (long long)nz, 0, 0};
    EdtPass px{(long long)ny * nz, nx, (long long)ny * nz, 0, 1, (long long)ny * nz, 0, 0};
    if ((s = launch(k_edt_pass<0, 0, 0>, k_edt_pass<0, 0, 1>, pz, (long long)n3, d_occ3, nullptr, t1, pass)) != TOPAY_OK) return s;
    if ((s = launch(k_edt_pass<1, 0, 0>, k_edt_pass<1, 0, 1>, py, (long long)n3, nullptr, t1, t2, pass)) != TOPAY_OK) return s;
    if ((s = launch(k_edt_pass<1, 1, 0>, k_edt_pass<1, 1, 1>, px, (long long)n3, nullptr, t2, e3, pass)) != TOPAY_OK) return s;
  }
  // One signed 2-D field from an occupancy grid: along y (lines x), along x (lines y), positive then negative part —
  // grid_map.cpp:125-207 and, with other seeds, 211-279, 283-351, 355-423
  auto field2d = [&](const signed char* occ, double* out) -> topay_status {
    if (small_lines) {
      const int w2 = pick_w32(ny);
      int* i1n = i1 + M * n2;
      direct(k_edt_direct<0, 0>, (long long)n2, ny, occ, nullptr, i1, nullptr, 0);
      direct(k_edt_direct<0, 0>, (long long)n2, ny, occ, nullptr, i1n, nullptr, 1);
      signed_x((long long)n2, nx, w2, ny, ny / w2, 0, ny / w2, i1, i1n, out);
      return TOPAY_OK;
    }
    EdtPass qy{(long long)nx, ny, (long long)nx, 0, (long long)ny, 1, 0, 0};
    EdtPass qx{(long long)ny, nx, (long long)ny, 0, 1, (long long)ny, 0, 0};
    for (int pass = 0; pass < 2; pass++) {
      topay_status s2;
      if ((s2 = launch(k_edt_pass<0, 0, 0>, k_edt_pass<0, 0, 1>, qy, (long long)n2, occ, nullptr, t1, pass)) != TOPAY_OK) return s2;
      if ((s2 = launch(k_edt_pass<1, 1, 0>, k_edt_pass<1, 1, 1>, qx, (long long)n2, nullptr, t1, out, pass)) != TOPAY_OK) return s2;
    }
    return TOPAY_OK;
  };
  auto threshold = [&](const double* field, signed char* occ) {
    const long long n = (long long)(M * n2);
    hipLaunchKernelGGL(k_edt_threshold, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, field, c->dp.chassis_colli_radius, occ, n);
  };
  if ((s = field2d(d_occ2, e2)) != TOPAY_OK) return s;              // esdf_buffer_2d
  threshold(e2, d_occ2t);
  if ((s = field2d(d_occ2t, e2i)) != TOPAY_OK) return s;            // esdf_buffer_2d_inflate (355-423)
  if (!occ2d_critical) {
    const long long n = (long long)(M * n2);
    hipLaunchKernelGGL(k_edt_project, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const signed char*)d_occ3, d_occ2c,
                       (long long)n2, nz, (long long)M);
  }
  if ((s = field2d(d_occ2c, e2s)) != TOPAY_OK) return s;            // 2-D critical (211-279)
  threshold(e2s, d_occ2t);
  if ((s = field2d(d_occ2t, e2c)) != TOPAY_OK) return s;            // critical inflate, stored in esdf_buffer_2d_critical (283-351)
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev1, c->stream));
  // the map slots point into the arena; descriptors as topay_set_map
  for (int k = 0; k < n_maps; k++) {
    const int map_id = first_map_id + k;
    c->map2d[map_id].release(); c->map3d[map_id].release(); c->map2d_inf[map_id].release(); c->map2d_crit[map_id].release();
    DevMap& m = c->hmaps[map_id];
    for (int i = 0; i < 3; i++) {
      m.origin[i] = desc->origin[i]; m.dims[i] = desc->dims[i];
      m.min_b[i] = desc->min_boundary[i]; m.max_b[i] = desc->max_boundary[i];
    }
    m.res = desc->resolution;
    m.res_inv = 1.0 / desc->resolution;
    m.esdf2d = (glb_cdp)(e2 + (size_t)k * n2);
    m.esdf3d = (glb_cdp)(e3 + (size_t)k * n3);
    m.esdf2d_inflate = (glb_cdp)(e2i + (size_t)k * n2);
    m.esdf2d_critical = (glb_cdp)(e2c + (size_t)k * n2);
    c->have_map[map_id] = 1;
  }
  HIPCHK(hipMemcpyAsync((char*)c->dmaps.p + sizeof(DevMap) * first_map_id, &c->hmaps[first_map_id], sizeof(DevMap) * n_maps,
                        hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->last_edt_ms = ms;
  // the construction's workspace (occupancy, two intermediate volumes, the envelope stacks, the staged results: about
  // five times the maps themselves) is not needed once the fields sit in their map slots
  DevBuf* ws[] = {&c->edt_occ, &c->edt_tmp1, &c->edt_tmp2, &c->edt_v, &c->edt_z, &c->edt_out2};
  for (DevBuf* b : ws) b->release();
  return TOPAY_OK;
}

topay_status topay_build_esdf_batch(topay_ctx* c, int n_maps, int first_map_id, const topay_map_desc_t* desc,
                                    const signed char* occ2d, const signed char* occ3d) {
  return topay_build_esdf_fields(c, n_maps, first_map_id, desc, occ2d, nullptr, occ3d);
}

topay_status topay_build_esdf(topay_ctx* c, int map_id, const topay_map_desc_t* desc, const signed char* occ2d,
                              const signed char* occ3d) {
  return topay_build_esdf_fields(c, 1, map_id, desc, occ2d, nullptr, occ3d);
}

// The two front-end fields of a map built on the device (GridMap::esdf_buffer_2d_inflate, esdf_buffer_2d_critical).
topay_status topay_get_map_fields(topay_ctx* c, int map_id, double* esdf2d_inflate, double* esdf2d_critical) {
  if (!c || map_id < 0 || map_id >= TOPAY_MAX_MAPS || !c->have_map[map_id]) return TOPAY_ERR_NO_MAP;
  const DevMap& m = c->hmaps[map_id];
  if (!m.esdf2d_inflate || !m.esdf2d_critical) { set_err("map slot was not built by topay_build_esdf*"); return TOPAY_ERR_NO_MAP; }
  HIPCHK(hipSetDevice(c->device));
  const size_t n2 = (size_t)m.dims[0] * m.dims[1];
  if (esdf2d_inflate) HIPCHK(memcpy_sync(c, esdf2d_inflate, (const void*)m.esdf2d_inflate, n2 * 8, hipMemcpyDeviceToHost));
  if (esdf2d_critical) HIPCHK(memcpy_sync(c, esdf2d_critical, (const void*)m.esdf2d_critical, n2 * 8, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

// Copy a resident map back (tests, or a caller that wants the GPU-built ESDF on the host); milliseconds of the last build.
topay_status topay_get_map(topay_ctx* c, int map_id, double* esdf2d, double* esdf3d, double* build_ms) {
  if (!c || map_id < 0 || map_id >= TOPAY_MAX_MAPS || !c->have_map[map_id]) return TOPAY_ERR_NO_MAP;
  HIPCHK(hipSetDevice(c->device));
  const DevMap& m = c->hmaps[map_id];
  const size_t n2 = (size_t)m.dims[0] * m.dims[1], n3 = n2 * m.dims[2];
  if (esdf2d) HIPCHK(memcpy_sync(c, esdf2d, (const void*)m.esdf2d, n2 * 8, hipMemcpyDeviceToHost));
  if (esdf3d) HIPCHK(memcpy_sync(c, esdf3d, (const void*)m.esdf3d, n3 * 8, hipMemcpyDeviceToHost));
  if (build_ms) *build_ms = c->last_edt_ms;
  return TOPAY_OK;
}

topay_status topay_set_init_traj(topay_ctx* c, int batch, const int* path_len, const double* init_paths,
                                 const double* boundary_vel, const double* boundary_acc, const int* map_ids) {
  if (!c || batch <= 0 || !path_len || !init_paths) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }   // inputs of a solve in flight stay untouched
  c->have_traj = false;
  c->solved = false;
  std::vector<long long> off(batch + 1, 0);
  int Pmax = 0;
  for (int b = 0; b < batch; b++) {
    if (path_len[b] < 2) { set_err("every init path needs at least 2 states"); return TOPAY_ERR_INVALID_ARG; }
    off[b + 1] = off[b] + path_len[b];
    Pmax = std::max(Pmax, path_len[b]);
  }
  std::vector<int> mids(batch, 0);
  for (int b = 0; b < batch; b++) {
    if (map_ids) mids[b] = map_ids[b];
    if (mids[b] < 0 || mids[b] >= TOPAY_MAX_MAPS || !c->have_map[mids[b]]) { set_err("map slot not set"); return TOPAY_ERR_NO_MAP; }
  }
  c->B = batch;
  c->Pmax = Pmax;
  c->h_map_id = mids;
  const size_t tot = (size_t)off[batch];
  topay_status s;
#define ENS(buf, bytes) if ((s = c->buf.ensure(bytes)) != TOPAY_OK) return s
  ENS(paths, tot * 10 * 8);
  ENS(path_off, (size_t)(batch + 1) * 8);
  ENS(path_len, (size_t)batch * 4);
  ENS(bvel, (size_t)batch * 20 * 8);
  ENS(bacc, (size_t)batch * 20 * 8);
  ENS(scratch, (size_t)batch * (3 * Pmax + 1 + TOPAY_MAX_N) * ND * 8);
  ENS(N, (size_t)batch * 4);
  ENS(s1_past, (size_t)batch * 4);
  ENS(map_id, (size_t)batch * 4);
  ENS(head, (size_t)batch * 27 * 8);
  ENS(tail, (size_t)batch * 27 * 8);
  ENS(start_xy, (size_t)batch * 2 * 8);
  ENS(goal_xy, (size_t)batch * 2 * 8);
  ENS(init_xy, (size_t)batch * 2 * TOPAY_MAX_N * 8);
  ENS(x0, (size_t)batch * (10 * TOPAY_MAX_N - 8) * 8);
  ENS(order, (size_t)batch * 4);
  HIPCHK(memcpy_sync(c, c->paths.p, init_paths, tot * 10 * 8, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, c->path_off.p, off.data(), (size_t)(batch + 1) * 8, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, c->path_len.p, path_len, (size_t)batch * 4, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, c->map_id.p, mids.data(), (size_t)batch * 4, hipMemcpyHostToDevice));
  if (boundary_vel) HIPCHK(memcpy_sync(c, c->bvel.p, boundary_vel, (size_t)batch * 20 * 8, hipMemcpyHostToDevice));
  else HIPCHK(hipMemsetAsync(c->bvel.p, 0, (size_t)batch * 20 * 8, c->stream));
  if (boundary_acc) HIPCHK(memcpy_sync(c, c->bacc.p, boundary_acc, (size_t)batch * 20 * 8, hipMemcpyHostToDevice));
  else HIPCHK(hipMemsetAsync(c->bacc.p, 0, (size_t)batch * 20 * 8, c->stream));
  DevBatch& d = c->db;
  memset(&d, 0, sizeof(d));
  d.B = batch;
  d.N = c->N.as<int>(); d.s1_past = c->s1_past.as<int>(); d.map_id = c->map_id.as<int>();
  d.head = c->head.as<double>(); d.tail = c->tail.as<double>();
  d.start_xy = c->start_xy.as<double>(); d.goal_xy = c->goal_xy.as<double>();
  d.init_xy = c->init_xy.as<double>(); d.x0 = c->x0.as<double>();
  d.order = c->order.as<int>();
  if ((s = run_init(c)) != TOPAY_OK) return s;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->hN.assign(batch, 0);
  HIPCHK(memcpy_sync(c, c->hN.data(), c->N.p, (size_t)batch * 4, hipMemcpyDeviceToHost));
  int Nmax = 0;
  for (int b = 0; b < batch; b++) {
    if (c->hN[b] <= 0) c->hN[b] = 0;  // needs more than TOPAY_MAX_N pieces: reported as failed, never launched
    Nmax = std::max(Nmax, c->hN[b]);
  }
  if (Nmax == 0) { set_err("every trajectory needs more pieces than TOPAY_MAX_N"); return TOPAY_ERR_TOO_MANY_PIECES; }
  c->Nmax = Nmax;
  c->h_poff.assign((size_t)batch + 1, 0);
  c->h_noff.assign((size_t)batch + 1, 0);
  for (int b = 0; b < batch; b++) {
    c->h_poff[b + 1] = c->h_poff[b] + c->hN[b];
    c->h_noff[b + 1] = c->h_noff[b] + (c->hN[b] > 0 ? 10 * c->hN[b] - 8 : 0);
  }
  const size_t P = (size_t)c->h_poff[batch], NN = (size_t)c->h_noff[batch];
  const int m = std::max(c->hp.s1_lbfgs.mem_size, c->hp.s2_lbfgs.mem_size);
  // launch order: longest trajectories first inside each row class (tail latency)
  std::vector<int> idx(batch);
  std::iota(idx.begin(), idx.end(), 0);
  // more pieces first, then more path states (both correlate ~0.45 with the number of evaluations a candidate needs)
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b2) {
    return c->hN[a] != c->hN[b2] ? c->hN[a] > c->hN[b2] : path_len[a] > path_len[b2];
  });
  for (auto& v : c->cls) v.clear();
  for (int b : idx) {
    if (c->hN[b] == 0) continue;
    c->cls[bucket_of(c->hN[b])].push_back(b);
  }
  c->h_path_len.assign(path_len, path_len + batch);
  {
    std::vector<int> ord;
    for (int k = topay_ctx::NBUCKET - 1; k >= 0; k--) ord.insert(ord.end(), c->cls[k].begin(), c->cls[k].end());
    ord.resize(batch, 0);
    HIPCHK(memcpy_sync(c, c->order.p, ord.data(), (size_t)batch * 4, hipMemcpyHostToDevice));
  }
  ENS(poff, ((size_t)batch + 1) * 8);
  ENS(noff, ((size_t)batch + 1) * 8);
  HIPCHK(memcpy_sync(c, c->poff.p, c->h_poff.data(), ((size_t)batch + 1) * 8, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, c->noff.p, c->h_noff.data(), ((size_t)batch + 1) * 8, hipMemcpyHostToDevice));
  // every block sized by the candidates' own pieces / decision vectors (the history, 2 m n doubles per candidate, is
  // by far the largest: 0.4 MB at the benchmark's mean of 11 pieces, 7 MB at 170)
  ENS(x, NN * 8);
  ENS(work, 4 * NN * 8);
  ENS(hist_s, (size_t)m * NN * 8);
  ENS(hist_y, (size_t)m * NN * 8);
  ENS(hist_ys, (size_t)batch * m * 8);
  ENS(hist_alpha, (size_t)batch * m * 8);
  ENS(lu, 84 * P * 8);
  ENS(success, (size_t)batch * 4);
  ENS(cost, (size_t)batch * 8);
  ENS(stats, (size_t)batch * 8 * 4);
  ENS(xyerr, (size_t)batch * 2 * 8);
  ENS(coef, 54 * P * 8);
  ENS(T, P * 8);
  ENS(knots, 2 * (P + batch) * 8);
  ENS(alm, (size_t)batch * 4 * 8);
  ENS(fout, (size_t)batch * 8);
  ENS(sbuf, 14 * TOPAY_EP * P * 8);
  ENS(mstash, 36 * TOPAY_EP * P * 8);
  ENS(elapsed, (size_t)batch * 8);
  ENS(startus, (size_t)batch * 8);
  ENS(hwid, (size_t)batch * 4);
  ENS(feas_flags, (size_t)batch * 2 * 4);
  ENS(feas_report, (size_t)batch * 38 * 8);
  ENS(interrupted, (size_t)batch * 4);
#undef ENS
  {
    DevBuf* all[] = {&c->paths, &c->path_off, &c->path_len, &c->bvel, &c->bacc, &c->scratch, &c->N, &c->s1_past, &c->map_id, &c->head, &c->tail,
                     &c->start_xy, &c->goal_xy, &c->init_xy, &c->x0, &c->order, &c->poff, &c->noff, &c->x, &c->work, &c->hist_s, &c->hist_y,
                     &c->hist_ys, &c->hist_alpha, &c->lu, &c->success, &c->cost, &c->stats, &c->xyerr, &c->coef, &c->T, &c->knots, &c->alm,
                     &c->fout, &c->sbuf, &c->mstash, &c->elapsed, &c->startus, &c->hwid};
    c->workspace_bytes = 0;
    for (DevBuf* q : all) c->workspace_bytes += q->bytes;
  }
  d.hist_m = m;
  d.poff = c->poff.as<long long>(); d.noff = c->noff.as<long long>();
  d.x = c->x.as<double>(); d.work = c->work.as<double>();
  d.hist_s = c->hist_s.as<double>(); d.hist_y = c->hist_y.as<double>();
  d.hist_ys = c->hist_ys.as<double>(); d.hist_alpha = c->hist_alpha.as<double>();
  d.lu = c->lu.as<double>();
  d.success = c->success.as<int>(); d.cost = c->cost.as<double>(); d.stats = c->stats.as<int>();
  d.xyerr = c->xyerr.as<double>(); d.coef = c->coef.as<double>(); d.T = c->T.as<double>();
  d.knots = c->knots.as<double>(); d.alm = c->alm.as<double>(); d.fout = c->fout.as<double>();
  d.sbuf = c->sbuf.as<double>();
  d.mstash = c->mstash.as<double>();
  d.elapsed_us = c->elapsed.as<double>();
  d.start_us = c->startus.as<double>();
  d.hw_id = c->hwid.as<int>();
  d.gate_in_solve = c->gate_in_solve ? 1 : 0;
  d.feas_flags = c->feas_flags.as<int>();
  d.feas_report = c->feas_report.as<double>();
  d.interrupted = c->interrupted.as<int>();
  HIPCHK(hipMemsetAsync(c->interrupted.p, 0, (size_t)batch * 4, c->stream));
  // a new batch has no planning-call groups until topay_set_groups says so
  c->h_group.clear();
  c->n_groups = 0;
  d.group_id = nullptr; d.group_tau = nullptr; d.cancel_budget = 0; d.cancel_flag = nullptr;
  c->gate_done = false;
  HIPCHK(hipMemsetAsync(c->elapsed.p, 0, (size_t)batch * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->success.p, 0, (size_t)batch * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->cost.p, 0xFF, (size_t)batch * 8, c->stream));   // never-launched candidates: cost = NaN
  HIPCHK(hipMemsetAsync(c->stats.p, 0, (size_t)batch * 32, c->stream));
  c->have_traj = true;
  return TOPAY_OK;
}

topay_status topay_reset(topay_ctx* c) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }   // inputs of a solve in flight stay untouched
  c->solved = false;
  return run_init(c);
}

}  // extern "C"

// Persistent grids.  A workgroup of class k occupies the fraction r_k of a compute unit -- the larger of its share of
// the register file (NW waves of 512 / occ registers on four SIMDs of 512) and of the 160 KB of LDS -- for the time its
// share of the class's work takes: work_k = sum of N^1.5 over the class (the cost of an evaluation grows with N, the
// number of evaluations slowly), divided by the speed-up of NW waves, times the slow-down of a wave that shares its SIMD.
// The grids are proportional to that workgroup-time and scaled so that together they ask for exactly the compute units
// there are (x oversubscription): all launches of a batch end together and none of their workgroups waits in the
// dispatcher.  nm[k] = longest candidate the class's LDS is sized for.
static void compute_grids(topay_ctx* c, const int* nm, double cus, int* grid) {
  const ClassDef* ct = class_table();
  double wt[topay_ctx::NBUCKET] = {0}, rk[topay_ctx::NBUCKET] = {0}, need = 0.0;
  for (int k = 0; k < topay_ctx::NBUCKET; k++) {
    grid[k] = 0;
    if (c->cls[k].empty()) continue;
    double t = ct[k].nw == 1 ? 1.0 : (ct[k].nw == 2 ? 1.0 / 1.48 : 0.5);   // time of a workgroup per unit of work
    const double regs = (double)ct[k].nw / (4.0 * ct[k].occ), lds = (double)class_lds_bytes(ct[k], nm[k]) / (160.0 * 1024.0);
    rk[k] = std::max(regs, lds);
    // a wave that shares its SIMD runs slower (two of them get through occ2_gain times the work of one); classes whose LDS
    // keeps them from sharing are not slowed down
    if (ct[k].occ == 2 && lds <= 0.1875) t *= 2.0 / c->occ2_gain;
    // the smallest class's workgroups cannot take over anybody's queue, the others can take over its: it gets less than its share
    // (measured in round 3, tools/experiments/r3_bias.sh, factor 1.0 / 0.9 / 0.8 / 0.7: serial step 1.00 / 0.99 / 0.98 / 0.97 s)
    static const double bias0 = [] { const char* e = exp_env("TOPAY_SHARE_BIAS0"); return e ? atof(e) : 0.8; }();
    if (k == 0) t *= bias0;
    for (int b : c->cls[k]) wt[k] += t * std::pow((double)c->hN[b], 1.5);
    need += wt[k] * rk[k];
  }
  if (need <= 0.0) return;
  const double G = cus / need;
  for (int k = 0; k < topay_ctx::NBUCKET; k++) {
    if (c->cls[k].empty()) continue;
    grid[k] = std::max(1, std::min((int)c->cls[k].size(), (int)std::floor(G * wt[k] + 0.5)));
  }
}

// Dynamic LDS above the 64 KB default needs the attribute; it is set once per device to the most a kernel can ask
// for (the launch itself passes the size it needs), not per launch: two host threads launching different contexts
// would otherwise interleave set(small), set(large), launch(large).
static std::once_flag g_attr_once[16];
static hipError_t g_attr_err[16];
static hipError_t set_kernel_attributes(int device) {
  std::call_once(g_attr_once[device % 16], [device] {
    (void)device;
    hipError_t e = hipSuccess;
    const ClassDef* ct = class_table();
    // The whole LDS of a compute unit for every kernel: an attribute below a launch's request is an error on a runtime
    // that enforces it, and nothing is gained by asking for less.
    const int lds = kLdsDoublesPerCU * 8;
    for (int k = 0; k < TOPAY_NBUCKET && e == hipSuccess; k++) {
      e = hipFuncSetAttribute((const void*)ct[k].solve, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ct[k].eval, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess && ct[k].lat) e = hipFuncSetAttribute((const void*)ct[k].lat, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    g_attr_err[device % 16] = e;
  });
  return g_attr_err[device % 16];
}

// Lowest class whose queue a workgroup of class k may go on with once its own is empty: a smaller class its kernel
// covers AND whose candidates it solves to the same bits (the division of the L-BFGS vectors over the threads depends on
// the waves per trajectory, so only classes with the same number of waves), and never across the long / common boundary
// (a resident workgroup of a long class holds LDS or whole compute units the common classes' workgroups want).
static int steal_floor(int k) {
  const ClassDef* ct = class_table();
  int lo = k;
  while (lo > 0 && ct[lo - 1].nw == ct[k].nw && ct[lo - 1].snw() == ct[k].snw() && ((lo - 1 >= kBigFirst) == (k >= kBigFirst))) lo--;
  return lo;
}

template <bool EVAL, typename... Args>
static topay_status launch_classes(topay_ctx* c, bool persistent, Args... args) {
  // One launch per N-bucket, each on its own stream so that the tail of one bucket overlaps the others.
  // Longest jobs first.  The context's main stream waits for all of them (events), so the caller's
  // ev0/ev1 pair on the main stream brackets the whole solve.
  const ClassDef* ct = class_table();
  int launches = 0, helper_launches = 0, off = 0;
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  HIPCHK(set_kernel_attributes(c->device));
  int slots = 0;
  if (persistent) {
    if (c->qnext.ensure(sizeof(int) * topay_ctx::NBUCKET) != TOPAY_OK) return TOPAY_ERR_NO_DEVICE;
    HIPCHK(hipMemsetAsync(c->qnext.p, 0, sizeof(int) * topay_ctx::NBUCKET, c->stream));
    slots = c->simd_slots;
  }
  int pgrid[topay_ctx::NBUCKET] = {0}, nmk[topay_ctx::NBUCKET] = {0};
  for (int k = 0; k < topay_ctx::NBUCKET; k++) {
    for (int b : c->cls[k]) nmk[k] = std::max(nmk[k], c->hN[b]);
    // classes that may take over each other's queues run the same kernel: its LDS must hold the longest candidate of any of them
    if (persistent && c->steal)
      for (int k2 = steal_floor(k); k2 < k; k2++)
        for (int b : c->cls[k2]) nmk[k] = std::max(nmk[k], c->hN[b]);
  }
  if (persistent) {
    // 8 % more workgroups than SIMD slots: in steady state 3-5 % of the SIMDs have no workgroup because the ones still
    // pending do not find LDS on the compute units where a SIMD is free (54-107 KB workgroups beside 21-36 KB ones); a few
    // pending workgroups more, mostly of the small classes, fill those.  Measured, interleaved on one box
    // (TOPAY_OVERSUBSCRIBE=1.0 / 1.08): 10.01 / 10.20, 10.06 / 10.19, 10.04 / 10.07k trajectories/s; 1.2 is no better.
    static const double over = [] { const char* e = exp_env("TOPAY_OVERSUBSCRIBE"); return e ? atof(e) : 1.08; }();
    compute_grids(c, nmk, slots / 4.0 * over, pgrid);
  }
  HIPCHK(hipEventRecord(c->bstart, c->stream));  // params + resets on the main stream come first
  for (int k = topay_ctx::NBUCKET - 1; k >= 0; k--) {
    const std::vector<int>& v = c->cls[k];
    const int nk = (int)v.size();
    if (nk == 0) continue;
    const int nm = nmk[k];
    DevBatch d = c->db;
    d.order = c->db.order + off;
    int grid = nk;
    if (persistent) {
      d.order = c->db.order;
      d.queue_next = c->qnext.as<int>();
      d.queue_class = k;
      d.queue_lowest = c->steal ? steal_floor(k) : k;
      int o2 = 0;
      for (int kk = topay_ctx::NBUCKET - 1; kk >= 0; kk--) {   // `order` holds the classes largest first
        d.queue_off[kk] = o2;
        d.queue_count[kk] = (int)c->cls[kk].size();
        o2 += d.queue_count[kk];
      }
      grid = pgrid[k];
    }
    off += nk;
    // helper-wave kernels (topay_set_latency_mode): a one-wave class of a small batch runs on four-wave workgroups whose
    // extra waves only join the evaluations -- same bits, shorter sample sweeps
    const bool lat = !EVAL && ct[k].lat && (c->latency_mode == 2 || (c->latency_mode == 1 && c->B <= c->simd_slots));
    size_t lds = class_lds_bytes(ct[k], nm);
    if (lat) {
      lds = (size_t)(lds_doubles_mw(nm, kLatWaves) + 8 + 40 + TOPAY_CMD_DOUBLES) * sizeof(double);
      grid = nk;   // a workgroup per candidate of the class
    }
    if (k == topay_ctx::NBUCKET - 2 && !c->cls[topay_ctx::NBUCKET - 1].empty() && c->bstream[k] == c->stream)
      HIPCHK(hipStreamCreateWithFlags(&c->bstream[k], hipStreamNonBlocking));   // both long classes in one batch: they must not serialise
    hipStream_t st = c->bstream[k];
    if (st != c->stream) HIPCHK(hipStreamWaitEvent(st, c->bstart, 0));
    if constexpr (EVAL) hipLaunchKernelGGL(ct[k].eval, dim3(grid), dim3(64 * ct[k].nw), lds, st, d, (const DevMap*)c->dmaps.p, args..., nm);
    else if (lat) { hipLaunchKernelGGL(ct[k].lat, dim3(grid), dim3(64 * kLatWaves), lds, st, d, (const DevMap*)c->dmaps.p, nm); helper_launches++; }
    else hipLaunchKernelGGL(ct[k].solve, dim3(grid), dim3(64 * ct[k].nw), lds, st, d, (const DevMap*)c->dmaps.p, nm);
    HIPCHK(hipGetLastError());
    if (st != c->stream) HIPCHK(hipEventRecord(c->bevent[k], st));
    launches++;
  }
  for (int k = 0; k < topay_ctx::NBUCKET; k++)
    if (!c->cls[k].empty() && c->bstream[k] != c->stream) HIPCHK(hipStreamWaitEvent(c->stream, c->bevent[k], 0));
  c->last_launches = launches;
  c->last_helper_launches = helper_launches;
  return TOPAY_OK;
}


static bool batch_done(topay_ctx* p) { return !p->pending || hipStreamQuery(p->stream) == hipSuccess; }

extern "C" {

topay_status topay_optimize_async(topay_ctx* c) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) {  // a second solve on a context whose first has not been waited for: finish that one first
    topay_status s0 = topay_synchronize(c);
    if (s0 != TOPAY_OK) return s0;
  }
  {
    // Dispatch gate.  Batches of different contexts run on different streams; issued at the same time their waves
    // would be dispatched alternately and both would end in the same long tail.  Holding the new batch back until
    // every candidate of the previous one is resident gives oldest-first scheduling without stream priorities: the
    // new waves take exactly the SIMDs the previous batch's tail leaves idle.  Host-side wait on a counter in pinned
    // memory (<= one bulk phase); results do not depend on it.
    std::lock_guard<std::mutex> lk(g_issue_mutex);
    topay_ctx* p = g_last_issued;
    if (c->gate && p && p != c && p->pending && p->device == c->device && p->h_started) {
      volatile int* cnt = p->h_started;
      const auto t0 = std::chrono::steady_clock::now();
      while (*cnt < p->n_gate) {
        if (batch_done(p)) break;  // finished (or never launched anything)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
          // scheduling only: the batch is issued anyway, but the caller can see that the hand-over did not happen
          c->gate_timeouts++;
          set_err("dispatch gate: the previous batch did not become resident within 120 s; issuing anyway");
          break;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
      }
    }
    c->h_started[0] = 0;
    int nl = 0, ng = 0;
    for (int k = 0; k < topay_ctx::NBUCKET; k++) {
      nl += (int)c->cls[k].size();
      if (k < kBigFirst) ng += (int)c->cls[k].size();
    }
    c->n_launched = nl;
    // The gate waits for the candidates of the three common classes only: the few workgroups of the two classes of
    // long candidates need 70 / 104 KB of LDS and may not find a compute unit with that much free until the previous
    // batch's tail -- holding the whole next batch back for them leaves the rest of the device idle meanwhile.
    c->n_gate = ng;
    c->db.gate_maxN = kBucketMaxN[kBigFirst - 1];   // (the common classes: up to 32 pieces)
    void* dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, c->h_started, 0));
    c->db.started = (int*)dp;
    g_last_issued = c;
  }
  HIPCHK(hipEventRecord(c->ev0, c->stream));
  // cancellation state of this solve: nobody has succeeded yet (clock "infinity"), nothing is interrupted
  c->h_cancel[0] = 0;
  {
    void* dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, c->h_cancel, 0));
    c->db.cancel_flag = (const int*)dp;
  }
  c->h_started[12] = 0;
  {
    void* dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, c->h_started + 12, 0));
    c->db.gate_truncated = (int*)dp;
  }
  c->db.cancel_budget = c->n_groups > 0 ? c->cancel_budget : 0;
  if (c->n_groups > 0) HIPCHK(hipMemsetAsync(c->group_tau.p, 0x7f, (size_t)c->n_groups * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->interrupted.p, 0, (size_t)c->B * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->feas_flags.p, 0, (size_t)c->B * 8, c->stream));
  c->gate_done = false;
  // candidates that were not launched keep success = 0 and cost = NaN
  HIPCHK(hipMemsetAsync(c->success.p, 0, (size_t)c->B * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->cost.p, 0xFF, (size_t)c->B * 8, c->stream));
  topay_status s = launch_classes<false>(c, c->persistent);
  if (s != TOPAY_OK) {
    // some class launches may already be running on the batch's buffers: nothing may touch them before they have ended
    for (int k = 0; k < topay_ctx::NBUCKET; k++)
      if (c->bstream[k]) (void)hipStreamSynchronize(c->bstream[k]);
    (void)hipStreamSynchronize(c->stream);
    return s;
  }
  HIPCHK(hipEventRecord(c->ev1, c->stream));
  c->pending = true;
  return TOPAY_OK;
}

topay_status topay_synchronize(topay_ctx* c) {
  if (!c) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->pending) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->last_ms = ms;
    c->solved = true;
    c->pending = false;
    // (a candidate whose history block could not hold the gate's scratch -- a small mem_size -- was left ungated by its
    // wave: the verdicts are then taken by the separate kernel, with scratch of the right size, at the first request)
    c->gate_done = c->gate_in_solve && c->h_started[12] == 0;
    if (c->n_groups > 0 && c->cancel_budget > 0) {
      // The rule, applied once more to the finished batch so that the outcome does not depend on WHEN a candidate saw its
      // group's clock: a candidate counts iff its own work clock is within cancel_budget of the smallest clock of a
      // feasible success of its planning call.  (A candidate stopped on the device had already passed that limit with
      // the clock it saw, which was no smaller than the final one; one that ran to its end before the first success of
      // its call was published is stopped here.)
      const int B = c->B;
      std::vector<int> succ(B), st((size_t)B * 8), fl((size_t)B * 2), intr(B), tau(c->n_groups);
      HIPCHK(memcpy_sync(c, succ.data(), c->success.p, (size_t)B * 4, hipMemcpyDeviceToHost));
      HIPCHK(memcpy_sync(c, st.data(), c->stats.p, (size_t)B * 32, hipMemcpyDeviceToHost));
      HIPCHK(memcpy_sync(c, fl.data(), c->feas_flags.p, (size_t)B * 8, hipMemcpyDeviceToHost));
      HIPCHK(memcpy_sync(c, intr.data(), c->interrupted.p, (size_t)B * 4, hipMemcpyDeviceToHost));
      HIPCHK(memcpy_sync(c, tau.data(), c->group_tau.p, (size_t)c->n_groups * 4, hipMemcpyDeviceToHost));
      bool changed = false;
      for (int b = 0; b < B; b++) {
        const int g = c->h_group[b];
        if (g < 0 || intr[b] || c->hN[b] == 0) continue;
        const long long clock = (long long)(st[(size_t)b * 8 + 2] + st[(size_t)b * 8 + 5]) * c->hN[b];
        if (clock > (long long)tau[g] + c->cancel_budget) {
          intr[b] = 1; succ[b] = 0; fl[2 * b] = 0; fl[2 * b + 1] = 0;
          st[(size_t)b * 8 + 3] = TOPAY_INTERRUPTED;
          changed = true;
        }
      }
      if (changed) {
        HIPCHK(memcpy_sync(c, c->success.p, succ.data(), (size_t)B * 4, hipMemcpyHostToDevice));
        HIPCHK(memcpy_sync(c, c->stats.p, st.data(), (size_t)B * 32, hipMemcpyHostToDevice));
        HIPCHK(memcpy_sync(c, c->feas_flags.p, fl.data(), (size_t)B * 8, hipMemcpyHostToDevice));
        HIPCHK(memcpy_sync(c, c->interrupted.p, intr.data(), (size_t)B * 4, hipMemcpyHostToDevice));
      }
    }
  }
  return TOPAY_OK;
}

// == the planner's thread group (planner.cpp:829-952): group_id[b] = planning call (scenario) of candidate b, -1 = none.
// With a positive cancel budget the candidates of a call that are still running `budget` piece-evaluations after the
// call's first success that passes the gate are interrupted (threads.interrupt_all() 100 ms after future_succ; the unit
// is alm_work_budget's: 24 000 = 1 s, so 100 ms = 2400).  Call after topay_set_init_traj; 0 / NULL switches it off.
// Launch order of the resident batch: inside every class longest first (the tail of a batch), or -- with the planner's
// cancellation -- shortest first (see topay_set_groups).
static topay_status upload_order(topay_ctx* c, bool shortest_first) {
  std::vector<int> ord;
  for (int k = topay_ctx::NBUCKET - 1; k >= 0; k--) {
    std::vector<int>& v = c->cls[k];
    if (shortest_first) std::stable_sort(v.begin(), v.end(), [&](int a, int b2) { return c->hN[a] < c->hN[b2]; });
    else std::stable_sort(v.begin(), v.end(), [&](int a, int b2) {
      return c->hN[a] != c->hN[b2] ? c->hN[a] > c->hN[b2] : (c->h_path_len[a] != c->h_path_len[b2] ? c->h_path_len[a] > c->h_path_len[b2] : a < b2);
    });
    ord.insert(ord.end(), v.begin(), v.end());
  }
  ord.resize(c->B, 0);
  HIPCHK(memcpy_sync(c, c->order.p, ord.data(), (size_t)c->B * 4, hipMemcpyHostToDevice));
  return TOPAY_OK;
}

topay_status topay_set_groups(topay_ctx* c, const int* group_id, int cancel_budget) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (cancel_budget < 0) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }
  const bool had_groups = c->n_groups > 0;
  c->h_group.clear();
  c->n_groups = 0;
  c->cancel_budget = cancel_budget;
  c->db.group_id = nullptr; c->db.group_tau = nullptr;
  if (!group_id || cancel_budget == 0) {
    if (had_groups) return upload_order(c, false);   // back to longest first
    return TOPAY_OK;
  }
  if (!c->gate_in_solve) { set_err("cancellation needs the in-solve feasibility gate (switched off in this experiments build: TOPAY_GATE_IN_SOLVE=0)"); return TOPAY_ERR_UNSUPPORTED; }
  // The in-solve gate's scratch is the candidate's dead L-BFGS history block (mem_size x n doubles twice); a candidate whose
  // block is too short is left to the separate kernel and would never publish its call's clock: the window would silently
  // stay shut.  64 rows hold the gate's panels and sample times of a trajectory three times as long as its initial guess.
  if (std::max(c->hp.s1_lbfgs.mem_size, c->hp.s2_lbfgs.mem_size) < 64) {
    set_err("cancellation window: the L-BFGS mem_size must be at least 64 (the in-solve gate works in the history block)");
    return TOPAY_ERR_INVALID_ARG;
  }
  // the caller's ids (any integers >= 0, e.g. global scenario numbers of a sharded sweep; -1 = no planning call) become
  // dense indices: the device holds one clock per planning call that is present, not one per possible id
  std::vector<int> dense(c->B, -1);
  {
    std::vector<int> ids;
    for (int b = 0; b < c->B; b++) {
      if (group_id[b] < -1) return TOPAY_ERR_INVALID_ARG;
      if (group_id[b] >= 0) ids.push_back(group_id[b]);
    }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    for (int b = 0; b < c->B; b++)
      if (group_id[b] >= 0) dense[b] = (int)(std::lower_bound(ids.begin(), ids.end(), group_id[b]) - ids.begin());
    c->n_groups = (int)ids.size();
  }
  const int ng = c->n_groups;
  c->h_group = dense;
  if (ng == 0) { c->cancel_budget = cancel_budget; return TOPAY_OK; }
  topay_status s;
  if ((s = c->group_id.ensure((size_t)c->B * 4)) != TOPAY_OK) return s;
  if ((s = c->group_tau.ensure((size_t)std::max(1, ng) * 4)) != TOPAY_OK) return s;
  HIPCHK(memcpy_sync(c, c->group_id.p, dense.data(), (size_t)c->B * 4, hipMemcpyHostToDevice));
  c->db.group_id = c->group_id.as<int>();
  c->db.group_tau = c->group_tau.as<int>();
  // Launch order with cancellation: shortest candidates first inside every class.  Without it the longest go first (they
  // are the tail of the batch); with it they are the ones the rule interrupts, and they can only be stopped early if the
  // short candidates of their planning call -- the ones that succeed first on the work clock -- have already run.  The
  // outcome does not depend on the order (the rule is applied to the candidates' own clocks), only the time saved does.
  return upload_order(c, true);
}

// Helper-wave kernels for small batches (include/topay.h)
topay_status topay_set_latency_mode(topay_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return TOPAY_ERR_INVALID_ARG;
  c->latency_mode = mode;
  return TOPAY_OK;
}

// threads.interrupt_all() for the solve in flight (planner.cpp:952): every candidate stops at its next interruption
// point (top of the ALM loop / next stage-2 evaluation); returns at once, topay_synchronize waits for the kernels.
topay_status topay_cancel(topay_ctx* c) {
  if (!c) return TOPAY_ERR_INVALID_ARG;
  if (c->h_cancel) __atomic_store_n(c->h_cancel, 1, __ATOMIC_RELEASE);
  return TOPAY_OK;
}

// interrupted[b] = 1: candidate b was stopped by the cancellation rule or by topay_cancel (no trajectory, success 0)
topay_status topay_get_interrupted(topay_ctx* c, int* interrupted) {
  if (!c || !c->have_traj || !interrupted) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }
  HIPCHK(memcpy_sync(c, interrupted, c->interrupted.p, (size_t)c->B * 4, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

topay_status topay_optimize(topay_ctx* c) {
  topay_status s = topay_optimize_async(c);
  if (s != TOPAY_OK) return s;
  return topay_synchronize(c);
}

topay_status topay_optimize_within(topay_ctx* c, double budget_ms, int* timed_out) {
  if (timed_out) *timed_out = 0;
  if (!(budget_ms > 0.0)) return TOPAY_ERR_INVALID_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  topay_status s = topay_optimize_async(c);
  if (s != TOPAY_OK) return s;
  while (hipStreamQuery(c->stream) == hipErrorNotReady) {
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() >= budget_ms) {
      (void)topay_cancel(c);
      if (timed_out) *timed_out = 1;
      break;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  return topay_synchronize(c);
}

topay_status topay_get_nmax(topay_ctx* c, int* nmax, int* Nmax) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (nmax) *nmax = 10 * c->Nmax - 8;
  if (Nmax) *Nmax = c->Nmax;
  return TOPAY_OK;
}

topay_status topay_get_batch(topay_ctx* c, int* success, double* cost, int* n_pieces) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  if (success) HIPCHK(memcpy_sync(c, success, c->success.p, (size_t)c->B * 4, hipMemcpyDeviceToHost));
  if (cost) HIPCHK(memcpy_sync(c, cost, c->cost.p, (size_t)c->B * 8, hipMemcpyDeviceToHost));
  if (n_pieces) memcpy(n_pieces, c->hN.data(), (size_t)c->B * 4);
  return TOPAY_OK;
}

topay_status topay_get_elapsed_us(topay_ctx* c, double* us, double* start_us, int* hw_id) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  if (us) HIPCHK(memcpy_sync(c, us, c->elapsed.p, (size_t)c->B * 8, hipMemcpyDeviceToHost));
  if (start_us) HIPCHK(memcpy_sync(c, start_us, c->startus.p, (size_t)c->B * 8, hipMemcpyDeviceToHost));
  if (hw_id) HIPCHK(memcpy_sync(c, hw_id, c->hwid.p, (size_t)c->B * 4, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

// MomaTraj playback of candidate i: car_seq (x, y, theta, t every 0.1 s; moma_traj_opt.h:40-69) and getState at the
// given times (113-137).  seq may be NULL; *n_seq receives the number of entries (capacity seq_cap rows of 4).
topay_status topay_playback(topay_ctx* c, int i, int n_times, const double* times, double* states, int seq_cap, double* seq,
                            int* n_seq) {
  if (!c || !c->have_traj || !c->solved) return TOPAY_ERR_NO_TRAJ;
  if (i < 0 || i >= c->B || n_times < 0 || (n_times > 0 && (!times || !states))) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  std::vector<double> hT((size_t)std::max(1, c->hN[i]));
  if (c->hN[i] > 0) HIPCHK(memcpy_sync(c, hT.data(), c->T.as<double>() + c->h_poff[i], (size_t)c->hN[i] * 8, hipMemcpyDeviceToHost));
  double t = 0.0;
  for (int k = 0; k < c->hN[i]; k++) t += hT[k];
  if (!(t > 0.0 && t < 1.0e4)) t = 0.0;
  const long long cap_panels = (long long)(t / 0.025) + 4;
  const long long nseq_max = cap_panels / 4 + 2;
  topay_status s;
  if ((s = c->feas_cseq.ensure((size_t)2 * (cap_panels + 1) * 8)) != TOPAY_OK) return s;
  if ((s = c->pb_io.ensure((size_t)(n_times * 11 + nseq_max * 4 + 2) * 8)) != TOPAY_OK) return s;
  double* d_times = c->pb_io.as<double>();
  double* d_states = d_times + n_times;
  double* d_seq = d_states + (size_t)n_times * 10;
  int* d_nseq = (int*)(d_seq + nseq_max * 4);
  if (n_times) HIPCHK(hipMemcpyAsync(d_times, times, (size_t)n_times * 8, hipMemcpyHostToDevice, c->stream));
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  hipLaunchKernelGGL(k_playback, dim3(1), dim3(64), 0, c->stream, c->db, i, c->feas_cseq.as<double>(), cap_panels, n_times,
                     (const double*)d_times, d_states, d_seq, d_nseq);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  int ns = 0;
  HIPCHK(memcpy_sync(c, &ns, d_nseq, 4, hipMemcpyDeviceToHost));
  if (n_seq) *n_seq = ns;
  if (seq && ns > 0) HIPCHK(memcpy_sync(c, seq, d_seq, (size_t)std::min(ns, seq_cap) * 4 * 8, hipMemcpyDeviceToHost));
  if (n_times) HIPCHK(memcpy_sync(c, states, d_states, (size_t)n_times * 10 * 8, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

topay_status topay_default_mesh_params(topay_mesh_params_t* p) {
  if (!p) return TOPAY_ERR_INVALID_ARG;
  const double ll[7] = {0.2405, 0.0, 0.256, 0.0, 0.21, 0.0, 0.144};              // moma_param.h:114
  const double lo[7] = {-3.1, -2.26, -3.1, -2.355, -3.1, -2.23, -6.28};          // moma_param.h:115
  for (int i = 0; i < 7; i++) {
    p->link_length[i] = ll[i];
    p->joint_pos_limit_min[i] = lo[i];
    for (int k = 0; k < 3; k++) { p->joint_offset[3 * i + k] = 0.0; p->joint_dof_axis[3 * i + k] = 0.0; }
    if (i < 6) {                                                                  // moma_param.h:77-90
      p->joint_offset[3 * i] = (i % 2 == 0) ? -1.5708 : 1.5708;
      p->joint_dof_axis[3 * i + 1] = (i % 2 == 0) ? -1.0 : 1.0;
    } else {
      p->joint_dof_axis[3 * i + 2] = 1.0;
    }
  }
  return TOPAY_OK;
}

topay_status topay_mesh_poses(topay_ctx* c, const topay_mesh_params_t* mesh, int n, const double* states, double* parts) {
  if (!c || !mesh || n < 0 || (n > 0 && (!states || !parts))) return TOPAY_ERR_INVALID_ARG;
  if (n == 0) return TOPAY_OK;
  HIPCHK(hipSetDevice(c->device));
  topay_status s;
  if ((s = c->pb_io.ensure((size_t)n * (10 + 77) * 8)) != TOPAY_OK) return s;
  double* d_st = c->pb_io.as<double>();
  double* d_parts = d_st + (size_t)n * 10;
  HIPCHK(hipMemcpyAsync(d_st, states, (size_t)n * 80, hipMemcpyHostToDevice, c->stream));
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  hipLaunchKernelGGL(k_mesh_pose, dim3((n + 63) / 64), dim3(64), 0, c->stream, *mesh, n, (const double*)d_st, d_parts);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(parts, d_parts, (size_t)n * 77 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return TOPAY_OK;
}

// Planner::toMeshMsg (planner.cpp:2003-2056).  The sample times and the arc length are running sums over the samples
// (host, in the reference's order); getState and getMeshPose of all samples run on the device.
topay_status topay_mesh_traj(topay_ctx* c, int i, const topay_mesh_params_t* mesh, int res, int cap_states, double* parts,
                             double* yaws, double* arc_lengths, int* n_states) {
  if (!c || !c->have_traj || !c->solved) return TOPAY_ERR_NO_TRAJ;
  if (i < 0 || i >= c->B || !mesh || res <= 0 || cap_states < res + 1 || !parts || !yaws || !arc_lengths || !n_states) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  std::vector<double> hT((size_t)std::max(1, c->hN[i]));
  if (c->hN[i] > 0) HIPCHK(memcpy_sync(c, hT.data(), c->T.as<double>() + c->h_poff[i], (size_t)c->hN[i] * 8, hipMemcpyDeviceToHost));
  double T = 0.0;
  for (int k = 0; k < c->hN[i]; k++) T += hT[k];
  if (!(T > 0.0 && T < 1.0e4)) { *n_states = 0; return TOPAY_OK; }
  const double intvl = T / res;
  std::vector<double> times;
  for (double t = 0.0; t < T && (int)times.size() < cap_states; t += intvl) times.push_back(t);
  const int n = (int)times.size();
  std::vector<double> st((size_t)(n + 1) * 10);
  times.push_back(0.0);                       // prev_state of the first sample = getState(0)
  topay_status s = topay_playback(c, i, n + 1, times.data(), st.data(), 0, nullptr, nullptr);
  if (s != TOPAY_OK) return s;
  if ((s = topay_mesh_poses(c, mesh, n, st.data(), parts)) != TOPAY_OK) return s;
  double acc = 0.0;
  const double* prev = &st[(size_t)n * 10];
  for (int k = 0; k < n; k++) {
    const double* cur = &st[(size_t)k * 10];
    const double dx = cur[0] - prev[0], dy = cur[1] - prev[1];
    acc += std::sqrt(dx * dx + dy * dy);
    arc_lengths[k] = acc;
    yaws[k] = cur[2];
    prev = cur;
  }
  *n_states = n;
  return TOPAY_OK;
}

// GridMap::isWholeBodyCollision (grid_map.h:613-650) of n states (x, y, theta, q1..q7) against map slot map_id:
// collide[i] = 1 when the state violates a joint limit, leaves the map or collides (front-end building block).
topay_status topay_whole_body_collision(topay_ctx* c, int map_id, int n, const double* states, int* collide) {
  if (!c || !states || !collide || n < 0 || map_id < 0 || map_id >= TOPAY_MAX_MAPS) return TOPAY_ERR_INVALID_ARG;
  if (!c->have_map[map_id]) return TOPAY_ERR_NO_MAP;
  if (n == 0) return TOPAY_OK;
  HIPCHK(hipSetDevice(c->device));
  topay_status s;
  if ((s = c->pb_io.ensure((size_t)n * 10 * 8 + (size_t)n * 4)) != TOPAY_OK) return s;
  double* d_st = c->pb_io.as<double>();
  int* d_out = (int*)(d_st + (size_t)n * 10);
  HIPCHK(hipMemcpyAsync(d_st, states, (size_t)n * 80, hipMemcpyHostToDevice, c->stream));
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  hipLaunchKernelGGL(k_whole_body, dim3((n + 63) / 64), dim3(64), 0, c->stream, (const DevMap*)c->dmaps.p, map_id, n,
                     (const double*)d_st, d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(collide, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return TOPAY_OK;
}

// GraphSearch::getDensePath (graph_search.cpp:119-176) for n_paths raw 2-D paths at once.
topay_status topay_dense_path(topay_ctx* c, int n_paths, const int* raw_len, const double* raw_xy, double step_size, const double* start_yaw,
                              const double* end_yaw, double v_max, double w_max, int cap_per_path, int* out_len, double* out) {
  if (!c || n_paths <= 0 || !raw_len || !raw_xy || !start_yaw || !end_yaw || !out_len || !out || cap_per_path <= 0 || !(step_size > 0.0))
    return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  std::vector<long long> off((size_t)n_paths + 1, 0);
  for (int p = 0; p < n_paths; p++) {
    if (raw_len[p] < 1) return TOPAY_ERR_INVALID_ARG;
    off[p + 1] = off[p] + raw_len[p];
  }
  const size_t tot = (size_t)off[n_paths];
  DevBuf d_raw, d_off, d_len, d_yaw, d_out, d_olen;
  topay_status s;
  if ((s = d_raw.ensure(tot * 16)) != TOPAY_OK || (s = d_off.ensure(((size_t)n_paths + 1) * 8)) != TOPAY_OK ||
      (s = d_len.ensure((size_t)n_paths * 4)) != TOPAY_OK || (s = d_yaw.ensure((size_t)n_paths * 16)) != TOPAY_OK ||
      (s = d_out.ensure((size_t)n_paths * cap_per_path * 32)) != TOPAY_OK || (s = d_olen.ensure((size_t)n_paths * 4)) != TOPAY_OK)
    return s;
  HIPCHK(hipMemcpyAsync(d_raw.p, raw_xy, tot * 16, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_off.p, off.data(), ((size_t)n_paths + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_len.p, raw_len, (size_t)n_paths * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_yaw.p, start_yaw, (size_t)n_paths * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_yaw.as<double>() + n_paths, end_yaw, (size_t)n_paths * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_dense_path, dim3((n_paths + 63) / 64), dim3(64), 0, c->stream, n_paths, (const double*)d_raw.as<double>(),
                     (const long long*)d_off.as<long long>(), (const int*)d_len.as<int>(), step_size, (const double*)d_yaw.as<double>(),
                     (const double*)(d_yaw.as<double>() + n_paths), v_max, w_max, cap_per_path, d_out.as<double>(), d_olen.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_len, d_olen.p, (size_t)n_paths * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(out, d_out.p, (size_t)n_paths * cap_per_path * 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  DevBuf* bufs[] = {&d_raw, &d_off, &d_len, &d_yaw, &d_out, &d_olen};
  for (DevBuf* b : bufs) b->release();
  return TOPAY_OK;
}

// MCRRTs::connectCollision (mcrrts.h:310-348): number of checks of every edge (lines 321-328; host arithmetic) ...
topay_status topay_connect_check_num(int n_edges, const double* rs_distance, const double* q_from, const double* q_to, double check_res,
                                     int* piece_num) {
  if (n_edges < 0 || !rs_distance || !q_from || !q_to || !piece_num || !(check_res > 0.0)) return TOPAY_ERR_INVALID_ARG;
  for (int e = 0; e < n_edges; e++) {
    const int check_num_car = (int)std::ceil(rs_distance[e] / check_res);
    double dmax = 0.0;
    for (int q = 0; q < 7; q++) dmax = std::max(dmax, std::fabs(q_to[7 * e + q] - q_from[7 * e + q]));
    const int check_num_theta = (int)std::ceil(dmax / check_res);
    piece_num[e] = std::max(std::max(check_num_car, check_num_theta), 3);
  }
  return TOPAY_OK;
}

// ... and the checks themselves (lines 330-345), every interpolated state of every edge in one launch.
topay_status topay_connect_collision(topay_ctx* c, int map_id, int n_edges, const int* piece_num, const double* car_poses, const double* q_from,
                                     const double* q_to, int* collide) {
  if (!c || n_edges < 0 || map_id < 0 || map_id >= TOPAY_MAX_MAPS || (n_edges > 0 && (!piece_num || !car_poses || !q_from || !q_to || !collide)))
    return TOPAY_ERR_INVALID_ARG;
  if (!c->have_map[map_id]) return TOPAY_ERR_NO_MAP;
  if (n_edges == 0) return TOPAY_OK;
  HIPCHK(hipSetDevice(c->device));
  std::vector<int> edge_of, idx;
  for (int e = 0; e < n_edges; e++) {
    if (piece_num[e] <= 0) return TOPAY_ERR_INVALID_ARG;
    for (int i = 0; i < piece_num[e]; i++) { edge_of.push_back(e); idx.push_back(i); }
  }
  const size_t nc = edge_of.size();
  DevBuf d_i, d_d;
  topay_status s;
  if ((s = d_i.ensure((2 * nc + 2 * (size_t)n_edges) * 4)) != TOPAY_OK || (s = d_d.ensure((3 * nc + 14 * (size_t)n_edges) * 8)) != TOPAY_OK) return s;
  int* d_edge = d_i.as<int>();
  int* d_idx = d_edge + nc;
  int* d_pn = d_idx + nc;
  int* d_col = d_pn + n_edges;
  double* d_car = d_d.as<double>();
  double* d_qf = d_car + 3 * nc;
  double* d_qt = d_qf + 7 * (size_t)n_edges;
  HIPCHK(hipMemcpyAsync(d_edge, edge_of.data(), nc * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_idx, idx.data(), nc * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_pn, piece_num, (size_t)n_edges * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(d_col, 0, (size_t)n_edges * 4, c->stream));
  HIPCHK(hipMemcpyAsync(d_car, car_poses, 3 * nc * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_qf, q_from, 7 * (size_t)n_edges * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_qt, q_to, 7 * (size_t)n_edges * 8, hipMemcpyHostToDevice, c->stream));
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  hipLaunchKernelGGL(k_connect, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, c->stream, (const DevMap*)c->dmaps.p, map_id, (long long)nc,
                     (const int*)d_edge, (const int*)d_idx, (const int*)d_pn, (const double*)d_car, (const double*)d_qf, (const double*)d_qt, d_col);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(collide, d_col, (size_t)n_edges * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  d_i.release();
  d_d.release();
  return TOPAY_OK;
}

topay_status topay_params_from_yaml(const char* path_or_text, topay_params_t* params, char* ignored, int ignored_cap) {
  if (!path_or_text || !params) return TOPAY_ERR_INVALID_ARG;
  std::string ign, err;
  const topay_status s = topay_yaml::apply_source(path_or_text, params, ign, err);
  if (s != TOPAY_OK) { set_err("topay_params_from_yaml: " + err); return s; }
  if (ignored && ignored_cap > 0) {
    strncpy(ignored, ign.c_str(), (size_t)ignored_cap - 1);
    ignored[ignored_cap - 1] = 0;
  }
  return TOPAY_OK;
}

topay_status topay_plan2d_jps(topay_ctx* c, int n, const int* map_ids, const double* start_xy, const double* end_xy, double threshold,
                              int cap_points, int* out_len, double* out_xy, int* stats) {
  if (!c || n < 0 || cap_points < 2 || (n > 0 && (!start_xy || !end_xy || !out_len || !out_xy))) return TOPAY_ERR_INVALID_ARG;
  if (n == 0) return TOPAY_OK;
  std::vector<int> mid((size_t)n, 0);
  long long ncell = 0;
  for (int p = 0; p < n; p++) {
    mid[p] = map_ids ? map_ids[p] : 0;
    if (mid[p] < 0 || mid[p] >= TOPAY_MAX_MAPS) return TOPAY_ERR_INVALID_ARG;
    if (!c->have_map[mid[p]]) return TOPAY_ERR_NO_MAP;
    ncell = std::max(ncell, (long long)c->hmaps[mid[p]].dims[0] * c->hmaps[mid[p]].dims[1]);
  }
  HIPCHK(hipSetDevice(c->device));
  // search state per instance: g (8) + parent, heap position, heap (3 x 4) + flags (1) bytes per cell; searches run in
  // chunks of at most 2 GB of it
  const size_t per = (size_t)ncell * 21;
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)2 << 30) / std::max<size_t>(per, 1)));
  ScopedDevBuf ws, io;
  topay_status s;
  if ((s = ws.ensure((size_t)chunk * per + 64)) != TOPAY_OK) return s;
  const size_t io_d = (size_t)n * 4 + (size_t)n * cap_points * 2, io_i = (size_t)n * 4;
  if ((s = io.ensure(io_d * 8 + io_i * 4)) != TOPAY_OK) { ws.release(); return s; }
  double* d_start = io.as<double>();
  double* d_end = d_start + 2 * (size_t)n;
  double* d_out = d_end + 2 * (size_t)n;
  int* d_mid = (int*)(d_out + (size_t)n * cap_points * 2);
  int* d_len = d_mid + n;
  int* d_stats = d_len + n;
  HIPCHK(hipMemcpyAsync(d_start, start_xy, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_end, end_xy, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_mid, mid.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  topay::JpsBatch B;
  B.cap = cap_points; B.ncell_max = ncell; B.map_id = d_mid; B.start = d_start; B.end = d_end; B.threshold = threshold;
  B.g = ws.as<double>();
  B.parent = (int*)(B.g + (size_t)chunk * ncell);
  B.hpos = B.parent + (size_t)chunk * ncell;
  B.heap = B.hpos + (size_t)chunk * ncell;
  B.flag = (unsigned char*)(B.heap + (size_t)chunk * ncell);
  B.out_len = d_len; B.out_xy = d_out; B.stats = d_stats;
  for (int i0 = 0; i0 < n; i0 += chunk) {
    B.inst0 = i0;
    B.n = std::min(chunk, n - i0);
    HIPCHK(hipMemsetAsync(B.flag, 0, (size_t)B.n * ncell, c->stream));
    hipLaunchKernelGGL(topay::k_jps, dim3((unsigned)B.n), dim3(64), 0, c->stream, (const DevMap*)c->dmaps.p, B);   // one wave per search
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipMemcpyAsync(out_len, d_len, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(out_xy, d_out, (size_t)n * cap_points * 16, hipMemcpyDeviceToHost, c->stream));
  if (stats) HIPCHK(hipMemcpyAsync(stats, d_stats, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  ws.release();
  io.release();
  return TOPAY_OK;
}

void topay_mcrrt_default_params(topay_mcrrt_params_t* p) {
  if (!p) return;
  p->goal_sample_rate = 0.4;      // planner/params/mcrrts.yaml
  p->check_colli_res = 0.01;
  p->rs_turning_radius = 1.0e-2;  // mcrrts.h:134
  p->max_iter = 1000;
  p->max_sample_tries = 64;
  p->node_cap = 2048;
  p->reserved = 0;
  p->seed = 42;
}

topay_status topay_mcrrt_plan(topay_ctx* c, int n, const int* map_ids, const int* path_len, const double* car_paths, const double* start,
                              const double* end, const topay_mcrrt_params_t* prm, unsigned long long first_instance, int cap_per_path,
                              int* wb_len, double* wb_path, int* stats, double* c_max) {
  if (!c || n < 0 || cap_per_path < 2 || (n > 0 && (!path_len || !car_paths || !start || !end || !wb_len || !wb_path))) return TOPAY_ERR_INVALID_ARG;
  topay_mcrrt_params_t P;
  if (prm) P = *prm;
  else topay_mcrrt_default_params(&P);
  if (P.max_iter < 0 || P.max_sample_tries < 1 || P.node_cap < 2 || !(P.check_colli_res > 0.0) || !(P.rs_turning_radius > 0.0)) return TOPAY_ERR_INVALID_ARG;
  if (n == 0) return TOPAY_OK;
  std::vector<long long> off((size_t)n + 1, 0);
  std::vector<int> mid((size_t)n, 0);
  for (int p = 0; p < n; p++) {
    if (path_len[p] < 2 || path_len[p] > cap_per_path || path_len[p] > 255) {
      set_err("topay_mcrrt_plan: chassis path " + std::to_string(p) + " has " + std::to_string(path_len[p]) + " layers (2.." +
              std::to_string(std::min(cap_per_path, 255)) + " supported: the reference's node key holds the layer in one character)");
      return TOPAY_ERR_INVALID_ARG;
    }
    off[p + 1] = off[p] + path_len[p];
    mid[p] = map_ids ? map_ids[p] : 0;
    if (mid[p] < 0 || mid[p] >= TOPAY_MAX_MAPS) return TOPAY_ERR_INVALID_ARG;
    if (!c->have_map[mid[p]]) return TOPAY_ERR_NO_MAP;
  }
  HIPCHK(hipSetDevice(c->device));
  const size_t nn = (size_t)n * P.node_cap, tot = (size_t)off[n];
  topay_status s;
  // inputs: map ids, lengths (int), offsets (i64), chassis paths, start, end (f64); outputs: wb_len, stats (int), wb, c_max (f64)
  const size_t in_i = (size_t)n * 2 + (size_t)n * 9, in_l = (size_t)n + 1, in_d = 4 * tot + 20 * (size_t)n + (size_t)n * cap_per_path * 10 + n;
  if ((s = c->mc_in.ensure(in_l * 8 + in_d * 8 + in_i * 4)) != TOPAY_OK || (s = c->mc_i.ensure(nn * 5 * 4)) != TOPAY_OK ||
      (s = c->mc_d.ensure(nn * 8 * 8)) != TOPAY_OK || (s = c->mc_k.ensure(nn * TOPAY_MC_KEYW * 8)) != TOPAY_OK ||
      (s = c->mc_rs.ensure((size_t)n * 2 * cap_per_path * sizeof(topay::RsPath))) != TOPAY_OK)
    return s;
  long long* d_off = c->mc_in.as<long long>();
  double* d_car = (double*)(d_off + in_l);
  double* d_start = d_car + 4 * tot;
  double* d_end = d_start + 10 * (size_t)n;
  double* d_wb = d_end + 10 * (size_t)n;
  double* d_cmax = d_wb + (size_t)n * cap_per_path * 10;
  int* d_mid = (int*)(d_cmax + n);
  int* d_len = d_mid + n;
  int* d_wlen = d_len + n;
  int* d_stats = d_wlen + n;
  HIPCHK(hipMemcpyAsync(d_off, off.data(), in_l * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_car, car_paths, 4 * tot * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_start, start, 10 * (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_end, end, 10 * (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_mid, mid.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_len, path_len, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemsetAsync(d_wb, 0, ((size_t)n * cap_per_path * 10 + n) * 8, c->stream));
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  topay::McrrtBatch B;
  B.n = n; B.layer_cap = cap_per_path; B.inst_base = first_instance;
  B.map_id = d_mid; B.car_off = d_off; B.car_len = d_len; B.car = d_car; B.start = d_start; B.end = d_end;
  B.P.goal_sample_rate = P.goal_sample_rate; B.P.check_colli_res = P.check_colli_res; B.P.rs_rho = P.rs_turning_radius;
  B.P.max_iter = P.max_iter; B.P.max_sample_tries = P.max_sample_tries; B.P.node_cap = P.node_cap; B.P.reserved = 0; B.P.seed = P.seed;
  int* ni = c->mc_i.as<int>();
  B.nd_layer = ni; B.nd_state = ni + nn; B.nd_parent = ni + 2 * nn; B.nd_nchild = ni + 3 * nn; B.nd_mark = ni + 4 * nn;
  B.nd_cost = c->mc_d.as<double>(); B.nd_q = B.nd_cost + nn; B.nd_key = c->mc_k.as<unsigned long long>();
  B.rs = (topay::RsPath*)c->mc_rs.p;
  B.wb_len = d_wlen; B.wb = d_wb; B.stats = d_stats; B.cmax = d_cmax;
  hipLaunchKernelGGL(topay::k_mcrrt, dim3((unsigned)n), dim3(64), 0, c->stream, (const DevMap*)c->dmaps.p, B);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(wb_len, d_wlen, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(wb_path, d_wb, (size_t)n * cap_per_path * 80, hipMemcpyDeviceToHost, c->stream));
  if (stats) HIPCHK(hipMemcpyAsync(stats, d_stats, (size_t)n * 32, hipMemcpyDeviceToHost, c->stream));
  if (c_max) HIPCHK(hipMemcpyAsync(c_max, d_cmax, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->mc_n = n;
  c->mc_node_cap = P.node_cap;
  return TOPAY_OK;
}

topay_status topay_mcrrt_nodes(topay_ctx* c, int instance, int cap, int* layer, int* state, int* parent, double* cost, double* q) {
  if (!c || instance < 0 || instance >= c->mc_n || cap < 0) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  const size_t nn = (size_t)c->mc_n * c->mc_node_cap, o = (size_t)instance * c->mc_node_cap;
  const size_t m = (size_t)std::min(cap, c->mc_node_cap);
  const int* ni = c->mc_i.as<int>();
  const double* nd = c->mc_d.as<double>();
  if (layer) HIPCHK(memcpy_sync(c, layer, ni + o, m * 4, hipMemcpyDeviceToHost));
  if (state) HIPCHK(memcpy_sync(c, state, ni + nn + o, m * 4, hipMemcpyDeviceToHost));
  if (parent) HIPCHK(memcpy_sync(c, parent, ni + 2 * nn + o, m * 4, hipMemcpyDeviceToHost));
  if (cost) HIPCHK(memcpy_sync(c, cost, nd + o, m * 8, hipMemcpyDeviceToHost));
  if (q) HIPCHK(memcpy_sync(c, q, nd + nn + 7 * o, m * 56, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

// ompl::base::ReedsSheppStateSpace(rho): distance and interpolate as the search uses them, for n pose pairs (one thread each)
topay_status topay_reeds_shepp(topay_ctx* c, int n, const double* from, const double* to, const double* t, double rho, double* distance,
                               int* word, double* lengths, double* pose) {
  if (!c || n < 0 || !(rho > 0.0) || (n > 0 && (!from || !to))) return TOPAY_ERR_INVALID_ARG;
  if (n == 0) return TOPAY_OK;
  HIPCHK(hipSetDevice(c->device));
  ScopedDevBuf d;
  topay_status s;
  if ((s = d.ensure((size_t)n * (3 + 3 + 1 + 1 + 5 + 3 + 1) * 8)) != TOPAY_OK) return s;
  double* d_from = d.as<double>();
  double* d_to = d_from + 3 * (size_t)n;
  double* d_t = d_to + 3 * (size_t)n;
  double* d_dist = d_t + n;
  double* d_len = d_dist + n;
  double* d_pose = d_len + 5 * (size_t)n;
  int* d_word = (int*)(d_pose + 3 * (size_t)n);
  HIPCHK(memcpy_sync(c, d_from, from, (size_t)n * 24, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, d_to, to, (size_t)n * 24, hipMemcpyHostToDevice));
  if (t) HIPCHK(memcpy_sync(c, d_t, t, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(topay::k_reeds_shepp, dim3((n + 63) / 64), dim3(64), 0, c->stream, n, (const double*)d_from, (const double*)d_to,
                     t ? (const double*)d_t : (const double*)nullptr, rho, d_dist, d_word, d_len, d_pose);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  if (distance) HIPCHK(memcpy_sync(c, distance, d_dist, (size_t)n * 8, hipMemcpyDeviceToHost));
  if (word) HIPCHK(memcpy_sync(c, word, d_word, (size_t)n * 4, hipMemcpyDeviceToHost));
  if (lengths) HIPCHK(memcpy_sync(c, lengths, d_len, (size_t)n * 40, hipMemcpyDeviceToHost));
  if (pose && t) HIPCHK(memcpy_sync(c, pose, d_pose, (size_t)n * 24, hipMemcpyDeviceToHost));
  d.release();
  return TOPAY_OK;
}

topay_status topay_get_total_durations(topay_ctx* c, double* total) {
  if (!c || !c->have_traj || !total) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  std::vector<double> hT((size_t)c->h_poff[c->B] + 1);
  HIPCHK(memcpy_sync(c, hT.data(), c->T.p, (size_t)c->h_poff[c->B] * 8, hipMemcpyDeviceToHost));
  for (int b = 0; b < c->B; b++) {
    double t = 0.0;
    for (int i = 0; i < c->hN[b]; i++) t += hT[(size_t)c->h_poff[b] + i];  // PolyTrajectory::getTotalDuration, minco.hpp:304-313
    total[b] = c->hN[b] > 0 ? t : 0.0 / 0.0;
  }
  return TOPAY_OK;
}

topay_status topay_get_alm(topay_ctx* c, double* alm) {
  if (!c || !c->have_traj || !alm) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(memcpy_sync(c, alm, c->alm.p, (size_t)c->B * 32, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

topay_status topay_get_stats(topay_ctx* c, int* stats) {
  if (!c || !c->have_traj || !stats) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(memcpy_sync(c, stats, c->stats.p, (size_t)c->B * 32, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

topay_status topay_get_result(topay_ctx* c, int i, int* success, double* cost, int* n_pieces, double* durations,
                              double* coeffs, double* knots_xy) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (i < 0 || i >= c->B) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }
  const int N = c->hN[i], rows = 6 * N;
  if (N == 0) {  // not representable (more than TOPAY_MAX_N pieces): failed candidate, nothing else to report
    if (success) *success = 0;
    if (cost) *cost = 0.0 / 0.0;
    if (n_pieces) *n_pieces = 0;
    return TOPAY_OK;
  }
  if (success) HIPCHK(memcpy_sync(c, success, c->success.as<int>() + i, 4, hipMemcpyDeviceToHost));
  if (cost) HIPCHK(memcpy_sync(c, cost, c->cost.as<double>() + i, 8, hipMemcpyDeviceToHost));
  if (n_pieces) *n_pieces = N;
  if (durations) HIPCHK(memcpy_sync(c, durations, c->T.as<double>() + c->h_poff[i], (size_t)N * 8, hipMemcpyDeviceToHost));
  if (coeffs) {
    std::vector<double> cm((size_t)9 * rows);
    HIPCHK(memcpy_sync(c, cm.data(), c->coef.as<double>() + 54 * c->h_poff[i], cm.size() * 8, hipMemcpyDeviceToHost));
    // getTraj(): per piece the 6x9 block transposed, highest order first — minco.hpp:908-921
    for (int p = 0; p < N; p++)
      for (int d = 0; d < 9; d++)
        for (int k = 0; k < 6; k++) coeffs[((size_t)p * 9 + d) * 6 + k] = cm[(size_t)d * rows + 6 * p + 5 - k];
  }
  if (knots_xy)
    HIPCHK(memcpy_sync(c, knots_xy, c->knots.as<double>() + 2 * (c->h_poff[i] + i), (size_t)2 * (N + 1) * 8, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

topay_status topay_get_results(topay_ctx* c, int n, const int* idx, int cap_pieces, int* piece_off, double* durations,
                               double* coeffs, double* knots_xy) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (n < 0 || (n > 0 && (!idx || !piece_off))) return TOPAY_ERR_INVALID_ARG;
  if (n == 0) return TOPAY_OK;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }
  if (!c->solved) { set_err("topay_get_results: the batch has not been optimised"); return TOPAY_ERR_NO_TRAJ; }
  std::vector<int> off((size_t)n + 1, 0);
  for (int k = 0; k < n; k++) {
    if (idx[k] < 0 || idx[k] >= c->B) return TOPAY_ERR_INVALID_ARG;
    off[k + 1] = off[k] + c->hN[idx[k]];
  }
  const int np = off[n];
  memcpy(piece_off, off.data(), ((size_t)n + 1) * sizeof(int));
  if (np > cap_pieces) { set_err("topay_get_results: cap_pieces too small for the selection"); return TOPAY_ERR_INVALID_ARG; }
  if (np == 0 || (!durations && !coeffs && !knots_xy)) return TOPAY_OK;
  // device staging: idx | piece_off | durations | coeffs | knots, one kernel, one copy back
  const size_t ints = (size_t)2 * n + 1, dbl = (size_t)np + (size_t)np * 54 + (size_t)2 * (np + n);
  topay_status s;
  if ((s = c->pb_io.ensure(ints * 4 + 8 + dbl * 8)) != TOPAY_OK) return s;
  int* d_idx = c->pb_io.as<int>();
  int* d_off = d_idx + n;
  double* d_dur = (double*)(((uintptr_t)(d_off + n + 1) + 7) & ~(uintptr_t)7);
  double* d_coef = d_dur + np;
  double* d_kn = d_coef + (size_t)np * 54;
  HIPCHK(hipMemcpyAsync(d_idx, idx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(d_off, off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  // a selected candidate that was never launched (zero pieces) still owns one knot pair of the packed output: zeros
  HIPCHK(hipMemsetAsync(d_kn, 0, (size_t)2 * (np + n) * 8, c->stream));
  hipLaunchKernelGGL(k_gather_results, dim3(n), dim3(64), 0, c->stream, c->db, n, (const int*)d_idx, (const int*)d_off, d_dur,
                     d_coef, d_kn);
  HIPCHK(hipGetLastError());
  std::vector<double> host(dbl);
  HIPCHK(memcpy_sync(c, host.data(), d_dur, dbl * 8, hipMemcpyDeviceToHost));
  if (durations) memcpy(durations, host.data(), (size_t)np * 8);
  if (coeffs) memcpy(coeffs, host.data() + np, (size_t)np * 54 * 8);
  if (knots_xy) memcpy(knots_xy, host.data() + np + (size_t)np * 54, (size_t)2 * (np + n) * 8);
  return TOPAY_OK;
}

topay_status topay_get_polytraj_msg(topay_ctx* c, int i, int cap_pieces, unsigned char* order, float* coeff, float* durations,
                                    signed char* directions, int* n_pieces) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (i < 0 || i >= c->B) return TOPAY_ERR_INVALID_ARG;
  const int N = c->hN[i];
  if (n_pieces) *n_pieces = N;
  if (order) *order = 5;
  if (N == 0) return TOPAY_OK;
  if (N > cap_pieces) return TOPAY_ERR_INVALID_ARG;
  std::vector<double> dur((size_t)N), cf((size_t)N * 54);
  topay_status s = topay_get_result(c, i, nullptr, nullptr, nullptr, dur.data(), cf.data(), nullptr);
  if (s != TOPAY_OK) return s;
  for (int p = 0; p < N; p++) {
    if (durations) durations[p] = (float)dur[p];
    if (coeff)
      for (int t = 0; t < 54; t++) coeff[(size_t)p * 54 + t] = (float)cf[(size_t)p * 54 + t];
    if (directions) {
      // arc-length rate (dimension 1) at the middle of the piece; coefficients are highest order first
      const double* a = &cf[(size_t)p * 54 + 6], t = 0.5 * dur[p];
      const double sd = ((((5.0 * a[0]) * t + 4.0 * a[1]) * t + 3.0 * a[2]) * t + 2.0 * a[3]) * t + a[4];
      directions[p] = sd < 0.0 ? -1 : 1;
    }
  }
  return TOPAY_OK;
}

topay_status topay_get_x(topay_ctx* c, int i, int* n, double* x) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (i < 0 || i >= c->B) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->hN[i] == 0) { if (n) *n = 0; return TOPAY_ERR_TOO_MANY_PIECES; }
  const int nn = 10 * c->hN[i] - 8;
  if (n) *n = nn;
  if (x) {
    // after optimize: the final iterate; before: the packed initial guess
    if (c->solved) HIPCHK(memcpy_sync(c, x, c->x.as<double>() + c->h_noff[i], (size_t)nn * 8, hipMemcpyDeviceToHost));
    else HIPCHK(memcpy_sync(c, x, c->x0.as<double>() + (size_t)i * (10 * TOPAY_MAX_N - 8), (size_t)nn * 8, hipMemcpyDeviceToHost));
  }
  return TOPAY_OK;
}

// Kernel for a forced number of waves per trajectory (test hook topay_eval_waves): the smallest template that holds N.
static bool class_for_waves(int N, int nw, ClassDef& out) {
  static const ClassDef w1[] = {{10, 1, 1, nullptr, k_eval1, 2}, {21, 2, 1, nullptr, k_eval2, 2}, {32, 3, 1, nullptr, k_eval3, 2},
                                {42, 4, 1, nullptr, k_eval4, 2}, {64, 6, 1, nullptr, k_eval6, 2}};
  static const ClassDef w2[] = {{42, 2, 2, nullptr, k_eval2w2, 2}, {64, 3, 2, nullptr, k_eval3w2, 2}};
  static const ClassDef w4[] = {{85, 2, 4, nullptr, k_eval2w4, 2}, {128, 3, 4, nullptr, k_eval3w4, 2}, {TOPAY_MAX_N, 4, 4, nullptr, k_eval4w4, 2}};
  const ClassDef* t = nw == 1 ? w1 : (nw == 2 ? w2 : (nw == 4 ? w4 : nullptr));
  const int cnt = nw == 1 ? 5 : (nw == 4 ? 3 : 2);
  if (!t) return false;
  for (int k = 0; k < cnt; k++)
    if (N <= t[k].max_n) { out = t[k]; return true; }
  return false;
}

static topay_status eval_one(topay_ctx* c, int stage, int i, const double* x, const double* alm_lambda, const double* alm_rho,
                             double* f, double* g, double* final_xy_error, bool commit, int force_nw = 0) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (i < 0 || i >= c->B || (stage != 1 && stage != 2) || !x) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }   // inputs of a solve in flight stay untouched
  const int N = c->hN[i], nn = 10 * N - 8;
  if (N == 0) return TOPAY_ERR_TOO_MANY_PIECES;
  HIPCHK(memcpy_sync(c, c->x.as<double>() + c->h_noff[i], x, (size_t)nn * 8, hipMemcpyHostToDevice));
  double alm[4] = {alm_lambda ? alm_lambda[0] : c->hp.alm_init_lambda[0], alm_lambda ? alm_lambda[1] : c->hp.alm_init_lambda[1],
                   alm_rho ? alm_rho[0] : c->hp.alm_init_rho[0], alm_rho ? alm_rho[1] : c->hp.alm_init_rho[1]};
  HIPCHK(memcpy_sync(c, c->alm.as<double>() + (size_t)i * 4, alm, 32, hipMemcpyHostToDevice));
  // single-block launch through a one-entry order array placed at the end of the order buffer
  ScopedDevBuf tmp;
  topay_status s = tmp.ensure(4);
  if (s != TOPAY_OK) return s;
  HIPCHK(memcpy_sync(c, tmp.p, &i, 4, hipMemcpyHostToDevice));
  DevBatch d = c->db;
  d.order = tmp.as<int>();
  ClassDef cd = class_table()[bucket_of(N)];   // the class (kernel, waves per trajectory) that also solves this candidate
  if (force_nw > 0 && !class_for_waves(N, force_nw, cd)) { set_err("no kernel with that many waves holds this candidate"); return TOPAY_ERR_UNSUPPORTED; }
  const size_t lds = class_lds_bytes(cd, N);
  if ((s = push_params(c)) != TOPAY_OK) return s;
  HIPCHK(set_kernel_attributes(c->device));
  if (force_nw > 0) HIPCHK(hipFuncSetAttribute((const void*)cd.eval, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsDoublesPerCU * 8));
  hipLaunchKernelGGL(cd.eval, dim3(1), dim3(64 * cd.nw), lds, c->stream, d, (const DevMap*)c->dmaps.p, stage | (commit ? 16 : 0), 1, N);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  tmp.release();
  if (f) HIPCHK(memcpy_sync(c, f, c->fout.as<double>() + i, 8, hipMemcpyDeviceToHost));
  if (g) HIPCHK(memcpy_sync(c, g, c->work.as<double>() + 4 * c->h_noff[i], (size_t)nn * 8, hipMemcpyDeviceToHost));
  if (final_xy_error) HIPCHK(memcpy_sync(c, final_xy_error, c->xyerr.as<double>() + 2 * i, 16, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}
}  // extern "C"  (eval_one is internal)

extern "C" {
topay_status topay_eval(topay_ctx* c, int stage, int i, const double* x, const double* alm_lambda, const double* alm_rho,
                        double* f, double* g, double* final_xy_error) {
  return eval_one(c, stage, i, x, alm_lambda, alm_rho, f, g, final_xy_error, false);
}

// Test hook: the same evaluation by the kernel with `waves` wavefronts per trajectory (1, 2 or 4) instead of the
// candidate's class default.  An evaluation is order-identical whatever the number of waves (topay_eval_mw.h): the
// results must agree bit for bit.
topay_status topay_eval_waves(topay_ctx* c, int stage, int i, int waves, const double* x, const double* alm_lambda, const double* alm_rho,
                              double* f, double* g, double* final_xy_error) {
  if (waves != 1 && waves != 2 && waves != 4) return TOPAY_ERR_INVALID_ARG;
  return eval_one(c, stage, i, x, alm_lambda, alm_rho, f, g, final_xy_error, false, waves);
}

// The spline of a given decision vector as candidate i's result (MomaTrajOpt keeps the MINCO state of its last cost
// evaluation, moma_traj_opt.h:943-946: getTraj() after an evaluation at x returns exactly this): one stage-2
// evaluation at x with the given ALM state, after which getTraj / playback / gate / message entry points serve x's
// trajectory.  Replay and warm-start entry; the cost stored is the stage-2 cost at x.
topay_status topay_load_solution(topay_ctx* c, int i, const double* x, const double* alm_lambda, const double* alm_rho) {
  topay_status s = eval_one(c, 2, i, x, alm_lambda, alm_rho, nullptr, nullptr, nullptr, true);
  if (s == TOPAY_OK) {
    c->solved = true;
    c->gate_done = false;   // (the gate of a loaded trajectory: the separate kernel)
    const int zero = 0;     // the candidate has a trajectory now, whatever a solve before left in its flag
    HIPCHK(memcpy_sync(c, c->interrupted.as<int>() + i, &zero, 4, hipMemcpyHostToDevice));
  }
  return s;
}

// Batched hook: evaluate every candidate `repeats` times at its packed initial guess x0 (ALM state = initial).
topay_status topay_eval_batch(topay_ctx* c, int stage, int repeats, double* f) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if ((stage != 1 && stage != 2) || repeats == 0) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }   // inputs of a solve in flight stay untouched
  // x <- x0 (strided copy), alm <- init
  std::vector<double> x0((size_t)c->B * (10 * TOPAY_MAX_N - 8)), xs((size_t)c->h_noff[c->B] + 1, 0.0), alm((size_t)c->B * 4);
  HIPCHK(memcpy_sync(c, x0.data(), c->x0.p, x0.size() * 8, hipMemcpyDeviceToHost));
  for (int b = 0; b < c->B; b++) {
    if (c->hN[b] == 0) continue;
    const int nn = 10 * c->hN[b] - 8;
    memcpy(&xs[(size_t)c->h_noff[b]], &x0[(size_t)b * (10 * TOPAY_MAX_N - 8)], (size_t)nn * 8);
    alm[4 * b] = c->hp.alm_init_lambda[0]; alm[4 * b + 1] = c->hp.alm_init_lambda[1];
    alm[4 * b + 2] = c->hp.alm_init_rho[0]; alm[4 * b + 3] = c->hp.alm_init_rho[1];
  }
  HIPCHK(memcpy_sync(c, c->x.p, xs.data(), (size_t)c->h_noff[c->B] * 8, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, c->alm.p, alm.data(), alm.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipEventRecord(c->ev0, c->stream));
  topay_status s = launch_classes<true>(c, false, stage, repeats);
  if (s != TOPAY_OK) return s;
  HIPCHK(hipEventRecord(c->ev1, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->last_ms = ms;
  if (f) HIPCHK(memcpy_sync(c, f, c->fout.p, (size_t)c->B * 8, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

topay_status topay_check_feasible(topay_ctx* c, int* feasible) {
  return topay_feasibility_report(c, feasible, nullptr, nullptr);
}

topay_status topay_feasibility_report(topay_ctx* c, int* feasible, int* strict, double* report) {
  if (!c || !c->have_traj || !c->solved) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  const int B = c->B;
  if (c->gate_done) {   // the solving waves have gated their own trajectories: verdicts and extremes are resident
    std::vector<int> fl((size_t)B * 2);
    HIPCHK(memcpy_sync(c, fl.data(), c->feas_flags.p, fl.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++) {
      if (feasible) feasible[b] = fl[2 * b];
      if (strict) strict[b] = fl[2 * b + 1];
    }
    if (report) HIPCHK(memcpy_sync(c, report, c->feas_report.p, (size_t)B * 38 * 8, hipMemcpyDeviceToHost));
    return TOPAY_OK;
  }
  // scratch is sized from the longest returned trajectory
  std::vector<double> hT((size_t)c->h_poff[B] + 1);
  HIPCHK(memcpy_sync(c, hT.data(), c->T.p, (size_t)c->h_poff[B] * 8, hipMemcpyDeviceToHost));
  double tmax = 0.0;
  for (int b = 0; b < B; b++) {
    double t = 0.0;
    for (int i = 0; i < c->hN[b]; i++) t += hT[(size_t)c->h_poff[b] + i];
    if (t > 0.0 && t < 1.0e4 && t > tmax) tmax = t;
  }
  const long long cap_panels = (long long)(tmax / 0.025) + 4, cap_samples = (long long)(tmax / 0.01) + 16;
  topay_status s;
  if ((s = c->feas_cseq.ensure((size_t)B * 2 * (cap_panels + 1) * 8)) != TOPAY_OK) return s;
  if ((s = c->feas_tk.ensure((size_t)B * cap_samples * 8)) != TOPAY_OK) return s;
  if ((s = c->feas_report.ensure((size_t)B * 38 * 8)) != TOPAY_OK) return s;
  if ((s = c->feas_flags.ensure((size_t)B * 2 * 4)) != TOPAY_OK) return s;
  topay_status ps = push_params(c);
  if (ps != TOPAY_OK) return ps;
  hipLaunchKernelGGL(k_feasible, dim3(B), dim3(64), 0, c->stream, c->db, (const DevMap*)c->dmaps.p, c->feas_cseq.as<double>(),
                     c->feas_tk.as<double>(), cap_panels, cap_samples, c->feas_report.as<double>(), c->feas_flags.as<int>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<int> fl((size_t)B * 2), intr(B);
  HIPCHK(memcpy_sync(c, fl.data(), c->feas_flags.p, fl.size() * 4, hipMemcpyDeviceToHost));
  // an interrupted candidate has no trajectory (its result block holds the spline of the evaluation it was stopped in):
  // its verdicts stay 0 / 0, as the in-solve path and the cancellation post-pass of topay_synchronize write them
  HIPCHK(memcpy_sync(c, intr.data(), c->interrupted.p, (size_t)B * 4, hipMemcpyDeviceToHost));
  bool changed = false;
  for (int b = 0; b < B; b++)
    if (intr[b] && (fl[2 * b] || fl[2 * b + 1])) { fl[2 * b] = 0; fl[2 * b + 1] = 0; changed = true; }
  if (changed) HIPCHK(memcpy_sync(c, c->feas_flags.p, fl.data(), fl.size() * 4, hipMemcpyHostToDevice));
  c->gate_done = true;   // resident until the next solve, load or re-initialisation (each resets it)
  for (int b = 0; b < B; b++) {
    if (feasible) feasible[b] = fl[2 * b];
    if (strict) strict[b] = fl[2 * b + 1];
  }
  if (report) HIPCHK(memcpy_sync(c, report, c->feas_report.p, (size_t)B * 38 * 8, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

// Debug / parity tooling: record f of every evaluation of the next topay_optimize (cap per candidate; 0 = off).
topay_status topay_set_trace(topay_ctx* c, int cap) {
  if (!c || !c->have_traj || cap < 0) return TOPAY_ERR_NO_TRAJ;
  HIPCHK(hipSetDevice(c->device));
  c->trace_cap = cap;
  c->db.trace = nullptr;
  c->db.trace_cap = 0;
  if (cap > 0) {
    topay_status s = c->trace.ensure((size_t)c->B * cap * 8);
    if (s != TOPAY_OK) return s;
    HIPCHK(hipMemsetAsync(c->trace.p, 0, (size_t)c->B * cap * 8, c->stream));
    c->db.trace = c->trace.as<double>();
    c->db.trace_cap = cap;
  }
  return TOPAY_OK;
}
topay_status topay_get_trace(topay_ctx* c, int i, double* out) {
  if (!c || !c->have_traj || c->trace_cap <= 0 || i < 0 || i >= c->B) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(memcpy_sync(c, out, c->trace.as<double>() + (size_t)i * c->trace_cap, (size_t)c->trace_cap * 8, hipMemcpyDeviceToHost));
  return TOPAY_OK;
}

// Test hook: evaluate the deterministic sin/cos/atan2 (and an IEEE sqrt/div probe) on the device.
topay_status topay_test_math(topay_ctx* c, int n, const double* a, const double* b, double* out4n) {
  if (!c || n <= 0 || !a || !b || !out4n) return TOPAY_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->device));
  DevBuf da, dbb, dout;
  topay_status s;
  if ((s = da.ensure((size_t)n * 8)) != TOPAY_OK || (s = dbb.ensure((size_t)n * 8)) != TOPAY_OK ||
      (s = dout.ensure((size_t)n * 32)) != TOPAY_OK)
    return s;
  HIPCHK(memcpy_sync(c, da.p, a, (size_t)n * 8, hipMemcpyHostToDevice));
  HIPCHK(memcpy_sync(c, dbb.p, b, (size_t)n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_math, dim3((n + 63) / 64), dim3(64), 0, c->stream, da.as<double>(), dbb.as<double>(), dout.as<double>(), n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(memcpy_sync(c, out4n, dout.p, (size_t)n * 32, hipMemcpyDeviceToHost));
  da.release(); dbb.release(); dout.release();
  return TOPAY_OK;
}

// ---- the multi-GPU exchange behind the C-ABI ------------------------------------------------------------------
// Scenarios shard over the GPUs of a node, one process per GPU, and nothing of the solve is shared; the one exchange
// is the all-gather of a fixed-size record per scenario.  RCCL is bound at run time (dlopen: the
// library has no link-time dependency on it, and inside a process that already carries an RCCL -- torch's -- the same
// one is used); the collective runs on a stream of its own, so a solve in flight on the context is not waited for.
namespace {
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, topay_comm_id_t, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("TOPAY_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.h) break;
    }
    if (!api.h) return;
    api.GetUniqueId = (int (*)(void*))dlsym(api.h, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, topay_comm_id_t, int))dlsym(api.h, "ncclCommInitRank");
    api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.h, "ncclAllGather");
    api.CommDestroy = (int (*)(void*))dlsym(api.h, "ncclCommDestroy");
    api.GetErrorString = (const char* (*)(int))dlsym(api.h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) { dlclose(api.h); api.h = nullptr; }
  });
  return api.h ? &api : nullptr;
}
topay_status rccl_fail(const char* what, int rc) {
  RcclApi* a = rccl_api();
  set_err(std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(rc) : "RCCL error"));
  return TOPAY_ERR_NO_DEVICE;
}
}  // namespace

topay_status topay_comm_unique_id(topay_comm_id_t* id) {
  if (!id) return TOPAY_ERR_INVALID_ARG;
  RcclApi* a = rccl_api();
  if (!a) { set_err("librccl.so not found (TOPAY_RCCL_LIB names it explicitly)"); return TOPAY_ERR_UNSUPPORTED; }
  const int rc = a->GetUniqueId(id);
  return rc == 0 ? TOPAY_OK : rccl_fail("ncclGetUniqueId", rc);
}

topay_status topay_comm_init(topay_ctx* c, const topay_comm_id_t* id, int world, int rank) {
  if (!c || !id || world <= 0 || rank < 0 || rank >= world) return TOPAY_ERR_INVALID_ARG;
  RcclApi* a = rccl_api();
  if (!a) { set_err("librccl.so not found (TOPAY_RCCL_LIB names it explicitly)"); return TOPAY_ERR_UNSUPPORTED; }
  HIPCHK(hipSetDevice(c->device));
  (void)topay_comm_destroy(c);
  const int rc = a->CommInitRank(&c->comm, world, *id, rank);
  if (rc != 0) { c->comm = nullptr; return rccl_fail("ncclCommInitRank", rc); }
  c->comm_world = world;
  c->comm_rank = rank;
  if (!c->comm_stream) HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  return TOPAY_OK;
}

topay_status topay_comm_destroy(topay_ctx* c) {
  if (!c) return TOPAY_ERR_INVALID_ARG;
  if (c->comm) {
    RcclApi* a = rccl_api();
    if (a) (void)a->CommDestroy(c->comm);
    c->comm = nullptr;
  }
  if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
  c->comm_world = 0;
  return TOPAY_OK;
}

// planner.cpp:999-1010 per scenario: of the candidates that count (optimizeTraj true AND the gate passed) the one with
// the shortest total duration.  scenario_of[b] = scenario id of candidate b (any ints); one record per distinct id in
// order of first appearance, best_candidate relative to the scenario's first candidate, -1 / status 0 without a winner.
topay_status topay_scenario_records(topay_ctx* c, const int* scenario_of, int cap_records, topay_record_t* records, int* n_records,
                                    int* winner_index /* cap_records, may be null: batch index of each winner or -1 */) {
  if (!c || !c->have_traj) return TOPAY_ERR_NO_TRAJ;
  if (c->pending) { topay_status ws = topay_synchronize(c); if (ws != TOPAY_OK) return ws; }   // (waits like the other getters)
  if (!c->solved) return TOPAY_ERR_NO_TRAJ;
  if (!scenario_of || !records || !n_records || cap_records < 0) return TOPAY_ERR_INVALID_ARG;
  const int B = c->B;
  std::vector<int> ok(B), feas(B);
  std::vector<double> cost(B), dur(B);
  topay_status s;
  if ((s = topay_get_batch(c, ok.data(), cost.data(), nullptr)) != TOPAY_OK) return s;
  if ((s = topay_check_feasible(c, feas.data())) != TOPAY_OK) return s;
  if ((s = topay_get_total_durations(c, dur.data())) != TOPAY_OK) return s;
  std::vector<int> ids, first, best;
  for (int b = 0; b < B; b++) {
    int r = -1;
    for (int k = (int)ids.size() - 1; k >= 0; k--)   // candidates of a scenario are adjacent in practice: found at once
      if (ids[k] == scenario_of[b]) { r = k; break; }
    if (r < 0) { ids.push_back(scenario_of[b]); first.push_back(b); best.push_back(-1); r = (int)ids.size() - 1; }
    if (ok[b] && feas[b] && (best[r] < 0 || dur[b] < dur[best[r]])) best[r] = b;
  }
  *n_records = (int)ids.size();
  if ((int)ids.size() > cap_records) { set_err("topay_scenario_records: cap_records too small"); return TOPAY_ERR_INVALID_ARG; }
  for (size_t r = 0; r < ids.size(); r++) {
    topay_record_t& q = records[r];
    q.scenario_id = ids[r];
    q.best_candidate = best[r] < 0 ? -1 : best[r] - first[r];
    q.status = best[r] < 0 ? 0 : 1;
    q.n_pieces = best[r] < 0 ? 0 : c->hN[best[r]];
    q.cost = best[r] < 0 ? 0.0 / 0.0 : cost[best[r]];
    q.duration = best[r] < 0 ? 0.0 / 0.0 : dur[best[r]];
    if (winner_index) winner_index[r] = best[r];
  }
  return TOPAY_OK;
}

// ncclAllGather of `per_rank` records from every rank (fewer valid ones are padded with scenario_id = INT_MIN); `all`
// receives world x per_rank records in rank order, *n_valid the number that are not padding (compacted to the front).
static void record_padding(topay_record_t& r) {
  r.scenario_id = INT32_MIN; r.best_candidate = -1; r.status = 0; r.n_pieces = 0; r.cost = 0.0; r.duration = 0.0;
}
topay_status topay_pack_records(const topay_record_t* mine, int n_mine, int per_rank, topay_record_t* block) {
  if (n_mine < 0 || per_rank <= 0 || n_mine > per_rank || (n_mine > 0 && !mine) || !block) return TOPAY_ERR_INVALID_ARG;
  for (int r = 0; r < per_rank; r++) {
    if (r < n_mine) block[r] = mine[r];
    else record_padding(block[r]);
  }
  return TOPAY_OK;
}
topay_status topay_unpack_records(const topay_record_t* gathered, int world, int per_rank, topay_record_t* all, int* n_valid) {
  if (!gathered || !all || world <= 0 || per_rank <= 0) return TOPAY_ERR_INVALID_ARG;
  const size_t tot = (size_t)world * per_rank;
  size_t n = 0;
  for (size_t k = 0; k < tot; k++)
    if (gathered[k].scenario_id != INT32_MIN) {
      const topay_record_t q = gathered[k];   // (gathered and all may be the same buffer: n <= k)
      all[n++] = q;
    }
  for (size_t k = n; k < tot; k++) record_padding(all[k]);
  if (n_valid) *n_valid = (int)n;
  return TOPAY_OK;
}

topay_status topay_gather_records(topay_ctx* c, const topay_record_t* mine, int n_mine, int per_rank, topay_record_t* all, int* n_valid) {
  if (!c || !c->comm) { set_err("topay_gather_records: no communicator (topay_comm_init)"); return TOPAY_ERR_INVALID_ARG; }
  if (n_mine < 0 || per_rank <= 0 || n_mine > per_rank || (n_mine > 0 && !mine) || !all) return TOPAY_ERR_INVALID_ARG;
  RcclApi* a = rccl_api();
  HIPCHK(hipSetDevice(c->device));
  const size_t bytes = (size_t)per_rank * sizeof(topay_record_t);
  topay_status s;
  if ((s = c->comm_send.ensure(bytes)) != TOPAY_OK || (s = c->comm_recv.ensure(bytes * c->comm_world)) != TOPAY_OK) return s;
  std::vector<topay_record_t> pad((size_t)per_rank);
  (void)topay_pack_records(mine, n_mine, per_rank, pad.data());
  HIPCHK(hipMemcpyAsync(c->comm_send.p, pad.data(), bytes, hipMemcpyHostToDevice, c->comm_stream));
  const int rc = a->AllGather(c->comm_send.p, c->comm_recv.p, bytes, 0 /* ncclInt8 */, c->comm, c->comm_stream);
  if (rc != 0) return rccl_fail("ncclAllGather", rc);
  std::vector<topay_record_t> got((size_t)per_rank * c->comm_world);
  HIPCHK(hipMemcpyAsync(got.data(), c->comm_recv.p, bytes * c->comm_world, hipMemcpyDeviceToHost, c->comm_stream));
  HIPCHK(hipStreamSynchronize(c->comm_stream));
  return topay_unpack_records(got.data(), c->comm_world, per_rank, all, n_valid);
}

// Launch class of a candidate with n_pieces pieces: waves per trajectory and decision-vector elements per thread of
// the kernel that solves (and, through topay_eval, evaluates) it.  The division of the L-BFGS vectors over the threads
// -- and with it the rounding of every dot product -- is a function of these two; parity tooling that restates the
// solver in the device's order needs them (oracle/: device-order mode).
topay_status topay_class_of(int n_pieces, int* waves, int* elements_per_thread, int* class_index) {
  if (n_pieces <= 0 || n_pieces > TOPAY_MAX_N) return TOPAY_ERR_TOO_MANY_PIECES;
  const int k = bucket_of(n_pieces);
  const ClassDef& cd = class_table()[k];
  if (waves) *waves = cd.snw();
  if (elements_per_thread) *elements_per_thread = 2 * cd.srmax();
  if (class_index) *class_index = k;
  return TOPAY_OK;
}

// Device memory held by the resident batch (everything topay_set_init_traj sized), in bytes.
topay_status topay_workspace_bytes(topay_ctx* c, unsigned long long* bytes) {
  if (!c || !bytes) return TOPAY_ERR_INVALID_ARG;
  *bytes = (unsigned long long)c->workspace_bytes;
  return TOPAY_OK;
}

topay_status topay_gate_timeouts(topay_ctx* c, int* n) {
  if (!c || !n) return TOPAY_ERR_INVALID_ARG;
  *n = c->gate_timeouts;
  return TOPAY_OK;
}

topay_status topay_last_kernel_ms(topay_ctx* c, double* ms, int* launches) {
  if (!c) return TOPAY_ERR_INVALID_ARG;
  if (ms) *ms = c->last_ms;
  if (launches) *launches = c->last_launches;
  return TOPAY_OK;
}

topay_status topay_last_helper_launches(topay_ctx* c, int* n) {
  if (!c || !n) return TOPAY_ERR_INVALID_ARG;
  *n = c->last_helper_launches;
  return TOPAY_OK;
}


#if defined(TOPAY_STAMPS) && !defined(TOPAY_CPU_EMU)
// diagnostic build only: per-phase cycle counters of the manipulator block (lane 0 of block 0)
topay_status topay_debug_mani_stamps(long long* out8) {
  HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(topay::g_mani_stamps), 64));
  return TOPAY_OK;
}
#endif
}  // extern "C"
