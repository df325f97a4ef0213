/*
 * topay.h — C-ABI of the MI355X-native batched (s,theta) trajectory optimizer.
 *
 * Drop-in boundary for ONE hot path of TopAY-Planner/TopAY: `MomaTrajOpt::optimizeTraj`
 * (reference: src/planner/src/moma_traj_opt.cpp:142-498) and what it calls per L-BFGS
 * iteration.  The reference exposes an in-process C++ class, one instance per candidate
 * thread (src/planner/include/planner/moma_traj_opt.h:613-674); this ABI is the same surface,
 * batch-first: one context optimises `batch` independent candidate trajectories at once,
 * one 64-lane wavefront per trajectory, on one GPU.
 *
 *   reference member                                   replaced by
 *   -------------------------------------------------  ------------------------------------
 *   MomaTrajOpt(GridMap::Ptr)        moma_traj_opt.h:651   topay_create + topay_set_map
 *   init(ros::NodeHandle&)           moma_traj_opt.h:845   topay_default_params / topay_create
 *   optimizeTraj(...) lines 146-357  moma_traj_opt.cpp     topay_set_init_traj
 *   optimizeTraj(...) lines 359-497  moma_traj_opt.cpp     topay_optimize
 *   getTraj(), traj_cost             moma_traj_opt.h:943   topay_get_result
 *   first/secondStageCostCallback    moma_traj_opt.cpp:817,885   topay_eval (test hook)
 *   printConstraintsSituations       moma_traj_opt.h:1052  topay_check_feasible
 *
 * Conventions: plain pointers and sizes, no C++/torch types, never throws.  Every call
 * returns a topay_status (0 = ok).  A context is single-caller (like one MomaTrajOpt); several
 * contexts may coexist (one per GPU / process).  All host buffers are caller-owned and copied;
 * the context owns its device memory.  Arithmetic is IEEE double throughout.
 */
#ifndef TOPAY_H
#define TOPAY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int topay_status;
enum {
  TOPAY_OK = 0,
  TOPAY_ERR_INVALID_ARG = -1,
  TOPAY_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime error; see topay_last_error() */
  TOPAY_ERR_NO_MAP = -3,
  TOPAY_ERR_NO_TRAJ = -4,
  TOPAY_ERR_TOO_MANY_PIECES = -5,
  TOPAY_ERR_UNSUPPORTED = -6
};

/* Per-trajectory solver status (reported by topay_get_result / topay_get_stats).  Values >= -1024
 * are the reference's L-BFGS codes (src/planner/include/utils/lbfgs.hpp:135-184). */
enum {
  TOPAY_LBFGS_CONVERGENCE = 0,
  TOPAY_LBFGS_STOP = 1,
  TOPAY_LBFGS_CANCELED = 2,
  TOPAY_LBFGSERR_INVALID_FUNCVAL = -1012,
  TOPAY_LBFGSERR_MINIMUMSTEP = -1011,
  TOPAY_LBFGSERR_MAXIMUMSTEP = -1010,
  TOPAY_LBFGSERR_MAXIMUMLINESEARCH = -1009,
  TOPAY_LBFGSERR_MAXIMUMITERATION = -1008,
  TOPAY_LBFGSERR_WIDTHTOOSMALL = -1007,
  TOPAY_LBFGSERR_INVALIDPARAMETERS = -1006,
  TOPAY_LBFGSERR_INCREASEGRADIENT = -1005
};

/* L-BFGS parameter block — mirrors lbfgs::lbfgs_parameter_t (lbfgs.hpp:15-129). */
typedef struct {
  int mem_size;
  int past;
  int max_iterations;
  int max_linesearch;
  double g_epsilon;
  double delta;
  double min_step;
  double max_step;
  double f_dec_coeff;
  double s_curv_coeff;
  double cautious_factor;
  double machine_prec;
} topay_lbfgs_params_t;

/* Optimizer parameters — mirror MomaTrajOptParam (moma_traj_opt.h:441-564) with the values of
 * src/planner/params/optimizer.yaml as defaults, plus the MomaParam constants the path reads
 * (src/simulator/fake_moma/include/fake_moma/moma_param.h:36-126). */
typedef struct {
  int int_K;            /* Simpson panels per piece; this build supports 12 only */
  int min_piece_num;
  double relu_mu;
  double sample_interval;
  double energy_weights[9];
  /* first stage */
  double s1_time_weight, s1_moment_weight, s1_acc_weight, s1_domega_weight, s1_path_pos_weight;
  int s1_normal_past, s1_shot_path_past;
  double s1_shot_path_horizon;
  topay_lbfgs_params_t s1_lbfgs;
  /* second stage */
  double s2_time_weight, s2_moment_weight, s2_acc_weight, s2_domega_weight;
  double s2_collision_weight, s2_mani_colli_weight, s2_self_colli_weight;
  double s2_mani_pos_weight, s2_mani_vel_weight, s2_mani_acc_weight, s2_mean_time_weight;
  topay_lbfgs_params_t s2_lbfgs;
  /* ALM on the end-point constraint (only entries [0],[1] of the reference's vectors are used) */
  double alm_init_lambda[2], alm_init_rho[2], alm_rho_max[2], alm_gamma[2];
  double alm_tolerance;
  /* Deterministic replacement of the reference's 1.0 s wall-clock cap on the ALM loop
   * (moma_traj_opt.cpp:403-407): maximum number of outer iterations. */
  int alm_max_outer;
  /* Second part of that replacement: the reference checks its 1.0 s clock only before starting another ALM round;
   * here no further round is started once the stage-2 work done so far -- evaluations x pieces, the unit the
   * cost of an evaluation is proportional to -- reaches this budget.  Default 24000 piece-evaluations: one second
   * of the CPU path at the ~42 us per piece-evaluation measured for the oracle.  0 = unlimited. */
  int alm_work_budget;
  /* robot (MomaParam) */
  double chassis_height, chassis_colli_radius;
  double max_v, max_a, max_w, max_dw;
  double colli_length[8];
  double colli_points[16];
  double colli_point_radius[16];
  double joint_pos_limit_max[7];
  double joint_vel_limit[7];
  double joint_acc_limit[7];
  double relative_R[9]; /* row-major */
  double relative_t[3];
} topay_params_t;

/* ESDF description — GridMap (src/map/include/map/grid_map.h) dense buffers:
 *   esdf2d[x*dims[1] + y]                        (grid_map.h:798-806)
 *   esdf3d[x*dims[1]*dims[2] + y*dims[2] + z]    (grid_map.h:808-816) */
typedef struct {
  double origin[3];
  double resolution;
  int dims[3];
  double min_boundary[3];
  double max_boundary[3];
} topay_map_desc_t;

typedef struct topay_ctx topay_ctx;

/* Fill `p` with the reference defaults (optimizer.yaml + MomaParam). */
topay_status topay_default_params(topay_params_t* p);

/* == MomaTrajOpt::init (src/planner/include/planner/moma_traj_opt.h:845-941) for a caller without a ROS parameter server:
 * the optimiser parameters from the reference's parameter file (src/planner/params/optimizer.yaml, under
 * `planner_node: moma_traj_opt:` or at top level).  path_or_text: a file name, or the YAML text itself.  Keys that are
 * absent keep the value *params has on entry (call topay_default_params first).  ignored (optional, NUL-terminated,
 * newline-separated): the keys init() reads but this path has no use for -- mean_time_lowb / mean_time_uppb (the
 * reference's penalty uses the literals 0.5 and 2.0, moma_traj_opt.cpp:1752-1769), first_stage/mean_time_weight,
 * second_stage/alm_data/... -- and any key init() does not know. */
topay_status topay_params_from_yaml(const char* path_or_text, topay_params_t* params, char* ignored, int ignored_cap);

/* Create a context on HIP device `device` (use LOCAL_RANK for one process per GPU). */
topay_status topay_create(const topay_params_t* params, int device, topay_ctx** out);
void topay_destroy(topay_ctx* ctx);

/* == assigning MomaTrajOpt::opt_param (a public member the planner may change between calls, moma_traj_opt.h:616):
 * replace the context's parameters.  Weights, L-BFGS and ALM settings take effect with the next solve / evaluation of
 * the resident batch; a change of sample_interval, min_piece_num, the velocity / acceleration limits or the history
 * depth invalidates the resident batch (call topay_set_init_traj again).  Waits for a solve in flight. */
topay_status topay_set_params(topay_ctx* ctx, const topay_params_t* params);
const char* topay_last_error(void);

/* Upload a map; the context keeps a device copy.  Up to TOPAY_MAX_MAPS maps may be resident (the
 * reference's benchmark loop uses a fresh map per episode); `map_id` selects the slot (0 for the
 * single-map case). */
#define TOPAY_MAX_MAPS 4096
topay_status topay_set_map(topay_ctx* ctx, int map_id, const topay_map_desc_t* desc, const double* esdf2d,
                           const double* esdf3d);

/* == GridMap::updateESDF (src/map/src/grid_map.cpp:125-521): build the 2-D and 3-D signed distance fields of map slot
 * `map_id` ON THE DEVICE from the occupancy grids the reference fills from its point cloud (grid_map.cpp:733-747):
 *   occ2d[x*dims[1] + y] (1 = a point below the chassis height), occ3d[x*dims[1]*dims[2] + y*dims[2] + z].
 * Afterwards the slot is as if topay_set_map had been called with the CPU-built fields (bit-identical values).
 * topay_get_map copies a resident map back and reports the duration of the last build in milliseconds. */
topay_status topay_build_esdf(topay_ctx* ctx, int map_id, const topay_map_desc_t* desc, const signed char* occ2d,
                              const signed char* occ3d);
/* n_maps maps of equal dimensions in one set of launches (the benchmark loop regenerates the map every episode): the
 * occupancy grids are concatenated map after map; slots first_map_id .. first_map_id + n_maps - 1 are filled. */
topay_status topay_build_esdf_batch(topay_ctx* ctx, int n_maps, int first_map_id, const topay_map_desc_t* desc,
                                    const signed char* occ2d, const signed char* occ3d);
topay_status topay_get_map(topay_ctx* ctx, int map_id, double* esdf2d, double* esdf3d, double* build_ms);

/* Map slots first_map_id .. first_map_id + n_maps - 1 of `ctx` become references to the resident maps of `owner` (same
 * device), without a copy: the maps are read-only for every entry point but the three that fill a slot.  The reference
 * hands one GridMap::Ptr to all of its optimisers (planner.cpp:59-75: every MomaTrajOpt gets the planner's grid_map);
 * here the contexts of the batches in flight share one set of fields the same way.  The library tracks the sharing:
 * when `owner` refills one of the slots (topay_set_map, topay_build_esdf*), or is destroyed, every context that shares
 * the slot first finishes its pending solve and then loses it -- its next use of the slot returns TOPAY_ERR_NO_MAP until
 * it is shared or filled again; nothing is left pointing at freed or half-written fields. */
topay_status topay_share_maps(topay_ctx* ctx, topay_ctx* owner, int first_map_id, int n_maps);

/* All five fields of GridMap::updateESDF (src/map/src/grid_map.cpp:125-521).  Besides esdf2d and esdf3d (above) the
 * reference builds two more 2-D fields for its front-end:
 *   esdf_buffer_2d_inflate   (355-423): the signed field of the cells where esdf2d < chassis_colli_radius;
 *   esdf_buffer_2d_critical  (211-351): the signed field of occ_buffer_2d_critical (any cloud point above the cell,
 *                            whatever its height, grid_map.cpp:733-747), then -- into the same buffer, as the reference
 *                            does at line 348 -- the field of the cells where that one is < chassis_colli_radius.
 * occ2d_critical may be NULL: the projection of occ3d onto the plane is used (what the reference's fill produces for a
 * cloud inside the map's height).  topay_build_esdf / _batch are this call with occ2d_critical = NULL.
 * topay_get_map_fields copies the two extra fields of a built map back (either pointer may be NULL). */
topay_status topay_build_esdf_fields(topay_ctx* ctx, int n_maps, int first_map_id, const topay_map_desc_t* desc,
                                     const signed char* occ2d, const signed char* occ2d_critical, const signed char* occ3d);
topay_status topay_get_map_fields(topay_ctx* ctx, int map_id, double* esdf2d_inflate, double* esdf2d_critical);

/* == optimizeTraj lines 146-357 for every batch member.
 *   path_len[b]      number of 10-d states of candidate b
 *   init_paths       ragged, sum(path_len) x 10, row-major (x, y, theta, q1..q7)
 *   boundary_vel/acc batch x 20, each a 10 x 2 column-major matrix as in the reference
 *                    (row 0 = v, row 1 = omega, rows 3-9 = joints; col 0 = start, col 1 = end); NULL = zeros
 *   map_ids          map slot per candidate, NULL = all slot 0
 * Uploads the raw paths (they stay resident), runs the init kernel, sizes the workspace.
 * The reference puts no bound on the number of pieces (moma_traj_opt.cpp:245, 300-321); this build solves up to 170
 * (a 255 s trajectory at the reference's 1.5 s sample_interval: what a four-wave workgroup holds in a compute unit's LDS).  A
 * candidate that needs more is reported as failed (success 0, cost NaN, n_pieces 0) without being launched and the
 * rest of the batch is solved normally; topay_get_batch's n_pieces lets the caller count such candidates. */
topay_status topay_set_init_traj(topay_ctx* ctx, int batch, const int* path_len, const double* init_paths,
                                 const double* boundary_vel, const double* boundary_acc, const int* map_ids);

/* Re-run the init kernel from the resident raw paths (restores x0 so the same batch can be
 * optimised again without uploading the paths a second time). */
topay_status topay_reset(topay_ctx* ctx);

/* == optimizeTraj lines 359-497 for every batch member (stage-1 L-BFGS, stage-2 ALM loop). */
topay_status topay_optimize(topay_ctx* ctx);

/* The same in two halves, so that a caller can keep several contexts (batches) in flight on one GPU: the tail of one
 * batch -- a few long candidates, most of the device idle -- then overlaps the bulk of the next.  topay_optimize ==
 * topay_optimize_async + topay_synchronize.  Results may be read after topay_synchronize.  The asynchronous call may
 * block until every candidate of the batch issued before it (by any context of the process on this device) has started:
 * the library hands the device over oldest batch first.  All contexts of a process share one parameter block in constant
 * memory: a solve whose parameters differ from those of a solve still in flight waits for that one to finish. */
topay_status topay_optimize_async(topay_ctx* ctx);
topay_status topay_synchronize(topay_ctx* ctx);

/* Batch-level results after topay_optimize: success[b] (0/1), cost[b] (traj_cost), n_pieces[b].
 * Any pointer may be NULL. */
topay_status topay_get_batch(topay_ctx* ctx, int* success, double* cost, int* n_pieces);

/* getTraj() of candidate i: durations[N]; coeffs[N][9][6] highest order first (minco.hpp:908-921);
 * knots_xy[(N+1)][2] Simpson-integrated piece end points.  Any output pointer may be NULL. */
topay_status topay_get_result(topay_ctx* ctx, int i, int* success, double* cost, int* n_pieces, double* durations,
                              double* coeffs, double* knots_xy);

/* getTraj() for a selection of candidates in ONE call and one device-to-host copy -- what a planner does with the
 * winners it has picked from topay_get_batch / topay_check_feasible / topay_get_total_durations (planner.cpp:999-1016:
 * the shortest feasible candidate of a scenario is the one that gets published).  Results are packed by pieces:
 *   idx[n]              candidate indices (a candidate that was never launched contributes zero pieces)
 *   piece_off[n + 1]    OUT: candidate k owns pieces [piece_off[k], piece_off[k+1]) of durations / coeffs
 *   durations[pieces]   coeffs[pieces][9][6] highest order first (minco.hpp:908-921)
 *   knots_xy            candidate k's N_k + 1 piece end points start at knots_xy[2 * (piece_off[k] + k)]
 *   cap_pieces          capacity of the output arrays in pieces; TOPAY_ERR_INVALID_ARG when the selection needs more */
topay_status topay_get_results(topay_ctx* ctx, int n, const int* idx, int cap_pieces, int* piece_off, double* durations,
                               double* coeffs, double* knots_xy);

/* Candidate i in the layout of src/planner/msg/PolyTraj.msg (uint8 order; std_msgs/Float32MultiArray[] coeff;
 * float32[] durations; int8[] directions -- the reference defines the message but never fills it, so the field
 * meanings are the obvious ones): *order = 5; coeff[N][9][6] = getTraj()'s coefficients as float32, one 9 x 6 array per
 * piece, highest order first; durations[N]; directions[N] = sign of the arc-length rate at the middle of the piece
 * (+1 forward, -1 reverse).  Capacity cap_pieces; *n_pieces receives N. */
topay_status topay_get_polytraj_msg(topay_ctx* ctx, int i, int cap_pieces, unsigned char* order, float* coeff,
                                    float* durations, signed char* directions, int* n_pieces);

/* Per-candidate solver counters, 8 ints each:
 * {stage1_ret, stage1_iters, stage1_evals, stage2_last_ret, stage2_iters, stage2_evals, alm_outer, sum_bound}
 * sum_bound = sum over stage-2 iterations of the two-loop history length (roofline accounting). */
topay_status topay_get_stats(topay_ctx* ctx, int* stats /* batch x 8 */);

/* Per-candidate optimisation time in microseconds, measured on the device (the reference logs the same quantity per
 * candidate, planner.cpp:893-905 "optimization time").  start_us (optional): when each solve started, on the same
 * device clock (only differences are meaningful); hw_id (optional): the hardware slot it ran on,
 * xcc << 16 | se << 12 | cu << 4 | simd (scheduling diagnostics).  Any pointer may be NULL. */
topay_status topay_get_elapsed_us(topay_ctx* ctx, double* us /* batch */, double* start_us /* batch */, int* hw_id /* batch */);

/* == GridMap::isWholeBodyCollision (src/map/include/map/grid_map.h:613-650) for n states (x, y, theta, q1..q7) against
 * map slot map_id: collide[i] = 1 when state i violates a joint limit, lies outside the map or collides (chassis, the 12
 * arm spheres against the environment, the chassis top and each other).  The check the front-end applies to every state
 * it samples or interpolates (mcrrts.cpp, planner.cpp:529-548). */
topay_status topay_whole_body_collision(topay_ctx* ctx, int map_id, int n, const double* states, int* collide);

/* First slice of the front-end that produces optimizeTraj's init paths.
 * topay_dense_path == GraphSearch::getDensePath (src/planner/src/graph_search.cpp:119-176) for n_paths raw 2-D paths:
 *   raw_len[p] points of path p, raw_xy ragged (sum x 2); out[p][cap_per_path][4] = (x, y, theta, dt) of the entries the
 *   reference keeps (dt > 1e-3, plus the final pose); out_len[p] = their number (entries beyond cap_per_path are counted,
 *   not written).
 * topay_connect_check_num / topay_connect_collision == MCRRTs::connectCollision (src/planner/include/planner/mcrrts.h:
 *   310-348) for n_edges edges of the joint-space tree.  The car poses along an edge come from OMPL's
 *   ReedsSheppStateSpace (distance, interpolate: mcrrts.h:318-324, 336) -- a third-party dependency of the reference,
 *   not part of it -- so the caller supplies rs_distance[e] and, for the piece_num[e] checks of edge e, the interpolated
 *   car poses car_poses[sum piece_num][3] at fractions i / piece_num[e]; the library interpolates the joints, builds the
 *   states and runs GridMap::isWholeBodyCollision on every one: collide[e] = 1 when any state of the edge collides. */
topay_status topay_dense_path(topay_ctx* ctx, int n_paths, const int* raw_len, const double* raw_xy, double step_size,
                              const double* start_yaw, const double* end_yaw, double v_max, double w_max, int cap_per_path,
                              int* out_len, double* out);
topay_status topay_connect_check_num(int n_edges, const double* rs_distance, const double* q_from /* n x 7 */,
                                     const double* q_to /* n x 7 */, double check_res, int* piece_num);
topay_status topay_connect_collision(topay_ctx* ctx, int map_id, int n_edges, const int* piece_num, const double* car_poses,
                                     const double* q_from, const double* q_to, int* collide);

/* == GraphSearch::plan2dJPS (src/planner/src/graph_search.cpp:53-117, with plan / getJpsSucc / jump / hasForced 178-475 and the
 * JPS2DNeib neighbour rules 583-669): the planner's direct chassis path (planner.cpp:816: threshold = chassis radius + 0.1)
 * for n (start, goal) pairs -- jump-point search on the 2-D distance field of each pair's map slot, then the zigzag cut
 * with GridMap::isLineCollisionGrid2d (grid_map.h:565-610).  out_len[p] = points of path p (0: no path; counts beyond
 * cap_points are reported, not written), out_xy + p * cap_points * 2 = its points, first = start, last = goal: the raw
 * path topay_dense_path takes.  stats (n x 2, optional): expanded nodes, jump points before the cut.  The open list
 * follows boost::heap::d_ary_heap<arity<2>> with the reference's compare_state (graph_search.h:20-38; boost is not part of
 * the reference's sources, its sift rules are restated), which is what decides between equal-cost paths.  One wavefront per
 * search (lane 0 runs the heap, the wave scans the jumps); the search state (21 bytes per map cell and search) is allocated
 * for the call. */
topay_status topay_plan2d_jps(topay_ctx* ctx, int n, const int* map_ids /* n, NULL = slot 0 */, const double* start_xy /* n x 2 */,
                              const double* end_xy /* n x 2 */, double threshold, int cap_points, int* out_len, double* out_xy, int* stats);

/* == MCRRTs::plan (src/planner/src/mcrrts.cpp:5-231; steer / rewire 336-400; the inline members of
 * src/planner/include/planner/mcrrts.h:153-348), the layered bidirectional search over the arm joints along a fixed chassis
 * path that produces optimizeTraj's init path -- n instances (candidate chassis paths) per call, one wavefront each.
 *   path_len[p], car_paths    chassis path of instance p: path_len[p] entries (x, y, theta, dt) as topay_dense_path /
 *                             GraphSearch::getDensePath return them (ragged, sum x 4); 2..min(cap_per_path, 255) entries
 *   start, end                n x 10 full states (x, y, theta, q1..q7); the joints seed the two trees
 *   wb_len[p], wb_path        the whole-body path: wb_len[p] states of 10 doubles (one per layer) at wb_path + p *
 *                             cap_per_path * 10; wb_len[p] = 0 when no path was found
 *   stats (n x 8, optional)   status (1 path, 0 none, -1 node pool full, -2 bad input), nodes, iterations, tree nodes,
 *                             anti-tree nodes, the two nodes that joined the trees, whole-body collision checks
 *   c_max (n, optional)       cost of the connection (mcrrts.cpp:74-79)
 * The chassis poses along a tree edge are ompl::base::ReedsSheppStateSpace(1e-2)'s (mcrrts.h:134, 318-324, 336); OMPL is a
 * third-party dependency that is not part of the reference's sources, its published algorithm (Reeds & Shepp 1990) is
 * implemented in the library (topay_reeds_shepp exposes it).  Where the reference is not reproducible the library is
 * deterministic (topay_mcrrt_params_t): every random draw is a pure function of (seed, first_instance + p, iteration,
 * slot) instead of a std::mt19937 seeded from std::random_device (mcrrts.h:89), and the 0.2 s wall-clock limits
 * (params/mcrrts.yaml, mcrrts.cpp:43, 143, mcrrts.h:225) are counts. */
typedef struct {
  double goal_sample_rate;     /* mcrrts/goal_sample_rate (0.4) */
  double check_colli_res;      /* mcrrts/check_colli_res (0.01): spacing of the collision checks along an edge */
  double rs_turning_radius;    /* ReedsSheppStateSpace(1.0e-2), mcrrts.h:134 */
  int max_iter;                /* iterations of the main loop (stands in for mcrrts/max_time) */
  int max_sample_tries;        /* resamplings of sampleState while the sample is in collision (mcrrts.h:225) */
  int node_cap;                /* nodes per instance; the search fails with status -1 when it needs more */
  int reserved;
  unsigned long long seed;
} topay_mcrrt_params_t;
void topay_mcrrt_default_params(topay_mcrrt_params_t* p);
topay_status topay_mcrrt_plan(topay_ctx* ctx, int n, const int* map_ids /* n, NULL = slot 0 */, const int* path_len, const double* car_paths,
                              const double* start, const double* end, const topay_mcrrt_params_t* params /* NULL = defaults */,
                              unsigned long long first_instance, int cap_per_path, int* wb_len, double* wb_path, int* stats, double* c_max);
/* Node table of instance `instance` of the last topay_mcrrt_plan, in creation order (diagnostics, parity tests): layer
 * (robo_state.first), state (MCRRTNode::NodeState: 1 EXPANDED, 2 IN_TREE, 3 IN_ANTI_TREE), parent (index, -1 none),
 * cost, q (7 per node).  Up to cap rows; any pointer may be NULL. */
topay_status topay_mcrrt_nodes(topay_ctx* ctx, int instance, int cap, int* layer, int* state, int* parent, double* cost, double* q);
/* ompl::base::ReedsSheppStateSpace(rho) for n pose pairs (x, y, theta): distance[i] = distance(from_i, to_i), word[i] /
 * lengths[i][5] = the shortest path's segment word (0..17, OMPL's reedsSheppPathType numbering) and signed segment lengths
 * in units of rho, pose[i] = interpolate(from_i, to_i, t[i]) when t is given.  Output pointers may be NULL. */
topay_status topay_reeds_shepp(topay_ctx* ctx, int n, const double* from, const double* to, const double* t /* n or NULL */, double rho,
                               double* distance, int* word, double* lengths, double* pose);

/* MomaTraj playback of candidate i (moma_traj_opt.h:26-137): car_seq -- (x, y, theta, t) every 0.1 s from Simpson
 * panels of 0.025 s, what the reference stores in the trajectory object and publishes -- and getState(t) at caller-given
 * times, states[n_times][10] = (x, y, theta, q1..q7).  seq (seq_cap rows of 4 doubles) may be NULL; *n_seq receives the
 * number of car_seq entries. */
topay_status topay_playback(topay_ctx* ctx, int i, int n_times, const double* times, double* states, int seq_cap,
                            double* seq, int* n_seq);

/* Mesh kinematics of MomaParam that only getMeshPose uses (moma_param.h:60-67, 77-90, 114-115). */
typedef struct topay_mesh_params_t {
  double link_length[7];
  double joint_pos_limit_min[7];
  double joint_offset[21];   /* 7 x 3 row-major: roll, pitch, yaw of the fixed rotation before joint i */
  double joint_dof_axis[21]; /* 7 x 3 row-major: the joint angle multiplies this (roll, pitch, yaw) triple */
} topay_mesh_params_t;
topay_status topay_default_mesh_params(topay_mesh_params_t* p);

/* == MomaParam::getMeshPose (moma_param.h:724-790) for n states (x, y, theta, q1..q7): parts[n][11][7], rows = chassis,
 * stump, link 1..7, end effector (a copy of link 7), end-effector collision point (position only, orientation zero);
 * columns = x, y, z, qw, qx, qy, qz as in MeshPart.msg. */
topay_status topay_mesh_poses(topay_ctx* ctx, const topay_mesh_params_t* mesh, int n, const double* states, double* parts);

/* == Planner::toMeshMsg (planner.cpp:2003-2056) of candidate i, the MeshTraj message the planner publishes for the winner
 * (planner.cpp:1014-1016): the trajectory sampled through MomaTraj::getState every T / res (res = 1000 in the
 * reference; t accumulated and compared `t < T` as there, so *n_states is res or res + 1), per sample the 11 mesh
 * poses, the yaw and the accumulated chassis arc length.  cap_states >= res + 1. */
topay_status topay_mesh_traj(topay_ctx* ctx, int i, const topay_mesh_params_t* mesh, int res, int cap_states, double* parts,
                             double* yaws, double* arc_lengths, int* n_states);

/* Total duration of every candidate's returned trajectory (MomaTraj::getTotalDuration): the quantity the planner ranks
 * the successful candidates of a scenario by (planner.cpp:999-1010). */
topay_status topay_get_total_durations(topay_ctx* ctx, double* total /* batch */);

/* ALM state (alm_lambda[2], alm_rho[2]) every candidate finished with; traj_cost is the stage-2 cost at the returned x
 * with this state (moma_traj_opt.cpp:398-401, 456-459). */
topay_status topay_get_alm(topay_ctx* ctx, double* alm /* batch x 4: lambda0, lambda1, rho0, rho1 */);

/* Number of decision variables of candidate i (n = 10N - 8) and the packed vector
 * x = [tau(N) | theta(N-1) | s(N) | Vq(7 x (N-1), column = knot)] (moma_traj_opt.cpp:324-344). */
topay_status topay_get_x(topay_ctx* ctx, int i, int* n, double* x);

/* Test hook == first/secondStageCostCallback (moma_traj_opt.cpp:817-955): evaluate stage `stage`
 * (1 or 2) of candidate i at x with ALM state (lambda, rho); returns f, g[n] and, for stage 2,
 * final_xy_error[2] (NULL allowed). */
topay_status topay_eval(topay_ctx* ctx, int stage, int i, const double* x, const double* alm_lambda,
                        const double* alm_rho, double* f, double* g, double* final_xy_error);

/* Test hook: topay_eval by the kernel with `waves` wavefronts per trajectory (1: N <= 64, 2: N <= 64, 4: N <= 170)
 * instead of the candidate's class default.  The evaluation is order-identical for any number of waves: f, g and the
 * end-point error must come out bit for bit equal. */
topay_status topay_eval_waves(topay_ctx* ctx, int stage, int i, int waves, const double* x, const double* alm_lambda,
                              const double* alm_rho, double* f, double* g, double* final_xy_error);

/* The spline of a given decision vector as candidate i's result.  MomaTrajOpt keeps the MINCO state of its last cost
 * evaluation and getTraj() returns it (moma_traj_opt.h:943-946; secondStageCostCallback, moma_traj_opt.cpp:885-955):
 * one stage-2 evaluation at x with the ALM state (lambda, rho), after which topay_get_result(s), topay_playback,
 * topay_mesh_traj, topay_get_polytraj_msg and the feasibility gate serve x's trajectory (replay / warm-start entry;
 * the stored cost is the stage-2 cost at x, success is set). */
topay_status topay_load_solution(topay_ctx* ctx, int i, const double* x, const double* alm_lambda, const double* alm_rho);

/* Batched form of the hook over candidates [0, batch): x is batch x nmax (row stride nmax from
 * topay_get_nmax), g likewise; used by bench.py to time the cost/gradient kernel on its own. */
topay_status topay_eval_batch(topay_ctx* ctx, int stage, int repeats, double* f /* batch */);
topay_status topay_get_nmax(topay_ctx* ctx, int* nmax, int* Nmax);

/* == printConstraintsSituations (moma_traj_opt.h:1052-1204), the gate the planner applies to every optimised candidate
 * (planner.cpp:878-880), for every candidate of the batch after topay_optimize: feasible[b] 0/1.  The trajectory is
 * sampled every 0.01 s through MomaTraj::getState (moma_traj_opt.h:26-137).
 * topay_feasibility_report also returns checkFeasible's verdict (948-1050: additionally requires the sphere clearances)
 * and the extreme values the verdicts are made from, 38 doubles per candidate:
 *   |v| |a| |omega| |domega| maxima, |q| max[7], |dq| max[7], |d2q| max[7], chassis min distance, sphere min distance[12].
 * Any output pointer may be NULL. */
topay_status topay_check_feasible(topay_ctx* ctx, int* feasible);
topay_status topay_feasibility_report(topay_ctx* ctx, int* feasible, int* strict, double* report /* batch x 38 */);

/* Debug / parity tooling: record the cost f of every evaluation made by the next topay_optimize
 * (at most `cap` per candidate, 0 switches it off); topay_get_trace copies candidate i's `cap` values. */
topay_status topay_set_trace(topay_ctx* ctx, int cap);
topay_status topay_get_trace(topay_ctx* ctx, int i, double* out);

/* Test hook for the deterministic elementary functions of the solver (topay_amd/csrc/topay_math.h):
 * out[4i..] = sin(a_i), cos(a_i), atan2(a_i, b_i), sqrt(|a_i|)/(1+|b_i|). */
topay_status topay_test_math(topay_ctx* ctx, int n, const double* a, const double* b, double* out4n);

/* == the planner's thread group and its cancellation (planner.cpp:829-952): the candidates of one planning call run
 * concurrently, and 100 ms after the first of them has succeeded (optimizeTraj true AND printConstraintsSituations
 * passed) the ones still running are interrupted (threads.interrupt_all(); interruption points: top of the ALM loop and
 * every second-stage cost evaluation, moma_traj_opt.cpp:402, 887) and count as failed.
 *   group_id[b]    planning call of candidate b, -1 = none (NULL: no groups)
 *   cancel_budget  the 100 ms in the deterministic unit of alm_work_budget: piece-evaluations (stage-1 + stage-2
 *                  evaluations x pieces; 24 000 = 1 s, so 2400 = 100 ms); 0 = no cancellation (the default)
 * The rule is applied on the candidates' own work clocks -- a candidate counts iff its clock at the end is within
 * cancel_budget of the smallest clock of a feasible success of its call -- so the outcome does not depend on how the
 * device happened to schedule the candidates; the solve stops a candidate as soon as it can see that it is past the limit.
 * Interrupted candidates: success 0, stage-2 status TOPAY_INTERRUPTED, topay_get_interrupted 1.
 * Call after topay_set_init_traj (a new batch starts without groups). */
#define TOPAY_INTERRUPTED (-2000)
topay_status topay_set_groups(topay_ctx* ctx, const int* group_id /* batch */, int cancel_budget);
/* threads.interrupt_all() (planner.cpp:952) for the solve in flight: every candidate stops at one of its next interruption
 * points (the flag lives in host memory; a running candidate looks at it ahead of every eighth stage-2 evaluation -- every
 * fourth / every one with more than 16 / 32 pieces -- i.e. within a millisecond or two; a workgroup before it takes another
 * candidate).  Returns at once (callable from another thread than the one in topay_synchronize). */
topay_status topay_cancel(topay_ctx* ctx);
topay_status topay_get_interrupted(topay_ctx* ctx, int* interrupted /* batch */);
/* The wall-clock knob of a planning call (max_replan_time, agent_benchmark_tables.yaml:6; the 1.0 s cap of the ALM loop,
 * moma_traj_opt.cpp:403-407, is of the same kind): topay_optimize_async, then topay_cancel if the batch has not finished
 * within budget_ms of wall time, then topay_synchronize.  Candidates that finished in time keep their results; the rest
 * return TOPAY_INTERRUPTED.  *timed_out (may be NULL) tells whether the cancel was needed.  Unlike the deterministic
 * budgets (alm_max_outer, alm_work_budget, topay_set_groups) the outcome depends on the machine and its load. */
topay_status topay_optimize_within(topay_ctx* ctx, double budget_ms, int* timed_out);

/* A planning call on an otherwise idle device (BASELINE configs[1]: 64 candidates of one scenario; planner.cpp:930-958 runs its
 * handful of candidates one after the other).  mode 1: a batch with at most one candidate per SIMD of the device (1024 on an MI355X: up
 * to there the longest candidate decides the time of the batch, measured) runs its candidates of up to 32 pieces on workgroups of
 * four waves -- the one-wave solver on the first, all four in every cost / gradient evaluation
 * (the sample passes of the two sweeps side by side).  Results are those of the default kernels bit for bit (the evaluation is
 * order-identical for any number of waves, the solver is the one-wave solver); what changes is the time of a solve.  mode 2: for
 * every batch (tests); mode 0 (the default): never.  Takes effect at the next topay_optimize / topay_optimize_async. */
topay_status topay_set_latency_mode(topay_ctx* ctx, int mode);

/* ---- multi-GPU: scenarios shard over the GPUs of a node (one process per GPU, nothing of the solve is shared); the one
 * exchange is the all-gather of a 32-byte record per scenario over RCCL (the reference's selection of
 * a scenario's winner, planner.cpp:999-1010, stays local).  A C++ planner shards with these four calls and no torch:
 *   rank 0: topay_comm_unique_id(&id), hands the 128 bytes to the other processes (file, pipe, ROS parameter ...);
 *   every rank: topay_comm_init(ctx, &id, world, rank); per step topay_scenario_records + topay_gather_records.
 * RCCL is bound at run time (librccl.so.1; TOPAY_RCCL_LIB overrides): TOPAY_ERR_UNSUPPORTED when it is not there. */
typedef struct { char internal[128]; } topay_comm_id_t;          /* == ncclUniqueId */
typedef struct {
  int scenario_id;      /* caller's id of the scenario (planning call) */
  int best_candidate;   /* index of the winner among the scenario's candidates, -1: none */
  int status;           /* 1: a candidate succeeded and passed the gate */
  int n_pieces;         /* pieces of the winner */
  double cost;          /* traj_cost of the winner */
  double duration;      /* total duration of the winner: what the planner ranks by */
} topay_record_t;
topay_status topay_comm_unique_id(topay_comm_id_t* id);
topay_status topay_comm_init(topay_ctx* ctx, const topay_comm_id_t* id, int world, int rank);
topay_status topay_comm_destroy(topay_ctx* ctx);
/* One record per scenario of the solved batch resident in ctx, in order of first appearance of its id in scenario_of
 * (batch entries); winner_index (may be NULL) receives the batch index of each scenario's winner or -1. */
topay_status topay_scenario_records(topay_ctx* ctx, const int* scenario_of, int cap_records, topay_record_t* records, int* n_records,
                                    int* winner_index);
/* ncclAllGather of per_rank records from every rank (n_mine <= per_rank valid ones); all[world x per_rank] receives them
 * in rank order with the valid ones compacted to the front, *n_valid their number.  Runs on a stream of its own. */
topay_status topay_gather_records(topay_ctx* ctx, const topay_record_t* mine, int n_mine, int per_rank, topay_record_t* all, int* n_valid);
/* The two halves of that exchange for a caller with a transport of its own (MPI, gloo, a ROS topic): the block a rank
 * contributes -- its n_mine records padded to per_rank entries with scenario_id = INT32_MIN -- and the reading of the
 * world x per_rank gathered entries: valid records compacted to the front of `all` in rank order, *n_valid their number,
 * the rest of `all` padding.  topay_gather_records is pack, ncclAllGather, unpack.  No device is touched. */
topay_status topay_pack_records(const topay_record_t* mine, int n_mine, int per_rank, topay_record_t* block);
topay_status topay_unpack_records(const topay_record_t* gathered, int world, int per_rank, topay_record_t* all, int* n_valid);

/* Launch class of a candidate with n_pieces pieces: the waves its SOLVER's vectors are divided over and the decision-vector
 * elements per thread of that solver.  For parity tooling: the rounding of the solver's dot products depends on how the
 * vectors are divided over the threads (an evaluation does not: it is order-identical for any number of waves).  Since
 * round 5 every class solves on one wave -- the long classes (more than 32 pieces) run their evaluations on four waves and
 * the solver on the first of them -- so *waves is 1 for every n_pieces.  Any output pointer may be NULL. */
topay_status topay_class_of(int n_pieces, int* waves, int* elements_per_thread, int* class_index);

/* Device memory (bytes) of the batch resident in the context: every block topay_set_init_traj sized.  The variable-length
 * blocks (L-BFGS history first of all) are packed by each candidate's own size, not strided by the longest member. */
topay_status topay_workspace_bytes(topay_ctx* ctx, unsigned long long* bytes);

/* Scheduling diagnostics: how often topay_optimize_async gave up waiting (120 s) for the previously issued batch to
 * become resident before issuing this context's batch (the dispatch gate only orders batches; results never depend on
 * it).  Non-zero means the device was stalled by something else; topay_last_error() then holds the message. */
topay_status topay_gate_timeouts(topay_ctx* ctx, int* n);

/* Device time (HIP events on the context's stream) of the last topay_optimize / topay_eval_batch
 * solve kernel(s), in milliseconds, and the number of launches it covered. */
topay_status topay_last_kernel_ms(topay_ctx* ctx, double* ms, int* launches);
/* How many of those launches ran on the helper-wave kernels (topay_set_latency_mode). */
topay_status topay_last_helper_launches(topay_ctx* ctx, int* n);

#ifdef __cplusplus
}
#endif
#endif /* TOPAY_H */
