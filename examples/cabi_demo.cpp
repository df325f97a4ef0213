// Minimal C++ caller of the C-ABI (include/topay.h): what the reference-side adapter of INTEGRATION.md does, without
// Eigen/ROS.  An obstacle-free 20 x 20 x 1.6 m map (distance fields built on the device from empty occupancy grids)
// and two straight-line candidates; prints success, cost, pieces, feasibility and the end point of each, then the
// scenario's record and its all-gather over RCCL (the multi-GPU exchange, world size 1).
//   g++ -std=c++17 -Iinclude examples/cabi_demo.cpp -o /tmp/cabi_demo topay_amd/lib/libtopay_hip.so -Wl,-rpath,$PWD/topay_amd/lib
#include <cmath>
#include <cstdio>
#include <vector>

#include "topay.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    topay_status s_ = (call);                                              \
    if (s_ != TOPAY_OK) {                                                  \
      std::fprintf(stderr, "%s -> %d: %s\n", #call, s_, topay_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  topay_params_t p;
  CHECK(topay_default_params(&p));
  topay_ctx* ctx = nullptr;
  CHECK(topay_create(&p, 0, &ctx));

  // map: one occupied voxel far away in a corner so that the distance fields are finite everywhere
  topay_map_desc_t d;
  const int nx = 200, ny = 200, nz = 16;
  d.resolution = 0.1;
  d.dims[0] = nx; d.dims[1] = ny; d.dims[2] = nz;
  d.origin[0] = -10.0; d.origin[1] = -10.0; d.origin[2] = 0.0;
  d.min_boundary[0] = -10.0; d.min_boundary[1] = -10.0; d.min_boundary[2] = 0.0;
  d.max_boundary[0] = 10.0; d.max_boundary[1] = 10.0; d.max_boundary[2] = 1.6;
  std::vector<signed char> occ2((size_t)nx * ny, 0), occ3((size_t)nx * ny * nz, 0);
  occ2[0] = 1;
  occ3[0] = 1;
  CHECK(topay_build_esdf(ctx, 0, &d, occ2.data(), occ3.data()));

  // two candidates: straight lines of 11 states from (-3, -2) to (3, 1) and to (2, 3), arm tucked
  const int batch = 2, len = 11;
  std::vector<int> path_len(batch, len);
  std::vector<double> paths((size_t)batch * len * 10, 0.0);
  const double goal[2][2] = {{3.0, 1.0}, {2.0, 3.0}};
  for (int b = 0; b < batch; b++) {
    const double x0 = -3.0, y0 = -2.0, th = std::atan2(goal[b][1] - y0, goal[b][0] - x0);
    for (int i = 0; i < len; i++) {
      double* s = &paths[((size_t)b * len + i) * 10];
      const double a = (double)i / (len - 1);
      s[0] = x0 + a * (goal[b][0] - x0);
      s[1] = y0 + a * (goal[b][1] - y0);
      s[2] = th;
      s[4] = 0.6; s[6] = 1.2; s[8] = 0.6;   // q2, q4, q6: a folded pose clear of the chassis
    }
  }
  CHECK(topay_set_init_traj(ctx, batch, path_len.data(), paths.data(), nullptr, nullptr, nullptr));
  CHECK(topay_optimize(ctx));

  std::vector<int> ok(batch), feasible(batch), n_pieces(batch);
  std::vector<double> cost(batch), total(batch);
  CHECK(topay_get_batch(ctx, ok.data(), cost.data(), n_pieces.data()));
  CHECK(topay_check_feasible(ctx, feasible.data()));
  CHECK(topay_get_total_durations(ctx, total.data()));
  int bad = 0;
  for (int b = 0; b < batch; b++) {
    std::vector<double> T(n_pieces[b]), coef((size_t)n_pieces[b] * 54), knots((size_t)(n_pieces[b] + 1) * 2);
    CHECK(topay_get_result(ctx, b, nullptr, nullptr, nullptr, T.data(), coef.data(), knots.data()));
    const double ex = knots[2 * n_pieces[b]] - goal[b][0], ey = knots[2 * n_pieces[b] + 1] - goal[b][1];
    std::printf("candidate %d: success %d cost %.6f pieces %d duration %.3f s feasible %d end-point error %.2e m\n", b, ok[b],
                cost[b], n_pieces[b], total[b], feasible[b], std::sqrt(ex * ex + ey * ey));
    if (!ok[b] || std::sqrt(ex * ex + ey * ey) > 0.01) bad++;
  }
  // What a sharded planner does with a solved batch: one 32-byte record per scenario (the winner: shortest duration
  // among the candidates that succeeded and passed the gate, planner.cpp:999-1010), all-gathered over RCCL.  World size 1
  // here -- the same calls with (world, rank) of the process on a node with several GPUs; rank 0's id travels out of band.
  const int scenario_of[batch] = {41, 41};            // both candidates belong to one planning call
  topay_record_t rec[2], all[2];
  int n_rec = 0, n_all = 0, winner[2] = {-1, -1};
  CHECK(topay_scenario_records(ctx, scenario_of, 2, rec, &n_rec, winner));
  std::printf("scenario %d: winner candidate %d, status %d, duration %.3f s\n", rec[0].scenario_id, rec[0].best_candidate, rec[0].status,
              rec[0].duration);
  if (n_rec != 1 || rec[0].scenario_id != 41) bad++;
  topay_comm_id_t id;
  const topay_status cs = topay_comm_unique_id(&id);
  if (cs == TOPAY_OK) {
    CHECK(topay_comm_init(ctx, &id, 1, 0));
    CHECK(topay_gather_records(ctx, rec, n_rec, 2, all, &n_all));
    const bool same = n_all == 1 && all[0].scenario_id == rec[0].scenario_id && all[0].best_candidate == rec[0].best_candidate &&
                      all[0].status == rec[0].status && all[0].n_pieces == rec[0].n_pieces && all[0].duration == rec[0].duration;
    std::printf("record gather over RCCL (world 1): %s\n", same ? "ok" : "MISMATCH");
    if (!same) bad++;
    CHECK(topay_comm_destroy(ctx));
  } else {
    std::printf("record gather skipped: %s\n", topay_last_error());
  }
  topay_destroy(ctx);
  return bad ? 2 : 0;
}
