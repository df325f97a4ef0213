// Minimal C++ caller of the C-ABI (include/topay.h): what the reference-side adapter of INTEGRATION.md does, without
// Eigen/ROS.  An obstacle-free 20 x 20 x 1.6 m map (distance fields built on the device from empty occupancy grids)
// and two straight-line candidates; prints success, cost, pieces, feasibility and the end point of each, then the
// scenario's record and its all-gather over RCCL (the multi-GPU exchange, world size 1), then the front-end chain
// start / goal -> 2-D path -> dense path -> joint-space search -> trajectory around a wall.
//   g++ -std=c++17 -Iinclude examples/cabi_demo.cpp -o /tmp/cabi_demo topay_amd/lib/libtopay_hip.so -Wl,-rpath,$PWD/topay_amd/lib
// `cabi_demo --exchange <rank> <world> <id file>`: only the record exchange, one process per GPU of a node (rank r on device
// r): rank 0 creates the communicator id and leaves it in the file, the others wait for it -- the out-of-band step a planner
// does over whatever it has (a file, a pipe, a ROS parameter).  Ranks contribute different numbers of records.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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

static int run_exchange(int rank, int world, const char* id_file) {
  topay_params_t p;
  CHECK(topay_default_params(&p));
  topay_ctx* ctx = nullptr;
  CHECK(topay_create(&p, rank, &ctx));
  topay_comm_id_t id;
  if (rank == 0) {
    CHECK(topay_comm_unique_id(&id));
    const std::string tmp = std::string(id_file) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
    std::fclose(f);
    if (std::rename(tmp.c_str(), id_file) != 0) { std::fprintf(stderr, "cannot publish %s\n", id_file); return 1; }
  } else {
    bool got = false;
    for (int tries = 0; tries < 600 && !got; tries++) {   // up to a minute
      if (FILE* f = std::fopen(id_file, "rb")) {
        got = std::fread(&id, sizeof(id), 1, f) == 1;
        std::fclose(f);
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (!got) { std::fprintf(stderr, "rank %d: no communicator id in %s\n", rank, id_file); return 1; }
  }
  CHECK(topay_comm_init(ctx, &id, world, rank));
  // rank r holds 2 + r records with ids 100 r + i; blocks of per_rank = 2 + world entries
  const int n_mine = 2 + rank, per_rank = 2 + world;
  std::vector<topay_record_t> mine(n_mine), all((size_t)world * per_rank);
  for (int i = 0; i < n_mine; i++) {
    mine[i].scenario_id = 100 * rank + i;
    mine[i].best_candidate = i % 3 - 1;
    mine[i].status = mine[i].best_candidate >= 0;
    mine[i].n_pieces = 4 + i;
    mine[i].cost = 10.0 * rank + i;
    mine[i].duration = 5.5 + rank;
  }
  int n_all = 0, bad = 0, expect = 0;
  for (int round = 0; round < 3; round++) {   // the exchange is per step: the communicator is reused
    CHECK(topay_gather_records(ctx, mine.data(), n_mine, per_rank, all.data(), &n_all));
    expect = 0;
    for (int r = 0; r < world; r++)
      for (int i = 0; i < 2 + r; i++, expect++) {
        const topay_record_t& q = all[expect];
        if (expect >= n_all || q.scenario_id != 100 * r + i || q.best_candidate != i % 3 - 1 || q.n_pieces != 4 + i || q.cost != 10.0 * r + i ||
            q.duration != 5.5 + r)
          bad++;
      }
    if (n_all != expect) bad++;
  }
  std::printf("rank %d of %d on device %d: record gather over RCCL, %d records from %d ranks: %s\n", rank, world, rank, n_all, world,
              bad ? "MISMATCH" : "ok");
  CHECK(topay_comm_destroy(ctx));
  topay_destroy(ctx);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc == 5 && std::strcmp(argv[1], "--exchange") == 0) return run_exchange(std::atoi(argv[2]), std::atoi(argv[3]), argv[4]);
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
  // The planner's front-end in front of the solve (planner.cpp:816-885), every step on the device: a wall across the map with
  // a gap, the direct 2-D path around it (GraphSearch::plan2dJPS), its dense path (getDensePath), the layered joint-space
  // search along it (MCRRTs::plan), and the result as the init path of optimizeTraj.
  {
    std::fill(occ2.begin(), occ2.end(), 0);
    std::fill(occ3.begin(), occ3.end(), 0);
    for (int y = 0; y < ny; y++) {
      if (y >= 150 && y < 170) continue;   // the gap: y in [5, 7) m
      for (int x = 98; x < 102; x++) {     // the wall: x in [-0.2, 0.2) m
        occ2[(size_t)x * ny + y] = 1;
        for (int z = 0; z < nz; z++) occ3[((size_t)x * ny + y) * nz + z] = 1;
      }
    }
    CHECK(topay_build_esdf(ctx, 1, &d, occ2.data(), occ3.data()));
    const int map1 = 1;
    double start[10] = {-4.0, -3.0, 0.3, 0.0, 0.6, 0.0, 1.2, 0.0, 0.6, 0.0}, end[10] = {4.0, -2.0, -0.5, 0.4, 0.3, -0.2, 1.0, 0.3, 0.8, 0.5};
    const int cap = 256;
    int jps_len = 0, jps_stats[2];
    std::vector<double> jps_xy((size_t)cap * 2);
    CHECK(topay_plan2d_jps(ctx, 1, &map1, start, end, p.chassis_colli_radius + 0.1, cap, &jps_len, jps_xy.data(), jps_stats));
    int dense_len = 0;
    std::vector<double> dense((size_t)cap * 4);
    if (jps_len >= 2 && jps_len <= cap)
      CHECK(topay_dense_path(ctx, 1, &jps_len, jps_xy.data(), 1.414, &start[2], &end[2], p.max_v, p.max_w, cap, &dense_len, dense.data()));
    int wb_len = 0, mstats[8] = {0};
    std::vector<double> wb((size_t)cap * 10);
    if (dense_len >= 2 && dense_len <= cap) {
      end[2] = dense[4 * (dense_len - 1) + 2];   // the dense path's last yaw: the goal's, normalised to its predecessor
      topay_mcrrt_params_t mc;
      topay_mcrrt_default_params(&mc);
      CHECK(topay_mcrrt_plan(ctx, 1, &map1, &dense_len, dense.data(), start, end, &mc, 0, cap, &wb_len, wb.data(), mstats, nullptr));
    }
    std::printf("front-end: JPS %d points (%d nodes expanded), dense path %d poses, joint search status %d (%d nodes, %d iterations), %d states\n",
                jps_len, jps_stats[0], dense_len, mstats[0], mstats[1], mstats[2], wb_len);
    if (wb_len >= 2) {
      CHECK(topay_set_init_traj(ctx, 1, &wb_len, wb.data(), nullptr, nullptr, &map1));
      CHECK(topay_optimize(ctx));
      int ok1 = 0, np1 = 0, feas1 = 0;
      double c1 = 0.0, t1 = 0.0;
      CHECK(topay_get_batch(ctx, &ok1, &c1, &np1));
      CHECK(topay_check_feasible(ctx, &feas1));
      CHECK(topay_get_total_durations(ctx, &t1));
      std::printf("front-end: trajectory success %d pieces %d duration %.3f s feasible %d\n", ok1, np1, t1, feas1);
      if (!ok1) bad++;
    } else {
      bad++;
    }
  }
  topay_destroy(ctx);
  return bad ? 2 : 0;
}
