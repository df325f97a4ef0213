// C entry points of the synthetic-workload harness (see workload.hpp).  Loaded through ctypes by
// harness/workload.py; used by bench.py and tests to build identical inputs for the HIP
// path and the oracle.
#include <atomic>

#include "workload.hpp"

using namespace topay_wl;

extern "C" {

void* wl_world_create(int kind, uint64_t seed, double size_xy, double size_z, double res, double cloud_res,
                      int n_keepouts, const double* keepouts_xy, int nthreads) {
  World* w = new World();
  std::vector<std::array<double, 2>> ko;
  for (int i = 0; i < n_keepouts; i++) ko.push_back({keepouts_xy[2 * i], keepouts_xy[2 * i + 1]});
  w->build(kind, seed, size_xy, size_z, res, cloud_res, ko, nthreads);
  return w;
}

void wl_world_destroy(void* h) { delete (World*)h; }

void wl_world_desc(void* h, int dims[3], double origin[3], double* res, double min_b[3], double max_b[3]) {
  World* w = (World*)h;
  for (int i = 0; i < 3; i++) {
    dims[i] = w->gm.voxel_num[i];
    origin[i] = w->gm.origin[i];
    min_b[i] = w->gm.min_b[i];
    max_b[i] = w->gm.max_b[i];
  }
  *res = w->gm.resolution;
}

const double* wl_world_esdf2d(void* h) { return ((World*)h)->gm.esdf2d.data(); }
const double* wl_world_esdf3d(void* h) { return ((World*)h)->gm.esdf3d.data(); }
const char* wl_world_occ2d(void* h) { return ((World*)h)->gm.occ2d.data(); }
const char* wl_world_occ3d(void* h) { return ((World*)h)->gm.occ3d.data(); }

// start/goal (x,y,theta) only, planner.cpp:498-512; map extent given explicitly so it can run before the map exists
void wl_sample_start_goal_xy(uint64_t seed, double size_xy, double start[3], double goal[3]) {
  GridMap gm;
  gm.init(size_xy, size_xy, 1.6, 1.0);
  Rng rng(seed);
  World::sampleStartGoalXY(gm, rng, 3.0, 8.0, start, goal);
}

int wl_sample_arm(void* h, uint64_t seed, double state[10]) {
  World* w = (World*)h;
  Rng rng(seed);
  return w->sampleArm(rng, state) ? 1 : 0;
}

int wl_sample_scenario(void* h, uint64_t seed, double start[10], double goal[10]) {
  return ((World*)h)->sampleScenario(seed, start, goal) ? 1 : 0;
}

int wl_whole_body_collision(void* h, const double* state) {
  World* w = (World*)h;
  return w->robot.isWholeBodyCollision(w->gm, state) ? 1 : 0;
}

double wl_whole_body_tie_slack(void* h, const double* state) {
  World* w = (World*)h;
  return w->robot.wholeBodyTieSlack(w->gm, state);
}

// returns number of candidates produced; out_paths holds sum(lens) x 10 doubles (capacity max_states)
int wl_init_paths(void* h, const double* start, const double* goal, int n_cand, uint64_t seed, double* out_paths,
                  int max_states, int* out_lens) {
  World* w = (World*)h;
  std::vector<double> paths;
  std::vector<int> lens;
  int made = w->initPaths(start, goal, n_cand, seed, paths, lens);
  size_t tot = 0;
  for (int l : lens) tot += l;
  if ((int)tot > max_states) return -1;
  std::memcpy(out_paths, paths.data(), paths.size() * sizeof(double));
  for (int i = 0; i < made; i++) out_lens[i] = lens[i];
  return made;
}

// standalone EDT entry points (used by tests: single occupied voxel => analytic distances)
void wl_edt(const char* occ2d, const char* occ3d, int nx, int ny, int nz, double res, double* esdf2d, double* esdf3d,
            int nthreads) {
  GridMap gm;
  gm.init(nx * res, ny * res, nz * res, res);
  gm.voxel_num[0] = nx; gm.voxel_num[1] = ny; gm.voxel_num[2] = nz;
  gm.occ2d.assign(occ2d, occ2d + (size_t)nx * ny);
  gm.occ3d.assign(occ3d, occ3d + (size_t)nx * ny * nz);
  gm.updateESDF2d();
  gm.updateESDF3d(nthreads);
  std::memcpy(esdf2d, gm.esdf2d.data(), gm.esdf2d.size() * sizeof(double));
  std::memcpy(esdf3d, gm.esdf3d.data(), gm.esdf3d.size() * sizeof(double));
}


// GraphSearch::getDensePath restatement (checker of topay_dense_path): returns the number of entries, writes at most cap
int wl_dense_path(const double* raw_xy, int n, double step_size, double start_yaw, double end_yaw, double v_max, double w_max, double* out,
                  int cap) {
  std::vector<std::array<double, 2>> raw((size_t)n);
  for (int i = 0; i < n; i++) raw[i] = {raw_xy[2 * i], raw_xy[2 * i + 1]};
  auto r = getDensePath(raw, step_size, start_yaw, end_yaw, v_max, w_max);
  for (size_t i = 0; i < r.size() && (int)i < cap; i++)
    for (int k = 0; k < 4; k++) out[4 * i + k] = r[i][k];
  return (int)r.size();
}

// the two front-end fields of updateESDF from the occupancy grids (occ2d_critical may be null: projection of occ3d)
void wl_edt_front_end_fields(const char* occ2d, const char* occ2d_critical, const char* occ3d, int nx, int ny, int nz, double res,
                             double chassis_radius, double* inflate, double* critical) {
  GridMap gm;
  gm.init(nx * res, ny * res, nz * res, res);
  gm.voxel_num[0] = nx; gm.voxel_num[1] = ny; gm.voxel_num[2] = nz;
  gm.occ2d.assign(occ2d, occ2d + (size_t)nx * ny);
  gm.updateESDF2d();
  std::vector<char> crit((size_t)nx * ny, 0);
  if (occ2d_critical) crit.assign(occ2d_critical, occ2d_critical + (size_t)nx * ny);
  else
    for (size_t c = 0; c < crit.size(); c++)
      for (int z = 0; z < nz; z++)
        if (occ3d[c * nz + z] == 1) crit[c] = 1;
  std::vector<double> inf, cr;
  gm.frontEndFields(crit, chassis_radius, inf, cr);
  std::memcpy(inflate, inf.data(), inf.size() * sizeof(double));
  std::memcpy(critical, cr.data(), cr.size() * sizeof(double));
}

// ---------------------------------------------------------------------------------------------
// Benchmark batch for the "tables" scene: S scenarios, each with its OWN map (the reference regenerates the
// map every episode with 1x1 m keep-outs at start and goal, planner.cpp:514-521, grid_map.cpp:755-772),
// n_cand candidate init paths per scenario.  Built by a thread pool; everything stays in the handle.
// ---------------------------------------------------------------------------------------------
struct TablesBatch {
  std::vector<World*> worlds;
  std::vector<std::array<double, 10>> starts, goals;
  std::vector<int> lens;          // per trajectory
  std::vector<int> scen;          // scenario of each trajectory
  std::vector<double> paths;      // ragged
  std::vector<double> dts;        // dt of every state (dense path)
};

// Scenarios with index >= g_keep_esdf3d drop their CPU-built 3-D distance field as soon as their init paths exist (it
// is only needed to sample the arm poses and by callers that hand the field to a CPU solver); -1 keeps all.  Saves
// 5 MB per scenario of host memory when the product builds the fields on the device from the occupancy grids.
static int g_keep_esdf3d = -1;
void wl_set_keep_esdf3d(int keep) { g_keep_esdf3d = keep; }
long long wl_world_esdf3d_size(void* h) { return (long long)((World*)h)->gm.esdf3d.size(); }

void* wl_tables_batch_create(int S, int n_cand, uint64_t base_seed, double size_xy, double size_z, double res,
                             double cloud_res, int nthreads) {
  TablesBatch* tb = new TablesBatch();
  tb->worlds.assign(S, nullptr);
  tb->starts.resize(S);
  tb->goals.resize(S);
  std::vector<std::vector<int>> lens(S);
  std::vector<std::vector<double>> paths(S), dts(S);
  if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
  std::atomic<int> next(0);
  auto worker = [&]() {
    while (true) {
      int sidx = next.fetch_add(1);
      if (sidx >= S) break;
      const uint64_t seed = base_seed + (uint64_t)sidx;
      for (int attempt = 0; attempt < 64; attempt++) {
        const uint64_t sd = seed * 1000 + attempt;
        double s3[3], g3[3];
        {
          GridMap gm;
          gm.init(size_xy, size_xy, size_z, 1.0);
          Rng rng(sd);
          World::sampleStartGoalXY(gm, rng, 3.0, 8.0, s3, g3);
        }
        World* w = new World();
        std::vector<std::array<double, 2>> ko = {{s3[0], s3[1]}, {g3[0], g3[1]}};
        w->build(0, sd, size_xy, size_z, res, cloud_res, ko, 1);
        double start[10] = {s3[0], s3[1], s3[2], 0, 0, 0, 0, 0, 0, 0}, goal[10] = {g3[0], g3[1], g3[2], 0, 0, 0, 0, 0, 0, 0};
        Rng ra(seed * 7919 + 2 * attempt), rb(seed * 7919 + 2 * attempt + 1);
        bool ok = w->sampleArm(ra, goal) && w->sampleArm(rb, start);
        std::vector<double> pth, dtv;
        std::vector<int> ln;
        if (ok) ok = w->initPaths(start, goal, n_cand, seed * 104729 + attempt, pth, ln, &dtv) == n_cand;
        if (!ok) { delete w; continue; }
        if (g_keep_esdf3d >= 0 && sidx >= g_keep_esdf3d) std::vector<double>().swap(w->gm.esdf3d);
        tb->worlds[sidx] = w;
        for (int q = 0; q < 10; q++) { tb->starts[sidx][q] = start[q]; tb->goals[sidx][q] = goal[q]; }
        lens[sidx] = ln;
        paths[sidx] = pth;
        dts[sidx] = dtv;
        break;
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(worker);
  for (auto& t : th) t.join();
  for (int sidx = 0; sidx < S; sidx++) {
    if (!tb->worlds[sidx]) continue;  // failed scenario: contributes nothing
    for (int l : lens[sidx]) { tb->lens.push_back(l); tb->scen.push_back(sidx); }
    tb->paths.insert(tb->paths.end(), paths[sidx].begin(), paths[sidx].end());
    tb->dts.insert(tb->dts.end(), dts[sidx].begin(), dts[sidx].end());
  }
  return tb;
}
void wl_tables_batch_destroy(void* h) {
  TablesBatch* tb = (TablesBatch*)h;
  for (World* w : tb->worlds) delete w;
  delete tb;
}
int wl_tables_batch_ntraj(void* h) { return (int)((TablesBatch*)h)->lens.size(); }
long long wl_tables_batch_nstates(void* h) { return (long long)((TablesBatch*)h)->paths.size() / 10; }
void wl_tables_batch_get(void* h, int* lens, int* scen, double* paths) {
  TablesBatch* tb = (TablesBatch*)h;
  std::memcpy(lens, tb->lens.data(), tb->lens.size() * sizeof(int));
  std::memcpy(scen, tb->scen.data(), tb->scen.size() * sizeof(int));
  std::memcpy(paths, tb->paths.data(), tb->paths.size() * sizeof(double));
}
// dt of every state of the batch's init paths (getDensePath's fourth component): with (x, y, theta) of the state it is the
// car path MCRRTs::plan takes
void wl_tables_batch_get_dt(void* h, double* dt) {
  TablesBatch* tb = (TablesBatch*)h;
  std::memcpy(dt, tb->dts.data(), tb->dts.size() * sizeof(double));
}
void* wl_tables_batch_world(void* h, int sidx) { return ((TablesBatch*)h)->worlds[sidx]; }

}  // extern "C"
