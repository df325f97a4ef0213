// C entry points of the synthetic-workload harness (see workload.hpp).  Loaded through ctypes by
// topay_amd/harness/workload.py; used by bench.py and tests to build identical inputs for the HIP
// path and the oracle.
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

}  // extern "C"
