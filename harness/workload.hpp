// Synthetic benchmark workload for the (s,theta) trajectory NLP: seeded maps, ESDF
// construction, start/goal scenarios and front-end stand-in init paths.
//
// This is HARNESS code (CPU, runs once per map / scenario, never timed): it produces the
// inputs the C-ABI consumes (esdf2d/esdf3d buffers, ragged init_paths).  It restates the
// reference's harness-side pieces so inputs have the reference's statistics:
//   map generators   simulator/random_map_generator/src/random_map_generator.cpp:207-325 (tables),
//                    342-443 (cuboids); include/random_map_generator/random_map.hpp:23-86 (Box)
//   occupancy fill   map/src/grid_map.cpp:716-798
//   ESDF (EDT)       map/src/grid_map.cpp:89-123 (fillESDF), 125-207 (2-D), 425-521 (3-D)
//   scenario sampler planner/src/planner.cpp:494-548; map/include/map/grid_map.h:613-650
//   dense path       planner/src/graph_search.cpp:119-176
// The reference seeds everything from std::random_device (not reproducible); here every
// generator takes an explicit seed and uses mt19937_64 with a hand-rolled uniform so the
// stream does not depend on the C++ library's distribution implementation.
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <queue>
#include <random>
#include <thread>
#include <vector>

namespace topay_wl {

struct Rng {
  std::mt19937_64 eng;
  explicit Rng(uint64_t seed) : eng(seed) {}
  double uni() { return (double)(eng() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
  double uni(double a, double b) { return a + (b - a) * uni(); }
  int uni_int(int a, int b) { return a + (int)(eng() % (uint64_t)(b - a + 1)); }
};

struct PtF { float x, y, z; };  // pcl::PointXYZ is float32

// random_map.hpp:23-86 — theta is always 0 in both generators, so boxes are axis aligned
struct Box {
  double pos[3], size[3];
  Box(double px, double py, double pz, double sx, double sy, double sz) {
    pos[0] = px; pos[1] = py; pos[2] = pz; size[0] = sx; size[1] = sy; size[2] = sz;
  }
  bool overlap2d(const Box& o) const {  // SAT on the two shared axes; touching counts (random_map.hpp:75)
    for (int a = 0; a < 2; a++) {
      double min1 = pos[a], max1 = pos[a] + size[a], min2 = o.pos[a], max2 = o.pos[a] + o.size[a];
      if (max1 < min2 || max2 < min1) return false;
    }
    return true;
  }
  bool overlap(const Box& o) const {  // random_map.hpp:80-84
    return overlap2d(o) && (pos[2] + size[2] > o.pos[2] && pos[2] < o.pos[2] + o.size[2]);
  }
  // random_map_generator.cpp:6-31 (theta = 0)
  void generatePCL(double resolution, std::vector<PtF>& out) const {
    int x_num = (int)std::ceil(size[0] / resolution), y_num = (int)std::ceil(size[1] / resolution),
        z_num = (int)std::ceil(size[2] / resolution);
    for (int i = 0; i < x_num; i++)
      for (int j = 0; j < y_num; j++)
        for (int k = 0; k < z_num; k++) {
          PtF pt;
          pt.x = (float)(i * resolution);
          pt.y = (float)(j * resolution);
          pt.z = (float)(k * resolution);
          double x = pt.x * 1.0 - pt.y * 0.0;
          double y = pt.x * 0.0 + pt.y * 1.0;
          pt.x = (float)(x + pos[0]);
          pt.y = (float)(y + pos[1]);
          pt.z = (float)(pt.z + pos[2]);
          out.push_back(pt);
        }
  }
};

struct MapGenParams {  // params/map_tables.yaml, map_cuboids.yaml
  int obs_num[2] = {40, 80};
  double wall_size_range[2] = {0.2, 0.8};
  double wall_height_range[2] = {0.4, 1.5};
  double float_size_range[2] = {0.3, 0.6};
  double float_height_range[2] = {0.4, 0.8};
  double resolution = 0.05;  // cloud resolution
  double size_x = 20.0, size_y = 20.0;
  double desk_length_range[2] = {0.75, 1.25};
  double desk_width_range[2] = {0.75, 1.25};
  double desk_height_range[2] = {0.5, 1.0};
  int desk_arrangement_range[2] = {1, 2};
};

// random_map_generator.cpp:103-123
inline void generateBoxCloud(const double size[3], double resolution, std::vector<PtF>& out) {
  int x_num = (int)std::ceil(size[0] / resolution), y_num = (int)std::ceil(size[1] / resolution),
      z_num = (int)std::ceil(size[2] / resolution);
  for (int i = 0; i < x_num; i++)
    for (int j = 0; j < y_num; j++)
      for (int k = 0; k < z_num; k++) out.push_back(PtF{(float)(i * resolution), (float)(j * resolution), (float)(k * resolution)});
}

// boundary walls — random_map_generator.cpp:214-233 / 349-368
inline void addBoundaryWalls(const MapGenParams& p, std::vector<PtF>& cloud) {
  const double res = p.resolution;
  std::vector<PtF> cb;
  double s1[3] = {p.size_x, res * 2.0, 1.0};
  generateBoxCloud(s1, res, cb);
  for (auto& q : cb) {
    PtF pt;
    pt.x = (float)(q.x - p.size_x / 2.0 - res);
    pt.y = (float)(q.y + p.size_y / 2.0 - res);
    pt.z = q.z;
    cloud.push_back(pt);
    pt.y = (float)(q.y - p.size_y / 2.0 - res);
    cloud.push_back(pt);
  }
  cb.clear();
  double s2[3] = {res * 2.0, p.size_y, 1.0};
  generateBoxCloud(s2, res, cb);
  for (auto& q : cb) {
    PtF pt;
    pt.x = (float)(q.x + p.size_x / 2.0 - res);
    pt.y = (float)(q.y - p.size_y / 2.0 - res);
    pt.z = q.z;
    cloud.push_back(pt);
    pt.x = (float)(q.x - p.size_x / 2.0 - res);
    cloud.push_back(pt);
  }
}

// one desk = 4 legs + top — random_map_generator.cpp:125-165 (theta = 0)
inline void generateDesk(double px, double py, double pz, double sx, double sy, double sz, double res,
                         std::vector<PtF>& cloud) {
  const double leg_width = 0.05, desktop_thickness = 0.05;
  double cx[4] = {px, px + (sx - leg_width), px, px + (sx - leg_width)};
  double cy[4] = {py, py, py + (sy - leg_width), py + (sy - leg_width)};
  for (int c = 0; c < 4; c++) Box(cx[c], cy[c], pz, leg_width, leg_width, sz).generatePCL(res, cloud);
  Box(px, py, sz, sx, sy, desktop_thickness).generatePCL(res, cloud);
}

// "tables" world — random_map_generator.cpp:207-325; keepouts = spawn boxes (grid_map.cpp:766-770)
inline void generateDeskCase(const MapGenParams& p, Rng& rng, const std::vector<Box>& keepouts,
                             std::vector<PtF>& cloud) {
  const double res = p.resolution;
  addBoundaryWalls(p, cloud);
  std::vector<Box> obs_boxes(keepouts.begin(), keepouts.end());
  long guard = 0;
  for (int i = 0; i < p.obs_num[0]; i++) {
    if (++guard > 2000000) break;
    double x = rng.uni(-p.size_x / 2.0, p.size_x / 2.0), y = rng.uni(-p.size_y / 2.0, p.size_y / 2.0);
    x = std::floor(x / res) * res + res / 2.0;
    y = std::floor(y / res) * res + res / 2.0;
    double size_x = rng.uni(p.desk_width_range[0], p.desk_width_range[1]);
    double size_y = rng.uni(p.desk_length_range[0], p.desk_length_range[1]);
    double height = rng.uni(p.desk_height_range[0], p.desk_height_range[1]);
    double row_arr = rng.uni_int(p.desk_arrangement_range[0], p.desk_arrangement_range[1]);
    double col_arr = rng.uni_int(p.desk_arrangement_range[0], p.desk_arrangement_range[1]);
    Box test_box(x, y, 0.0, size_x * row_arr, size_y * col_arr, height);
    bool collision = false;
    for (auto& other : obs_boxes)
      if ((collision = test_box.overlap(other))) { i--; break; }
    if (collision) continue;
    obs_boxes.push_back(test_box);
    for (int r = 0; r < (int)row_arr; r++)
      for (int c = 0; c < (int)col_arr; c++)
        generateDesk(x + r * size_x, y + c * size_y, 0.0, size_x, size_y, height, res, cloud);
  }
  for (int i = 0; i < p.obs_num[1]; i++) {
    if (++guard > 2000000) break;
    double x = rng.uni(-p.size_x / 2.0, p.size_x / 2.0), y = rng.uni(-p.size_y / 2.0, p.size_y / 2.0);
    x = std::floor(x / res) * res + res / 2.0;
    y = std::floor(y / res) * res + res / 2.0;
    double bx = rng.uni(p.wall_size_range[0], p.wall_size_range[1]);
    double by = rng.uni(p.wall_size_range[0], p.wall_size_range[1]);
    double bz = rng.uni(p.wall_height_range[0], p.wall_height_range[1]);
    Box box(x, y, 0.0, bx, by, bz);
    bool collision = false;
    for (auto& other : obs_boxes)
      if ((collision = box.overlap(other))) { i--; break; }
    if (collision) continue;
    obs_boxes.push_back(box);
    box.generatePCL(res, cloud);
  }
}

// "cuboids" world — random_map_generator.cpp:342-443
inline void generateCuboidCase(const MapGenParams& p, Rng& rng, std::vector<PtF>& cloud) {
  const double res = p.resolution;
  addBoundaryWalls(p, cloud);
  Box spawn_box(-0.5, -0.5, -0.5, 1.0, 1.0, 1.0);
  std::vector<Box> obs_boxes;
  long guard = 0;
  for (int k = 0; k < 2; k++)
    for (int j = 0; j < p.obs_num[k]; j++) {
      if (++guard > 2000000) break;
      double x = rng.uni(-p.size_x / 2.0, p.size_x / 2.0), y = rng.uni(-p.size_y / 2.0, p.size_y / 2.0);
      x = std::floor(x / res) * res + res / 2.0;
      y = std::floor(y / res) * res + res / 2.0;
      double bs[3], height = 0.0;
      if (k == 0) {
        bs[0] = rng.uni(p.wall_size_range[0], p.wall_size_range[1]);
        bs[1] = rng.uni(p.wall_size_range[0], p.wall_size_range[1]);
        bs[2] = rng.uni(p.wall_height_range[0], p.wall_height_range[1]);
      } else {
        bs[0] = rng.uni(p.float_size_range[0], p.float_size_range[1]);
        bs[1] = rng.uni(p.float_size_range[0], p.float_size_range[1]);
        bs[2] = rng.uni(p.float_size_range[0], p.float_size_range[1]);
        height = rng.uni(p.float_height_range[0], p.float_height_range[1]);
      }
      Box box(x, y, height, bs[0], bs[1], bs[2]);
      bool collision = false;
      for (auto& other : obs_boxes)
        if ((collision = box.overlap(other))) break;
      if (collision || box.overlap2d(spawn_box)) { j--; continue; }
      obs_boxes.push_back(box);
      std::vector<PtF> cb;
      box.generatePCL(res, cb);
      for (auto& pt : cb) {
        float free_range = 0.5f;
        if (pt.x > -free_range && pt.x < free_range && pt.y > -free_range && pt.y < free_range) continue;
        cloud.push_back(pt);
      }
    }
}

// Felzenszwalb lower-envelope 1-D pass — grid_map.cpp:89-123
template <typename FG, typename FS>
inline void fillESDF(FG f_get_val, FS f_set_val, int start, int end, int size, int* v, double* z) {
  (void)size;
  int k = start;
  v[start] = start;
  z[start] = -std::numeric_limits<double>::max();
  z[start + 1] = std::numeric_limits<double>::max();
  for (int q = start + 1; q <= end; q++) {
    k++;
    double s;
    do {
      k--;
      s = ((f_get_val(q) + q * q) - (f_get_val(v[k]) + v[k] * v[k])) / (2 * q - 2 * v[k]);
    } while (s <= z[k]);
    k++;
    v[k] = q;
    z[k] = s;
    z[k + 1] = std::numeric_limits<double>::max();
  }
  k = start;
  for (int q = start; q <= end; q++) {
    while (z[k + 1] < q) k++;
    double val = (q - v[k]) * (q - v[k]) + f_get_val(v[k]);
    f_set_val(q, val);
  }
}

struct GridMap {
  double map_size[3] = {20.0, 20.0, 1.6};
  double resolution = 0.1, resolution_inv = 10.0;
  double min_b[3], max_b[3], origin[3];
  int voxel_num[3];
  double chassis_height = 0.155;
  std::vector<char> occ2d, occ3d;
  std::vector<double> esdf2d, esdf3d;

  void init(double sx, double sy, double sz, double res) {  // grid_map.cpp:6-66
    map_size[0] = sx; map_size[1] = sy; map_size[2] = sz;
    resolution = res;
    for (int i = 0; i < 3; i++) { min_b[i] = -map_size[i] / 2.0; max_b[i] = map_size[i] / 2.0; }
    min_b[2] = 0.0;
    max_b[2] = map_size[2];
    for (int i = 0; i < 3; i++) origin[i] = min_b[i];
    resolution_inv = 1.0 / resolution;
    for (int i = 0; i < 3; i++) voxel_num[i] = (int)std::ceil(map_size[i] / resolution);
    occ2d.assign((size_t)voxel_num[0] * voxel_num[1], 0);
    occ3d.assign((size_t)voxel_num[0] * voxel_num[1] * voxel_num[2], 0);
  }
  size_t addr2(int x, int y) const { return (size_t)x * voxel_num[1] + y; }
  size_t addr3(int x, int y, int z) const { return (size_t)x * voxel_num[1] * voxel_num[2] + (size_t)y * voxel_num[2] + z; }

  // occupancy fill — grid_map.cpp:733-747 / 778-792
  void fillOccupancy(const std::vector<PtF>& pc) {
    std::fill(occ2d.begin(), occ2d.end(), 0);
    std::fill(occ3d.begin(), occ3d.end(), 0);
    for (auto& p : pc) {
      int ix = (int)std::floor(((double)p.x - origin[0]) * resolution_inv);
      int iy = (int)std::floor(((double)p.y - origin[1]) * resolution_inv);
      int iz = (int)std::floor(((double)p.z - origin[2]) * resolution_inv);
      bool in2 = !(ix < 0 || iy < 0 || ix > voxel_num[0] - 1 || iy > voxel_num[1] - 1);
      if (in2 && p.z < chassis_height) occ2d[addr2(ix, iy)] = 1;
      if (in2 && !(iz < 0 || iz > voxel_num[2] - 1)) occ3d[addr3(ix, iy, iz)] = 1;
    }
  }

  // One signed 2-D field from an occupancy grid (1 = seed of the positive part, 0 = seed of the negative part): the
  // construction grid_map.cpp:125-207 applies to occ_buffer_2d and, with other seeds, at 211-279, 283-351 and 355-423.
  void signedField2d(const std::vector<char>& occ, std::vector<double>& out) const {
    const int rows = voxel_num[0], cols = voxel_num[1];
    const int mx = std::max(rows, cols);
    std::vector<double> tmp((size_t)rows * cols), dist((size_t)rows * cols), neg((size_t)rows * cols);
    std::vector<int> v(mx + 2);
    std::vector<double> z(mx + 2);
    const double DMAX = std::numeric_limits<double>::max();
    for (int pass = 0; pass < 2; pass++) {
      std::vector<double>& outbuf = pass == 0 ? dist : neg;
      for (int x = 0; x < rows; x++)
        fillESDF([&](int y) { return (occ[addr2(x, y)] == 1) == (pass == 0) ? 0.0 : DMAX; },
                 [&](int y, double val) { tmp[addr2(x, y)] = val; }, 0, cols - 1, cols, v.data(), z.data());
      for (int y = 0; y < cols; y++)
        fillESDF([&](int x) { return tmp[addr2(x, y)]; },
                 [&](int x, double val) { outbuf[addr2(x, y)] = resolution * std::sqrt(val); }, 0, rows - 1, rows,
                 v.data(), z.data());
    }
    out.assign((size_t)rows * cols, 0.0);
    for (size_t i = 0; i < out.size(); i++) {
      out[i] = dist[i];
      if (neg[i] > 0.0) out[i] += (-neg[i] + resolution);
    }
  }
  // The two front-end fields of updateESDF: esdf_buffer_2d_inflate (grid_map.cpp:355-423) and esdf_buffer_2d_critical
  // (211-279, overwritten by its own inflation at 283-351).  occ_crit = occ_buffer_2d_critical (733-747).
  void frontEndFields(const std::vector<char>& occ_crit, double chassis_radius, std::vector<double>& inflate,
                      std::vector<double>& critical) const {
    std::vector<char> seeds(esdf2d.size());
    for (size_t i = 0; i < seeds.size(); i++) seeds[i] = esdf2d[i] < chassis_radius ? 1 : 0;
    signedField2d(seeds, inflate);
    std::vector<double> crit0;
    signedField2d(occ_crit, crit0);
    for (size_t i = 0; i < seeds.size(); i++) seeds[i] = crit0[i] < chassis_radius ? 1 : 0;
    signedField2d(seeds, critical);
  }

  // 2-D signed EDT — grid_map.cpp:125-207
  void updateESDF2d() {
    const int rows = voxel_num[0], cols = voxel_num[1];
    const int mx = std::max(rows, cols);
    std::vector<double> tmp((size_t)rows * cols), dist((size_t)rows * cols), neg((size_t)rows * cols);
    std::vector<int> v(mx + 2);
    std::vector<double> z(mx + 2);
    const double DMAX = std::numeric_limits<double>::max();
    for (int pass = 0; pass < 2; pass++) {
      std::vector<double>& outbuf = pass == 0 ? dist : neg;
      for (int x = 0; x < rows; x++)
        fillESDF([&](int y) { return (occ2d[addr2(x, y)] == 1) == (pass == 0) ? 0.0 : DMAX; },
                 [&](int y, double val) { tmp[addr2(x, y)] = val; }, 0, cols - 1, cols, v.data(), z.data());
      for (int y = 0; y < cols; y++)
        fillESDF([&](int x) { return tmp[addr2(x, y)]; },
                 [&](int x, double val) { outbuf[addr2(x, y)] = resolution * std::sqrt(val); }, 0, rows - 1, rows,
                 v.data(), z.data());
    }
    esdf2d.assign((size_t)rows * cols, 0.0);
    for (size_t i = 0; i < esdf2d.size(); i++) {
      esdf2d[i] = dist[i];
      if (neg[i] > 0.0) esdf2d[i] += (-neg[i] + resolution);
    }
  }

  // 3-D signed EDT — grid_map.cpp:425-521.  Threaded over independent 1-D lines and run in two
  // sweeps (pos, neg) that share two scratch volumes so the 4 GB config-5 map stays within RAM.
  void updateESDF3d(int nthreads = 0) {
    const int nx = voxel_num[0], ny = voxel_num[1], nz = voxel_num[2];
    const size_t total = (size_t)nx * ny * nz;
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    std::vector<double> tmp1(total), tmp2(total);
    esdf3d.assign(total, 0.0);
    const double DMAX = std::numeric_limits<double>::max();
    auto parallel_for = [&](int n, const std::function<void(int, int, std::vector<int>&, std::vector<double>&)>& fn) {
      std::vector<std::thread> th;
      int chunk = (n + nthreads - 1) / nthreads;
      for (int t = 0; t < nthreads; t++) {
        int lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([&, lo, hi]() {
          int mx = std::max(nx, std::max(ny, nz));
          std::vector<int> v(mx + 2);
          std::vector<double> z(mx + 2);
          fn(lo, hi, v, z);
        });
      }
      for (auto& t : th) t.join();
    };
    for (int pass = 0; pass < 2; pass++) {
      parallel_for(nx, [&](int lo, int hi, std::vector<int>& v, std::vector<double>& z) {
        for (int x = lo; x < hi; x++)
          for (int y = 0; y < ny; y++)
            fillESDF([&](int zz) { return (occ3d[addr3(x, y, zz)] == 1) == (pass == 0) ? 0.0 : DMAX; },
                     [&](int zz, double val) { tmp1[addr3(x, y, zz)] = val; }, 0, nz - 1, nz, v.data(), z.data());
      });
      parallel_for(nx, [&](int lo, int hi, std::vector<int>& v, std::vector<double>& z) {
        for (int x = lo; x < hi; x++)
          for (int zz = 0; zz < nz; zz++)
            fillESDF([&](int y) { return tmp1[addr3(x, y, zz)]; },
                     [&](int y, double val) { tmp2[addr3(x, y, zz)] = val; }, 0, ny - 1, ny, v.data(), z.data());
      });
      parallel_for(ny, [&](int lo, int hi, std::vector<int>& v, std::vector<double>& z) {
        for (int y = lo; y < hi; y++)
          for (int zz = 0; zz < nz; zz++)
            fillESDF([&](int x) { return tmp2[addr3(x, y, zz)]; },
                     [&](int x, double val) {
                       double dd = resolution * std::sqrt(val);
                       if (pass == 0) esdf3d[addr3(x, y, zz)] = dd;
                       else if (dd > 0.0) esdf3d[addr3(x, y, zz)] += (-dd + resolution);
                     },
                     0, nx - 1, nx, v.data(), z.data());
      });
    }
  }

  // ---- queries used by the harness (value-only forms, grid_map.h:256-362, 727-885)
  bool isInMap2d(double px, double py) const {
    if (px < min_b[0] + 1e-4 || py < min_b[1] + 1e-4) return false;
    if (px > max_b[0] - 1e-4 || py > max_b[1] - 1e-4) return false;
    return true;
  }
  bool isInMap3d(double px, double py, double pz) const {
    if (px < min_b[0] + 1e-4 || py < min_b[1] + 1e-4 || pz < min_b[2] + 1e-4) return false;
    if (px > max_b[0] - 1e-4 || py > max_b[1] - 1e-4 || pz > max_b[2] - 1e-4) return false;
    return true;
  }
  int bnd(int id, int a) const { return std::max(std::min(id, voxel_num[a] - 1), 0); }
  double getDistance2d(double px, double py) const {
    if (!isInMap2d(px, py)) return 1e+10;
    double p[2] = {px, py}, diff[2];
    int idx[2];
    for (int a = 0; a < 2; a++) {
      double pm = p[a] - 0.5 * resolution;
      idx[a] = (int)std::floor((pm - origin[a]) * resolution_inv);
      double ip = (idx[a] + 0.5) * resolution + origin[a];
      diff[a] = (p[a] - ip) * resolution_inv;
    }
    double v[2][2];
    for (int x = 0; x < 2; x++)
      for (int y = 0; y < 2; y++) v[x][y] = esdf2d[addr2(bnd(idx[0] + x, 0), bnd(idx[1] + y, 1))];
    double v0 = v[0][0] * (1 - diff[0]) + v[1][0] * diff[0];
    double v1 = v[0][1] * (1 - diff[0]) + v[1][1] * diff[0];
    return v0 * (1 - diff[1]) + v1 * diff[1];
  }
  double getDistance3d(double px, double py, double pz) const {
    if (!isInMap3d(px, py, pz)) return 1e+10;
    double p[3] = {px, py, pz}, diff[3];
    int idx[3];
    for (int a = 0; a < 3; a++) {
      double pm = p[a] - 0.5 * resolution;
      idx[a] = (int)std::floor((pm - origin[a]) * resolution_inv);
      double ip = (idx[a] + 0.5) * resolution + origin[a];
      diff[a] = (p[a] - ip) * resolution_inv;
    }
    double v[2][2][2];
    for (int x = 0; x < 2; x++)
      for (int y = 0; y < 2; y++)
        for (int zz = 0; zz < 2; zz++) v[x][y][zz] = esdf3d[addr3(bnd(idx[0] + x, 0), bnd(idx[1] + y, 1), bnd(idx[2] + zz, 2))];
    double v00 = v[0][0][0] * (1 - diff[0]) + v[1][0][0] * diff[0];
    double v01 = v[0][0][1] * (1 - diff[0]) + v[1][0][1] * diff[0];
    double v10 = v[0][1][0] * (1 - diff[0]) + v[1][1][0] * diff[0];
    double v11 = v[0][1][1] * (1 - diff[0]) + v[1][1][1] * diff[0];
    double v0 = v00 * (1 - diff[1]) + v10 * diff[1];
    double v1 = v01 * (1 - diff[1]) + v11 * diff[1];
    return v0 * (1.0 - diff[2]) + v1 * diff[2];
  }
  bool isCollision2d(double px, double py, double thr) const {  // grid_map.h:511-536
    if (isInMap2d(px, py)) return getDistance2d(px, py) < thr;
    return true;
  }
  bool isCollision3d(double px, double py, double pz, double thr) const {  // grid_map.h:705-725
    if (isInMap3d(px, py, pz)) return getDistance3d(px, py, pz) < thr;
    return true;
  }
};

// Robot constants needed by the scenario sampler (fake_moma/moma_param.h:36-144, 203-247)
struct RobotModel {
  double chassis_height = 0.155, chassis_colli_radius = 0.4;
  double colli_length[8] = {0.139, 0.1015, 0.1525, 0.1035, 0.1285, 0.0815, 0.144, 0.05};
  double colli_points[16] = {0.139 - 0.09, 0.139, 0.0, 0.1015, 0.1525 - 0.08, 0.1525, 0.0, 0.1035,
                             0.1285 - 0.07, 0.1285, 0.0, 0.0815, 0.144 - 0.07, 0.144, 0.0, 0.1};
  double colli_radius[16] = {0.06, 0.06, 0.0, 0.08, 0.055, 0.055, 0.0, 0.07,
                             0.055, 0.055, 0.0, 0.06, 0.055, 0.055, 0.0, 0.08};  // after the 0.055 floor
  double qmin[7] = {-3.1, -2.26, -3.1, -2.355, -3.1, -2.23, -6.28};
  double qmax[7] = {3.1, 2.26, 3.1, 2.355, 3.1, 2.23, 6.28};
  double rel_t[3] = {0.0, 0.115, 0.016};
  double rel_R[3][3] = {{0.7071068, 0.7071068, 0.0}, {-0.7071068, 0.7071068, 0.0}, {0.0, 0.0, 1.0}};
  static void mm(const double a[3][3], const double b[3][3], double r[3][3]) {
    double t[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) { t[i][j] = 0; for (int k = 0; k < 3; k++) t[i][j] += a[i][k] * b[k][j]; }
    std::memcpy(r, t, sizeof(t));
  }
  int getColliPts(const double* s, double pts[12][4]) const {
    double p[3] = {s[0], s[1], chassis_height};
    double R[3][3] = {{std::cos(s[2]), -std::sin(s[2]), 0}, {std::sin(s[2]), std::cos(s[2]), 0}, {0, 0, 1}};
    for (int i = 0; i < 3; i++) p[i] += R[i][0] * rel_t[0] + R[i][1] * rel_t[1] + R[i][2] * rel_t[2];
    mm(R, rel_R, R);
    int n = 0;
    for (int i = 0; i < 8; i++) {
      for (int j = 0; j < 2; j++) {
        if (colli_points[i * 2 + j] == 0.0) continue;
        for (int a = 0; a < 3; a++) pts[n][a] = p[a] + R[a][2] * colli_points[i * 2 + j];
        pts[n][3] = colli_radius[i * 2 + j];
        n++;
      }
      for (int a = 0; a < 3; a++) p[a] += R[a][2] * colli_length[i];
      if (i == 7) break;
      double q = s[3 + i];
      double D[3][3];
      if (i % 2 == 0) { double t[3][3] = {{std::cos(q), -std::sin(q), 0}, {std::sin(q), std::cos(q), 0}, {0, 0, 1}}; std::memcpy(D, t, sizeof(t)); }
      else { double t[3][3] = {{std::cos(q), 0, std::sin(q)}, {0, 1, 0}, {-std::sin(q), 0, std::cos(q)}}; std::memcpy(D, t, sizeof(t)); }
      mm(R, D, R);
    }
    return n;
  }
  // grid_map.h:613-650
  bool isWholeBodyCollision(const GridMap& gm, const double* s) const {
    for (int i = 0; i < 7; i++)
      if (s[3 + i] > qmax[i] || s[3 + i] < qmin[i]) return true;
    if (gm.isCollision2d(s[0], s[1], chassis_colli_radius)) return true;
    double pts[12][4];
    int n = getColliPts(s, pts);
    for (int i = 0; i < n; i++) {
      if (gm.isCollision3d(pts[i][0], pts[i][1], pts[i][2], pts[i][3])) return true;
      double dx = pts[i][0] - s[0], dy = pts[i][1] - s[1];
      if (i > 2 && pts[i][2] < chassis_height + pts[i][3] && std::sqrt(dx * dx + dy * dy) < chassis_colli_radius + pts[i][3])
        return true;
      for (int j = i + 1; j < n; j++) {
        if (j - i <= 1) continue;  // collision_matrix: self + adjacent spheres exempt (moma_param.h:128-143)
        double d0 = pts[i][0] - pts[j][0], d1 = pts[i][1] - pts[j][1], d2 = pts[i][2] - pts[j][2];
        if (std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) < pts[i][3] + pts[j][3]) return true;
      }
    }
    return false;
  }
  // The smallest slack |lhs - rhs| over every threshold comparison isWholeBodyCollision evaluates for this state (all
  // of them, without the early exits): a state whose verdict could flip under last-bit rounding differences has a
  // slack of the order of 1e-16; the parity tests only accept a differing verdict from such a tie.
  double wholeBodyTieSlack(const GridMap& gm, const double* s) const {
    double m = 1e300;
    auto cmp = [&](double a, double b) { m = std::min(m, std::fabs(a - b)); };
    for (int i = 0; i < 7; i++) { cmp(s[3 + i], qmax[i]); cmp(s[3 + i], qmin[i]); }
    if (gm.isInMap2d(s[0], s[1])) cmp(gm.getDistance2d(s[0], s[1]), chassis_colli_radius);
    double pts[12][4];
    int n = getColliPts(s, pts);
    for (int i = 0; i < n; i++) {
      if (gm.isInMap3d(pts[i][0], pts[i][1], pts[i][2])) cmp(gm.getDistance3d(pts[i][0], pts[i][1], pts[i][2]), pts[i][3]);
      double dx = pts[i][0] - s[0], dy = pts[i][1] - s[1];
      if (i > 2) { cmp(pts[i][2], chassis_height + pts[i][3]); cmp(std::sqrt(dx * dx + dy * dy), chassis_colli_radius + pts[i][3]); }
      for (int j = i + 2; j < n; j++) {
        double d0 = pts[i][0] - pts[j][0], d1 = pts[i][1] - pts[j][1], d2 = pts[i][2] - pts[j][2];
        cmp(std::sqrt(d0 * d0 + d1 * d1 + d2 * d2), pts[i][3] + pts[j][3]);
      }
    }
    return m;
  }
};

// ---- front-end stand-in ---------------------------------------------------------------
struct Cell { int x, y; };

// 8-connected A* on esdf2d >= thr (cf. planner.cpp:816: chassis radius + 0.1)
inline bool astar2d(const GridMap& gm, Cell s, Cell g, double thr, std::vector<Cell>& out) {
  const int nx = gm.voxel_num[0], ny = gm.voxel_num[1];
  auto freec = [&](int x, int y) { return x >= 0 && y >= 0 && x < nx && y < ny && gm.esdf2d[gm.addr2(x, y)] >= thr; };
  if (!freec(s.x, s.y) || !freec(g.x, g.y)) return false;
  std::vector<double> gsc((size_t)nx * ny, 1e300);
  std::vector<int> parent((size_t)nx * ny, -1);
  typedef std::pair<double, int> QE;
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> pq;
  auto h = [&](int x, int y) { double dx = std::abs(x - g.x), dy = std::abs(y - g.y); return (dx + dy) + (1.41421356237 - 2.0) * std::min(dx, dy); };
  int sid = s.x * ny + s.y, gid = g.x * ny + g.y;
  gsc[sid] = 0;
  pq.push({h(s.x, s.y), sid});
  while (!pq.empty()) {
    auto [f, id] = pq.top();
    pq.pop();
    int x = id / ny, y = id % ny;
    if (id == gid) break;
    if (f > gsc[id] + h(x, y) + 1e-9) continue;
    for (int dx = -1; dx <= 1; dx++)
      for (int dy = -1; dy <= 1; dy++) {
        if (!dx && !dy) continue;
        int xx = x + dx, yy = y + dy;
        if (!freec(xx, yy)) continue;
        double c = (dx && dy) ? 1.41421356237 : 1.0;
        int nid = xx * ny + yy;
        if (gsc[id] + c < gsc[nid] - 1e-12) {
          gsc[nid] = gsc[id] + c;
          parent[nid] = id;
          pq.push({gsc[nid] + h(xx, yy), nid});
        }
      }
  }
  if (parent[gid] < 0 && gid != sid) return false;
  std::vector<Cell> rev;
  for (int id = gid; id >= 0; id = parent[id]) { rev.push_back({id / ny, id % ny}); if (id == sid) break; }
  out.assign(rev.rbegin(), rev.rend());
  return true;
}

// Bresenham line check on the 2-D ESDF — grid_map.h:565-611
inline bool lineFree(const GridMap& gm, Cell a, Cell b, double thr) {
  int dx = std::abs(b.x - a.x), dy = std::abs(b.y - a.y);
  int sx = (a.x < b.x) ? 1 : -1, sy = (a.y < b.y) ? 1 : -1;
  int err = dx - dy, x0 = a.x, y0 = a.y;
  while (true) {
    if (gm.esdf2d[gm.addr2(x0, y0)] < thr) return false;
    if (x0 == b.x && y0 == b.y) break;
    int e2 = 2 * err;
    if (e2 > -dy) { err -= dy; x0 += sx; }
    if (e2 < dx) { err += dx; y0 += sy; }
  }
  return true;
}

inline void shortcut(const GridMap& gm, const std::vector<Cell>& in, double thr, std::vector<Cell>& out) {
  out.clear();
  size_t i = 0;
  out.push_back(in[0]);
  while (i + 1 < in.size()) {
    size_t j = in.size() - 1;
    while (j > i + 1 && !lineFree(gm, in[i], in[j], thr)) j--;
    out.push_back(in[j]);
    i = j;
  }
}

inline void normalizeAngle(double ref, double& a) {
  while (ref - a > M_PI) a += 2 * M_PI;
  while (ref - a < -M_PI) a -= 2 * M_PI;
}

// graph_search.cpp:119-176.  (No fused multiply-adds here: this restatement is also the checker of the device's
// topay_dense_path, and the reference's build has none either.)
__attribute__((optimize("fp-contract=off")))
inline std::vector<std::array<double, 4>> getDensePath(const std::vector<std::array<double, 2>>& raw, double step_size,
                                                       double start_yaw, double end_yaw, double v_max, double w_max) {
  std::vector<std::array<double, 2>> dense;
  dense.push_back(raw[0]);
  for (size_t i = 1; i < raw.size(); i++) {
    double dx = raw[i][0] - raw[i - 1][0], dy = raw[i][1] - raw[i - 1][1];
    double len = std::sqrt(dx * dx + dy * dy);
    // Eigen::normalized(): a zero-length vector stays zero (graph_search.cpp:128)
    double ux = len > 0.0 ? dx / len : 0.0, uy = len > 0.0 ? dy / len : 0.0;
    int times = (int)std::max(std::ceil(len / step_size), 1.0);
    double step = len / times;
    for (int j = 1; j <= times; j++) dense.push_back({raw[i - 1][0] + (step * j) * ux, raw[i - 1][1] + (step * j) * uy});
  }
  std::vector<std::array<double, 4>> sp;
  sp.push_back({dense[0][0], dense[0][1], start_yaw, 0.0});
  double cur_theta = std::atan2(dense[1][1] - dense[0][1], dense[1][0] - dense[0][0]);
  normalizeAngle(start_yaw, cur_theta);
  sp.back()[3] = std::fabs(cur_theta - start_yaw) / w_max;
  sp.push_back({dense[0][0], dense[0][1], cur_theta, 0.0});
  for (size_t i = 1; i + 1 < dense.size(); i++) {
    double px = dense[i][0], py = dense[i][1];
    double ax = px - sp.back()[0], ay = py - sp.back()[1];
    sp.back()[3] = std::sqrt(ax * ax + ay * ay) / v_max;
    sp.push_back({px, py, sp.back()[2], 0.0});
    cur_theta = std::atan2(dense[i + 1][1] - dense[i][1], dense[i + 1][0] - dense[i][0]);
    normalizeAngle(sp.back()[2], cur_theta);
    sp.back()[3] = std::fabs(cur_theta - sp.back()[2]) / w_max;
    sp.push_back({px, py, cur_theta, 0.0});
  }
  double px = dense.back()[0], py = dense.back()[1];
  double ax = px - sp.back()[0], ay = py - sp.back()[1];
  sp.back()[3] = std::sqrt(ax * ax + ay * ay) / v_max;
  sp.push_back({px, py, sp.back()[2], 0.0});
  cur_theta = end_yaw;
  normalizeAngle(sp.back()[2], cur_theta);
  sp.back()[3] = std::fabs(cur_theta - sp.back()[2]) / w_max;
  sp.push_back({px, py, cur_theta, 0.0});
  std::vector<std::array<double, 4>> result;
  for (size_t i = 0; i + 1 < sp.size(); i++)
    if (sp[i][3] > 1.0e-3) result.push_back(sp[i]);
  result.push_back(sp.back());
  return result;
}

struct World {
  GridMap gm;
  RobotModel robot;
  MapGenParams mp;
  int kind = 0;  // 0 tables, 1 cuboids
  uint64_t seed = 42;

  // scale: map edge multiplier for config 5 (obstacle counts scale with area)
  void build(int kind_, uint64_t seed_, double size_xy, double size_z, double res, double cloud_res,
             const std::vector<std::array<double, 2>>& keepouts, int nthreads) {
    kind = kind_;
    seed = seed_;
    mp = MapGenParams();
    mp.size_x = mp.size_y = size_xy;
    mp.resolution = cloud_res;
    double area_scale = (size_xy * size_xy) / (20.0 * 20.0);
    if (kind == 0) { mp.obs_num[0] = (int)std::lround(40 * area_scale); mp.obs_num[1] = (int)std::lround(80 * area_scale); }
    else { mp.obs_num[0] = (int)std::lround(80 * area_scale); mp.obs_num[1] = (int)std::lround(80 * area_scale); }
    gm.init(size_xy, size_xy, size_z, res);
    Rng rng(seed);
    std::vector<PtF> cloud;
    if (kind == 0) {
      std::vector<Box> ko;
      for (auto& k : keepouts) ko.push_back(Box(k[0] - 0.5, k[1] - 0.5, 0.0, 1.0, 1.0, 1.0));  // grid_map.cpp:766-770
      generateDeskCase(mp, rng, ko, cloud);
    } else {
      generateCuboidCase(mp, rng, cloud);
    }
    gm.fillOccupancy(cloud);
    if (nthreads < 0) return;   // occupancy only (the caller builds the distance fields on the device)
    gm.updateESDF2d();
    gm.updateESDF3d(nthreads);
  }

  // planner.cpp:498-512: goal then start; accept 3 <= dist <= 8
  static void sampleStartGoalXY(const GridMap& gm, Rng& rng, double lo, double hi, double start[3], double goal[3]) {
    while (true) {
      goal[0] = rng.uni(gm.min_b[0] + 2.0, gm.max_b[0] - 2.0);
      goal[1] = rng.uni(gm.min_b[1] + 2.0, gm.max_b[1] - 2.0);
      goal[2] = rng.uni(-M_PI, M_PI);
      start[0] = rng.uni(gm.min_b[0] + 2.0, gm.max_b[0] - 2.0);
      start[1] = rng.uni(gm.min_b[1] + 2.0, gm.max_b[1] - 2.0);
      start[2] = rng.uni(-M_PI, M_PI);
      double d = std::hypot(start[0] - goal[0], start[1] - goal[1]);
      if (d < lo || d > hi) continue;
      return;
    }
  }
  // planner.cpp:529-548: joints U[min,max] until no whole-body collision (1 s cap -> try cap)
  bool sampleArm(Rng& rng, double state[10], int max_tries = 2000) const {
    for (int t = 0; t < max_tries; t++) {
      for (int i = 0; i < 7; i++) state[3 + i] = (robot.qmax[i] - robot.qmin[i]) * rng.uni() + robot.qmin[i];
      if (!robot.isWholeBodyCollision(gm, state)) return true;
    }
    return false;
  }
  // scenario on an EXISTING map (cuboids, or tables map reused): rejection-sample until both ends free
  bool sampleScenario(uint64_t sseed, double start[10], double goal[10]) const {
    Rng rng(sseed);
    for (int attempt = 0; attempt < 10000; attempt++) {
      sampleStartGoalXY(gm, rng, 3.0, 8.0, start, goal);
      if (gm.isCollision2d(start[0], start[1], 0.5) || gm.isCollision2d(goal[0], goal[1], 0.5)) continue;
      if (!sampleArm(rng, goal)) continue;
      if (!sampleArm(rng, start)) continue;
      return true;
    }
    return false;
  }

  Cell toCell(double x, double y) const {
    return {(int)std::floor((x - gm.origin[0]) * gm.resolution_inv), (int)std::floor((y - gm.origin[1]) * gm.resolution_inv)};
  }
  std::array<double, 2> cellCenter(Cell c) const {
    return {(c.x + 0.5) * gm.resolution + gm.origin[0], (c.y + 0.5) * gm.resolution + gm.origin[1]};
  }

  // Candidate init paths: candidate 0 = direct A*; others detour through a seeded random free waypoint.
  // Returns states (P x 10 each) appended to `paths`, lengths to `lens`.
  // dts (optional): the dt of every dense-path entry (the fourth component of getDensePath's output, which MCRRTs::plan needs)
  int initPaths(const double start[10], const double goal[10], int n_cand, uint64_t pseed,
                std::vector<double>& paths, std::vector<int>& lens, std::vector<double>* dts = nullptr) const {
    const double thr = robot.chassis_colli_radius + 0.1;
    Rng rng(pseed);
    Cell cs = toCell(start[0], start[1]), cg = toCell(goal[0], goal[1]);
    int made = 0;
    double dsg = std::hypot(start[0] - goal[0], start[1] - goal[1]);
    auto cells_len = [&](const std::vector<Cell>& cs_) {
      double L = 0.0;
      for (size_t i = 1; i < cs_.size(); i++) L += std::hypot((double)(cs_[i].x - cs_[i - 1].x), (double)(cs_[i].y - cs_[i - 1].y));
      return L * gm.resolution;
    };
    double min_len = -1.0;  // length of the shortest (direct) candidate
    for (int c = 0; c < n_cand; c++) {
      std::vector<Cell> cells;
      bool ok = false, via = false;
      for (int tries = 0; tries < 200 && !ok; tries++) {
        cells.clear();
        if (c == 0 && tries == 0) {
          ok = astar2d(gm, cs, cg, thr, cells);
          if (!ok) break;
        } else {
          // waypoint within an ellipse-ish neighbourhood of the segment
          double mx = 0.5 * (start[0] + goal[0]), my = 0.5 * (start[1] + goal[1]);
          double r = 0.5 * dsg + 1.0;
          double wx = mx + rng.uni(-r, r), wy = my + rng.uni(-r, r);
          if (!gm.isInMap2d(wx, wy)) continue;
          Cell cw = toCell(wx, wy);
          if (cw.x < 0 || cw.y < 0 || cw.x >= gm.voxel_num[0] || cw.y >= gm.voxel_num[1]) continue;
          if (gm.esdf2d[gm.addr2(cw.x, cw.y)] < thr) continue;
          std::vector<Cell> a, b;
          if (!astar2d(gm, cs, cw, thr, a) || !astar2d(gm, cw, cg, thr, b)) continue;
          // shortcut each half separately so the detour waypoint survives (distinct candidate)
          std::vector<Cell> sa, sb;
          shortcut(gm, a, thr, sa);
          shortcut(gm, b, thr, sb);
          cells = sa;
          cells.insert(cells.end(), sb.begin() + 1, sb.end());
          // TopologyPRM::selectShortPaths keeps a candidate only if it is shorter than ratio_to_short (2.0) times the
          // shortest one (topo_prm.cpp:384-407, params/topo_prm.yaml:17)
          if (min_len > 0.0 && cells_len(cells) >= 2.0 * min_len) continue;
          via = true;
          ok = true;
        }
      }
      if (!ok) {
        if (c == 0) return 0;
        // fall back to the direct path again (still a valid candidate)
        if (!astar2d(gm, cs, cg, thr, cells)) return made;
      }
      std::vector<Cell> sc;
      if (via) sc = cells;
      else shortcut(gm, cells, thr, sc);
      if (c == 0) min_len = cells_len(sc);
      std::vector<std::array<double, 2>> raw;
      raw.push_back({start[0], start[1]});
      for (size_t i = 1; i + 1 < sc.size(); i++) raw.push_back(cellCenter(sc[i]));
      raw.push_back({goal[0], goal[1]});
      // drop zero-length segments
      std::vector<std::array<double, 2>> raw2;
      for (auto& p : raw)
        if (raw2.empty() || std::hypot(p[0] - raw2.back()[0], p[1] - raw2.back()[1]) > 1e-6) raw2.push_back(p);
      if (raw2.size() < 2) raw2.push_back({goal[0], goal[1]});
      auto dense = getDensePath(raw2, 1.414, start[2], goal[2], 1.0, 1.25);
      // joints: linear in cumulative time (mcrrts.cpp:25-33 two-node case generalised)
      double total = 0.0;
      for (size_t i = 0; i + 1 < dense.size(); i++) total += dense[i][3];
      double acc = 0.0;
      for (size_t i = 0; i < dense.size(); i++) {
        double f = total > 0 ? acc / total : (i + 1 == dense.size() ? 1.0 : 0.0);
        if (i + 1 == dense.size()) f = 1.0;
        paths.push_back(dense[i][0]);
        paths.push_back(dense[i][1]);
        paths.push_back(dense[i][2]);
        for (int q = 0; q < 7; q++) paths.push_back(start[3 + q] + f * (goal[3 + q] - start[3 + q]));
        if (dts) dts->push_back(dense[i][3]);
        acc += dense[i][3];
      }
      lens.push_back((int)dense.size());
      made++;
    }
    return made;
  }
};

}  // namespace topay_wl
