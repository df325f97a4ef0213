// First slice of the front-end that feeds optimizeTraj, batched on the device:
//   dense_path()      GraphSearch::getDensePath   planner/src/graph_search.cpp:119-176  (raw 2-D path -> (x, y, theta, dt))
//   connect edge check MCRRTs::connectCollision    planner/include/planner/mcrrts.h:310-348, on top of
//                      GridMap::isWholeBodyCollision (whole_body_collision(), topay_feas.h)
// The car poses along an edge come from OMPL's ReedsSheppStateSpace (distance / interpolate, mcrrts.h:318-324, 336): a
// third-party dependency of the reference that is not part of /root/reference.  The caller keeps using it for that
// geometry (a handful of flops per pose); what is batched here is everything else of the function -- the number of
// checks, the joint interpolation and the whole-body collision test of every interpolated state.
#pragma once
#include "topay_feas.h"

namespace topay {

// One path per thread (the construction is a short sequential scan).  Returns the number of entries the reference's
// result vector has; at most `cap` of them are written.
__device__ __forceinline__ int dense_path(const double* raw, int nraw, double step_size, double start_yaw, double end_yaw, double v_max,
                                          double w_max, double* out, int cap) {
  const double PI = 3.14159265358979323846;
  auto normalize = [&](double ref, double& a) {
    while (ref - a > PI) a += 2 * PI;
    while (ref - a < -PI) a -= 2 * PI;
  };
  // generator of dense_path[1], dense_path[2], ... (graph_search.cpp:124-137)
  int seg = 1, j = 1, times = 0;
  double step = 0.0, ux = 0.0, uy = 0.0;
  auto seg_setup = [&]() {
    const double dx = raw[2 * seg] - raw[2 * (seg - 1)], dy = raw[2 * seg + 1] - raw[2 * (seg - 1) + 1];
    const double len = sqrt(dx * dx + dy * dy);
    // Eigen's normalized() returns the zero vector for a zero-length input (a repeated raw point): the samples of such a
    // leg are the point itself
    ux = len > 0.0 ? dx / len : 0.0;
    uy = len > 0.0 ? dy / len : 0.0;
    const double t = ceil(len / step_size);
    times = (int)(t > 1.0 ? t : 1.0);
    step = len / times;
    j = 1;
  };
  bool more = nraw >= 2;
  if (more) seg_setup();
  auto next = [&](double& x, double& y) {
    x = raw[2 * (seg - 1)] + (step * j) * ux;
    y = raw[2 * (seg - 1) + 1] + (step * j) * uy;
    j++;
    if (j > times) {
      seg++;
      if (seg < nraw) seg_setup();
      else more = false;
    }
  };
  int cnt = 0;
  double px = raw[0], py = raw[1], pth = start_yaw, pw = 0.0;   // the pending entry (sampled_path.back())
  auto flush = [&]() {   // result keeps an entry only when its dt exceeds 1e-3 (graph_search.cpp:170-172)
    if (pw > 1.0e-3) {
      if (cnt < cap) { out[4 * cnt] = px; out[4 * cnt + 1] = py; out[4 * cnt + 2] = pth; out[4 * cnt + 3] = pw; }
      cnt++;
    }
  };
  if (!more) {   // fewer than two raw points: the reference would read past its vector; report the start pose only
    if (cap > 0) { out[0] = px; out[1] = py; out[2] = pth; out[3] = 0.0; }
    return 1;
  }
  double qx, qy;
  next(qx, qy);   // dense_path[1]
  double cur = det_atan2(qy - py, qx - px);
  normalize(start_yaw, cur);
  pw = fabs(cur - start_yaw) / w_max;
  flush();
  pth = cur; pw = 0.0;
  while (more) {   // qx, qy = interior point i, (nx, ny) = dense_path[i + 1]
    double nx, ny;
    next(nx, ny);
    const double ax = qx - px, ay = qy - py;
    pw = sqrt(ax * ax + ay * ay) / v_max;
    flush();
    px = qx; py = qy; pw = 0.0;
    cur = det_atan2(ny - qy, nx - qx);
    normalize(pth, cur);
    pw = fabs(cur - pth) / w_max;
    flush();
    pth = cur; pw = 0.0;
    qx = nx; qy = ny;
  }
  {
    const double ax = qx - px, ay = qy - py;
    pw = sqrt(ax * ax + ay * ay) / v_max;
    flush();
    px = qx; py = qy; pw = 0.0;
    cur = end_yaw;
    normalize(pth, cur);
    pw = fabs(cur - pth) / w_max;
    flush();
    pth = cur;
  }
  if (cnt < cap) { out[4 * cnt] = px; out[4 * cnt + 1] = py; out[4 * cnt + 2] = pth; out[4 * cnt + 3] = 0.0; }
  return cnt + 1;
}

__global__ void k_dense_path(int n, const double* raw, const long long* off, const int* len, double step_size, const double* syaw,
                             const double* eyaw, double v_max, double w_max, int cap, double* out, int* out_len) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  out_len[p] = dense_path(raw + 2 * off[p], len[p], step_size, syaw[p], eyaw[p], v_max, w_max, out + (size_t)p * cap * 4, cap);
}

// connectCollision, lines 327-345: check i of edge e is the state (car pose i of the edge, q_from + (q_to - q_from) * i / n_e);
// the edge collides when any of its states does (the reference returns at the first; the flag is the same).
__global__ void k_connect(const DevMap* maps, int map_id, long long n_checks, const int* edge_of, const int* idx_in_edge, const int* piece_num,
                          const double* car, const double* q_from, const double* q_to, int* collide) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_checks) return;
  const int e = edge_of[t], i = idx_in_edge[t];
  const double temp_i = 1.0 * i / (1.0 * piece_num[e]);
  double st[10];
  st[0] = car[3 * t]; st[1] = car[3 * t + 1]; st[2] = car[3 * t + 2];
#pragma unroll
  for (int q = 0; q < 7; q++) {
    const double a = q_from[7 * e + q];
    st[3 + q] = a + (q_to[7 * e + q] - a) * temp_i;
  }
  const DevMap M = maps[map_id];
  if (whole_body_collision(M, st)) atomicOr(collide + e, 1);
}

}  // namespace topay
