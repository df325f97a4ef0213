// The planner's direct 2-D chassis path on the device: GraphSearch::plan2dJPS (planner/src/graph_search.cpp:53-117) --
// jump-point search on the 2-D distance field (plan / getJpsSucc / jump / hasForced / recoverPath, graph_search.cpp:
// 178-306, 365-392, 439-475; the pruned / forced neighbour rules of JPS2DNeib, 583-669), then the zigzag cut with
// GridMap::isLineCollisionGrid2d (src/map/include/map/grid_map.h:565-610).  The open list is the reference's
// boost::heap::d_ary_heap<arity<2>, mutable_<true>> with compare_state (graph_search.h:20-38); boost is a third-party
// dependency that is not part of /root/reference, its sift rules are restated (they decide which of several equal-cost
// paths comes out).
// A best-first search is one sequential chain of heap operations, so the unit of parallelism is the search: one WAVE per
// (start, goal) pair -- a benchmark sweep plans 1024 scenarios at once, a planner its handful per cycle.  Lane 0 owns the
// search state (HBM, per instance: g, parent, heap position per cell, one byte of flags + direction, and the heap) and
// runs the serial part -- pop, successor bookkeeping, heap sifts; the jumps, which only read the map, are the wave's: a
// straight jump tests 64 successive cells at a time (free / goal / forced neighbour) and a ballot finds the first cell
// at which the sequential loop would have stopped.  The recursion of jump() is unrolled: a diagonal jump is a loop that
// tries the two straight jumps of every cell it passes.
#pragma once
#include "topay_front.h"

namespace topay {

struct JpsBatch {
  int n, cap;                   // searches, capacity of a returned path (points)
  int inst0;                    // first instance of this launch (the workspace is indexed by instance - inst0)
  long long ncell_max;          // workspace stride per instance (cells)
  const int* map_id;
  const double* start;          // n x 2
  const double* end;            // n x 2
  double threshold;
  // workspace
  double* g;                    // [inst][cell]
  int* parent;                  // [inst][cell]
  int* hpos;                    // [inst][cell]  position in the heap
  unsigned char* flag;          // [inst][cell]  bit 0 seen, 1 opened, 2 closed, bits 3-4 dx + 1, bits 5-6 dy + 1
  int* heap;                    // [inst][cell]
  // results
  int* out_len;                 // points of the path (0: none; may exceed cap: counted, not written)
  double* out_xy;               // [inst][cap][2]
  int* stats;                   // [inst][2]: expanded nodes, jump points of the raw path
};

struct JpsCtx {
  DevMap M;
  int nx, ny, xg, yg;
  double thr;
  double* g; int* parent; int* hpos; unsigned char* flag; int* heap;
  int hn;   // heap size
};

__device__ __forceinline__ bool jps_free(const JpsCtx& C, int x, int y) {
  if (x < 0 || x >= C.nx || y < 0 || y >= C.ny) return false;
  return !(C.M.esdf2d[(size_t)x * C.ny + y] < C.thr);
}
__device__ __forceinline__ double jps_h(const JpsCtx& C, int id) {
  const int x = id / C.ny, y = id % C.ny;
  return 1.0 * sqrt((double)((x - C.xg) * (x - C.xg) + (y - C.yg) * (y - C.yg)));
}
// compare_state: node a has lower priority than node b
__device__ __forceinline__ bool jps_less(const JpsCtx& C, int a, int b) {
  const double f1 = C.g[a] + jps_h(C, a), f2 = C.g[b] + jps_h(C, b);
  if ((f1 >= f2 - 0.000001) && (f1 <= f2 + 0.000001)) return C.g[a] < C.g[b];
  return f1 > f2;
}
__device__ __forceinline__ void jps_heap_swap(JpsCtx& C, int i, int j) {
  const int a = C.heap[i], b = C.heap[j];
  C.heap[i] = b; C.heap[j] = a;
  C.hpos[b] = i; C.hpos[a] = j;
}
__device__ inline void jps_siftup(JpsCtx& C, int index) {
  while (index != 0) {
    const int parent = (index - 1) / 2;
    if (jps_less(C, C.heap[parent], C.heap[index])) { jps_heap_swap(C, parent, index); index = parent; }
    else return;
  }
}
__device__ inline void jps_siftdown(JpsCtx& C, int index) {
  while (2 * index + 1 < C.hn) {
    int mc = 2 * index + 1;
    if (mc + 1 < C.hn && jps_less(C, C.heap[mc], C.heap[mc + 1])) mc = mc + 1;   // the first of the largest children
    if (!jps_less(C, C.heap[mc], C.heap[index])) { jps_heap_swap(C, mc, index); index = mc; }
    else return;
  }
}
__device__ inline void jps_push(JpsCtx& C, int id) {
  C.heap[C.hn] = id;
  C.hpos[id] = C.hn;
  C.hn++;
  jps_siftup(C, C.hn - 1);
}
__device__ inline int jps_pop(JpsCtx& C) {
  const int top = C.heap[0];
  jps_heap_swap(C, 0, C.hn - 1);
  C.hn--;
  if (C.hn > 0) jps_siftdown(C, 0);
  return top;
}
// forced neighbours of a move (dx, dy) arriving at (x, y): FNeib's f1 cells (graph_search.cpp:633-664)
__device__ __forceinline__ void jps_f1(int dx, int dy, int dev, int& fx, int& fy, int& nx, int& ny) {
  if (dx != 0 && dy != 0) {
    if (dev == 0) { fx = -dx; fy = 0; nx = -dx; ny = dy; }
    else { fx = 0; fy = -dy; nx = dx; ny = -dy; }
  } else {
    fx = 0; fy = dev == 0 ? 1 : -1;
    if (dx == 0) { fx = fy; fy = 0; }
    nx = dx + fx; ny = dy + fy;
  }
}
__device__ __forceinline__ bool jps_has_forced(const JpsCtx& C, int x, int y, int dx, int dy) {
  for (int fn = 0; fn < 2; ++fn) {
    int fx, fy, nx, ny;
    jps_f1(dx, dy, fn, fx, fy, nx, ny);
    if (!jps_free(C, x + fx, y + fy)) return true;
  }
  return false;
}
// jump along a straight direction (norm1 = 1): no inner jumps.  Wave-collective (uniform arguments): lane l tests cell
// base + l; the first cell that is blocked, the goal, or has a forced neighbour is where the sequential loop stops.
__device__ inline bool jps_jump_straight(const JpsCtx& C, int lane, int x, int y, int dx, int dy, int& ox, int& oy) {
  for (int base = 1;; base += 64) {
    const int i = base + lane, cx = x + i * dx, cy = y + i * dy;
    const bool fr = jps_free(C, cx, cy);
    const bool stop = !fr || (cx == C.xg && cy == C.yg) || jps_has_forced(C, cx, cy, dx, dy);
    const unsigned long long m = __ballot(stop);
    if (m) {
      const int f = __ffsll((long long)m) - 1;
      if (!__shfl((int)fr, f)) return false;
      ox = x + (base + f) * dx;
      oy = y + (base + f) * dy;
      return true;
    }
  }
}
// jump(): straight or diagonal; a diagonal step first tries the two straight jumps from the new cell (ns[id][.][0..1])
__device__ inline bool jps_jump(const JpsCtx& C, int lane, int x, int y, int dx, int dy, int& ox, int& oy) {
  if (dx == 0 || dy == 0) return jps_jump_straight(C, lane, x, y, dx, dy, ox, oy);
  for (;;) {
    x += dx; y += dy;
    if (!jps_free(C, x, y)) return false;
    ox = x; oy = y;
    if (x == C.xg && y == C.yg) return true;
    if (jps_has_forced(C, x, y, dx, dy)) return true;
    int tx, ty;
    if (jps_jump_straight(C, lane, x, y, dx, 0, tx, ty)) return true;
    if (jps_jump_straight(C, lane, x, y, 0, dy, tx, ty)) return true;
  }
}
__device__ __forceinline__ void jps_pos_to_index(const DevMap& M, double px, double py, int& ix, int& iy) {
  ix = (int)floor((px - M.origin[0]) * M.res_inv);
  iy = (int)floor((py - M.origin[1]) * M.res_inv);
}
__device__ inline bool jps_line_collides(const JpsCtx& C, double ax, double ay, double bx, double by) {   // isLineCollisionGrid2d
  int x0, y0, x1, y1;
  jps_pos_to_index(C.M, ax, ay, x0, y0);
  jps_pos_to_index(C.M, bx, by, x1, y1);
  const int dx = abs(x1 - x0), dy = abs(y1 - y0);
  const int sx = (x0 < x1) ? 1 : -1, sy = (y0 < y1) ? 1 : -1;
  int err = dx - dy;
  for (;;) {
    if (x0 < 0 || y0 < 0 || x0 >= C.nx || y0 >= C.ny) return true;   // (the reference would read outside its buffer)
    if (C.M.esdf2d[(size_t)x0 * C.ny + y0] < C.thr) return true;
    if (x0 == x1 && y0 == y1) break;
    const int e2 = 2 * err;
    if (e2 > -dy) { err -= dy; x0 += sx; }
    if (e2 < dx) { err += dx; y0 += sy; }
  }
  return false;
}

__global__ void __launch_bounds__(64) k_jps(const DevMap* maps, const JpsBatch B) {
  const int p = blockIdx.x, lane = threadIdx.x;
  if (p >= B.n) return;
  JpsCtx C;
  const int gp = B.inst0 + p;   // global instance: inputs and results (the workspace is per launch)
  C.M = maps[B.map_id[gp]];
  C.nx = C.M.dims[0]; C.ny = C.M.dims[1];
  C.thr = B.threshold;
  const size_t wo = (size_t)p * (size_t)B.ncell_max;
  C.g = B.g + wo; C.parent = B.parent + wo; C.hpos = B.hpos + wo; C.flag = B.flag + wo; C.heap = B.heap + wo;
  C.hn = 0;
  const double sx = B.start[2 * (size_t)gp], sy = B.start[2 * (size_t)gp + 1], ex = B.end[2 * (size_t)gp], ey = B.end[2 * (size_t)gp + 1];
  int* stats = B.stats + 2 * (size_t)gp;
  double* out = B.out_xy + (size_t)gp * B.cap * 2;
  if (lane == 0) { stats[0] = 0; stats[1] = 0; B.out_len[gp] = 0; }
  int xs, ys, xg, yg;
  jps_pos_to_index(C.M, sx, sy, xs, ys);
  jps_pos_to_index(C.M, ex, ey, xg, yg);
  if (xs < 0 || ys < 0 || xs >= C.nx || ys >= C.ny || xg < 0 || yg < 0 || xg >= C.nx || yg >= C.ny) return;
  C.xg = xg; C.yg = yg;
  const int goal_id = xg * C.ny + yg, start_id = xs * C.ny + ys;
  // flag byte: 1 seen, 2 opened, 4 closed, (dx + 1) << 3, (dy + 1) << 5.  Everything below that touches g / parent / flag /
  // the heap is lane 0's.
  if (lane == 0) {
    C.g[start_id] = 0.0;
    C.parent[start_id] = -1;
    C.flag[start_id] = 1 | 2 | (1 << 3) | (1 << 5);
    jps_push(C, start_id);
  }
  int expanded = 0, cur = -1;
  int state = 0;   // 0 searching, 1 goal popped, 2 given up
  for (;;) {
    int cdx = 0, cdy = 0;
    if (lane == 0) {
      expanded++;
      cur = jps_pop(C);
      C.flag[cur] |= 4;
      if (cur == goal_id) state = 1;
      cdx = ((C.flag[cur] >> 3) & 3) - 1;
      cdy = ((C.flag[cur] >> 5) & 3) - 1;
    }
    state = __shfl(state, 0);
    if (state) break;
    cur = __shfl(cur, 0); cdx = __shfl(cdx, 0); cdy = __shfl(cdy, 0);
    const int cx = cur / C.ny, cy = cur % C.ny;
    const int norm1 = abs(cdx) + abs(cdy);
    const int num_neib = norm1 == 0 ? 8 : (norm1 == 1 ? 1 : 3), num_fneib = norm1 == 0 ? 0 : 2;
    bool astar_error = false;   // (lane 0's)
    for (int dev = 0; dev < num_neib + num_fneib; ++dev) {
      int nxn, nyn, dx, dy;
      if (dev < num_neib) {
        if (norm1 == 0) {   // Neib, case 0
          const int tx[8] = {1, -1, 0, 1, -1, 0, 1, -1}, ty[8] = {0, 0, 1, 1, 1, -1, -1, -1};
          dx = tx[dev]; dy = ty[dev];
        } else if (norm1 == 1) {
          dx = cdx; dy = cdy;
        } else {
          dx = dev == 1 ? 0 : cdx;
          dy = dev == 0 ? 0 : cdy;
        }
        if (!jps_jump(C, lane, cx, cy, dx, dy, nxn, nyn)) continue;
      } else {
        int fx, fy;
        jps_f1(cdx, cdy, dev - num_neib, fx, fy, dx, dy);
        if (jps_free(C, cx + fx, cy + fy)) continue;
        if (!jps_jump(C, lane, cx, cy, dx, dy, nxn, nyn)) continue;
      }
      if (lane != 0 || astar_error) continue;   // (after "ASTAR ERROR!" the reference has left the loop: nothing more is processed)
      const int nid = nxn * C.ny + nyn;
      if (!(C.flag[nid] & 1)) {
        C.flag[nid] = (unsigned char)(1 | ((dx + 1) << 3) | ((dy + 1) << 5));
        C.g[nid] = 1.0e300 * 1.0e300;   // infinity
        C.parent[nid] = -1;
      }
      const double cost = sqrt((double)((nxn - cx) * (nxn - cx) + (nyn - cy) * (nyn - cy)));
      const double tentative = C.g[cur] + cost;
      if (tentative < C.g[nid]) {
        C.parent[nid] = cur;
        C.g[nid] = tentative;
        const unsigned char fl = C.flag[nid];
        if ((fl & 2) && !(fl & 4)) {
          jps_siftup(C, C.hpos[nid]);   // pq_.increase(heapkey)
          int ndx = nxn - cx, ndy = nyn - cy;
          if (ndx != 0) ndx /= abs(ndx);
          if (ndy != 0) ndy /= abs(ndy);
          C.flag[nid] = (unsigned char)((fl & 7) | ((ndx + 1) << 3) | ((ndy + 1) << 5));
        } else if ((fl & 2) && (fl & 4)) {
          astar_error = true;   // "ASTAR ERROR!": the reference gives up
        } else {
          C.flag[nid] = fl | 2;
          jps_push(C, nid);
        }
      }
    }
    if (lane == 0 && (astar_error || expanded >= 10000000 || C.hn == 0)) state = 2;
    state = __shfl(state, 0);
    if (state) break;
  }
  if (lane != 0) return;   // the rest is serial: path recovery and the zigzag cut
  const bool found = state == 1;
  stats[0] = expanded;
  if (!found) return;
  // recoverPath: ids from the goal back to the start, kept in the heap array (free now)
  int m = 0;
  for (int nd = cur; nd >= 0; nd = C.parent[nd]) {
    C.heap[m++] = nd;
    if (nd == start_id) break;
  }
  stats[1] = m;
  // raw path in travel order: point i = cell of heap[m - 1 - i]; first = start, last = end
  auto raw = [&](int i, double& x, double& y) {
    if (i == 0) { x = sx; y = sy; return; }
    if (i == m - 1) { x = ex; y = ey; return; }
    const int id = C.heap[m - 1 - i];
    x = (id / C.ny + 0.5) * C.M.res + C.M.origin[0];
    y = (id % C.ny + 0.5) * C.M.res + C.M.origin[1];
  };
  int cnt = 0;
  auto emit = [&](double x, double y) {
    if (cnt < B.cap) { out[2 * cnt] = x; out[2 * cnt + 1] = y; }
    cnt++;
  };
  if (m < 2) {   // start and goal in one cell: raw_path has the single point `end` (front = start, then back = end)
    emit(ex, ey);
    B.out_len[gp] = cnt;
    return;
  }
  // cut zigzag segment (graph_search.cpp:79-114)
  const double inf = 1.0e300 * 1.0e300;
  auto dist = [](double ax, double ay, double bx, double by) { const double dx = ax - bx, dy = ay - by; return sqrt(dx * dx + dy * dy); };
  double p1x, p1y, p2x, p2y, prx, pry;
  raw(0, p1x, p1y);
  raw(1, p2x, p2y);
  prx = p1x; pry = p1y;
  emit(p1x, p1y);
  double cost1 = !jps_line_collides(C, p1x, p1y, p2x, p2y) ? dist(p1x, p1y, p2x, p2y) : inf;
  for (int i = 1; i < m - 1; i++) {
    raw(i, p1x, p1y);
    raw(i + 1, p2x, p2y);
    const double cost2 = !jps_line_collides(C, p1x, p1y, p2x, p2y) ? dist(p1x, p1y, p2x, p2y) : inf;
    const double cost3 = !jps_line_collides(C, prx, pry, p2x, p2y) ? dist(prx, pry, p2x, p2y) : inf;
    if (cost3 < cost1 + cost2) cost1 = cost3;
    else {
      emit(p1x, p1y);
      cost1 = dist(p1x, p1y, p2x, p2y);
      prx = p1x; pry = p1y;
    }
  }
  raw(m - 1, p1x, p1y);
  emit(p1x, p1y);
  B.out_len[gp] = cnt;
}

}  // namespace topay
