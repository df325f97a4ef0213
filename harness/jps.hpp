// CPU restatement of the 2-D jump-point search that produces the planner's direct chassis path:
//   GraphSearch::plan2dJPS                 /root/reference/src/planner/src/graph_search.cpp:53-117
//   GraphSearch::plan / getJpsSucc / jump / hasForced / recoverPath      graph_search.cpp:178-306, 365-392, 439-475
//   JPS2DNeib (the pruned / forced neighbour tables)                      graph_search.cpp:583-669, graph_search.h:77-104
//   GridMap::posToIndex2d / indexToPos2d / isCollisionIndx2d / isLineCollisionGrid2d    src/map/include/map/grid_map.h:538-610, 744-781
// and of the one third-party piece it uses: boost::heap::d_ary_heap<arity<2>, mutable_<true>> (graph_search.h:37-38) --
// boost is not part of /root/reference; DAryHeap below restates push / pop / increase of boost/heap/d_ary_heap.hpp (sift-up
// swaps while the parent compares less, sift-down swaps with the first largest child unless that child compares less,
// pop moves the last element to the root).  Which of several equal-cost paths the search returns depends on exactly these
// rules and on compare_state's tolerance (graph_search.h:20-30).  PARITY UNPINNED for the heap (no boost here).
// Test infrastructure only: the checker of topay_plan2d_jps.
#pragma once
#include <cmath>
#include <limits>
#include <memory>
#include <vector>

#include "workload.hpp"

namespace topay_wl {

struct JpsState {
  int id, x, y, dx, dy;
  int parentId = -1;
  int heapkey = -1;   // position in the heap (boost: a handle)
  double g = std::numeric_limits<double>::infinity();
  double h = 0.0;
  bool opened = false, closed = false;
  JpsState(int id_, int x_, int y_, int dx_, int dy_) : id(id_), x(x_), y(y_), dx(dx_), dy(dy_) {}
};
typedef std::shared_ptr<JpsState> JpsStatePtr;

// compare_state (graph_search.h:20-30): "a1 has lower priority than a2"
inline bool jps_less(const JpsStatePtr& a1, const JpsStatePtr& a2) {
  double f1 = a1->g + a1->h;
  double f2 = a2->g + a2->h;
  if ((f1 >= f2 - 0.000001) && (f1 <= f2 + 0.000001)) return a1->g < a2->g;
  return f1 > f2;
}

struct DAryHeap {   // boost::heap::d_ary_heap, arity 2, mutable
  std::vector<JpsStatePtr> q_;
  bool empty() const { return q_.empty(); }
  void clear() { q_.clear(); }
  void swap_(size_t a, size_t b) {
    std::swap(q_[a], q_[b]);
    q_[a]->heapkey = (int)a;
    q_[b]->heapkey = (int)b;
  }
  void siftup(size_t index) {
    while (index != 0) {
      size_t parent = (index - 1) / 2;
      if (jps_less(q_[parent], q_[index])) { swap_(parent, index); index = parent; }
      else return;
    }
  }
  size_t top_child_index(size_t index) const {
    size_t first = index * 2 + 1, last = std::min(first + 2, q_.size());
    size_t best = first;   // std::max_element: the first of the largest
    for (size_t i = first + 1; i < last; i++)
      if (jps_less(q_[best], q_[i])) best = i;
    return best;
  }
  void siftdown(size_t index) {
    while (index * 2 + 1 < q_.size()) {
      size_t mc = top_child_index(index);
      if (!jps_less(q_[mc], q_[index])) { swap_(mc, index); index = mc; }
      else return;
    }
  }
  int push(const JpsStatePtr& v) {
    q_.push_back(v);
    v->heapkey = (int)q_.size() - 1;
    siftup(q_.size() - 1);
    return v->heapkey;
  }
  JpsStatePtr top() const { return q_.front(); }
  void pop() {
    std::swap(q_.front(), q_.back());
    q_.pop_back();
    if (q_.empty()) return;
    q_.front()->heapkey = 0;
    siftdown(0);
  }
  void increase(int handle) { siftup((size_t)handle); }
};

struct JPS2DNeib {   // graph_search.cpp:583-669
  int ns[9][2][8];
  int f1[9][2][2];
  int f2[9][2][2];
  static constexpr int nsz[3][2] = {{8, 0}, {1, 2}, {3, 2}};
  JPS2DNeib() {
    int id = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        int norm1 = std::abs(dx) + std::abs(dy);
        for (int dev = 0; dev < nsz[norm1][0]; ++dev) Neib(dx, dy, norm1, dev, ns[id][0][dev], ns[id][1][dev]);
        for (int dev = 0; dev < nsz[norm1][1]; ++dev) FNeib(dx, dy, norm1, dev, f1[id][0][dev], f1[id][1][dev], f2[id][0][dev], f2[id][1][dev]);
        id++;
      }
  }
  // The neighbours that are always expanded (graph_search.cpp:609-636): all eight at the start node in the order
  // (1,0) (-1,0) (0,1) (1,1) (-1,1) (0,-1) (1,-1) (-1,-1); the move itself for a straight move; for a diagonal move its
  // horizontal part, its vertical part, then the move.
  static void Neib(int dx, int dy, int norm1, int dev, int& tx, int& ty) {
    if (norm1 == 0) {
      static const int ex[8] = {1, -1, 0, 1, -1, 0, 1, -1}, ey[8] = {0, 0, 1, 1, 1, -1, -1, -1};
      tx = ex[dev];
      ty = ey[dev];
    } else if (norm1 == 1) {
      tx = dx;
      ty = dy;
    } else {
      tx = dev == 1 ? 0 : dx;
      ty = dev == 0 ? 0 : dy;
    }
  }
  // The cells whose being blocked forces a neighbour, and that neighbour (graph_search.cpp:638-669).  Straight move: the
  // two cells beside the arrival cell (across the direction of travel), neighbour = that side, one step on.  Diagonal
  // move: the cells behind the arrival cell along either axis, neighbour = the move mirrored in that axis.
  static void FNeib(int dx, int dy, int norm1, int dev, int& fx, int& fy, int& nx, int& ny) {
    if (norm1 == 1) {
      const int side = dev == 0 ? 1 : -1;
      fx = dx == 0 ? side : 0;
      fy = dx == 0 ? 0 : side;
      nx = dx + fx;
      ny = dy + fy;
    } else if (norm1 == 2) {
      fx = dev == 0 ? -dx : 0;
      fy = dev == 0 ? 0 : -dy;
      nx = dev == 0 ? -dx : dx;
      ny = dev == 0 ? dy : -dy;
    }
  }
};

struct GraphSearch {
  const GridMap& map_;
  double safe_dis_;
  int xDim_, yDim_, xGoal_ = 0, yGoal_ = 0;
  double eps_ = 1;
  DAryHeap pq_;
  std::vector<JpsStatePtr> hm_;
  std::vector<bool> seen_;
  std::vector<JpsStatePtr> path_;
  JPS2DNeib jn2d_;
  int expand_iteration = 0;

  GraphSearch(const GridMap& m, double safe_dis) : map_(m), safe_dis_(safe_dis), xDim_(m.voxel_num[0]), yDim_(m.voxel_num[1]) {
    hm_.resize((size_t)xDim_ * yDim_);
    seen_.resize((size_t)xDim_ * yDim_, false);
  }
  int coordToId(int x, int y) const { return x * yDim_ + y; }   // GridMap::toAddress2d
  bool isCollisionIndx2d(int x, int y, double thr) const { return map_.esdf2d[(size_t)coordToId(x, y)] < thr; }
  bool isFree(int x, int y) const {
    if (x < 0 || x >= xDim_ || y < 0 || y >= yDim_) return false;
    return !isCollisionIndx2d(x, y, safe_dis_);
  }
  double getHeur(int x, int y) const { return eps_ * std::sqrt((double)((x - xGoal_) * (x - xGoal_) + (y - yGoal_) * (y - yGoal_))); }
  void posToIndex2d(const double* pos, int* id) const {
    id[0] = (int)std::floor((pos[0] - map_.origin[0]) * map_.resolution_inv);
    id[1] = (int)std::floor((pos[1] - map_.origin[1]) * map_.resolution_inv);
  }
  void indexToPos2d(int ix, int iy, double* pos) const {
    pos[0] = (ix + 0.5) * map_.resolution + map_.origin[0];
    pos[1] = (iy + 0.5) * map_.resolution + map_.origin[1];
  }
  bool isLineCollisionGrid2d(const double* p1, const double* p2, double thr) const {   // grid_map.h:565-610
    int start[2], end[2];
    posToIndex2d(p1, start);
    posToIndex2d(p2, end);
    int dx = std::abs(end[0] - start[0]), dy = std::abs(end[1] - start[1]);
    int sx = (start[0] < end[0]) ? 1 : -1, sy = (start[1] < end[1]) ? 1 : -1;
    int err = dx - dy, x0 = start[0], y0 = start[1];
    while (true) {
      if (x0 < 0 || y0 < 0 || x0 >= xDim_ || y0 >= yDim_) return true;   // (the reference would read outside its buffer)
      if (isCollisionIndx2d(x0, y0, thr)) return true;
      if (x0 == end[0] && y0 == end[1]) break;
      int e2 = 2 * err;
      if (e2 > -dy) { err -= dy; x0 += sx; }
      if (e2 < dx) { err += dx; y0 += sy; }
    }
    return false;
  }
  bool hasForced(int x, int y, int dx, int dy) const {
    const int id = (dx + 1) + 3 * (dy + 1);
    for (int fn = 0; fn < 2; ++fn) {
      int nx = x + jn2d_.f1[id][0][fn], ny = y + jn2d_.f1[id][1][fn];
      if (!isFree(nx, ny)) return true;
    }
    return false;
  }
  bool jump(int x, int y, int dx, int dy, int& new_x, int& new_y) const {   // graph_search.cpp:439-461
    new_x = x + dx;
    new_y = y + dy;
    if (!isFree(new_x, new_y)) return false;
    if (new_x == xGoal_ && new_y == yGoal_) return true;
    if (hasForced(new_x, new_y, dx, dy)) return true;
    const int id = (dx + 1) + 3 * (dy + 1);
    const int norm1 = std::abs(dx) + std::abs(dy);
    int num_neib = JPS2DNeib::nsz[norm1][0];
    for (int k = 0; k < num_neib - 1; ++k) {
      int new_new_x, new_new_y;
      if (jump(new_x, new_y, jn2d_.ns[id][0][k], jn2d_.ns[id][1][k], new_new_x, new_new_y)) return true;
    }
    return jump(new_x, new_y, dx, dy, new_x, new_y);
  }
  void getJpsSucc(const JpsStatePtr& curr, std::vector<int>& succ_ids, std::vector<double>& succ_costs) {
    const int norm1 = std::abs(curr->dx) + std::abs(curr->dy);
    int num_neib = JPS2DNeib::nsz[norm1][0], num_fneib = JPS2DNeib::nsz[norm1][1];
    int id = (curr->dx + 1) + 3 * (curr->dy + 1);
    for (int dev = 0; dev < num_neib + num_fneib; ++dev) {
      int new_x, new_y, dx, dy;
      if (dev < num_neib) {
        dx = jn2d_.ns[id][0][dev];
        dy = jn2d_.ns[id][1][dev];
        if (!jump(curr->x, curr->y, dx, dy, new_x, new_y)) continue;
      } else {
        int nx = curr->x + jn2d_.f1[id][0][dev - num_neib], ny = curr->y + jn2d_.f1[id][1][dev - num_neib];
        if (!isFree(nx, ny)) {
          dx = jn2d_.f2[id][0][dev - num_neib];
          dy = jn2d_.f2[id][1][dev - num_neib];
          if (!jump(curr->x, curr->y, dx, dy, new_x, new_y)) continue;
        } else {
          continue;
        }
      }
      int new_id = coordToId(new_x, new_y);
      if (!seen_[new_id]) {
        seen_[new_id] = true;
        hm_[new_id] = std::make_shared<JpsState>(new_id, new_x, new_y, dx, dy);
        hm_[new_id]->h = getHeur(new_x, new_y);
      }
      succ_ids.push_back(new_id);
      succ_costs.push_back(std::sqrt((double)((new_x - curr->x) * (new_x - curr->x) + (new_y - curr->y) * (new_y - curr->y))));
    }
  }
  bool plan(int xStart, int yStart, int xGoal, int yGoal, int maxExpand) {   // graph_search.cpp:178-306
    pq_.clear();
    path_.clear();
    std::fill(seen_.begin(), seen_.end(), false);
    int goal_id = coordToId(xGoal, yGoal);
    xGoal_ = xGoal; yGoal_ = yGoal;
    int start_id = coordToId(xStart, yStart);
    JpsStatePtr currNode_ptr = std::make_shared<JpsState>(start_id, xStart, yStart, 0, 0);
    currNode_ptr->g = 0;
    currNode_ptr->h = getHeur(xStart, yStart);
    currNode_ptr->heapkey = pq_.push(currNode_ptr);
    currNode_ptr->opened = true;
    hm_[currNode_ptr->id] = currNode_ptr;
    seen_[currNode_ptr->id] = true;
    expand_iteration = 0;
    while (true) {
      expand_iteration++;
      currNode_ptr = pq_.top();
      pq_.pop();
      currNode_ptr->closed = true;
      if (currNode_ptr->id == goal_id) break;
      std::vector<int> succ_ids;
      std::vector<double> succ_costs;
      getJpsSucc(currNode_ptr, succ_ids, succ_costs);
      for (int s = 0; s < (int)succ_ids.size(); s++) {
        JpsStatePtr& child_ptr = hm_[succ_ids[s]];
        double tentative_gval = currNode_ptr->g + succ_costs[s];
        if (tentative_gval < child_ptr->g) {
          child_ptr->parentId = currNode_ptr->id;
          child_ptr->g = tentative_gval;
          if (child_ptr->opened && !child_ptr->closed) {
            pq_.increase(child_ptr->heapkey);
            child_ptr->dx = (child_ptr->x - currNode_ptr->x);
            child_ptr->dy = (child_ptr->y - currNode_ptr->y);
            if (child_ptr->dx != 0) child_ptr->dx /= std::abs(child_ptr->dx);
            if (child_ptr->dy != 0) child_ptr->dy /= std::abs(child_ptr->dy);
          } else if (child_ptr->opened && child_ptr->closed) {
            return false;   // "ASTAR ERROR!"
          } else {
            child_ptr->heapkey = pq_.push(child_ptr);
            child_ptr->opened = true;
          }
        }
      }
      if (maxExpand > 0 && expand_iteration >= maxExpand) return false;
      if (pq_.empty()) return false;
    }
    // recoverPath
    JpsStatePtr node = currNode_ptr;
    path_.push_back(node);
    while (node && node->id != start_id) {
      node = hm_[node->parentId];
      path_.push_back(node);
    }
    return true;
  }
  // plan2dJPS (graph_search.cpp:53-117): positions of the returned 2-D path, empty when there is none
  std::vector<std::array<double, 2>> plan2dJPS(const double* start, const double* end, double threshold) {
    safe_dis_ = threshold;
    std::vector<std::array<double, 2>> raw_path;
    int idx_s[2], idx_e[2];
    posToIndex2d(start, idx_s);
    posToIndex2d(end, idx_e);
    for (int a = 0; a < 2; a++)   // (the reference indexes its buffers without a check)
      if (idx_s[a] < 0 || idx_e[a] < 0 || idx_s[a] >= (a ? yDim_ : xDim_) || idx_e[a] >= (a ? yDim_ : xDim_)) return raw_path;
    if (!plan(idx_s[0], idx_s[1], idx_e[0], idx_e[1], 10000000)) return raw_path;
    for (size_t i = 0; i < path_.size(); i++) {
      std::array<double, 2> p;
      indexToPos2d(path_[i]->x, path_[i]->y, p.data());
      raw_path.push_back(p);
    }
    std::reverse(raw_path.begin(), raw_path.end());
    raw_path.front() = {start[0], start[1]};
    raw_path.back() = {end[0], end[1]};
    if (raw_path.size() < 2) return raw_path;
    // cut zigzag segment
    std::vector<std::array<double, 2>> optimized_path;
    std::array<double, 2> pose1 = raw_path[0], pose2 = raw_path[1], prev_pose = pose1;
    optimized_path.push_back(pose1);
    double cost1, cost2, cost3;
    auto norm = [](const std::array<double, 2>& a, const std::array<double, 2>& b) {
      double dx = a[0] - b[0], dy = a[1] - b[1];
      return std::sqrt(dx * dx + dy * dy);
    };
    const double inf = std::numeric_limits<double>::infinity();
    if (!isLineCollisionGrid2d(pose1.data(), pose2.data(), safe_dis_)) cost1 = norm(pose1, pose2);
    else cost1 = inf;
    for (unsigned int i = 1; i < raw_path.size() - 1; i++) {
      pose1 = raw_path[i];
      pose2 = raw_path[i + 1];
      if (!isLineCollisionGrid2d(pose1.data(), pose2.data(), safe_dis_)) cost2 = norm(pose1, pose2);
      else cost2 = inf;
      if (!isLineCollisionGrid2d(prev_pose.data(), pose2.data(), safe_dis_)) cost3 = norm(prev_pose, pose2);
      else cost3 = inf;
      if (cost3 < cost1 + cost2) cost1 = cost3;
      else {
        optimized_path.push_back(raw_path[i]);
        cost1 = norm(pose1, pose2);
        prev_pose = pose1;
      }
    }
    optimized_path.push_back(raw_path.back());
    return optimized_path;
  }
};

}  // namespace topay_wl
