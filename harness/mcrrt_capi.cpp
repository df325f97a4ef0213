// C entry points of the MCRRT / Reeds-Shepp restatement (harness/mcrrt.hpp) for the Python tests.  A library of its own,
// built with -ffp-contract=off: the search compares costs computed with a*b+c expressions, and a fused multiply-add
// on the host would differ from the device (built with contraction off) in the last bit.
#include "mcrrt.hpp"
#include "jps.hpp"

using namespace topay_wl;

extern "C" {

// One planning instance on `world` (a World* of libtopay_workload.so: same header, same layout).  car_path: L x (x, y,
// theta, dt) as getDensePath returns it.  wb: up to L x 10.  stats[8]: status (1 path, 0 none, -1 node pool full), nodes,
// iterations, tree count, anti-tree count, index of path_node_1, of path_node_2, whole-body checks.  nodes: up to
// nodes_cap rows in creation order (may be null).
int wl_mcrrt_plan(void* world, const double* start, const double* end, int L, const double* car_path, const McrrtParams* prm,
                  unsigned long long inst, int track_slack, double* wb, int* wb_len, int* stats, double* cmax, double* min_slack,
                  int nodes_cap, McrrtNodeRec* nodes) {
  const World& w = *(const World*)world;
  MCRRTs m(w, *prm, inst);
  m.track_slack = track_slack != 0;
  std::vector<std::array<double, 4>> path(L);
  for (int i = 0; i < L; i++) for (int a = 0; a < 4; a++) path[i][a] = car_path[4 * i + a];
  std::vector<std::array<double, 10>> out;
  const int st = m.plan(start, end, path, out);
  *wb_len = (int)out.size();
  for (size_t i = 0; i < out.size(); i++) std::memcpy(wb + 10 * i, out[i].data(), 10 * sizeof(double));
  stats[0] = st;
  stats[1] = (int)m.by_index.size();
  stats[2] = m.iterations;
  stats[3] = m.tree_count_;
  stats[4] = m.anti_tree_count_;
  stats[5] = m.path_node_1 ? m.path_node_1->index : -1;
  stats[6] = m.path_node_2 ? m.path_node_2->index : -1;
  stats[7] = (int)(m.n_checks & 0x7fffffff);
  *cmax = m.c_max;
  *min_slack = m.min_slack;
  if (nodes)
    for (int i = 0; i < (int)m.by_index.size() && i < nodes_cap; i++) {
      const MCRRTs::Node* n = m.by_index[i];
      nodes[i].layer = n->layer;
      nodes[i].state = (int)n->node_state;
      nodes[i].parent = n->parent ? n->parent->index : -1;
      nodes[i].cost = n->cost;
      std::memcpy(nodes[i].q, n->q, sizeof(n->q));
    }
  return st;
}

// ReedsSheppStateSpace(rho): the shortest path's word (0..17), its five signed segment lengths (units of rho) and
// distance(); interpolate(from, to, t).
double wl_rs_path(double rho, const double* from, const double* to, int* type, double* lengths) {
  ReedsShepp rs(rho);
  ReedsShepp::Path p = rs.reedsShepp(from, to);
  *type = p.type;
  for (int i = 0; i < 5; i++) lengths[i] = p.length[i];
  return rho * p.total;
}
void wl_rs_interpolate(double rho, const double* from, const double* to, double t, double* out) { ReedsShepp(rho).interpolate(from, to, t, out); }

// GraphSearch::plan2dJPS on `world`'s 2-D distance field: returns the number of path points (0 = no path), writes at most
// cap of them; stats[0] = expanded nodes, stats[1] = jump points of the raw path (before the zigzag cut).
int wl_plan2d_jps(void* world, const double* start, const double* end, double threshold, int cap, double* out_xy, int* stats) {
  const World& w = *(const World*)world;
  GraphSearch gs(w.gm, threshold);
  auto path = gs.plan2dJPS(start, end, threshold);
  stats[0] = gs.expand_iteration;
  stats[1] = (int)gs.path_.size();
  for (size_t i = 0; i < path.size() && (int)i < cap; i++) { out_xy[2 * i] = path[i][0]; out_xy[2 * i + 1] = path[i][1]; }
  return (int)path.size();
}

double wl_mcrrt_u01(unsigned long long seed, unsigned long long inst, unsigned long long iter, unsigned long long slot) {
  return mcrrt_u01(seed, inst, iter, slot);
}
}
