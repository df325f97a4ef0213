// The layered joint-space search of the front-end on the device: one wavefront per planning
// instance (one candidate chassis path of one scenario), thousands of instances per launch.
//   MCRRTs::plan            planner/src/mcrrts.cpp:5-231      the bidirectional tree over (layer of the chassis path, q1..q7)
//   MCRRTs::steer / rewire  mcrrts.cpp:336-400
//   inline members          planner/include/planner/mcrrts.h:153-348 (estHeuristic, getKey, genNodeFromState, sampleState,
//                           getNearestNode, linkNode + updateCosts, mergeTree, feasibleCheck, connectCollision)
//   ompl::base::ReedsSheppStateSpace::distance / interpolate (mcrrts.h:318-324, 336) -- OMPL is a third-party dependency
//                           that is not part of /root/reference: its published algorithm (Reeds & Shepp 1990, formulas
//                           8.1-8.11, candidate order and tie rules of OMPL's ReedsSheppStateSpace.cpp) is written out below.
// The tree itself is sequential (every iteration depends on the tree the previous ones left); what a wave parallelises
// is the work inside an iteration: the up to several hundred interpolated states of an edge's collision check (lane =
// state: Reeds-Shepp pose + joint interpolation + GridMap::isWholeBodyCollision), the resampling tries of sampleState
// (lane = try), the scans of the node table (nearest node, key lookup, cost propagation: lane = node) and the
// 2 (L - 1) Reeds-Shepp words of the chassis path (lane = edge, once per instance: every edge of the tree joins
// neighbouring layers, so its chassis motion is one of those).
// Deterministic where the reference is not (include/topay.h, topay_mcrrt_params_t): counter-based random numbers,
// iteration / try / node caps instead of wall-clock limits.  harness/mcrrt.hpp is the CPU restatement with the same
// rules; tests compare the trees node by node.
#pragma once
#include "topay_front.h"

namespace topay {

struct McrrtParams {   // == topay_mcrrt_params_t
  double goal_sample_rate, check_colli_res, rs_rho;
  int max_iter, max_sample_tries, node_cap, reserved;
  unsigned long long seed;
};

__device__ __forceinline__ unsigned long long mcrrt_mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// draw `slot` of iteration `iter` of instance `inst`: a pure function, so lanes can evaluate tries in parallel
__device__ __forceinline__ double mcrrt_u01(unsigned long long seed, unsigned long long inst, unsigned long long iter, unsigned long long slot) {
  unsigned long long h = mcrrt_mix(seed + inst * 0x9E3779B97F4A7C15ull);
  h = mcrrt_mix(h + iter * 0xD1342543DE82EF95ull);
  h = mcrrt_mix(h + slot * 0x94D049BB133111EBull);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------------------------------------------------------
// Reeds-Shepp shortest path between two poses, in units of the turning radius.  A path is one of 18 words of up to five
// segments (left arc / right arc / straight) with signed lengths.  The candidates are eight base formulas, each tried
// for the four symmetries (as is, time-flipped, reflected, both) and some also for the reversed problem; the first
// strictly shorter candidate wins, in OMPL's order: CSC, CCC, CCCC, CCSC, CCSCC.
// ---------------------------------------------------------------------------------------------------------------
struct RsPath {
  double len[5];
  double total;
  int type;
  int pad;
};
#define TOPAY_RS_PI 3.14159265358979323846
__device__ __forceinline__ double rs_mod2pi(double x) {
  double v = fmod(x, 2.0 * TOPAY_RS_PI);
  if (v < -TOPAY_RS_PI) v += 2.0 * TOPAY_RS_PI;
  else if (v > TOPAY_RS_PI) v -= 2.0 * TOPAY_RS_PI;
  return v;
}
__device__ __forceinline__ double rs_sin(double x) { double s, c; det_sincos(x, &s, &c); return s; }
__device__ __forceinline__ double rs_cos(double x) { double s, c; det_sincos(x, &s, &c); return c; }
// asin / acos through the library's own atan2 (the device's and the host's libm differ in the last bits; the emulator
// build of this file has to give the device's bits)
__device__ __forceinline__ double rs_asin(double x) { return det_atan2(x, sqrt((1.0 - x) * (1.0 + x))); }
__device__ __forceinline__ double rs_acos(double x) { return det_atan2(sqrt((1.0 - x) * (1.0 + x)), x); }
__device__ __forceinline__ void rs_tau_omega(double u, double v, double xi, double eta, double phi, double& tau, double& omega) {
  const double delta = rs_mod2pi(u - v);
  const double A = rs_sin(u) - rs_sin(delta), B = rs_cos(u) - rs_cos(delta) - 1.0;
  const double t1 = det_atan2(eta * A - xi * B, xi * A + eta * B), t2 = 2.0 * (rs_cos(delta) - rs_cos(v) - rs_cos(u)) + 3;
  tau = (t2 < 0) ? rs_mod2pi(t1 + TOPAY_RS_PI) : rs_mod2pi(t1);
  omega = rs_mod2pi(tau - u + v - phi);
}
// base formula f of the paper's section 8: 0 = 8.1 L+S+L+, 1 = 8.2 L+S+R+, 2 = 8.3/8.4 L+R-L, 3 = 8.7 L+R+uL-uR-,
// 4 = 8.8 L+R-uL-uR+, 5 = 8.9 L+R-S-L-, 6 = 8.10 L+R-S-R-, 7 = 8.11 L+R-S-L-R+
__device__ inline bool rs_base(int f, double x, double y, double phi, double& t, double& u, double& v) {
  const double ZERO = 10.0 * 2.220446049250313e-16;
  double sp, cp;
  det_sincos(phi, &sp, &cp);
  if (f == 0) {
    const double a = x - sp, b = y - 1.0 + cp;
    u = sqrt(a * a + b * b);
    t = det_atan2(b, a);
    if (t >= -ZERO) {
      v = rs_mod2pi(phi - t);
      if (v >= -ZERO) return true;
    }
    return false;
  }
  if (f == 1) {
    const double a = x + sp, b = y - 1.0 - cp;
    double u1 = sqrt(a * a + b * b);
    const double t1 = det_atan2(b, a);
    u1 = u1 * u1;
    if (u1 >= 4.0) {
      u = sqrt(u1 - 4.0);
      const double theta = det_atan2(2.0, u);
      t = rs_mod2pi(t1 + theta);
      v = rs_mod2pi(t - phi);
      return t >= -ZERO && v >= -ZERO;
    }
    return false;
  }
  if (f == 2) {
    const double xi = x - sp, eta = y - 1.0 + cp;
    const double u1 = sqrt(xi * xi + eta * eta), theta = det_atan2(eta, xi);
    if (u1 <= 4.0) {
      u = -2.0 * rs_asin(0.25 * u1);
      t = rs_mod2pi(theta + 0.5 * u + TOPAY_RS_PI);
      v = rs_mod2pi(phi - t + u);
      return t >= -ZERO && u <= ZERO;
    }
    return false;
  }
  if (f == 3) {
    const double xi = x + sp, eta = y - 1.0 - cp, r = 0.25 * (2.0 + sqrt(xi * xi + eta * eta));
    if (r <= 1.0) {
      u = rs_acos(r);
      rs_tau_omega(u, -u, xi, eta, phi, t, v);
      return t >= -ZERO && v <= ZERO;
    }
    return false;
  }
  if (f == 4) {
    const double xi = x + sp, eta = y - 1.0 - cp, r = (20.0 - xi * xi - eta * eta) / 16.0;
    if (r >= 0 && r <= 1) {
      u = -rs_acos(r);
      if (u >= -0.5 * TOPAY_RS_PI) {
        rs_tau_omega(u, u, xi, eta, phi, t, v);
        return t >= -ZERO && v >= -ZERO;
      }
    }
    return false;
  }
  if (f == 5) {
    const double xi = x - sp, eta = y - 1.0 + cp;
    const double r0 = sqrt(xi * xi + eta * eta), theta = det_atan2(eta, xi);
    if (r0 >= 2.0) {
      const double r = sqrt(r0 * r0 - 4.0);
      u = 2.0 - r;
      t = rs_mod2pi(theta + det_atan2(r, -2.0));
      v = rs_mod2pi(phi - 0.5 * TOPAY_RS_PI - t);
      return t >= -ZERO && u <= ZERO && v <= ZERO;
    }
    return false;
  }
  if (f == 6) {
    const double xi = x + sp, eta = y - 1.0 - cp;
    const double r0 = sqrt(eta * eta + xi * xi), theta = det_atan2(xi, -eta);   // polar(-eta, xi)
    if (r0 >= 2.0) {
      t = theta;
      u = 2.0 - r0;
      v = rs_mod2pi(t + 0.5 * TOPAY_RS_PI - phi);
      return t >= -ZERO && u <= ZERO && v <= ZERO;
    }
    return false;
  }
  {
    const double xi = x + sp, eta = y - 1.0 - cp;
    const double r0 = sqrt(xi * xi + eta * eta);
    if (r0 >= 2.0) {
      u = 4.0 - sqrt(r0 * r0 - 4.0);
      if (u <= ZERO) {
        t = rs_mod2pi(det_atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta));
        v = rs_mod2pi(t - phi);
        return t >= -ZERO && v >= -ZERO;
      }
    }
    return false;
  }
}
__device__ __forceinline__ void rs_set(RsPath& P, int type, double a, double b, double c, double d, double e) {
  P.type = type;
  P.len[0] = a; P.len[1] = b; P.len[2] = c; P.len[3] = d; P.len[4] = e;
  P.total = fabs(a) + fabs(b) + fabs(c) + fabs(d) + fabs(e);
}
// the four symmetries of base formula f on (x, y, phi); layout = where (t, u, v) go in the word
__device__ inline void rs_family(int f, int base_type, int layout, double x, double y, double phi, double& Lmin, RsPath& P) {
  const double hp = 0.5 * TOPAY_RS_PI;
  for (int k = 0; k < 4; k++) {
    const bool tf = (k & 1) != 0, rf = (k & 2) != 0;   // time flip, reflection
    double t, u, v;
    if (!rs_base(f, tf ? -x : x, rf ? -y : y, (tf != rf) ? -phi : phi, t, u, v)) continue;
    const double L = (layout == 2 || layout == 3) ? fabs(t) + 2.0 * fabs(u) + fabs(v) : fabs(t) + fabs(u) + fabs(v);
    if (!(Lmin > L)) continue;
    const double s = tf ? -1.0 : 1.0;
    const int type = base_type + (rf ? 1 : 0);
    if (layout == 0) rs_set(P, type, s * t, s * u, s * v, 0.0, 0.0);
    else if (layout == 1) rs_set(P, type, s * v, s * u, s * t, 0.0, 0.0);               // the reversed problem: segments in reverse order
    else if (layout == 2) rs_set(P, type, s * t, s * u, -(s * u), s * v, 0.0);
    else if (layout == 3) rs_set(P, type, s * t, s * u, s * u, s * v, 0.0);
    else if (layout == 4) rs_set(P, type, s * t, -(s * hp), s * u, s * v, 0.0);
    else if (layout == 5) rs_set(P, type, s * v, s * u, -(s * hp), s * t, 0.0);
    else rs_set(P, type, s * t, -(s * hp), s * u, -(s * hp), s * v);
    Lmin = L;
  }
}
// word numbering (segment types of word w): rs_segments()
__device__ inline void rs_shortest(double x, double y, double phi, RsPath& P) {
  P.type = 0;
  P.len[0] = 1.7976931348623157e308; P.len[1] = P.len[2] = P.len[3] = P.len[4] = 0.0;
  P.total = 1.7976931348623157e308;
  double sp, cp;
  det_sincos(phi, &sp, &cp);
  const double xb = x * cp + y * sp, yb = x * sp - y * cp;
  double Lmin = P.total;                       // CSC
  rs_family(0, 14, 0, x, y, phi, Lmin, P);
  rs_family(1, 12, 0, x, y, phi, Lmin, P);
  Lmin = P.total;                              // CCC
  rs_family(2, 0, 0, x, y, phi, Lmin, P);
  rs_family(2, 0, 1, xb, yb, phi, Lmin, P);
  Lmin = P.total;                              // CCCC
  rs_family(3, 2, 2, x, y, phi, Lmin, P);
  rs_family(4, 2, 3, x, y, phi, Lmin, P);
  Lmin = P.total - 0.5 * TOPAY_RS_PI;          // CCSC
  rs_family(5, 4, 4, x, y, phi, Lmin, P);
  rs_family(6, 8, 4, x, y, phi, Lmin, P);
  rs_family(5, 6, 5, xb, yb, phi, Lmin, P);
  rs_family(6, 10, 5, xb, yb, phi, Lmin, P);
  Lmin = P.total - TOPAY_RS_PI;                // CCSCC
  rs_family(7, 16, 6, x, y, phi, Lmin, P);
}
__device__ inline void rs_between(const double* from, const double* to, double rho, RsPath& P) {
  const double dx = to[0] - from[0], dy = to[1] - from[1];
  double s, c;
  det_sincos(from[2], &s, &c);
  const double x = c * dx + s * dy, y = -s * dx + c * dy, phi = to[2] - from[2];
  rs_shortest(x / rho, y / rho, phi, P);
}
// segment i of word w: 1 left, 2 straight, 3 right, 0 none
__device__ __forceinline__ int rs_segment(int w, int i) {
  // two bits per segment, first segment in the low bits
  const unsigned short T[18] = {0x01D, 0x037, 0x0DD, 0x077, 0x06D, 0x0E7, 0x079, 0x0DB, 0x0ED, 0x067, 0x07B, 0x0D9, 0x039, 0x01B, 0x019,
                                0x03B, 0x36D, 0x1E7};
  return (T[w] >> (2 * i)) & 3;
}
// ReedsSheppStateSpace::interpolate(from, path, t): the pose at fraction t of the path that starts at `from`
__device__ inline void rs_interpolate(const double* from, const RsPath& P, double t, double rho, double* out) {
  double seg = t * P.total;
  double sx = 0.0, sy = 0.0, yaw = from[2];
  for (int i = 0; i < 5 && seg > 0; ++i) {
    double v;
    if (P.len[i] < 0) {
      v = (-seg < P.len[i]) ? P.len[i] : -seg;   // std::max(-seg, length)
      seg += v;
    } else {
      v = (P.len[i] < seg) ? P.len[i] : seg;     // std::min(seg, length)
      seg -= v;
    }
    const double phi = yaw;
    const int ty = rs_segment(P.type, i);
    double s0, c0;
    det_sincos(phi, &s0, &c0);
    if (ty == 1) {
      double s1, c1;
      det_sincos(phi + v, &s1, &c1);
      sx = sx + s1 - s0;
      sy = sy - c1 + c0;
      yaw = phi + v;
    } else if (ty == 3) {
      double s1, c1;
      det_sincos(phi - v, &s1, &c1);
      sx = sx - s1 + s0;
      sy = sy + c1 - c0;
      yaw = phi - v;
    } else if (ty == 2) {
      sx = sx + v * c0;
      sy = sy + v * s0;
    }
  }
  out[0] = sx * rho + from[0];
  out[1] = sy * rho + from[1];
  double w = fmod(yaw, 2.0 * TOPAY_RS_PI);   // SO2StateSpace::enforceBounds
  if (w < -TOPAY_RS_PI) w += 2.0 * TOPAY_RS_PI;
  else if (w >= TOPAY_RS_PI) w -= 2.0 * TOPAY_RS_PI;
  out[2] = w;
}

// distance() and interpolate(from, to, t) for n pose pairs, one thread each (test hook and a caller's own edge checks)
__global__ void __launch_bounds__(64) k_reeds_shepp(int n, const double* from, const double* to, const double* t, double rho, double* dist, int* word, double* lengths,
                              double* pose) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  RsPath P;
  rs_between(from + 3 * (size_t)i, to + 3 * (size_t)i, rho, P);
  dist[i] = rho * P.total;
  word[i] = P.type;
  for (int k = 0; k < 5; k++) lengths[5 * (size_t)i + k] = P.len[k];
  if (t) {
    const double ti = t[i];
    double* o = pose + 3 * (size_t)i;
    if (ti >= 1.0) { o[0] = to[3 * (size_t)i]; o[1] = to[3 * (size_t)i + 1]; o[2] = to[3 * (size_t)i + 2]; }
    else if (ti <= 0.0) { o[0] = from[3 * (size_t)i]; o[1] = from[3 * (size_t)i + 1]; o[2] = from[3 * (size_t)i + 2]; }
    else rs_interpolate(from + 3 * (size_t)i, P, ti, rho, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// the search
// ---------------------------------------------------------------------------------------------------------------
#define TOPAY_MC_KEYW 5          // key bytes: 1 (layer) + 7 x up to 5 ("-1234") = 36 <= 40
enum { MC_EXPANDED = 1, MC_IN_TREE = 2, MC_IN_ANTI = 3 };   // MCRRTNode::NodeState (mcrrts.h:30-36)

struct McrrtBatch {
  int n, layer_cap;
  unsigned long long inst_base;
  const int* map_id;
  const long long* car_off;   // first entry of instance p's chassis path in `car`
  const int* car_len;         // L_p
  const double* car;          // ragged [sum L][4] = (x, y, theta, dt): getDensePath's output
  const double* start;        // n x 10
  const double* end;          // n x 10
  McrrtParams P;
  // node tables (n x node_cap) and Reeds-Shepp words (n x 2 layer_cap)
  int* nd_layer; int* nd_state; int* nd_parent; int* nd_nchild; int* nd_mark;
  double* nd_cost; double* nd_q; unsigned long long* nd_key;
  RsPath* rs;
  // results
  int* wb_len; double* wb; int* stats; double* cmax;
};

struct Mc {
  int lane, L, n, near_min, near_max, overflow, stamp;
  unsigned long long inst, iter;
  long long checks;
  DevMap M;
  McrrtParams P;
  const double* car;
  int* layer; int* state; int* parent; int* nchild; int* mark;
  double* cost; double* q; unsigned long long* key;
  const RsPath* rs;
};

__device__ __forceinline__ double mc_layer_time(const Mc& C, int a, int b) {
  double time = 0.0;
  const int lo = a < b ? a : b, hi = a > b ? a : b;
  for (int i = lo; i < hi; ++i) time += C.car[4 * i + 3];
  return time;
}
__device__ __forceinline__ double mc_est(const Mc& C, int l1, const double* q1, int l2, const double* q2) {   // estHeuristic
  const double time = mc_layer_time(C, l1, l2);
  double n1 = 0.0;
#pragma unroll
  for (int i = 0; i < 7; i++) n1 += fabs(q1[i] - q2[i]);
  return n1 / time;
}
__device__ __forceinline__ void mc_load_q(const Mc& C, int i, double* q) {
#pragma unroll
  for (int a = 0; a < 7; a++) q[a] = C.q[7 * (size_t)i + a];
}
// getKey: the layer as one character, then std::to_string((int) round(q * 100)) of the seven joints, concatenated; bytes
// packed most significant first, so that comparing the words as unsigned numbers is std::string's order
__device__ inline void mc_key(int layer, const double* q, unsigned long long* key) {
  for (int w = 0; w < TOPAY_MC_KEYW; w++) key[w] = 0ull;
  int pos = 0;
  auto put = [&](unsigned c) {
    if (pos < 8 * TOPAY_MC_KEYW) key[pos >> 3] |= (unsigned long long)(c & 0xFFu) << (8 * (7 - (pos & 7)));
    pos++;
  };
  put((unsigned)layer);
  for (int i = 0; i < 7; i++) {
    int v = (int)round(q[i] * 100.0);
    if (v < 0) put('-');
    unsigned m = v < 0 ? (unsigned)(-(long long)v) : (unsigned)v;
    unsigned div = 1;
    while (m / div >= 10) div *= 10;
    for (; div > 0; div /= 10) put('0' + (m / div) % 10);
  }
}
__device__ __forceinline__ int mc_key_less(const unsigned long long* a, const unsigned long long* b) {
  for (int w = 0; w < TOPAY_MC_KEYW; w++) {
    if (a[w] < b[w]) return 1;
    if (a[w] > b[w]) return 0;
  }
  return 0;
}
__device__ inline int mc_find(const Mc& C, const unsigned long long* key) {
  for (int base = 0; base < C.n; base += 64) {
    const int i = base + C.lane;
    bool eq = i < C.n;
    if (eq)
      for (int w = 0; w < TOPAY_MC_KEYW; w++) eq = eq && C.key[(size_t)i * TOPAY_MC_KEYW + w] == key[w];
    const unsigned long long m = __ballot(eq);
    if (m) return base + __ffsll((long long)m) - 1;
  }
  return -1;
}
// genNodeFromState: the node with this key, created (EXPANDED, no parent) when there is none
__device__ inline int mc_gen(Mc& C, int layer, const double* q) {
  unsigned long long key[TOPAY_MC_KEYW];
  mc_key(layer, q, key);
  const int f = mc_find(C, key);
  if (f >= 0) return f;
  if (C.n >= C.P.node_cap) { C.overflow = 1; return -1; }
  const int i = C.n;
  if (C.lane == 0) {
    C.layer[i] = layer; C.state[i] = MC_EXPANDED; C.parent[i] = -1; C.nchild[i] = 0; C.mark[i] = 0; C.cost[i] = 0.0;
    for (int a = 0; a < 7; a++) C.q[7 * (size_t)i + a] = q[a];
    for (int w = 0; w < TOPAY_MC_KEYW; w++) C.key[(size_t)i * TOPAY_MC_KEYW + w] = key[w];
  }
  C.n = i + 1;
  wave_global_sync();
  return i;
}
// connectCollision between neighbouring layers: true when any of the interpolated states is in whole-body collision
__device__ inline bool mc_edge_collides(Mc& C, int l_from, const double* q_from, int l_to, const double* q_to) {
  if (l_to < 0 || l_to >= C.L) return true;   // (the reference would index its chassis path out of range)
  const RsPath pth = (l_to == l_from + 1) ? C.rs[l_from] : C.rs[(C.L - 1) + l_to];
  const double* from = C.car + 4 * l_from;
  const int check_num_car = (int)ceil(C.P.rs_rho * pth.total / C.P.check_colli_res);
  double delta[7], linf = 0.0;
#pragma unroll
  for (int a = 0; a < 7; a++) {
    delta[a] = q_to[a] - q_from[a];
    linf = fmax(linf, fabs(delta[a]));
  }
  const int check_num_theta = (int)ceil(linf / C.P.check_colli_res);
  int pn = check_num_car > check_num_theta ? check_num_car : check_num_theta;
  if (pn < 3) pn = 3;
  const double piece = 1.0 * pn;
  for (int base = 0; base < pn; base += 64) {
    const int i = base + C.lane;
    bool hit = false;
    if (i < pn) {
      const double temp_i = 1.0 * i / piece;
      double st[10];
      if (temp_i <= 0.0) { st[0] = from[0]; st[1] = from[1]; st[2] = from[2]; }
      else rs_interpolate(from, pth, temp_i, C.P.rs_rho, st);
#pragma unroll
      for (int a = 0; a < 7; a++) st[3 + a] = q_from[a] + delta[a] * temp_i;
      hit = whole_body_collision(C.M, st);
    }
    C.checks += (pn - base < 64) ? pn - base : 64;
    if (__any(hit)) return true;
  }
  return false;
}
// steer (mcrrts.cpp:336-376): one layer from `node` towards the target state, joint speeds clamped
__device__ inline int mc_steer(Mc& C, int node, int tl, const double* tq) {
  const int nl = C.layer[node];
  double nq[7], snew[7];
  mc_load_q(C, node, nq);
  const double time = mc_layer_time(C, nl, tl);
  const bool anti = C.state[node] == MC_IN_ANTI;
  if (anti && nl < 1) return -1;   // (the reference would read car_path[-1])
  const int new_idx = anti ? nl - 1 : nl + 1;
  const double dtl = anti ? C.car[4 * (nl - 1) + 3] : C.car[4 * nl + 3];
#pragma unroll
  for (int i = 0; i < 7; i++) {
    double vel = (tq[i] - nq[i]) / time;
    const double lim = g_P.joint_vel_limit[i];
    vel = (lim < vel) ? lim : vel;        // std::min(vel, v_limit)
    vel = (vel < -lim) ? -lim : vel;      // std::max(.., -v_limit)
    snew[i] = nq[i] + vel * dtl;
  }
  if (mc_edge_collides(C, nl, nq, new_idx, snew)) return -1;
  return mc_gen(C, new_idx, snew);
}
// linkNode + updateCosts (mcrrts.h:163-171, 253-264).  The children of a node are the nodes whose parent it is; the
// recursion over them becomes a sweep over the table per generation (children lie one layer further from the root).
__device__ inline void mc_link(Mc& C, int par, int child) {
  const int pre = C.parent[child];
  if (pre == par) return;
  wave_global_sync();   // (every lane has read the old parent before lane 0 replaces it)
  if (C.lane == 0) {
    if (pre >= 0) C.nchild[pre] -= 1;
    C.parent[child] = par;
    C.nchild[par] += 1;
  }
  wave_global_sync();
  double qp[7], qc[7];
  mc_load_q(C, par, qp);
  mc_load_q(C, child, qc);
  const double now_cost = C.cost[par] + mc_est(C, C.layer[par], qp, C.layer[child], qc);
  const bool unchanged = C.cost[child] == now_cost;
  wave_global_sync();
  if (unchanged) return;
  if (C.lane == 0) C.cost[child] = now_cost;
  if (C.nchild[child] == 0) { wave_global_sync(); return; }
  int stamp = ++C.stamp;
  if (C.lane == 0) C.mark[child] = stamp;
  wave_global_sync();
  for (;;) {
    bool changed = false;
    for (int base = 0; base < C.n; base += 64) {
      const int i = base + C.lane;
      if (i < C.n) {
        const int p = C.parent[i];
        if (p >= 0 && C.mark[p] == stamp) {
          double a[7], b[7];
          mc_load_q(C, p, a);
          mc_load_q(C, i, b);
          const double nc = C.cost[p] + mc_est(C, C.layer[p], a, C.layer[i], b);
          if (!(C.cost[i] == nc)) {
            C.cost[i] = nc;
            C.mark[i] = stamp + 1;
            changed = true;
          }
        }
      }
    }
    wave_global_sync();
    stamp = ++C.stamp;
    if (!__any(changed)) break;
  }
}
__device__ __forceinline__ void mc_update_min_max(Mc& C, int node) {
  const int st = C.state[node], l = C.layer[node];
  if (st == MC_IN_TREE && l > C.near_max) C.near_max = l;
  if (st == MC_IN_ANTI && l < C.near_min) C.near_min = l;
}
// getNearestNode (mcrrts.h:231-251): the node of the tree (anti: of the anti-tree) in the layer next to the state's with
// the smallest estHeuristic below 1e12; the reference walks its std::map in key order and keeps the first minimum
__device__ inline int mc_nearest(const Mc& C, int sl, const double* sq, bool anti) {
  int near_layer;
  if (anti) near_layer = (sl + 1 > C.near_min) ? sl + 1 : C.near_min;
  else near_layer = (C.near_max < sl - 1) ? C.near_max : sl - 1;
  if (near_layer < 0 || near_layer >= C.L) return -1;
  const int want = anti ? MC_IN_ANTI : MC_IN_TREE;
  const double time = mc_layer_time(C, near_layer, sl);
  double best = 1.0e12;
  int besti = -1;
  for (int base = 0; base < C.n; base += 64) {
    const int i = base + C.lane;
    if (i < C.n && C.layer[i] == near_layer && C.state[i] == want) {
      double n1 = 0.0;
#pragma unroll
      for (int a = 0; a < 7; a++) n1 += fabs(C.q[7 * (size_t)i + a] - sq[a]);
      const double d = n1 / time;
      bool take = d < best;
      if (!take && besti >= 0 && d == best) take = mc_key_less(C.key + (size_t)i * TOPAY_MC_KEYW, C.key + (size_t)besti * TOPAY_MC_KEYW) != 0;
      if (take) { best = d; besti = i; }
    }
  }
  // across the lanes: smallest distance, ties to the smaller key
  double m = best;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_xor(m, off));
  unsigned long long cand = __ballot(besti >= 0 && best == m);
  int res = -1;
  while (cand) {
    const int l = __ffsll((long long)cand) - 1;
    cand &= cand - 1;
    const int i = __shfl(besti, l);
    if (res < 0 || mc_key_less(C.key + (size_t)i * TOPAY_MC_KEYW, C.key + (size_t)res * TOPAY_MC_KEYW)) res = i;
  }
  return res;
}
__device__ inline bool mc_feasible(const Mc& C, int a, int b) {   // feasibleCheck (mcrrts.h:293-308)
  const double time = mc_layer_time(C, C.layer[a], C.layer[b]);
  bool ok = true;
  for (int i = 0; i < 7; ++i) {
    double dif = C.q[7 * (size_t)b + i] - C.q[7 * (size_t)a + i];
    if (dif > TOPAY_RS_PI) dif = 2.0 * TOPAY_RS_PI - fabs(dif);
    const double vel = fabs(dif) / time;
    if (g_P.joint_vel_limit[i] - vel < 0.0) ok = false;
  }
  return ok;
}
// rewire (mcrrts.cpp:378-400): nodes of the next layer of q_new's tree that get cheaper through q_new.  (The reference
// visits them in key order; each re-link only changes costs further down its own subtree, so the order is immaterial.)
__device__ inline void mc_rewire(Mc& C, int q_new) {
  const int l = C.layer[q_new];
  if (l < 1) return;
  const int st = C.state[q_new];
  const bool anti = st == MC_IN_ANTI;
  const int next_idx = anti ? l - 1 : l + 1;
  if ((anti && next_idx < C.near_min) || (!anti && next_idx > C.near_max)) return;
  double qn[7];
  mc_load_q(C, q_new, qn);
  const int n0 = C.n;
  for (int t = 0; t < n0; t++) {
    if (t == q_new || C.layer[t] != next_idx || C.state[t] != st) continue;
    double qt[7];
    mc_load_q(C, t, qt);
    if (C.cost[t] > C.cost[q_new] + mc_est(C, l, qn, next_idx, qt) && mc_feasible(C, q_new, t) && !mc_edge_collides(C, l, qn, next_idx, qt))
      mc_link(C, q_new, t);
  }
}
// sampleState (mcrrts.h:210-229): a random inner layer and the first of max_sample_tries joint samples that is free of
// whole-body collision (the last one when none is)
__device__ inline void mc_sample(Mc& C, int& sl, double* sq) {
  const int n_in = C.L - 2;
  int idx = 1 + (int)floor(mcrrt_u01(C.P.seed, C.inst, C.iter, 1) * n_in);
  if (idx > n_in) idx = n_in;
  sl = idx;
  const int tries = C.P.max_sample_tries;
  for (int base = 0; base < tries; base += 64) {
    const int t = base + C.lane;
    double st[10];
    st[0] = C.car[4 * idx]; st[1] = C.car[4 * idx + 1]; st[2] = C.car[4 * idx + 2];
    bool ok = false;
    if (t < tries) {
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const double hi = g_P.joint_pos_limit_max[i], lo = -hi;   // joint_pos_limit_min == -max (moma_param.h:115-116)
        st[3 + i] = lo + (hi - lo) * mcrrt_u01(C.P.seed, C.inst, C.iter, 2 + 7 * (unsigned long long)t + i);
      }
      ok = !whole_body_collision(C.M, st);
    }
    C.checks += (tries - base < 64) ? tries - base : 64;
    const unsigned long long m = __ballot(ok);
    int src = -1;
    if (m) src = __ffsll((long long)m) - 1;
    else if (base + 64 >= tries) src = (tries - 1) - base;
    if (src >= 0) {
#pragma unroll
      for (int i = 0; i < 7; i++) sq[i] = __shfl(st[3 + i], src);
      return;
    }
  }
}

__global__ void __launch_bounds__(64) k_mcrrt(const DevMap* maps, const McrrtBatch B) {
  const int p = blockIdx.x;
  if (p >= B.n) return;
  Mc C;
  C.lane = threadIdx.x;
  C.L = B.car_len[p];
  C.P = B.P;
  C.M = maps[B.map_id[p]];
  C.car = B.car + 4 * B.car_off[p];
  const size_t nb = (size_t)p * B.P.node_cap;
  C.layer = B.nd_layer + nb; C.state = B.nd_state + nb; C.parent = B.nd_parent + nb; C.nchild = B.nd_nchild + nb; C.mark = B.nd_mark + nb;
  C.cost = B.nd_cost + nb; C.q = B.nd_q + 7 * nb; C.key = B.nd_key + TOPAY_MC_KEYW * nb;
  RsPath* rs = B.rs + (size_t)p * 2 * B.layer_cap;
  C.rs = rs;
  C.n = 0; C.overflow = 0; C.stamp = 0; C.checks = 0;
  C.inst = B.inst_base + (unsigned long long)p;
  C.iter = 0;
  const int L = C.L;
  C.near_min = L - 1;
  C.near_max = 0;
  int* stats = B.stats + 8 * (size_t)p;
  double* wb = B.wb + (size_t)p * B.layer_cap * 10;
  const double* start = B.start + 10 * (size_t)p;
  const double* end = B.end + 10 * (size_t)p;
  auto finish = [&](int status, int iters, int tc, int ac, int p1, int p2, double cmax, int wlen) {
    if (C.lane == 0) {
      stats[0] = status; stats[1] = C.n; stats[2] = iters; stats[3] = tc; stats[4] = ac; stats[5] = p1; stats[6] = p2;
      stats[7] = (int)(C.checks & 0x7fffffff);
      B.cmax[p] = cmax;
      B.wb_len[p] = wlen;
    }
  };
  if (L < 2 || L > B.layer_cap || L > 255) { finish(-2, 0, 0, 0, -1, -1, 1.0e+6, 0); return; }
  // the Reeds-Shepp words of the chassis path: [i] = layer i -> i + 1, [L - 1 + i] = layer i + 1 -> i
  for (int e = C.lane; e < 2 * (L - 1); e += 64) {
    const bool back = e >= L - 1;
    const int i = back ? e - (L - 1) : e;
    RsPath P;
    rs_between(C.car + 4 * (back ? i + 1 : i), C.car + 4 * (back ? i : i + 1), B.P.rs_rho, P);
    rs[e] = P;
  }
  wave_global_sync();
  double sq[7], eq[7];
  for (int a = 0; a < 7; a++) { sq[a] = start[3 + a]; eq[a] = end[3 + a]; }
  const int start_node = mc_gen(C, 0, sq);
  if (C.lane == 0) { C.state[start_node] = MC_IN_TREE; C.cost[start_node] = 0.0; }
  wave_global_sync();
  const int end_node = mc_gen(C, L - 1, eq);
  if (C.lane == 0) { C.state[end_node] = MC_IN_ANTI; C.cost[end_node] = 0.0; }
  wave_global_sync();
  if (L == 2) {   // mcrrts.cpp:25-33
    if (mc_edge_collides(C, 0, sq, 1, eq)) { finish(0, 0, 1, 1, -1, -1, 1.0e+6, 0); return; }
    if (C.lane < 10) { wb[C.lane] = start[C.lane]; wb[10 + C.lane] = end[C.lane]; }
    finish(1, 0, 1, 1, -1, -1, 1.0e+6, 2);
    return;
  }
  int tree_count = 1, anti_count = 1, iters = 0, p1 = -1, p2 = -1;
  bool connected = false;
  double c_max = 1.0e+6;
  auto count = [&](int node) { if (C.state[node] == MC_IN_TREE) ++tree_count; else ++anti_count; };
  auto set_state = [&](int node, int st) { if (C.lane == 0) C.state[node] = st; wave_global_sync(); };
  auto est_nodes = [&](int a, int b) {
    double qa[7], qb[7];
    mc_load_q(C, a, qa);
    mc_load_q(C, b, qb);
    return mc_est(C, C.layer[a], qa, C.layer[b], qb);
  };
  for (C.iter = 0; (int)C.iter < B.P.max_iter; C.iter++) {
    if (connected || C.overflow) break;
    iters++;
    const bool anti = tree_count > anti_count;
    int rl;
    double rq[7];
    if (mcrrt_u01(B.P.seed, C.inst, C.iter, 0) < B.P.goal_sample_rate) {
      rl = anti ? 0 : L - 1;
      for (int a = 0; a < 7; a++) rq[a] = anti ? sq[a] : eq[a];
    } else {
      mc_sample(C, rl, rq);
    }
    const int q_nearest = mc_nearest(C, rl, rq, anti);
    if (q_nearest < 0) continue;
    const int q_new = mc_steer(C, q_nearest, rl, rq);
    if (q_new < 0) continue;
    const int st_new = C.state[q_new], st_near = C.state[q_nearest];
    if ((!anti && st_new == MC_IN_ANTI) || (anti && st_new == MC_IN_TREE)) {   // q_new is a node of the other tree: connected
      connected = true;
      const double cost = C.cost[q_nearest] + C.cost[q_new] + est_nodes(q_nearest, q_new);
      if (cost < c_max) { c_max = cost; p1 = q_nearest; p2 = q_new; }
      continue;
    }
    if (st_new == MC_EXPANDED || (st_new == st_near && C.cost[q_new] > C.cost[q_nearest] + est_nodes(q_nearest, q_new))) {
      mc_link(C, q_nearest, q_new);
      set_state(q_new, st_near);
      count(q_new);
      mc_update_min_max(C, q_new);
      mc_rewire(C, q_new);
      const int nl = C.layer[q_new];
      double nq[7];
      mc_load_q(C, q_new, nq);
      const int q_near_opp = mc_nearest(C, nl, nq, !anti);
      if (q_near_opp < 0) continue;
      int q_new_opp = mc_steer(C, q_near_opp, nl, nq);
      if (q_new_opp < 0) continue;
      const int st_o = C.state[q_new_opp], st_no = C.state[q_near_opp];
      if ((anti && st_o == MC_IN_ANTI) || (!anti && st_o == MC_IN_TREE)) {
        connected = true;
        const double cost = C.cost[q_new_opp] + C.cost[q_near_opp] + est_nodes(q_new_opp, q_near_opp);
        if (cost < c_max) { c_max = cost; p1 = q_new_opp; p2 = q_near_opp; }
        continue;
      }
      if (st_o == MC_EXPANDED || (st_o == st_no && C.cost[q_new_opp] > C.cost[q_near_opp] + est_nodes(q_near_opp, q_new_opp))) {
        mc_link(C, q_near_opp, q_new_opp);
        set_state(q_new_opp, st_no);
        mc_update_min_max(C, q_new_opp);
        mc_rewire(C, q_new_opp);
        count(q_new_opp);
        // "try connecting tree once" (mcrrts.cpp:135-193): keep steering the opposite tree towards q_new
        while (C.layer[q_new] != C.layer[q_new_opp]) {
          const int q_new_2 = mc_steer(C, q_new_opp, nl, nq);
          if (q_new_2 < 0) break;
          const int s2 = C.state[q_new_2], so = C.state[q_new_opp];
          if (s2 == MC_EXPANDED || (s2 == so && C.cost[q_new_2] > C.cost[q_new_opp] + est_nodes(q_new_opp, q_new_2))) {
            mc_link(C, q_new_opp, q_new_2);
            set_state(q_new_2, so);
            mc_update_min_max(C, q_new_2);
            count(q_new_2);
            q_new_opp = q_new_2;
          } else if ((!anti && s2 == MC_IN_TREE) || (anti && s2 == MC_IN_ANTI)) {
            connected = true;
            const double cost = C.cost[q_new_2] + C.cost[q_new_opp] + est_nodes(q_new_2, q_new_opp);
            if (cost < c_max) { c_max = cost; p1 = q_new_2; p2 = q_new_opp; }
            break;
          } else if (s2 == so && C.cost[q_new_2] < C.cost[q_new_opp] + est_nodes(q_new_opp, q_new_2)) {
            q_new_opp = q_new_2;
          } else {
            break;
          }
        }
      }
    }
  }
  if (C.overflow) { finish(-1, iters, tree_count, anti_count, p1, p2, c_max, 0); return; }
  if (!connected) { finish(0, iters, tree_count, anti_count, p1, p2, c_max, 0); return; }
  // mergeTree + the walk from the end node (mcrrts.h:266-291, mcrrts.cpp:205-215) without re-linking the table: the tree
  // side from the IN_TREE member of the connecting pair up to the start, reversed, then the anti-tree side down to the end
  int wlen = 0;
  if (C.lane == 0) {
    const int q1 = C.state[p1] == MC_IN_TREE ? p1 : p2, q2 = C.state[p1] == MC_IN_TREE ? p2 : p1;
    int na = 0;
    for (int nidx = q1; nidx >= 0 && na <= L; nidx = C.parent[nidx]) na++;
    int k = na - 1;
    auto emit = [&](int slot, int nidx) {
      if (slot < 0 || slot >= B.layer_cap) return;
      const int l = C.layer[nidx];
      wb[10 * slot] = C.car[4 * l]; wb[10 * slot + 1] = C.car[4 * l + 1]; wb[10 * slot + 2] = C.car[4 * l + 2];
      for (int a = 0; a < 7; a++) wb[10 * slot + 3 + a] = C.q[7 * (size_t)nidx + a];
    };
    for (int nidx = q1; nidx >= 0 && k >= 0; nidx = C.parent[nidx]) emit(k--, nidx);
    wlen = na;
    for (int nidx = q2; nidx >= 0 && wlen <= L; nidx = C.parent[nidx]) emit(wlen++, nidx);
  }
  wlen = __shfl(wlen, 0);
  finish(1, iters, tree_count, anti_count, p1, p2, c_max, wlen);
}

}  // namespace topay
