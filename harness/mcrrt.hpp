// CPU restatement of the layered joint-space search that produces optimizeTraj's init paths:
//   MCRRTs::plan            /root/reference/src/planner/src/mcrrts.cpp:5-231
//   MCRRTs::steer, rewire   mcrrts.cpp:336-400
//   the inline members      /root/reference/src/planner/include/planner/mcrrts.h:120-348
// plus the one third-party piece the search calls on every edge: ompl::base::ReedsSheppStateSpace (distance,
// interpolate; mcrrts.h:318-324, 336).  OMPL is not part of /root/reference (the reference installs it with apt --
// Dockerfile:30-33, ros-noetic-ompl / libompl-dev, i.e. 1.5.2 / 1.4.2, no lock file); ReedsShepp below restates its
// published algorithm (Reeds & Shepp 1990, formulas 8.1-8.11, in the order and with the tie rules of OMPL's
// src/ompl/base/spaces/src/ReedsSheppStateSpace.cpp).  PARITY UNPINNED for that piece: no OMPL here to run it against.
//
// Test infrastructure only: it is the checker of topay_mcrrt_plan (the device search), never part of the product path.
//
// What is different from the reference, and the same in the device code (include/topay.h: topay_mcrrt_params_t):
//   * random numbers: the reference seeds std::mt19937 from std::random_device (mcrrts.h:89), so no run of it is
//     reproducible; here every draw is a pure function of (seed, instance, iteration, slot) -- mcrrt_u01 -- so that the
//     restatement and the device grow the same tree and can be compared node by node;
//   * wall-clock limits (max_time, mcrrts.cpp:43, 143, mcrrts.h:225) become counts: max_iter iterations of the main loop,
//     max_sample_tries resamplings in sampleState, node_cap nodes.
// Plain libm arithmetic in the reference's operation order (this file is built with -ffp-contract=off; the L1 norm of
// estHeuristic is summed in index order, Eigen's packet order is a build detail of the reference).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "workload.hpp"

namespace topay_wl {

// ------------------------------------------------------------------------------------------------------------------
// random numbers shared with the device (topay_amd/csrc/topay_mcrrt.h: mcrrt_u01)
// ------------------------------------------------------------------------------------------------------------------
inline uint64_t mcrrt_mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double mcrrt_u01(uint64_t seed, uint64_t inst, uint64_t iter, uint64_t slot) {
  uint64_t h = mcrrt_mix(seed + inst * 0x9E3779B97F4A7C15ull);
  h = mcrrt_mix(h + iter * 0xD1342543DE82EF95ull);
  h = mcrrt_mix(h + slot * 0x94D049BB133111EBull);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// ------------------------------------------------------------------------------------------------------------------
// ompl::base::ReedsSheppStateSpace, turning radius rho
// ------------------------------------------------------------------------------------------------------------------
struct ReedsShepp {
  enum Seg { NOP = 0, LEFT = 1, STRAIGHT = 2, RIGHT = 3 };
  struct Path {
    int type = 0;
    double length[5] = {1.7976931348623157e308, 0.0, 0.0, 0.0, 0.0};
    double total = 1.7976931348623157e308;
    Path() {}
    Path(int type_, double t, double u, double v, double w = 0.0, double x = 0.0) : type(type_) {
      length[0] = t; length[1] = u; length[2] = v; length[3] = w; length[4] = x;
      total = std::fabs(t) + std::fabs(u) + std::fabs(v) + std::fabs(w) + std::fabs(x);
    }
  };
  static const int* segments(int type) {
    static const int T[18][5] = {
        {LEFT, RIGHT, LEFT, NOP, NOP},         {RIGHT, LEFT, RIGHT, NOP, NOP},        {LEFT, RIGHT, LEFT, RIGHT, NOP},
        {RIGHT, LEFT, RIGHT, LEFT, NOP},       {LEFT, RIGHT, STRAIGHT, LEFT, NOP},    {RIGHT, LEFT, STRAIGHT, RIGHT, NOP},
        {LEFT, STRAIGHT, RIGHT, LEFT, NOP},    {RIGHT, STRAIGHT, LEFT, RIGHT, NOP},   {LEFT, RIGHT, STRAIGHT, RIGHT, NOP},
        {RIGHT, LEFT, STRAIGHT, LEFT, NOP},    {RIGHT, STRAIGHT, RIGHT, LEFT, NOP},   {LEFT, STRAIGHT, LEFT, RIGHT, NOP},
        {LEFT, STRAIGHT, RIGHT, NOP, NOP},     {RIGHT, STRAIGHT, LEFT, NOP, NOP},     {LEFT, STRAIGHT, LEFT, NOP, NOP},
        {RIGHT, STRAIGHT, RIGHT, NOP, NOP},    {LEFT, RIGHT, STRAIGHT, LEFT, RIGHT},  {RIGHT, LEFT, STRAIGHT, RIGHT, LEFT}};
    return T[type];
  }
  static constexpr double pi = 3.14159265358979323846;
  static constexpr double twopi = 2.0 * pi;
  static constexpr double ZERO = 10.0 * 2.220446049250313e-16;

  double rho;
  explicit ReedsShepp(double rho_) : rho(rho_) {}

  static double mod2pi(double x) {
    double v = std::fmod(x, twopi);
    if (v < -pi) v += twopi;
    else if (v > pi) v -= twopi;
    return v;
  }
  static void polar(double x, double y, double& r, double& theta) {
    r = std::sqrt(x * x + y * y);
    theta = std::atan2(y, x);
  }
  static void tauOmega(double u, double v, double xi, double eta, double phi, double& tau, double& omega) {
    double delta = mod2pi(u - v), A = std::sin(u) - std::sin(delta), B = std::cos(u) - std::cos(delta) - 1.0;
    double t1 = std::atan2(eta * A - xi * B, xi * A + eta * B), t2 = 2.0 * (std::cos(delta) - std::cos(v) - std::cos(u)) + 3;
    tau = (t2 < 0) ? mod2pi(t1 + pi) : mod2pi(t1);
    omega = mod2pi(tau - u + v - phi);
  }
  // formula 8.1
  static bool LpSpLp(double x, double y, double phi, double& t, double& u, double& v) {
    polar(x - std::sin(phi), y - 1.0 + std::cos(phi), u, t);
    if (t >= -ZERO) {
      v = mod2pi(phi - t);
      if (v >= -ZERO) return true;
    }
    return false;
  }
  // formula 8.2
  static bool LpSpRp(double x, double y, double phi, double& t, double& u, double& v) {
    double t1, u1;
    polar(x + std::sin(phi), y - 1.0 - std::cos(phi), u1, t1);
    u1 = u1 * u1;
    if (u1 >= 4.0) {
      double theta;
      u = std::sqrt(u1 - 4.0);
      theta = std::atan2(2.0, u);
      t = mod2pi(t1 + theta);
      v = mod2pi(t - phi);
      return t >= -ZERO && v >= -ZERO;
    }
    return false;
  }
  // formula 8.3 / 8.4
  static bool LpRmL(double x, double y, double phi, double& t, double& u, double& v) {
    double xi = x - std::sin(phi), eta = y - 1.0 + std::cos(phi), u1, theta;
    polar(xi, eta, u1, theta);
    if (u1 <= 4.0) {
      u = -2.0 * std::asin(0.25 * u1);
      t = mod2pi(theta + 0.5 * u + pi);
      v = mod2pi(phi - t + u);
      return t >= -ZERO && u <= ZERO;
    }
    return false;
  }
  // formula 8.7
  static bool LpRupLumRm(double x, double y, double phi, double& t, double& u, double& v) {
    double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho_ = 0.25 * (2.0 + std::sqrt(xi * xi + eta * eta));
    if (rho_ <= 1.0) {
      u = std::acos(rho_);
      tauOmega(u, -u, xi, eta, phi, t, v);
      return t >= -ZERO && v <= ZERO;
    }
    return false;
  }
  // formula 8.8
  static bool LpRumLumRp(double x, double y, double phi, double& t, double& u, double& v) {
    double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho_ = (20.0 - xi * xi - eta * eta) / 16.0;
    if (rho_ >= 0 && rho_ <= 1) {
      u = -std::acos(rho_);
      if (u >= -0.5 * pi) {
        tauOmega(u, u, xi, eta, phi, t, v);
        return t >= -ZERO && v >= -ZERO;
      }
    }
    return false;
  }
  // formula 8.9
  static bool LpRmSmLm(double x, double y, double phi, double& t, double& u, double& v) {
    double xi = x - std::sin(phi), eta = y - 1.0 + std::cos(phi), rho_, theta;
    polar(xi, eta, rho_, theta);
    if (rho_ >= 2.0) {
      double r = std::sqrt(rho_ * rho_ - 4.0);
      u = 2.0 - r;
      t = mod2pi(theta + std::atan2(r, -2.0));
      v = mod2pi(phi - 0.5 * pi - t);
      return t >= -ZERO && u <= ZERO && v <= ZERO;
    }
    return false;
  }
  // formula 8.10
  static bool LpRmSmRm(double x, double y, double phi, double& t, double& u, double& v) {
    double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho_, theta;
    polar(-eta, xi, rho_, theta);
    if (rho_ >= 2.0) {
      t = theta;
      u = 2.0 - rho_;
      v = mod2pi(t + 0.5 * pi - phi);
      return t >= -ZERO && u <= ZERO && v <= ZERO;
    }
    return false;
  }
  // formula 8.11
  static bool LpRmSLmRp(double x, double y, double phi, double& t, double& u, double& v) {
    double xi = x + std::sin(phi), eta = y - 1.0 - std::cos(phi), rho_, theta;
    polar(xi, eta, rho_, theta);
    if (rho_ >= 2.0) {
      u = 4.0 - std::sqrt(rho_ * rho_ - 4.0);
      if (u <= ZERO) {
        t = mod2pi(std::atan2((4.0 - u) * xi - 2.0 * eta, -2.0 * xi + (u - 4.0) * eta));
        v = mod2pi(t - phi);
        return t >= -ZERO && v >= -ZERO;
      }
    }
    return false;
  }
  static void CSC(double x, double y, double phi, Path& path) {
    double t, u, v, Lmin = path.total, L;
    if (LpSpLp(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(14, t, u, v); Lmin = L; }
    if (LpSpLp(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(14, -t, -u, -v); Lmin = L; }   // timeflip
    if (LpSpLp(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(15, t, u, v); Lmin = L; }      // reflect
    if (LpSpLp(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(15, -t, -u, -v); Lmin = L; }   // timeflip + reflect
    if (LpSpRp(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(12, t, u, v); Lmin = L; }
    if (LpSpRp(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(12, -t, -u, -v); Lmin = L; }
    if (LpSpRp(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(13, t, u, v); Lmin = L; }
    if (LpSpRp(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(13, -t, -u, -v); Lmin = L; }
  }
  static void CCC(double x, double y, double phi, Path& path) {
    double t, u, v, Lmin = path.total, L;
    if (LpRmL(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(0, t, u, v); Lmin = L; }
    if (LpRmL(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(0, -t, -u, -v); Lmin = L; }
    if (LpRmL(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(1, t, u, v); Lmin = L; }
    if (LpRmL(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(1, -t, -u, -v); Lmin = L; }
    // backwards
    double xb = x * std::cos(phi) + y * std::sin(phi), yb = x * std::sin(phi) - y * std::cos(phi);
    if (LpRmL(xb, yb, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(0, v, u, t); Lmin = L; }
    if (LpRmL(-xb, yb, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(0, -v, -u, -t); Lmin = L; }
    if (LpRmL(xb, -yb, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(1, v, u, t); Lmin = L; }
    if (LpRmL(-xb, -yb, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(1, -v, -u, -t); Lmin = L; }
  }
  static void CCCC(double x, double y, double phi, Path& path) {
    double t, u, v, Lmin = path.total, L;
    if (LpRupLumRm(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(2, t, u, -u, v); Lmin = L; }
    if (LpRupLumRm(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(2, -t, -u, u, -v); Lmin = L; }
    if (LpRupLumRm(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(3, t, u, -u, v); Lmin = L; }
    if (LpRupLumRm(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(3, -t, -u, u, -v); Lmin = L; }
    if (LpRumLumRp(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(2, t, u, u, v); Lmin = L; }
    if (LpRumLumRp(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(2, -t, -u, -u, -v); Lmin = L; }
    if (LpRumLumRp(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(3, t, u, u, v); Lmin = L; }
    if (LpRumLumRp(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + 2.0 * std::fabs(u) + std::fabs(v))) { path = Path(3, -t, -u, -u, -v); Lmin = L; }
  }
  static void CCSC(double x, double y, double phi, Path& path) {
    double t, u, v, Lmin = path.total - 0.5 * pi, L;
    if (LpRmSmLm(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(4, t, -0.5 * pi, u, v); Lmin = L; }
    if (LpRmSmLm(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(4, -t, 0.5 * pi, -u, -v); Lmin = L; }
    if (LpRmSmLm(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(5, t, -0.5 * pi, u, v); Lmin = L; }
    if (LpRmSmLm(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(5, -t, 0.5 * pi, -u, -v); Lmin = L; }
    if (LpRmSmRm(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(8, t, -0.5 * pi, u, v); Lmin = L; }
    if (LpRmSmRm(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(8, -t, 0.5 * pi, -u, -v); Lmin = L; }
    if (LpRmSmRm(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(9, t, -0.5 * pi, u, v); Lmin = L; }
    if (LpRmSmRm(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(9, -t, 0.5 * pi, -u, -v); Lmin = L; }
    // backwards
    double xb = x * std::cos(phi) + y * std::sin(phi), yb = x * std::sin(phi) - y * std::cos(phi);
    if (LpRmSmLm(xb, yb, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(6, v, u, -0.5 * pi, t); Lmin = L; }
    if (LpRmSmLm(-xb, yb, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(6, -v, -u, 0.5 * pi, -t); Lmin = L; }
    if (LpRmSmLm(xb, -yb, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(7, v, u, -0.5 * pi, t); Lmin = L; }
    if (LpRmSmLm(-xb, -yb, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(7, -v, -u, 0.5 * pi, -t); Lmin = L; }
    if (LpRmSmRm(xb, yb, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(10, v, u, -0.5 * pi, t); Lmin = L; }
    if (LpRmSmRm(-xb, yb, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(10, -v, -u, 0.5 * pi, -t); Lmin = L; }
    if (LpRmSmRm(xb, -yb, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(11, v, u, -0.5 * pi, t); Lmin = L; }
    if (LpRmSmRm(-xb, -yb, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(11, -v, -u, 0.5 * pi, -t); Lmin = L; }
  }
  static void CCSCC(double x, double y, double phi, Path& path) {
    double t, u, v, Lmin = path.total - pi, L;
    if (LpRmSLmRp(x, y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(16, t, -0.5 * pi, u, -0.5 * pi, v); Lmin = L; }
    if (LpRmSLmRp(-x, y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(16, -t, 0.5 * pi, -u, 0.5 * pi, -v); Lmin = L; }
    if (LpRmSLmRp(x, -y, -phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(17, t, -0.5 * pi, u, -0.5 * pi, v); Lmin = L; }
    if (LpRmSLmRp(-x, -y, phi, t, u, v) && Lmin > (L = std::fabs(t) + std::fabs(u) + std::fabs(v))) { path = Path(17, -t, 0.5 * pi, -u, 0.5 * pi, -v); Lmin = L; }
  }
  static Path reedsShepp(double x, double y, double phi) {
    Path path;
    CSC(x, y, phi, path);
    CCC(x, y, phi, path);
    CCCC(x, y, phi, path);
    CCSC(x, y, phi, path);
    CCSCC(x, y, phi, path);
    return path;
  }
  Path reedsShepp(const double from[3], const double to[3]) const {
    double x1 = from[0], y1 = from[1], th1 = from[2];
    double x2 = to[0], y2 = to[1], th2 = to[2];
    double dx = x2 - x1, dy = y2 - y1, c = std::cos(th1), s = std::sin(th1);
    double x = c * dx + s * dy, y = -s * dx + c * dy, phi = th2 - th1;
    return reedsShepp(x / rho, y / rho, phi);
  }
  double distance(const double from[3], const double to[3]) const { return rho * reedsShepp(from, to).total; }
  // interpolate(from, to, t, state): the path is computed on every call (firstTime = true)
  void interpolate(const double from[3], const double to[3], double t, double out[3]) const {
    if (t >= 1.0) { out[0] = to[0]; out[1] = to[1]; out[2] = to[2]; return; }
    if (t <= 0.0) { out[0] = from[0]; out[1] = from[1]; out[2] = from[2]; return; }
    interpolate(from, reedsShepp(from, to), t, out);
  }
  void interpolate(const double from[3], const Path& path, double t, double out[3]) const {
    double seg = t * path.total, phi, v;
    double sx = 0.0, sy = 0.0, syaw = from[2];
    const int* type = segments(path.type);
    for (unsigned int i = 0; i < 5 && seg > 0; ++i) {
      if (path.length[i] < 0) {
        v = std::max(-seg, path.length[i]);
        seg += v;
      } else {
        v = std::min(seg, path.length[i]);
        seg -= v;
      }
      phi = syaw;
      switch (type[i]) {
        case LEFT:
          sx = sx + std::sin(phi + v) - std::sin(phi);
          sy = sy - std::cos(phi + v) + std::cos(phi);
          syaw = phi + v;
          break;
        case RIGHT:
          sx = sx - std::sin(phi - v) + std::sin(phi);
          sy = sy + std::cos(phi - v) - std::cos(phi);
          syaw = phi - v;
          break;
        case STRAIGHT:
          sx = sx + v * std::cos(phi);
          sy = sy + v * std::sin(phi);
          break;
        case NOP:
          break;
      }
    }
    out[0] = sx * rho + from[0];
    out[1] = sy * rho + from[1];
    // SO2StateSpace::enforceBounds
    double w = std::fmod(syaw, 2.0 * pi);
    if (w < -pi) w += 2.0 * pi;
    else if (w >= pi) w -= 2.0 * pi;
    out[2] = w;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// MCRRTs
// ------------------------------------------------------------------------------------------------------------------
struct McrrtParams {          // == topay_mcrrt_params_t (include/topay.h)
  double goal_sample_rate = 0.4;   // params/mcrrts.yaml
  double check_colli_res = 0.01;
  double rs_turning_radius = 1.0e-2;   // mcrrts.h:134
  int max_iter = 1000;
  int max_sample_tries = 64;
  int node_cap = 2048;
  int reserved = 0;
  uint64_t seed = 42;
};

struct McrrtNodeRec {   // one row of the node table, in creation order (what the tests compare)
  int layer, state, parent;
  double cost;
  double q[7];
};

struct MCRRTs {
  enum NodeState { INIT, EXPANDED, IN_TREE, IN_ANTI_TREE };
  struct Node {
    std::string key;
    double cost = 0.0;   // (uninitialised in the reference; never read before linkNode sets it)
    NodeState node_state = INIT;
    int layer = 0;       // robo_state.first
    double q[7];         // robo_state.second
    Node* parent = nullptr;
    std::map<std::string, Node*> children;
    int index = 0;       // creation order (not in the reference; for the node-by-node comparison)
  };
  typedef Node* NodePtr;
  struct MCState { int first; double second[7]; };

  const World& world;
  McrrtParams prm;
  ReedsShepp reeds_shepp;
  uint64_t inst = 0, iter = 0;
  static constexpr int state_dim = 7;
  double joint_vel_limit[7];

  int near_min_idx = 1 << 20, near_max_idx = 0;
  bool connected = false;
  double c_max = 1.0e+6;
  std::vector<std::array<double, 4>> car_path;
  std::map<std::string, NodePtr> node_pool;
  std::vector<NodePtr> by_index;
  bool overflow = false;
  long long n_checks = 0;
  double min_slack = 1e300;   // smallest margin of any discrete decision taken (collision thresholds, ceil of the check counts)
  bool track_slack = false;
  int tree_count_ = 1, anti_tree_count_ = 1, iterations = 0;
  NodePtr path_node_1 = nullptr, path_node_2 = nullptr;

  MCRRTs(const World& w, const McrrtParams& p, uint64_t instance) : world(w), prm(p), reeds_shepp(p.rs_turning_radius), inst(instance) {
    for (int i = 0; i < 7; i++) joint_vel_limit[i] = 2.35;   // moma_param.h:118
  }
  ~MCRRTs() { for (auto& kv : node_pool) delete kv.second; }

  void reset(const std::vector<std::array<double, 4>>& path) {   // mcrrts.h:139-151
    connected = false;
    for (auto& kv : node_pool) delete kv.second;
    node_pool.clear();
    by_index.clear();
    c_max = 1.0e+6;
    car_path = path;
    near_min_idx = (int)car_path.size() - 1;
    near_max_idx = 0;
  }
  void updateMinMaxIdx(NodePtr node) {
    if (node->node_state == IN_TREE && node->layer > near_max_idx) near_max_idx = node->layer;
    if (node->node_state == IN_ANTI_TREE && node->layer < near_min_idx) near_min_idx = node->layer;
  }
  double layerTime(int a, int b) const {
    double time = 0.0;
    int min_idx = (a < b) ? a : b, max_idx = (a > b) ? a : b;
    for (int i = min_idx; i < max_idx; ++i) time += car_path[i][3];
    return time;
  }
  double estHeuristic(int l1, const double* q1, int l2, const double* q2) const {   // mcrrts.h:174-183
    double time = layerTime(l1, l2);
    double n1 = 0.0;
    for (int i = 0; i < state_dim; i++) n1 += std::fabs(q1[i] - q2[i]);
    return n1 / time;
  }
  double estHeuristic(NodePtr a, NodePtr b) const { return estHeuristic(a->layer, a->q, b->layer, b->q); }
  void updateCosts(NodePtr node) {   // mcrrts.h:163-171
    double now_cost = node->parent->cost + estHeuristic(node->parent, node);
    if (node->cost == now_cost) return;
    node->cost = now_cost;
    for (auto& kv : node->children) updateCosts(kv.second);
  }
  static std::string getKey(int layer, const double* q) {   // mcrrts.h:185-191
    std::string res(1, (char)layer);
    for (int i = 0; i < state_dim; ++i) res += std::to_string((int)(std::round(q[i] * 100.0)));
    return res;
  }
  NodePtr genNodeFromState(int layer, const double* q) {   // mcrrts.h:193-208
    std::string key = getKey(layer, q);
    auto it = node_pool.find(key);
    if (it != node_pool.end()) return it->second;
    if ((int)by_index.size() >= prm.node_cap) { overflow = true; return nullptr; }
    NodePtr node = new Node;
    node->node_state = EXPANDED;
    node->layer = layer;
    std::memcpy(node->q, q, sizeof(node->q));
    node->key = key;
    node->index = (int)by_index.size();
    by_index.push_back(node);
    node_pool.insert(std::make_pair(key, node));
    return node;
  }
  bool wholeBody(const double* st) {
    n_checks++;
    if (track_slack) min_slack = std::min(min_slack, world.robot.wholeBodyTieSlack(world.gm, st));
    return world.robot.isWholeBodyCollision(world.gm, st);
  }
  MCState sampleState() {   // mcrrts.h:210-229; draws: slot 1 = layer, slot 2 + 7 t + j = joint j of try t
    int n_in = (int)car_path.size() - 2;
    int idx = 1 + (int)std::floor(mcrrt_u01(prm.seed, inst, iter, 1) * n_in);
    if (idx > n_in) idx = n_in;
    double full[10];
    full[0] = car_path[idx][0]; full[1] = car_path[idx][1]; full[2] = car_path[idx][2];
    MCState s;
    s.first = idx;
    for (int t = 0; t < prm.max_sample_tries; t++) {
      for (int i = 0; i < state_dim; ++i)
        s.second[i] = world.robot.qmin[i] + (world.robot.qmax[i] - world.robot.qmin[i]) * mcrrt_u01(prm.seed, inst, iter, 2 + 7 * t + i);
      std::memcpy(full + 3, s.second, sizeof(s.second));
      if (!wholeBody(full)) break;
    }
    return s;
  }
  NodePtr getNearestNode(const MCState& state, bool anti) {   // mcrrts.h:231-251
    int near_layer = anti ? std::max(state.first + 1, near_min_idx) : std::min(near_max_idx, state.first - 1);
    NodePtr q_near = nullptr;
    double min_dis = 1.0e12;
    for (auto it = node_pool.lower_bound(std::string(1, (char)near_layer)); it != node_pool.end() && it->second->layer == near_layer; ++it) {
      NodePtr node = it->second;
      if ((!anti && node->node_state == IN_TREE) || (anti && node->node_state == IN_ANTI_TREE)) {
        if (estHeuristic(node->layer, node->q, state.first, state.second) < min_dis) {
          min_dis = estHeuristic(node->layer, node->q, state.first, state.second);
          q_near = node;
        }
      }
    }
    return q_near;
  }
  void linkNode(NodePtr parent, NodePtr child) {   // mcrrts.h:253-264
    NodePtr pre_parent = child->parent;
    if (pre_parent == parent) return;
    if (pre_parent != nullptr) pre_parent->children.erase(child->key);
    child->parent = parent;
    parent->children.insert(std::make_pair(child->key, child));
    updateCosts(child);
  }
  bool feasibleCheck(NodePtr a, NodePtr b) const {   // mcrrts.h:293-308
    double time = layerTime(a->layer, b->layer);
    for (int i = 0; i < state_dim; ++i) {
      double dif = b->q[i] - a->q[i];
      if (dif > M_PI) dif = 2.0 * M_PI - std::fabs(dif);
      double vel = std::fabs(dif) / time;
      if (joint_vel_limit[i] - vel < 0.0) return false;
    }
    return true;
  }
  bool connectCollision(int l_cur, const double* q_cur, int l_next, const double* q_next) {   // mcrrts.h:310-348
    if (l_next < 0 || l_next >= (int)car_path.size()) return true;   // (the reference would index car_path out of range)
    double cur_state[10], next_state[10];
    for (int a = 0; a < 3; a++) { cur_state[a] = car_path[l_cur][a]; next_state[a] = car_path[l_next][a]; }
    std::memcpy(cur_state + 3, q_cur, 7 * sizeof(double));
    std::memcpy(next_state + 3, q_next, 7 * sizeof(double));
    const double dist_over_res = reeds_shepp.distance(cur_state, next_state) / prm.check_colli_res;
    int check_num_car = (int)std::ceil(dist_over_res);
    double delta_theta[7], linf = 0.0;
    for (int i = 0; i < 7; i++) { delta_theta[i] = next_state[3 + i] - cur_state[3 + i]; linf = std::max(linf, std::fabs(delta_theta[i])); }
    int check_num_theta = (int)std::ceil(linf / prm.check_colli_res);
    if (track_slack) {
      min_slack = std::min(min_slack, std::fabs(dist_over_res - std::round(dist_over_res)) + (dist_over_res < 3.0 ? 1.0 : 0.0));
      min_slack = std::min(min_slack, std::fabs(linf / prm.check_colli_res - std::round(linf / prm.check_colli_res)) + (linf / prm.check_colli_res < 3.0 ? 1.0 : 0.0));
    }
    double piece_num_temp = 1.0 * std::max(std::max(check_num_car, check_num_theta), 3);
    for (int i = 0; i < piece_num_temp; ++i) {
      double temp_state[10];
      double temp_i = 1.0 * i / piece_num_temp;
      reeds_shepp.interpolate(cur_state, next_state, temp_i, temp_state);
      for (int a = 0; a < 7; a++) temp_state[3 + a] = cur_state[3 + a] + delta_theta[a] * temp_i;
      if (wholeBody(temp_state)) return true;
    }
    return false;
  }
  NodePtr steer(NodePtr node, const MCState& target_state) {   // mcrrts.cpp:336-376
    double diff[7], vel[7], state_new[7];
    for (int i = 0; i < 7; i++) diff[i] = target_state.second[i] - node->q[i];
    double time = layerTime(node->layer, target_state.first);
    for (int i = 0; i < state_dim; ++i) {
      vel[i] = diff[i] / time;
      double v_limit = joint_vel_limit[i];
      vel[i] = std::max(std::min(vel[i], v_limit), -v_limit);
    }
    int new_idx;
    bool anti = (node->node_state == IN_ANTI_TREE);
    if (anti) {
      if (node->layer < 1) return nullptr;   // (the reference would read car_path[-1])
      for (int i = 0; i < 7; i++) state_new[i] = node->q[i] + vel[i] * car_path[node->layer - 1][3];
      new_idx = node->layer - 1;
    } else {
      for (int i = 0; i < 7; i++) state_new[i] = node->q[i] + vel[i] * car_path[node->layer][3];
      new_idx = node->layer + 1;
    }
    if (connectCollision(node->layer, node->q, new_idx, state_new)) return nullptr;
    return genNodeFromState(new_idx, state_new);
  }
  void rewire(NodePtr q_new) {   // mcrrts.cpp:378-400
    if (q_new == nullptr || q_new->layer < 1) return;
    bool anti = (q_new->node_state == IN_ANTI_TREE);
    int next_idx = anti ? q_new->layer - 1 : q_new->layer + 1;
    if ((anti && next_idx < near_min_idx) || (!anti && next_idx > near_max_idx)) return;
    for (auto it = node_pool.lower_bound(std::string(1, (char)next_idx)); it != node_pool.end() && it->second->layer == next_idx; ++it) {
      NodePtr q_temp = it->second;
      if (q_temp->node_state != q_new->node_state) continue;
      if (q_temp->cost > q_new->cost + estHeuristic(q_new, q_temp) && feasibleCheck(q_new, q_temp) &&
          !connectCollision(q_new->layer, q_new->q, q_temp->layer, q_temp->q))
        linkNode(q_new, q_temp);
    }
  }

  // MCRRTs::plan, mcrrts.cpp:5-231.  start / end: (x, y, theta, q1..q7).  Returns 1 (path), 0 (none), -1 (node pool full).
  int plan(const double* start, const double* end, const std::vector<std::array<double, 4>>& path, std::vector<std::array<double, 10>>& wb_path) {
    reset(path);
    wb_path.clear();
    const int L = (int)path.size();
    NodePtr start_node = genNodeFromState(0, start + 3);
    start_node->node_state = IN_TREE;
    start_node->cost = 0.0;
    NodePtr end_node = genNodeFromState(L - 1, end + 3);
    end_node->node_state = IN_ANTI_TREE;
    end_node->cost = 0.0;
    tree_count_ = 1;
    anti_tree_count_ = 1;
    bool anti = false;
    path_node_1 = path_node_2 = nullptr;
    iterations = 0;
    if (L == 2) {
      if (connectCollision(start_node->layer, start_node->q, end_node->layer, end_node->q)) return 0;
      std::array<double, 10> a, b;
      std::memcpy(a.data(), start, sizeof(a));
      std::memcpy(b.data(), end, sizeof(b));
      wb_path.push_back(a);
      wb_path.push_back(b);
      return 1;
    }
    MCState start_state, end_state;
    start_state.first = 0; std::memcpy(start_state.second, start + 3, sizeof(start_state.second));
    end_state.first = L - 1; std::memcpy(end_state.second, end + 3, sizeof(end_state.second));
    auto as_state = [](NodePtr n) { MCState s; s.first = n->layer; std::memcpy(s.second, n->q, sizeof(s.second)); return s; };
    for (iter = 0; (int)iter < prm.max_iter; iter++) {
      if (connected || overflow) break;
      iterations++;
      anti = tree_count_ > anti_tree_count_;
      MCState rand_state;
      if (mcrrt_u01(prm.seed, inst, iter, 0) < prm.goal_sample_rate) rand_state = anti ? start_state : end_state;
      else rand_state = sampleState();
      NodePtr q_nearest = getNearestNode(rand_state, anti);
      if (q_nearest == nullptr) continue;
      NodePtr q_new = steer(q_nearest, rand_state);
      if (q_new == nullptr) continue;
      if ((!anti && q_new->node_state == IN_ANTI_TREE) || (anti && q_new->node_state == IN_TREE)) {
        connected = true;
        double cost = q_nearest->cost + q_new->cost + estHeuristic(q_nearest, q_new);
        if (cost < c_max) { c_max = cost; path_node_1 = q_nearest; path_node_2 = q_new; }
        continue;
      }
      if (q_new->node_state == EXPANDED ||
          (q_new->node_state == q_nearest->node_state && q_new->cost > q_nearest->cost + estHeuristic(q_nearest, q_new))) {
        linkNode(q_nearest, q_new);
        q_new->node_state = q_nearest->node_state;
        if (q_new->node_state == IN_TREE) ++tree_count_;
        else ++anti_tree_count_;
        updateMinMaxIdx(q_new);
        rewire(q_new);
        NodePtr q_near_opp = getNearestNode(as_state(q_new), !anti);
        if (q_near_opp == nullptr) continue;
        NodePtr q_new_opp = steer(q_near_opp, as_state(q_new));
        if (q_new_opp == nullptr) continue;
        if ((anti && q_new_opp->node_state == IN_ANTI_TREE) || (!anti && q_new_opp->node_state == IN_TREE)) {
          connected = true;
          double cost = q_new_opp->cost + q_near_opp->cost + estHeuristic(q_new_opp, q_near_opp);
          if (cost < c_max) { c_max = cost; path_node_1 = q_new_opp; path_node_2 = q_near_opp; }
          continue;
        }
        if (q_new_opp->node_state == EXPANDED ||
            (q_new_opp->node_state == q_near_opp->node_state && q_new_opp->cost > q_near_opp->cost + estHeuristic(q_near_opp, q_new_opp))) {
          linkNode(q_near_opp, q_new_opp);
          q_new_opp->node_state = q_near_opp->node_state;
          updateMinMaxIdx(q_new_opp);
          rewire(q_new_opp);
          if (q_new_opp->node_state == IN_TREE) ++tree_count_;
          else ++anti_tree_count_;
          // try connecting tree once
          while (q_new->layer != q_new_opp->layer) {
            NodePtr q_new_2 = steer(q_new_opp, as_state(q_new));
            if (q_new_2 == nullptr) break;
            if (q_new_2->node_state == EXPANDED ||
                (q_new_2->node_state == q_new_opp->node_state && q_new_2->cost > q_new_opp->cost + estHeuristic(q_new_opp, q_new_2))) {
              linkNode(q_new_opp, q_new_2);
              q_new_2->node_state = q_new_opp->node_state;
              updateMinMaxIdx(q_new_2);
              if (q_new_2->node_state == IN_TREE) ++tree_count_;
              else ++anti_tree_count_;
              q_new_opp = q_new_2;
            } else if ((!anti && q_new_2->node_state == IN_TREE) || (anti && q_new_2->node_state == IN_ANTI_TREE)) {
              connected = true;
              double cost = q_new_2->cost + q_new_opp->cost + estHeuristic(q_new_2, q_new_opp);
              if (cost < c_max) { c_max = cost; path_node_1 = q_new_2; path_node_2 = q_new_opp; }
              break;
            } else if (q_new_2->node_state == q_new_opp->node_state && q_new_2->cost < q_new_opp->cost + estHeuristic(q_new_opp, q_new_2)) {
              q_new_opp = q_new_2;
            } else {
              break;
            }
          }
        }
      }
    }
    if (overflow) return -1;
    if (!connected) return 0;
    // mergeTree (mcrrts.h:266-291) + the walk from end_node (mcrrts.cpp:205-215), without touching the node table: the
    // tree side from path_node's IN_TREE member up to the start, reversed, then the anti-tree side down to the end
    NodePtr q1, q2;
    if (path_node_1->node_state == IN_TREE) { q1 = path_node_1; q2 = path_node_2; }
    else { q1 = path_node_2; q2 = path_node_1; }
    std::vector<NodePtr> chain;
    for (NodePtr n = q1; n != nullptr; n = n->parent) chain.push_back(n);
    std::reverse(chain.begin(), chain.end());
    for (NodePtr n = q2; n != nullptr; n = n->parent) chain.push_back(n);
    for (NodePtr n : chain) {
      std::array<double, 10> full;
      for (int a = 0; a < 3; a++) full[a] = car_path[n->layer][a];
      std::memcpy(full.data() + 3, n->q, 7 * sizeof(double));
      wb_path.push_back(full);
    }
    return 1;
  }
};

}  // namespace topay_wl
