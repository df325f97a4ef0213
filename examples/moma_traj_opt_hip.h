// Adapter a TopAY maintainer adds to the reference (e.g. as src/planner/include/planner/moma_traj_opt_hip.h): the
// reference's own MomaTrajOpt surface -- constructor from the shared GridMap, optimizeTraj(init_path, boundary_vel,
// boundary_acc) -> bool, getTraj(), traj_cost, printConstraintsSituations (moma_traj_opt.h:646-674, 943-946) --
// forwarding to the C-ABI of include/topay.h.  It is compiled in the reference's catkin workspace (it needs Eigen and
// the reference's MomaTraj / GridMap); this repository only syntax-checks it against tests/stubs/ (tests/test_cabi.py).
#pragma once
#include <stdexcept>
#include <vector>

#include "topay.h"                   // this repository: include/topay.h
#include "map/grid_map.h"            // GridMap: getOrigin, getResolution, getVoxelNum, min_boundary, max_boundary, getESDFBuffer2d/3d
#include "planner/moma_traj_opt.h"   // MomaTraj, PolyTrajectory<9, 5>, CoefficientMat

class MomaTrajOptHip {
 public:
  typedef std::shared_ptr<MomaTrajOptHip> Ptr;

  explicit MomaTrajOptHip(GridMap::Ptr map, int device = 0) {
    topay_params_t p;
    topay_default_params(&p);        // == optimizer.yaml + MomaParam defaults; override fields from opt_param when they differ
    if (topay_create(&p, device, &ctx_) != TOPAY_OK) throw std::runtime_error(topay_last_error());
    topay_set_latency_mode(ctx_, 1);   // a planning call is a handful of candidates: helper waves in every evaluation (same bits)
    setMap(map);
  }
  ~MomaTrajOptHip() { topay_destroy(ctx_); }
  MomaTrajOptHip(const MomaTrajOptHip&) = delete;
  MomaTrajOptHip& operator=(const MomaTrajOptHip&) = delete;

  // GridMap keeps its distance fields as std::vector<double>, x-major (grid_map.h:213-216, 798-816): passed as they are.
  void setMap(GridMap::Ptr map) {
    topay_map_desc_t d;
    const Eigen::Vector3d origin = map->getOrigin();            // grid_map.h:206
    Eigen::Vector3i num;
    map->getVoxelNum(num);                                      // grid_map.h:179
    for (int k = 0; k < 3; k++) {
      d.origin[k] = origin(k);
      d.dims[k] = num(k);
      d.min_boundary[k] = map->min_boundary(k);                 // public members, grid_map.h:85-86
      d.max_boundary[k] = map->max_boundary(k);
    }
    d.resolution = map->getResolution();                        // grid_map.h:203
    if (topay_set_map(ctx_, 0, &d, map->getESDFBuffer2d().data(), map->getESDFBuffer3d().data()) != TOPAY_OK)
      throw std::runtime_error(topay_last_error());
  }

  // same signature as MomaTrajOpt::optimizeTraj (moma_traj_opt.h:660-662)
  bool optimizeTraj(std::vector<Eigen::VectorXd> init_path, const Eigen::MatrixXd& boundary_vel,
                    const Eigen::MatrixXd& boundary_acc) {
    const int len = (int)init_path.size();
    if (len < 2) return false;
    start_state_ = init_path.front().head(3);                   // MomaTrajOpt::start_state (moma_traj_opt.cpp:146-150)
    std::vector<double> flat((size_t)len * 10);
    for (int i = 0; i < len; i++)
      for (int k = 0; k < 10; k++) flat[(size_t)i * 10 + k] = init_path[i](k);
    // Eigen matrices are column-major 10 x 2: exactly the layout topay_set_init_traj expects
    if (topay_set_init_traj(ctx_, 1, &len, flat.data(), boundary_vel.data(), boundary_acc.data(), nullptr) != TOPAY_OK)
      return false;
    if (topay_optimize(ctx_) != TOPAY_OK) return false;
    int ok = 0;
    topay_get_result(ctx_, 0, &ok, &traj_cost, &n_pieces_, nullptr, nullptr, nullptr);
    return ok != 0;
  }

  // MomaTrajOpt::getTraj(): durations + CoefficientMat<9, 5> per piece, highest order first (minco.hpp:908-921)
  MomaTraj getTraj() const {
    std::vector<double> T((size_t)n_pieces_), c((size_t)n_pieces_ * 54);
    topay_get_result(ctx_, 0, nullptr, nullptr, nullptr, T.data(), c.data(), nullptr);
    std::vector<CoefficientMat<9, 5>> cm((size_t)n_pieces_);
    for (int i = 0; i < n_pieces_; i++)
      for (int d = 0; d < 9; d++)
        for (int k = 0; k < 6; k++) cm[(size_t)i](d, k) = c[((size_t)i * 9 + d) * 6 + k];
    return MomaTraj(PolyTrajectory<9, 5>(T, cm), start_state_);  // as moma_traj_opt.h:943-946
  }

  // the gate of planner.cpp:878-880, on the trajectory optimizeTraj has just produced
  bool printConstraintsSituations(const MomaTraj&) {
    int feasible = 0;
    return topay_check_feasible(ctx_, &feasible) == TOPAY_OK && feasible != 0;
  }

  double traj_cost = 0.0;

 private:
  topay_ctx* ctx_ = nullptr;
  int n_pieces_ = 0;
  Eigen::Vector3d start_state_ = Eigen::Vector3d::Zero();
};
