// syntax-check stand-in for planner/moma_traj_opt.h:26-45 and utils/minco.hpp:27-40, 268-290, see tests/stubs/README.md
#pragma once
#include <vector>
#include <Eigen/Eigen>
template <int Dim, int Order> using CoefficientMat = Eigen::Matrix<double, Dim, Order + 1>;
template <int Dim, int Order> struct PolyTrajectory {
  PolyTrajectory() = default;
  PolyTrajectory(const std::vector<double>& durs, const std::vector<CoefficientMat<Dim, Order>>& cMats);
};
struct MomaTraj {
  MomaTraj() {}
  MomaTraj(PolyTrajectory<9, 5> ploy_traj_, const Eigen::Vector3d& start_state_);
};
