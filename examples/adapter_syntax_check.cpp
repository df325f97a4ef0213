// Translation unit for `g++ -fsyntax-only`: instantiates every member of the reference-side adapter.
#include "moma_traj_opt_hip.h"
bool adapter_syntax_check(GridMap::Ptr map, std::vector<Eigen::VectorXd> path, const Eigen::MatrixXd& bv, const Eigen::MatrixXd& ba) {
  MomaTrajOptHip opt(map, 0);
  const bool ok = opt.optimizeTraj(path, bv, ba) && opt.printConstraintsSituations(opt.getTraj());
  return ok && opt.traj_cost >= 0.0;
}
