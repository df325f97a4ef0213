// syntax-check stand-in for src/map/include/map/grid_map.h:79-217, see tests/stubs/README.md
#pragma once
#include <memory>
#include <vector>
#include <Eigen/Eigen>
class GridMap {
 public:
  typedef std::shared_ptr<GridMap> Ptr;
  Eigen::Vector3d min_boundary, max_boundary;
  Eigen::Vector3d getOrigin();
  double getResolution();
  void getVoxelNum(Eigen::Vector3i& voxel_num);
  const std::vector<double>& getESDFBuffer2d() const;
  const std::vector<double>& getESDFBuffer3d() const;
};
