// TEST INFRASTRUCTURE ONLY: libg2o_mini_types_slam3d.so -- what libg2o_types_slam3d.so is to g2o (see g2o_mini_types_sba.cpp).
#include "g2o/types/slam3d/edge_se3.h"
namespace g2o {
bool VertexSE3::write(std::ostream& os) const {
  const Vector6d v = internal::toVectorMQT(estimate());
  for (int i = 0; i < 6; ++i) os << v[i] << " ";
  return os.good();
}
bool EdgeSE3::write(std::ostream& os) const {
  const Vector6d v = internal::toVectorMQT(measurement());
  for (int i = 0; i < 6; ++i) os << v[i] << " ";
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) os << " " << _information(i, j);
  return os.good();
}
}  // namespace g2o
