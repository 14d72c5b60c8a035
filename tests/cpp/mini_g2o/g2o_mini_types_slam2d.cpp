// TEST INFRASTRUCTURE ONLY: libg2o_mini_types_slam2d.so -- what libg2o_types_slam2d.so is to g2o (see g2o_mini_types_sba.cpp).
#include "g2o/types/slam2d/edge_se2.h"
namespace g2o {
bool VertexSE2::write(std::ostream& os) const {
  const Vector3d p = estimate().toVector();
  os << p[0] << " " << p[1] << " " << p[2];
  return os.good();
}
bool EdgeSE2::write(std::ostream& os) const {
  const Vector3d p = measurement().toVector();
  os << p[0] << " " << p[1] << " " << p[2];
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j) os << " " << _information(i, j);
  return os.good();
}
}  // namespace g2o
