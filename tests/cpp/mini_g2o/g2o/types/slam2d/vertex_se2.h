// TEST INFRASTRUCTURE ONLY: VertexSE2 of the test host, interface and update rule of
// /root/reference/g2o/types/slam2d/vertex_se2.h:41-59 (translation += (dx, dy), angle = normalize(angle + dtheta)).
#ifndef G2O_MINI_VERTEX_SE2_H
#define G2O_MINI_VERTEX_SE2_H
#include "se2.h"
namespace g2o {
class VertexSE2 : public BaseVertex<3, SE2> {
 public:
  virtual bool write(std::ostream& os) const;           // vertex_se2.cpp:50-55 (out of line: libg2o_mini_types_slam2d.so)
  virtual void oplusImpl(const double* d) {   // additive on all three coordinates, the angle wrapped into [-pi, pi)
    const SE2& T = _estimate;
    _estimate = SE2(T.translation()[0] + d[0], T.translation()[1] + d[1], normalize_theta(T.rotation().angle() + d[2]));
  }
};
}  // namespace g2o
#endif
