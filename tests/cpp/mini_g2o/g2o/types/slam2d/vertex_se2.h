// TEST INFRASTRUCTURE ONLY: VertexSE2 of the test host, interface and update rule of
// /root/reference/g2o/types/slam2d/vertex_se2.h:41-59 (translation += (dx, dy), angle = normalize(angle + dtheta)).
#ifndef G2O_MINI_VERTEX_SE2_H
#define G2O_MINI_VERTEX_SE2_H
#include "se2.h"
namespace g2o {
class VertexSE2 : public BaseVertex<3, SE2> {
 public:
  virtual void oplusImpl(const double* update) {
    Vector2d t = _estimate.translation();
    t[0] += update[0];
    t[1] += update[1];
    const double angle = normalize_theta(_estimate.rotation().angle() + update[2]);
    _estimate.setTranslation(t);
    _estimate.setRotation(Rotation2Dd(angle));
  }
};
}  // namespace g2o
#endif
