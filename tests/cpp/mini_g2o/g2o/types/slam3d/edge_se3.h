// TEST INFRASTRUCTURE ONLY: EdgeSE3 of the test host, interface of /root/reference/g2o/types/slam3d/edge_se3.h and the error of
// edge_se3.cpp:48-53 (toVectorMQT(Z^-1 Xi^-1 Xj)).  Its Jacobians are NUMERIC (central differences over oplus, what
// BaseBinaryEdge does when a type brings none: base_binary_edge.hpp:130-205) -- the reference's analytic ones
// (isometry3d_gradients.h) are restated on the device (pg_se3_linearize_kernel) and in the oracle, and this host is what
// those are compared with through the plugin.
#ifndef G2O_MINI_EDGE_SE3_H
#define G2O_MINI_EDGE_SE3_H
#include "vertex_se3.h"
namespace g2o {
class EdgeSE3 : public BaseBinaryEdge<6, Eigen::Isometry3d, VertexSE3, VertexSE3> {
 public:
  virtual void computeError() {
    const VertexSE3* from = static_cast<const VertexSE3*>(_vertices[0]);
    const VertexSE3* to = static_cast<const VertexSE3*>(_vertices[1]);
    const Vector6d e = internal::toVectorMQT(_inverseMeasurement * from->estimate().inverse() * to->estimate());
    for (int i = 0; i < 6; ++i) _error[i] = e[i];
  }
  void setMeasurement(const Eigen::Isometry3d& m) {
    _measurement = m;
    _inverseMeasurement = m.inverse();
  }
  using BaseBinaryEdge<6, Eigen::Isometry3d, VertexSE3, VertexSE3>::linearizeOplus;
  virtual void linearizeOplus() {
    const double delta = 1e-9, scalar = 1. / (2 * delta);
    ErrorVector keep = _error;
    for (int side = 0; side < 2; ++side) {
      VertexSE3* v = static_cast<VertexSE3*>(_vertices[side]);
      for (int d = 0; d < 6; ++d) {
        double add[6] = {0, 0, 0, 0, 0, 0};
        double e1[6], e2[6];
        v->push();
        add[d] = delta;
        v->oplus(add);
        computeError();
        for (int i = 0; i < 6; ++i) e1[i] = _error[i];
        v->pop();
        v->push();
        add[d] = -delta;
        v->oplus(add);
        computeError();
        for (int i = 0; i < 6; ++i) e2[i] = _error[i];
        v->pop();
        for (int i = 0; i < 6; ++i) (side ? _jacobianOplusXj : _jacobianOplusXi)(i, d) = scalar * (e1[i] - e2[i]);
      }
    }
    _error = keep;
  }
 protected:
  Eigen::Isometry3d _inverseMeasurement;
};
}  // namespace g2o
#endif
