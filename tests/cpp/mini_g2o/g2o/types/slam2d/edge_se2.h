// TEST INFRASTRUCTURE ONLY: EdgeSE2 of the test host, interface of /root/reference/g2o/types/slam2d/edge_se2.h:41-57 and the
// analytic Jacobians of edge_se2.cpp:76-99: error = (Z^-1 (Xi^-1 Xj)).toVector(), Jacobians rotated by Z^-1.
#ifndef G2O_MINI_EDGE_SE2_H
#define G2O_MINI_EDGE_SE2_H
#include "vertex_se2.h"
namespace g2o {
class EdgeSE2 : public BaseBinaryEdge<3, SE2, VertexSE2, VertexSE2> {
 public:
  virtual bool write(std::ostream& os) const;           // edge_se2.cpp:54-62 (out of line: libg2o_mini_types_slam2d.so)
  virtual void computeError() {
    const VertexSE2* v1 = static_cast<const VertexSE2*>(_vertices[0]);
    const VertexSE2* v2 = static_cast<const VertexSE2*>(_vertices[1]);
    const SE2 delta = _inverseMeasurement * (v1->estimate().inverse() * v2->estimate());
    const Vector3d e = delta.toVector();
    for (int i = 0; i < 3; ++i) _error[i] = e[i];
  }
  void setMeasurement(const SE2& m) {
    _measurement = m;
    _inverseMeasurement = m.inverse();
  }
  using BaseBinaryEdge<3, SE2, VertexSE2, VertexSE2>::linearizeOplus;
  virtual void linearizeOplus() {
    const VertexSE2* vi = static_cast<const VertexSE2*>(_vertices[0]);
    const VertexSE2* vj = static_cast<const VertexSE2*>(_vertices[1]);
    const double ti = vi->estimate().rotation().angle(), si = std::sin(ti), ci = std::cos(ti);
    const double dx = vj->estimate().translation()[0] - vi->estimate().translation()[0];
    const double dy = vj->estimate().translation()[1] - vi->estimate().translation()[1];
    double Ji[3][3] = {{-ci, -si, -si * dx + ci * dy}, {si, -ci, -ci * dx - si * dy}, {0, 0, -1}};
    double Jj[3][3] = {{ci, si, 0}, {-si, ci, 0}, {0, 0, 1}};
    const double tz = _inverseMeasurement.rotation().angle(), cz = std::cos(tz), sz = std::sin(tz);
    const double Z[3][3] = {{cz, -sz, 0}, {sz, cz, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double a = 0, b = 0;
        for (int k = 0; k < 3; ++k) {
          a += Z[r][k] * Ji[k][c];
          b += Z[r][k] * Jj[k][c];
        }
        _jacobianOplusXi(r, c) = a;
        _jacobianOplusXj(r, c) = b;
      }
  }
 protected:
  SE2 _inverseMeasurement;
};
}  // namespace g2o
#endif
