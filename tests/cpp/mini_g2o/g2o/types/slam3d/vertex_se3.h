// TEST INFRASTRUCTURE ONLY (see ../../../g2o_mini.h): VertexSE3 of the test host over a small rigid-motion class with the
// members of Eigen::Isometry3d that g2o's slam3d types and the adapter use (linear(), translation(), operator*, inverse()).
// Interface and update rule of /root/reference/g2o/types/slam3d/vertex_se3.h:107-116 (estimate <- estimate * increment,
// increment from (x, y, z, qx, qy, qz) with qw = sqrt(1 - |q|^2): isometry3d_mappings.cpp:84-91,117-122).
#ifndef G2O_MINI_VERTEX_SE3_H
#define G2O_MINI_VERTEX_SE3_H
#include <cmath>
#include "../../../g2o_mini.h"
namespace Eigen {
class Isometry3d {
 public:
  Isometry3d() { _R.setIdentity(); _t[0] = _t[1] = _t[2] = 0.; }
  Isometry3d(const Matrix3d& R, const Vector3d& t) : _R(R), _t(t) {}
  const Matrix3d& linear() const { return _R; }
  Matrix3d& linear() { return _R; }
  const Vector3d& translation() const { return _t; }
  Vector3d& translation() { return _t; }
  Isometry3d operator*(const Isometry3d& o) const {
    Isometry3d r;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += _R(i, k) * o._R(k, j);
        r._R(i, j) = s;
      }
      double s = _t[i];
      for (int k = 0; k < 3; ++k) s += _R(i, k) * o._t[k];
      r._t[i] = s;
    }
    return r;
  }
  Isometry3d inverse() const {
    Isometry3d r;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) r._R(i, j) = _R(j, i);
      double s = 0;
      for (int k = 0; k < 3; ++k) s -= _R(k, i) * _t[k];
      r._t[i] = s;
    }
    return r;
  }
 private:
  Matrix3d _R;
  Vector3d _t;
};
}  // namespace Eigen
namespace g2o {
namespace internal {
// (x, y, z) of the unit quaternion of the rotation, w >= 0 (isometry3d_mappings.cpp:38-44,77-82,94-99)
inline Vector6d toVectorMQT(const Eigen::Isometry3d& T) {
  Eigen::Quaterniond q(T.linear());
  double n = std::sqrt(q.w() * q.w() + q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
  if (q.w() < 0) n = -n;
  Vector6d v;
  for (int i = 0; i < 3; ++i) v[i] = T.translation()[i];
  v[3] = q.x() / n; v[4] = q.y() / n; v[5] = q.z() / n;
  return v;
}
inline Eigen::Isometry3d fromVectorMQT(const Vector6d& v) {
  Eigen::Isometry3d T;
  const double w2 = 1. - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
  if (w2 >= 0) T.linear() = Eigen::Quaterniond(std::sqrt(w2), v[3], v[4], v[5]).toRotationMatrix();
  for (int i = 0; i < 3; ++i) T.translation()[i] = v[i];
  return T;
}
}  // namespace internal
class VertexSE3 : public BaseVertex<6, Eigen::Isometry3d> {
 public:
  virtual bool write(std::ostream& os) const;           // vertex_se3.cpp:62-68 (out of line: libg2o_mini_types_slam3d.so)
  virtual void oplusImpl(const double* update) {
    Vector6d v;
    for (int i = 0; i < 6; ++i) v[i] = update[i];
    _estimate = _estimate * internal::fromVectorMQT(v);
  }
};
}  // namespace g2o
#endif
