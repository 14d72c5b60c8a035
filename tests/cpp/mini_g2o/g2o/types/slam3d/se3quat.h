// TEST INFRASTRUCTURE ONLY (see ../../../g2o_mini.h): the members of g2o::SE3Quat the test host and the adapter's fast
// path use (/root/reference/g2o/types/slam3d/se3quat.h:41-300), over a rotation matrix + translation.
#ifndef G2O_MINI_SE3QUAT_H
#define G2O_MINI_SE3QUAT_H
#include "../../../g2o_mini.h"
namespace g2o {
class SE3Quat {
 public:
  SE3Quat() { _R.setIdentity(); }
  SE3Quat(const Eigen::Matrix3d& R, const Vector3d& t) : _R(R), _t(t) {}
  const Vector3d& translation() const { return _t; }                        // :96
  Eigen::Quaterniond rotation() const { return Eigen::Quaterniond(_R); }    // :100 (the reference stores the quaternion)
  const Eigen::Matrix3d& rotationMatrix() const { return _R; }
  Vector3d map(const Vector3d& xyz) const {                                 // :217
    Vector3d r;
    for (int i = 0; i < 3; ++i) r[i] = _R(i, 0) * xyz[0] + _R(i, 1) * xyz[1] + _R(i, 2) * xyz[2] + _t[i];
    return r;
  }
  SE3Quat operator*(const SE3Quat& o) const {                               // :108
    SE3Quat r;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) r._R(i, j) = _R(i, 0) * o._R(0, j) + _R(i, 1) * o._R(1, j) + _R(i, 2) * o._R(2, j);
      r._t[i] = _R(i, 0) * o._t[0] + _R(i, 1) * o._t[1] + _R(i, 2) * o._t[2] + _t[i];
    }
    return r;
  }
  // exponential map, update = (omega, upsilon) (:223-257)
  static SE3Quat exp(const Vector6d& u) {
    const double theta = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    Eigen::Matrix3d Om, Om2, R, V;
    Om(0, 1) = -u[2]; Om(0, 2) = u[1]; Om(1, 0) = u[2]; Om(1, 2) = -u[0]; Om(2, 0) = -u[1]; Om(2, 1) = u[0];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Om2(i, j) = Om(i, 0) * Om(0, j) + Om(i, 1) * Om(1, j) + Om(i, 2) * Om(2, j);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        const double I = i == j ? 1.0 : 0.0;
        if (theta < 0.00001) {
          R(i, j) = I + Om(i, j) + Om2(i, j);
          V(i, j) = R(i, j);
        } else {
          R(i, j) = I + std::sin(theta) / theta * Om(i, j) + (1 - std::cos(theta)) / (theta * theta) * Om2(i, j);
          V(i, j) = I + (1 - std::cos(theta)) / (theta * theta) * Om(i, j) + (theta - std::sin(theta)) / (theta * theta * theta) * Om2(i, j);
        }
      }
    SE3Quat r;
    r._R = R;
    for (int i = 0; i < 3; ++i) r._t[i] = V(i, 0) * u[3] + V(i, 1) * u[4] + V(i, 2) * u[5];
    return r;
  }
 private:
  Eigen::Matrix3d _R;
  Vector3d _t;
};
}  // namespace g2o
#endif
