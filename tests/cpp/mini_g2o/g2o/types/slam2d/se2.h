// TEST INFRASTRUCTURE ONLY (see ../../../g2o_mini.h): planar rigid motion with the public interface of
// /root/reference/g2o/types/slam2d/se2.h -- translation(), rotation().angle(), operator*, inverse(), toVector() -- written
// for the test host (an angle and two coordinates; no Eigen).
#ifndef G2O_MINI_SE2_H
#define G2O_MINI_SE2_H
#include <cmath>
#include "../../../g2o_mini.h"
namespace g2o {
inline double normalize_theta(double theta) {            // g2o/stuff/misc.h: into [-pi, pi)
  if (theta >= -M_PI && theta < M_PI) return theta;
  const double multiplier = std::floor(theta / (2 * M_PI));
  theta = theta - multiplier * 2 * M_PI;
  if (theta >= M_PI) theta -= 2 * M_PI;
  if (theta < -M_PI) theta += 2 * M_PI;
  return theta;
}
class Rotation2Dd {
 public:
  explicit Rotation2Dd(double a = 0.) : _a(a) {}
  double angle() const { return _a; }
  double& angle() { return _a; }
 private:
  double _a;
};
class SE2 {
 public:
  SE2() : _R(0.) { _t[0] = _t[1] = 0.; }
  SE2(double x, double y, double theta) : _R(theta) { _t[0] = x; _t[1] = y; }
  const Vector2d& translation() const { return _t; }
  void setTranslation(const Vector2d& t) { _t = t; }
  const Rotation2Dd& rotation() const { return _R; }
  void setRotation(const Rotation2Dd& R) { _R = R; }
  SE2 operator*(const SE2& o) const {                    // (R, t)(R', t') = (R R', t + R t')
    const double c = std::cos(_R.angle()), s = std::sin(_R.angle());
    return SE2(_t[0] + c * o._t[0] - s * o._t[1], _t[1] + s * o._t[0] + c * o._t[1], normalize_theta(_R.angle() + o._R.angle()));
  }
  SE2 inverse() const {                                  // (R', -R' t)
    const double c = std::cos(_R.angle()), s = std::sin(_R.angle());
    return SE2(-(c * _t[0] + s * _t[1]), -(-s * _t[0] + c * _t[1]), normalize_theta(-_R.angle()));
  }
  Vector3d toVector() const {
    Vector3d v;
    v[0] = _t[0]; v[1] = _t[1]; v[2] = _R.angle();
    return v;
  }
 private:
  Rotation2Dd _R;
  Vector2d _t;
};
}  // namespace g2o
#endif
