// TEST INFRASTRUCTURE ONLY (see ../../../g2o_mini.h): the bundle-adjustment types of the test host with the public
// interface of /root/reference/g2o/types/sba/types_six_dof_expmap.h:48-153 and types_sba.h:92-118 -- CameraParameters,
// VertexSE3Expmap (world -> camera, update exp(delta) * T with delta = (omega, upsilon)), VertexSBAPointXYZ,
// EdgeProjectXYZ2UV (vertex 0 = point, vertex 1 = pose; error = measurement - projection; analytic Jacobians as
// types_six_dof_expmap.cpp:288-326).
#ifndef G2O_MINI_TYPES_SIX_DOF_EXPMAP_H
#define G2O_MINI_TYPES_SIX_DOF_EXPMAP_H
#include "../slam3d/se3quat.h"
namespace g2o {
class CameraParameters {                                // types_six_dof_expmap.h:48-85
 public:
  CameraParameters() : focal_length(1.), baseline(0.5) {}
  CameraParameters(double f, const Vector2d& pp, double b) : focal_length(f), principle_point(pp), baseline(b) {}
  Vector2d cam_map(const Vector3d& trans_xyz) const {
    Vector2d r;
    r[0] = trans_xyz[0] / trans_xyz[2] * focal_length + principle_point[0];
    r[1] = trans_xyz[1] / trans_xyz[2] * focal_length + principle_point[1];
    return r;
  }
  double focal_length;
  Vector2d principle_point;
  double baseline;
};
class VertexSE3Expmap : public BaseVertex<6, SE3Quat> { // :91-109
 public:
  virtual bool write(std::ostream& os) const;           // types_six_dof_expmap.cpp:88-103 (out of line: libg2o_mini_types_sba.so)
  virtual void oplusImpl(const double* update_) {
    Vector6d u;
    for (int i = 0; i < 6; ++i) u[i] = update_[i];
    setEstimate(SE3Quat::exp(u) * estimate());
  }
};
class VertexSBAPointXYZ : public BaseVertex<3, Vector3d> {   // types_sba.h:92-118
 public:
  virtual bool write(std::ostream& os) const;
  virtual void oplusImpl(const double* update) { for (int i = 0; i < 3; ++i) _estimate[i] += update[i]; }
};
class EdgeProjectXYZ2UV : public BaseBinaryEdge<2, Vector2d, VertexSBAPointXYZ, VertexSE3Expmap> {   // :133-153
 public:
  EdgeProjectXYZ2UV() : _cam(0) {}
  virtual bool write(std::ostream& os) const;
  virtual void computeError() {
    const VertexSE3Expmap* v1 = static_cast<const VertexSE3Expmap*>(_vertices[1]);
    const VertexSBAPointXYZ* v2 = static_cast<const VertexSBAPointXYZ*>(_vertices[0]);
    const Vector2d proj = _cam->cam_map(v1->estimate().map(v2->estimate()));
    _error[0] = _measurement[0] - proj[0];
    _error[1] = _measurement[1] - proj[1];
  }
  using BaseBinaryEdge<2, Vector2d, VertexSBAPointXYZ, VertexSE3Expmap>::linearizeOplus;
  virtual void linearizeOplus() {
    const VertexSE3Expmap* vj = static_cast<const VertexSE3Expmap*>(_vertices[1]);
    const VertexSBAPointXYZ* vi = static_cast<const VertexSBAPointXYZ*>(_vertices[0]);
    const SE3Quat& T = vj->estimate();
    const Vector3d xyz_trans = T.map(vi->estimate());
    const double x = xyz_trans[0], y = xyz_trans[1], z = xyz_trans[2], z_2 = z * z, f = _cam->focal_length;
    const double tmp[2][3] = {{f, 0, -x / z * f}, {0, f, -y / z * f}};
    const Eigen::Matrix3d& R = T.rotationMatrix();
    for (int r = 0; r < 2; ++r)
      for (int c = 0; c < 3; ++c) _jacobianOplusXi(r, c) = -1. / z * (tmp[r][0] * R(0, c) + tmp[r][1] * R(1, c) + tmp[r][2] * R(2, c));
    _jacobianOplusXj(0, 0) = x * y / z_2 * f;
    _jacobianOplusXj(0, 1) = -(1 + (x * x / z_2)) * f;
    _jacobianOplusXj(0, 2) = y / z * f;
    _jacobianOplusXj(0, 3) = -1. / z * f;
    _jacobianOplusXj(0, 4) = 0;
    _jacobianOplusXj(0, 5) = x / z_2 * f;
    _jacobianOplusXj(1, 0) = (1 + y * y / z_2) * f;
    _jacobianOplusXj(1, 1) = -x * y / z_2 * f;
    _jacobianOplusXj(1, 2) = -x / z * f;
    _jacobianOplusXj(1, 3) = 0;
    _jacobianOplusXj(1, 4) = -1. / z * f;
    _jacobianOplusXj(1, 5) = y / z_2 * f;
  }
  CameraParameters* _cam;
};
}  // namespace g2o
#endif
