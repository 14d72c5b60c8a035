// TEST INFRASTRUCTURE ONLY: EdgeSE3 of the test host, interface of /root/reference/g2o/types/slam3d/edge_se3.h and the error of
// edge_se3.cpp:48-53 (toVectorMQT(Z^-1 Xi^-1 Xj)) with ANALYTIC Jacobians (edge_se3.cpp:64-75 -> isometry3d_gradients.h:86-192),
// so that the generic path of the plugin (this host's Jacobians uploaded) and its device fast path (pg_se3_linearize_kernel)
// agree to rounding (round 3 had central differences here and a 1e-2 tolerance).
#ifndef G2O_MINI_EDGE_SE3_H
#define G2O_MINI_EDGE_SE3_H
#include "vertex_se3.h"
namespace g2o {
class EdgeSE3 : public BaseBinaryEdge<6, Eigen::Isometry3d, VertexSE3, VertexSE3> {
 public:
  virtual bool write(std::ostream& os) const;           // edge_se3.cpp:77-87 (out of line: libg2o_mini_types_slam3d.so)
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
  // d(qx, qy, qz) / d vec(R) (3 x 9, vec column-major, D row-major): the partial derivatives of the four-case conversion
  // R -> unit quaternion with w >= 0 (what /root/reference/g2o/types/slam3d/dquat2mat.cpp tabulates)
  static void dq_dR(const Eigen::Matrix3d& R, double* D) {
    for (int i = 0; i < 27; ++i) D[i] = 0;
    const double tr = R(0, 0) + R(1, 1) + R(2, 2);
    double qw;
    if (tr > 0) {
      const double w = 0.5 * std::sqrt(tr + 1.0);
      qw = w;
      const double num[3] = {R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1)};
      const int pa[3] = {2, 0, 1}, pb[3] = {1, 2, 0};
      for (int c = 0; c < 3; ++c) {
        const double dd = -num[c] / (32.0 * w * w * w);
        D[c * 9 + 0] = D[c * 9 + 4] = D[c * 9 + 8] = dd;
        D[c * 9 + pa[c] + 3 * pb[c]] = 0.25 / w;
        D[c * 9 + pb[c] + 3 * pa[c]] = -0.25 / w;
      }
    } else {
      int i;
      if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) i = 0;
      else if (R(1, 1) > R(2, 2)) i = 1;
      else i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      const double s = 0.5 * std::sqrt(1.0 + R(i, i) - R(j, j) - R(k, k));
      qw = (R(k, j) - R(j, k)) / (4.0 * s);
      D[i * 9 + i + 3 * i] = 1.0 / (8.0 * s);
      D[i * 9 + j + 3 * j] = -1.0 / (8.0 * s);
      D[i * 9 + k + 3 * k] = -1.0 / (8.0 * s);
      const int other[2] = {j, k};
      for (int o = 0; o < 2; ++o) {
        const int c = other[o];
        const double num = R(c, i) + R(i, c), dd = num / (32.0 * s * s * s);
        D[c * 9 + c + 3 * i] += 0.25 / s;
        D[c * 9 + i + 3 * c] += 0.25 / s;
        D[c * 9 + i + 3 * i] += -dd;
        D[c * 9 + j + 3 * j] += dd;
        D[c * 9 + k + 3 * k] += dd;
      }
    }
    if (qw <= 0)
      for (int i = 0; i < 27; ++i) D[i] = -D[i];
  }
  // ANALYTIC Jacobians of e = toVectorMQT(Z^-1 Xi^-1 Xj) with respect to the minimal updates X <- X fromVectorMQT(d)
  // (edge_se3.cpp:64-75 -> isometry3d_gradients.h:86-192 with the sensor offsets at identity), written here from the chain
  // rule: translation part E.t = Ra (tbc), rotation part through dq_dR of E.R = Ra Rbc.
  virtual void linearizeOplus() {
    const VertexSE3* from = static_cast<const VertexSE3*>(_vertices[0]);
    const VertexSE3* to = static_cast<const VertexSE3*>(_vertices[1]);
    const Eigen::Isometry3d A = _inverseMeasurement, B = from->estimate().inverse() * to->estimate(), E = A * B;
    const Eigen::Matrix3d &Ra = A.linear(), &Rab = E.linear(), &Rbc = B.linear();
    const Vector3d& tbc = B.translation();
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 6; ++c) _jacobianOplusXi(r, c) = _jacobianOplusXj(r, c) = 0.0;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) {
        _jacobianOplusXi(r, c) = -Ra(r, c);
        _jacobianOplusXj(r, c) = Rab(r, c);
      }
    {  // d t_e / d q_i = Ra skew(2 tbc)
      const double x = 2 * tbc[0], y = 2 * tbc[1], z = 2 * tbc[2];
      const double S[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) {
          double s = 0;
          for (int m = 0; m < 3; ++m) s += Ra(r, m) * S[m][c];
          _jacobianOplusXi(r, 3 + c) = s;
        }
    }
    double D[27];
    dq_dR(Rab, D);
    {  // d q_e / d q_i: dR_e = Ra (skew(2 e_a)^T Rbc)
      const double r11 = 2 * Rbc(0, 0), r12 = 2 * Rbc(0, 1), r13 = 2 * Rbc(0, 2), r21 = 2 * Rbc(1, 0), r22 = 2 * Rbc(1, 1), r23 = 2 * Rbc(1, 2),
                   r31 = 2 * Rbc(2, 0), r32 = 2 * Rbc(2, 1), r33 = 2 * Rbc(2, 2);
      const double S[3][9] = {{0, 0, 0, r31, r32, r33, -r21, -r22, -r23}, {-r31, -r32, -r33, 0, 0, 0, r11, r12, r13},
                              {r21, r22, r23, -r11, -r12, -r13, 0, 0, 0}};   // row-wise 3 x 3
      for (int a = 0; a < 3; ++a) {
        double M[9];
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 3; ++r) {
            double s = 0;
            for (int m = 0; m < 3; ++m) s += Ra(r, m) * S[a][m * 3 + c];
            M[r + 3 * c] = s;
          }
        for (int r = 0; r < 3; ++r) {
          double s = 0;
          for (int m = 0; m < 9; ++m) s += D[r * 9 + m] * M[m];
          _jacobianOplusXi(3 + r, 3 + a) = s;
        }
      }
    }
    {  // d q_e / d q_j: dR_e = Rab skew(2 e_a)
      const double S[3][9] = {{0, 0, 0, 0, 0, -2, 0, 2, 0}, {0, 0, 2, 0, 0, 0, -2, 0, 0}, {0, -2, 0, 2, 0, 0, 0, 0, 0}};
      for (int a = 0; a < 3; ++a) {
        double M[9];
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 3; ++r) {
            double s = 0;
            for (int m = 0; m < 3; ++m) s += Rab(r, m) * S[a][m * 3 + c];
            M[r + 3 * c] = s;
          }
        for (int r = 0; r < 3; ++r) {
          double s = 0;
          for (int m = 0; m < 9; ++m) s += D[r * 9 + m] * M[m];
          _jacobianOplusXj(3 + r, 3 + a) = s;
        }
      }
    }
  }
 protected:
  Eigen::Isometry3d _inverseMeasurement;
};
}  // namespace g2o
#endif
