/* ===========================================================================
 * TEST INFRASTRUCTURE ONLY (see g2o_oracle.c header).  CPU restatement of the
 * input producers -- error vectors, analytic Jacobians and oplus -- for the edge
 * types BASELINE.json's configs use.  References (relative to /root/reference):
 *   EdgeSE2::computeError            g2o/types/slam2d/edge_se2.h:46-52
 *   EdgeSE2::linearizeOplus          g2o/types/slam2d/edge_se2.cpp:76-99
 *   VertexSE2::oplusImpl             g2o/types/slam2d/vertex_se2.h:51-58
 *   SE2 compose/inverse              g2o/types/slam2d/se2.h:59-89
 *   normalize_theta                  g2o/stuff/misc.h:94-107
 *   EdgeProjectXYZ2UV::computeError  g2o/types/sba/types_six_dof_expmap.h:139-147
 *   EdgeProjectXYZ2UV::linearizeOplus g2o/types/sba/types_six_dof_expmap.cpp:288-326
 *   CameraParameters::cam_map        g2o/types/sba/types_six_dof_expmap.cpp:69-75
 *   VertexSE3Expmap::oplusImpl       g2o/types/sba/types_six_dof_expmap.h:101-104
 *   SE3Quat::exp                     g2o/types/slam3d/se3quat.h:223-257
 *   VertexSBAPointXYZ::oplusImpl     g2o/types/sba/types_sba.h:151-155
 * Poses of the BA problem are kept as (R column-major 3x3, t) world->camera; the
 * reference keeps a unit quaternion -- same group element, roundoff-level difference.
 * ======================================================================== */
#define _USE_MATH_DEFINES
#define _DEFAULT_SOURCE
#include <math.h>
#include <string.h>

static double normalize_theta(double theta) {
  if (theta >= -M_PI && theta < M_PI) return theta;
  double multiplier = floor(theta / (2 * M_PI));
  theta = theta - multiplier * 2 * M_PI;
  if (theta >= M_PI) theta -= 2 * M_PI;
  if (theta < -M_PI) theta += 2 * M_PI;
  return theta;
}

/* SE2 as (x, y, theta) */
static void se2_inverse(const double* a, double* r) {
  double th = normalize_theta(-a[2]); double c = cos(th), s = sin(th);
  r[0] = c * (-a[0]) - s * (-a[1]); r[1] = s * (-a[0]) + c * (-a[1]); r[2] = th;
}
static void se2_mul(const double* a, const double* b, double* r) {
  double c = cos(a[2]), s = sin(a[2]);
  double x = a[0] + c * b[0] - s * b[1], y = a[1] + s * b[0] + c * b[1];
  r[0] = x; r[1] = y; r[2] = normalize_theta(a[2] + b[2]);
}

/* poses: [nv][3]; vi, vj: vertex ids; meas: [n][3].  Outputs column-major 3x3 Jacobians. */
void orc_se2_edges(int n, const double* poses, const int* vi, const int* vj, const double* meas,
                   double* J0, double* J1, double* err) {
  for (int k = 0; k < n; ++k) {
    const double* xi = poses + 3 * (size_t)vi[k]; const double* xj = poses + 3 * (size_t)vj[k];
    double invm[3], invi[3], t1[3], delta[3];
    se2_inverse(meas + 3 * (size_t)k, invm);
    se2_inverse(xi, invi);
    se2_mul(invi, xj, t1);
    se2_mul(invm, t1, delta);
    if (err) { err[3 * k] = delta[0]; err[3 * k + 1] = delta[1]; err[3 * k + 2] = delta[2]; }
    if (!J0) continue;
    double thetai = xi[2]; double dtx = xj[0] - xi[0], dty = xj[1] - xi[1];
    double si = sin(thetai), ci = cos(thetai);
    double A[9], B[9];   /* row-major scratch A[r*3+c] */
    A[0] = -ci; A[1] = -si; A[2] = -si * dtx + ci * dty;
    A[3] = si;  A[4] = -ci; A[5] = -ci * dtx - si * dty;
    A[6] = 0;   A[7] = 0;   A[8] = -1;
    B[0] = ci;  B[1] = si;  B[2] = 0;
    B[3] = -si; B[4] = ci;  B[5] = 0;
    B[6] = 0;   B[7] = 0;   B[8] = 1;
    double cz = cos(invm[2]), sz = sin(invm[2]);
    double Z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double* o0 = J0 + 9 * (size_t)k; double* o1 = J1 + 9 * (size_t)k;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
      double a = 0, b = 0;
      for (int m = 0; m < 3; ++m) { a += Z[r * 3 + m] * A[m * 3 + c]; b += Z[r * 3 + m] * B[m * 3 + c]; }
      o0[r + 3 * c] = a; o1[r + 3 * c] = b;
    }
  }
}
/* hidx[v] = hessian index or -1; x: solution vector (pose part), block size 3 */
void orc_se2_oplus(int nv, double* poses, const int* hidx, const double* x) {
  for (int v = 0; v < nv; ++v) {
    if (hidx[v] < 0) continue;
    const double* u = x + 3 * (size_t)hidx[v]; double* p = poses + 3 * (size_t)v;
    p[0] += u[0]; p[1] += u[1]; p[2] = normalize_theta(p[2] + u[2]);
  }
}

/* ---- bundle adjustment: EdgeProjectXYZ2UV.  cams: [nc][12] = R (col-major) | t.  pts: [np][3]. */
void orc_ba_edges(int n, const double* cams, const double* pts, const int* cam_idx, const int* pt_idx,
                  const double* meas, double f, double cx, double cy,
                  double* Jpt /* [n][2x3] */, double* Jcam /* [n][2x6] */, double* err /* [n][2] */) {
  for (int k = 0; k < n; ++k) {
    const double* T = cams + 12 * (size_t)cam_idx[k]; const double* X = pts + 3 * (size_t)pt_idx[k];
    double x = T[0] * X[0] + T[3] * X[1] + T[6] * X[2] + T[9];
    double y = T[1] * X[0] + T[4] * X[1] + T[7] * X[2] + T[10];
    double z = T[2] * X[0] + T[5] * X[1] + T[8] * X[2] + T[11];
    if (err) { err[2 * k] = meas[2 * k] - (x / z * f + cx); err[2 * k + 1] = meas[2 * k + 1] - (y / z * f + cy); }
    if (!Jpt) continue;
    double z_2 = z * z;
    double tmp[6] = {f, 0, -x / z * f, 0, f, -y / z * f};   /* row-major 2x3 */
    double* A = Jpt + 6 * (size_t)k;
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
      double t = 0; for (int m = 0; m < 3; ++m) t += tmp[r * 3 + m] * T[m + 3 * c];
      A[r + 2 * c] = -1. / z * t;
    }
    double* B = Jcam + 12 * (size_t)k;
    B[0 + 2 * 0] = x * y / z_2 * f;         B[0 + 2 * 1] = -(1 + (x * x / z_2)) * f; B[0 + 2 * 2] = y / z * f;
    B[0 + 2 * 3] = -1. / z * f;             B[0 + 2 * 4] = 0;                        B[0 + 2 * 5] = x / z_2 * f;
    B[1 + 2 * 0] = (1 + y * y / z_2) * f;   B[1 + 2 * 1] = -x * y / z_2 * f;         B[1 + 2 * 2] = -x / z * f;
    B[1 + 2 * 3] = 0;                       B[1 + 2 * 4] = -1. / z * f;              B[1 + 2 * 5] = y / z_2 * f;
  }
}

static void skew3(const double* w, double* S /* col-major */) {
  S[0] = 0; S[1] = w[2]; S[2] = -w[1]; S[3] = -w[2]; S[4] = 0; S[5] = w[0]; S[6] = w[1]; S[7] = -w[0]; S[8] = 0;
}
static void mm3(const double* A, const double* B, double* C) {
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { double t = 0; for (int m = 0; m < 3; ++m) t += A[r + 3 * m] * B[m + 3 * c]; C[r + 3 * c] = t; }
}
/* estimate <- exp(update) * estimate ; update = (omega, upsilon) */
void orc_ba_oplus_cams(int nc, double* cams, const int* hidx, const double* x) {
  for (int v = 0; v < nc; ++v) {
    if (hidx[v] < 0) continue;
    const double* u = x + 6 * (size_t)hidx[v]; double* T = cams + 12 * (size_t)v;
    double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double Om[9], Om2[9], R[9], V[9];
    skew3(u, Om); mm3(Om, Om, Om2);
    for (int i = 0; i < 9; ++i) { double I = (i % 4 == 0);
      if (theta < 0.00001) { R[i] = I + Om[i] + Om2[i]; V[i] = R[i]; }
      else { R[i] = I + sin(theta) / theta * Om[i] + (1 - cos(theta)) / (theta * theta) * Om2[i];
             V[i] = I + (1 - cos(theta)) / (theta * theta) * Om[i] + (theta - sin(theta)) / (theta * theta * theta) * Om2[i]; } }
    double Rn[9], tn[3];
    mm3(R, T, Rn);
    for (int r = 0; r < 3; ++r) tn[r] = R[r] * T[9] + R[r + 3] * T[10] + R[r + 6] * T[11] + V[r] * u[3] + V[r + 3] * u[4] + V[r + 6] * u[5];
    memcpy(T, Rn, sizeof(Rn)); T[9] = tn[0]; T[10] = tn[1]; T[11] = tn[2];
  }
}
void orc_ba_oplus_pts(int np, double* pts, const int* hidx_local /* landmark index or -1 */, const double* xl) {
  for (int v = 0; v < np; ++v) { if (hidx_local[v] < 0) continue; for (int i = 0; i < 3; ++i) pts[3 * (size_t)v + i] += xl[3 * (size_t)hidx_local[v] + i]; }
}
