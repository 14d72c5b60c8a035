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
#ifdef _OPENMP   /* the reference linearises inside its parallel loop over the edges, block_solver.hpp:527-531 */
#pragma omp parallel for default(shared) if (n > 100)
#endif
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

/* ===========================================================================
 * 3-D pose graphs: VertexSE3 / EdgeSE3 (config 2, sphere).  References:
 *   EdgeSE3::computeError            g2o/types/slam3d/edge_se3.cpp:48-53
 *   EdgeSE3::linearizeOplus          g2o/types/slam3d/edge_se3.cpp:63-75
 *   computeEdgeSE3Gradient           g2o/types/slam3d/isometry3d_gradients.h:86-192 (gcc branch)
 *   skew / skewT                     g2o/types/slam3d/isometry3d_gradients.h:41-84
 *   compute_dq_dR                    g2o/types/slam3d/dquat2mat.cpp:9-62 (+ the Maxima-generated
 *                                    partials, re-derived by hand below from the same four-case
 *                                    rotation->quaternion formulas)
 *   toVectorMQT/toCompactQuaternion/fromCompactQuaternion/fromVectorQT
 *                                    g2o/types/slam3d/isometry3d_mappings.cpp:32-44,77-136
 *   VertexSE3::oplusImpl             g2o/types/slam3d/vertex_se3.h:107-116
 *   EdgeSE3::read (quaternion normalisation) g2o/types/slam3d/edge_se3.cpp:14-36
 * Isometries are stored as T[12] = R (column-major 3x3) | t.
 * ======================================================================== */
#define R_(T, a, b) (T)[(a) + 3 * (b)]

static void iso_mul(const double* A, const double* B, double* C) {
  double R[9], t[3];
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { double s = 0; for (int m = 0; m < 3; ++m) s += R_(A, r, m) * R_(B, m, c); R[r + 3 * c] = s; }
  for (int r = 0; r < 3; ++r) t[r] = R_(A, r, 0) * B[9] + R_(A, r, 1) * B[10] + R_(A, r, 2) * B[11] + A[9 + r];
  memcpy(C, R, sizeof(R)); C[9] = t[0]; C[10] = t[1]; C[11] = t[2];
}
static void iso_inv(const double* A, double* C) {
  double R[9], t[3];
  for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R[r + 3 * c] = R_(A, c, r);
  for (int r = 0; r < 3; ++r) t[r] = -(R[r] * A[9] + R[r + 3] * A[10] + R[r + 6] * A[11]);
  memcpy(C, R, sizeof(R)); C[9] = t[0]; C[10] = t[1]; C[11] = t[2];
}
/* Quaterniond(w,x,y,z).toRotationMatrix() (no normalisation, like Eigen) */
static void quat_to_R(double w, double x, double y, double z, double* R) {
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R_(R, 0, 0) = 1 - (tyy + tzz); R_(R, 0, 1) = txy - twz; R_(R, 0, 2) = txz + twy;
  R_(R, 1, 0) = txy + twz; R_(R, 1, 1) = 1 - (txx + tzz); R_(R, 1, 2) = tyz - twx;
  R_(R, 2, 0) = txz - twy; R_(R, 2, 1) = tyz + twx; R_(R, 2, 2) = 1 - (txx + tyy);
}
/* Quaterniond(R) (four cases), then normalize + w >= 0 (isometry3d_mappings.cpp:38-44); q = (x,y,z,w) */
static int R_to_quat(const double* R, double* q) {
  double tr = R_(R, 0, 0) + R_(R, 1, 1) + R_(R, 2, 2);
  int which;
  if (tr > 0) { double t = sqrt(tr + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
    q[0] = (R_(R, 2, 1) - R_(R, 1, 2)) * t; q[1] = (R_(R, 0, 2) - R_(R, 2, 0)) * t; q[2] = (R_(R, 1, 0) - R_(R, 0, 1)) * t; which = 0; }
  else { int i = 0; if (R_(R, 1, 1) > R_(R, 0, 0)) i = 1; if (R_(R, 2, 2) > R_(R, i, i)) i = 2; int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R_(R, i, i) - R_(R, j, j) - R_(R, k, k) + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
    q[3] = (R_(R, k, j) - R_(R, j, k)) * t; q[j] = (R_(R, j, i) + R_(R, i, j)) * t; q[k] = (R_(R, k, i) + R_(R, i, k)) * t; which = i + 1; }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
  if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
  return which;
}
/* d(qx,qy,qz)/d vec(R) (3 x 9, column-major vec), partials of the case formulas of dquat2mat.cpp:9-43 */
static void dq_dR(const double* R, double* D /* row-major 3x9: D[row*9 + col] */) {
  for (int i = 0; i < 27; ++i) D[i] = 0;
#define COL(a, b) ((a) + 3 * (b))
  double r00 = R_(R, 0, 0), r11 = R_(R, 1, 1), r22 = R_(R, 2, 2);
  double tr = r00 + r11 + r22, qw;
  if (tr > 0) {
    double w = 0.5 * sqrt(tr + 1.0); qw = w;
    double num[3] = {R_(R, 2, 1) - R_(R, 1, 2), R_(R, 0, 2) - R_(R, 2, 0), R_(R, 1, 0) - R_(R, 0, 1)};
    int pa[3] = {2, 0, 1}, pb[3] = {1, 2, 0};     /* numerator of component c is r[pa][pb] - r[pb][pa] */
    for (int c = 0; c < 3; ++c) {
      double dd = -num[c] / (32.0 * w * w * w);
      D[c * 9 + COL(0, 0)] = dd; D[c * 9 + COL(1, 1)] = dd; D[c * 9 + COL(2, 2)] = dd;
      D[c * 9 + COL(pa[c], pb[c])] = 0.25 / w; D[c * 9 + COL(pb[c], pa[c])] = -0.25 / w;
    }
  } else {
    int i = 0; if ((r00 > r11) & (r00 > r22)) i = 0; else if (r11 > r22) i = 1; else i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = 0.5 * sqrt(1.0 + R_(R, i, i) - R_(R, j, j) - R_(R, k, k));   /* q_i */
    qw = (R_(R, k, j) - R_(R, j, k)) / (4.0 * s);
    /* q_i = s: d/dr_ii = 1/(8s), d/dr_jj = d/dr_kk = -1/(8s) */
    D[i * 9 + COL(i, i)] = 1.0 / (8.0 * s); D[i * 9 + COL(j, j)] = -1.0 / (8.0 * s); D[i * 9 + COL(k, k)] = -1.0 / (8.0 * s);
    /* q_j = (r_ji + r_ij) / (4 s), q_k = (r_ki + r_ik) / (4 s) */
    int other[2] = {j, k};
    for (int o = 0; o < 2; ++o) { int c = other[o]; double num = R_(R, c, i) + R_(R, i, c);
      D[c * 9 + COL(c, i)] += 0.25 / s; D[c * 9 + COL(i, c)] += 0.25 / s;
      double dd = num / (32.0 * s * s * s);
      D[c * 9 + COL(i, i)] += -dd; D[c * 9 + COL(j, j)] += dd; D[c * 9 + COL(k, k)] += dd; }
  }
  if (qw <= 0) for (int i = 0; i < 27; ++i) D[i] = -D[i];
#undef COL
}

/* qt: [n][7] = x y z qx qy qz qw  ->  T [n][12]; normalize != 0 for EDGE_SE3:QUAT measurements */
void orc_se3_from_qt(int n, const double* qt, int normalize, double* T) {
  for (int k = 0; k < n; ++k) {
    const double* v = qt + 7 * (size_t)k; double* o = T + 12 * (size_t)k;
    double q[4] = {v[3], v[4], v[5], v[6]};
    if (normalize) { double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (int i = 0; i < 4; ++i) q[i] /= nn; }
    quat_to_R(q[3], q[0], q[1], q[2], o);
    o[9] = v[0]; o[10] = v[1]; o[11] = v[2];
  }
}

void orc_se3_edges(int n, const double* poses, const int* vi, const int* vj, const double* meas /* [n][12] */,
                   double* J0, double* J1, double* err) {
  for (int e = 0; e < n; ++e) {
    const double* Xi = poses + 12 * (size_t)vi[e]; const double* Xj = poses + 12 * (size_t)vj[e]; const double* Z = meas + 12 * (size_t)e;
    double A[12], Xii[12], B[12], E[12], q[4];
    iso_inv(Z, A); iso_inv(Xi, Xii); iso_mul(Xii, Xj, B); iso_mul(A, B, E);
    R_to_quat(E, q);
    if (err) { double* r = err + 6 * (size_t)e; r[0] = E[9]; r[1] = E[10]; r[2] = E[11]; r[3] = q[0]; r[4] = q[1]; r[5] = q[2]; }
    if (!J0) continue;
    double Ji[36], Jj[36];   /* column-major 6x6 */
    for (int i = 0; i < 36; ++i) Ji[i] = Jj[i] = 0;
    const double* Ra = A; const double* Rab = E; const double* Rbc = B; const double* tbc = B + 9;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { Ji[r + 6 * c] = -R_(Ra, r, c); Jj[r + 6 * c] = R_(Rab, r, c); }
    { /* dte/dqi = Ra * skewT(tbc);  skewT rows: [0 -z y; z 0 -x; -y x 0] with doubled components */
      double x = 2 * tbc[0], y = 2 * tbc[1], z = 2 * tbc[2];
      double S[9]; R_(S, 0, 0) = 0; R_(S, 0, 1) = -z; R_(S, 0, 2) = y; R_(S, 1, 0) = z; R_(S, 1, 1) = 0; R_(S, 1, 2) = -x; R_(S, 2, 0) = -y; R_(S, 2, 1) = x; R_(S, 2, 2) = 0;
      for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { double s = 0; for (int m = 0; m < 3; ++m) s += R_(Ra, r, m) * R_(S, m, c); Ji[r + 6 * (3 + c)] = s; }
      /* dte/dqj = Rab * skew(tc) = 0 (tc = 0) */ }
    double D[27]; dq_dR(E, D);
    { /* dre/dqi: M = [vec(Ra*Sxt) vec(Ra*Syt) vec(Ra*Szt)], skewT(Rbc) */
      double r11 = 2 * R_(Rbc, 0, 0), r12 = 2 * R_(Rbc, 0, 1), r13 = 2 * R_(Rbc, 0, 2), r21 = 2 * R_(Rbc, 1, 0), r22 = 2 * R_(Rbc, 1, 1), r23 = 2 * R_(Rbc, 1, 2),
             r31 = 2 * R_(Rbc, 2, 0), r32 = 2 * R_(Rbc, 2, 1), r33 = 2 * R_(Rbc, 2, 2);
      double S[3][9] = {{0, 0, 0, r31, r32, r33, -r21, -r22, -r23}, {-r31, -r32, -r33, 0, 0, 0, r11, r12, r13}, {r21, r22, r23, -r11, -r12, -r13, 0, 0, 0}};  /* row-wise */
      for (int a = 0; a < 3; ++a) { double M[9];
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { double s = 0; for (int m = 0; m < 3; ++m) s += R_(Ra, r, m) * S[a][m * 3 + c]; M[r + 3 * c] = s; }
        for (int r = 0; r < 3; ++r) { double s = 0; for (int m = 0; m < 9; ++m) s += D[r * 9 + m] * M[m]; Ji[(3 + r) + 6 * (3 + a)] = s; } } }
    { /* dre/dqj: M = [vec(Rab*Sx) ...], skew(Rc = I) */
      double S[3][9] = {{0, 0, 0, 0, 0, -2, 0, 2, 0}, {0, 0, 2, 0, 0, 0, -2, 0, 0}, {0, -2, 0, 2, 0, 0, 0, 0, 0}};   /* row-wise, Rc = I */
      for (int a = 0; a < 3; ++a) { double M[9];
        for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) { double s = 0; for (int m = 0; m < 3; ++m) s += R_(Rab, r, m) * S[a][m * 3 + c]; M[r + 3 * c] = s; }
        for (int r = 0; r < 3; ++r) { double s = 0; for (int m = 0; m < 9; ++m) s += D[r * 9 + m] * M[m]; Jj[(3 + r) + 6 * (3 + a)] = s; } } }
    memcpy(J0 + 36 * (size_t)e, Ji, sizeof(Ji)); memcpy(J1 + 36 * (size_t)e, Jj, sizeof(Jj));
  }
}

/* estimate <- estimate * fromVectorMQT(update) */
void orc_se3_oplus(int nv, double* poses, const int* hidx, const double* x) {
  for (int v = 0; v < nv; ++v) {
    if (hidx[v] < 0) continue;
    const double* u = x + 6 * (size_t)hidx[v]; double* T = poses + 12 * (size_t)v;
    double inc[12]; double w = 1 - (u[3] * u[3] + u[4] * u[4] + u[5] * u[5]);
    if (w < 0) { for (int i = 0; i < 9; ++i) inc[i] = (i % 4 == 0); } else quat_to_R(sqrt(w), u[3], u[4], u[5], inc);
    inc[9] = u[0]; inc[10] = u[1]; inc[11] = u[2];
    double out[12]; iso_mul(T, inc, out); memcpy(T, out, sizeof(out));
  }
}
