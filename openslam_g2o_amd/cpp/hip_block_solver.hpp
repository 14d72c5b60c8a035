// Host-side C++ mirror of g2o::Solver / g2o::BlockSolver<Traits> over the libg2ohip C ABI.
//
// Same member names, argument meaning and error behaviour as the reference
//   /root/reference/g2o/core/solver.h:44-149         (init, buildStructure, buildSystem, solve,
//                                                     setLambda, restoreDiagonal, x, b, vectorSize)
//   /root/reference/g2o/core/block_solver.h:98-178   (BlockSolver<Traits>)
// but over flat index / Jacobian arrays, so it compiles without g2o or Eigen.  The adapter that
// derives from g2o::BlockSolverBase and fills these arrays from a SparseOptimizer is shown in
// INTEGRATION.md (it needs the consumer's g2o + Eigen headers).  Failure is `false` + text on
// stderr, like the reference path; no exceptions.
#pragma once
#include <cstdio>
#include <vector>

#include "g2ohip.h"

namespace g2o_hip {

template <int PoseDim, int LandmarkDim>
class HipBlockSolver {
 public:
  explicit HipBlockSolver(int device = 0) {
    if (g2ohip_create(&h_, PoseDim, LandmarkDim, device) != G2OHIP_OK) report("g2ohip_create");
  }
  ~HipBlockSolver() { g2ohip_destroy(h_); }
  HipBlockSolver(const HipBlockSolver&) = delete;
  HipBlockSolver& operator=(const HipBlockSolver&) = delete;
  bool valid() const { return h_ != nullptr; }

  // Solver::init(optimizer, online)
  bool init() { return ok(g2ohip_init(h_), "init"); }
  // graph topology: one edge set per edge type; indices are hessianIndex values (-1 fixed)
  int addEdgeSet(int errorDim, int n, const int32_t* v0, const int32_t* v1) {
    int rc = g2ohip_add_edge_set(h_, errorDim, n, v0, v1);
    if (rc < 0) report("addEdgeSet");
    return rc;
  }
  // Solver::buildStructure(zeroBlocks)
  bool buildStructure(int numPoses, int numLandmarks, bool schur) {
    bool r = ok(g2ohip_build_structure(h_, numPoses, numLandmarks, schur ? 1 : 0), "buildStructure");
    if (r) {
      x_.assign(g2ohip_vector_size(h_), 0.0);
      b_.assign(g2ohip_vector_size(h_), 0.0);
    }
    return r;
  }
  bool setEdgeData(int set, const double* J0, const double* J1, const double* omega, const double* err, bool onDevice = false) {
    return ok(g2ohip_set_edge_data(h_, set, J0, J1, omega, err, onDevice ? 1 : 0), "setEdgeData");
  }
  bool setRobustKernelHuber(int set, double delta) { return ok(g2ohip_set_robust_kernel(h_, set, G2OHIP_KERNEL_HUBER, delta), "setRobustKernel"); }
  // any kernel of robust_kernel_impl.cpp: G2OHIP_KERNEL_{HUBER, PSEUDOHUBER, CAUCHY, SATURATED, DCS}
  bool setRobustKernel(int set, int kind, double delta) { return ok(g2ohip_set_robust_kernel(h_, set, kind, delta), "setRobustKernel"); }
  // Solver::buildSystem(): afterwards b() holds -J' Omega e (block_solver.hpp:551-557)
  bool buildSystem() {
    if (!ok(g2ohip_build_system(h_), "buildSystem")) return false;
    return ok(g2ohip_copy_b(h_, b_.data()), "b");
  }
  // Solver::solve(): false iff not positive definite; x() valid afterwards, b() untouched
  bool solve() {
    int rc = g2ohip_solve(h_);
    if (rc == G2OHIP_NOT_PD) return false;
    if (!ok(rc, "solve")) return false;
    return ok(g2ohip_copy_x(h_, x_.data()), "x");
  }
  bool setLambda(double lambda, bool backup = false) { return ok(g2ohip_set_lambda(h_, lambda, backup ? 1 : 0), "setLambda"); }
  void restoreDiagonal() { (void)ok(g2ohip_restore_diagonal(h_), "restoreDiagonal"); }
  double chi2() {
    double c = 0;
    (void)ok(g2ohip_chi2(h_, &c), "chi2");
    return c;
  }
  double maxDiagonal() {
    double m = 0;
    (void)ok(g2ohip_max_diagonal(h_, &m), "maxDiagonal");
    return m;
  }
  // BlockSolverBase::multiplyHessian(dest, src)
  void multiplyHessian(double* dest, const double* src) { (void)ok(g2ohip_multiply_hessian(h_, dest, src), "multiplyHessian"); }
  // Solver::computeMarginals (block_solver.hpp:489-498): out[i] = column-major PoseDim x PoseDim block (rows[i], cols[i])
  bool computeMarginals(int n, const int32_t* rows, const int32_t* cols, double* out) {
    return g2ohip_compute_marginals(h_, n, rows, cols, out) == G2OHIP_OK;
  }
  // LinearSolverPCG instead of the direct solver (linear_solver_pcg.h:53-85)
  bool usePCG(bool on, double tolerance = 1e-6) {
    return ok(g2ohip_set_option(h_, "linear_solver", on ? 1.0 : 0.0), "set_option") &&
           ok(g2ohip_set_option(h_, "pcg_tolerance", tolerance), "set_option");
  }
  double* x() { return x_.data(); }
  const double* b() const { return b_.data(); }
  size_t vectorSize() const { return x_.size(); }
  bool supportsSchur() const { return true; }
  g2ohip_solver* handle() { return h_; }

 private:
  bool ok(int rc, const char* what) {
    if (rc >= 0) return true;
    report(what);
    return false;
  }
  void report(const char* what) { std::fprintf(stderr, "g2o_hip::HipBlockSolver::%s: %s\n", what, g2ohip_last_error()); }
  g2ohip_solver* h_ = nullptr;
  std::vector<double> x_, b_;
};

typedef HipBlockSolver<6, 3> HipBlockSolver_6_3;  // block_solver.h:180-186
typedef HipBlockSolver<3, 2> HipBlockSolver_3_2;
typedef HipBlockSolver<7, 3> HipBlockSolver_7_3;

}  // namespace g2o_hip
