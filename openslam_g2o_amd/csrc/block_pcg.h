// Preconditioned conjugate gradients on a symmetric positive definite block matrix given as the upper triangle
// in block-CCS (the reduced pose system Hschur, or Hpp without Schur), resident on one MI355X.
//
// Replaces g2o::LinearSolverPCG<MatrixType>::solve
//     /root/reference/g2o/solvers/pcg/linear_solver_pcg.hpp:79-196   (block-Jacobi preconditioner J_i = A_ii^-1,
//     x0 = 0, stop when r'Jr <= max(tolerance * r0'Jr0, residual of the previous solve) or after maxIter
//     iterations, _residual = 0.5 * r'Jr kept for the next call: linear_solver_pcg.h:53-57,91-95)
// Design: everything stays on the device, including the scalars (alpha, beta, the stopping test); the host
// launches blocks of iterations and looks at the "done" flag once per block.  The symmetric product
// y = A d is a row-wise GATHER over a per-vertex entry list built once per pattern (no fp64 atomics: the
// reference's serial loop order is replaced by a fixed gather order, results are deterministic); dot products
// are two-stage reductions with a fixed partition.
#pragma once
#include <functional>

#include "common.h"

namespace g2ohip {

struct PcgOptions {
  double tolerance = 1e-6;         // LinearSolverPCG::_tolerance
  bool absolute_tolerance = true;  // _absoluteTolerance
  int max_iter = -1;               // _maxIter (< 0: the matrix dimension)
  int check_every = 16;            // host looks at the device "done" flag once per this many iterations
};

class BlockPCG {
 public:
  explicit BlockPCG(int block_size) : bs_(block_size) {}
  PcgOptions opt;
  // pattern: upper block-CCS (rows <= col, sorted, diagonal present in every column)
  void analyze(int nb, const int* colptr, const int* rowidx, hipStream_t st);
  bool analyzed() const { return nb_ > 0; }
  // x = A \ b; A values [nnzb][bs*bs] column-major blocks in pattern order; device pointers.  Returns false when a
  // diagonal block is singular / the iteration broke down (NaN).  Synchronises st.
  bool solve(const double* dA, const double* d_b, double* d_x, hipStream_t st);
  // The same iteration with the matrix given as an operator (matrix-free Schur complement: linear_solver_pcg.hpp:79-196 only
  // ever needs A*d and the diagonal blocks): diag_blocks [nb][bs*bs] column-major (block i = A_ii), apply(d_in, d_out)
  // queues d_out = A d_in on st.  analyze_operator(nb) instead of analyze().
  void analyze_operator(int nb, hipStream_t st);
  bool solve_operator(const double* d_diag_blocks, const std::function<void(const double*, double*)>& apply, const double* d_b,
                      double* d_x, hipStream_t st);
  // d_out = A d_in with the analysed pattern (deterministic gather form of the symmetric product)
  void multiply(const double* dA, const double* d_in, double* d_out, hipStream_t st);
  int last_iterations() const { return iters_; }
  double residual() const { return residual_; }   // 0.5 * r'Jr of the last solve
  void reset_residual() { residual_ = -1.0; }

 private:
  int bs_, nb_ = 0, iters_ = 0;
  double residual_ = -1.0;
  DevBuf<int> d_diag, d_ent_ptr, d_ent;   // per block row: diagonal block id; entries (block id << 1 | transposed, other block row)
  DevBuf<int> d_ent_other;
  DevBuf<double> d_zero_scal;   // multiply(): an all-zero scalar block (the product kernel checks its "done" flag)
  DevBuf<double> d_J, d_r, d_d, d_q, d_s, d_part, d_scal;
  int n_part_ = 0;
};

}  // namespace g2ohip
