// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// Thin caller around the REFERENCE's own compiled code (built by oracle/Makefile
// from /root/reference/EXTERNAL/csparse/*.c and
// /root/reference/g2o/solvers/csparse/csparse_helper.cpp into oracle/_ref/).
// It drives those functions in the order LinearSolverCSparse does:
//   computeSymbolicDecomposition  g2o/solvers/csparse/linear_solver_csparse.h:246-308
//   solve                         g2o/solvers/csparse/linear_solver_csparse.h:106-142
// so that a scalar upper-triangular CCS matrix + rhs gives the reference's x.
// Nothing here re-implements CSparse arithmetic: cs_amd, cs_symperm, cs_etree,
// cs_post, cs_counts, cs_cumsum, cs_pinv and csparse_extension::cs_cholsolsymb
// are the reference's objects.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "g2o/solvers/csparse/csparse_helper.h"

extern "C" {

// Block AMD ordering exactly as linear_solver_csparse.h:252-263: cs_amd(1, blockPattern).
// Ap/Ai: block-level upper-triangular CCS pattern (what fillBlockStructure produces,
// sparse_block_matrix.hpp:519-545).  P_out[nb] = block permutation.
int ref_block_amd(int nb, const int* Ap, const int* Ai, int* P_out) {
  cs aux;
  aux.nzmax = Ap[nb];
  aux.m = aux.n = nb;
  aux.p = const_cast<int*>(Ap);
  aux.i = const_cast<int*>(Ai);
  aux.x = NULL;
  aux.nz = -1;
  int* P = cs_amd(1, &aux);
  if (!P) return 0;
  memcpy(P_out, P, sizeof(int) * nb);
  cs_free(P);
  return 1;
}

// Scalar AMD ordering as cs_schol(1, A) would compute it (non-block path, :249-250).
int ref_scalar_amd(int n, const int* Ap, const int* Ai, int* P_out) {
  cs aux;
  aux.nzmax = Ap[n]; aux.m = aux.n = n;
  aux.p = const_cast<int*>(Ap); aux.i = const_cast<int*>(Ai); aux.x = NULL; aux.nz = -1;
  int* P = cs_amd(1, &aux);
  if (!P) return 0;
  memcpy(P_out, P, sizeof(int) * n);
  cs_free(P);
  return 1;
}

struct RefSym {
  css* S;
  int n;
  std::vector<double> work;
  std::vector<int> iwork;
};

// Symbolic step with a given scalar permutation (linear_solver_csparse.h:283-298).
void* ref_symbolic(int n, const int* Ap, const int* Ai, const int* scalarPerm) {
  cs A;
  A.nzmax = Ap[n]; A.m = A.n = n;
  A.p = const_cast<int*>(Ap); A.i = const_cast<int*>(Ai); A.x = NULL; A.nz = -1;
  RefSym* R = new RefSym;
  R->n = n;
  css* S = (css*)cs_calloc(1, sizeof(css));
  S->pinv = cs_pinv(scalarPerm, n);
  cs* C = cs_symperm(&A, S->pinv, 0);
  S->parent = cs_etree(C, 0);
  int* post = cs_post(S->parent, n);
  int* c = cs_counts(C, S->parent, post, 0);
  cs_free(post);
  cs_spfree(C);
  S->cp = (int*)cs_malloc(n + 1, sizeof(int));
  S->unz = S->lnz = cs_cumsum(S->cp, c, n);
  cs_free(c);
  R->S = S;
  R->work.resize(2 * (size_t)n);   // _csWorkspace sizing, linear_solver_csparse.h:114-120
  R->iwork.resize(4 * (size_t)n);
  return R;
}

double ref_lnz(void* h) { return ((RefSym*)h)->S->lnz; }

// Numeric factor + solve: csparse_extension::cs_cholsolsymb (csparse_helper.cpp:56-81).
// x (in: b, out: solution).  Returns 1 ok, 0 not positive definite.
int ref_cholsolve(void* h, const int* Ap, const int* Ai, const double* Ax, double* x) {
  RefSym* R = (RefSym*)h;
  cs A;
  A.nzmax = Ap[R->n]; A.m = A.n = R->n;
  A.p = const_cast<int*>(Ap); A.i = const_cast<int*>(Ai); A.x = const_cast<double*>(Ax); A.nz = -1;
  return g2o::csparse_extension::cs_cholsolsymb(&A, x, R->S, R->work.data(), R->iwork.data());
}

void ref_free(void* h) {
  RefSym* R = (RefSym*)h;
  cs_sfree(R->S);
  delete R;
}

}  // extern "C"
