/* ===========================================================================
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (oracle) of g2o's BlockSolver<p,l>
 * hot path.  Never linked into, imported by or called from the product path
 * (openslam_g2o_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * It is a restatement, in plain C99, of the algorithm in (all paths relative
 * to /root/reference):
 *   buildStructure      g2o/core/block_solver.hpp:142-295
 *   buildSystem         g2o/core/block_solver.hpp:501-560
 *   constructQuadraticForm (binary)  g2o/core/base_binary_edge.hpp:54-120
 *   constructQuadraticForm (unary)   g2o/core/base_unary_edge.hpp:42-72
 *   robustInformation   g2o/core/base_edge.h:96-102
 *   Huber               g2o/core/robust_kernel_impl.cpp:65-78
 *   setLambda/restore   g2o/core/block_solver.hpp:563-604
 *   solve (Schur, back-substitution)  g2o/core/block_solver.hpp:353-486
 *   fillCCS (scalar upper CCS)        g2o/core/sparse_block_matrix_ccs.h:143-199
 *   up-looking Cholesky + solves      g2o/solvers/csparse/csparse_helper.cpp:56-143,
 *                                     EXTERNAL/csparse/cs_{ereach,etree,lsolve,ltsolve,ipvec,pvec}.c
 *   LM helpers (lambda init, scale)   g2o/core/optimization_algorithm_levenberg.cpp:149-172
 *
 * Parity pin: validated against the reference's own compiled CSparse path
 * (oracle/_ref, built from the reference sources by oracle/Makefile) in
 * tests/test_oracle_ref.py, and against dense numpy solves.  The reference has no
 * assertion-bearing tests or golden vectors for this path.
 * Everything is serial fp64 with int32 indices, like the reference default build
 * (G2O_USE_OPENMP OFF, CMakeLists.txt:137).
 * ======================================================================== */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_MAX_SETS 16

typedef struct {
  int d, n, dim0, dim1, unary;
  int *v0, *v1;                 /* hessian indices, -1 == fixed (or absent for unary) */
  const double *J0, *J1, *omega, *err;
  double huber_delta;           /* <= 0: no robust kernel */
  int kernel_kind;              /* which robust kernel huber_delta parametrises (0/1: Huber; orc_set_robust_kernel) */
  long *o00, *o11, *o01;        /* offsets (in doubles) of the mapped blocks, -1 == none */
  char *k00, *k11, *k01;        /* 0 none, 1 Hpp, 2 Hll, 3 Hpl */
  char *tr01;                   /* edge writes the off-diagonal block transposed */
} OrcSet;

/* n-ary edges (BaseMultiEdge, g2o/core/base_multi_edge.h): `arity` vertices per edge, one Jacobian array per vertex position,
 * the off-diagonal blocks of an edge kept in the order of internal::computeUpperTriangleIndex (base_multi_edge.hpp:28-32) */
#define ORC_MAX_MSETS 8
#define ORC_MAX_ARITY 4
#define ORC_MAX_PAIRS (ORC_MAX_ARITY * (ORC_MAX_ARITY - 1) / 2)
typedef struct {
  int d, n, arity;
  int *v;                       /* [n][arity] hessian indices, -1 == fixed */
  const double *J[ORC_MAX_ARITY], *omega, *err;
  double delta; int kernel_kind; /* delta <= 0: no robust kernel */
  long *odiag; char *kdiag;     /* [n][arity]: the vertex's diagonal block (1 Hpp, 2 Hll) */
  long *ooff; char *koff, *troff; /* [n][ORC_MAX_PAIRS] at computeUpperTriangleIndex(i, j): _hessian[idx] (.transposed) */
} OrcMulti;

typedef struct {
  int n;                        /* scalar dimension */
  int nb, bs;                   /* blocks, block size */
  int *Ap, *Ai; double *Ax;     /* scalar upper CCS (fillCCS) */
  int *perm, *pinv;             /* scalar permutation */
  int *Cp, *Ci; double *Cx; int *Cmap; /* C = P A P' upper, Cmap: C entry <- A entry */
  int *parent, *Lp, *Li; double *Lx;
  double lnz;
  int *iw; double *xw;
  int have_symbolic;
} OrcChol;

typedef struct {
  int p, l, nP, nL, doSchur;
  int sizeP, sizeL;
  int nsets; OrcSet sets[ORC_MAX_SETS];
  int nmsets; OrcMulti msets[ORC_MAX_MSETS];
  int *pp_colptr, *pp_row, *pp_diag; int pp_nnzb; double *Hpp;
  int *pl_colptr, *pl_row; int pl_nnzb; double *Hpl;
  double *Hll;
  int *hs_colptr, *hs_row; int hs_nnzb; double *Hschur;
  double *Dinv, *coeff, *bschur, *x, *b;
  double *bkP, *bkL;
  OrcChol chol;
  int n_extra; int *extra_r, *extra_c;  /* extra structural Hschur blocks (sharding support, no reference analogue) */
  int ordering;                 /* 0 natural, 1 own block minimum degree, 2 external block perm */
  int *ext_block_perm;
  double t_schur, t_linear, t_numeric;  /* seconds, last solve */
} Orc;

/* OpenMP build (libg2o_oracle_omp.so, -fopenmp): the reference's optional G2O_OPENMP regions -- buildSystem parallel over
 * edges with a lock per vertex (block_solver.hpp:527, base_binary_edge.hpp:69-72,114-117), the Schur complement parallel
 * over landmarks with a lock per pose row (block_solver.hpp:378-380,411-413).  The default build has no threads
 * (the reference's default: G2O_USE_OPENMP OFF, CMakeLists.txt:137). */
#ifdef _OPENMP
#include <omp.h>
static omp_lock_t* g_locks = NULL;
static long g_nlocks = 0;
static void ensure_locks(long n) {
  if (n <= g_nlocks) return;
  g_locks = (omp_lock_t*)realloc(g_locks, sizeof(omp_lock_t) * (size_t)n);
  for (long i = g_nlocks; i < n; ++i) omp_init_lock(&g_locks[i]);
  g_nlocks = n;
}
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif

static double now_s(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ---------------------------------------------------------------- utilities */
static int cmp_i64(const void* a, const void* b) {
  long long x = *(const long long*)a, y = *(const long long*)b;
  return (x > y) - (x < y);
}
static int find_row(const int* colptr, const int* row, int c, int r) {
  int lo = colptr[c], hi = colptr[c + 1] - 1;
  while (lo <= hi) { int m = (lo + hi) >> 1; if (row[m] == r) return m; if (row[m] < r) lo = m + 1; else hi = m - 1; }
  return -1;
}
/* keys (c * N + r) sorted unique -> block CCS with ascending rows per column
 * (== iteration order of the reference's std::map<int,Block*> columns). */
static void keys_to_ccs(long long* keys, long nk, long long N, int ncols, int** colptr, int** row, int* nnzb) {
  qsort(keys, (size_t)nk, sizeof(long long), cmp_i64);
  long u = 0;
  for (long i = 0; i < nk; ++i) if (i == 0 || keys[i] != keys[i - 1]) keys[u++] = keys[i];
  *colptr = (int*)calloc((size_t)ncols + 1, sizeof(int));
  *row = (int*)malloc(sizeof(int) * (size_t)(u > 0 ? u : 1));
  for (long i = 0; i < u; ++i) { int c = (int)(keys[i] / N); (*colptr)[c + 1]++; (*row)[i] = (int)(keys[i] % N); }
  for (int c = 0; c < ncols; ++c) (*colptr)[c + 1] += (*colptr)[c];
  *nnzb = (int)u;
}

/* ------------------------------------------------------------- construction */
Orc* orc_create(int p, int l, int nP, int nL, int doSchur) {
  Orc* s = (Orc*)calloc(1, sizeof(Orc));
  s->p = p; s->l = l; s->nP = nP; s->nL = nL; s->doSchur = doSchur && nL > 0;
  s->sizeP = p * nP; s->sizeL = l * nL;
  s->ordering = 1;
  return s;
}

/* v1 == NULL -> unary edges (dim1 = 0).  Indices are hessianIndex values in
 * indexMapping order: poses [0,nP), landmarks [nP, nP+nL), -1 fixed
 * (sparse_optimizer.cpp:166-190). */
int orc_add_edge_set(Orc* s, int d, int n, const int* v0, const int* v1) {
  if (s->nsets >= ORC_MAX_SETS) return -1;
  OrcSet* e = &s->sets[s->nsets];
  memset(e, 0, sizeof(*e));
  e->d = d; e->n = n; e->unary = (v1 == NULL);
  e->v0 = (int*)malloc(sizeof(int) * (size_t)n); memcpy(e->v0, v0, sizeof(int) * (size_t)n);
  e->v1 = (int*)malloc(sizeof(int) * (size_t)n);
  if (v1) memcpy(e->v1, v1, sizeof(int) * (size_t)n); else for (int i = 0; i < n; ++i) e->v1[i] = -1;
  return s->nsets++;
}
/* vertex class dimensions of a set: taken from the first non-fixed index on each side */
void orc_set_dims(Orc* s, int set, int dim0, int dim1) { s->sets[set].dim0 = dim0; s->sets[set].dim1 = dim1; }

static int vdim(const Orc* s, int idx) { return idx < s->nP ? s->p : s->l; }

/* internal::computeUpperTriangleIndex, base_multi_edge.hpp:28-32 (i < j) */
static int upper_triangle_index(int i, int j) { int elemsUpToCol = ((j - 1) * j) / 2; return elemsUpToCol + i; }

/* An edge set of n-ary edges (BaseMultiEdge<D, E>): v[n][arity] hessian indices, -1 fixed; vertex dimensions follow the index
 * (pose / landmark).  Returns the multi-set id. */
int orc_add_multi_edge_set(Orc* s, int d, int n, int arity, const int* v) {
  if (s->nmsets >= ORC_MAX_MSETS || arity < 2 || arity > ORC_MAX_ARITY) return -1;
  OrcMulti* e = &s->msets[s->nmsets];
  memset(e, 0, sizeof(*e));
  e->d = d; e->n = n; e->arity = arity;
  e->v = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1) * arity);
  memcpy(e->v, v, sizeof(int) * (size_t)n * arity);
  return s->nmsets++;
}
/* J[i]: [n][d x dim_i] column-major Jacobians of vertex position i (_jacobianOplus[i]); delta > 0: robust kernel `kind`
 * (0 / 1 Huber, as orc_set_robust_kernel) */
void orc_set_multi_edge_data(Orc* s, int mset, const double* const* J, const double* omega, const double* err, double delta, int kind) {
  OrcMulti* e = &s->msets[mset];
  for (int i = 0; i < e->arity; ++i) e->J[i] = J[i];
  e->omega = omega; e->err = err; e->delta = delta; e->kernel_kind = kind;
}

/* block_solver.hpp:142-295 */
int orc_build_structure(Orc* s) {
  const int nP = s->nP, nL = s->nL, p = s->p, l = s->l;
  long npairs_pp = nP, npairs_pl = 0;
  for (int si = 0; si < s->nsets; ++si) {
    OrcSet* e = &s->sets[si];
    for (int k = 0; k < e->n; ++k) {
      int a = e->v0[k], b = e->v1[k];
      if (a < 0 || b < 0) continue;
      int ma = a >= nP, mb = b >= nP;
      if (!ma && !mb) npairs_pp++; else if (ma != mb) npairs_pl++; else return -2; /* landmark-landmark edge: asserted away at :383 */
    }
  }
  for (int si = 0; si < s->nmsets; ++si) {      /* every vertex pair (i < j) of an n-ary edge: block_solver.hpp:208-251 */
    OrcMulti* e = &s->msets[si];
    for (int k = 0; k < e->n; ++k)
      for (int i = 0; i < e->arity; ++i)
        for (int j = i + 1; j < e->arity; ++j) {
          int a = e->v[(size_t)k * e->arity + i], b = e->v[(size_t)k * e->arity + j];
          if (a < 0 || b < 0) continue;
          int ma = a >= nP, mb = b >= nP;
          if (!ma && !mb) npairs_pp++; else if (ma != mb) npairs_pl++; else return -2;
        }
  }
  long long* kpp = (long long*)malloc(sizeof(long long) * (size_t)(npairs_pp + 1));
  long long* kpl = (long long*)malloc(sizeof(long long) * (size_t)(npairs_pl + 1));
  long ipp = 0, ipl = 0;
  for (int i = 0; i < nP; ++i) kpp[ipp++] = (long long)i * nP + i;       /* diagonal :178-194 */
  for (int si = 0; si < s->nsets; ++si) {
    OrcSet* e = &s->sets[si];
    for (int k = 0; k < e->n; ++k) {
      int a = e->v0[k], b = e->v1[k];
      if (a < 0 || b < 0) continue;
      int ma = a >= nP, mb = b >= nP;
      if (!ma && !mb) { int r = a < b ? a : b, c = a < b ? b : a; kpp[ipp++] = (long long)c * nP + r; }
      else { int pose = ma ? b : a, lm = (ma ? a : b) - nP; kpl[ipl++] = (long long)lm * nP + pose; }
    }
  }
  for (int si = 0; si < s->nmsets; ++si) {
    OrcMulti* e = &s->msets[si];
    for (int k = 0; k < e->n; ++k)
      for (int i = 0; i < e->arity; ++i)
        for (int j = i + 1; j < e->arity; ++j) {
          int a = e->v[(size_t)k * e->arity + i], b = e->v[(size_t)k * e->arity + j];
          if (a < 0 || b < 0) continue;
          int ma = a >= nP, mb = b >= nP;
          if (!ma && !mb) { int r = a < b ? a : b, c = a < b ? b : a; kpp[ipp++] = (long long)c * nP + r; }
          else { int pose = ma ? b : a, lm = (ma ? a : b) - nP; kpl[ipl++] = (long long)lm * nP + pose; }
        }
  }
  keys_to_ccs(kpp, ipp, nP, nP, &s->pp_colptr, &s->pp_row, &s->pp_nnzb);
  if (nL > 0) keys_to_ccs(kpl, ipl, nP, nL, &s->pl_colptr, &s->pl_row, &s->pl_nnzb);
  free(kpp); free(kpl);
  s->pp_diag = (int*)malloc(sizeof(int) * (size_t)(nP > 0 ? nP : 1));
  for (int c = 0; c < nP; ++c) s->pp_diag[c] = find_row(s->pp_colptr, s->pp_row, c, c);
  s->Hpp = (double*)calloc((size_t)s->pp_nnzb * p * p + 1, sizeof(double));
  s->Hpl = (double*)calloc((size_t)s->pl_nnzb * p * l + 1, sizeof(double));
  s->Hll = (double*)calloc((size_t)nL * l * l + 1, sizeof(double));
  s->x = (double*)calloc((size_t)s->sizeP + s->sizeL + 1, sizeof(double));
  s->b = (double*)calloc((size_t)s->sizeP + s->sizeL + 1, sizeof(double));
  s->coeff = (double*)calloc((size_t)s->sizeP + s->sizeL + 1, sizeof(double));
  s->bschur = (double*)calloc((size_t)s->sizeP + 1, sizeof(double));
  s->Dinv = (double*)calloc((size_t)nL * l * l + 1, sizeof(double));
  s->bkP = (double*)calloc((size_t)s->sizeP + 1, sizeof(double));
  s->bkL = (double*)calloc((size_t)s->sizeL + 1, sizeof(double));

  /* edge -> mapped block memory (mapHessianMemory) */
  for (int si = 0; si < s->nsets; ++si) {
    OrcSet* e = &s->sets[si];
    size_t n = (size_t)(e->n > 0 ? e->n : 1);
    e->o00 = (long*)malloc(sizeof(long) * n); e->o11 = (long*)malloc(sizeof(long) * n); e->o01 = (long*)malloc(sizeof(long) * n);
    e->k00 = (char*)calloc(n, 1); e->k11 = (char*)calloc(n, 1); e->k01 = (char*)calloc(n, 1); e->tr01 = (char*)calloc(n, 1);
    for (int k = 0; k < e->n; ++k) {
      int a = e->v0[k], b = e->v1[k];
      e->o00[k] = e->o11[k] = e->o01[k] = -1;
      if (a >= 0) { if (a < nP) { e->k00[k] = 1; e->o00[k] = (long)s->pp_diag[a] * p * p; } else { e->k00[k] = 2; e->o00[k] = (long)(a - nP) * l * l; } }
      if (b >= 0) { if (b < nP) { e->k11[k] = 1; e->o11[k] = (long)s->pp_diag[b] * p * p; } else { e->k11[k] = 2; e->o11[k] = (long)(b - nP) * l * l; } }
      if (a >= 0 && b >= 0) {
        int ma = a >= nP, mb = b >= nP;
        if (!ma && !mb) {                         /* :221-229 */
          int tr = a > b; int r = tr ? b : a, c = tr ? a : b;
          e->k01[k] = 1; e->tr01[k] = (char)tr; e->o01[k] = (long)find_row(s->pp_colptr, s->pp_row, c, r) * p * p;
        } else {                                  /* :240-250 */
          int pose = ma ? b : a, lm = (ma ? a : b) - nP;
          e->k01[k] = 3; e->tr01[k] = (char)ma;   /* v0 marginalized -> transposed write */
          e->o01[k] = (long)find_row(s->pl_colptr, s->pl_row, lm, pose) * p * l;
        }
      }
    }
  }
  /* n-ary edges: mapHessianMemory(d, i, j, rowMajor) for every free pair, base_multi_edge.hpp:128-160 via block_solver.hpp:208-251 */
  for (int si = 0; si < s->nmsets; ++si) {
    OrcMulti* e = &s->msets[si];
    const int ar = e->arity;
    size_t n = (size_t)(e->n > 0 ? e->n : 1);
    e->odiag = (long*)malloc(sizeof(long) * n * ar); e->kdiag = (char*)calloc(n * ar, 1);
    e->ooff = (long*)malloc(sizeof(long) * n * ORC_MAX_PAIRS); e->koff = (char*)calloc(n * ORC_MAX_PAIRS, 1); e->troff = (char*)calloc(n * ORC_MAX_PAIRS, 1);
    for (int k = 0; k < e->n; ++k) {
      for (int i = 0; i < ar; ++i) {
        int a = e->v[(size_t)k * ar + i];
        e->odiag[(size_t)k * ar + i] = -1;
        if (a >= 0) { if (a < nP) { e->kdiag[(size_t)k * ar + i] = 1; e->odiag[(size_t)k * ar + i] = (long)s->pp_diag[a] * p * p; } else { e->kdiag[(size_t)k * ar + i] = 2; e->odiag[(size_t)k * ar + i] = (long)(a - nP) * l * l; } }
      }
      for (int i = 0; i < ar; ++i)
        for (int j = i + 1; j < ar; ++j) {
          const size_t q = (size_t)k * ORC_MAX_PAIRS + upper_triangle_index(i, j);
          int a = e->v[(size_t)k * ar + i], b = e->v[(size_t)k * ar + j];
          e->ooff[q] = -1;
          if (a < 0 || b < 0) continue;
          int ma = a >= nP, mb = b >= nP;
          if (!ma && !mb) {                         /* :221-229: transposedBlock = ind1 > ind2 */
            int tr = a > b; int r = tr ? b : a, c = tr ? a : b;
            e->koff[q] = 1; e->troff[q] = (char)tr; e->ooff[q] = (long)find_row(s->pp_colptr, s->pp_row, c, r) * p * p;
          } else {                                  /* :237-249: v1 marginalized -> the block is written transposed */
            int pose = ma ? b : a, lm = (ma ? a : b) - nP;
            e->koff[q] = 3; e->troff[q] = (char)ma;
            e->ooff[q] = (long)find_row(s->pl_colptr, s->pl_row, lm, pose) * p * l;
          }
        }
    }
  }
  if (!s->doSchur) return 0;
  /* Schur pattern: Hpp pattern U {(i1,i2): i1<=i2 co-observe a landmark}  :262-290 */
  long cnt = s->pp_nnzb;
  for (int c = 0; c < nL; ++c) { long k = s->pl_colptr[c + 1] - s->pl_colptr[c]; cnt += k * (k + 1) / 2; }
  cnt += s->n_extra;
  long long* ks = (long long*)malloc(sizeof(long long) * (size_t)(cnt + 1));
  long is = 0;
  for (int k = 0; k < s->n_extra; ++k) ks[is++] = (long long)s->extra_c[k] * nP + s->extra_r[k];
  for (int c = 0; c < nP; ++c) for (int q = s->pp_colptr[c]; q < s->pp_colptr[c + 1]; ++q) ks[is++] = (long long)c * nP + s->pp_row[q];
  for (int c = 0; c < nL; ++c)
    for (int q1 = s->pl_colptr[c]; q1 < s->pl_colptr[c + 1]; ++q1)
      for (int q2 = q1; q2 < s->pl_colptr[c + 1]; ++q2) ks[is++] = (long long)s->pl_row[q2] * nP + s->pl_row[q1];
  keys_to_ccs(ks, is, nP, nP, &s->hs_colptr, &s->hs_row, &s->hs_nnzb);
  free(ks);
  s->Hschur = (double*)calloc((size_t)s->hs_nnzb * p * p + 1, sizeof(double));
  return 0;
}

void orc_add_schur_pattern(Orc* s, int n, const int* rows, const int* cols) {
  s->extra_r = (int*)realloc(s->extra_r, sizeof(int) * (size_t)(s->n_extra + n + 1));
  s->extra_c = (int*)realloc(s->extra_c, sizeof(int) * (size_t)(s->n_extra + n + 1));
  memcpy(s->extra_r + s->n_extra, rows, sizeof(int) * (size_t)n);
  memcpy(s->extra_c + s->n_extra, cols, sizeof(int) * (size_t)n);
  s->n_extra += n;
}

void orc_set_edge_data(Orc* s, int set, const double* J0, const double* J1, const double* omega, const double* err, double huber_delta) {
  OrcSet* e = &s->sets[set];
  e->J0 = J0; e->J1 = J1; e->omega = omega; e->err = err; e->huber_delta = huber_delta;
}

static void robustify(int kind, double delta, double e, double* rho);
/* rho(e), rho'(e), rho''(e) of kernel `kind` (tests: consistency of the three, known values) */
void orc_robustify(int kind, double delta, double e, double* rho) { robustify(kind, delta, e, rho); }
/* other kernels than Huber for the set (the delta stays the one given to orc_set_edge_data) */
void orc_set_robust_kernel(Orc* s, int set, int kind) { s->sets[set].kernel_kind = kind; }

static double* blkptr(Orc* s, char kind, long off) {
  switch (kind) { case 1: return s->Hpp + off; case 2: return s->Hll + off; case 3: return s->Hpl + off; default: return NULL; }
}

/* The robust kernels of robust_kernel_impl.cpp, rho[0..2] = rho(e), rho'(e), rho''(e):
 * kind 1 Huber :65-78, 2 PseudoHuber :80-89, 3 Cauchy :91-99, 4 Saturated :101-113, 5 DCS :116-126 (delta = phi) */
static void robustify(int kind, double delta, double e, double* rho) {
  double dsqr = delta * delta, dsqrReci = 1. / dsqr;
  switch (kind) {
    case 1:
      if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
      else { double sqrte = sqrt(e); rho[0] = 2 * sqrte * delta - dsqr; rho[1] = delta / sqrte; rho[2] = -0.5 * rho[1] / e; }
      break;
    case 2: { double aux1 = dsqrReci * e + 1.0, aux2 = sqrt(aux1);
      rho[0] = 2 * dsqr * (aux2 - 1); rho[1] = 1. / aux2; rho[2] = -0.5 * dsqrReci * rho[1] / aux1; } break;
    case 3: { double aux = dsqrReci * e + 1.0;
      rho[0] = dsqr * log(aux); rho[1] = 1. / aux; rho[2] = -dsqrReci * rho[1] * rho[1]; } break;
    case 4:
      if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; } else { rho[0] = dsqr; rho[1] = 0.; rho[2] = 0.; }
      break;
    case 5: { double scale = (2.0 * delta) / (delta + e); if (scale >= 1.0) scale = 1.0;
      rho[0] = scale * e * scale; rho[1] = scale * scale; rho[2] = 0; } break;
    default: rho[0] = e; rho[1] = 1.; rho[2] = 0.;
  }
}
static double edge_chi2(const OrcSet* e, int k) {   /* base_edge.h:58-61: e' Omega e */
  const int d = e->d; const double* O = e->omega + (size_t)k * d * d; const double* r = e->err + (size_t)k * d;
  double c = 0;
  for (int i = 0; i < d; ++i) { double t = 0; for (int j = 0; j < d; ++j) t += O[i + d * j] * r[j]; c += r[i] * t; }
  return c;
}

/* y(dimA) += A' w ;  A is d x dimA column-major */
static void atx(const double* A, int d, int dimA, const double* w, double* y) {
  for (int c = 0; c < dimA; ++c) { double t = 0; for (int i = 0; i < d; ++i) t += A[i + d * c] * w[i]; y[c] += t; }
}
/* H(dimA x dimB, col-major, ld dimA) += A' O B */
static void atob(const double* A, int dimA, const double* O, const double* B, int dimB, int d, double* H) {
  double AtO[8 * 8];
  for (int a = 0; a < dimA; ++a) for (int j = 0; j < d; ++j) { double t = 0; for (int i = 0; i < d; ++i) t += A[i + d * a] * O[i + d * j]; AtO[a + dimA * j] = t; }
  for (int bcol = 0; bcol < dimB; ++bcol) for (int a = 0; a < dimA; ++a) {
    double t = 0; for (int j = 0; j < d; ++j) t += AtO[a + dimA * j] * B[j + d * bcol];
    H[a + dimA * bcol] += t;
  }
}

/* block_solver.hpp:501-560 with base_{binary,unary}_edge.hpp quadratic forms */
int orc_build_system(Orc* s) {
  const int p = s->p, l = s->l;
  memset(s->b, 0, sizeof(double) * (size_t)(s->sizeP + s->sizeL));          /* clearQuadraticForm :508-512 */
  memset(s->Hpp, 0, sizeof(double) * (size_t)s->pp_nnzb * p * p);            /* _Hpp->clear() */
  memset(s->Hll, 0, sizeof(double) * (size_t)s->nL * l * l);
  memset(s->Hpl, 0, sizeof(double) * (size_t)s->pl_nnzb * p * l);
  for (int si = 0; si < s->nsets; ++si) {
    OrcSet* e = &s->sets[si];
    const int d = e->d, d0 = e->dim0, d1 = e->dim1;
#ifdef _OPENMP
    ensure_locks((long)s->nP + s->nL);
#pragma omp parallel for default(shared) if (e->n > 100)
#endif
    for (int k = 0; k < e->n; ++k) {
      const int a = e->v0[k], bidx = e->v1[k];
      const int fromNotFixed = a >= 0, toNotFixed = (!e->unary) && bidx >= 0;
      if (!fromNotFixed && !toNotFixed) continue;
#ifdef _OPENMP
      /* from->lockQuadraticForm(); to->lockQuadraticForm();  (ascending order here: the reference locks in edge order) */
      { const int l0 = fromNotFixed ? a : -1, l1 = toNotFixed ? bidx : -1;
        const int lo = (l0 >= 0 && (l1 < 0 || l0 < l1)) ? l0 : l1, hi = (lo == l0) ? l1 : l0;
        if (lo >= 0) omp_set_lock(&g_locks[lo]);
        if (hi >= 0 && hi != lo) omp_set_lock(&g_locks[hi]); }
#endif
      const double* A = e->J0 + (size_t)k * d * d0;
      const double* B = e->unary ? NULL : e->J1 + (size_t)k * d * d1;
      const double* O = e->omega + (size_t)k * d * d;
      const double* r = e->err + (size_t)k * d;
      double omega_r[8], Ow[64];
      for (int i = 0; i < d; ++i) { double t = 0; for (int j = 0; j < d; ++j) t += O[i + d * j] * r[j]; omega_r[i] = -t; }
      const double* Ouse = O;
      if (e->huber_delta > 0) {              /* robust branch, base_binary_edge.hpp:92-112 */
        double rho[3]; robustify(e->kernel_kind > 0 ? e->kernel_kind : 1, e->huber_delta, edge_chi2(e, k), rho);
        for (int i = 0; i < d; ++i) omega_r[i] *= rho[1];
        for (int i = 0; i < d * d; ++i) Ow[i] = rho[1] * O[i];           /* base_edge.h:96-102 */
        Ouse = Ow;
      }
      if (fromNotFixed) {
        double* bv = s->b + (a < s->nP ? (size_t)a * p : (size_t)s->sizeP + (size_t)(a - s->nP) * l);
        atx(A, d, d0, omega_r, bv);
        atob(A, d0, Ouse, A, d0, d, blkptr(s, e->k00[k], e->o00[k]));
        if (toNotFixed) {
          double* H = blkptr(s, e->k01[k], e->o01[k]);
          if (e->tr01[k]) atob(B, d1, Ouse, A, d0, d, H); else atob(A, d0, Ouse, B, d1, d, H);
        }
      }
      if (toNotFixed) {
        double* bv = s->b + (bidx < s->nP ? (size_t)bidx * p : (size_t)s->sizeP + (size_t)(bidx - s->nP) * l);
        atx(B, d, d1, omega_r, bv);
        atob(B, d1, Ouse, B, d1, d, blkptr(s, e->k11[k], e->o11[k]));
      }
#ifdef _OPENMP
      if (toNotFixed) omp_unset_lock(&g_locks[bidx]);
      if (fromNotFixed && !(toNotFixed && a == bidx)) omp_unset_lock(&g_locks[a]);
#endif
    }
  }
  /* n-ary edges: BaseMultiEdge::constructQuadraticForm (base_multi_edge.hpp:35-49) + computeQuadraticForm (:170-222) */
  for (int si = 0; si < s->nmsets; ++si) {
    OrcMulti* e = &s->msets[si];
    const int d = e->d, ar = e->arity;
    for (int k = 0; k < e->n; ++k) {
      const double* O = e->omega + (size_t)k * d * d;
      const double* r = e->err + (size_t)k * d;
      double omega_r[8], Ow[64];
      for (int i = 0; i < d; ++i) { double t = 0; for (int j = 0; j < d; ++j) t += O[i + d * j] * r[j]; omega_r[i] = -t; }   /* - _information * _error */
      const double* Ouse = O;
      if (e->delta > 0) {                          /* :38-45 */
        double c = 0, rho[3];
        for (int i = 0; i < d; ++i) { double t = 0; for (int j = 0; j < d; ++j) t += O[i + d * j] * r[j]; c += r[i] * t; }
        robustify(e->kernel_kind > 0 ? e->kernel_kind : 1, e->delta, c, rho);
        for (int i = 0; i < d; ++i) omega_r[i] *= rho[1];
        for (int i = 0; i < d * d; ++i) Ow[i] = rho[1] * O[i];
        Ouse = Ow;
      }
      for (int i = 0; i < ar; ++i) {               /* :173-221 */
        const int a = e->v[(size_t)k * ar + i];
        if (a < 0) continue;                       /* istatus = !from->fixed() */
        const int da = vdim(s, a);
        const double* A = e->J[i] + (size_t)k * d * da;
        double* bv = s->b + (a < s->nP ? (size_t)a * p : (size_t)s->sizeP + (size_t)(a - s->nP) * l);
        atob(A, da, Ouse, A, da, d, blkptr(s, e->kdiag[(size_t)k * ar + i], e->odiag[(size_t)k * ar + i]));   /* fromMap += AtO * A */
        atx(A, d, da, omega_r, bv);                                                                           /* fromB += A' weightedError */
        for (int j = i + 1; j < ar; ++j) {
          const int bidx = e->v[(size_t)k * ar + j];
          if (bidx < 0) continue;                  /* jstatus */
          const int db = vdim(s, bidx);
          const double* B = e->J[j] + (size_t)k * d * db;
          const size_t q = (size_t)k * ORC_MAX_PAIRS + upper_triangle_index(i, j);
          double* H = blkptr(s, e->koff[q], e->ooff[q]);
          if (e->troff[q]) atob(B, db, Ouse, A, da, d, H);   /* hhelper.transposed: += B' AtO' */
          else atob(A, da, Ouse, B, db, d, H);               /* += AtO * B */
        }
      }
    }
  }
  return 0;
}

/* activeRobustChi2: sparse_optimizer.cpp:100-114 */
double orc_chi2(Orc* s) {
  double chi = 0;
  for (int si = 0; si < s->nsets; ++si) {
    OrcSet* e = &s->sets[si];
    for (int k = 0; k < e->n; ++k) {
      double c = edge_chi2(e, k);
      if (e->huber_delta > 0) { double rho[3]; robustify(e->kernel_kind > 0 ? e->kernel_kind : 1, e->huber_delta, c, rho); c = rho[0]; }
      chi += c;
    }
  }
  for (int si = 0; si < s->nmsets; ++si) {
    OrcMulti* e = &s->msets[si];
    const int d = e->d;
    for (int k = 0; k < e->n; ++k) {
      const double* O = e->omega + (size_t)k * d * d; const double* r = e->err + (size_t)k * d;
      double c = 0;
      for (int i = 0; i < d; ++i) { double t = 0; for (int j = 0; j < d; ++j) t += O[i + d * j] * r[j]; c += r[i] * t; }
      if (e->delta > 0) { double rho[3]; robustify(e->kernel_kind > 0 ? e->kernel_kind : 1, e->delta, c, rho); c = rho[0]; }
      chi += c;
    }
  }
  return chi;
}

/* block_solver.hpp:563-604 */
void orc_set_lambda(Orc* s, double lambda, int backup) {
  const int p = s->p, l = s->l;
  for (int i = 0; i < s->nP; ++i) { double* B = s->Hpp + (size_t)s->pp_diag[i] * p * p;
    for (int j = 0; j < p; ++j) { if (backup) s->bkP[i * p + j] = B[j + p * j]; B[j + p * j] += lambda; } }
  for (int i = 0; i < s->nL; ++i) { double* B = s->Hll + (size_t)i * l * l;
    for (int j = 0; j < l; ++j) { if (backup) s->bkL[i * l + j] = B[j + l * j]; B[j + l * j] += lambda; } }
}
void orc_set_lambda_split(Orc* s, double lambda_pose, double lambda_landmark, int backup) {
  const int p = s->p, l = s->l;
  for (int i = 0; i < s->nP; ++i) { double* B = s->Hpp + (size_t)s->pp_diag[i] * p * p;
    for (int j = 0; j < p; ++j) { if (backup) s->bkP[i * p + j] = B[j + p * j]; B[j + p * j] += lambda_pose; } }
  for (int i = 0; i < s->nL; ++i) { double* B = s->Hll + (size_t)i * l * l;
    for (int j = 0; j < l; ++j) { if (backup) s->bkL[i * l + j] = B[j + l * j]; B[j + l * j] += lambda_landmark; } }
}
void orc_restore_diagonal(Orc* s) {
  const int p = s->p, l = s->l;
  for (int i = 0; i < s->nP; ++i) { double* B = s->Hpp + (size_t)s->pp_diag[i] * p * p; for (int j = 0; j < p; ++j) B[j + p * j] = s->bkP[i * p + j]; }
  for (int i = 0; i < s->nL; ++i) { double* B = s->Hll + (size_t)i * l * l; for (int j = 0; j < l; ++j) B[j + l * j] = s->bkL[i * l + j]; }
}
/* computeLambdaInit: optimization_algorithm_levenberg.cpp:149-163 (tau applied by caller) */
double orc_max_diagonal(Orc* s) {
  const int p = s->p, l = s->l; double m = 0;
  for (int i = 0; i < s->nP; ++i) { const double* B = s->Hpp + (size_t)s->pp_diag[i] * p * p; for (int j = 0; j < p; ++j) if (fabs(B[j + p * j]) > m) m = fabs(B[j + p * j]); }
  for (int i = 0; i < s->nL; ++i) { const double* B = s->Hll + (size_t)i * l * l; for (int j = 0; j < l; ++j) if (fabs(B[j + l * j]) > m) m = fabs(B[j + l * j]); }
  return m;
}
/* computeScale: optimization_algorithm_levenberg.cpp:165-172 */
double orc_compute_scale(Orc* s, double lambda) {
  double sc = 0; const int n = s->sizeP + s->sizeL;
  for (int j = 0; j < n; ++j) sc += s->x[j] * (lambda * s->x[j] + s->b[j]);
  return sc;
}

/* ------------------------------------------------------- small dense inverse
 * Eigen's fixed-size inverse (D->inverse(), block_solver.hpp:389) is the closed
 * cofactor form for sizes <= 4; restated for 1,2,3; Gauss-Jordan beyond. */
static void small_inverse(const double* D, int n, double* R) {
  if (n == 1) { R[0] = 1.0 / D[0]; return; }
  if (n == 2) { double det = D[0] * D[3] - D[2] * D[1]; double id = 1.0 / det;
    R[0] = D[3] * id; R[1] = -D[1] * id; R[2] = -D[2] * id; R[3] = D[0] * id; return; }
  if (n == 3) {
#define M(i, j) D[(i) + 3 * (j)]
    double c00 = M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1);
    double c10 = M(1, 2) * M(2, 0) - M(1, 0) * M(2, 2);   /* cofactor(0,1) */
    double c20 = M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0);   /* cofactor(0,2) */
    double det = M(0, 0) * c00 + M(0, 1) * c10 + M(0, 2) * c20;
    double id = 1.0 / det;
    R[0 + 3 * 0] = c00 * id;
    R[1 + 3 * 0] = c10 * id;
    R[2 + 3 * 0] = c20 * id;
    R[0 + 3 * 1] = (M(0, 2) * M(2, 1) - M(0, 1) * M(2, 2)) * id;
    R[1 + 3 * 1] = (M(0, 0) * M(2, 2) - M(0, 2) * M(2, 0)) * id;
    R[2 + 3 * 1] = (M(0, 1) * M(2, 0) - M(0, 0) * M(2, 1)) * id;
    R[0 + 3 * 2] = (M(0, 1) * M(1, 2) - M(0, 2) * M(1, 1)) * id;
    R[1 + 3 * 2] = (M(0, 2) * M(1, 0) - M(0, 0) * M(1, 2)) * id;
    R[2 + 3 * 2] = (M(0, 0) * M(1, 1) - M(0, 1) * M(1, 0)) * id;
#undef M
    return;
  }
  double T[8 * 16];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { T[i + n * j] = D[i + n * j]; T[i + n * (n + j)] = (i == j); }
  for (int c = 0; c < n; ++c) {
    int piv = c; for (int i = c + 1; i < n; ++i) if (fabs(T[i + n * c]) > fabs(T[piv + n * c])) piv = i;
    if (piv != c) for (int j = 0; j < 2 * n; ++j) { double t = T[c + n * j]; T[c + n * j] = T[piv + n * j]; T[piv + n * j] = t; }
    double ip = 1.0 / T[c + n * c];
    for (int j = 0; j < 2 * n; ++j) T[c + n * j] *= ip;
    for (int i = 0; i < n; ++i) if (i != c) { double f = T[i + n * c]; if (f != 0) for (int j = 0; j < 2 * n; ++j) T[i + n * j] -= f * T[c + n * j]; }
  }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) R[i + n * j] = T[i + n * (n + j)];
}

/* ------------------------------------------------------------ linear solver */
/* fillCCS: sparse_block_matrix_ccs.h:143-199 (upper triangle; diagonal blocks contribute rows 0..c) */
static void chol_fill_ccs(OrcChol* C, int nb, int bs, const int* colptr, const int* row, const double* val, int structure) {
  const int n = nb * bs;
  if (structure) {
    long nz = 0;
    for (int c = 0; c < nb; ++c) for (int q = colptr[c]; q < colptr[c + 1]; ++q) nz += (row[q] == c) ? (long)bs * (bs + 1) / 2 : (long)bs * bs;
    C->n = n; C->nb = nb; C->bs = bs;
    free(C->Ap); free(C->Ai); free(C->Ax);
    C->Ap = (int*)malloc(sizeof(int) * ((size_t)n + 1)); C->Ai = (int*)malloc(sizeof(int) * (size_t)(nz + 1)); C->Ax = (double*)malloc(sizeof(double) * (size_t)(nz + 1));
  }
  int nz = 0;
  for (int bc = 0; bc < nb; ++bc) {
    for (int c = 0; c < bs; ++c) {
      if (structure) C->Ap[bc * bs + c] = nz;
      for (int q = colptr[bc]; q < colptr[bc + 1]; ++q) {
        const double* B = val + (size_t)q * bs * bs;
        int elems = bs; if (row[q] == bc) elems = c + 1;
        for (int r = 0; r < elems; ++r) { C->Ax[nz] = B[r + bs * c]; if (structure) C->Ai[nz] = row[q] * bs + r; ++nz; }
      }
    }
  }
  if (structure) C->Ap[n] = nz;
}

/* --- own exact minimum-degree on the block graph (stand-in for cs_amd where oracle/_ref is absent) */
typedef struct { int* a; int n, cap; } ivec;
static void iv_push(ivec* v, int x) { if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 8; v->a = (int*)realloc(v->a, sizeof(int) * (size_t)v->cap); } v->a[v->n++] = x; }
typedef struct { int deg, v; } hent;
static void heap_push(hent** h, int* n, int* cap, hent e) {
  if (*n == *cap) { *cap = *cap ? 2 * *cap : 1024; *h = (hent*)realloc(*h, sizeof(hent) * (size_t)*cap); }
  int i = (*n)++; (*h)[i] = e;
  while (i > 0) { int pa = (i - 1) / 2; hent* H = *h; if (H[pa].deg < H[i].deg || (H[pa].deg == H[i].deg && H[pa].v <= H[i].v)) break; hent t = H[pa]; H[pa] = H[i]; H[i] = t; i = pa; }
}
static hent heap_pop(hent* H, int* n) {
  hent top = H[0]; H[0] = H[--(*n)]; int i = 0;
  for (;;) { int l = 2 * i + 1, r = l + 1, m = i;
    if (l < *n && (H[l].deg < H[m].deg || (H[l].deg == H[m].deg && H[l].v < H[m].v))) m = l;
    if (r < *n && (H[r].deg < H[m].deg || (H[r].deg == H[m].deg && H[r].v < H[m].v))) m = r;
    if (m == i) break;
    hent t = H[m]; H[m] = H[i]; H[i] = t; i = m; }
  return top;
}
static int cmp_int(const void* a, const void* b) { return (*(const int*)a > *(const int*)b) - (*(const int*)a < *(const int*)b); }
static void block_min_degree(int nb, const int* colptr, const int* row, int* perm) {
  ivec* adj = (ivec*)calloc((size_t)nb, sizeof(ivec));
  for (int c = 0; c < nb; ++c) for (int q = colptr[c]; q < colptr[c + 1]; ++q) { int r = row[q]; if (r != c) { iv_push(&adj[c], r); iv_push(&adj[r], c); } }
  for (int v = 0; v < nb; ++v) { qsort(adj[v].a, (size_t)adj[v].n, sizeof(int), cmp_int); int u = 0; for (int i = 0; i < adj[v].n; ++i) if (i == 0 || adj[v].a[i] != adj[v].a[i - 1]) adj[v].a[u++] = adj[v].a[i]; adj[v].n = u; }
  char* done = (char*)calloc((size_t)nb, 1);
  hent* H = NULL; int hn = 0, hc = 0;
  for (int v = 0; v < nb; ++v) { hent e = {adj[v].n, v}; heap_push(&H, &hn, &hc, e); }
  int* tmp = (int*)malloc(sizeof(int) * (size_t)(nb + 1));
  int k = 0;
  while (hn > 0) {
    hent e = heap_pop(H, &hn);
    if (done[e.v] || e.deg != adj[e.v].n) continue;
    int v = e.v; done[v] = 1; perm[k++] = v;
    ivec nb_v = adj[v];
    for (int i = 0; i < nb_v.n; ++i) {
      int u = nb_v.a[i];
      /* adj[u] = (adj[u] U adj[v]) \ {u, v}  -- sorted merge */
      int a = 0, b = 0, t = 0; ivec* au = &adj[u];
      while (a < au->n || b < nb_v.n) {
        int x;
        if (b >= nb_v.n || (a < au->n && au->a[a] < nb_v.a[b])) x = au->a[a++];
        else if (a >= au->n || nb_v.a[b] < au->a[a]) x = nb_v.a[b++];
        else { x = au->a[a]; ++a; ++b; }
        if (x != u && x != v) tmp[t++] = x;
      }
      if (t > au->cap) { au->cap = t + 8; au->a = (int*)realloc(au->a, sizeof(int) * (size_t)au->cap); }
      memcpy(au->a, tmp, sizeof(int) * (size_t)t); au->n = t;
      hent ne = {t, u}; heap_push(&H, &hn, &hc, ne);
    }
    free(adj[v].a); adj[v].a = NULL; adj[v].n = adj[v].cap = 0;
  }
  for (int v = 0; v < nb; ++v) free(adj[v].a);
  free(adj); free(done); free(H); free(tmp);
}

/* symbolic: C = P A P' (upper), elimination tree, column counts by row-pattern walks.
 * Follows the flow of linear_solver_csparse.h:283-298 with own code for
 * cs_symperm / cs_etree / counts (counts are obtained by running the cs_ereach
 * pattern walk for every row instead of cs_counts' skeleton algorithm: same result). */
static int chol_symbolic(OrcChol* C, const int* scalar_perm) {
  const int n = C->n; const int* Ap = C->Ap; const int* Ai = C->Ai;
  free(C->perm); free(C->pinv); free(C->Cp); free(C->Ci); free(C->Cx); free(C->Cmap); free(C->parent); free(C->Lp); free(C->Li); free(C->Lx); free(C->iw); free(C->xw);
  C->perm = (int*)malloc(sizeof(int) * (size_t)n); C->pinv = (int*)malloc(sizeof(int) * (size_t)n);
  for (int k = 0; k < n; ++k) { C->perm[k] = scalar_perm ? scalar_perm[k] : k; C->pinv[C->perm[k]] = k; }
  const int nz = Ap[n];
  C->Cp = (int*)calloc((size_t)n + 1, sizeof(int)); C->Ci = (int*)malloc(sizeof(int) * (size_t)(nz + 1)); C->Cx = (double*)malloc(sizeof(double) * (size_t)(nz + 1)); C->Cmap = (int*)malloc(sizeof(int) * (size_t)(nz + 1));
  int* w = (int*)calloc((size_t)n + 1, sizeof(int));
  for (int j = 0; j < n; ++j) { int j2 = C->pinv[j]; for (int q = Ap[j]; q < Ap[j + 1]; ++q) { int i = Ai[q]; if (i > j) continue; int i2 = C->pinv[i]; w[i2 > j2 ? i2 : j2]++; } }
  { int sum = 0; for (int j = 0; j < n; ++j) { C->Cp[j] = sum; sum += w[j]; w[j] = C->Cp[j]; } C->Cp[n] = sum; }
  for (int j = 0; j < n; ++j) { int j2 = C->pinv[j]; for (int q = Ap[j]; q < Ap[j + 1]; ++q) { int i = Ai[q]; if (i > j) continue; int i2 = C->pinv[i];
      int dst = w[i2 > j2 ? i2 : j2]++; C->Ci[dst] = i2 < j2 ? i2 : j2; C->Cmap[dst] = q; } }
  /* etree (Liu) with path compression -- EXTERNAL/csparse/cs_etree.c semantics */
  C->parent = (int*)malloc(sizeof(int) * (size_t)n);
  int* anc = w;
  for (int k = 0; k < n; ++k) {
    C->parent[k] = -1; anc[k] = -1;
    for (int q = C->Cp[k]; q < C->Cp[k + 1]; ++q) {
      int i = C->Ci[q];
      while (i != -1 && i < k) { int inext = anc[i]; anc[i] = k; if (inext == -1) C->parent[i] = k; i = inext; }
    }
  }
  /* column counts */
  int* cnt = (int*)calloc((size_t)n + 1, sizeof(int));
  int* mark = (int*)malloc(sizeof(int) * (size_t)n);
  for (int k = 0; k < n; ++k) mark[k] = -1;
  for (int k = 0; k < n; ++k) {
    mark[k] = k; cnt[k]++;
    for (int q = C->Cp[k]; q < C->Cp[k + 1]; ++q) { int i = C->Ci[q]; if (i > k) continue; for (; mark[i] != k; i = C->parent[i]) { cnt[i]++; mark[i] = k; } }
  }
  C->Lp = (int*)malloc(sizeof(int) * ((size_t)n + 1));
  { long sum = 0; for (int k = 0; k < n; ++k) { C->Lp[k] = (int)sum; sum += cnt[k]; } C->Lp[n] = (int)sum; C->lnz = (double)sum;
    if (sum > 2000000000L) { free(cnt); free(mark); free(w); return -1; } }
  C->Li = (int*)malloc(sizeof(int) * (size_t)(C->Lp[n] + 1)); C->Lx = (double*)malloc(sizeof(double) * (size_t)(C->Lp[n] + 1));
  C->iw = (int*)malloc(sizeof(int) * 3 * (size_t)(n + 1)); C->xw = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  free(cnt); free(mark); free(w);
  C->have_symbolic = 1;
  return 0;
}

/* numeric up-looking Cholesky + solve: csparse_helper.cpp:56-143 (same loop structure and
 * operation order as cs_chol_workspace / cs_ereach / cs_lsolve / cs_ltsolve).  b in, x out. */
static int chol_numeric_solve(OrcChol* C, const double* b, double* xout) {
  const int n = C->n; const int *Cp = C->Cp, *Ci = C->Ci, *parent = C->parent; int *Lp = C->Lp, *Li = C->Li; double* Lx = C->Lx;
  for (int q = 0; q < Cp[n]; ++q) C->Cx[q] = C->Ax[C->Cmap[q]];
  const double* Cx = C->Cx;
  int* c = C->iw; int* s = C->iw + n; int* mark = C->iw + 2 * n; double* x = C->xw;
  for (int k = 0; k < n; ++k) { c[k] = Lp[k]; mark[k] = -1; x[k] = 0; }
  for (int k = 0; k < n; ++k) {
    int top = n; mark[k] = k;
    for (int q = Cp[k]; q < Cp[k + 1]; ++q) {           /* cs_ereach.c:3-23 */
      int i = Ci[q]; if (i > k) continue;
      int len = 0;
      for (; mark[i] != k; i = parent[i]) { s[len++] = i; mark[i] = k; }
      while (len > 0) s[--top] = s[--len];
    }
    x[k] = 0;
    for (int q = Cp[k]; q < Cp[k + 1]; ++q) if (Ci[q] <= k) x[Ci[q]] = Cx[q];
    double d = x[k]; x[k] = 0;
    for (; top < n; ++top) {
      int i = s[top];
      double lki = x[i] / Lx[Lp[i]];
      x[i] = 0;
      for (int q = Lp[i] + 1; q < c[i]; ++q) x[Li[q]] -= Lx[q] * lki;
      d -= lki * lki;
      int q = c[i]++; Li[q] = k; Lx[q] = lki;
    }
    if (d <= 0) return 0;                                 /* not positive definite, :136 */
    int q = c[k]++; Li[q] = k; Lx[q] = sqrt(d);
  }
  /* x = P b ; L\ ; L'\ ; P' x */
  for (int k = 0; k < n; ++k) x[C->pinv[k]] = b[k];       /* cs_ipvec */
  for (int j = 0; j < n; ++j) { x[j] /= Lx[Lp[j]]; for (int q = Lp[j] + 1; q < Lp[j + 1]; ++q) x[Li[q]] -= Lx[q] * x[j]; }   /* cs_lsolve */
  for (int j = n - 1; j >= 0; --j) { for (int q = Lp[j] + 1; q < Lp[j + 1]; ++q) x[j] -= Lx[q] * x[Li[q]]; x[j] /= Lx[Lp[j]]; } /* cs_ltsolve */
  for (int k = 0; k < n; ++k) xout[k] = x[C->pinv[k]];    /* cs_pvec */
  return 1;
}

void orc_set_ordering(Orc* s, int mode, const int* block_perm) {
  s->ordering = mode;
  free(s->ext_block_perm); s->ext_block_perm = NULL;
  if (mode == 2 && block_perm) { s->ext_block_perm = (int*)malloc(sizeof(int) * (size_t)s->nP); memcpy(s->ext_block_perm, block_perm, sizeof(int) * (size_t)s->nP); }
  s->chol.have_symbolic = 0;
}

/* LinearSolverCSparse::solve, linear_solver_csparse.h:106-142 */
static int linear_solve(Orc* s, int nb, int bs, const int* colptr, const int* row, const double* val, double* x, const double* b) {
  OrcChol* C = &s->chol;
  chol_fill_ccs(C, nb, bs, colptr, row, val, !C->have_symbolic);
  if (!C->have_symbolic) {
    int* bperm = (int*)malloc(sizeof(int) * (size_t)(nb + 1));
    if (s->ordering == 1) block_min_degree(nb, colptr, row, bperm);
    else if (s->ordering == 2 && s->ext_block_perm) memcpy(bperm, s->ext_block_perm, sizeof(int) * (size_t)nb);
    else for (int i = 0; i < nb; ++i) bperm[i] = i;
    int* sperm = (int*)malloc(sizeof(int) * (size_t)(nb * bs + 1));
    int k = 0; for (int i = 0; i < nb; ++i) for (int j = 0; j < bs; ++j) sperm[k++] = bperm[i] * bs + j;   /* :267-281 */
    int rc = chol_symbolic(C, sperm);
    free(bperm); free(sperm);
    if (rc) return 0;
  }
  double t = now_s();
  int ok = chol_numeric_solve(C, b, x);
  s->t_numeric = now_s() - t;
  return ok;
}

/* standalone narrow-seam solve for KAT tests: block CCS upper + b -> x.  perm (block) may be NULL (natural) */
int orc_linear_solve_blocks(int nb, int bs, const int* colptr, const int* row, const double* val, const int* block_perm, int use_min_degree, const double* b, double* x, double* lnz_out) {
  Orc tmp; memset(&tmp, 0, sizeof(tmp)); tmp.nP = nb;
  tmp.ordering = block_perm ? 2 : (use_min_degree ? 1 : 0);
  if (block_perm) { tmp.ext_block_perm = (int*)malloc(sizeof(int) * (size_t)nb); memcpy(tmp.ext_block_perm, block_perm, sizeof(int) * (size_t)nb); }
  int ok = linear_solve(&tmp, nb, bs, colptr, row, val, x, b);
  if (lnz_out) *lnz_out = tmp.chol.lnz;
  OrcChol* C = &tmp.chol;
  free(C->Ap); free(C->Ai); free(C->Ax); free(C->perm); free(C->pinv); free(C->Cp); free(C->Ci); free(C->Cx); free(C->Cmap); free(C->parent); free(C->Lp); free(C->Li); free(C->Lx); free(C->iw); free(C->xw);
  free(tmp.ext_block_perm);
  return ok;
}
/* scalar CCS export of a block matrix for feeding oracle/_ref (returns nnz; arrays sized by caller via count call) */
long orc_fill_scalar_ccs(int nb, int bs, const int* colptr, const int* row, const double* val, int* Ap, int* Ai, double* Ax) {
  OrcChol C; memset(&C, 0, sizeof(C));
  chol_fill_ccs(&C, nb, bs, colptr, row, val, 1);
  long nz = C.Ap[nb * bs];
  if (Ap) { memcpy(Ap, C.Ap, sizeof(int) * ((size_t)nb * bs + 1)); memcpy(Ai, C.Ai, sizeof(int) * (size_t)nz); memcpy(Ax, C.Ax, sizeof(double) * (size_t)nz); }
  free(C.Ap); free(C.Ai); free(C.Ax);
  return nz;
}

/* ----------------------------------------------------------------- solve() */
/* block_solver.hpp:367-444: Schur complement (K5-K8) */
void orc_solve_schur(Orc* s) {
  const int p = s->p, l = s->l, nP = s->nP, nL = s->nL;
  double t = now_s();
  /* _Hschur = _Hpp keeping the pattern of _Hschur  :373-374 */
  memset(s->Hschur, 0, sizeof(double) * (size_t)s->hs_nnzb * p * p);
  for (int c = 0; c < nP; ++c) for (int q = s->pp_colptr[c]; q < s->pp_colptr[c + 1]; ++q) {
    int dq = find_row(s->hs_colptr, s->hs_row, c, s->pp_row[q]);
    double* dst = s->Hschur + (size_t)dq * p * p; const double* src = s->Hpp + (size_t)q * p * p;
    for (int i = 0; i < p * p; ++i) dst[i] += src[i];
  }
  memset(s->coeff, 0, sizeof(double) * (size_t)s->sizeP);
#ifdef _OPENMP
  ensure_locks((long)s->nP + s->nL);
#pragma omp parallel for default(shared) schedule(dynamic, 10)
#endif
  for (int lm = 0; lm < nL; ++lm) {
    const double* D = s->Hll + (size_t)lm * l * l;
    double* Dinv = s->Dinv + (size_t)lm * l * l;
    small_inverse(D, l, Dinv);                                     /* :389 */
    double db0[8], db[8];
    for (int j = 0; j < l; ++j) db0[j] = s->b[s->sizeP + lm * l + j];
    for (int i = 0; i < l; ++i) { double tt = 0; for (int j = 0; j < l; ++j) tt += Dinv[i + l * j] * db0[j]; db[i] = tt; }
    for (int q1 = s->pl_colptr[lm]; q1 < s->pl_colptr[lm + 1]; ++q1) {
      const int i1 = s->pl_row[q1];
      const double* Bi = s->Hpl + (size_t)q1 * p * l;
      double BDinv[8 * 8];
      for (int r = 0; r < p; ++r) for (int cc = 0; cc < l; ++cc) { double tt = 0; for (int k = 0; k < l; ++k) tt += Bi[r + p * k] * Dinv[k + l * cc]; BDinv[r + p * cc] = tt; }
#ifdef _OPENMP
      omp_set_lock(&g_locks[i1]);                                   /* ScopedOpenMPMutex mutexLock(&_coefficientsMutex[i1]) :411 */
#endif
      for (int r = 0; r < p; ++r) { double tt = 0; for (int k = 0; k < l; ++k) tt += Bi[r + p * k] * db[k]; s->coeff[i1 * p + r] += tt; }   /* :412 */
      for (int q2 = q1; q2 < s->pl_colptr[lm + 1]; ++q2) {          /* lower_bound start at i2 >= i1, :418-419 */
        const int i2 = s->pl_row[q2];
        const double* Bj = s->Hpl + (size_t)q2 * p * l;
        double* H = s->Hschur + (size_t)find_row(s->hs_colptr, s->hs_row, i2, i1) * p * p;
        for (int r = 0; r < p; ++r) for (int cc = 0; cc < p; ++cc) { double tt = 0; for (int k = 0; k < l; ++k) tt += BDinv[r + p * k] * Bj[cc + p * k]; H[r + p * cc] -= tt; }  /* :430 */
      }
#ifdef _OPENMP
      omp_unset_lock(&g_locks[i1]);                                 /* (scope of the mutex: the body of the outer loop) */
#endif
    }
  }
  memcpy(s->bschur, s->b, sizeof(double) * (size_t)s->sizeP);       /* :435-439 */
  for (int i = 0; i < s->sizeP; ++i) s->bschur[i] -= s->coeff[i];
  s->t_schur = now_s() - t;
}
/* :446-457 linear solve of the reduced system */
int orc_solve_reduced(Orc* s) {
  double t = now_s();
  int ok = linear_solve(s, s->nP, s->p, s->hs_colptr, s->hs_row, s->Hschur, s->x, s->bschur);
  s->t_linear = now_s() - t;
  return ok;
}
/* landmark back-substitution :459-483 */
void orc_solve_back_substitute(Orc* s) {
  const int p = s->p, l = s->l, nL = s->nL;
  double* xp = s->x; double* cp = s->coeff; double* xl = s->x + s->sizeP; double* cl = s->coeff + s->sizeP; const double* bl = s->b + s->sizeP;
  for (int i = 0; i < s->sizeP; ++i) cp[i] = -xp[i];
  memcpy(cl, bl, sizeof(double) * (size_t)s->sizeL);
  for (int lm = 0; lm < nL; ++lm)                                   /* rightMultiply: sparse_block_matrix_ccs.h:103-128 */
    for (int q = s->pl_colptr[lm]; q < s->pl_colptr[lm + 1]; ++q) {
      const double* B = s->Hpl + (size_t)q * p * l; const double* src = cp + (size_t)s->pl_row[q] * p;
      for (int cc = 0; cc < l; ++cc) { double tt = 0; for (int r = 0; r < p; ++r) tt += B[r + p * cc] * src[r]; cl[lm * l + cc] += tt; }
    }
  memset(xl, 0, sizeof(double) * (size_t)s->sizeL);
  for (int lm = 0; lm < nL; ++lm) {                                 /* sparse_block_matrix_diagonal.h:77-99 */
    const double* Dinv = s->Dinv + (size_t)lm * l * l;
    for (int i = 0; i < l; ++i) { double tt = 0; for (int j = 0; j < l; ++j) tt += Dinv[i + l * j] * cl[lm * l + j]; xl[lm * l + i] += tt; }
  }
}

/* block_solver.hpp:353-486.  Returns 1 ok, 0 not positive definite. */
int orc_solve(Orc* s) {
  if (!s->doSchur) {
    double t = now_s();
    int ok = linear_solve(s, s->nP, s->p, s->pp_colptr, s->pp_row, s->Hpp, s->x, s->b);
    s->t_linear = now_s() - t; s->t_schur = 0;
    return ok;
  }
  orc_solve_schur(s);
  if (!orc_solve_reduced(s)) return 0;
  orc_solve_back_substitute(s);
  return 1;
}

/* dest += H * src with H = [Hpp Hpl; Hpl' Hll] (Hpp upper-stored symmetric):
 * BlockSolverBase::multiplyHessian block_solver.h:83-91 / sparse_block_matrix.hpp:257-282,
 * extended to the full system for residual checks. */
void orc_multiply_full(Orc* s, const double* src, double* dest) {
  const int p = s->p, l = s->l, nP = s->nP, nL = s->nL;
  for (int c = 0; c < nP; ++c) for (int q = s->pp_colptr[c]; q < s->pp_colptr[c + 1]; ++q) {
    int r = s->pp_row[q]; const double* B = s->Hpp + (size_t)q * p * p;
    for (int i = 0; i < p; ++i) for (int j = 0; j < p; ++j) {
      dest[r * p + i] += B[i + p * j] * src[c * p + j];
      if (r != c) dest[c * p + j] += B[i + p * j] * src[r * p + i];
    }
  }
  for (int lm = 0; lm < nL; ++lm) {
    for (int q = s->pl_colptr[lm]; q < s->pl_colptr[lm + 1]; ++q) {
      int r = s->pl_row[q]; const double* B = s->Hpl + (size_t)q * p * l;
      for (int i = 0; i < p; ++i) for (int j = 0; j < l; ++j) {
        dest[r * p + i] += B[i + p * j] * src[s->sizeP + lm * l + j];
        dest[s->sizeP + lm * l + j] += B[i + p * j] * src[r * p + i];
      }
    }
    const double* D = s->Hll + (size_t)lm * l * l;
    for (int i = 0; i < l; ++i) for (int j = 0; j < l; ++j) dest[s->sizeP + lm * l + i] += D[i + l * j] * src[s->sizeP + lm * l + j];
  }
}

/* --------------------------------------------------------------- accessors */
double* orc_x(Orc* s) { return s->x; }
double* orc_b(Orc* s) { return s->b; }
double* orc_bschur(Orc* s) { return s->bschur; }
double* orc_Hpp(Orc* s) { return s->Hpp; }
double* orc_Hpl(Orc* s) { return s->Hpl; }
double* orc_Hll(Orc* s) { return s->Hll; }
double* orc_Hschur(Orc* s) { return s->Hschur; }
double* orc_Dinv(Orc* s) { return s->Dinv; }
int orc_pp_nnzb(Orc* s) { return s->pp_nnzb; }
int orc_pl_nnzb(Orc* s) { return s->pl_nnzb; }
int orc_hs_nnzb(Orc* s) { return s->hs_nnzb; }
const int* orc_pp_colptr(Orc* s) { return s->pp_colptr; }
const int* orc_pp_row(Orc* s) { return s->pp_row; }
const int* orc_pl_colptr(Orc* s) { return s->pl_colptr; }
const int* orc_pl_row(Orc* s) { return s->pl_row; }
const int* orc_hs_colptr(Orc* s) { return s->hs_colptr; }
const int* orc_hs_row(Orc* s) { return s->hs_row; }
double orc_lnz(Orc* s) { return s->chol.lnz; }
double orc_time(Orc* s, int which) { return which == 0 ? s->t_schur : which == 1 ? s->t_linear : s->t_numeric; }

void orc_destroy(Orc* s) {
  if (!s) return;
  for (int si = 0; si < s->nsets; ++si) { OrcSet* e = &s->sets[si]; free(e->v0); free(e->v1); free(e->o00); free(e->o11); free(e->o01); free(e->k00); free(e->k11); free(e->k01); free(e->tr01); }
  for (int si = 0; si < s->nmsets; ++si) { OrcMulti* e = &s->msets[si]; free(e->v); free(e->odiag); free(e->kdiag); free(e->ooff); free(e->koff); free(e->troff); }
  free(s->pp_colptr); free(s->pp_row); free(s->pp_diag); free(s->Hpp); free(s->pl_colptr); free(s->pl_row); free(s->Hpl); free(s->Hll);
  free(s->hs_colptr); free(s->hs_row); free(s->Hschur); free(s->Dinv); free(s->coeff); free(s->bschur); free(s->x); free(s->b); free(s->bkP); free(s->bkL);
  OrcChol* C = &s->chol;
  free(C->Ap); free(C->Ai); free(C->Ax); free(C->perm); free(C->pinv); free(C->Cp); free(C->Ci); free(C->Cx); free(C->Cmap); free(C->parent); free(C->Lp); free(C->Li); free(C->Lx); free(C->iw); free(C->xw);
  free(s->ext_block_perm); free(s->extra_r); free(s->extra_c);
  free(s);
}

/* ========================================================================
 * LinearSolverPCG<MatrixType>::solve, g2o/solvers/pcg/linear_solver_pcg.hpp:79-196 (+ defaults
 * linear_solver_pcg.h:53-57): block-Jacobi preconditioned CG on the upper block-CCS matrix.
 *   J_i = A_ii^-1 (:96, Eigen inverse(); here Gauss-Jordan with partial pivoting),
 *   x = 0, r = b, d = J r, dn = r'd, d0 = tol*dn (raised to the previous call's residual when absolute, :127-131),
 *   loop (:135-151): stop when dn <= d0; q = A d (mult :192-213: diagonal blocks, then every upper block and its
 *   transpose); a = dn/d'q; x += a d; r -= a q; s = J r; ba = r's / dn; d = s + ba d.
 *   residual_inout: in = _residual of the previous solve (<= 0: none), out = 0.5 * dn (:153).
 * Test infrastructure only.
 * ======================================================================== */
static int small_inverse_gj(int n, const double* A /* col-major */, double* R) {
  double M[16 * 32];
  if (n > 16) return 0;
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) { M[r * 2 * n + c] = A[r + n * c]; M[r * 2 * n + n + c] = (r == c); }
  for (int c = 0; c < n; ++c) {
    int p = c; double best = fabs(M[c * 2 * n + c]);
    for (int r = c + 1; r < n; ++r) if (fabs(M[r * 2 * n + c]) > best) { best = fabs(M[r * 2 * n + c]); p = r; }
    if (best == 0.0) return 0;
    if (p != c) for (int k = 0; k < 2 * n; ++k) { double t = M[c * 2 * n + k]; M[c * 2 * n + k] = M[p * 2 * n + k]; M[p * 2 * n + k] = t; }
    double inv = 1.0 / M[c * 2 * n + c];
    for (int k = 0; k < 2 * n; ++k) M[c * 2 * n + k] *= inv;
    for (int r = 0; r < n; ++r) if (r != c) { double f = M[r * 2 * n + c]; if (f != 0.0) for (int k = 0; k < 2 * n; ++k) M[r * 2 * n + k] -= f * M[c * 2 * n + k]; }
  }
  for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) R[r + n * c] = M[r * 2 * n + n + c];
  return 1;
}

int orc_pcg_solve_blocks(int nb, int bs, const int* colptr, const int* row, const double* val, const double* b, double* x,
                         double tolerance, int absolute, int max_iter, double* residual_inout, int* iterations) {
  const int n = nb * bs, bb = bs * bs;
  double* J = (double*)malloc(sizeof(double) * (size_t)nb * bb);
  int* diag = (int*)malloc(sizeof(int) * (size_t)nb);
  double* r = (double*)malloc(sizeof(double) * (size_t)n * 4);
  double *d = r + n, *q = d + n, *s = q + n;
  int ok = 1;
  for (int c = 0; c < nb && ok; ++c) {
    diag[c] = -1;
    for (int k = colptr[c]; k < colptr[c + 1]; ++k) if (row[k] == c) diag[c] = k;
    if (diag[c] < 0 || !small_inverse_gj(bs, val + (size_t)diag[c] * bb, J + (size_t)c * bb)) ok = 0;
  }
  if (!ok) { free(J); free(diag); free(r); return 0; }
#define MULT_DIAG(M_, idx_, src_, dst_) \
  for (int i_ = 0; i_ < nb; ++i_) { const double* B_ = (M_) + (size_t)(idx_) * bb; \
    for (int rr = 0; rr < bs; ++rr) { double t_ = 0; for (int cc = 0; cc < bs; ++cc) t_ += B_[rr + bs * cc] * (src_)[i_ * bs + cc]; (dst_)[i_ * bs + rr] = t_; } }
  for (int i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; }
  MULT_DIAG(J, i_, r, d)
  double dn = 0; for (int i = 0; i < n; ++i) dn += r[i] * d[i];
  double d0 = tolerance * dn;
  if (absolute && residual_inout && *residual_inout > 0.0 && *residual_inout > d0) d0 = *residual_inout;
  int maxit = max_iter < 0 ? n : max_iter, it;
  for (it = 0; it < maxit; ++it) {
    if (dn <= d0) break;
    MULT_DIAG(val, diag[i_], d, q)
    for (int c = 0; c < nb; ++c)
      for (int k = colptr[c]; k < colptr[c + 1]; ++k) {
        const int rw = row[k]; if (rw == c) continue;
        const double* B = val + (size_t)k * bb;
        for (int rr = 0; rr < bs; ++rr) { double t = 0; for (int cc = 0; cc < bs; ++cc) t += B[rr + bs * cc] * d[c * bs + cc]; q[rw * bs + rr] += t; }
        for (int cc = 0; cc < bs; ++cc) { double t = 0; for (int rr = 0; rr < bs; ++rr) t += B[rr + bs * cc] * d[rw * bs + rr]; q[c * bs + cc] += t; }
      }
    double dq = 0; for (int i = 0; i < n; ++i) dq += d[i] * q[i];
    const double a = dn / dq;
    for (int i = 0; i < n; ++i) { x[i] += a * d[i]; r[i] -= a * q[i]; }
    MULT_DIAG(J, i_, r, s)
    const double dold = dn;
    dn = 0; for (int i = 0; i < n; ++i) dn += r[i] * s[i];
    const double ba = dn / dold;
    for (int i = 0; i < n; ++i) d[i] = s[i] + ba * d[i];
  }
#undef MULT_DIAG
  if (residual_inout) *residual_inout = 0.5 * dn;
  if (iterations) *iterations = it;
  free(J); free(diag); free(r);
  return 1;
}
