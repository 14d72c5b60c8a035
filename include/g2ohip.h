/* g2ohip.h -- C ABI of the MI355X-native sparse block solver for g2o's GN/LM inner loop.
 *
 * Drop-in boundary for g2o's BlockSolver<p,l> path.  Two seams:
 *
 *   wide seam   g2o::Solver            /root/reference/g2o/core/solver.h:44-149
 *               g2o::BlockSolver<T>    /root/reference/g2o/core/block_solver.h:98-178
 *   narrow seam g2o::LinearSolver<M>   /root/reference/g2o/core/linear_solver.h:40-81
 *
 * Every entry point below names the reference member it replaces.  Plain C, plain
 * pointers and sizes, no C++/Eigen/torch types.  All matrices are fp64, all indices int32
 * (the reference's hessianIndex / csi=int, EXTERNAL/csparse/cs.h:26).  Dense blocks are
 * column-major like Eigen's default (config.h.in:16-20).
 *
 * Threading: a handle is not thread-safe; all calls for one handle must come from one
 * thread (the thread that calls SparseOptimizer::optimize()).  Errors are
 * reported by return code (the reference path uses no exceptions); g2ohip_last_error()
 * gives the text.  The library never falls back to a CPU path: without a usable HIP
 * device every compute call returns G2OHIP_ERR_HIP.
 */
#ifndef G2OHIP_H
#define G2OHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2OHIP_OK 0
#define G2OHIP_NOT_PD 1            /* Cholesky hit a pivot <= 0: solve() == false (csparse_helper.cpp:136) */
#define G2OHIP_REPEAT 2            /* phased sharded solve only (g2ohip_exchange_status): a rank's dependency-driven launch gave
                                      up waiting; run the phases of this solve again on EVERY rank (g2ohip_solve_sharded does) */
#define G2OHIP_ERR_ARG (-1)
#define G2OHIP_ERR_HIP (-2)
#define G2OHIP_ERR_STATE (-3)
#define G2OHIP_ERR_UNSUPPORTED (-4)

#define G2OHIP_KERNEL_NONE 0
#define G2OHIP_KERNEL_HUBER 1      /* RobustKernelHuber, g2o/core/robust_kernel_impl.cpp:65-78 */
#define G2OHIP_KERNEL_PSEUDOHUBER 2 /* RobustKernelPseudoHuber :80-89 */
#define G2OHIP_KERNEL_CAUCHY 3     /* RobustKernelCauchy :91-99 */
#define G2OHIP_KERNEL_SATURATED 4  /* RobustKernelSaturated :101-113 */
#define G2OHIP_KERNEL_DCS 5        /* RobustKernelDCS :116-126 (delta = phi) */

/* which-matrix selectors for the inspection calls */
#define G2OHIP_HPP 0
#define G2OHIP_HPL 1
#define G2OHIP_HLL 2
#define G2OHIP_HSCHUR 3
#define G2OHIP_DINV 4

typedef struct g2ohip_solver g2ohip_solver;
typedef struct g2ohip_linear_solver g2ohip_linear_solver;

/* Per-call timing/size record; same fields as G2OBatchStatistics (g2o/core/batch_stats.h:40-77)
 * so `g2o -stats` columns line up.  Times in seconds, measured with HIP events on the
 * solver's stream (only filled while profiling is enabled, see g2ohip_set_profiling). */
typedef struct g2ohip_stats {
  double timeQuadraticForm;          /* buildSystem                          */
  double timeSchurComplement;        /* K5-K8                                */
  double timeSymbolicDecomposition;  /* host ordering + symbolic, last build */
  double timeNumericDecomposition;   /* multifrontal factorisation           */
  double timeLinearSolution;         /* triangular solves                    */
  double timeLinearSolver;           /* numeric + solves                     */
  double timeBackSubstitution;       /* landmark back-substitution           */
  size_t hessianDimension, hessianPoseDimension, hessianLandmarkDimension;
  size_t choleskyNNZ;                /* scalar nnz(L)                        */
  size_t numFronts, numLevels, maxFrontDim;
  size_t iterationsLinearSolver;     /* PCG iterations of the last solve (G2OBatchStatistics::iterationsLinearSolver) */
  double timeResiduals;              /* device front end: last error-only evaluation (computeActiveErrors)   */
  double timeLinearize;              /* device front end: last error + Jacobian evaluation                   */
  double timeUpdate;                 /* device front end: last oplus over all vertices (SparseOptimizer::update) */
  size_t dependencyFallbacks;        /* dependency-driven launches that gave up waiting and were repeated level by level (0 expected) */
  size_t bandChains;                 /* leaf chains of the elimination tree factorised by the sliding-window band kernel */
  size_t bandCholeskyNNZ, bandPivots; /* ... their share of choleskyNNZ and of the pivot columns (scalars) */
  size_t shardedCollectives;         /* all-reduces of the last g2ohip_solve_sharded (2 with option sharded_merge where the partition allows, else 3) */
  size_t treeBackwardGroups;         /* groups of fronts (one workgroup each) that sweep the tree levels backward (option tree_backward; 0: task by task) */
  double choleskyFlops;              /* floating-point operations of one numeric factorisation (dense-front count of the symbolic analysis: sum over the fronts of npiv m^2 - npiv^2 m + npiv^3 / 3) */
} g2ohip_stats;
/* (G2OBatchStatistics::timeIteration / levenbergIterations / chi2 belong to the caller's optimisation loop:
 *  openslam_g2o_amd/lm.py fills them and prints the `g2o -stats` line, batch_stats.cpp:49-82.) */

const char* g2ohip_last_error(void);
int g2ohip_device_count(void);

/* ---- wide seam: g2o::BlockSolver<BlockSolverTraits<pose_dim, landmark_dim>> ------------- */

/* BlockSolver(LinearSolverType*) ctor, block_solver.h:116-120.  device = HIP ordinal. */
int g2ohip_create(g2ohip_solver** out, int pose_dim, int landmark_dim, int device);
/* ~BlockSolver, block_solver.hpp:135-140 */
void g2ohip_destroy(g2ohip_solver* s);
/* Optional: run on a caller-owned hipStream_t (e.g. torch's current stream).  NULL = own stream. */
int g2ohip_set_stream(g2ohip_solver* s, void* hip_stream);

/* Solver::init(optimizer, online), block_solver.hpp:606-620: forget numeric + symbolic state. */
int g2ohip_init(g2ohip_solver* s);

/* Graph topology, replacing the walk over activeEdges() in buildStructure
 * (block_solver.hpp:206-254).  One call per homogeneous edge type ("edge set"): all edges
 * of a set share error_dim and the vertex classes of their two sides.
 *   v0, v1: hessianIndex of vertex 0 / vertex 1 in indexMapping order
 *           (sparse_optimizer.cpp:166-190): poses are [0, num_poses), landmarks
 *           [num_poses, num_poses+num_landmarks), -1 = fixed vertex.
 *   v1 == NULL: unary edges (BaseUnaryEdge, base_unary_edge.hpp:42-72).
 * Returns the set id (>= 0) or an error code (< 0).  Must precede g2ohip_build_structure. */
int g2ohip_add_edge_set(g2ohip_solver* s, int error_dim, int n_edges, const int32_t* v0, const int32_t* v1);
/* n-ary edges (BaseMultiEdge::constructQuadraticForm, base_multi_edge.hpp:170-222: H_ii and b_i once per vertex, H_ij once per
 * pair i < j, the robust weight of the edge on all of them): ONE binary edge set per vertex pair (i, j) of the edges, each handed
 * the pair's two Jacobians and the edges' information / error / robust kernel, with the parts another pair already contributes
 * switched off -- G2OHIP_PART_NO_VERTEX0 / _NO_VERTEX1: no diagonal block and no right-hand side for that side of the set,
 * G2OHIP_PART_NO_CHI2: the set's edges are not counted in g2ohip_chi2 / g2ohip_trial_stats.  E.g. three vertices: pair (0, 1)
 * whole, pair (0, 2) with NO_VERTEX0 | NO_VERTEX1 | NO_CHI2, pair (1, 2) with NO_VERTEX0 | NO_CHI2 (vertex 2's own terms).  Generic path on one GPU;
 * before g2ohip_build_structure. */
#define G2OHIP_PART_NO_VERTEX0 1
#define G2OHIP_PART_NO_VERTEX1 2
#define G2OHIP_PART_NO_CHI2 4
int g2ohip_set_edge_set_parts(g2ohip_solver* s, int set, int parts);

/* Solver::buildStructure(), block_solver.hpp:142-295.  do_schur mirrors setSchur()
 * (OptimizationAlgorithmWithHessian::init, optimization_algorithm_with_hessian.cpp:55-69). */
int g2ohip_build_structure(g2ohip_solver* s, int num_poses, int num_landmarks, int do_schur);

/* Per-iteration edge data = what linearizeOplus()/computeError() leave in each edge
 * (jacobianOplusXi/Xj, information, error; base_binary_edge.hpp:54-63), flat in edge order:
 *   J0 [n][error_dim x dim(v0)] column-major, J1 likewise (NULL for unary sets),
 *   omega [n][error_dim x error_dim], err [n][error_dim].
 * on_device != 0: the pointers are device pointers that stay valid until the next
 * g2ohip_build_system returns (zero-copy); otherwise host arrays, copied to the device. */
int g2ohip_set_edge_data(g2ohip_solver* s, int set, const double* J0, const double* J1, const double* omega,
                         const double* err, int on_device);
/* The errors of a set alone ([n][error_dim], host array) after g2ohip_set_edge_data with host arrays: what computeActiveErrors
 * (sparse_optimizer.cpp:61-74) leaves in the edges at TRIAL estimates -- chi2 of a Levenberg-Marquardt trial reads nothing else;
 * the Jacobians and information matrices stay what the system was built from. */
int g2ohip_set_edge_errors(g2ohip_solver* s, int set, const double* err);
/* OptimizableGraph::Edge::setRobustKernel (g2o.cpp:322-336); delta as RobustKernel::setDelta. */
int g2ohip_set_robust_kernel(g2ohip_solver* s, int set, int kind, double delta);
/* The same per EDGE: in g2o the robust kernel is a member of the edge (optimizable_graph.h:436-443, asked per edge by
 * base_binary_edge.hpp:92-112), so a pose graph with kernels on its loop closures only is still ONE set of EdgeSE2 / EdgeSE3.
 * kind [n], delta [n] (kind 0 = none for that edge); kind == NULL returns to the set-level kernel.  Not for a set bound to the
 * BA front end (g2ohip_ba_set_edges_classes carries the kernels there: G2OHIP_ERR_STATE).  Edges appended later by
 * g2ohip_update_structure start with no kernel until this is called again for the grown set. */
int g2ohip_set_robust_kernel_per_edge(g2ohip_solver* s, int set, const int32_t* kind, const double* delta);

/* Solver::buildSystem(), block_solver.hpp:501-560 (clear + per-edge constructQuadraticForm +
 * gather b).  Returns G2OHIP_OK on success (the reference returns 0 and nobody checks). */
int g2ohip_build_system(g2ohip_solver* s);
/* SparseOptimizer::activeRobustChi2(), sparse_optimizer.cpp:100-114, from the current edge data. */
int g2ohip_chi2(g2ohip_solver* s, double* chi2);

/* Solver::setLambda(lambda, backup) / restoreDiagonal(), block_solver.hpp:563-604.
 * With the Schur complement enabled the damping is virtual: lambda is held in device scalars and applied where
 * the diagonal is consumed (Hll + lambda I before the landmark inversion, Hpp + lambda I when Hschur is
 * formed), so Hpp / Hll in HBM are not modified and restoreDiagonal is exact by construction.  Everything read
 * through this ABI (g2ohip_copy_values, g2ohip_multiply_hessian, g2ohip_max_diagonal, Hschur, Dinv, x) shows the
 * damped system exactly as after the reference's setLambda; only raw g2ohip_device_array views of Hpp / Hll are
 * undamped.  Without Schur the diagonal is modified and restored in place as in the reference. */
int g2ohip_set_lambda(g2ohip_solver* s, double lambda, int backup);
int g2ohip_restore_diagonal(g2ohip_solver* s);
/* max_j |H_jj| over all free vertices: the quantity computeLambdaInit reads through
 * v->hessian(j,j) (optimization_algorithm_levenberg.cpp:149-163). */
int g2ohip_max_diagonal(g2ohip_solver* s, double* out);
/* sum_j x_j (lambda x_j + b_j): OptimizationAlgorithmLevenberg::computeScale (:165-172). */
int g2ohip_compute_scale(g2ohip_solver* s, double lambda, double* out);
/* The scalar diagonal of H (poses then landmarks, hessian-index order; vector_size() doubles) with the current damping:
 * what OptimizationAlgorithmLevenberg::computeLambdaInit reads through v->hessian(j, j) in the vertices' mapped memory
 * (optimization_algorithm_levenberg.cpp:149-163, base_vertex.h:62-110) -- the g2o adapter mirrors it on the host. */
int g2ohip_copy_diagonal(g2ohip_solver* s, double* diag_host);

/* Solver::solve(), block_solver.hpp:353-486: Schur complement, sparse block Cholesky of the
 * reduced pose system, landmark back-substitution.  G2OHIP_OK | G2OHIP_NOT_PD | error.
 * b is left untouched (block_solver.hpp:435-436). */
int g2ohip_solve(g2ohip_solver* s);

/* Solver::vectorSize(), x(), b() (solver.h:80-86).  x/b layout: poses then landmarks
 * (block_solver.hpp:551-557).  copy_* synchronise and copy to host; *_device return the
 * resident vectors (valid until destroy / build_structure). */
size_t g2ohip_vector_size(g2ohip_solver* s);
int g2ohip_copy_x(g2ohip_solver* s, double* x_host);
/* Overwrite the resident increment x (what g2ohip_ba_update / g2ohip_pg_update apply): SparseOptimizer::update
 * takes an arbitrary vector, e.g. Dogleg's h_dl (optimization_algorithm_dogleg.cpp:179). */
int g2ohip_set_x(g2ohip_solver* s, const double* x_host);
int g2ohip_copy_b(g2ohip_solver* s, double* b_host);
const double* g2ohip_x_device(g2ohip_solver* s);
const double* g2ohip_b_device(g2ohip_solver* s);

/* dest += H * src over the full [Hpp Hpl; Hpl' Hll] system (host vectors of vectorSize()).
 * Superset of BlockSolverBase::multiplyHessian (block_solver.h:83-91,142), used by Dogleg and
 * by residual checks. */
int g2ohip_multiply_hessian(g2ohip_solver* s, double* dest_host, const double* src_host);

int g2ohip_sync(g2ohip_solver* s);
/* HIP-event timing on the solver's stream.  0: off; 1: every kernel slot (g2ohip_kernel_time) and the stage
 * timers of g2ohip_get_stats; 2 + k: kernel slot k only (two event records per iteration -- the records are
 * not free next to sub-millisecond iterations). */
int g2ohip_set_profiling(g2ohip_solver* s, int enabled);
int g2ohip_get_stats(g2ohip_solver* s, g2ohip_stats* out);
/* Per-kernel HIP-event timing on the solver's stream (while profiling is enabled): slot in
 * [0, g2ohip_kernel_slots()), accumulated seconds and launch count since the last reset. */
int g2ohip_kernel_slots(void);
const char* g2ohip_kernel_name(int slot);
int g2ohip_kernel_time(g2ohip_solver* s, int slot, double* total_seconds, long* launches, int reset);
/* One Levenberg-Marquardt trial with a single host synchronisation (optimization_algorithm_levenberg.cpp:96-128 asks for
 * the solve status, the new chi2 and computeScale one after the other): g2ohip_solve_async queues g2ohip_solve and
 * leaves its status on the device; after the caller has updated the estimates and re-evaluated the errors,
 * g2ohip_trial_stats returns that status (solve_ok 1/0), activeRobustChi2 and computeScale(lambda) = x'(lambda x + b)
 * together.  Without a pending g2ohip_solve_async it just evaluates the two sums.  *solve_ok: 1 solved, 0 not positive
 * definite, 2 the solve has to be REPEATED (a dependency-driven launch gave up waiting -- never observed, the safety net of
 * DESIGN.md section 2; g2ohip_solve repeats by itself, the asynchronous pair leaves it to the caller: pop the estimates,
 * run the trial again).
 * g2ohip_trial_stats_begin queues the sums and their read-back WITHOUT waiting; the next g2ohip_trial_stats then only waits and
 * returns them (its lambda is ignored).  For a caller with host work to overlap: the adapter's LM driver queues the first trial of
 * the NEXT iteration this way and writes the accepted estimates into g2o's vertices (SparseOptimizer::update's effect,
 * sparse_optimizer.cpp:421-437) while the device works. */
/* The reduced (Schur) system as an operator, never formed (what "linear_solver" 2 iterates on; callers that run their own
 * Krylov loop, e.g. sharded over GPUs with one all-reduce of the product per iteration, use these directly):
 * prepare: Dinv = (Hll + lambda_l I)^-1, bschur = b_p - Hpl Dinv b_l (g2ohip_device_array 100) and the diagonal blocks
 * Hpp_ii + lambda_p I - sum B Dinv B' (g2ohip_device_array 107, [nP][p*p]); apply: out = (Hpp + lambda_p I - Hpl Dinv Hpl') in
 * on device vectors of nP*p doubles.  With landmarks sharded over ranks both are this rank's summands. */
int g2ohip_schur_operator_prepare(g2ohip_solver* s);
int g2ohip_schur_operator_apply(g2ohip_solver* s, const double* in_device, double* out_device);
int g2ohip_solve_async(g2ohip_solver* s);
int g2ohip_trial_stats_begin(g2ohip_solver* s, double lambda);
int g2ohip_trial_stats(g2ohip_solver* s, double lambda, int* solve_ok, double* chi2, double* scale);
/* Options (name, value).  Linear solver of the reduced system: "linear_solver" 0 = multifrontal block Cholesky
 * (replaces LinearSolverCSparse / LinearSolverCholmod), 1 = block-Jacobi preconditioned CG (replaces
 * LinearSolverPCG, g2o/solvers/pcg/linear_solver_pcg.hpp:79-196; 2 = the same iteration matrix-free: with the Schur
 * complement on, Hschur is never formed -- Hschur v = Hpp v + lambda v - Hpl Dinv Hpl' v, exact block-Jacobi preconditioner)
 * with "pcg_tolerance" (1e-6),
 * "pcg_absolute_tolerance" (1), "pcg_max_iterations" (-1 = dimension), like LinearSolverPCG's setters
 * (linear_solver_pcg.h:74-85); iterations of the last solve in g2ohip_stats.iterationsLinearSolver.
 * Ordering / symbolic knobs (before g2ohip_build_structure): "nd_leaf" (nested-dissection leaf size in blocks, 32),
 * "max_sn_scalars" (48) / "max_sn_scalars_lds" (24: fronts that fit LDS), "relax_zeros", "relax_front_bytes",
 * "lds_front_bytes", "fuse_chains", "wave_front_tasks", "dep_levels" (64: task levels that may share one
 * dependency-driven launch, 0/1 = one launch per level), "dep_backward" (1), "dep_delay", "dep_spin_limit", "dep_acq_rel" (0; 1 =
 * release / acquire on the dependency counters, for A/B validation), "band_kernel" (1: leaf chains of a band in the one-wave
 * sliding-window kernel), "fuse_fwd_any" (1: forward sweep fused into the factor kernel whatever a front's children),
 * "lds_mfma" ((4 << 16) | 96: LDS fronts with at least that many pivot blocks / boundary rows get one MFMA trailing update),
 * "big_front_passes" (1: large fronts as whole-GPU passes with an MFMA update) / "big_front_min_dim" (180);
 * for those passes "inplace_chains", "mfma_diag", "fuse_panel", "overlap_level_halves", "hoist_big_assembly",
 * "fuse_big_forward", "split_sweeps" / "split_sweeps_min_dim" (512), "merge_backward_levels", "merge_diag_panel" (all 1: see INTEGRATION.md, options).
 * Kernel knobs: "schur_tile_bytes", "schur_group", "fuse_landmark_inverse", "fuse_schur_reduce" (1: g2ohip_solve on
 * one GPU folds the Schur reduction into the factorisation; Hschur is then written only when it is asked for),
 * "ba_fused", "ba_store_ll" (0: Hll and the errors of the fused BA path reach HBM only when a reader asks), "use_graph",
 * "mask_solution", "sharded_virtual" (1: on a rank the factorisation reads Hpp and the partial blocks itself, only the boundary
 * blocks of the reduced system are reduced and exchanged as blocks), "sharded_graph" (1: g2ohip_solve_sharded as one hipGraph where nothing crosses the host; 2: with RCCL too),
 * "sharded_merge" (1: the boundary blocks / b_p of a sharded solve travel in the all-reduce of the subtree roots -- two collectives per
 * solve instead of three -- whenever only the shared top of the tree consumes them; 0: always three),
 * "comm_emulate" (timing only).  G2OHIP_OPTIONS="name=value,..." in the environment sets options for every solver of a process: g2ohip_create and
 * g2ohip_ls_create apply it (so the g2o plugin sees it too); a malformed or, for g2ohip_create, unknown entry fails the
 * creation with G2OHIP_ERR_ARG; an explicit g2ohip_set_option afterwards wins. */
int g2ohip_set_option(g2ohip_solver* s, const char* name, double value);

/* Inspection for parity tests (saveHessian-like, block_solver.hpp:628-632): block patterns
 * (column pointers + row block indices, ascending rows per column) and raw block values. */
int g2ohip_get_nnzb(g2ohip_solver* s, int which, int* nnzb);
int g2ohip_get_pattern(g2ohip_solver* s, int which, int32_t* colptr, int32_t* rowidx);
int g2ohip_copy_values(g2ohip_solver* s, int which, double* values_host);
/* Device pointers + element counts of resident arrays (for multi-GPU exchange through RCCL):
 * which = G2OHIP_HSCHUR (values), or 100 = bschur. */
int g2ohip_device_array(g2ohip_solver* s, int which, double** ptr, size_t* count);
/* BlockSolver::computeMarginals / LinearSolver::solvePattern (block_solver.hpp:489-498, linear_solver.h:63-69,
 * marginal_covariance_cholesky.cpp:71-220): blocks (rows[i], cols[i]) (pose block indices) of the inverse of Hpp
 * with the current damping -- what the reference hands to solvePattern (`*_Hpp`, block_solver.hpp:492), also when
 * the Schur complement is on.  Option "marginals_reduced" = 1 (g2ohip_set_option) inverts the reduced pose system
 * instead: the pose marginals with the landmarks integrated out (a deviation from the reference, off by default).
 * Blocks inside the pattern of the factor come from ONE sparse-inverse pass over the frontal matrices (all of them at
 * once: what MarginalCovarianceCholesky's recursion computes entry by entry), the others from a pair of triangular
 * sweeps per column; option "marginals_recursion" = 0 forces the column path.
 * out [n][p*p], column-major blocks.  After g2ohip_build_system.  G2OHIP_OK | G2OHIP_NOT_PD. */
int g2ohip_compute_marginals(g2ohip_solver* s, int n_blocks, const int32_t* rows, const int32_t* cols, double* out);
/* The per-edge data the next g2ohip_build_system will consume, copied to the host (inspection / tests of the
 * device-side producers): J0 [n][d*dim0], J1 [n][d*dim1], err [n][d]; any pointer may be NULL. */
int g2ohip_copy_edge_data(g2ohip_solver* s, int set, double* J0, double* J1, double* err);

/* Drops every edge set (and the device front-end bindings) of the handle: the next g2ohip_add_edge_set starts a new
 * graph.  What BlockSolver::buildStructure does implicitly when it is called again (block_solver.hpp:142-204 deallocates
 * and rebuilds Hpp / Hll / Hpl): a second optimize() on the same solver, or a rebuild after online growth. */
int g2ohip_clear_edge_sets(g2ohip_solver* s);

/* Solver::updateStructure(vset, edges), block_solver.hpp:297-351: online growth WITHOUT Schur complement.  The new pose
 * vertices take the hessian indices [num_poses, num_poses + num_new_poses) (SparseOptimizer::updateInitialization appends them
 * to the index mapping, sparse_optimizer.cpp:269-352); the new edges are appended to an existing edge set (same error
 * dimension and vertex classes; for another edge type call g2ohip_add_edge_set first and pass n_new_edges = 0 here).  The
 * structure is rebuilt from the enlarged topology: g2ohip_vector_size grows, per-edge data of the touched set has to be
 * handed over again for ALL its edges (old ones first); a device front end bound to the touched set (g2ohip_pg_set_edges)
 * is unbound -- call g2ohip_pg_set_edges + g2ohip_pg_set_estimates again after growth (g2ohip_pg_linearize returns
 * G2OHIP_ERR_STATE until then).  G2OHIP_ERR_UNSUPPORTED where the reference aborts (a system with
 * marginalised vertices, :313-316). */
int g2ohip_update_structure(g2ohip_solver* s, int num_new_poses, int set, int n_new_edges, const int32_t* v0, const int32_t* v1);

/* Multi-GPU sharding support (openslam_g2o_amd/distributed.py).  Each rank holds a landmark
 * shard; to give every rank the SAME Hschur block layout (so the partial Schur complements
 * can be summed element-wise by one RCCL all-reduce) the union pattern is declared up front:
 * extra structural blocks (row <= col, pose block indices) of the reduced system.  Must
 * precede g2ohip_build_structure. */
int g2ohip_add_schur_pattern(g2ohip_solver* s, int n_blocks, const int32_t* rows, const int32_t* cols);
/* setLambda with separate damping of the pose and landmark diagonals: every rank damps its own
 * landmarks, only one rank adds lambda to the (summed) pose diagonal. */
int g2ohip_set_lambda_split(g2ohip_solver* s, double lambda_pose, double lambda_landmark, int backup);

/* Subtree-distributed factorisation of the reduced system: the top of the
 * elimination-task tree is split into >= world subtrees; rank r factorises its own subtrees, every rank
 * factorises the few shared tasks above them redundantly.  g2ohip_set_partition precedes
 * g2ohip_build_structure (all ranks must pass the same union Schur pattern and options: the symbolic
 * analysis is deterministic, so every rank derives the same partition).  Per solve:
 *   g2ohip_solve_reduced_local   own subtrees: factor + forward sweep, pack the subtree roots' update
 *                                matrices/vectors into the exchange buffer (g2ohip_device_array 103)
 *   [caller: all-reduce(SUM) of buffer 103 over the ranks]
 *   g2ohip_solve_reduced_shared  shared top: factor + forward/backward, backward sweep of the own
 *                                subtrees, mask x_p entries owned elsewhere (g2ohip_device_array 104)
 *   [caller: all-reduce(SUM) of buffer 104]
 *   g2ohip_solve_reduced_finish  un-permute x_p; G2OHIP_NOT_PD if a local pivot was <= 0
 * With option "mask_solution" = 0 the masking is skipped: every rank then holds valid x_p for its own and
 * the shared poses, and the caller exchanges only the few foreign poses its landmarks observe (after
 * g2ohip_solve_reduced_finish, on g2ohip_device_array 101).
 * g2ohip_get_partition: owner rank (-1 = shared) of every pose block and of every Hschur block (the rank
 * that consumes its value) -- what the caller needs to exchange only the boundary blocks.
 * g2ohip_partition_poses: the same partition (block_consumer may be NULL) from a bare block pattern, host only (no device), so
 * landmarks can be dealt to ranks before build_structure; options_from (may be NULL) supplies the ordering
 * options (g2ohip_set_option) so that the result matches what that solver will compute.  In partitioned
 * mode g2ohip_set_lambda[_split] damps a pose block only on the rank that consumes its diagonal block. */
int g2ohip_set_partition(g2ohip_solver* s, int rank, int world);
int g2ohip_solve_reduced_local(g2ohip_solver* s);
int g2ohip_solve_reduced_shared(g2ohip_solver* s);
int g2ohip_solve_reduced_finish(g2ohip_solver* s);
/* The same exchange without host round trips (what openslam_g2o_amd/distributed.py does per solve): the caller
 * registers once the reduced-system blocks / poses on partition boundaries (with 0/1 keep flags: a rank keeps the
 * summed value only of what it consumes) and the foreign "halo" poses its landmarks observe (flag 1 where this rank
 * owns the value); then per solve
 *   g2ohip_solve_schur, g2ohip_exchange_pack(1), [all-reduce g2ohip_device_array 105], g2ohip_exchange_unpack(1),
 *   g2ohip_solve_reduced_local, [all-reduce 103], g2ohip_solve_reduced_shared, g2ohip_solve_reduced_finish_async,
 *   g2ohip_exchange_pack(3), [all-reduce 106: halo x_p + failure flags], g2ohip_exchange_unpack(3),
 *   g2ohip_solve_back_substitute, g2ohip_exchange_status  (the only synchronisation; G2OHIP_NOT_PD if any rank failed,
 *   G2OHIP_REPEAT -- on every rank alike -- if the phases are to be run again: see the define). */
int g2ohip_exchange_setup(g2ohip_solver* s, int n_blocks, const int32_t* block_idx, const double* block_keep, int n_poses,
                          const int32_t* pose_idx, const double* pose_keep, int n_halo, const int32_t* halo_idx, const double* halo_mine);
int g2ohip_exchange_pack(g2ohip_solver* s, int which);
int g2ohip_exchange_unpack(g2ohip_solver* s, int which);
int g2ohip_exchange_status(g2ohip_solver* s);
int g2ohip_solve_reduced_finish_async(g2ohip_solver* s);
int g2ohip_get_partition(g2ohip_solver* s, int32_t* pose_owner, int32_t* block_consumer);

/* ---- collectives inside the library: the N > 1 path for C / C++ consumers -------------------------------------------
 * One process per GPU; every rank creates its solver, registers ITS shard of the edges, calls g2ohip_set_partition +
 * g2ohip_build_structure + g2ohip_exchange_setup (the index lists come from g2ohip_partition_poses, see
 * openslam_g2o_amd/distributed.py for the host-side recipe) and attaches a communicator:
 *   RCCL over xGMI: rank 0 calls g2ohip_comm_unique_id, hands the 128 bytes to the other ranks by any means (MPI, a file,
 *   a socket), every rank calls g2ohip_comm_init_rccl -- ncclCommInitRank; librccl is bound at run time, libg2ohip.so
 *   itself has no link dependency on it;
 *   host callback: g2ohip_comm_init_host with an in-place all-reduce over host memory (op 0 = sum, 1 = max; MPI_Allreduce
 *   fits directly): device buffers are staged through pinned memory.  For ranks sharing one GPU (tests) and boxes
 *   without peer access; not a performance path;
 *   peer mailboxes (opt-in): g2ohip_comm_init_peer -- every rank exports a device mailbox (hipIpcGetMemHandle; the handles
 *   travel through the host all-reduce once), an all-reduce is one kernel that stores the payload into the rank's slot of
 *   every peer's mailbox over xGMI and one that waits for all slots and adds them in rank order: two launches for the
 *   latency-sized payloads of the sharded solve instead of a ring.  slot_doubles = capacity of a slot (larger payloads go
 *   in pieces).  A peer that does not deliver within the wait limit (5 s) fails the call that next synchronises
 *   (G2OHIP_ERR_STATE) instead of hanging the device.  Verified with several processes on one GPU only; RCCL is the default.
 * g2ohip_solve_sharded then runs the whole linear solve (BlockSolver::solve, block_solver.hpp:353-486, on the sharded
 * system): local Schur pass | all-reduce of the boundary blocks of the reduced system and boundary b_p | own subtrees of
 * the elimination tree | all-reduce of the subtree roots' update matrices | shared top + backward sweep | all-reduce of
 * the halo x_p and the failure flags | back-substitution of the own landmarks.  G2OHIP_OK | G2OHIP_NOT_PD (on every rank
 * alike).  x_p is valid for own, shared and halo poses, x_l for the own landmarks.
 * The *_sharded scalars are the Levenberg-Marquardt quantities over all ranks (activeRobustChi2, computeLambdaInit's
 * maximum, computeScale: optimization_algorithm_levenberg.cpp:149-172). */
typedef int (*g2ohip_host_allreduce_fn)(void* ctx, double* host_buffer, size_t count, int op);
int g2ohip_comm_unique_id(char* id128);
int g2ohip_comm_init_rccl(g2ohip_solver* s, int rank, int world, const char* id128);
int g2ohip_comm_init_host(g2ohip_solver* s, int rank, int world, g2ohip_host_allreduce_fn fn, void* ctx);
int g2ohip_comm_init_peer(g2ohip_solver* s, int rank, int world, g2ohip_host_allreduce_fn fn, void* ctx, size_t slot_doubles);
int g2ohip_comm_destroy(g2ohip_solver* s);
int g2ohip_comm_all_reduce(g2ohip_solver* s, double* device_buffer, size_t count, int op);   /* in place, on the solver's stream */
int g2ohip_solve_sharded(g2ohip_solver* s);
int g2ohip_chi2_sharded(g2ohip_solver* s, double* chi2);
int g2ohip_max_diagonal_sharded(g2ohip_solver* s, double* out);
int g2ohip_compute_scale_sharded(g2ohip_solver* s, double lambda, double* out);
int g2ohip_partition_poses(const g2ohip_solver* options_from, int block_dim, int n_blocks, const int32_t* colptr,
                           const int32_t* rowidx, int world, int32_t* pose_owner, int32_t* block_consumer);

/* Split solve for the multi-GPU path: g2ohip_solve == schur + reduced + back_substitute. */
int g2ohip_solve_schur(g2ohip_solver* s);            /* K5-K8: Hschur, bschur, Dinv          */
int g2ohip_solve_reduced(g2ohip_solver* s);          /* K9-K12: x_p = Hschur \ bschur        */
int g2ohip_solve_back_substitute(g2ohip_solver* s);  /* K13: x_l = Dinv (b_l - Hpl' x_p)     */

/* ---- page-locked host buffers ---------------------------------------------------------------
 * The boundary takes plain host pointers (Solver::_x / _b are `new double[]` of g2o's own Solver::resizeVector,
 * g2o/core/solver.cpp:46-70).  Pageable memory crosses PCIe through the driver's staging copy (~6-10 GB/s); a caller that
 * keeps handing over the SAME buffers every iteration -- the adapter: estimates up, b / x / diagonal down -- registers them
 * once (hipHostRegister) and the copies run at the link rate.  Unregister before the buffer is freed or reallocated. */
int g2ohip_host_register(g2ohip_solver* s, void* ptr, size_t bytes);
int g2ohip_host_unregister(g2ohip_solver* s, void* ptr);

/* ---- device-resident bundle-adjustment front end (a "next" row) -----
 * For graphs of EdgeProjectXYZ2UV (g2o/types/sba/types_six_dof_expmap.h:127-150) the library can
 * produce errors and Jacobians itself and keep the vertex estimates on the device, so a whole
 * Levenberg-Marquardt trial loop (optimization_algorithm_levenberg.cpp:57-146) runs without moving
 * Jacobians or estimates over PCIe.  Estimates: cams [n][12] = R (column-major) | t, world -> camera
 * (what VertexSE3Expmap holds, types_six_dof_expmap.cpp:88-103); points [n][3]. */
/* edge set `set` (added with error_dim 2, vertex 0 = point, vertex 1 = pose): per-edge indices into
 * the estimate arrays (all vertices, fixed ones included), measurements [n][2], information [n][2x2]
 * (NULL = identity), CameraParameters (focal_length, principle_point). After g2ohip_build_structure. */
int g2ohip_ba_set_edges(g2ohip_solver* s, int set, const int32_t* cam_vertex, const int32_t* point_vertex, const double* meas,
                        const double* info, double focal_length, double cx, double cy);
/* The same with EDGE CLASSES: the edges of one set may differ in their CameraParameters (every EdgeProjectXYZ2UV carries
 * its own _cam, types_six_dof_expmap.h:133-153) and in their robust kernel (OptimizableGraph::Edge::setRobustKernel,
 * optimizable_graph.h:436-443; base_binary_edge.hpp:92-112 asks each edge for its own).  class_params [n_classes][5] =
 * (focal_length, principle_point x, y, robust kernel kind as in g2ohip_set_robust_kernel, delta), edge_class [n] the class of
 * every edge (NULL with one class).  1 <= n_classes <= 128; with more than one class the set must be on the fused path
 * (one observation per (pose, landmark) pair, no fixed landmark) -- G2OHIP_ERR_ARG otherwise, and
 * g2ohip_set_robust_kernel on the set is refused (G2OHIP_ERR_STATE) while the classes are bound. */
int g2ohip_ba_set_edges_classes(g2ohip_solver* s, int set, const int32_t* cam_vertex, const int32_t* point_vertex, const double* meas,
                                const double* info, int n_classes, const double* class_params, const int32_t* edge_class);
/* setEstimate for every vertex + the index mapping: cam_hidx[v] = hessianIndex (-1 fixed),
 * point_hidx[v] = landmark index (0-based, i.e. hessianIndex - num_poses) or -1. */
int g2ohip_ba_set_estimates(g2ohip_solver* s, int n_cams, const double* cams, const int32_t* cam_hidx, int n_points,
                            const double* points, const int32_t* point_hidx);
int g2ohip_ba_get_estimates(g2ohip_solver* s, double* cams, double* points);
/* The estimates of SELECTED vertices (indices into the arrays of g2ohip_ba_set_estimates): cams [n_cams][12], points [n_points][3].
 * What a caller with a few host-side edges reads of an LM trial -- BaseUnaryEdge / BaseMultiEdge::computeError
 * (base_unary_edge.hpp:42-72) need the estimates of the vertices THEY touch, not all 1.1 M -- next to the full read-back. */
int g2ohip_ba_get_estimates_of(g2ohip_solver* s, int n_cams, const int32_t* cam_index, double* cams, int n_points, const int32_t* point_index,
                               double* points);
/* The same read-back (what SparseOptimizer::update leaves in the vertices, sparse_optimizer.cpp:422-432, fetched for the caller's
 * setEstimate loop) started ASYNCHRONOUSLY behind everything queued so far -- typically right after g2ohip_ba_update of an LM
 * trial -- on a copy stream of the library, in pieces: piece 0 = the cameras, pieces 1 .. point_pieces (<= 16) = the points in
 * equal ranges.  g2ohip_ba_fetch_estimates_wait(piece) returns once that piece is in the caller's buffer (page-locked buffers,
 * g2ohip_host_register, make the copy truly asynchronous): the caller writes piece k into its vertices while piece k + 1 is
 * still in flight and the device evaluates the trial's errors.  The buffers must stay untouched until the last piece has been
 * waited for or the next g2ohip_ba_set_estimates; device-side writers of the estimates (update, pop) wait for the copy themselves. */
int g2ohip_ba_fetch_estimates_begin(g2ohip_solver* s, double* cams, double* points, int point_pieces);
int g2ohip_ba_fetch_estimates_wait(g2ohip_solver* s, int piece);
/* computeActiveErrors() (+ linearizeOplus() when jacobians != 0) into the set's edge data
 * (sparse_optimizer.cpp:61-76, types_six_dof_expmap.cpp:288-326). */
int g2ohip_ba_linearize(g2ohip_solver* s, int jacobians);
/* SparseOptimizer::update(x): oplus of every free vertex with the current solution (sparse_optimizer.cpp:421-434). */
int g2ohip_ba_update(g2ohip_solver* s);
/* SparseOptimizer::push / pop / discardTop on all vertices (one level, what LM needs; sparse_optimizer.cpp:599-650). */
int g2ohip_ba_push(g2ohip_solver* s);
int g2ohip_ba_pop(g2ohip_solver* s);
int g2ohip_ba_discard_top(g2ohip_solver* s);

/* ---- device-resident pose-graph front end ----------------------------
 * Error + Jacobian producers and vertex updates for pose graphs, so a Gauss-Newton / LM iteration needs no
 * host round trip: type 1 = EdgeSE2 / VertexSE2 (g2o/types/slam2d/edge_se2.h:51-57, edge_se2.cpp:76-99,
 * vertex_se2.h:55-59), estimates and measurements (x, y, theta), information [n][3x3];
 * type 2 = EdgeSE3 / VertexSE3 (g2o/types/slam3d/edge_se3.cpp:48-75, isometry3d_gradients.h:39-126,
 * vertex_se3.h:107-116), estimates and measurements as isometries [12] = R (column-major) | t,
 * information [n][6x6].  Edge set `set` was added with error_dim 3 / 6 and hessian indices of its two
 * vertices; vi/vj index the estimate array (all vertices, fixed ones included), hidx[v] = hessianIndex or -1.
 * The calls mirror the g2ohip_ba_* ones (set_edges after g2ohip_build_structure). */
int g2ohip_pg_set_edges(g2ohip_solver* s, int set, int type, const int32_t* vi, const int32_t* vj, const double* meas,
                        const double* info);
int g2ohip_pg_set_estimates(g2ohip_solver* s, int n_vertices, const double* poses, const int32_t* hidx);
int g2ohip_pg_get_estimates(g2ohip_solver* s, double* poses);
int g2ohip_pg_linearize(g2ohip_solver* s, int jacobians);
int g2ohip_pg_update(g2ohip_solver* s);
int g2ohip_pg_push(g2ohip_solver* s);
int g2ohip_pg_pop(g2ohip_solver* s);
int g2ohip_pg_discard_top(g2ohip_solver* s);

/* ---- narrow seam: g2o::LinearSolver<MatrixType> ------------------------------------------- */

/* LinearSolver ctor for MatrixType = block_dim x block_dim. */
int g2ohip_ls_create(g2ohip_linear_solver** out, int block_dim, int device);
void g2ohip_ls_destroy(g2ohip_linear_solver* ls);
/* LinearSolver::init(), linear_solver_csparse.h:97-104: drop the symbolic factorisation. */
int g2ohip_ls_init(g2ohip_linear_solver* ls);
/* LinearSolver::solve(A, x, b), linear_solver_csparse.h:106-142.  A: symmetric, upper blocks
 * only (rowidx[q] <= column), ascending rows per column, values [nnzb][bd x bd] column-major,
 * diagonal blocks fully stored.  Same pattern between init() calls (linear_solver.h:86-105).
 * x, b: host vectors of n_blocks*block_dim.  G2OHIP_OK | G2OHIP_NOT_PD | error. */
int g2ohip_ls_solve(g2ohip_linear_solver* ls, int n_blocks, const int32_t* colptr, const int32_t* rowidx,
                    const double* values, double* x, const double* b);
/* LinearSolver::solvePattern(spinv, blockIndices, A), linear_solver.h:71 / linear_solver_csparse.h:190-221 (factorise,
 * then MarginalCovarianceCholesky::computeCovariance): blocks (rows[i], cols[i]) of A^-1, out [n_req][bd x bd]
 * column-major.  Blocks inside the pattern of the factor come from one sparse-inverse pass over the frontal matrices,
 * the others from unit right-hand sides.  G2OHIP_OK | G2OHIP_NOT_PD | error. */
int g2ohip_ls_solve_pattern(g2ohip_linear_solver* ls, int n_blocks, const int32_t* colptr, const int32_t* rowidx,
                            const double* values, int n_req, const int32_t* rows, const int32_t* cols, double* out);
int g2ohip_ls_get_stats(g2ohip_linear_solver* ls, g2ohip_stats* out);
int g2ohip_ls_set_option(g2ohip_linear_solver* ls, const char* name, double value);

#ifdef __cplusplus
}
#endif
#endif /* G2OHIP_H */
