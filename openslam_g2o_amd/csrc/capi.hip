// extern "C" surface of libg2ohip (include/g2ohip.h).  Exceptions never cross the boundary:
// every entry point maps failures to the reference's convention -- a return code plus text
// (the reference path itself uses `false` + cerr).
#include "../../include/g2ohip.h"

#include <cstdlib>
#include <cstring>
#include <string>

#include "block_solver.h"

namespace g2ohip {
std::string& last_error_ref() {
  static thread_local std::string e;
  return e;
}
}  // namespace g2ohip

using namespace g2ohip;

struct g2ohip_solver {
  std::unique_ptr<BlockSolver> impl;
};

// Narrow seam: LinearSolver<MatrixType> over the same multifrontal engine.
struct g2ohip_linear_solver {
  int bs = 0, device = 0;
  hipStream_t st = nullptr;
  std::unique_ptr<SparseCholesky> chol;
  CholOptions opt;
  std::vector<int> colptr, rowidx;  // pattern the symbolic factorisation was built for
  // a second analysis for solvePattern() on a DIFFERENT pattern between two init() calls: g2o's BlockSolver factorises
  // Hschur in solve() and hands Hpp (a sub-pattern) to solvePattern() of the same LinearSolver (block_solver.hpp:489-499)
  std::unique_ptr<SparseCholesky> chol2;
  std::vector<int> colptr2, rowidx2;
  DevBuf<double> dA, db, dx;
  EventTimer tn, tl;
  double t_numeric = 0, t_solve = 0;
};

namespace {
template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const ArgFailure& e) {
    set_error(e.what());
    return G2OHIP_ERR_ARG;
  } catch (const StateFailure& e) {
    set_error(e.what());
    return G2OHIP_ERR_STATE;
  } catch (const HipFailure& e) {
    set_error(e.what());
    return G2OHIP_ERR_HIP;
  } catch (const std::exception& e) {
    set_error(e.what());
    return G2OHIP_ERR_HIP;
  }
}
#define REQUIRE_HANDLE(s)                 \
  if (!(s) || !(s)->impl) {               \
    set_error("null solver handle");      \
    return G2OHIP_ERR_ARG;                \
  }
}  // namespace

// G2OHIP_OPTIONS="name=value,..." in the environment: options for every solver handle of the process, applied at creation
// (an explicit g2ohip_set_option afterwards still wins).  Parsed HERE so that C / C++ consumers -- the g2o plugin -- see it
// too.  A malformed entry fails the creation with G2OHIP_ERR_ARG and a message; `unknown_ok` lets the narrow-seam handle
// skip names that only the block solver knows.
template <class Setter>
static int apply_env_options(Setter&& set, bool unknown_ok) {
  const char* env = std::getenv("G2OHIP_OPTIONS");
  if (!env || !*env) return G2OHIP_OK;
  std::string all(env);
  size_t pos = 0;
  while (pos <= all.size()) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    std::string kv = all.substr(pos, end - pos);
    pos = end + 1;
    const size_t a = kv.find_first_not_of(" \t"), b = kv.find_last_not_of(" \t");
    if (a == std::string::npos) continue;
    kv = kv.substr(a, b - a + 1);
    const size_t eq = kv.find('=');
    char* stop = nullptr;
    const double v = eq == std::string::npos ? 0.0 : std::strtod(kv.c_str() + eq + 1, &stop);
    if (eq == std::string::npos || eq == 0 || stop == kv.c_str() + eq + 1 || (stop && *stop && *stop != ' ')) {
      set_error("G2OHIP_OPTIONS: malformed entry '" + kv + "' (want name=value)");
      return G2OHIP_ERR_ARG;
    }
    std::string name = kv.substr(0, eq);
    name.erase(name.find_last_not_of(" \t") + 1);
    const int rc = set(name.c_str(), v);
    if (rc != G2OHIP_OK && !unknown_ok) {
      set_error("G2OHIP_OPTIONS: unknown option '" + name + "'");
      return rc;
    }
  }
  return G2OHIP_OK;
}

extern "C" {

const char* g2ohip_last_error(void) { return last_error_ref().c_str(); }

int g2ohip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int g2ohip_create(g2ohip_solver** out, int pose_dim, int landmark_dim, int device) {
  if (!out) return G2OHIP_ERR_ARG;
  *out = nullptr;
  return guarded([&] {
    auto h = std::make_unique<g2ohip_solver>();
    h->impl = std::make_unique<BlockSolver>(pose_dim, landmark_dim, device);
    g2ohip_solver* raw = h.get();
    const int rc = apply_env_options([&](const char* n, double v) { return g2ohip_set_option(raw, n, v); }, false);
    if (rc != G2OHIP_OK) return rc;
    *out = h.release();
    return G2OHIP_OK;
  });
}

void g2ohip_destroy(g2ohip_solver* s) { delete s; }

int g2ohip_set_stream(g2ohip_solver* s, void* hip_stream) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_stream((hipStream_t)hip_stream);
    return G2OHIP_OK;
  });
}

int g2ohip_init(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->init();
    return G2OHIP_OK;
  });
}

int g2ohip_add_edge_set(g2ohip_solver* s, int error_dim, int n_edges, const int32_t* v0, const int32_t* v1) {
  REQUIRE_HANDLE(s);
  return guarded([&] { return s->impl->add_edge_set(error_dim, n_edges, v0, v1); });
}

int g2ohip_set_edge_set_parts(g2ohip_solver* s, int set, int parts) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_edge_set_parts(set, parts);
    return G2OHIP_OK;
  });
}

int g2ohip_build_structure(g2ohip_solver* s, int num_poses, int num_landmarks, int do_schur) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->build_structure(num_poses, num_landmarks, do_schur != 0);
    return G2OHIP_OK;
  });
}

int g2ohip_set_edge_data(g2ohip_solver* s, int set, const double* J0, const double* J1, const double* omega, const double* err,
                         int on_device) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_edge_data(set, J0, J1, omega, err, on_device != 0);
    return G2OHIP_OK;
  });
}

int g2ohip_set_edge_errors(g2ohip_solver* s, int set, const double* err) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_edge_errors(set, err);
    return G2OHIP_OK;
  });
}

int g2ohip_set_robust_kernel(g2ohip_solver* s, int set, int kind, double delta) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_robust_kernel(set, kind, delta);
    return G2OHIP_OK;
  });
}

int g2ohip_set_robust_kernel_per_edge(g2ohip_solver* s, int set, const int32_t* kind, const double* delta) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_robust_kernel_per_edge(set, kind, delta);
    return G2OHIP_OK;
  });
}

int g2ohip_build_system(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->build_system();
    return G2OHIP_OK;
  });
}

int g2ohip_chi2(g2ohip_solver* s, double* chi2) {
  REQUIRE_HANDLE(s);
  if (!chi2) return G2OHIP_ERR_ARG;
  return guarded([&] {
    *chi2 = s->impl->chi2();
    return G2OHIP_OK;
  });
}

int g2ohip_set_lambda(g2ohip_solver* s, double lambda, int backup) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_lambda(lambda, backup != 0);
    return G2OHIP_OK;
  });
}

int g2ohip_set_lambda_split(g2ohip_solver* s, double lambda_pose, double lambda_landmark, int backup) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_lambda_split(lambda_pose, lambda_landmark, backup != 0);
    return G2OHIP_OK;
  });
}

int g2ohip_add_schur_pattern(g2ohip_solver* s, int n_blocks, const int32_t* rows, const int32_t* cols) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->add_schur_pattern(n_blocks, rows, cols);
    return G2OHIP_OK;
  });
}

int g2ohip_clear_edge_sets(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->clear_edge_sets();
    return G2OHIP_OK;
  });
}

int g2ohip_update_structure(g2ohip_solver* s, int num_new_poses, int set, int n_new_edges, const int32_t* v0, const int32_t* v1) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    if (!s->impl->update_structure(num_new_poses, set, n_new_edges, v0, v1)) {
      set_error("updateStructure(): Schur not supported");   // (the reference's message, block_solver.hpp:314)
      return G2OHIP_ERR_UNSUPPORTED;
    }
    return G2OHIP_OK;
  });
}

int g2ohip_restore_diagonal(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->restore_diagonal();
    return G2OHIP_OK;
  });
}

int g2ohip_max_diagonal(g2ohip_solver* s, double* out) {
  REQUIRE_HANDLE(s);
  if (!out) return G2OHIP_ERR_ARG;
  return guarded([&] {
    *out = s->impl->max_diagonal();
    return G2OHIP_OK;
  });
}

int g2ohip_compute_scale(g2ohip_solver* s, double lambda, double* out) {
  REQUIRE_HANDLE(s);
  if (!out) return G2OHIP_ERR_ARG;
  return guarded([&] {
    *out = s->impl->compute_scale(lambda);
    return G2OHIP_OK;
  });
}

int g2ohip_solve(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] { return s->impl->solve() ? G2OHIP_NOT_PD : G2OHIP_OK; });
}
int g2ohip_solve_schur(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->solve_schur();
    return G2OHIP_OK;
  });
}
int g2ohip_solve_reduced(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] { return s->impl->solve_reduced() ? G2OHIP_NOT_PD : G2OHIP_OK; });
}
int g2ohip_set_partition(g2ohip_solver* s, int rank, int world) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_partition(rank, world);
    return G2OHIP_OK;
  });
}
int g2ohip_solve_reduced_local(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->solve_reduced_local();
    return G2OHIP_OK;
  });
}
int g2ohip_solve_reduced_shared(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->solve_reduced_shared();
    return G2OHIP_OK;
  });
}
int g2ohip_solve_reduced_finish(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] { return s->impl->solve_reduced_finish() ? G2OHIP_NOT_PD : G2OHIP_OK; });
}
int g2ohip_schur_operator_prepare(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->schur_operator_prepare();
    return G2OHIP_OK;
  });
}
int g2ohip_schur_operator_apply(g2ohip_solver* s, const double* in_device, double* out_device) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    if (!in_device || !out_device) throw g2ohip::ArgFailure("g2ohip_schur_operator_apply: null vector");
    s->impl->schur_operator_apply(in_device, out_device);
    return G2OHIP_OK;
  });
}
int g2ohip_solve_async(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->solve_async();
    return G2OHIP_OK;
  });
}
int g2ohip_trial_stats_begin(g2ohip_solver* s, double lambda) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->trial_stats_begin(lambda);
    return G2OHIP_OK;
  });
}
int g2ohip_trial_stats(g2ohip_solver* s, double lambda, int* solve_ok, double* chi2, double* scale) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    if (!solve_ok || !chi2 || !scale) throw g2ohip::ArgFailure("g2ohip_trial_stats: null output");
    s->impl->trial_stats(lambda, solve_ok, chi2, scale);
    return G2OHIP_OK;
  });
}
int g2ohip_solve_reduced_finish_async(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->solve_reduced_finish_async();
    return G2OHIP_OK;
  });
}
int g2ohip_exchange_setup(g2ohip_solver* s, int n_blocks, const int32_t* block_idx, const double* block_keep, int n_poses,
                          const int32_t* pose_idx, const double* pose_keep, int n_halo, const int32_t* halo_idx, const double* halo_mine) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    if ((n_blocks > 0 && (!block_idx || !block_keep)) || (n_poses > 0 && (!pose_idx || !pose_keep)) ||
        (n_halo > 0 && (!halo_idx || !halo_mine)))
      throw g2ohip::ArgFailure("g2ohip_exchange_setup: null array");
    s->impl->exchange_setup(n_blocks, block_idx, block_keep, n_poses, pose_idx, pose_keep, n_halo, halo_idx, halo_mine);
    return G2OHIP_OK;
  });
}
int g2ohip_exchange_pack(g2ohip_solver* s, int which) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->exchange_pack(which);
    return G2OHIP_OK;
  });
}
int g2ohip_exchange_unpack(g2ohip_solver* s, int which) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->exchange_unpack(which);
    return G2OHIP_OK;
  });
}
int g2ohip_exchange_status(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    const int rc = s->impl->exchange_status();
    return rc == 0 ? G2OHIP_OK : (rc == 2 ? G2OHIP_REPEAT : G2OHIP_NOT_PD);
  });
}
int g2ohip_get_partition(g2ohip_solver* s, int32_t* pose_owner, int32_t* block_consumer) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->partition_info(pose_owner, block_consumer);
    return G2OHIP_OK;
  });
}
// Host-only: partition of the reduced system's block columns over `world` ranks (no device needed).
int g2ohip_partition_poses(const g2ohip_solver* options_from, int block_dim, int n_blocks, const int32_t* colptr, const int32_t* rowidx,
                           int world, int32_t* pose_owner, int32_t* block_consumer) {
  if (!colptr || !rowidx || !pose_owner || n_blocks <= 0 || world < 1) return G2OHIP_ERR_ARG;
  return guarded([&] {
    SparseCholesky chol(block_dim);
    if (options_from) chol.opt = options_from->impl->chol_opt;
    chol.opt.rank = 0;
    chol.opt.world = world;
    chol.analyze(n_blocks, colptr, rowidx, nullptr, /*host_only=*/true);
    const CholSymbolic& S = chol.symbolic();
    std::copy(S.pose_owner.begin(), S.pose_owner.end(), pose_owner);
    if (block_consumer) std::copy(S.block_consumer.begin(), S.block_consumer.end(), block_consumer);
    return G2OHIP_OK;
  });
}
int g2ohip_solve_back_substitute(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->solve_back_substitute();
    return G2OHIP_OK;
  });
}

size_t g2ohip_vector_size(g2ohip_solver* s) { return (s && s->impl) ? s->impl->vector_size() : 0; }

int g2ohip_copy_x(g2ohip_solver* s, double* x_host) {
  REQUIRE_HANDLE(s);
  if (!x_host) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->copy_x(x_host);
    return G2OHIP_OK;
  });
}
int g2ohip_copy_b(g2ohip_solver* s, double* b_host) {
  REQUIRE_HANDLE(s);
  if (!b_host) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->copy_b(b_host);
    return G2OHIP_OK;
  });
}
const double* g2ohip_x_device(g2ohip_solver* s) { return (s && s->impl) ? s->impl->x_device() : nullptr; }
const double* g2ohip_b_device(g2ohip_solver* s) { return (s && s->impl) ? s->impl->b_device() : nullptr; }

int g2ohip_multiply_hessian(g2ohip_solver* s, double* dest_host, const double* src_host) {
  REQUIRE_HANDLE(s);
  if (!dest_host || !src_host) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->multiply_hessian(dest_host, src_host);
    return G2OHIP_OK;
  });
}

int g2ohip_sync(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->sync();
    return G2OHIP_OK;
  });
}

int g2ohip_set_profiling(g2ohip_solver* s, int enabled) {
  REQUIRE_HANDLE(s);
  // 0: off; 1: every kernel slot + the stage timers of g2ohip_get_stats; 2 + k: kernel slot k only
  s->impl->profiling = enabled == 1;
  s->impl->prof.enabled = enabled != 0;
  s->impl->prof.only = enabled >= 2 ? enabled - 2 : -1;
  return G2OHIP_OK;
}

int g2ohip_kernel_time(g2ohip_solver* s, int slot, double* total_seconds, long* launches, int reset) {
  REQUIRE_HANDLE(s);
  if (slot < 0 || slot >= KernelProf::kNumSlots || !total_seconds || !launches) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->prof.collect();
    *total_seconds = s->impl->prof.total[slot];
    *launches = s->impl->prof.launches[slot];
    if (reset) {
      s->impl->prof.total[slot] = 0;
      s->impl->prof.launches[slot] = 0;
    }
    return G2OHIP_OK;
  });
}
const char* g2ohip_kernel_name(int slot) { return KernelProf::name(slot); }
int g2ohip_kernel_slots(void) { return KernelProf::kNumSlots; }

static void fill_chol_stats(const CholStats* cs, g2ohip_stats* out) {
  if (!cs) return;
  out->timeSymbolicDecomposition = cs->t_symbolic;
  out->choleskyNNZ = cs->nnzL;
  out->numFronts = cs->n_fronts;
  out->numLevels = cs->n_levels;
  out->maxFrontDim = cs->max_front_dim;
  out->bandChains = cs->n_band;
  out->bandCholeskyNNZ = cs->nnzL_band;
  out->bandPivots = cs->piv_band;
  out->treeBackwardGroups = cs->n_tree_groups;
  out->choleskyFlops = cs->flops;
}

int g2ohip_get_stats(g2ohip_solver* s, g2ohip_stats* out) {
  REQUIRE_HANDLE(s);
  if (!out) return G2OHIP_ERR_ARG;
  std::memset(out, 0, sizeof(*out));
  BlockSolver& b = *s->impl;
  out->timeQuadraticForm = b.times.quadratic;
  out->timeSchurComplement = b.times.schur;
  out->timeNumericDecomposition = b.times.numeric;
  out->timeLinearSolution = b.times.linsolve;
  out->timeLinearSolver = b.times.numeric + b.times.linsolve;
  out->timeBackSubstitution = b.times.backsub;
  out->hessianPoseDimension = (size_t)b.nP() * b.p();
  out->hessianLandmarkDimension = (size_t)b.nL() * b.l();
  out->hessianDimension = out->hessianPoseDimension + out->hessianLandmarkDimension;
  fill_chol_stats(b.chol_stats(), out);
  out->iterationsLinearSolver = (size_t)b.pcg_iterations;
  out->timeResiduals = b.times.residuals;
  out->timeLinearize = b.times.linearize;
  out->timeUpdate = b.times.update;
  out->dependencyFallbacks = b.dependency_fallbacks;
  out->shardedCollectives = b.sharded_collectives;
  return G2OHIP_OK;
}

int g2ohip_set_option(g2ohip_solver* s, const char* name, double value) {
  REQUIRE_HANDLE(s);
  if (!name) return G2OHIP_ERR_ARG;
  if (!std::strcmp(name, "nd_leaf")) s->impl->chol_opt.nd_leaf = (int)value;
  else if (!std::strcmp(name, "max_sn_scalars")) s->impl->chol_opt.max_sn_scalars = (int)value;
  else if (!std::strcmp(name, "max_sn_scalars_lds")) s->impl->chol_opt.max_sn_scalars_lds = (int)value;
  else if (!std::strcmp(name, "dep_levels")) s->impl->chol_opt.dep_levels = (int)value;
  else if (!std::strcmp(name, "band_kernel")) s->impl->chol_opt.band_kernel = (int)value;
  else if (!std::strcmp(name, "tree_backward")) s->impl->chol_opt.tree_backward = (int)value;
  else if (!std::strcmp(name, "big_group")) s->impl->chol_opt.big_group = (int)value;
  else if (!std::strcmp(name, "big_group_min_rows")) s->impl->chol_opt.big_group_min_rows = (int)value;
  else if (!std::strcmp(name, "dep_backward")) s->impl->chol_opt.dep_backward = (int)value;
  else if (!std::strcmp(name, "big_front_passes")) s->impl->chol_opt.big_front_passes = (int)value;
  else if (!std::strcmp(name, "dep_spin_limit")) s->impl->chol_opt.dep_spin_limit = (int)value;
  else if (!std::strcmp(name, "schur_tile_bytes")) s->impl->schur_tile_bytes = (size_t)value;
  else if (!std::strcmp(name, "comm_emulate")) s->impl->comm_emulate = (int)value;
  else if (!std::strcmp(name, "mask_solution")) s->impl->mask_solution = value != 0;
  else if (!std::strcmp(name, "linear_solver")) s->impl->linear_solver = (int)value;       // 0 Cholesky, 1 PCG
  else if (!std::strcmp(name, "pcg_tolerance")) s->impl->pcg_opt.tolerance = value;
  else if (!std::strcmp(name, "pcg_max_iterations")) s->impl->pcg_opt.max_iter = (int)value;
  else if (!std::strcmp(name, "pcg_absolute_tolerance")) s->impl->pcg_opt.absolute_tolerance = value != 0;
  else if (!std::strcmp(name, "pcg_check_every")) s->impl->pcg_opt.check_every = std::max(1, (int)value);
  else if (!std::strcmp(name, "fuse_schur_reduce")) s->impl->fuse_schur_reduce = value != 0;
  else if (!std::strcmp(name, "marginals_reduced")) s->impl->marginals_reduced = value != 0;
  else if (!std::strcmp(name, "marginals_recursion")) s->impl->marginals_recursion = value != 0;
  else if (!std::strcmp(name, "use_graph")) s->impl->use_graph = value != 0;
  else if (!std::strcmp(name, "sharded_graph")) s->impl->sharded_graph = (int)value;
  else if (!std::strcmp(name, "sharded_merge")) s->impl->sharded_merge = (int)value;
  else if (!std::strcmp(name, "sharded_selftest")) s->impl->sharded_selftest = (int)value;
  else if (!std::strcmp(name, "setup_overlap")) s->impl->setup_overlap = value != 0.0;
  else if (!std::strcmp(name, "sharded_selftest_break")) s->impl->selftest_break = (int)value;   // (tests: corrupt the first variant solve)
  else if (!std::strcmp(name, "ba_fused")) s->impl->ba_fused = value != 0;
  else if (!std::strcmp(name, "ba_fuse_landmarks")) s->impl->ba_fuse_landmarks = value != 0;
  else {
    set_error(std::string("unknown option ") + name);
    return G2OHIP_ERR_ARG;
  }
  s->impl->invalidate_graphs();   // options change kernel arguments / launch shapes
  return G2OHIP_OK;
}

int g2ohip_get_nnzb(g2ohip_solver* s, int which, int* nnzb) {
  REQUIRE_HANDLE(s);
  if (!nnzb) return G2OHIP_ERR_ARG;
  return guarded([&] {
    *nnzb = s->impl->nnzb(which);
    return G2OHIP_OK;
  });
}
int g2ohip_get_pattern(g2ohip_solver* s, int which, int32_t* colptr, int32_t* rowidx) {
  REQUIRE_HANDLE(s);
  if (!colptr || !rowidx) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->get_pattern(which, colptr, rowidx);
    return G2OHIP_OK;
  });
}
int g2ohip_copy_values(g2ohip_solver* s, int which, double* values_host) {
  REQUIRE_HANDLE(s);
  if (!values_host) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->copy_values(which, values_host);
    return G2OHIP_OK;
  });
}
int g2ohip_device_array(g2ohip_solver* s, int which, double** ptr, size_t* count) {
  REQUIRE_HANDLE(s);
  if (!ptr || !count) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->device_array(which, ptr, count);
    return G2OHIP_OK;
  });
}

// ---- page-locked host buffers -------------------------------------------------------------
int g2ohip_host_register(g2ohip_solver* s, void* ptr, size_t bytes) {
  REQUIRE_HANDLE(s);
  if (!ptr || !bytes) return G2OHIP_ERR_ARG;
  return guarded([&] {
    G2OHIP_HIP_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return G2OHIP_OK;
  });
}
int g2ohip_host_unregister(g2ohip_solver* s, void* ptr) {
  REQUIRE_HANDLE(s);
  if (!ptr) return G2OHIP_ERR_ARG;
  return guarded([&] {
    G2OHIP_HIP_CHECK(hipHostUnregister(ptr));
    return G2OHIP_OK;
  });
}

// ---- device-resident bundle-adjustment front end ---------------------------------------
int g2ohip_ba_set_edges(g2ohip_solver* s, int set, const int32_t* cam_vertex, const int32_t* point_vertex, const double* meas,
                        const double* info, double f, double cx, double cy) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_set_edges(set, cam_vertex, point_vertex, meas, info, f, cx, cy);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_set_edges_classes(g2ohip_solver* s, int set, const int32_t* cam_vertex, const int32_t* point_vertex, const double* meas,
                                const double* info, int n_classes, const double* class_params, const int32_t* edge_class) {
  REQUIRE_HANDLE(s);
  if (n_classes < 1 || !class_params) {
    set_error("ba_set_edges_classes: at least one class with its parameters");
    return G2OHIP_ERR_ARG;
  }
  return guarded([&] {
    s->impl->ba_set_edges_classes(set, cam_vertex, point_vertex, meas, info, class_params[0], class_params[1], class_params[2], n_classes,
                                  class_params, edge_class);
    if (n_classes == 1) s->impl->set_robust_kernel(set, (int)class_params[3], class_params[4]);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_set_estimates(g2ohip_solver* s, int n_cams, const double* cams, const int32_t* cam_hidx, int n_points,
                            const double* points, const int32_t* point_hidx) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_set_estimates(n_cams, cams, cam_hidx, n_points, points, point_hidx);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_get_estimates(g2ohip_solver* s, double* cams, double* points) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_get_estimates(cams, points);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_get_estimates_of(g2ohip_solver* s, int n_cams, const int32_t* cam_index, double* cams, int n_points, const int32_t* point_index,
                               double* points) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_get_estimates_of(n_cams, cam_index, cams, n_points, point_index, points);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_fetch_estimates_begin(g2ohip_solver* s, double* cams, double* points, int point_pieces) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_fetch_begin(cams, points, point_pieces);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_fetch_estimates_wait(g2ohip_solver* s, int piece) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_fetch_wait(piece);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_linearize(g2ohip_solver* s, int jacobians) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_linearize(jacobians != 0);
    return G2OHIP_OK;
  });
}
int g2ohip_ba_update(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_update();
    return G2OHIP_OK;
  });
}
int g2ohip_ba_push(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_push();
    return G2OHIP_OK;
  });
}
int g2ohip_ba_pop(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_pop();
    return G2OHIP_OK;
  });
}
int g2ohip_ba_discard_top(g2ohip_solver* s) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->ba_discard_top();
    return G2OHIP_OK;
  });
}

int g2ohip_comm_unique_id(char* id128) {
  if (!id128) return G2OHIP_ERR_ARG;
  return guarded([&] { Comm::unique_id(id128); return G2OHIP_OK; });
}
int g2ohip_comm_init_rccl(g2ohip_solver* s, int rank, int world, const char* id128) {
  if (!s || !id128 || world < 1 || rank < 0 || rank >= world) return G2OHIP_ERR_ARG;
  return guarded([&] { s->impl->comm_init_rccl(rank, world, id128); return G2OHIP_OK; });
}
int g2ohip_comm_init_host(g2ohip_solver* s, int rank, int world, g2ohip_host_allreduce_fn fn, void* ctx) {
  if (!s || !fn || world < 1 || rank < 0 || rank >= world) return G2OHIP_ERR_ARG;
  return guarded([&] { s->impl->comm.init_host(rank, world, fn, ctx); return G2OHIP_OK; });
}
int g2ohip_comm_init_peer(g2ohip_solver* s, int rank, int world, g2ohip_host_allreduce_fn fn, void* ctx, size_t slot_doubles) {
  if (!s || !fn || world < 1 || rank < 0 || rank >= world) return G2OHIP_ERR_ARG;
  return guarded([&] { s->impl->comm_init_peer(rank, world, fn, ctx, slot_doubles); return G2OHIP_OK; });
}
int g2ohip_comm_destroy(g2ohip_solver* s) {
  if (!s) return G2OHIP_ERR_ARG;
  return guarded([&] { s->impl->comm.destroy(); return G2OHIP_OK; });
}
int g2ohip_comm_all_reduce(g2ohip_solver* s, double* device_buffer, size_t count, int op) {
  if (!s || (count && !device_buffer)) return G2OHIP_ERR_ARG;
  return guarded([&] { s->impl->comm_all_reduce(device_buffer, count, op); return G2OHIP_OK; });
}
int g2ohip_solve_sharded(g2ohip_solver* s) {
  if (!s) return G2OHIP_ERR_ARG;
  return guarded([&] { return s->impl->solve_sharded() ? G2OHIP_NOT_PD : G2OHIP_OK; });
}
int g2ohip_chi2_sharded(g2ohip_solver* s, double* chi2) {
  if (!s || !chi2) return G2OHIP_ERR_ARG;
  return guarded([&] { *chi2 = s->impl->chi2_sharded(); return G2OHIP_OK; });
}
int g2ohip_max_diagonal_sharded(g2ohip_solver* s, double* out) {
  if (!s || !out) return G2OHIP_ERR_ARG;
  return guarded([&] { *out = s->impl->max_diagonal_sharded(); return G2OHIP_OK; });
}
int g2ohip_compute_scale_sharded(g2ohip_solver* s, double lambda, double* out) {
  if (!s || !out) return G2OHIP_ERR_ARG;
  return guarded([&] { *out = s->impl->compute_scale_sharded(lambda); return G2OHIP_OK; });
}
int g2ohip_copy_diagonal(g2ohip_solver* s, double* diag_host) {
  if (!s) return G2OHIP_ERR_ARG;
  return guarded([&] { s->impl->copy_diagonal(diag_host); return G2OHIP_OK; });
}
int g2ohip_compute_marginals(g2ohip_solver* s, int n_blocks, const int32_t* rows, const int32_t* cols, double* out) {
  REQUIRE_HANDLE(s);
  return guarded([&] { return s->impl->compute_marginals(n_blocks, rows, cols, out) ? G2OHIP_NOT_PD : G2OHIP_OK; });
}
int g2ohip_set_x(g2ohip_solver* s, const double* x_host) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->set_x(x_host);
    return G2OHIP_OK;
  });
}
int g2ohip_copy_edge_data(g2ohip_solver* s, int set, double* J0, double* J1, double* err) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->copy_edge_data(set, J0, J1, err);
    return G2OHIP_OK;
  });
}
int g2ohip_pg_set_edges(g2ohip_solver* s, int set, int type, const int32_t* vi, const int32_t* vj, const double* meas,
                        const double* info) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->pg_set_edges(set, type, vi, vj, meas, info);
    return G2OHIP_OK;
  });
}
int g2ohip_pg_set_estimates(g2ohip_solver* s, int n_vertices, const double* poses, const int32_t* hidx) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->pg_set_estimates(n_vertices, poses, hidx);
    return G2OHIP_OK;
  });
}
int g2ohip_pg_get_estimates(g2ohip_solver* s, double* poses) {
  REQUIRE_HANDLE(s);
  if (!poses) return G2OHIP_ERR_ARG;
  return guarded([&] {
    s->impl->pg_get_estimates(poses);
    return G2OHIP_OK;
  });
}
int g2ohip_pg_linearize(g2ohip_solver* s, int jacobians) {
  REQUIRE_HANDLE(s);
  return guarded([&] {
    s->impl->pg_linearize(jacobians != 0);
    return G2OHIP_OK;
  });
}
#define G2OHIP_PG_SIMPLE(NAME, METHOD)   \
  int NAME(g2ohip_solver* s) {           \
    REQUIRE_HANDLE(s);                   \
    return guarded([&] {                 \
      s->impl->METHOD();                 \
      return G2OHIP_OK;                  \
    });                                  \
  }
G2OHIP_PG_SIMPLE(g2ohip_pg_update, pg_update)
G2OHIP_PG_SIMPLE(g2ohip_pg_push, pg_push)
G2OHIP_PG_SIMPLE(g2ohip_pg_pop, pg_pop)
G2OHIP_PG_SIMPLE(g2ohip_pg_discard_top, pg_discard_top)
#undef G2OHIP_PG_SIMPLE

// ---- narrow seam ---------------------------------------------------------------------
int g2ohip_ls_create(g2ohip_linear_solver** out, int block_dim, int device) {
  if (!out) return G2OHIP_ERR_ARG;
  *out = nullptr;
  return guarded([&] {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw HipFailure("no HIP device available: libg2ohip has no CPU fallback");
    if (device < 0 || device >= count) throw ArgFailure("bad device ordinal");
    if (!(block_dim == 3 || block_dim == 6 || block_dim == 7)) throw ArgFailure("block_dim must be 3, 6 or 7");
    auto h = std::make_unique<g2ohip_linear_solver>();
    h->bs = block_dim;
    h->device = device;
    G2OHIP_HIP_CHECK(hipSetDevice(device));
    G2OHIP_HIP_CHECK(hipStreamCreate(&h->st));
    g2ohip_linear_solver* raw = h.get();
    const int rc = apply_env_options([&](const char* n, double v) { return g2ohip_ls_set_option(raw, n, v); }, true);
    if (rc != G2OHIP_OK) {
      (void)hipStreamDestroy(h->st);
      return rc;
    }
    *out = h.release();
    return G2OHIP_OK;
  });
}
void g2ohip_ls_destroy(g2ohip_linear_solver* ls) {
  if (!ls) return;
  if (ls->st) (void)hipStreamDestroy(ls->st);
  delete ls;
}
int g2ohip_ls_init(g2ohip_linear_solver* ls) {
  if (!ls) return G2OHIP_ERR_ARG;
  ls->chol.reset();
  ls->colptr.clear();
  ls->rowidx.clear();
  ls->chol2.reset();
  ls->colptr2.clear();
  ls->rowidx2.clear();
  return G2OHIP_OK;
}
namespace {
bool same_pattern(const std::vector<int>& cp, const std::vector<int>& ri, int n_blocks, const int32_t* colptr, const int32_t* rowidx) {
  return (int)cp.size() == n_blocks + 1 && std::memcmp(cp.data(), colptr, sizeof(int) * (n_blocks + 1)) == 0 && (int)ri.size() == colptr[n_blocks] &&
         std::memcmp(ri.data(), rowidx, sizeof(int) * ri.size()) == 0;
}
void analyze_into(g2ohip_linear_solver* ls, std::unique_ptr<SparseCholesky>& chol, std::vector<int>& cp, std::vector<int>& ri, int n_blocks,
                  const int32_t* colptr, const int32_t* rowidx) {
  chol = std::make_unique<SparseCholesky>(ls->bs);
  chol->opt = ls->opt;
  cp.assign(colptr, colptr + n_blocks + 1);
  ri.assign(rowidx, rowidx + colptr[n_blocks]);
  chol->analyze(n_blocks, colptr, rowidx, ls->st);
}
// factorise; a dependency-driven launch that gave up waiting is not "not positive definite": once more, level by level
bool factor_checked(SparseCholesky& chol, const double* dA, hipStream_t st) {
  chol.factor(dA, st);
  bool bad = chol.failed(st);
  if (bad && chol.dependency_stall()) {
    chol.factor(dA, st);
    bad = chol.failed(st);
  }
  return !bad;
}
}  // namespace
int g2ohip_ls_solve(g2ohip_linear_solver* ls, int n_blocks, const int32_t* colptr, const int32_t* rowidx, const double* values,
                    double* x, const double* b) {
  if (!ls || !colptr || !rowidx || !values || !x || !b || n_blocks <= 0) return G2OHIP_ERR_ARG;
  return guarded([&] {
    G2OHIP_HIP_CHECK(hipSetDevice(ls->device));
    const int nnzb = colptr[n_blocks];
    const int bs = ls->bs;
    // first call after init(): symbolic factorisation (linear_solver_csparse.h:110-112).  The pattern has to stay the same
    // until the next init() (linear_solver.h:86-105); a different one is analysed anew instead of being factorised with a
    // stale structure
    if (!ls->chol || !same_pattern(ls->colptr, ls->rowidx, n_blocks, colptr, rowidx)) analyze_into(ls, ls->chol, ls->colptr, ls->rowidx, n_blocks, colptr, rowidx);
    const size_t n = (size_t)n_blocks * bs;
    ls->dA.upload(values, (size_t)nnzb * bs * bs, ls->st);
    ls->db.upload(b, n, ls->st);
    ls->dx.alloc(n);
    ls->tn.start(ls->st);
    ls->chol->factor(ls->dA.p, ls->st);
    ls->tn.stop(ls->st);
    ls->tl.start(ls->st);
    ls->chol->solve(ls->db.p, ls->dx.p, ls->st);
    ls->tl.stop(ls->st);
    bool bad = ls->chol->failed(ls->st);
    if (bad && ls->chol->dependency_stall()) {   // a dependency-driven launch gave up waiting (not "not positive definite"):
      ls->chol->factor(ls->dA.p, ls->st);        // the solver has switched to one launch per level -- once more
      ls->chol->solve(ls->db.p, ls->dx.p, ls->st);
      bad = ls->chol->failed(ls->st);
    }
    ls->t_numeric = ls->tn.seconds();
    ls->t_solve = ls->tl.seconds();
    if (bad) return G2OHIP_NOT_PD;
    ls->dx.download(x, n, ls->st);
    return G2OHIP_OK;
  });
}
namespace {
__global__ void ls_gather_inverse_kernel(int n, int p, const long long* __restrict__ off, const int* __restrict__ ld,
                                         const int* __restrict__ tr, const double* __restrict__ Z, double* __restrict__ out) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= (size_t)n * p * p) return;
  const int b = (int)(t / (p * p)), e = (int)(t % (p * p)), i = e % p, j = e / p;
  if (off[b] < 0) return;
  out[t] = tr[b] ? Z[off[b] + j + (long long)ld[b] * i] : Z[off[b] + i + (long long)ld[b] * j];
}
__global__ void ls_unit_kernel(double* v, size_t n, size_t k) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t < n) v[t] = t == k ? 1.0 : 0.0;
}
}  // namespace

int g2ohip_ls_solve_pattern(g2ohip_linear_solver* ls, int n_blocks, const int32_t* colptr, const int32_t* rowidx, const double* values,
                            int n_req, const int32_t* rows, const int32_t* cols, double* out) {
  if (!ls || !colptr || !rowidx || !values || n_blocks <= 0 || n_req < 0 || (n_req > 0 && (!rows || !cols || !out))) return G2OHIP_ERR_ARG;
  for (int i = 0; i < n_req; ++i)
    if (rows[i] < 0 || rows[i] >= n_blocks || cols[i] < 0 || cols[i] >= n_blocks) return G2OHIP_ERR_ARG;
  return guarded([&] {
    G2OHIP_HIP_CHECK(hipSetDevice(ls->device));
    const int nnzb = colptr[n_blocks];
    const int bs = ls->bs;
    // the pattern of solve() -> its analysis; another pattern (BlockSolver::computeMarginals hands Hpp to the LinearSolver
    // that factorised Hschur, without init(): block_solver.hpp:489-499) -> a second analysis kept next to it
    SparseCholesky* ch = nullptr;
    if (!ls->chol) analyze_into(ls, ls->chol, ls->colptr, ls->rowidx, n_blocks, colptr, rowidx);
    if (same_pattern(ls->colptr, ls->rowidx, n_blocks, colptr, rowidx)) {
      ch = ls->chol.get();
    } else {
      if (!ls->chol2 || !same_pattern(ls->colptr2, ls->rowidx2, n_blocks, colptr, rowidx)) analyze_into(ls, ls->chol2, ls->colptr2, ls->rowidx2, n_blocks, colptr, rowidx);
      ch = ls->chol2.get();
    }
    ls->dA.upload(values, (size_t)nnzb * bs * bs, ls->st);
    if (!factor_checked(*ch, ls->dA.p, ls->st)) return G2OHIP_NOT_PD;
    if (n_req == 0) return G2OHIP_OK;
    ch->sparse_inverse(ls->st);
    std::vector<long long> off(n_req, -1);
    std::vector<int> ldv(n_req, 0), trv(n_req, 0);
    int found = 0;
    for (int i = 0; i < n_req; ++i) {
      bool tr = false;
      if (ch->inverse_block(rows[i], cols[i], &off[i], &ldv[i], &tr)) {
        trv[i] = tr ? 1 : 0;
        ++found;
      } else {
        off[i] = -1;
      }
    }
    const size_t pp = (size_t)bs * bs;
    if (found > 0) {
      DevBuf<long long> d_off;
      DevBuf<int> d_ld, d_tr;
      DevBuf<double> d_out;
      d_off.upload(off, ls->st);
      d_ld.upload(ldv, ls->st);
      d_tr.upload(trv, ls->st);
      d_out.alloc((size_t)n_req * pp);
      const size_t total = (size_t)n_req * pp;
      hipLaunchKernelGGL(ls_gather_inverse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ls->st, n_req, bs, d_off.p, d_ld.p, d_tr.p,
                         ch->inverse_slab(), d_out.p);
      std::vector<double> ho(total);
      d_out.download(ho.data(), total, ls->st);
      for (int i = 0; i < n_req; ++i)
        if (off[i] >= 0) std::copy(ho.begin() + (size_t)i * pp, ho.begin() + (size_t)(i + 1) * pp, out + (size_t)i * pp);
    }
    if (found < n_req) {   // pairs outside the pattern of the factor: a pair of sweeps per column
      const size_t n = (size_t)n_blocks * bs;
      DevBuf<double> rhs, sol;
      rhs.alloc(n);
      sol.alloc(n);
      std::vector<double> h(n);
      for (int i = 0; i < n_req; ++i) {
        if (off[i] >= 0) continue;
        for (int k = 0; k < bs; ++k) {
          hipLaunchKernelGGL(ls_unit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ls->st, rhs.p, n, (size_t)cols[i] * bs + k);
          ch->solve(rhs.p, sol.p, ls->st);
          sol.download(h.data(), n, ls->st);
          for (int r = 0; r < bs; ++r) out[(size_t)i * pp + r + (size_t)bs * k] = h[(size_t)rows[i] * bs + r];
        }
      }
    }
    return G2OHIP_OK;
  });
}
int g2ohip_ls_get_stats(g2ohip_linear_solver* ls, g2ohip_stats* out) {
  if (!ls || !out) return G2OHIP_ERR_ARG;
  std::memset(out, 0, sizeof(*out));
  if (ls->chol) fill_chol_stats(&ls->chol->stats(), out);
  out->timeNumericDecomposition = ls->t_numeric;
  out->timeLinearSolution = ls->t_solve;
  out->timeLinearSolver = ls->t_numeric + ls->t_solve;
  return G2OHIP_OK;
}
int g2ohip_ls_set_option(g2ohip_linear_solver* ls, const char* name, double value) {
  if (!ls || !name) return G2OHIP_ERR_ARG;
  if (!std::strcmp(name, "nd_leaf")) ls->opt.nd_leaf = (int)value;
  else if (!std::strcmp(name, "max_sn_scalars")) ls->opt.max_sn_scalars = (int)value;
  else if (!std::strcmp(name, "max_sn_scalars_lds")) ls->opt.max_sn_scalars_lds = (int)value;
  else if (!std::strcmp(name, "dep_levels")) ls->opt.dep_levels = (int)value;
  else if (!std::strcmp(name, "band_kernel")) ls->opt.band_kernel = (int)value;
  else if (!std::strcmp(name, "tree_backward")) ls->opt.tree_backward = (int)value;
  else if (!std::strcmp(name, "big_group")) ls->opt.big_group = (int)value;
  else if (!std::strcmp(name, "big_group_min_rows")) ls->opt.big_group_min_rows = (int)value;
  else if (!std::strcmp(name, "dep_backward")) ls->opt.dep_backward = (int)value;
  else if (!std::strcmp(name, "big_front_passes")) ls->opt.big_front_passes = (int)value;
  else if (!std::strcmp(name, "dep_spin_limit")) ls->opt.dep_spin_limit = (int)value;
  else return G2OHIP_ERR_ARG;
  return G2OHIP_OK;
}

}  // extern "C"
