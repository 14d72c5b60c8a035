// Device-resident replacement of g2o::BlockSolver<BlockSolverTraits<p,l>>
//   /root/reference/g2o/core/block_solver.h:98-178, block_solver.hpp (all)
// over flat index/Jacobian arrays instead of the pointer-chasing SparseBlockMatrix
// (sparse_block_matrix.h:61-220: vector<map<int,Block*>>, one heap block per tile).
//
// Data layout in HBM (all fp64 column-major blocks, int32 indices):
//   Hpp, Hschur : block-CCS upper triangle, ascending rows per column, values [nnzb][p*p]
//   Hpl         : block-CCS by landmark column (== _HplCCS, block_solver.h:157), rows = pose
//                 index ascending, values [nnzb][p*l]
//   Hll, Dinv   : dense [nL][l*l];  b, x : [nP*p | nL*l]  (block_solver.hpp:551-557)
// Scatter reductions (many edges -> one vertex block, many landmarks -> one Hschur block)
// are done destination-major from precomputed contributor lists: deterministic, no fp64
// atomics, same summation order as the reference's serial loops.
#pragma once
#include <memory>

#include "common.h"
#include "comm.h"
#include "block_pcg.h"
#include "sparse_cholesky.h"

namespace g2ohip {

struct EdgeSet {
  int d = 0, n = 0, dim0 = 0, dim1 = 0;
  bool unary = false;
  int parts = 0;           // G2OHIP_PART_* bits: what of its quadratic form the set does NOT contribute (set_edge_set_parts)
  std::vector<int> v0, v1;
  int kernel_kind = 0;
  double delta = 1.0;
  DevBuf<double> rk;       // per-edge robust kernels [n][2] = (kind, delta), or empty: the set-level pair above (set_robust_kernel_per_edge)
  // per-iteration data (device); either own buffers or caller-owned device pointers
  const double *J0 = nullptr, *J1 = nullptr, *omega = nullptr, *err = nullptr;
  DevBuf<double> own_J0, own_J1, own_omega, own_err;
  bool has_data = false;   // Jacobians + information + errors available for build_system
  bool external = false;   // the arrays belong to the caller (set_edge_data on_device): they may change behind the solver's back
  bool has_err = false;    // errors + information available (enough for chi2)
  // destination-major contributor lists
  DevBuf<int> vp_ptr, vp_ent, vl_ptr, vl_ent;  // per pose / per landmark: (edge << 1 | side)
  DevBuf<int> vp_act;                          // poses with at least one entry (when that is not all of them)
  int n_vp_act = 0;
  std::vector<int> h_vp_ent, h_vl_ent;         // host copies (pose- / landmark-major copies of per-edge inputs)
  std::vector<int> h_vl_ptr;                   // ... and the landmark list bounds (lane slots of the fused Schur tiles)
  DevBuf<int> op_dst, op_ptr, op_ent;          // Hpp off-diagonal blocks: dest block id, (edge << 1 | transposed)
  DevBuf<int> ol_dst, ol_ptr, ol_ent;          // Hpl blocks
  int n_op = 0, n_ol = 0;
  long n_vp_ent = 0, n_vl_ent = 0;
  bool touches_pose = false, touches_lm = false, first_pose = false, first_lm = false, first_op = false, first_ol = false;
};

struct SolverTimes {
  double quadratic = 0, schur = 0, numeric = 0, linsolve = 0, backsub = 0, residuals = 0, linearize = 0, update = 0;
};

class BlockSolver {
 public:
  BlockSolver(int p, int l, int device);
  ~BlockSolver();

  void init();
  void clear_edge_sets();
  int add_edge_set(int d, int n, const int* v0, const int* v1);
  void set_edge_set_parts(int set, int parts);
  void add_schur_pattern(int n, const int* rows, const int* cols);
  void build_structure(int nP, int nL, bool do_schur);
  bool update_structure(int new_poses, int set, int n, const int* v0, const int* v1);
  void set_edge_data(int set, const double* J0, const double* J1, const double* omega, const double* err, bool on_device);
  void set_edge_errors(int set, const double* err);
  void set_robust_kernel(int set, int kind, double delta);
  void set_robust_kernel_per_edge(int set, const int* kind, const double* delta);
  void build_system();
  double chi2();
  void set_lambda(double lambda, bool backup);
  void set_lambda_split(double lambda_pose, double lambda_landmark, bool backup);
  void restore_diagonal();
  double max_diagonal();
  double compute_scale(double lambda);
  int solve();                 // 0 ok, 1 not PD
  void solve_schur();
  int solve_reduced();
  // multi-GPU split of solve_reduced (openslam_g2o_amd/distributed.py): own subtrees -> exchange of the
  // subtree-root update matrices/vectors -> shared top of the tree + back down -> exchange of x_p
  void set_partition(int rank, int world);
  void solve_reduced_local();
  void solve_reduced_shared();
  int solve_reduced_finish();
  void partition_info(int* pose_owner, int* block_consumer);
  int compute_marginals(int n, const int* rows, const int* cols, double* out);
  void copy_diagonal(double* host);
  void copy_edge_data(int set, double* J0, double* J1, double* err);
  void pg_set_edges(int set, int type, const int* vi, const int* vj, const double* meas, const double* info);
  void pg_set_estimates(int nv, const double* poses, const int* hidx);
  void pg_get_estimates(double* poses);
  void pg_linearize(bool jacobians);
  void pg_update();
  void pg_push();
  void pg_pop();
  void pg_discard_top();
  void solve_back_substitute();
  void multiply_hessian(double* dest_host, const double* src_host);

  size_t vector_size() const { return (size_t)nP_ * p_ + (size_t)nL_ * l_; }
  void set_x(const double* h);
  void copy_x(double* h);
  void copy_b(double* h);
  const double* x_device() const { return d_x.p; }
  const double* b_device() {
    ensure_bl();
    return d_b.p;
  }
  void sync();
  void set_stream(hipStream_t st);
  hipStream_t stream() const { return st_; }

  int nnzb(int which) const;
  void get_pattern(int which, int* colptr, int* rowidx) const;
  void copy_values(int which, double* h);
  void device_array(int which, double** ptr, size_t* count);

  // ---- device-resident bundle-adjustment front end: error / Jacobian
  // producers, oplus and the estimate stack for EdgeProjectXYZ2UV graphs, so that a whole LM trial
  // loop needs no host round trip of Jacobians or estimates
  void ba_set_edges(int set, const int* cam_vertex, const int* point_vertex, const double* meas, const double* info, double f,
                    double cx, double cy);
  void ba_set_edges_classes(int set, const int* cam_vertex, const int* point_vertex, const double* meas, const double* info, double f,
                            double cx, double cy, int n_classes, const double* class_params, const int* edge_class);
  void ba_set_estimates(int n_cams, const double* cams, const int* cam_hidx, int n_points, const double* points, const int* point_hidx);
  void ba_get_estimates(double* cams, double* points);
  void ba_get_estimates_of(int n_cams, const int* cam_idx, double* cams, int n_points, const int* point_idx, double* points);   // selected vertices
  void ba_fetch_begin(double* cams, double* points, int point_pieces);   // the same, asynchronous and in pieces (see the definition)
  void ba_fetch_wait(int piece);
  static constexpr int kFetchMaxPieces = 17;
  void ba_linearize(bool jacobians);
  void ba_update();
  void ba_push();
  void ba_pop();
  void ba_discard_top();

  bool profiling = false;
  KernelProf prof;
  SolverTimes times;
  CholOptions chol_opt;
  size_t schur_tile_bytes = 39 * 1024;     // LDS budget of one Schur tile
  int comm_emulate = 0;                    // TIMING ONLY: solve_sharded of one rank of an N-rank job run alone, the all-reduces skipped
                                           // (results are wrong; bench.py --emulate r/N: the per-rank time an N-GPU run is bounded by)
  static constexpr bool rhs_prefill = true;                 // solve(): schur_rhs_kernel also writes the permuted right-hand side and clears the factorisation's status word
  bool rhs_prefilled_ = false;             // ... and has done so for the next solve_reduced_device
  bool ba_fused = true;                    // evaluate BA errors/Jacobians inside the assembly kernels (no J arrays)
  // hipGraph replay of the launch-bound kernel sequences (one launch per tree level: factorisation with the
  // fused forward sweep, backward sweep): ~30 launches per iteration, which is what limits a rank once the
  // per-rank work shrinks (multi-GPU).  Needs a non-default stream.  Timing events sit between the graphs.
  bool use_graph = false;
  static constexpr int sharded_virtual = 1;                 // sharded solve: the factorisation reads Hpp + partial blocks itself (as on one GPU); only the
                                           // boundary blocks of the reduced system are reduced, exchanged and read back as blocks
  int sharded_merge = 1;                   // sharded solve: TWO all-reduces instead of three -- the boundary blocks and b_p travel with the
                                           // subtree roots (after the own subtrees) whenever only the shared top of the tree consumes them
  size_t sharded_collectives = 0;          // all-reduces issued by the last solve_sharded_once (stats)
  bool setup_overlap = true;               // build_structure: the symbolic analysis of the reduced system on a thread of its own next to the Schur tiles' set-up
  int sharded_selftest = 1;                // the first solve_sharded of a structure runs twice (as configured / reference schedule) and falls back to the
                                           // reference schedule when the two disagree (see solve_sharded)
  bool selftest_fallback() const { return selftest_fallback_; }
  int selftest_break = 0;                  // (tests only) rank 0 corrupts its copy of the configured variant's solution: the self-test must fall back on EVERY rank
  int sharded_graph = 1;                   // solve_sharded as ONE hipGraph (the collectives inside it): 1 = when nothing has to cross the
                                           // host (comm_emulate, the peer mailboxes: their two kernels per all-reduce), 2 = with RCCL as well (ncclAllReduce captured into the graph; not
                                           // exercised on hardware here -- opt-in); 0 = one graph per phase, plain launches in between
  void invalidate_graphs();
  int linear_solver = 0;                   // 0: multifrontal block Cholesky, 1: block-Jacobi PCG (LinearSolverPCG) on Hschur,
                                           // 2: the same PCG matrix-free (Schur complement never formed; Schur mode only)
  PcgOptions pcg_opt;
  int pcg_iterations = 0;
  static constexpr bool schur_sort_dests = true;            // order a tile's destinations by entry count (lockstep lane groups)
  int num_cus_ = 256;
  static constexpr bool fuse_landmark_inverse = true;       // invert the landmark blocks inside the Schur tile kernel
  static constexpr bool overlap_assembly = false;           // fused BA assembly: pose-side kernel on a side stream next to the landmark-side one (measured: 0.2 ms SLOWER per iteration, kept as a switch)
  bool fuse_schur_reduce = true;           // solve(): the factorisation assembles its fronts from Hpp and the tiles' partial
                                           // blocks directly; Hschur is only written out when somebody asks for it
  bool tiles_cover_all_ = false;
  bool mask_solution = true;               // solve_reduced_shared zeroes the x_p entries other ranks own (all-reduce of x_p)
  static constexpr int schur_group = 0;                     // lanes per destination in the Schur tile kernel (0 = auto)
  const CholStats* chol_stats() const { return chol_ ? &chol_->stats() : nullptr; }
  int p() const { return p_; }
  int l() const { return l_; }
  int nP() const { return nP_; }
  int nL() const { return nL_; }
  bool schur() const { return schur_; }

 private:
  int p_, l_, device_;
  int nP_ = 0, nL_ = 0;
  bool schur_ = false, structured_ = false, system_built_ = false;
  hipStream_t st_ = nullptr;
  bool own_stream_ = false;
  std::vector<std::unique_ptr<EdgeSet>> sets_;
  std::vector<std::pair<int, int>> extra_hs_;  // extra structural blocks (row, col) of the reduced system
  // host patterns
  std::vector<int> pp_colptr, pp_row, pp_diag, pl_colptr, pl_row, hs_colptr, hs_row;
  // device matrices
  enum Seg { kSegFactor = 0, kSegBackward, kSegLocal, kSegShared, kSegSharedBack, kSegFactorBand, kSegFactorRest, kSegShardedAll, kSegFactorPre, kNumSeg };
  bool in_outer_seg_ = false;   // an enclosing segment is being captured / run: inner run_seg calls are transparent
  struct GraphSeg {
    hipGraph_t g = nullptr;
    hipGraphExec_t e = nullptr;
    int state = 0;   // 0: run plainly once (first-use initialisation), 1: capture, 2: replay
  };
  GraphSeg segs_[kNumSeg];
  template <class F>
  void run_seg(int id, F&& body);
  void build_system_impl();
  void solve_schur_impl(bool want_matrix = true);
  void launch_schur_reduce(bool matrix);
  void ensure_hschur();
  bool virtual_reduced_ok();
  void solve_reduced_device();
  int solve_reduced_impl();
 public:
  // Multi-GPU halo exchange without host round trips (openslam_g2o_amd/distributed.py): index lists and keep-masks
  // live on the device, pack / unpack are single kernels, the status travels inside the last buffer.
  // LM trial without intermediate host round trips: solve_async() queues the whole solve and leaves the status on the
  // device; trial_stats() (after the caller's update / error evaluation) returns that status together with chi2 and
  // computeScale(lambda) = x'(lambda x + b) behind ONE synchronisation
  // the reduced system as an operator (matrix-free PCG, here and sharded in distributed.py)
  void schur_operator_prepare();
  void schur_operator_apply(const double* din, double* dout);
  void solve_async();
  void trial_stats(double lambda, int* ok, double* chi2, double* scale);
  void trial_stats_begin(double lambda);   // the queued half of trial_stats (kernels + read-back, no synchronisation)
  void exchange_setup(int nbb, const int* bblock, const double* hkeep, int nbp, const int* bpose, const double* bkeep, int nh,
                      const int* halo, const double* hmine);
  void exchange_pack(int which);     // 1: boundary Hschur blocks + boundary bschur -> buffer 105; 3: halo x (masked) + status -> 106
  void exchange_unpack(int which);   // 1: buffer 105 (x keep) -> Hschur / bschur; 3: buffer 106 -> x
  void solve_reduced_finish_async(); // un-permute x_p, no status read
  int exchange_status();             // after exchange_unpack(3): 0 ok, 1 some rank's factorisation failed (synchronises)
  // Collectives inside the library (comm.h): the whole sharded solve and the LM scalars without a Python / torch layer.
  Comm comm;
  void comm_init_rccl(int rank, int world, const char* id128);
  void comm_init_peer(int rank, int world, HostAllReduceFn fn, void* ctx, size_t slot_doubles);
  void comm_all_reduce(double* dev, size_t n, int op);
  int solve_sharded();
  int solve_sharded_once();                       // Schur pass, three all-reduces, subtree-distributed factorisation, back-substitution
  double chi2_sharded();                     // activeRobustChi2 over all ranks
  double max_diagonal_sharded();             // computeLambdaInit's maximum over the SUMMED pose diagonal and all landmarks
  double compute_scale_sharded(double lambda);
 private:
  struct Exchange {
    int nbb = 0, nbp = 0, nh = 0;
    DevBuf<int> bblock, bpose, halo;
    DevBuf<double> hkeep, bkeep, hmine, buf1, buf3;
    bool valid = false;       // exchange_setup has been called for the current structure (build_structure resets the whole record)
    bool merged = false;      // the boundary blocks / b_p ride in the all-reduce of the subtree roots (sharded_merge)
    double* tail = nullptr;   // ... their place behind the Cholesky's exchange buffer
  } ex_;
  bool selftest_done_ = false, selftest_fallback_ = false;
  int solve_sharded_repeat();
  void solve_back_substitute_impl();
  void launch_boundary_reduce();
  void drop_graph_segments();
  bool sv_ready_ = false, sv_now_ = false;
  size_t hpp_blocks_ = 0;
  std::vector<int> sv_base_, sv_diag_, sv_ptr_, sv_slot_;
  DevBuf<int> d_sv_slot;
  void solve_reduced_local_impl();
  void solve_reduced_shared_impl();
  DevBuf<unsigned char> d_lam_mask;
  DevBuf<double> d_lam;                    // {lambda_pose, lambda_landmark}: virtual damping (Schur mode)
  double lam_pose_ = 0.0, lam_lm_ = 0.0;
  std::vector<unsigned char> lam_mask_h_;
  DevBuf<int> d_active;                    // multi-GPU: reduced-system blocks this rank forms (schur_reduce)
  int n_active_ = -1;                      // -1: all
  std::vector<int> rd_cnt_h_, rd_ptr_h_, rd_slot_h_, hs_src_h_, hs_diag_h_;
  DevBuf<int> d_pose_diag;                 // pose -> its diagonal block of the reduced system
  bool hschur_valid_ = true, virt_now_ = false;
 public:
  size_t dependency_fallbacks = 0;   // dependency-driven launches that gave up and were repeated level by level
  static constexpr bool ba_skip_hpl = true;            // fused BA path: Hpl is not written at all while nobody reads it (ensure_hpl)
  bool ba_fuse_landmarks = true;      // ... and the landmark side (Hll, b_l, errors) is assembled by the Schur tiles of the solve
  static constexpr int ba_store_ll = 0;                // ... which then also write Hll and the errors to memory (0: only b_l and Dinv, what the solve reads)
  static constexpr bool ba_recompute_backsub = true;   // fused BA path: back-substitution from the Jacobians instead of reading Hpl
  static constexpr double fuse_reduce_max_partials = 6.0;   // solve() folds the Schur reduction into the factorisation only while a block of the reduced system has at most this many partial blocks on average
  bool marginals_recursion = true;  // compute_marginals: all entries on the pattern of L in one top-down pass (sparse inverse) instead of one pair of sweeps per column
  bool marginals_reduced = false;   // compute_marginals: invert the reduced pose system instead of Hpp alone (the reference inverts Hpp)
 private:
  hipStream_t side_ = nullptr;
  hipEvent_t side_fork_ = nullptr, side_join_ = nullptr;
  hipStream_t fetch_st_ = nullptr;                         // ba_fetch_begin: the copy stream of the asynchronous estimate read-back
  hipEvent_t fetch_fork_ = nullptr, fetch_ev_[kFetchMaxPieces] = {};
  int fetch_pieces_ = 0;
  void ba_fetch_fence();
  DevBuf<double> d_red_multi;              // trial_stats: partial sums of every reduction of the call
  double* h_trial_ = nullptr;              // ... their pinned host copy (+ the factorisation status word): ONE synchronisation per trial
  size_t h_trial_n_ = 0;
  int sync_status_ = -1;                   // status of a solve_async() that had to run synchronously (-1: none)
  struct TrialPending {                    // trial_stats_begin() -> trial_stats()
    bool begun = false, bad = false, need_chi = false;
    int mode = 0;                          // 1: the status word of an asynchronous solve is part of the read-back
    size_t hn = 0;
    std::vector<int> nblk;
  } trial_;
  hipEvent_t trial_ev_ = nullptr;
  bool deferred_status_ = false;           // a solve_async() whose status has not been read yet
  bool chi2_valid_ = false;                // chi2_value_ matches the errors / kernels of every edge set
  double chi2_value_ = 0.0;
  DevBuf<double> d_Hpp, d_Hpl, d_Hll, d_Hschur, d_Dinv, d_db, d_b, d_x, d_bschur, d_bkP, d_bkL, d_red;
  DevBuf<int> d_pp_diag, d_pl_colptr, d_pl_row, d_pl_lm;
  DevBuf<int> d_hs_src;                    // Hschur block -> Hpp block id or -1
  DevBuf<int> d_hs_diag;                   // Hschur block -> pose index when diagonal, else -1
  // Schur tiles: landmark ranges, their destination blocks and LDS-local contributor entries
  DevBuf<int> d_tile_lm0, d_tile_td0, d_td_diag, d_td_ptr, d_te_pack, d_rd_ptr, d_rd_slot;
  DevBuf<int2> d_tile_q2;         // per tile: second Hpl block range (first block, count) -- tiles of a split landmark
  int n_split_tiles_ = 0;
  std::vector<int> tile_lm0_h_;   // first landmark of every tile (+ one past the last)
  DevBuf<unsigned short> d_te_lm, d_slot_lm;
  DevBuf<double> d_Pd, d_Pr;               // per (tile, destination) partial blocks / rhs
  int n_tiles_ = 0;
  long n_td_ = 0;
  size_t schur_lds_bytes_ = 0;
  // multiply_hessian pattern
  DevBuf<int> d_pp_colptr, d_pp_row;
  long n_sc_ = 0;
  std::unique_ptr<SparseCholesky> chol_;
  std::unique_ptr<BlockPCG> pcg_;
  // linear_solver 2: PCG on the reduced system WITHOUT forming it (Hschur v = Hpp v + lambda v - Hpl Dinv Hpl' v)
  std::unique_ptr<BlockPCG> pcg_mf_, pcg_hpp_;
  DevBuf<int> d_pm_ptr, d_pm_q, d_pm_lm;     // pose-major lists of the Hpl blocks (block id, landmark)
  DevBuf<int> d_mh_ptr, d_mh_q, d_mh_lm;     // ... the same for multiply_hessian (any mode)
  std::unique_ptr<BlockPCG> mh_hpp_;         // gather-form symmetric product with Hpp (multiply_hessian)
  DevBuf<double> d_mf_l, d_mf_diag, d_mf_zero;
  bool mf_ready_ = false;
  int solve_matrix_free();
  void mf_prepare_lists();
  void ba_validate();
  void ba_validate_edges(const struct EdgeSet& es, const int* cam_v, const int* pt_v, size_t n) const;
  bool ba_recompute_ok() const;
  bool ba_skip_hpl_ok() const;
  bool ba_fuse_ll_ok() const;
  int ba_lm_group() const;
  void ensure_hpl();
  void ensure_side();
  void launch_ba_poses(hipStream_t sp);
  void ensure_ll();
  void ensure_bl();
  void launch_ba_landmarks(bool write_hpl);
  bool hpl_valid_ = true;
  bool ll_valid_ = true;   // Hll, b_l and the errors of the fused BA path match the last build_system
  bool ll_hbm_partial_ = false;   // ... but the Schur tiles that assembled them wrote b_l only (Hll and the errors stayed on chip)
  void pg_validate();
  struct BaFrontEnd {
    int set = -1, n_edges = 0, n_cams = 0, n_points = 0;
    double f = 0, cx = 0, cy = 0;
    int n_classes = 1;          // edge classes (ba_set_edges_classes): > 1 = the class of an observation rides in the top byte of its
    DevBuf<double> ctab;        // camera index, ctab[5 c] = (f, cx, cy, robust kernel kind, delta)
    std::vector<double> h_ctab;
    DevBuf<int> cam_v, pt_v, cam_hidx, pt_hidx, edge_hpl;
    std::vector<int> h_cam_v, h_pt_v, h_cam_hidx, h_pt_hidx;   // host copies: index validation (ba_validate)
    // which estimates the assembled system was built from (back-substitution re-evaluates the Jacobians from them)
    long est_version = 0, bak_version = 0, sys_version = -1;
    int sys_kind = 0;
    double sys_delta = 0.0;
    bool omega_identity = false;   // information = identity for the whole set (info == NULL): not read per edge
    bool err_valid = false, jac_valid = false;   // errors / Jacobians of the set match the current estimates
    DevBuf<double> chi_part;                     // per workgroup of ba_linearize_kernel: partial chi2 (folded into d_red_multi)
    bool fused_ok = false;   // every Hpl block has exactly one observation: fused on-the-fly assembly allowed
    DevBuf<double> meas, cams, pts, cams_bak, pts_bak;
    DevBuf<double> meas_pm, omega_pm;   // pose-major copies (observation-list order of the pose side)
    DevBuf<int> pt_pm, cam_pm;
    DevBuf<double> meas_lm, omega_lm;   // landmark-major copies
    DevBuf<int> cam_q, pt_q;            // per Hpl block (block order): camera / point of its observation
    DevBuf<double> meas_q, omega_q;
    DevBuf<int> cam_lm, pt_lm, hpl_lm, row_lm;   // (row_lm: pose block row of the observation's Hpl block, -1 = fixed pose)
    // Schur tiles that assemble their landmarks (ba_schur_tile_kernel<G, true>): lane slots of every tile -- observation
    // (12 bits, relative to the tile's first one; 0xfff: none) | list length (8 bits) | landmark inside the tile at the
    // first lane of its list, position in the list elsewhere (11 bits) | bit 31: not a first lane; a landmark never
    // straddles a wavefront -- and per tile (first slot, slots, first observation, longest list)
    DevBuf<int4> ll_rec;      // per lane slot: (slot word, camera, point, Hpl block or -1)
    DevBuf<int> ll_edge;      // ... edge id (error store)
    DevBuf<int> ll_row;       // ... pose block row of the observation's Hpl block (-1: fixed pose)
    DevBuf<double> ll_meas, ll_omega;   // ... measurement, information
    DevBuf<int4> tile_ll;
    DevBuf<int> sel_idx;      // ba_get_estimates_of: the selected cameras | points
    DevBuf<double> sel_out;
    bool ll_slots_ok = false;
    bool has_backup = false;
  } ba_;
  struct PgFrontEnd {   // pose-graph front end: type 1 = EdgeSE2 (x, y, theta), 2 = EdgeSE3 (isometries T[12])
    int set = -1, type = 0, nv = 0;
    DevBuf<int> vi, vj, hidx;
    std::vector<int> h_vi, h_vj, h_hidx;
    DevBuf<double> meas, poses, poses_bak;
    bool has_backup = false;
    bool err_valid = false, jac_valid = false;
  } pg_;
  EventTimer tq_, ts_, tn_, tl_, tb_, tfe_;
  void require_structure() const;
  double reduce_sum_finish(int nblocks);
};

}  // namespace g2ohip
