// Multifrontal sparse *block* Cholesky for symmetric positive definite block matrices with
// uniform bs x bs blocks (bs = pose dimension), resident on one MI355X.
//
// Replaces, for the reduced pose system, the reference's linear solver back end
//   LinearSolverCSparse::solve / computeSymbolicDecomposition
//       /root/reference/g2o/solvers/csparse/linear_solver_csparse.h:106-142,246-308
//   csparse_extension::cs_chol_workspace / cs_cholsolsymb
//       /root/reference/g2o/solvers/csparse/csparse_helper.cpp:56-143
// (twin: LinearSolverCholmod, g2o/solvers/cholmod/linear_solver_cholmod.h:79-154).
//
// Design (MI355X-first, not a translation of the up-looking CSparse loop, which is strictly
// sequential over rows):
//   * ordering on the BLOCK graph, like the reference (linear_solver_csparse.h:252-281), but
//     nested dissection instead of AMD so that the elimination tree is wide and shallow --
//     on a GPU tree height is the critical path, fill is secondary;
//   * block elimination tree -> supernodes -> dense frontal matrices; fronts of one tree
//     level are independent and are factorised by one kernel launch (one workgroup per
//     front, the front held in LDS when it fits, in an HBM scratch slab otherwise);
//   * update (Schur) matrices are passed child -> parent through HBM (extend-add), the L
//     panels stay resident for the two triangular sweeps;
//   * "not positive definite" (pivot <= 0, csparse_helper.cpp:136) is a device flag read back
//     once per solve.
#pragma once
#include <functional>

#include "common.h"

namespace g2ohip {

struct CholOptions {
  int nd_leaf = 0;           // nested-dissection leaf size (blocks); 0: 32 (and more, see analyze) for band-shaped graphs, 4 for the others
  int max_sn_scalars = 48;   // supernode (pivot panel) width cap, scalars
  int max_sn_scalars_lds = 0;   // ... for the fronts small enough for LDS; 0: 24 for band-shaped graphs (the register fronts), 48 otherwise
  static constexpr double relax_zeros = 0.25; // relaxed amalgamation: tolerated share of explicit zero blocks in a panel
  static constexpr size_t lds_front_bytes = 256 * 1024;  // fronts up to this DENSE size (m*m*8) are candidates for LDS (stored packed: half) ...
  static constexpr size_t lds_budget_bytes = 150 * 1024; // ... if blocks + vectors + index tables fit this per-workgroup LDS budget
  static constexpr bool fuse_chains = true;   // fuse parent/only-child chains into one workgroup task
  static constexpr int max_chain_fronts = 0;  // > 0: chains longer than this are cut into equal segments
  int rank = 0, world = 1;   // multi-GPU: this rank factorises its own subtrees + (redundantly) the shared top of the tree
  int dep_levels = 64;                   // > 1: up to this many consecutive task levels share ONE launch; a parent task waits for its
                                         // children through device-scope counters instead of the launch boundary
  static constexpr int fuse_fwd_any = 1;                  // forward sweep fused into the factor kernel whatever the number / size of a front's children
  static constexpr int lds_mfma = (4 << 16) | 96;                     // LDS fronts with at least (low 16 bits) boundary rows and (high bits) pivot blocks: pivot steps update the panel only, ONE MFMA rank-npiv
                                         // update of the trailing matrix afterwards (0: off -- every pivot block updates the whole trailing matrix)
  int dep_spin_limit = 1 << 21;          // polls (~0.2 us each) before a waiting workgroup gives up and flags status 2
  static constexpr int big_front_min_dim = 180;           // ... for the launches whose largest front has at least this many rows
  static constexpr int wide_front_doubles = 5000;         // launches whose largest LDS front has this many packed doubles (100 rows) use 512 threads per front
  int big_front_passes = 1;              // scratch-slab (large) fronts: whole-GPU passes instead of one workgroup per front
  int dep_backward = 1;                  // the backward sweep uses the same dependency-driven groups (parents first)
  static constexpr int dep_delay = 0;                     // launch-order distance between a task and its parent inside a wide dependency-driven launch
  static constexpr int wave_front_tasks = 1024;           // launches at least this wide use two waves (128 threads) per front
  static constexpr size_t wave_front_bytes = 0;           // (unused)
  static constexpr int mfma_diag = 1;                     // scratch-slab fronts: pivot block on the matrix cores (four waves) instead of one wave with v_readlane broadcasts
  static constexpr int fuse_panel = 1;                    // scratch-slab fronts: panel solve and trailing update of a level in one launch
  static constexpr int inplace_chains = 1;                // chains of scratch-slab fronts with identical rows (panels of one large supernode) are factorised in place
  static constexpr int hoist_big_assembly = 1;            // zero fill + original blocks of ALL scratch-slab fronts of a phase in two launches up front (their slab regions are never reused)
  static constexpr int fuse_big_forward = 1;              // forward step of scratch-slab fronts inside the pivot-block and panel kernels (levels on the fused panel path)
  int big_group = 8;                     // panels of a long in-place chain (one large supernode) are grouped: inside a group a panel's rank-npiv update
                                         // touches only the columns of the group's remaining panels, the rest of the trailing matrix gets ONE rank-(group)
                                         // update behind the group's last panel -- a quarter of the passes over a frontal matrix that lives in HBM (1: off)
  static constexpr int group_forward_side = 1;            // ... and the forward steps of such panels run on a side stream next to the next panel's factorisation
  int big_group_min_rows = 1024;         // ... for chains whose first front has at least this many rows (smaller ones are latency chains, not traffic:
                                         // 512 takes sphere from 1.41 to 2.0 ms)
#ifndef G2OHIP_BIG_MERGE_TILES
#define G2OHIP_BIG_MERGE_TILES 256
#endif
  static constexpr int big_merge_tiles = G2OHIP_BIG_MERGE_TILES;             // scratch-slab levels of at most this many 64 x 64 tiles run the fused panel kernel (panel solve + update
                                         // [+ pivot blocks, merge_diag_panel] in one launch); wider levels the separate whole-GPU passes
  static constexpr int merge_diag_panel = 1;              // pivot blocks and panel tiles of a level of scratch-slab fronts in ONE launch (tiles wait for their front's flag)
  static constexpr int merge_backward_levels = 1;         // backward step of consecutive levels of scratch-slab fronts in ONE launch (workgroups wait for their parent front's flag)
  static constexpr int split_sweeps = 1;                  // forward / backward step of scratch-slab fronts by several workgroups per front (256 boundary rows each)
  static constexpr int split_sweeps_min_dim = 512;        // ... on levels whose largest such front has at least this many rows
  static constexpr int big_gather = 1;                    // scratch-slab levels in the merged pivot-block / panel launch: the children's update matrices are added where the
                                         // frontal matrix is loaded (inverse block maps) instead of by one extend-add pass per child ordinal in front of it
  static constexpr int lazy_level_joins = 1;              // ... the two streams wait for each other only where a front has a child on the other one (0: at every such level)
  static constexpr int overlap_level_halves = 1;          // levels with LDS fronts AND scratch-slab fronts: the two halves on two streams, the forward step of the large fronts next to the following level
  static constexpr int wave_kernel = 1;                   // small fronts (<= 24 pivot columns, <= 48 boundary rows): one wavefront per task, the front in registers
  static constexpr size_t relax_front_bytes = 42 * 1024;  // relaxed merges only while the front stays this small (3 workgroups per CU)
  int band_kernel = 1;                   // leaf chains of a band (+ one dense border) on the sliding-window kernel (band_chain.inc)
  int tree_backward = 1;                 // backward sweep of the tree levels of a dependency-driven group by GROUPS of fronts: one sixteen-wave workgroup
                                         // per subtree of up to sixteen small fronts, hand-offs inside a group through LDS (tree_backward_kernel); 0: task by task
};

struct CholStats {
  size_t nnzL = 0;        // scalar nnz(L) incl. diagonal
  size_t n_fronts = 0, n_levels = 0, n_tasks = 0, max_front_dim = 0;
  size_t n_band = 0;      // leaf chains on the band kernel
  size_t nnzL_band = 0, piv_band = 0;   // ... their share of nnz(L) and of the pivot columns (scalars)
  size_t n_tree_groups = 0;   // groups of fronts of the backward sweep (tree_backward_kernel)
  double flops = 0;       // factorisation flops (dense-front count)
  double t_symbolic = 0;  // seconds, host
  size_t bytes_L = 0, bytes_U = 0;
};

// Host-side symbolic result (kept for tests / introspection).
struct CholSymbolic {
  int nb = 0, bs = 0;
  std::vector<int> perm, iperm;            // perm[new] = old block
  std::vector<int> parent;                 // block etree (permuted indices)
  std::vector<int> sn_start;               // supernode/front f covers block cols [sn_start[f], sn_start[f+1])
  std::vector<int> f_ns, f_nb, f_parent, f_level;
  std::vector<int> rows_off, rows;         // boundary rows (permuted block idx) per front
  std::vector<int> rel_off, rel;           // per front: local block index of each boundary row in the parent front
  std::vector<long long> L_off, U_off, w_off;
  std::vector<int> asm_off;                // per front, into asm_q/asm_pos
  std::vector<int> asm_q, asm_pos;         // source block id; packed lr | lc<<15 | tr<<30
  std::vector<int> child_off, children;
  std::vector<int> task_ptr, task_fronts;    // tasks (chains of fronts fused into one workgroup)
  std::vector<int> level_ptr, level_fronts;  // task ids grouped by task level (leaves first)
  // multi-GPU partition (world > 1): owner rank of every task (-1 = shared top-of-tree task, done by all
  // ranks), of every block column (original order) and of every block of the analysed pattern
  std::vector<int> task_owner, pose_owner, block_consumer;
  std::vector<int> xroots;                   // subtree-root tasks whose update matrix / vector is exchanged
  long long L_total = 0, U_total = 0, w_total = 0;
};

// One 64-byte record per front: everything the factor kernel needs, fetched with scalar loads.
// Extend-add descriptor of one child (stored contiguously per parent).
struct ChildDesc {
  long long U_off;   // packed lower-triangular block storage of the child's update matrix
  int nbc;           // boundary blocks of the child
  int crel_start;    // start of its relative-index list inside the parent's crel segment
  int cmap_start;    // start of its packed-block -> destination-block map inside the parent's cmap segment
  int w_off;         // offset (doubles) of the child's update vector in the solve workspace
};
// One 128-byte record per front: everything the kernels need, fetched with scalar loads; the first
// two children are embedded so the common case needs no second dependent load.
constexpr int kVirtInts = 5;

struct FrontRec {
  int ns, nb, c0, asm_off, asm_cnt, child_off, child_cnt, crel_off, crel_cnt, cmap_off, cmap_cnt, tri_cnt;
  long long L_off, U_off;
  ChildDesc ch[2];
  int rows_off, w_off, pad[2];   // boundary row list; update vector in the solve workspace; pad[0]: three-children fast path,
                                 // pad[1]: parent front (bits 0-23) | children to wait for (24-30) | signal the parent (31)
};

// One leaf CHAIN of fronts whose rows are a band (every pivot block couples to at most the next four blocks) plus one
// dense border of at most four blocks: what nested dissection leaves at the bottom of the tree of a camera trajectory.
// band_chain.inc factorises such a chain on a sliding window of 16-column tiles.
struct BandChainRec {
  int f_first, nfronts;   // the chain's fronts (consecutive)
  int nblk, nS, nR;       // pivot blocks; border blocks; blocks of the band that stay (the boundary rows that continue the band)
  int ntiles;             // 16-row tiles of the band that hold original entries
  int tab_off, tab_n;     // per-chain tables (ints): front records | one record per pivot block (at pad[1]) | tile -> first record (at pad[0])
  int ent0, nent;         // records of the chain's original blocks, by band tile (two int4 each)
  int c0;                 // first pivot block column (permuted order)
  int ublk;               // 8 x 4 bits: boundary position (in the last front) of the staying band blocks (0-3) and of the border blocks (4-7)
  int pad[4];
};
// tree_backward_kernel: the fronts it takes (pivot columns, boundary rows: scalars) and the fronts (waves) of a group
constexpr int kTreePiv = 24, kTreeBnd = 48, kTreeWaves = 16;
constexpr int kBandFrontInts = 17;   // first pivot block, pivot scalars, L offset (2), m, local row of band blocks +0..+7, of border blocks 0..3

struct CholPlanDev {
  const int2* slots;                   // launch slot -> (first front id, chain length)
  const int *task_ptr, *task_fronts;   // task t = chain of fronts task_fronts[task_ptr[t] .. task_ptr[t+1])
  const FrontRec* rec;
  const ChildDesc* cdesc;
  const int* crel;
  const int* cmap;   // per parent: child's packed U block -> (row | col << 16) block of the parent front
  const int* cinv;         // scratch-slab fronts whose children are gathered at load time: per front a header (children, offsets of their update
                           // matrices) and per child ordinal the map front block -> the child's boundary block (-1: none)
  const int2* cinv_slot;   // ... per launch slot: (offset into cinv or -1, ints)
  const int *gtab, *gtab_off;   // grouped in-place chains: per group's last panel (gtab_off[front], -1: none) the earlier panels of the group --
                                // count, then (L offset low, high, rows of the front, pivot columns, row offset) each
  const int* tri;    // row-major enumeration of a lower triangle: idx -> (i | j << 16)
  const int *f_ns, *f_nb, *f_c0, *rows_off, *rows, *rel_off, *rel;
  const long long *L_off, *U_off, *w_off;
  const int *asm_off, *asm_q, *asm_pos, *child_off, *children;
  double *L, *U, *w;
  // virtual source (set_virtual_blocks): per assembly entry the base block (-1: none), pos with bit 31 = diagonal
  // block, and kVirtInts ints (count, three partial slots, first index into vslots)
  const int *asm_vq, *asm_vpos, *asm_v, *vslots;
  const int* asm_r8;   // the same per assembly entry as ONE 32-byte record (base, pos, count, three partial slots, first list index, 0): wave kernel, scalar loads
  const double *vbase, *vparts, *vlam;
  int vsplit;
  int* status;
  int* ready;        // dependency-driven launches: children finished so far, per front
  int dep_spin_limit;
  int lds_mfma;
  long long* dbg;   // G2OHIP_CHOL_STAMPS builds only: per-launch wall-clock stamps of workgroup 0
  long long* tl;    // ... and (start, end) wall-clock of every workgroup of the wave-kernel launch
  // band chains (band_chain.inc)
  const BandChainRec* band_rec;
  const int* band_tab;
  const int4* band_ent;    // records of the original blocks per chain and band tile (two int4 each, see the analysis) -- plain source
  const int4* band_entv;   // the same with the virtual source's fields (set_virtual_blocks)
};

class SparseCholesky {
 public:
  explicit SparseCholesky(int block_size) : bs_(block_size) {}
  CholOptions opt;

  // Host symbolic analysis of an upper-triangular block-CCS pattern (rows <= col, sorted).
  // host_only: no device work at all (partition queries, CPU tests).
  void analyze(int nb, const int* colptr, const int* rowidx, hipStream_t st, bool host_only = false);
  bool analyzed() const { return analyzed_; }
  void reset() { analyzed_ = false; }

  // Numeric factorisation from device block values [nnzb][bs*bs] (column-major blocks, same
  // block order as the analysed pattern).  Asynchronous on st.
  void factor(const double* dA, hipStream_t st);
  // x = A \ b with device vectors of nb*bs (original block order).  Asynchronous.
  void solve(const double* d_b, double* d_x, hipStream_t st);
  // ---- phased interface for the multi-GPU path (phase 0: this rank's subtrees, phase 1: shared top)
  // fwd: fused forward sweep (after solve_begin); parts: bit 0 = the band chains' launch (band_chain.inc), bit 1 = everything
  // else -- two calls with 1 and 2 let a caller time the two halves separately
  void factor_phase(const double* dA, int phase, hipStream_t st, bool fwd = false, int parts = 3);
  bool has_band_chains(int phase) const;
  // called with 0 right before and 1 right after the band chains' launch (per-kernel timing by the caller; plain launches only)
  std::function<void(int)> band_hook;
  void factor_solve(const double* dA, const double* d_b, double* d_x, hipStream_t st);
  void solve_begin(const double* d_b, hipStream_t st);            // permute the right-hand side in
  void solve_forward_phase(int phase, hipStream_t st);
  void solve_backward_phase(int phase, hipStream_t st);
  void solve_end(double* d_x, hipStream_t st);                    // permute the solution out
  void pack_exchange(hipStream_t st);                             // own subtree roots: U, w -> exchange buffer (others zero)
  void unpack_exchange(hipStream_t st);                           // foreign subtree roots: exchange buffer -> U, w
  void mask_solution(hipStream_t st);                             // zero the entries other ranks own (before the x all-reduce)
  // ---- entries of the INVERSE on the pattern of L (Takahashi / Erisman-Tinney recursion over the frontal matrices, top
  // down): after factor(), sparse_inverse() fills one dense m x m "inverse front" per frontal matrix; block (r, c)
  // (original block indices) of A^-1 is inverse_block(): its address in the slab, or false when (r, c) is not in
  // the pattern of L + L'.  One GPU only (world == 1).
  void sparse_inverse(hipStream_t st);
  bool inverse_block(int r, int c, long long* offset, int* ld, bool* transposed) const;
  const double* inverse_slab() const { return d_Z.p; }
  double* exchange_buffer(size_t* count) { *count = xbuf_count_; return d_xbuf.p; }
  // room for `n` more doubles BEHIND the subtree-root segments of the exchange buffer (the caller's own payload travels in
  // the same all-reduce: BlockSolver's boundary blocks); pack_exchange clears and fills the head only
  double* reserve_exchange_tail(size_t n);
  double* permuted_solution(size_t* count) { *count = (size_t)sym_.nb * bs_; return d_xp.p; }
  // A caller that produces the right-hand side itself may write it permuted (xp[iperm[old] * bs + r]) and clear the status word
  // in the same kernel: solve_begin and the memset of factor_phase(phase 0) are then two launches less (skip_status_clear)
  const int* inverse_permutation_device() const { return d_iperm.p; }
  int* status_word_device() { return d_status.p; }
  bool skip_status_clear = false;   // consumed by the next factor_phase(phase 0)
  int* status_flag() { return d_status.p; }
  // Synchronises st and returns true when the last factorisation met a pivot <= 0.
  // The matrix to factorise given as  A[q] = base[base_idx[q]] (+ lam[0] on the diagonal of diagonal blocks)
  //                                          - sum_k parts[part_slot[k]],  k in [part_ptr[q], part_ptr[q+1])
  // (q = block in pattern order; partial blocks stored [row part][column][row inside the part] with BS/2 rows per
  // part when `split`, else plain column-major; block `zero_slot` of parts is all zeros).  factor_phase(nullptr,..)
  // then assembles the fronts straight from it: the Schur complement is never written out.
  struct VirtualBlocks {
    const int *base_idx, *is_diag, *part_ptr, *part_slot;   // host
    const int* d_part_slot;                                 // device copy of part_slot
    const double *base, *parts, *lam;                       // device
    int zero_slot;
    bool split;
  };
  void set_virtual_blocks(const VirtualBlocks& vb, hipStream_t st);
  bool has_virtual_blocks() const { return d_asm_v.p != nullptr; }
  void set_virtual_split(bool split) { plan_.vsplit = split ? 1 : 0; }
  const int* status_device() const { return d_status.p; }   // 0 ok, 1 non-positive pivot, 2 dependency wait gave up
  bool failed(hipStream_t st);
  // the same in two halves, for a caller that has its own read-back to synchronise on: status_async enqueues the copy of
  // the status word into (pinned) host memory, failed_with interprets the value once the stream has been synchronised
  void status_async(int* host, hipStream_t st) { G2OHIP_HIP_CHECK(hipMemcpyAsync(host, d_status.p, sizeof(int), hipMemcpyDeviceToHost, st)); }
  bool failed_with(int h, hipStream_t st);
  // true once after failed() saw a dependency-driven launch give up waiting; those launches are off from then on
  // (the caller drops its captured graphs and repeats the solve)
  bool dependency_stall() { const bool v = dep_stalled_; dep_stalled_ = false; return v; }

  const CholStats& stats() const { return stats_; }
  const CholSymbolic& symbolic() const { return sym_; }
  int block_size() const { return bs_; }

 private:
  int bs_;
  bool analyzed_ = false;
  CholSymbolic sym_;
  CholStats stats_;
  // device plan
  DevBuf<int> d_f_ns, d_f_nb, d_f_c0, d_rows_off, d_rows, d_rel_off, d_rel, d_asm_off, d_asm_q, d_asm_pos,
      d_child_off, d_children, d_level_fronts, d_perm, d_iperm, d_status;
  DevBuf<long long> d_L_off, d_U_off, d_w_off, d_scratch_off;
  DevBuf<int> d_scratch_ld;   // leading dimension of a scratch-slab front (that of the chain's first front)
  // sparse inverse: slab of the inverse fronts, per-front offsets / parents, per front level the launch lists
  DevBuf<double> d_Z;
  DevBuf<long long> d_zoff;
  DevBuf<int> d_fparent;
  DevBuf<int4> d_spinv_work;   // (front, first row / column of the chunk, -, -)
  std::vector<long long> zoff_h_;
  struct SpInvLevel { int f_begin, f_count, rc_begin, rc_count; };   // into d_spinv_work: one entry per front, one per 64-row chunk of its boundary
  std::vector<SpInvLevel> spinv_levels_;
  bool spinv_planned_ = false;
  int spinv_npiv_max_ = 0;
  DevBuf<FrontRec> d_rec;
  DevBuf<ChildDesc> d_cdesc;
  DevBuf<int> d_crel, d_cmap, d_tri, d_task_ptr, d_task_fronts, d_cinv, d_gtab, d_gtab_off;
  DevBuf<int2> d_cinv_slot;
  DevBuf<double> d_L, d_U, d_w, d_y, d_xp, d_scratch;
  int hz_begin_[2] = {0, 0}, hz_count_[2] = {0, 0}, ha_begin_[2] = {0, 0}, ha_count_[2] = {0, 0};   // phase-wide fill / assembly chunks (d_big_tiles)
  // merged backward launches: per phase, runs of consecutive levels (top level first) of scratch-slab fronts only
  struct BwGroup { int top_level, bottom_level, begin, count; };
  std::vector<BwGroup> bw_groups_[2];
  std::vector<int> bw_of_level_[2];   // level -> index into bw_groups_ or -1
  DevBuf<int> d_sw_flag;              // per front: its pivot part of the solution is in memory (merged launches)
  DevBuf<double> d_sw_part;     // multi-workgroup backward step: per row chunk the partial L21' x (64 doubles)
  DevBuf<int> d_sw_cnt;         // ... and per launch slot the chunks that have delivered (the last one finishes the front and resets it)
  DevBuf<double> d_sweep_vec;   // vectors of the triangular sweeps of fronts too large for LDS
  // per level launch info
  struct LevelLaunch {
    int lds_begin = 0, lds_count = 0, lds_max_m = 0;     // index range in d_level_fronts
    int glb_begin = 0, glb_count = 0, glb_max_m = 0;
    int max_panel = 0;                                   // max m*npiv (doubles) for solve kernels
    int max_m = 0;
    int fz_begin = 0, fz_count = 0;                      // zero-fill chunks of the scratch-slab fronts that start a region at this level
    bool hoisted = false;                                // its fill / assembly chunks are also in the phase-wide lists (hz_ / ha_)
    int lds_max_panel = 0;                               // max_panel over the LDS / register fronts only
    int sw_begin = 0, sw_count = 0;                      // row chunks of the scratch-slab fronts for the multi-workgroup sweeps (d_big_tiles); 0: not eligible
    int lds_vec_m = 0;                                   // largest front dimension among the LDS / register fronts only (their vectors in LDS)
    int lds_idx_ints = 0, glb_idx_ints = 0;              // staged index lists (ints) per front, max over the launch
    bool fuse_fwd = true;                                // every LDS front of the launch is within the fused forward sweep's limits
    int sm_count = 0, sm_max_m = 0, sm_idx_ints = 0;     // leading part of the lds range: small fronts, one wave each
    bool wv = false;                                     // every front fits the register-resident wave kernel (wave_front.inc)
    int wv_pn = 0, wv_idx_ints = 0;                      // ... its panel region (doubles) and index tables (ints), max over the launch
    bool grouped = false;                                // a front of the launch is the LAST panel of a group of an in-place chain (its tiles carry the group's columns: separate kernels)
    bool group_in = false;                               // ... a panel INSIDE a group (its tiles stop at the group's end: the fused kernels clip)
    int bt_begin = 0, bt_count = 0;                      // 64 x 64 trailing-update tiles of the scratch-slab fronts (d_big_tiles)
    // scratch-slab fronts as whole-GPU passes (all ranges index d_big_tiles): assembly chunks, one extend-add pass
    // per child ordinal, row chunks of the panel solve; big_ok: every such front has at most 64 pivot columns
    int ba_begin = 0, ba_count = 0, tr_begin = 0, tr_count = 0;
    std::vector<std::pair<int, int>> be_pass;
    long long glb_scratch = 0;                           // doubles of the scratch slab this launch uses
    bool big_ok = false;
    bool eg_ok = false;                                  // the level's extend-add as one launch (big_extend_gather_kernel: every front has a map table, 2..7 children)
    int eg_begin = 0, eg_count = 0;                      // ... its chunks (d_big_tiles)
    int eg_maxc = 7;                                     // ... children per front at most
    bool eg_write = false;                               // ... and it WRITES the regions of its fronts (no zero fill for them; their original blocks, and those of the
                                                         // fronts continued in place behind them, are added behind it: la_*)
    int la_begin = 0, la_count = 0;
    bool tr_all = false;                                 // every scratch-slab front of the level has boundary rows (big_panel_solve_kernel)
    // levels whose LDS fronts run on a side stream next to the scratch-slab passes (overlap_level_halves): split_ok = the level
    // qualifies by its structure; fork = an LDS front of it has a child that ran on the main stream since the side stream last
    // waited for it; join = a front of the main part has a child that ran on the side stream since the main stream last did
    bool split_ok = false, fork = false, join = false;
    bool gather = false;                                 // every scratch-slab front of the level can gather its children's update matrices at load time (cinv tables): no extend-add passes
  };
  std::vector<LevelLaunch> launches_[2];   // [0] own tasks, [1] shared top-of-tree tasks
  struct FactorGroup {
    LevelLaunch LL; int first_level, last_level; bool dep; int band_count = 0, band_rec0 = 0, band_ent_cap = 0, band_tab_cap = 0;
    int tb_grp0 = 0, tb_ngrp = 0, tb_low = 0;   // tree_backward: its groups (d_tb_grec), the launch slots (lowest levels) left to the per-task kernel
  };
  std::vector<FactorGroup> groups_[2];     // factorisation launches: runs of levels (dep: one launch, in-kernel dependencies)
  DevBuf<int> d_ready;
  DevBuf<int4> d_tb_grec;    // tree_backward: per group (first entry of d_tb_front, fronts, levels, front to wait for)
  DevBuf<int2> d_tb_front;   // ... per group front (front, level inside the group | tasks to release << 8), level by level
  DevBuf<int> d_tb_rows;     // ... the boundary row lists (indexed like d_rows) with the rows a front of the same group owns replaced by -1 - (offset in LDS)
  struct SegCopy { long long a, b; int n, flags; };  // exchange segment: a = offset in U (or w: flag 2), b = offset in xbuf; flag 1 = mine
  DevBuf<SegCopy> d_xseg;
  DevBuf<double> d_xbuf, d_xmask;
  DevBuf<long long> d_dbg;
  DevBuf<int2> d_slots, d_fslots, d_bslots;
  DevBuf<int> d_asm_vq, d_asm_vpos, d_asm_v, d_asm_r8;
  DevBuf<BandChainRec> d_band_rec;
  DevBuf<int> d_band_tab;
  DevBuf<int4> d_band_ent, d_band_entv;
  std::vector<int4> band_ent_h_;        // host copy of d_band_ent (set_virtual_blocks derives d_band_entv from it)
  std::vector<int> band_ent_asm_;       // assembly entry (index into asm_q) of every band entry
  DevBuf<int4> d_big_tiles;
  int n_slots_ = 0;
  bool dep_off_ = false, dep_stalled_ = false;
  int n_xseg_ = 0;
  [[maybe_unused]] int dbg_launch_ = 0;   // (G2OHIP_CHOL_STAMPS builds)
  size_t xbuf_count_ = 0;
  void launch_factor(const LevelLaunch& LL, const double* dA, bool fwd, hipStream_t st, bool dep = false, int parts = 3);
  bool band_usable(const FactorGroup& G, const double* dA) const;
  void launch_band(const FactorGroup& G, const double* dA, bool fused, hipStream_t st, bool dep);
  hipStream_t side_[2] = {nullptr, nullptr};   // factor_phase: the two halves of a level / the forward step of its large fronts
  hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
  void launch_solve(const LevelLaunch& LL, bool fwd, hipStream_t st, bool glb_only = false, bool dep = false, bool skip_glb = false, int dep_tail = 0);
  bool big_forward_carried(const LevelLaunch& LL) const;
  static void prepare_kernels();
  int merge_tiles_of(const LevelLaunch& LL) const { return (LL.grouped || LL.group_in) ? -1 : opt.big_merge_tiles; }   // (grouped chains: the separate kernels)   // the forward step of the level's scratch-slab fronts rides along in their factorisation
  CholPlanDev plan_{};
};

// Nested-dissection ordering of a symmetric block graph (CSR without self loops).
void nested_dissection(int n, const std::vector<int>& xadj, const std::vector<int>& adj, int leaf,
                       std::vector<int>& perm);

}  // namespace g2ohip
