// g2o adapter over the C ABI of libg2ohip (include/g2ohip.h): what a g2o maintainer adds to the g2o tree
// (e.g. as g2o/solvers/hip/block_solver_hip.h) to run SparseOptimizer::optimize() on an MI355X.
//
//   BlockSolverHip<p, l>        wide seam: replaces g2o::BlockSolver<BlockSolverTraits<p, l>> behind g2o::Solver
//                               (/root/reference/g2o/core/solver.h:44-149, block_solver.h:83-178)
//   LinearSolverHip<MatrixType> narrow seam: replaces LinearSolverCSparse / LinearSolverCholmod behind
//                               g2o::LinearSolver<MatrixType> (/root/reference/g2o/core/linear_solver.h:40-81)
//
// Everything here is C++ ABI of the consumer's g2o build (vtables, Eigen types), so it is compiled against the
// consumer's g2o + Eigen; this repository syntax-checks it against tests/cpp/g2o_decl (declarations of the g2o
// members it touches, tests/test_adapter_syntax.py).  Registration as a plugin: solver_hip.cpp.
#ifndef G2O_HIP_SOLVER_H
#define G2O_HIP_SOLVER_H

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <typeinfo>
#include <utility>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

#include "g2o/core/batch_stats.h"
#include "g2o/core/block_solver.h"
#include "g2o/core/jacobian_workspace.h"
#include "g2o/core/linear_solver.h"
#include "g2o/core/robust_kernel_impl.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/stuff/timeutil.h"
// The device fast paths recognise g2o's edge / vertex types by typeid.  Those classes have out-of-line virtuals, so their
// typeinfo objects live in libg2o_types_sba / _slam2d / _slam3d: a plugin built with a fast path has to LINK the matching
// type library (g2o_cli loads plugins with dlopen(RTLD_LAZY) and no RTLD_GLOBAL, dl_wrapper.cpp:118 -- an unresolved typeinfo
// makes the whole plugin fail to load).  Each fast path is a compile-time switch (default on); with all three off the plugin
// depends on libg2o_core and libg2ohip only and still offers the generic path and the narrow seam.
#ifndef G2OHIP_FASTPATH_SBA
#define G2OHIP_FASTPATH_SBA 1
#endif
#ifndef G2OHIP_FASTPATH_SLAM2D
#define G2OHIP_FASTPATH_SLAM2D 1
#endif
#ifndef G2OHIP_FASTPATH_SLAM3D
#define G2OHIP_FASTPATH_SLAM3D 1
#endif
#if G2OHIP_FASTPATH_SBA
#include "g2o/types/sba/types_six_dof_expmap.h"   // EdgeProjectXYZ2UV, VertexSE3Expmap, VertexSBAPointXYZ (link types_sba)
#endif
#if G2OHIP_FASTPATH_SLAM2D
#include "g2o/types/slam2d/edge_se2.h"            // EdgeSE2, VertexSE2 (link types_slam2d)
#endif
#if G2OHIP_FASTPATH_SLAM3D
#include "g2o/types/slam3d/edge_se3.h"            // EdgeSE3, VertexSE3 (link types_slam3d)
#endif
#include "g2ohip.h"

namespace g2o {

// ---------------------------------------------------------------------------------------------------------
// What a device-resident iteration needs from the solver BESIDES g2o::Solver: the graph-side operations of
// OptimizationAlgorithmLevenberg / GaussNewton::solve that SparseOptimizer runs on the host (computeActiveErrors,
// activeRobustChi2, update, push / pop / discardTop: sparse_optimizer.cpp:61-114,421-434,599-650) executed on the estimates the
// device front ends hold.  Implemented by BlockSolverHip when EVERY active edge sits on a device front end; used by
// OptimizationAlgorithmLevenbergHip / GaussNewtonHip (g2o_hip_algorithm.h), which fall back to g2o's host loop otherwise.
// ---------------------------------------------------------------------------------------------------------
// Persistent helper threads for the adapter's per-vertex / per-edge host loops: chunk t of [0, n) always runs on thread t (the
// callers that merge per-chunk results rely on the chunking, not on who runs it); the calling thread takes chunk 0.  Spawning
// seven threads per loop cost ~0.1 ms, twice per write-back of an iteration that is 1.3 ms on the device.
class HipWorkers {
 public:
  HipWorkers() : _stop(false), _gen(0), _pending(0), _n(0), _chunk(0), _active(0), _fn(0) {}
  ~HipWorkers() {
    {
      std::unique_lock<std::mutex> lk(_m);
      _stop.store(true);
    }
    _wake.notify_all();
    for (size_t t = 0; t < _threads.size(); ++t) _threads[t].join();
  }
  // fn(begin, end) on nt chunks of [0, n); returns when all of them are done
  void run(size_t n, size_t nt, const std::function<void(size_t, size_t)>& fn) {
    const size_t chunk = (n + nt - 1) / nt;
    while (_threads.size() + 1 < nt) {
      const size_t id = _threads.size() + 1;
      _threads.push_back(std::thread(&HipWorkers::loop, this, id, _gen.load()));   // (_gen is written by this thread only)
    }
    size_t pending = 0;
    for (size_t t = 1; t < nt; ++t)
      if (t * chunk < n) ++pending;
    {
      std::unique_lock<std::mutex> lk(_m);
      _fn = &fn;
      _n = n;
      _chunk = chunk;
      _active = nt;
      _pending.store(pending);
      _gen.fetch_add(1, std::memory_order_release);
    }
    _wake.notify_all();
    fn((size_t)0, chunk < n ? chunk : n);
    // (the chunks are equal: the others finish within microseconds of this one -- spin, no condition variable on the way back)
    while (_pending.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  }

 private:
  // A worker spins on the generation counter for a few milliseconds after its last job (an optimize() run hands out a job per
  // millisecond: waking 31 sleeping threads through a condition variable cost 0.3-0.5 ms per loop), then sleeps.
  void loop(size_t id, unsigned long long seen) {
    // (the spin is bounded by TIME -- G2OHIP_ADAPTER_SPIN_US, default 1 500 us: one Levenberg-Marquardt iteration of the
    // measured runs -- not by a yield count whose duration depends on the host's load: an embedding application gets its cores
    // back a millisecond and a half after the solver's last parallel loop)
    static const long spin_us = [] {
      const char* e = std::getenv("G2OHIP_ADAPTER_SPIN_US");
      return e ? std::atol(e) : 1500L;
    }();
    for (;;) {
      unsigned spins = 0;
      std::chrono::steady_clock::time_point t0;
      while (!_stop.load(std::memory_order_relaxed) && _gen.load(std::memory_order_acquire) == seen) {
        if ((++spins & 63) == 1 && spins == 1) t0 = std::chrono::steady_clock::now();
        const bool keep = spin_us > 0 && ((spins & 63) != 0 ||
                          std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < spin_us);
        if (keep) {
          std::this_thread::yield();
        } else {
          std::unique_lock<std::mutex> lk(_m);
          while (!_stop.load() && _gen.load() == seen) _wake.wait(lk);
        }
      }
      if (_stop.load()) return;
      const std::function<void(size_t, size_t)>* fn = 0;
      size_t b = 0, e = 0;
      {
        std::unique_lock<std::mutex> lk(_m);   // (the job's fields are published under the lock)
        seen = _gen.load();
        b = id * _chunk;
        e = b + _chunk < _n ? b + _chunk : _n;
        if (id >= _active || b >= _n) continue;
        fn = _fn;
      }
      (*fn)(b, e);
      _pending.fetch_sub(1, std::memory_order_release);
    }
  }
  std::mutex _m;
  std::condition_variable _wake;
  std::vector<std::thread> _threads;
  std::atomic<bool> _stop;
  std::atomic<unsigned long long> _gen;
  std::atomic<size_t> _pending;
  size_t _n, _chunk, _active;
  const std::function<void(size_t, size_t)>* _fn;
};

class HipDeviceGraph {
 public:
  virtual ~HipDeviceGraph() {}
  // The accepted estimates into the vertices, PIPELINED: devFetchBegin right behind devUpdate of a trial starts the read-back next
  // to the trial's error evaluation; devFetchEnd (the trial was accepted) writes the pieces into the vertices as they arrive,
  // devFetchCancel (rejected: the vertices keep what they hold) only waits for the copy.  false from devFetchBegin: not offered,
  // the caller uses devGetEstimates.
  virtual bool devFetchBegin() { return false; }
  virtual bool devFetchEnd() { return false; }
  virtual void devFetchCancel() {}
  virtual bool deviceResident() const = 0;             // every active edge is bound to a device front end
  virtual bool devEstimatesValid() const = 0;          // the device holds estimates for the current structure
  virtual bool devSetEstimates() = 0;                  // vertices -> device (setEstimate of every vertex the front ends know)
  virtual bool devGetEstimates() = 0;                  // device -> vertices (the free ones)
  virtual bool devLinearize(bool jacobians) = 0;       // computeActiveErrors (+ linearizeOplus) at the device estimates
  virtual bool devChi2(double& chi2) = 0;              // activeRobustChi2 of the errors last evaluated
  virtual bool devBuildSystem() = 0;                   // Solver::buildSystem without the b / diagonal read-back
  virtual bool devMaxDiagonal(double& d) = 0;          // what computeLambdaInit reads (levenberg.cpp:149-163)
  virtual bool devComputeScale(double lambda, double& scale) = 0;   // computeScale (levenberg.cpp:165-172)
  virtual int devSolve() = 0;                          // Solver::solve without the x read-back: 1 solved, 0 not positive definite, -1 error
  virtual bool devSolveAsync() = 0;                    // the same, queued; status with the trial's sums:
  virtual int devTrialStats(double lambda, double& chi2, double& scale) = 0;   // 1 / 0 as devSolve, 2 repeat the trial, -1 error
  virtual bool devUpdate() = 0;                        // SparseOptimizer::update(x) with the resident x
  virtual bool devPush() = 0;
  virtual bool devPop() = 0;
  virtual bool devDiscardTop() = 0;
  // Look-ahead (Levenberg-Marquardt driver): once a trial has been accepted the driver queues the head of the NEXT solve() --
  // errors, buildSystem, push, setLambda, solve, update, errors, the trial's sums (devTrialStatsBegin: no synchronisation) --
  // BEFORE it writes the accepted estimates into the vertices, so that the device works while the host stores.  The solver
  // remembers that such a trial is in flight (devSetLookAheadPending); the next solve() of the same optimize() consumes it,
  // anything else that enters the solver first (a new optimize(), computeMarginals, buildSystem, the destructor) drops it:
  // the sums are drained and the estimates popped, the device holds what the vertices hold (devDropLookAhead).
  virtual bool devCanLookAhead() const { return false; }
  virtual bool devHybrid() const { return false; }     // host-linearised groups next to the front end: they read (a few of) the vertices
  virtual bool devTrialStatsBegin(double /*lambda*/) { return false; }
  virtual bool devLookAheadPending() const { return false; }
  virtual void devSetLookAheadPending(bool) {}
  virtual void devDropLookAhead() {}
  virtual void devSetQueueing(bool) {}               // (while the look-ahead is being queued: no timing synchronisation)
};

// ---------------------------------------------------------------------------------------------------------
// Wide seam.  Edges are grouped into homogeneous SETS (error dimension, vertex dimensions, unary / binary, robust
// kernel): one g2ohip edge set per group, flat arrays in the group's edge order.
//   generic path: the edges keep producing errors and Jacobians on the CPU (computeError / linearizeOplus of the
//     consumer's edge types) and the adapter uploads them every iteration (E * (d*dim0 + d*dim1 + d*d + d) doubles:
//     1.0 GB at the metric configuration); assembly, damping, Schur complement, factorisation and back-substitution
//     run on the device;
//   fast path (default; setFastPath(false) or G2OHIP_ADAPTER_FASTPATH=0 turn it off): EVERY g2o::EdgeProjectXYZ2UV over
//     VertexSBAPointXYZ / VertexSE3Expmap is bound to the library's device-resident bundle-adjustment front end
//     (g2ohip_ba_*) as ONE edge set, whatever CameraParameters (each edge carries its own _cam,
//     types_six_dof_expmap.h:133-153) and whatever robust kernel it has: the distinct (focal length, principal point,
//     kernel, delta) combinations become edge classes (g2ohip_ba_set_edges_classes, up to 128).  The measurements go to
//     the device once, per iteration only the ESTIMATES are uploaded (12 doubles per camera + 3 per point: 34 MB at
//     the metric configuration) and the errors / Jacobians of types_six_dof_expmap.cpp:288-326 are evaluated inside the
//     assembly kernels.  Next to it ONE homogeneous group of EdgeSE2 or EdgeSE3 may sit on the pose-graph front end
//     (g2ohip_pg_*); every other group (priors, other edge types, a second pose-graph group) stays generic.
//     chi2 (computeActiveErrors / activeRobustChi2) stays with the optimizer on the CPU.
// ---------------------------------------------------------------------------------------------------------
template <int p, int l>
class BlockSolverHip : public BlockSolverBase, public HipDeviceGraph {
 public:
  explicit BlockSolverHip(int device = 0) : _h(0), _doSchur(true), _writeDebug(false), _nP(0), _nL(0), _fastPath(true), _fastGroups(0), _devValid(false), _hybrid(false), _touchedPushed(false), _fetchBegun(false), _lookPending(false), _lookEnabled(true), _queueing(false), _lambdaInForce(false), _undampedRetries(0) {
    const char* la = std::getenv("G2OHIP_ADAPTER_LOOKAHEAD");   // 0: the LM driver never queues the next iteration's first trial early (A/B)
    if (la && la[0] == '0') _lookEnabled = false;
    const char* fp = std::getenv("G2OHIP_ADAPTER_FASTPATH");
    if (fp && fp[0] == '0') _fastPath = false;
    const char* pin = std::getenv("G2OHIP_ADAPTER_PINNED");
    _pin = !(pin && pin[0] == '0');
    const char* th = std::getenv("G2OHIP_ADAPTER_THREADS");   // host threads of the estimate gather / write-back loops (1 = serial)
    // (the CPUs this PROCESS may run on -- affinity mask / cpuset of a container --, not the machine's: hardware_concurrency counts
    // cores a container was never given)
    unsigned hc = std::thread::hardware_concurrency();
#if defined(__linux__)
    {
      cpu_set_t cs;
      CPU_ZERO(&cs);
      if (sched_getaffinity(0, sizeof(cs), &cs) == 0) {
        const int n = CPU_COUNT(&cs);
        if (n > 0 && (hc == 0 || (unsigned)n < hc)) hc = (unsigned)n;
      }
    }
#endif
    // (default: an eighth of the hardware threads, 8 to 32 -- the write-back of 1.1 M vertices is a cache miss per vertex and
    // scales with the threads up to there: 1.8 / 1.05 / 1.06 ms on 16 / 32 / 64 of the 256 threads of the measurement host)
    _threads = th ? std::atoi(th) : (int)std::min(32u, std::max(8u, hc / 8));
    if (hc > 0 && _threads > (int)hc) _threads = (int)hc;
    if (_threads < 1) _threads = 1;
    const char* tm = std::getenv("G2OHIP_ADAPTER_TIMING");
    _timing = tm && tm[0] != '0';
    std::memset(&_phase, 0, sizeof(_phase));
    if (g2ohip_create(&_h, p, l, device) != G2OHIP_OK) {
      std::cerr << "BlockSolverHip: " << g2ohip_last_error() << std::endl;   // (no exceptions on this path)
      _h = 0;
    }
  }
  virtual ~BlockSolverHip() {
    devDropLookAhead();
    if (_lookQueued > 0 && std::getenv("G2OHIP_ADAPTER_VERBOSE"))
      std::cerr << "BlockSolverHip: look-ahead trials queued " << _lookQueued << ", dropped " << _lookDropped << std::endl;
    if (_timing && _phase.buildSystems > 0) printPhases(std::cerr);
    unpinAll();
    if (_h) g2ohip_destroy(_h);
  }

  // Where the time of buildSystem() / solve() goes, accumulated over the calls (G2OHIP_ADAPTER_TIMING=1 also synchronises the
  // device between the phases so that each one is charged to its own line; printed as one JSON line by the destructor)
  struct Phases {
    double hostLinearize;    // generic path: linearizeOplus of every edge + gathering the Jacobians (CPU)
    double upload;           // estimates (fast path) or Jacobians / information / errors (generic path) to the device
    double deviceBuild;      // g2ohip_build_system: errors + Jacobians (fast path) and the assembly, on the device
    double downloadB;        // b() and the diagonal mirror computeLambdaInit reads, to the host
    double deviceSolve;      // g2ohip_solve: Schur complement, factorisation, sweeps, back-substitution
    double downloadX;        // x() to the host
    double fetchWait, fetchCams, fetchPoints;   // pipelined write-back (devFetchEnd): waiting for pieces | cameras into the vertices | points
    int fetches;
    int buildSystems, solves;
  };
  const Phases& phases() const { return _phase; }
  void printPhases(std::ostream& os) const {
    const double nb = _phase.buildSystems > 0 ? _phase.buildSystems : 1, ns = _phase.solves > 0 ? _phase.solves : 1;
    os << "{\"g2ohip_adapter_phases_ms\": {\"buildSystem_calls\": " << _phase.buildSystems << ", \"solve_calls\": " << _phase.solves
       << ", \"host_linearize\": " << 1e3 * _phase.hostLinearize / nb << ", \"upload\": " << 1e3 * _phase.upload / nb << ", \"device_build\": "
       << 1e3 * _phase.deviceBuild / nb << ", \"download_b_diag\": " << 1e3 * _phase.downloadB / nb << ", \"device_solve\": "
       << 1e3 * _phase.deviceSolve / ns << ", \"download_x\": " << 1e3 * _phase.downloadX / ns << ", \"write_back_wait\": " << 1e3 * _phase.fetchWait / (_phase.fetches > 0 ? _phase.fetches : 1)
       << ", \"write_back_cameras\": " << 1e3 * _phase.fetchCams / (_phase.fetches > 0 ? _phase.fetches : 1) << ", \"write_back_points\": "
       << 1e3 * _phase.fetchPoints / (_phase.fetches > 0 ? _phase.fetches : 1) << ", \"threads\": " << _threads << ", \"pinned\": " << (_pin ? 1 : 0)
       << ", \"fast_groups\": " << _fastGroups << "}}" << std::endl;
  }

  // block_solver.hpp:606-620: store the optimizer, drop numeric / symbolic state
  virtual bool init(SparseOptimizer* optimizer, bool online = false) {
    (void)online;
    devDropLookAhead();
    _optimizer = optimizer;
    _groups.clear();
    _multi.clear();
    return _h && g2ohip_init(_h) == G2OHIP_OK;
  }

  // block_solver.hpp:142-295: index arrays from indexMapping() / activeEdges(); called at iteration 0 only
  // (optimization_algorithm_levenberg.cpp:62-68), so the edge sets are registered exactly once per init()
  virtual bool buildStructure(bool zeroBlocks = false) {
    (void)zeroBlocks;
    devDropLookAhead();
    return buildStructureImpl(_fastPath, _fastPath);
  }

 private:
  // tryBA: merge every EdgeProjectXYZ2UV into one edge set with classes for the BA front end; tryPG: offer the groups to
  // the pose-graph front end.  A front end that refuses (a marginalised camera, a fixed point on the fused path, more than
  // 128 classes, ...) sends the call back here without it: the edges then go to their generic per-key groups.
  bool buildStructureImpl(bool tryBA, bool tryPG) {
    if (!_h || !_optimizer) return false;
    double tSetup = get_monotonic_time(), setupMs[4] = {0., 0., 0., 0.};   // grouping | edge sets + buffers | g2ohip_build_structure | front ends
    unpinAll();                                        // (the buffers below are about to be reallocated)
    if (g2ohip_init(_h) != G2OHIP_OK) return fail("init");
    if (g2ohip_clear_edge_sets(_h) != G2OHIP_OK) return fail("clear_edge_sets");   // (a second optimize(), online growth: a new graph)
    _groups.clear();
    _multi.clear();
    _devValid = false;                                 // (the front ends are about to be bound again: no estimates behind them yet)
    // poses first, then marginalized vertices, each in index order (sparse_optimizer.cpp:174-187)
    _nP = _nL = 0;
    size_t diagDoubles = 0;
    const size_t nV = _optimizer->indexMapping().size();
    {
      // (1.1 M vertex objects at the metric configuration, a cache miss each: walked on the host threads; the order -- poses, then
      // marginalised vertices -- makes a vertex's place in the mirror a function of its index, checked below)
      std::atomic<long long> nPose(0), nLm(0), firstLm((long long)nV), lastPose(-1);
      std::atomic<int> badDim(0);
      parallelFor(nV, [&](size_t b, size_t e) {
        long long np = 0, nl = 0, fl = (long long)nV, lp = -1;
        for (size_t i = b; i < e; ++i) {
          OptimizableGraph::Vertex* v = _optimizer->indexMapping()[i];
          const bool lm = v->marginalized();
          if (v->dimension() != (lm ? l : p)) badDim.store(v->dimension() ? v->dimension() : -1);
          if (lm) { ++nl; fl = std::min(fl, (long long)i); } else { ++np; lp = (long long)i; }
        }
        nPose += np;
        nLm += nl;
        long long cur = firstLm.load();
        while (fl < cur && !firstLm.compare_exchange_weak(cur, fl)) {}
        cur = lastPose.load();
        while (lp > cur && !lastPose.compare_exchange_weak(cur, lp)) {}
      });
      if (badDim.load()) {
        std::cerr << "BlockSolverHip: vertex dimension " << badDim.load() << " does not fit <" << p << "," << l << ">" << std::endl;
        return false;
      }
      _nP = (int)nPose.load();
      _nL = (int)nLm.load();
      if (lastPose.load() >= firstLm.load()) {
        std::cerr << "BlockSolverHip: indexMapping() does not hold the poses before the marginalized vertices" << std::endl;
        return false;
      }
      diagDoubles = (size_t)_nP * p * p + (size_t)_nL * l * l;
    }
    // host mirror of the diagonal blocks: OptimizationAlgorithmLevenberg::computeLambdaInit reads v->hessian(j, j)
    // through the vertices' mapped memory (optimization_algorithm_levenberg.cpp:149-163, base_vertex.h:62-110)
    _diagMirror.assign(diagDoubles, 0.0);
    parallelFor(nV, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        OptimizableGraph::Vertex* v = _optimizer->indexMapping()[i];
        const size_t nP = (size_t)_nP;
        const size_t off = i < nP ? i * p * p : nP * p * p + (i - nP) * l * l;
        v->setColInHessian((int)(i < nP ? i * p : nP * p + (i - nP) * l));   // block_solver.hpp:170,177
        v->mapHessianMemory(&_diagMirror[off]);
      }
    });
    // group the active edges
    std::map<GroupKey, size_t> index;
    int baGroup = -1;
    std::map<ClassKey, int> classIndex;
    ClassKey lastCk = ClassKey();
    int lastClass = -1;
    _baClasses.clear();
    // Large graphs: the EdgeProjectXYZ2UV edges are classified on the host threads first (5 M edges x a handful of virtual calls and
    // typeid comparisons is 0.15 s on one core) -- contiguous chunks of the edge list, every chunk its own class list in order of
    // first appearance, merged in chunk order: the class numbers and the edge order of the sequential loop.  Everything else
    // (and every graph below G2OHIP_ADAPTER_PAR_MIN edges, default 200 000) takes the loop below.
    std::vector<unsigned char> taken;
    {
      const size_t nE = _optimizer->activeEdges().size();
      const char* pm = std::getenv("G2OHIP_ADAPTER_PAR_MIN");
      const size_t parMin = pm ? (size_t)std::atol(pm) : 200000;
      if (tryBA && p == 6 && l == 3 && _threads > 1 && nE >= parMin && nE >= 8192) {
        struct Chunk {
          std::vector<ClassKey> cks;
          std::vector<int32_t> cls, v0, v1;
          std::vector<OptimizableGraph::Edge*> edges;
          bool bad;
          Chunk() : bad(false) {}
        };
        const size_t nt = (size_t)_threads, step = (nE + nt - 1) / nt;   // (the chunking of parallelFor)
        std::vector<Chunk> chunks(nt);
        taken.assign(nE, 0);
        const OptimizableGraph::EdgeContainer& act = _optimizer->activeEdges();
        parallelFor(nE, [&](size_t b, size_t e_) {
          Chunk& c = chunks[b / step];
          int last = -1;
          for (size_t k = b; k < e_; ++k) {
            OptimizableGraph::Edge* e = act[k];
            if (e->vertices().size() != 2) continue;
            OptimizableGraph::Vertex* v0 = static_cast<OptimizableGraph::Vertex*>(e->vertex(0));
            OptimizableGraph::Vertex* v1 = static_cast<OptimizableGraph::Vertex*>(e->vertex(1));
            GroupKey key;
            key.d = e->dimension();
            key.dim0 = v0->dimension();
            key.dim1 = v1->dimension();
            kernelOf(e->robustKernel(), key.kernel, key.delta);
            if (key.kernel < 0) {
              c.bad = true;                                // (reported by the loop below)
              return;
            }
            ClassKey ck;
            if (!projectEdgeClass(e, key, ck)) continue;
            if (last < 0 || ck < c.cks[last] || c.cks[last] < ck) {
              last = -1;
              for (size_t q = 0; q < c.cks.size() && last < 0; ++q)
                if (!(ck < c.cks[q]) && !(c.cks[q] < ck)) last = (int)q;
              if (last < 0) {
                last = (int)c.cks.size();
                c.cks.push_back(ck);
              }
            }
            c.edges.push_back(e);
            c.cls.push_back(last);
            c.v0.push_back(v0->hessianIndex());
            c.v1.push_back(v1->hessianIndex());
            taken[k] = 1;
          }
        });
        bool bad = false;
        size_t total = 0;
        for (size_t c = 0; c < nt; ++c) {
          bad = bad || chunks[c].bad;
          total += chunks[c].edges.size();
        }
        if (bad || total == 0) {
          taken.clear();
        } else {
          baGroup = (int)_groups.size();
          _groups.push_back(Group());
          Group& g = _groups.back();
          g.key.d = 2;
          g.key.dim0 = 3;
          g.key.dim1 = 6;
          g.key.kernel = 0;                                // (the class table owns the kernels)
          g.key.delta = 0.0;
          g.fast = 1;
          g.edges.reserve(total);
          g.cls.reserve(total);
          g.v0.reserve(total);
          g.v1.reserve(total);
          for (size_t c = 0; c < nt; ++c) {
            Chunk& ch = chunks[c];
            std::vector<int> remap(ch.cks.size());
            for (size_t q = 0; q < ch.cks.size(); ++q) {
              typename std::map<ClassKey, int>::iterator ic = classIndex.find(ch.cks[q]);
              if (ic == classIndex.end()) {
                if (classIndex.size() >= 128) return buildStructureImpl(false, tryPG);
                ic = classIndex.insert(std::make_pair(ch.cks[q], (int)classIndex.size())).first;
                const double row[5] = {ch.cks[q].f, ch.cks[q].cx, ch.cks[q].cy, (double)ch.cks[q].kernel, ch.cks[q].delta};
                _baClasses.insert(_baClasses.end(), row, row + 5);
              }
              remap[q] = ic->second;
            }
            g.edges.insert(g.edges.end(), ch.edges.begin(), ch.edges.end());
            g.v0.insert(g.v0.end(), ch.v0.begin(), ch.v0.end());
            g.v1.insert(g.v1.end(), ch.v1.begin(), ch.v1.end());
            for (size_t k = 0; k < ch.cls.size(); ++k) g.cls.push_back(remap[ch.cls[k]]);
          }
        }
      }
    }
    for (size_t k = 0; k < _optimizer->activeEdges().size(); ++k) {
      if (!taken.empty() && taken[k]) continue;
      OptimizableGraph::Edge* e = _optimizer->activeEdges()[k];
      const size_t nv = e->vertices().size();
      if (nv < 1 || nv > (size_t)kMaxMultiVertices) {
        std::cerr << "BlockSolverHip: edges with " << nv << " vertices are not supported (1 to " << kMaxMultiVertices << ")" << std::endl;
        return false;
      }
      if (nv > 2) {
        // BaseMultiEdge (base_multi_edge.hpp:170-222): one binary edge set per vertex pair, see MultiGroup
        MultiGroup* mg = 0;
        int kind;
        double delta;
        kernelOf(e->robustKernel(), kind, delta);
        if (kind < 0) {
          std::cerr << "BlockSolverHip: robust kernel " << typeid(*e->robustKernel()).name() << " has no device counterpart" << std::endl;
          return false;
        }
        for (size_t m = 0; m < _multi.size() && !mg; ++m) {
          bool same = _multi[m].d == e->dimension() && _multi[m].arity == (int)nv;
          for (size_t i = 0; i < nv && same; ++i) same = _multi[m].dims[i] == static_cast<OptimizableGraph::Vertex*>(e->vertex(i))->dimension();
          if (same) mg = &_multi[m];
        }
        if (!mg) {
          _multi.push_back(MultiGroup());
          mg = &_multi.back();
          mg->d = e->dimension();
          mg->arity = (int)nv;
          for (size_t i = 0; i < nv; ++i) mg->dims[i] = static_cast<OptimizableGraph::Vertex*>(e->vertex(i))->dimension();
        }
        mg->edges.push_back(e);
        mg->rkKind.push_back(kind);
        mg->rkDelta.push_back(kind ? delta : 0.0);
        for (size_t i = 0; i < nv; ++i) mg->v[i].push_back(static_cast<OptimizableGraph::Vertex*>(e->vertex(i))->hessianIndex());
        continue;
      }
      OptimizableGraph::Vertex* v0 = static_cast<OptimizableGraph::Vertex*>(e->vertex(0));
      OptimizableGraph::Vertex* v1 = nv == 2 ? static_cast<OptimizableGraph::Vertex*>(e->vertex(1)) : 0;
      GroupKey key;
      key.d = e->dimension();
      key.dim0 = v0->dimension();
      key.dim1 = v1 ? v1->dimension() : 0;
      kernelOf(e->robustKernel(), key.kernel, key.delta);
      if (key.kernel < 0) {
        std::cerr << "BlockSolverHip: robust kernel " << typeid(*e->robustKernel()).name() << " has no device counterpart" << std::endl;
        return false;
      }
      ClassKey ck;
      if (tryBA && projectEdgeClass(e, key, ck)) {         // one set for all of them, the differences become classes
        if (baGroup < 0) {
          baGroup = (int)_groups.size();
          _groups.push_back(Group());
          _groups.back().key = key;
          _groups.back().key.kernel = 0;                   // (the class table owns the kernels)
          _groups.back().key.delta = 0.0;
          _groups.back().fast = 1;
        }
        int cid;
        if (lastClass >= 0 && !(ck < lastCk) && !(lastCk < ck)) {   // (runs of edges of one class: no map lookup)
          cid = lastClass;
        } else {
          typename std::map<ClassKey, int>::iterator ic = classIndex.find(ck);
          if (ic == classIndex.end()) {
            if (classIndex.size() >= 128) return buildStructureImpl(false, tryPG);
            ic = classIndex.insert(std::make_pair(ck, (int)classIndex.size())).first;
            const double row[5] = {ck.f, ck.cx, ck.cy, (double)ck.kernel, ck.delta};
            _baClasses.insert(_baClasses.end(), row, row + 5);
          }
          cid = ic->second;
          lastCk = ck;
          lastClass = cid;
        }
        Group& g = _groups[baGroup];
        g.edges.push_back(e);
        g.cls.push_back(cid);
        g.v0.push_back(v0->hessianIndex());
        g.v1.push_back(v1->hessianIndex());
        continue;
      }
      typename std::map<GroupKey, size_t>::iterator it = index.find(key);
      if (it == index.end()) {
        it = index.insert(std::make_pair(key, _groups.size())).first;
        _groups.push_back(Group());
        _groups.back().key = key;
      }
      Group& g = _groups[it->second];
      g.edges.push_back(e);
      g.rkKind.push_back(key.kernel);
      g.rkDelta.push_back(key.kernel ? key.delta : 0.0);
      g.v0.push_back(v0->hessianIndex());              // -1 when fixed (optimizable_graph.h:299)
      if (v1) g.v1.push_back(v1->hessianIndex());
    }
    setupMs[0] = 1e3 * lap(tSetup);
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      Group& g = _groups[gi];
      const int n = (int)g.edges.size();
      g.set = g2ohip_add_edge_set(_h, g.key.d, n, g.v0.data(), g.key.dim1 ? g.v1.data() : 0);
      if (g.set < 0) return fail("add_edge_set");
      if (!g.fast) {                                     // one kernel for the set where the edges agree, else one per edge
        bool uniform = true;
        for (size_t k = 1; k < g.rkKind.size() && uniform; ++k) uniform = g.rkKind[k] == g.rkKind[0] && g.rkDelta[k] == g.rkDelta[0];
        if (uniform && !g.rkKind.empty() && g.rkKind[0] > 0) {
          if (g2ohip_set_robust_kernel(_h, g.set, g.rkKind[0], g.rkDelta[0]) != G2OHIP_OK) return fail("set_robust_kernel");
        } else if (!uniform) {
          if (g2ohip_set_robust_kernel_per_edge(_h, g.set, g.rkKind.data(), g.rkDelta.data()) != G2OHIP_OK) return fail("set_robust_kernel_per_edge");
        }
      }
      // (the staging buffers of the generic path -- 0.96 GB at the metric configuration -- are allocated by the first buildSystem
      // that finds the group still generic: a group on a device front end never needs them)
      g.J0.clear();
      g.J1.clear();
      g.Om.clear();
      g.err.clear();
    }
    for (size_t m = 0; m < _multi.size(); ++m) {
      MultiGroup& mg = _multi[m];
      const int n = (int)mg.edges.size();
      bool uniform = true;
      for (size_t k = 1; k < mg.rkKind.size() && uniform; ++k) uniform = mg.rkKind[k] == mg.rkKind[0] && mg.rkDelta[k] == mg.rkDelta[0];
      mg.pairs.clear();
      for (int i = 0; i < mg.arity; ++i)
        for (int j = i + 1; j < mg.arity; ++j) {
          typename MultiGroup::Pair pr;
          pr.i = i;
          pr.j = j;
          // vertex i's own terms (H_ii, b_i) come from the pair (i, i + 1), the last vertex's from the pair (arity - 2, arity - 1),
          // chi2 from the pair (0, 1): everything once, as base_multi_edge.hpp:170-222 adds it
          pr.parts = (j == i + 1 ? 0 : G2OHIP_PART_NO_VERTEX0) | ((i == mg.arity - 2 && j == mg.arity - 1) ? 0 : G2OHIP_PART_NO_VERTEX1) |
                     ((i == 0 && j == 1) ? 0 : G2OHIP_PART_NO_CHI2);
          pr.set = g2ohip_add_edge_set(_h, mg.d, n, mg.v[i].data(), mg.v[j].data());
          if (pr.set < 0) return fail("add_edge_set");
          if (g2ohip_set_edge_set_parts(_h, pr.set, pr.parts) != G2OHIP_OK) return fail("set_edge_set_parts");
          if (uniform && !mg.rkKind.empty() && mg.rkKind[0] > 0) {
            if (g2ohip_set_robust_kernel(_h, pr.set, mg.rkKind[0], mg.rkDelta[0]) != G2OHIP_OK) return fail("set_robust_kernel");
          } else if (!uniform) {
            if (g2ohip_set_robust_kernel_per_edge(_h, pr.set, mg.rkKind.data(), mg.rkDelta.data()) != G2OHIP_OK) return fail("set_robust_kernel_per_edge");
          }
          mg.pairs.push_back(pr);
        }
      mg.Om.clear();
      mg.err.clear();
    }
    setupMs[1] = 1e3 * lap(tSetup);
    // The walk over the projection edges that the device front end needs (vertex tables in order of first appearance,
    // measurements, information matrices: 0.15 s of pointer chasing over 5 M edge objects at the metric configuration) reads the
    // graph only: it runs on a thread of its own next to g2ohip_build_structure (0.45 s of host work on the index arrays).
    std::thread gatherThread;
    if (baGroup >= 0 && _threads > 1) gatherThread = std::thread(&BlockSolverHip::gatherProjectXYZ2UV, this, &_groups[baGroup]);
    const bool structureOk = g2ohip_build_structure(_h, _nP, _nL, _doSchur ? 1 : 0) == G2OHIP_OK;
    if (gatherThread.joinable()) gatherThread.join();
    else if (baGroup >= 0) gatherProjectXYZ2UV(&_groups[baGroup]);
    if (!structureOk) return fail("build_structure");
    setupMs[2] = 1e3 * lap(tSetup);
    resizeVector(g2ohip_vector_size(_h));              // Solver::_x, _b (solver.cpp:46-70)
    _diag.assign(g2ohip_vector_size(_h), 0.0);
    pinDoubles(_x, g2ohip_vector_size(_h));            // what crosses PCIe every iteration is page-locked once (g2ohip_host_register)
    pinDoubles(_b, g2ohip_vector_size(_h));
    pinDoubles(_diag.data(), _diag.size());
    _fastGroups = 0;
    if (baGroup >= 0) {
      if (!bindProjectXYZ2UV(_groups[baGroup])) return buildStructureImpl(false, tryPG);   // everything generic (the set is re-registered per key)
      ++_fastGroups;
    }
    if (tryPG)
      for (size_t gi = 0; gi < _groups.size(); ++gi) {     // the pose-graph front end holds one set
        Group& g = _groups[gi];
        if (g.fast) continue;
        if (bindSE2(g)) g.fast = 2;
        else if (bindSE3(g)) g.fast = 3;
        if (g.fast) {
          ++_fastGroups;
          break;
        }
      }
    planHybrid();
    setupMs[3] = 1e3 * lap(tSetup);
    if (_timing)
      std::cerr << "{\"g2ohip_adapter_setup_ms\": {\"grouping\": " << setupMs[0] << ", \"edge_sets_and_buffers\": " << setupMs[1]
                << ", \"build_structure\": " << setupMs[2] << ", \"front_ends_and_pinning\": " << setupMs[3] << "}}" << std::endl;
    return true;
  }

 public:

  //! device-resident front end for homogeneous EdgeProjectXYZ2UV groups (on by default)
  void setFastPath(bool on) { _fastPath = on; }
  bool fastPathActive() const { return _fastGroups > 0; }
  int fastGroups() const { return _fastGroups; }       // edge sets bound to a device front end (BA: 1 for all its classes)

  // Online growth (block_solver.hpp:297-351).  SparseOptimizer::updateInitialization has appended the new vertices to
  // indexMapping() and the new edges to activeEdges() before it calls this (sparse_optimizer.cpp:269-352), so the index
  // arrays are simply read again: the library rebuilds contributor lists, pattern and symbolic analysis from them.  With
  // marginalised vertices the reference aborts (:313-316); here the call reports failure.
  virtual bool updateStructure(const std::vector<HyperGraph::Vertex*>& vset, const HyperGraph::EdgeSet& edges) {
    (void)edges;
    devDropLookAhead();
    for (size_t i = 0; i < vset.size(); ++i)
      if (static_cast<OptimizableGraph::Vertex*>(vset[i])->marginalized()) {
        std::cerr << "updateStructure(): Schur not supported" << std::endl;
        return false;
      }
    if (_doSchur && _nL > 0) {
      std::cerr << "updateStructure(): Schur not supported" << std::endl;
      return false;
    }
    return buildStructure(false);
  }

  // block_solver.hpp:501-560.  After it returns b() holds -J' Omega e (poses then landmarks) and H is assembled on
  // the device.  The reference returns 0 and nobody checks (optimization_algorithm_levenberg.cpp:81).
  // The groups no device front end takes (and the n-ary edges): errors / Jacobians by the edges' own computeError /
  // linearizeOplus on the host, flat arrays to the library.  computeErrors: the caller is a device-resident driver (nobody ran
  // computeActiveErrors on the host); jacobians = false: the errors alone (chi2 of a Levenberg-Marquardt trial).
  bool hostGeneric(bool computeErrors, bool jacobians) {
    JacobianWorkspace& ws = _optimizer->jacobianWorkspace();
    double t = get_monotonic_time();
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      Group& g = _groups[gi];
      if (g.fast) continue;
      const int d = g.key.d, d0 = g.key.dim0, d1 = g.key.dim1;
      if (g.err.empty() && !g.edges.empty()) {
        // The library takes a side's dimension from the hessian indices; a side whose vertices are ALL fixed (e.g. a
        // localisation graph over fixed points) is taken as a pose side there (p columns), whatever the vertex type: the
        // buffers are sized for the larger of the two so that g2ohip_set_edge_data never reads past their end.
        const size_t n = g.edges.size();
        const int w0 = d0 > p ? d0 : p, w1 = d1 ? (d1 > p ? d1 : p) : 0;
        g.J0.assign(n * d * w0, 0.0);
        g.J1.assign(n * d * w1, 0.0);
        g.Om.assign(n * d * d, 0.0);
        g.err.assign(n * d, 0.0);
      }
      for (size_t k = 0; k < g.edges.size(); ++k) {
        OptimizableGraph::Edge* e = g.edges[k];
        if (computeErrors) e->computeError();
        if (!jacobians) {
          std::memcpy(&g.err[k * d], e->errorData(), sizeof(double) * d);
          continue;
        }
        e->linearizeOplus(ws);                         // block_solver.hpp:531
        // Jacobians sit column-major (d x dim) in the workspace (base_binary_edge.h: Map onto workspaceForVertex(i));
        // a fixed vertex has no Jacobian and is never read on the device (index -1)
        if (g.v0[k] >= 0) std::memcpy(&g.J0[k * d * d0], ws.workspaceForVertex(0), sizeof(double) * d * d0);
        if (d1 && g.v1[k] >= 0) std::memcpy(&g.J1[k * d * d1], ws.workspaceForVertex(1), sizeof(double) * d * d1);
        std::memcpy(&g.Om[k * d * d], e->informationData(), sizeof(double) * d * d);
        std::memcpy(&g.err[k * d], e->errorData(), sizeof(double) * d);
      }
      _phase.hostLinearize += lap(t);
      if (!jacobians) {
        if (g2ohip_set_edge_errors(_h, g.set, g.err.data()) != G2OHIP_OK) return fail("set_edge_errors");
      } else if (g2ohip_set_edge_data(_h, g.set, g.J0.data(), d1 ? g.J1.data() : 0, g.Om.data(), g.err.data(), /*on_device*/ 0) != G2OHIP_OK) {
        return fail("set_edge_data");
      }
      _phase.upload += lap(t);
    }
    for (size_t m = 0; m < _multi.size(); ++m) {         // n-ary edges: linearised once, their Jacobians dealt to the pair sets
      MultiGroup& mg = _multi[m];
      const int d = mg.d;
      const size_t n = mg.edges.size();
      if (mg.err.empty() && n) {
        mg.Om.assign(n * d * d, 0.0);
        mg.err.assign(n * d, 0.0);
        for (size_t q = 0; q < mg.pairs.size(); ++q) {
          typename MultiGroup::Pair& pr = mg.pairs[q];
          pr.J0.assign(n * d * (mg.dims[pr.i] > p ? mg.dims[pr.i] : p), 0.0);   // (sized as the generic groups' buffers, above)
          pr.J1.assign(n * d * (mg.dims[pr.j] > p ? mg.dims[pr.j] : p), 0.0);
        }
      }
      for (size_t k = 0; k < n; ++k) {
        OptimizableGraph::Edge* e = mg.edges[k];
        if (computeErrors) e->computeError();
        if (!jacobians) {
          std::memcpy(&mg.err[k * d], e->errorData(), sizeof(double) * d);
          continue;
        }
        e->linearizeOplus(ws);
        for (size_t q = 0; q < mg.pairs.size(); ++q) {
          typename MultiGroup::Pair& pr = mg.pairs[q];
          const int di = mg.dims[pr.i], dj = mg.dims[pr.j];
          if (mg.v[pr.i][k] >= 0) std::memcpy(&pr.J0[k * d * di], ws.workspaceForVertex(pr.i), sizeof(double) * d * di);
          if (mg.v[pr.j][k] >= 0) std::memcpy(&pr.J1[k * d * dj], ws.workspaceForVertex(pr.j), sizeof(double) * d * dj);
        }
        std::memcpy(&mg.Om[k * d * d], e->informationData(), sizeof(double) * d * d);
        std::memcpy(&mg.err[k * d], e->errorData(), sizeof(double) * d);
      }
      _phase.hostLinearize += lap(t);
      for (size_t q = 0; q < mg.pairs.size(); ++q) {
        typename MultiGroup::Pair& pr = mg.pairs[q];
        if (!jacobians) {
          if (g2ohip_set_edge_errors(_h, pr.set, mg.err.data()) != G2OHIP_OK) return fail("set_edge_errors");
        } else if (g2ohip_set_edge_data(_h, pr.set, pr.J0.data(), pr.J1.data(), mg.Om.data(), mg.err.data(), /*on_device*/ 0) != G2OHIP_OK) {
          return fail("set_edge_data");
        }
      }
      _phase.upload += lap(t);
    }
    return true;
  }

  virtual bool buildSystem() {
    if (!_h) return false;
    devDropLookAhead();
    // (the host-linearised groups first: handing a set its data for the first time drops the front ends' cached evaluations)
    if (!hostGeneric(/*computeErrors=*/false, /*jacobians=*/true)) return false;   // (the errors are current: computeActiveErrors)
    double t = get_monotonic_time();
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      Group& g = _groups[gi];
      if (g.fast) {                                    // estimates up, errors + Jacobians on the device
        if (!(g.fast == 2 ? uploadPosesSE2() : (g.fast == 3 ? uploadPosesSE3() : uploadEstimates()))) return false;
        _devValid = true;
        _phase.upload += lap(t);
      }
    }
    if (g2ohip_build_system(_h) != G2OHIP_OK) return fail("build_system");
    if (_timing) g2ohip_sync(_h);
    _phase.deviceBuild += lap(t);
    if (g2ohip_copy_b(_h, _b) != G2OHIP_OK) return fail("copy_b");          // LM reads b() (levenberg.cpp:169)
    refreshDiagonalMirror();
    _phase.downloadB += lap(t);
    ++_phase.buildSystems;
    return true;
  }

  // block_solver.hpp:353-486: x() <- solution of (H (+ lambda I)) x = b, full length; b() untouched; false iff not
  // positive definite
  // Gauss-Newton drop-in on numerically singular systems (round 6).  The reference's up-looking Cholesky under block-AMD gets through
  // an undamped long camera chain (kappa x eps >= 1) where the pivot test of csparse_helper.cpp:136 happens to pass, and
  // OptimizationAlgorithmGaussNewton applies that step (optimization_algorithm_gauss_newton.cpp:73-85); the nested-dissection
  // factorisation meets d <= 0 there and used to return Fail.  An UNDAMPED solve (no setLambda in force: Gauss-Newton, Dogleg's
  // Gauss-Newton step) that breaks down is therefore repeated ONCE with lambda = 1e-14 x the largest diagonal entry -- from there on
  // both solvers succeed and agree to kappa x eps (profiles/r5_gn_lambda0.txt) -- and says so on cerr.  Levenberg-Marquardt trials
  // (setLambda in force) keep the reference's behaviour: solve() == false, the driver raises lambda.
  int solveUndampedRetry() {
    int rc = g2ohip_solve(_h);
    if (rc == G2OHIP_NOT_PD && !_lambdaInForce) {
      double md = 0.;
      if (g2ohip_max_diagonal(_h, &md) == G2OHIP_OK && md > 0.) {
        const double lam = 1e-14 * md;
        std::cerr << "HipBlockSolver: the undamped system is numerically singular (non-positive pivot); solving once more with lambda = "
                  << lam << " (1e-14 x the largest diagonal entry)" << std::endl;
        if (g2ohip_set_lambda(_h, lam, 1) == G2OHIP_OK) {
          rc = g2ohip_solve(_h);
          g2ohip_restore_diagonal(_h);
          ++_undampedRetries;
        }
      }
    }
    return rc;
  }
  int undampedRetries() const { return _undampedRetries; }

  virtual bool solve() {
    if (!_h) return false;
    devDropLookAhead();
    double t = get_monotonic_time();
    const int rc = solveUndampedRetry();
    if (rc != G2OHIP_OK) {
      if (rc != G2OHIP_NOT_PD) fail("solve");
      else if (_writeDebug) {   // linear_solver_csparse.h:127-133: the matrix the Cholesky was given, loadable by Octave
        std::cerr << "Cholesky failure, writing debug.txt (Hessian loadable by Octave)" << std::endl;
        writeOctave("debug.txt", (_doSchur && _nL > 0) ? G2OHIP_HSCHUR : G2OHIP_HPP, /*fixed=*/false);
      }
      return false;
    }
    _phase.deviceSolve += lap(t);
    if (g2ohip_copy_x(_h, _x) != G2OHIP_OK) return fail("copy_x");
    _phase.downloadX += lap(t);
    ++_phase.solves;
    G2OBatchStatistics* gs = G2OBatchStatistics::globalStats();
    if (gs) {                                          // batch_stats.h:40-77
      g2ohip_stats st;
      if (g2ohip_get_stats(_h, &st) == G2OHIP_OK) {
        gs->timeSchurComplement = st.timeSchurComplement;
        gs->timeSymbolicDecomposition = st.timeSymbolicDecomposition;
        gs->timeNumericDecomposition = st.timeNumericDecomposition;
        gs->timeLinearSolver = st.timeLinearSolver;
        gs->choleskyNNZ = st.choleskyNNZ;
        gs->hessianPoseDimension = st.hessianPoseDimension;
        gs->hessianLandmarkDimension = st.hessianLandmarkDimension;
        gs->hessianDimension = st.hessianPoseDimension + st.hessianLandmarkDimension;
      }
      // (timeLinearSolution is accumulated by the algorithm around solve(), levenberg.cpp:105)
    }
    return true;
  }

  // block_solver.hpp:563-604: lambda on every scalar diagonal entry of Hpp and Hll; exact restore
  virtual bool setLambda(double lambda, bool backup = false) {
    devDropLookAhead();
    _lambdaInForce = true;
    return _h && g2ohip_set_lambda(_h, lambda, backup ? 1 : 0) == G2OHIP_OK;
  }
  virtual void restoreDiagonal() {
    devDropLookAhead();
    _lambdaInForce = false;
    if (_h) g2ohip_restore_diagonal(_h);
  }

  // block_solver.hpp:489-498: blocks of the inverse of Hpp
  virtual bool computeMarginals(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices) {
    if (!_h) return false;
    devDropLookAhead();
    const double t = get_monotonic_time();
    std::vector<int32_t> r(blockIndices.size()), c(blockIndices.size());
    for (size_t i = 0; i < blockIndices.size(); ++i) {
      r[i] = blockIndices[i].first;
      c[i] = blockIndices[i].second;
    }
    std::vector<double> out(blockIndices.size() * p * p);
    if (g2ohip_compute_marginals(_h, (int)blockIndices.size(), r.data(), c.data(), out.data()) != G2OHIP_OK) return false;
    // the block layout of the result, as MarginalCovarianceCholesky::computeCovariance sets it up
    // (marginal_covariance_cholesky.cpp:156-160): one p x p block per pose
    std::vector<int> rbi(_nP);
    for (int i = 0; i < _nP; ++i) rbi[i] = (i + 1) * p;
    if (_nP > 0) spinv = SparseBlockMatrix<MatrixXd>(&rbi[0], &rbi[0], _nP, _nP, true);
    for (size_t i = 0; i < blockIndices.size(); ++i) {
      MatrixXd* blk = spinv.block(r[i], c[i], true);   // allocated p x p, column-major like `out`
      std::memcpy(blk->data(), &out[i * p * p], sizeof(double) * p * p);
    }
    G2OBatchStatistics* gs = G2OBatchStatistics::globalStats();
    if (gs) gs->timeMarginals = get_monotonic_time() - t;
    return true;
  }

  virtual bool supportsSchur() { return true; }
  virtual bool schur() { return _doSchur; }
  virtual void setSchur(bool s) { _doSchur = s; }
  virtual void setWriteDebug(bool b) { _writeDebug = b; }
  virtual bool writeDebug() const { return _writeDebug; }
  // block_solver.hpp:628-632: _Hpp->writeOctave(fileName, true) -- the upper blocks of Hpp as an Octave sparse matrix (both triangles,
  // sorted by column, nine fixed digits: sparse_block_matrix.hpp:548-589)
  virtual bool saveHessian(const std::string& fileName) const {
    const_cast<BlockSolverHip*>(this)->devDropLookAhead();
    return _h && writeOctave(fileName, G2OHIP_HPP, /*fixed=*/true);
  }
  // BlockSolverBase (block_solver.h:83-91), used by OptimizationAlgorithmDogleg: dest = H * src
  virtual void multiplyHessian(double* dest, const double* src) const {
    const_cast<BlockSolverHip*>(this)->devDropLookAhead();
    if (_h) g2ohip_multiply_hessian(_h, dest, src);
  }

  // ---- HipDeviceGraph: the graph side of an iteration on the device front ends (g2ohip_ba_* / g2ohip_pg_*)
  // every active edge on a device front end -- or (hybrid) the bundle-adjustment front end plus groups the host linearises
  // whose free vertices are all cameras / points of that front end
  virtual bool deviceResident() const { return _h && !_groups.empty() && ((_multi.empty() && _fastGroups == (int)_groups.size()) || _hybrid); }
  bool hybridLoop() const { return _hybrid; }
  virtual bool devEstimatesValid() const { return _devValid; }
  virtual bool devSetEstimates() {
    double t = get_monotonic_time();
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      const int fast = _groups[gi].fast;
      if (!fast) continue;                               // (hybrid loop: a host-linearised group has nothing on the device)
      if (!(fast == 1 ? setEstimatesBA() : (fast == 2 ? setPosesSE2() : setPosesSE3()))) return false;
    }
    _devValid = true;
    _phase.upload += lap(t);
    return true;
  }
  virtual bool devGetEstimates() {
    double t = get_monotonic_time();
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      const int fast = _groups[gi].fast;
      if (!fast) continue;
      if (!(fast == 1 ? getEstimatesBA() : (fast == 2 ? getPosesSE2() : getPosesSE3()))) return false;
    }
    _phase.downloadX += lap(t);
    return true;
  }
  // (pipelined write-back: the bundle-adjustment group's estimates in pieces -- cameras, then four point ranges; a pose-graph
  // group next to it is small and read back whole at the end)
  virtual bool devFetchBegin() {
#if G2OHIP_FASTPATH_SBA
    if (!_pin) return false;                           // (pageable buffers: the "asynchronous" copy would block the caller)
    for (size_t gi = 0; gi < _groups.size(); ++gi)
      if (_groups[gi].fast == 1) {
        if (g2ohip_ba_fetch_estimates_begin(_h, _camBuf.data(), _pointBuf.data(), kFetchPieces) != G2OHIP_OK) return fail("ba_fetch_estimates_begin");
        _fetchBegun = true;
        return true;
      }
#endif
    return false;
  }
  virtual void devFetchCancel() {
    (void)g2ohip_ba_fetch_estimates_wait(_h, kFetchPieces);
    _fetchBegun = false;
  }
  virtual bool devFetchEnd() {
    double t = get_monotonic_time();
    _fetchBegun = false;
#if G2OHIP_FASTPATH_SBA
    // ONE parallel region: the calling thread waits for the pieces and announces them, every thread writes its share of a
    // piece as soon as it is there (a region per piece paid the hand-out five times)
    {
      const size_t nc = _cams.size(), np = _points.size(), step = (np + kFetchPieces - 1) / kFetchPieces;
      const size_t nt = (size_t)(_threads > 1 ? _threads : 1);
      std::atomic<int> arrived(0), bad(0);
      double waited = 0.;
      _workers.run(nt, nt, [&, nc, np, step, nt](size_t tb, size_t) {
        const size_t tid = tb;
        for (int piece = 0; piece <= kFetchPieces; ++piece) {
          if (tid == 0) {
            const double tw = get_monotonic_time();
            if (g2ohip_ba_fetch_estimates_wait(_h, piece) != G2OHIP_OK) bad.store(1);
            waited += get_monotonic_time() - tw;
            arrived.store(piece + 1, std::memory_order_release);
          } else {
            while (arrived.load(std::memory_order_acquire) <= piece) std::this_thread::yield();
          }
          if (bad.load()) return;
          const size_t first = piece == 0 ? 0 : std::min(np, (piece - 1) * step), last = piece == 0 ? nc : std::min(np, first + step);
          const size_t n = last - first, chunk = (n + nt - 1) / nt, b = std::min(n, tid * chunk), e = std::min(n, b + chunk);
          if (piece == 0) scatterCamsRange(b, e);
          else scatterPointsRange(first + b, first + e);
        }
      });
      if (bad.load()) return fail("ba_fetch_estimates_wait");
      _phase.fetchWait += waited;
      _phase.fetchPoints += get_monotonic_time() - t - waited;
    }
    ++_phase.fetches;
#endif
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      const int fast = _groups[gi].fast;
      if (fast == 2 ? !getPosesSE2() : (fast == 3 ? !getPosesSE3() : false)) return false;
    }
    _phase.downloadX += lap(t);
    return true;
  }
  virtual bool devLinearize(bool jacobians) {
    if (_hybrid) {
      // Hybrid loop: the groups no front end takes are linearised by their own edge types on the host.  With the Jacobians
      // (iteration start) the vertices hold the current estimates (write-back of the previous iteration); for the errors of a
      // TRIAL the vertices those edges touch -- usually a handful -- get the trial estimates from the device first, under a
      // push() that devPop / devDiscardTop resolve with the device's own stack.
      if (!jacobians && !refreshTouched()) return false;
      if (!hostGeneric(/*computeErrors=*/true, jacobians)) return false;
      // the errors of a LOOK-AHEAD trial: the caller must not find the speculative estimates in its vertices (nor their errors in
      // its edges) when solve() returns -- the touched vertices go back to the accepted estimates at once; if the next solve()
      // accepts the trial, devDiscardTop writes the cached trial estimates into them
      if (!jacobians && _queueing) touchedBackToAccepted();
    }
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      const int fast = _groups[gi].fast;
      if (!fast) continue;
      if (fast == 1 ? g2ohip_ba_linearize(_h, jacobians ? 1 : 0) != G2OHIP_OK : g2ohip_pg_linearize(_h, jacobians ? 1 : 0) != G2OHIP_OK) return fail("linearize");
    }
    return true;
  }
  virtual bool devChi2(double& chi2) { return g2ohip_chi2(_h, &chi2) == G2OHIP_OK || fail("chi2"); }
  virtual bool devBuildSystem() {
    double t = get_monotonic_time();
    if (g2ohip_build_system(_h) != G2OHIP_OK) return fail("build_system");
    if (_timing && !_queueing) g2ohip_sync(_h);
    _phase.deviceBuild += lap(t);
    ++_phase.buildSystems;
    return true;
  }
  // look-ahead (see HipDeviceGraph): only the bundle-adjustment front end with its pipelined write-back -- a pose-graph group's
  // estimates are read back whole at the end of devFetchEnd, i.e. AFTER a queued update would have moved them.  The hybrid loop's
  // host groups read only the vertices they touch, and those hold the accepted estimates already (refreshTouched of the accepted
  // trial, kept by discardTop): the driver queues the trial up to its update, writes back, and evaluates the host errors then.
  virtual bool devCanLookAhead() const {
    if (!_lookEnabled || !_pin || _groups.empty()) return false;
    if (!_hybrid && !_multi.empty()) return false;
    bool ba = false;
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      if (_groups[gi].fast == 1) ba = true;
      else if (_groups[gi].fast != 0 || !_hybrid) return false;
    }
    return ba;
  }
  virtual bool devHybrid() const { return _hybrid; }
  virtual bool devTrialStatsBegin(double lambda) { return g2ohip_trial_stats_begin(_h, lambda) == G2OHIP_OK || fail("trial_stats_begin"); }
  virtual bool devLookAheadPending() const { return _lookPending; }
  virtual void devSetLookAheadPending(bool on) {
    _lookPending = on;
    if (on) ++_lookQueued;
  }
  virtual void devSetQueueing(bool on) { _queueing = on; }
  virtual void devDropLookAhead() {
    if (!_lookPending) return;
    _lookPending = false;
    int ok = 0;
    double chi2 = 0., scale = 0.;
    const bool drained = g2ohip_trial_stats(_h, 0., &ok, &chi2, &scale) == G2OHIP_OK;   // (drains the queued read-back)
    if (_fetchBegun) devFetchCancel();
    const bool popped = devPop();                       // the device holds what the vertices hold again
    if (!drained || !popped) _devValid = false;         // ... unless one of the two failed: the next solve() uploads the estimates afresh
    ++_lookDropped;
  }
  int lookAheadsDropped() const { return _lookDropped; }
  virtual bool devMaxDiagonal(double& d) { return g2ohip_max_diagonal(_h, &d) == G2OHIP_OK || fail("max_diagonal"); }
  virtual bool devComputeScale(double lambda, double& scale) { return g2ohip_compute_scale(_h, lambda, &scale) == G2OHIP_OK || fail("compute_scale"); }
  virtual int devSolve() {
    double t = get_monotonic_time();
    const int rc = solveUndampedRetry();
    _phase.deviceSolve += lap(t);
    ++_phase.solves;
    if (rc == G2OHIP_OK) return 1;
    if (rc == G2OHIP_NOT_PD) return 0;
    fail("solve");
    return -1;
  }
  virtual bool devSolveAsync() {
    ++_phase.solves;
    return g2ohip_solve_async(_h) == G2OHIP_OK || fail("solve_async");
  }
  virtual int devTrialStats(double lambda, double& chi2, double& scale) {
    double t = get_monotonic_time();
    int ok = 0;
    if (g2ohip_trial_stats(_h, lambda, &ok, &chi2, &scale) != G2OHIP_OK) {
      fail("trial_stats");
      return -1;
    }
    _phase.deviceSolve += lap(t);                      // (the one synchronisation of a trial: solve, update and errors end here)
    return ok;
  }
  virtual bool devUpdate() { return forEachFrontEnd(g2ohip_ba_update, g2ohip_pg_update, "update"); }
  virtual bool devPush() { return forEachFrontEnd(g2ohip_ba_push, g2ohip_pg_push, "push"); }
  virtual bool devPop() {
    if (_touchedPushed)
      for (size_t i = 0; i < _touched.size(); ++i) _touched[i].v->pop();
    _touchedPushed = false;
    _touchedSpec = false;                                // (a look-ahead trial that was rejected or dropped: the vertices never saw it)
    return forEachFrontEnd(g2ohip_ba_pop, g2ohip_pg_pop, "pop");
  }
  virtual bool devDiscardTop() {
    if (_touchedPushed)
      for (size_t i = 0; i < _touched.size(); ++i) _touched[i].v->discardTop();
    _touchedPushed = false;
    if (_touchedSpec) {                                  // an accepted look-ahead trial: its estimates of the touched vertices, from the cache
      setTouchedFromCache();
      hostErrorsOnly();
      _touchedSpec = false;
    }
    return forEachFrontEnd(g2ohip_ba_discard_top, g2ohip_pg_discard_top, "discard_top");
  }

  g2ohip_solver* handle() const { return _h; }

 private:
  struct GroupKey {
    int d, dim0, dim1, kernel;
    double delta;
    // (kernel / delta: of the edge being classified -- NOT part of the ordering: in g2o the robust kernel belongs to the edge,
    // so edges that differ in it only share a group and the group hands the library one kernel per edge)
    bool operator<(const GroupKey& o) const {
      if (d != o.d) return d < o.d;
      if (dim0 != o.dim0) return dim0 < o.dim0;
      return dim1 < o.dim1;
    }
  };
  struct Group {
    GroupKey key;
    int set;
    int fast;                                            // 0: generic path; device front end 1: EdgeProjectXYZ2UV (g2ohip_ba_*), 2 / 3: EdgeSE2 / EdgeSE3 (g2ohip_pg_*, type 1 / 2)
    std::vector<OptimizableGraph::Edge*> edges;
    std::vector<int32_t> v0, v1, cls, rkKind;            // (cls: edge class of a BA group's edges; rkKind / rkDelta: robust kernel per edge)
    std::vector<double> rkDelta;
    std::vector<double> J0, J1, Om, err;
    Group() : key(), set(-1), fast(0) {}
  };
  // n-ary edges of one shape (error dimension, number of vertices, their dimensions): BaseMultiEdge::constructQuadraticForm
  // (base_multi_edge.hpp:170-222) adds H_ii, b_i per vertex and H_ij per vertex pair; here every pair (i, j) is ONE binary edge
  // set of the library over the edges' vertices i and j, and g2ohip_set_edge_set_parts switches off what another pair of the
  // same edges already contributes.  The host linearises an edge once and deals its Jacobians to the pair sets.
  enum { kMaxMultiVertices = 4 };
  struct MultiGroup {
    struct Pair {
      int i, j, set, parts;
      std::vector<double> J0, J1;
    };
    int d, arity, dims[kMaxMultiVertices];
    std::vector<OptimizableGraph::Edge*> edges;
    std::vector<int32_t> v[kMaxMultiVertices], rkKind;
    std::vector<double> rkDelta, Om, err;
    std::vector<Pair> pairs;
  };
  struct ClassKey {                                      // what distinguishes two EdgeProjectXYZ2UV on the device: intrinsics + robust kernel
    double f, cx, cy, delta;
    int kernel;
    bool operator<(const ClassKey& o) const {
      if (f != o.f) return f < o.f;
      if (cx != o.cx) return cx < o.cx;
      if (cy != o.cy) return cy < o.cy;
      if (kernel != o.kernel) return kernel < o.kernel;
      return delta < o.delta;
    }
  };

  // robust kernel -> the kind numbers of g2ohip_set_robust_kernel (robust_kernel_impl.h:77-140); 0: none, -1: unknown
  static void kernelOf(const RobustKernel* k, int& kind, double& delta) {
    kind = 0;
    delta = 0.0;
    if (!k) return;
    delta = k->delta();
    if (dynamic_cast<const RobustKernelHuber*>(k)) kind = 1;
    else if (dynamic_cast<const RobustKernelPseudoHuber*>(k)) kind = 2;
    else if (dynamic_cast<const RobustKernelCauchy*>(k)) kind = 3;
    else if (dynamic_cast<const RobustKernelSaturated*>(k)) kind = 4;
    else if (dynamic_cast<const RobustKernelDCS*>(k)) kind = 5;
    else kind = -1;
  }

  void refreshDiagonalMirror() {
    if (g2ohip_copy_diagonal(_h, _diag.data()) != G2OHIP_OK) return;
    size_t off = 0, s = 0;
    for (size_t i = 0; i < _optimizer->indexMapping().size(); ++i) {
      const int dim = _optimizer->indexMapping()[i]->dimension();
      for (int j = 0; j < dim; ++j) _diagMirror[off + (size_t)j * (dim + 1)] = _diag[s + j];
      off += (size_t)dim * dim;
      s += dim;
    }
  }

  // vertex -> position in an estimate table, in order of first appearance: free vertices are keyed by their hessianIndex (an
  // array lookup), only the fixed ones -- they have none -- go through a map (5 M std::map lookups cost a second at the metric size)
  struct SlotTable {
    std::vector<int> byHidx;
    std::map<const HyperGraph::Vertex*, int> fixed;
    int count;
    explicit SlotTable(size_t n) : byHidx(n, -1), count(0) {}
    int slot(const OptimizableGraph::Vertex* v, bool& isNew) {
      const int hi = v->hessianIndex();
      isNew = false;
      if (hi >= 0 && (size_t)hi < byHidx.size()) {
        if (byHidx[hi] < 0) {
          byHidx[hi] = count++;
          isNew = true;
        }
        return byHidx[hi];
      }
      std::map<const HyperGraph::Vertex*, int>::iterator it = fixed.find(v);
      if (it == fixed.end()) {
        it = fixed.insert(std::make_pair((const HyperGraph::Vertex*)v, count++)).first;
        isNew = true;
      }
      return it->second;
    }
  };

  // fn(begin, end) over [0, n) on up to _threads host threads: the loops that read or write one estimate per vertex (1.1 M
  // vertices at the metric configuration; setEstimate touches the vertex it is called on only)
  template <class Fn>
  void parallelFor(size_t n, Fn fn) const {
    const size_t nt = (_threads > 1 && n >= 8192) ? (size_t)_threads : 1;
    if (nt == 1) {
      fn((size_t)0, n);
      return;
    }
    _workers.run(n, nt, std::function<void(size_t, size_t)>(fn));
  }

  bool forEachFrontEnd(int (*ba)(g2ohip_solver*), int (*pg)(g2ohip_solver*), const char* what) {
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      const int fast = _groups[gi].fast;
      if (fast && (fast == 1 ? ba(_h) : pg(_h)) != G2OHIP_OK) return fail(what);
    }
    return true;
  }

  // A block matrix of the library (upper triangle, p x p blocks) as the text file the reference's writeOctave / writeCs2Octave produce
  // (csparse_helper.cpp:145-197): "r c value" with one-based indices, both triangles, sorted by column then row.
  bool writeOctave(const std::string& fileName, int which, bool fixed) const {
    int nnzb = 0;
    if (g2ohip_get_nnzb(_h, which, &nnzb) != G2OHIP_OK || nnzb <= 0) return false;
    std::vector<int32_t> colptr(_nP + 1), rowidx(nnzb);
    std::vector<double> val((size_t)nnzb * p * p);
    if (g2ohip_get_pattern(_h, which, colptr.data(), rowidx.data()) != G2OHIP_OK || g2ohip_copy_values(_h, which, val.data()) != G2OHIP_OK) return false;
    struct Entry {
      int r, c;
      double x;
      bool operator<(const Entry& o) const { return c < o.c || (c == o.c && r < o.r); }
    };
    std::vector<Entry> entries;
    for (int c = 0; c < _nP; ++c)
      for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
        const int r = rowidx[q];
        for (int cc = 0; cc < p; ++cc)
          for (int rr = 0; rr < p; ++rr) {
            const double x = val[(size_t)q * p * p + rr + p * cc];   // (column-major blocks)
            if (r == c && rr > cc) continue;                          // (the diagonal blocks hold both halves: once)
            Entry e = {r * p + rr, c * p + cc, x};
            entries.push_back(e);
            if (e.r != e.c) {
              Entry m = {e.c, e.r, x};
              entries.push_back(m);
            }
          }
      }
    std::sort(entries.begin(), entries.end());
    std::string name = fileName;
    const std::string::size_type lastDot = name.find_last_of('.');
    if (lastDot != std::string::npos) name = name.substr(0, lastDot);
    std::ofstream fout(fileName.c_str());
    fout << "# name: " << name << std::endl << "# type: sparse matrix" << std::endl << "# nnz: " << entries.size() << std::endl;
    fout << "# rows: " << _nP * p << std::endl << "# columns: " << _nP * p << std::endl;
    fout << std::setprecision(9);
    if (fixed) fout << std::fixed;
    fout << std::endl;
    for (size_t i = 0; i < entries.size(); ++i) fout << entries[i].r + 1 << " " << entries[i].c + 1 << " " << entries[i].x << std::endl;
    return fout.good();
  }

  static double lap(double& t) {
    const double now = get_monotonic_time(), d = now - t;
    t = now;
    return d;
  }
  // page-locked registration of a buffer that crosses PCIe every iteration; a refusal (the driver's limit on locked memory) is
  // not an error: the copies then go through the driver's staging buffer as before
  void pinDoubles(double* ptr, size_t n) {
    if (!_pin || !ptr || !n) return;
    if (g2ohip_host_register(_h, ptr, n * sizeof(double)) == G2OHIP_OK) _pinned.push_back(ptr);
  }
  void unpinAll() {
    for (size_t i = 0; i < _pinned.size(); ++i) g2ohip_host_unregister(_h, _pinned[i]);
    _pinned.clear();
  }

  bool fail(const char* what) const {
    std::cerr << "BlockSolverHip::" << what << ": " << g2ohip_last_error() << std::endl;
    return false;
  }

  // ---- fast path: EdgeProjectXYZ2UV (types_six_dof_expmap.h:133-153; vertex 0 = VertexSBAPointXYZ, vertex 1 = VertexSE3Expmap).
  // projectEdgeClass: is this edge one, and which (intrinsics, robust kernel) class does it belong to
#if G2OHIP_FASTPATH_SBA
  static bool projectEdgeClass(OptimizableGraph::Edge* e, const GroupKey& key, ClassKey& ck) {
    if (p != 6 || l != 3 || key.d != 2 || key.dim0 != 3 || key.dim1 != 6) return false;
    if (typeid(*e) != typeid(EdgeProjectXYZ2UV)) return false;
    if (typeid(*e->vertex(0)) != typeid(VertexSBAPointXYZ) || typeid(*e->vertex(1)) != typeid(VertexSE3Expmap)) return false;
    const CameraParameters* c = static_cast<const EdgeProjectXYZ2UV*>(e)->_cam;
    if (!c) return false;
    ck.f = c->focal_length;
    ck.cx = c->principle_point[0];
    ck.cy = c->principle_point[1];
    ck.kernel = key.kernel;
    ck.delta = key.kernel ? key.delta : 0.0;
    return true;
  }
  // the merged group is handed to g2ohip_ba_set_edges_classes; false = the front end refused, the caller regroups generically
  // what the binding below needs of the graph (no library call: may run next to g2ohip_build_structure)
  std::vector<int32_t> _baCamOf, _baPointOf;
  std::vector<double> _baMeas, _baInfo;
  bool _baIdentity;
  void gatherProjectXYZ2UV(Group* gp) {
    Group& g = *gp;
    // estimate tables over every vertex the group touches (fixed ones included), in order of first appearance
    _cams.clear();
    _points.clear();
    SlotTable camIndex((size_t)_nP + _nL), pointIndex((size_t)_nP + _nL);
    const size_t n = g.edges.size();
    std::vector<int32_t>&camOf = _baCamOf, &pointOf = _baPointOf;
    std::vector<double>&meas = _baMeas, &info = _baInfo;
    camOf.resize(n);
    pointOf.resize(n);
    meas.resize(2 * n);
    info.resize(4 * n);
    bool identity = true;
    for (size_t k = 0; k < n; ++k) {
      EdgeProjectXYZ2UV* e = static_cast<EdgeProjectXYZ2UV*>(g.edges[k]);
      VertexSBAPointXYZ* vp = static_cast<VertexSBAPointXYZ*>(e->vertex(0));
      VertexSE3Expmap* vc = static_cast<VertexSE3Expmap*>(e->vertex(1));
      bool isNew;
      camOf[k] = camIndex.slot(vc, isNew);
      if (isNew) _cams.push_back(vc);
      pointOf[k] = pointIndex.slot(vp, isNew);
      if (isNew) _points.push_back(vp);
      meas[2 * k] = e->measurement()[0];
      meas[2 * k + 1] = e->measurement()[1];
      const double* om = e->informationData();            // 2 x 2, column-major
      for (int q = 0; q < 4; ++q) info[4 * k + q] = om[q];
      identity = identity && om[0] == 1.0 && om[1] == 0.0 && om[2] == 0.0 && om[3] == 1.0;
    }
    _baIdentity = identity;
  }
  bool bindProjectXYZ2UV(Group& g) {
    if (g.edges.empty()) return false;
    const size_t n = g.edges.size();
    if (_baCamOf.size() != n) gatherProjectXYZ2UV(&g);
    const std::vector<int32_t>&camOf = _baCamOf, &pointOf = _baPointOf;
    const std::vector<double>&meas = _baMeas, &info = _baInfo;
    const bool identity = _baIdentity;
    _camHidx.resize(_cams.size());
    _pointHidx.resize(_points.size());
    {
      std::atomic<int> foreign(0);
      parallelFor(_cams.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
          _camHidx[i] = _cams[i]->hessianIndex();
          if (_camHidx[i] >= _nP) foreign.store(1);       // a marginalized camera: not this front end's layout
        }
      });
      parallelFor(_points.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
          const int hi = _points[i]->hessianIndex();
          if (hi >= 0 && hi < _nP) foreign.store(1);      // a point that is not marginalized: likewise
          _pointHidx[i] = hi < 0 ? -1 : hi - _nP;
        }
      });
      if (foreign.load()) return false;
    }
    const int nClasses = (int)(_baClasses.size() / 5);
    if (g2ohip_ba_set_edges_classes(_h, g.set, camOf.data(), pointOf.data(), meas.data(), identity ? 0 : info.data(), nClasses, _baClasses.data(),
                                    g.cls.data()) != G2OHIP_OK) {
      std::cerr << "BlockSolverHip: fast path not available (" << g2ohip_last_error() << "), using the generic path" << std::endl;
      return false;
    }
    std::vector<int32_t>().swap(_baCamOf);             // (5 M observations: 140 MB of staging the library has copied)
    std::vector<int32_t>().swap(_baPointOf);
    std::vector<double>().swap(_baMeas);
    std::vector<double>().swap(_baInfo);
    _camBuf.assign(12 * _cams.size(), 0.0);
    _pointBuf.assign(3 * _points.size(), 0.0);
    pinDoubles(_camBuf.data(), _camBuf.size());
    pinDoubles(_pointBuf.data(), _pointBuf.size());
    if (std::getenv("G2OHIP_ADAPTER_VERBOSE"))
      std::cerr << "BlockSolverHip: device front end (g2ohip_ba_*) for " << n << " EdgeProjectXYZ2UV, " << _cams.size() << " cameras, " << _points.size()
                << " points, " << nClasses << " edge class" << (nClasses == 1 ? "" : "es") << std::endl;
    return true;
  }
#else
  static bool projectEdgeClass(OptimizableGraph::Edge*, const GroupKey&, ClassKey&) { return false; }
  void gatherProjectXYZ2UV(Group*) {}
  bool bindProjectXYZ2UV(Group&) { return false; }
  bool uploadEstimates() { return false; }
#endif

  // setEstimate of every camera (R column-major | t, world -> camera as VertexSE3Expmap holds it) and point, then the
  // device evaluates errors and Jacobians (computeActiveErrors + linearizeOplus of the group)
  // A homogeneous group of EdgeSE2 over VertexSE2 (types/slam2d/edge_se2.h:41-57, vertex_se2.h:41-59; p = 3): error,
  // Jacobians (edge_se2.cpp:76-99) and the quadratic forms are evaluated on the device from the estimates (x, y, theta) --
  // g2ohip_pg_* with type 1.  The types are recognised by typeid: no member of them is called that is not inline.
#if G2OHIP_FASTPATH_SLAM2D
  bool bindSE2(Group& g) {
    if (p != 3 || g.key.d != 3 || g.key.dim0 != 3 || g.key.dim1 != 3 || g.edges.empty()) return false;
    for (size_t k = 0; k < g.edges.size(); ++k) {
      if (typeid(*g.edges[k]) != typeid(EdgeSE2)) return false;
      if (typeid(*g.edges[k]->vertex(0)) != typeid(VertexSE2) || typeid(*g.edges[k]->vertex(1)) != typeid(VertexSE2)) return false;
    }
    _pgVerts.clear();
    SlotTable index((size_t)_nP + _nL);
    const size_t n = g.edges.size();
    std::vector<int32_t> vi(n), vj(n);
    std::vector<double> meas(3 * n), info(9 * n);
    for (size_t k = 0; k < n; ++k) {
      EdgeSE2* e = static_cast<EdgeSE2*>(g.edges[k]);
      for (int side = 0; side < 2; ++side) {
        VertexSE2* v = static_cast<VertexSE2*>(e->vertex(side));
        bool isNew;
        (side ? vj : vi)[k] = index.slot(v, isNew);
        if (isNew) _pgVerts.push_back(v);
      }
      meas[3 * k] = e->measurement().translation()[0];
      meas[3 * k + 1] = e->measurement().translation()[1];
      meas[3 * k + 2] = e->measurement().rotation().angle();
      const double* om = e->informationData();            // 3 x 3, column-major
      for (int q = 0; q < 9; ++q) info[9 * k + q] = om[q];
    }
    _pgHidx.resize(_pgVerts.size());
    for (size_t i = 0; i < _pgVerts.size(); ++i) _pgHidx[i] = _pgVerts[i]->hessianIndex();
    _pgBuf.assign(3 * _pgVerts.size(), 0.0);
    if (g2ohip_pg_set_edges(_h, g.set, 1, vi.data(), vj.data(), meas.data(), info.data()) != G2OHIP_OK) {
      std::cerr << "BlockSolverHip: fast path not available (" << g2ohip_last_error() << "), using the generic path" << std::endl;
      return false;
    }
    if (std::getenv("G2OHIP_ADAPTER_VERBOSE"))
      std::cerr << "BlockSolverHip: device fast path for " << n << " EdgeSE2 edges over " << _pgVerts.size() << " vertices" << std::endl;
    pinDoubles(_pgBuf.data(), _pgBuf.size());          // (only once the front end has taken the group: the buffer stays)
    return true;
  }
  bool setPosesSE2() {
    for (size_t i = 0; i < _pgVerts.size(); ++i) {
      const SE2& T = _pgVerts[i]->estimate();
      _pgBuf[3 * i] = T.translation()[0];
      _pgBuf[3 * i + 1] = T.translation()[1];
      _pgBuf[3 * i + 2] = T.rotation().angle();
    }
    if (g2ohip_pg_set_estimates(_h, (int)_pgVerts.size(), _pgBuf.data(), _pgHidx.data()) != G2OHIP_OK) return fail("pg_set_estimates");
    return true;
  }
  bool uploadPosesSE2() {
    if (!setPosesSE2()) return false;
    if (g2ohip_pg_linearize(_h, 1) != G2OHIP_OK) return fail("pg_linearize");
    return true;
  }
  bool getPosesSE2() {                                   // device -> setEstimate of the free vertices (fixed ones never change)
    if (g2ohip_pg_get_estimates(_h, _pgBuf.data()) != G2OHIP_OK) return fail("pg_get_estimates");
    for (size_t i = 0; i < _pgVerts.size(); ++i)
      if (_pgHidx[i] >= 0) _pgVerts[i]->setEstimate(SE2(_pgBuf[3 * i], _pgBuf[3 * i + 1], _pgBuf[3 * i + 2]));
    return true;
  }
#else
  bool bindSE2(Group&) { return false; }
  bool uploadPosesSE2() { return false; }
  bool setPosesSE2() { return false; }
  bool getPosesSE2() { return false; }
#endif

#if G2OHIP_FASTPATH_SLAM3D
  // A homogeneous group of EdgeSE3 over VertexSE3 (types/slam3d/edge_se3.cpp:48-75, vertex_se3.h:107-116; p = 6): g2ohip_pg_*
  // with type 2, estimates and measurements as isometries [12] = R (column-major) | t.
  static void isometryTo12(const Eigen::Isometry3d& T, double* c) {
    for (int col = 0; col < 3; ++col)
      for (int row = 0; row < 3; ++row) c[row + 3 * col] = T.linear()(row, col);
    for (int row = 0; row < 3; ++row) c[9 + row] = T.translation()[row];
  }
  bool bindSE3(Group& g) {
    if (p != 6 || g.key.d != 6 || g.key.dim0 != 6 || g.key.dim1 != 6 || g.edges.empty()) return false;
    for (size_t k = 0; k < g.edges.size(); ++k) {
      if (typeid(*g.edges[k]) != typeid(EdgeSE3)) return false;
      if (typeid(*g.edges[k]->vertex(0)) != typeid(VertexSE3) || typeid(*g.edges[k]->vertex(1)) != typeid(VertexSE3)) return false;
    }
    _pg3Verts.clear();
    SlotTable index((size_t)_nP + _nL);
    const size_t n = g.edges.size();
    std::vector<int32_t> vi(n), vj(n);
    std::vector<double> meas(12 * n), info(36 * n);
    for (size_t k = 0; k < n; ++k) {
      EdgeSE3* e = static_cast<EdgeSE3*>(g.edges[k]);
      for (int side = 0; side < 2; ++side) {
        VertexSE3* v = static_cast<VertexSE3*>(e->vertex(side));
        bool isNew;
        (side ? vj : vi)[k] = index.slot(v, isNew);
        if (isNew) _pg3Verts.push_back(v);
      }
      isometryTo12(e->measurement(), &meas[12 * k]);
      const double* om = e->informationData();            // 6 x 6, column-major
      for (int q = 0; q < 36; ++q) info[36 * k + q] = om[q];
    }
    _pgHidx.resize(_pg3Verts.size());
    for (size_t i = 0; i < _pg3Verts.size(); ++i) _pgHidx[i] = _pg3Verts[i]->hessianIndex();
    _pgBuf.assign(12 * _pg3Verts.size(), 0.0);
    if (g2ohip_pg_set_edges(_h, g.set, 2, vi.data(), vj.data(), meas.data(), info.data()) != G2OHIP_OK) {
      std::cerr << "BlockSolverHip: fast path not available (" << g2ohip_last_error() << "), using the generic path" << std::endl;
      return false;
    }
    if (std::getenv("G2OHIP_ADAPTER_VERBOSE"))
      std::cerr << "BlockSolverHip: device fast path for " << n << " EdgeSE3 edges over " << _pg3Verts.size() << " vertices" << std::endl;
    pinDoubles(_pgBuf.data(), _pgBuf.size());
    return true;
  }
  bool setPosesSE3() {
    for (size_t i = 0; i < _pg3Verts.size(); ++i) isometryTo12(_pg3Verts[i]->estimate(), &_pgBuf[12 * i]);
    if (g2ohip_pg_set_estimates(_h, (int)_pg3Verts.size(), _pgBuf.data(), _pgHidx.data()) != G2OHIP_OK) return fail("pg_set_estimates");
    return true;
  }
  bool uploadPosesSE3() {
    if (!setPosesSE3()) return false;
    if (g2ohip_pg_linearize(_h, 1) != G2OHIP_OK) return fail("pg_linearize");
    return true;
  }
  bool getPosesSE3() {
    if (g2ohip_pg_get_estimates(_h, _pgBuf.data()) != G2OHIP_OK) return fail("pg_get_estimates");
    for (size_t i = 0; i < _pg3Verts.size(); ++i) {
      if (_pgHidx[i] < 0) continue;
      Eigen::Isometry3d T = _pg3Verts[i]->estimate();   // (keeps whatever the type holds besides R | t)
      const double* c = &_pgBuf[12 * i];
      for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) T.linear()(row, col) = c[row + 3 * col];
      for (int row = 0; row < 3; ++row) T.translation()[row] = c[9 + row];
      _pg3Verts[i]->setEstimate(T);
    }
    return true;
  }
#else
  bool bindSE3(Group&) { return false; }
  bool uploadPosesSE3() { return false; }
  bool setPosesSE3() { return false; }
  bool getPosesSE3() { return false; }
#endif

#if G2OHIP_FASTPATH_SBA
  bool setEstimatesBA() {
    parallelFor(_cams.size(), [this](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        const SE3Quat& T = _cams[i]->estimate();
        const Eigen::Matrix3d R = T.rotation().toRotationMatrix();
        double* c = &_camBuf[12 * i];
        for (int col = 0; col < 3; ++col)
          for (int row = 0; row < 3; ++row) c[row + 3 * col] = R(row, col);
        for (int row = 0; row < 3; ++row) c[9 + row] = T.translation()[row];
      }
    });
    parallelFor(_points.size(), [this](size_t b, size_t e) {
      for (size_t i = b; i < e; ++i)
        for (int row = 0; row < 3; ++row) _pointBuf[3 * i + row] = _points[i]->estimate()[row];
    });
    if (g2ohip_ba_set_estimates(_h, (int)_cams.size(), _camBuf.data(), _camHidx.data(), (int)_points.size(), _pointBuf.data(), _pointHidx.data()) != G2OHIP_OK)
      return fail("ba_set_estimates");
    return true;
  }
  bool uploadEstimates() {
    if (!setEstimatesBA()) return false;
    if (g2ohip_ba_linearize(_h, 1) != G2OHIP_OK) return fail("ba_linearize");
    return true;
  }
  bool getEstimatesBA() {                                // device -> setEstimate of the free cameras and points
    if (g2ohip_ba_get_estimates(_h, _camBuf.data(), _pointBuf.data()) != G2OHIP_OK) return fail("ba_get_estimates");
    scatterCams();
    scatterPoints(0, _points.size());
    return true;
  }
  void scatterCams() {
    parallelFor(_cams.size(), [this](size_t b, size_t e) { scatterCamsRange(b, e); });
  }
  void scatterPoints(size_t first, size_t last) {
    parallelFor(last - first, [this, first](size_t b, size_t e) { scatterPointsRange(first + b, first + e); });
  }
  void scatterCamsRange(size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      if (_camHidx[i] < 0) continue;
      const double* c = &_camBuf[12 * i];
      Eigen::Matrix3d R;
      Eigen::Vector3d t;
      for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) R(row, col) = c[row + 3 * col];
      for (int row = 0; row < 3; ++row) t[row] = c[9 + row];
      _cams[i]->setEstimate(SE3Quat(R, t));           // (se3quat.h:58-60: quaternion of R, normalised)
    }
  }
  void scatterPointsRange(size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      // (a million vertex objects scattered over the heap: every store is a cache miss; the misses of the next vertices
      // are started here instead of one at a time)
      if (i + 12 < e) __builtin_prefetch(_points[i + 12]->estimate().data(), 1, 1);
      if (_pointHidx[i] < 0) continue;
      Eigen::Vector3d x;
      for (int row = 0; row < 3; ++row) x[row] = _pointBuf[3 * i + row];
      _points[i]->setEstimate(x);
    }
  }
  // Hybrid loop: may the device-resident drivers take a graph that has host-linearised groups next to the bundle-adjustment
  // front end?  Yes when every FREE vertex those groups touch is a camera or a point of the front end (the device updates and
  // stacks it; a vertex no front end knows would miss its oplus) and no pose-graph front end is bound next to it.
  // G2OHIP_ADAPTER_HYBRID=0: never (such graphs then run g2o's host loop, as before).
  void planHybrid() {
    _hybrid = false;
    _touchedPushed = false;
    _touched.clear();
    const char* hy = std::getenv("G2OHIP_ADAPTER_HYBRID");
    if (hy && hy[0] == '0') return;
    bool haveBA = false, other = !_multi.empty();
    for (size_t gi = 0; gi < _groups.size(); ++gi) {
      if (_groups[gi].fast == 1) haveBA = true;
      else if (_groups[gi].fast == 0) other = true;
      else return;
    }
    if (!haveBA || !other) return;
    std::vector<int> camOf(_nP > 0 ? _nP : 1, -1), ptOf(_nL > 0 ? _nL : 1, -1);
    for (size_t i = 0; i < _cams.size(); ++i)
      if (_camHidx[i] >= 0 && _camHidx[i] < _nP) camOf[_camHidx[i]] = (int)i;
    for (size_t i = 0; i < _points.size(); ++i)
      if (_pointHidx[i] >= 0 && _pointHidx[i] < _nL) ptOf[_pointHidx[i]] = (int)i;
    std::map<OptimizableGraph::Vertex*, Touched> seen;
    std::vector<OptimizableGraph::Edge*> es;
    for (size_t gi = 0; gi < _groups.size(); ++gi)
      if (!_groups[gi].fast) es.insert(es.end(), _groups[gi].edges.begin(), _groups[gi].edges.end());
    for (size_t m = 0; m < _multi.size(); ++m) es.insert(es.end(), _multi[m].edges.begin(), _multi[m].edges.end());
    for (size_t k = 0; k < es.size(); ++k)
      for (size_t i = 0; i < es[k]->vertices().size(); ++i) {
        OptimizableGraph::Vertex* v = static_cast<OptimizableGraph::Vertex*>(es[k]->vertex(i));
        const int h = v->hessianIndex();
        if (h < 0 || seen.count(v)) continue;          // (a fixed vertex keeps its estimate)
        Touched t;
        t.v = v;
        t.kind = v->marginalized() ? 1 : 0;
        t.idx = t.kind ? (h - _nP >= 0 && h - _nP < _nL ? ptOf[h - _nP] : -1) : (h < _nP ? camOf[h] : -1);
        if (t.idx < 0) return;                         // a vertex the front end does not hold
        if (t.kind ? static_cast<OptimizableGraph::Vertex*>(_points[t.idx]) != v : static_cast<OptimizableGraph::Vertex*>(_cams[t.idx]) != v) return;
        seen[v] = t;
      }
    for (typename std::map<OptimizableGraph::Vertex*, Touched>::const_iterator it = seen.begin(); it != seen.end(); ++it) _touched.push_back(it->second);
    _touchedCams.clear();
    _touchedPoints.clear();
    for (size_t i = 0; i < _touched.size(); ++i) (_touched[i].kind ? _touchedPoints : _touchedCams).push_back(_touched[i].idx);
    _touchedCamEst.assign(12 * _touchedCams.size() + 1, 0.);
    _touchedPointEst.assign(3 * _touchedPoints.size() + 1, 0.);
    _hybrid = true;
    if (std::getenv("G2OHIP_ADAPTER_VERBOSE"))
      std::cerr << "BlockSolverHip: hybrid device loop -- " << es.size() << " host-linearised edges over " << _touched.size() << " free vertices of the device front end" << std::endl;
  }
  // the trial estimates of the touched vertices from the device (the read-back started behind the trial's update, or a plain one)
  void setTouchedFromCache() {
    size_t ic = 0, ip = 0;
    for (size_t i = 0; i < _touched.size(); ++i) {
      if (_touched[i].kind) {
        const double* xs = &_touchedPointEst[3 * ip++];
        Eigen::Vector3d x;
        for (int row = 0; row < 3; ++row) x[row] = xs[row];
        _points[_touched[i].idx]->setEstimate(x);
      } else {
        const double* c = &_touchedCamEst[12 * ic++];
        Eigen::Matrix3d R;
        Eigen::Vector3d t;
        for (int col = 0; col < 3; ++col)
          for (int row = 0; row < 3; ++row) R(row, col) = c[row + 3 * col];
        for (int row = 0; row < 3; ++row) t[row] = c[9 + row];
        _cams[_touched[i].idx]->setEstimate(SE3Quat(R, t));
      }
    }
  }
  // computeError of the host-linearised edges at what the vertices hold now (their _error members only: nothing goes to the device)
  void hostErrorsOnly() {
    for (size_t gi = 0; gi < _groups.size(); ++gi)
      if (!_groups[gi].fast)
        for (size_t k = 0; k < _groups[gi].edges.size(); ++k) _groups[gi].edges[k]->computeError();
    for (size_t m = 0; m < _multi.size(); ++m)
      for (size_t k = 0; k < _multi[m].edges.size(); ++k) _multi[m].edges[k]->computeError();
  }
  void touchedBackToAccepted() {
    if (_touchedPushed)
      for (size_t i = 0; i < _touched.size(); ++i) _touched[i].v->pop();
    _touchedPushed = false;
    _touchedSpec = true;
    hostErrorsOnly();
  }
  // (only THEIR estimates cross PCIe here -- g2ohip_ba_get_estimates_of, a gather and a copy of a few hundred bytes --: the full
  // read-back of the trial, 34 MB at the metric configuration, keeps running next to the host's error evaluation instead of
  // being waited for in front of it)
  bool refreshTouched() {
    if (_touched.empty()) return true;
    if (g2ohip_ba_get_estimates_of(_h, (int)_touchedCams.size(), _touchedCams.data(), _touchedCamEst.data(), (int)_touchedPoints.size(),
                                   _touchedPoints.data(), _touchedPointEst.data()) != G2OHIP_OK)
      return fail("ba_get_estimates_of");
    for (size_t i = 0; i < _touched.size(); ++i) _touched[i].v->push();
    setTouchedFromCache();
    _touchedPushed = true;
    return true;
  }
#else
  bool setEstimatesBA() { return false; }
  bool getEstimatesBA() { return false; }
  void planHybrid() { _hybrid = false; }
  bool refreshTouched() { return true; }
  void setTouchedFromCache() {}
  void hostErrorsOnly() {}
  void touchedBackToAccepted() {}
#endif

  g2ohip_solver* _h;
  bool _doSchur, _writeDebug;
  int _nP, _nL;
  std::vector<Group> _groups;
  std::vector<MultiGroup> _multi;                      // n-ary edge groups (generic path)
  std::vector<double> _diagMirror, _diag;
  bool _fastPath;
  int _fastGroups;                                     // groups bound to a device front end (Group::fast)
  bool _devValid;                                      // the front ends hold estimates for the current structure
  bool _pin, _timing;
  int _threads;
  // hybrid loop (deviceResident with host-linearised groups): the free vertices those groups touch, as (vertex, camera / point, index)
  struct Touched {
    OptimizableGraph::Vertex* v;
    int kind, idx;                                     // 0: _cams[idx], 1: _points[idx]
  };
  std::vector<Touched> _touched;
  std::vector<int32_t> _touchedCams, _touchedPoints;   // their indices in _cams / _points, in the order of _touched
  std::vector<double> _touchedCamEst, _touchedPointEst;
  bool _hybrid, _touchedPushed, _fetchBegun;
  bool _touchedSpec = false;                           // the cache holds the touched vertices' estimates of a look-ahead trial the vertices have not seen
  int _lookDropped = 0, _lookQueued = 0;
  bool _lookPending, _lookEnabled, _queueing;          // look-ahead trial in flight | allowed (G2OHIP_ADAPTER_LOOKAHEAD) | being queued (no timing synchronisation)
  bool _lambdaInForce;                                 // between setLambda and restoreDiagonal (a Levenberg-Marquardt trial): no undamped retry
  int _undampedRetries;                                // undamped solves repeated with lambda = 1e-14 x max diag (solveUndampedRetry)
  enum { kFetchPieces = 4 };                           // point ranges of the pipelined write-back (devFetchBegin / devFetchEnd)
  mutable HipWorkers _workers;                         // persistent helper threads of parallelFor
  std::vector<void*> _pinned;                          // buffers registered with g2ohip_host_register
  Phases _phase;
  std::vector<double> _baClasses;                      // class table of the BA group: (f, cx, cy, kernel kind, delta) per class
#if G2OHIP_FASTPATH_SLAM2D
  std::vector<VertexSE2*> _pgVerts;
#endif
#if G2OHIP_FASTPATH_SLAM3D
  std::vector<VertexSE3*> _pg3Verts;
#endif
  std::vector<int32_t> _pgHidx;
  std::vector<double> _pgBuf;
#if G2OHIP_FASTPATH_SBA
  std::vector<VertexSE3Expmap*> _cams;
  std::vector<VertexSBAPointXYZ*> _points;
#endif
  std::vector<int32_t> _camHidx, _pointHidx;
  std::vector<double> _camBuf, _pointBuf;
};


// ---------------------------------------------------------------------------------------------------------
// Narrow seam: g2o's own CPU assembly and Schur complement stay, only the sparse Cholesky solve moves to the device
// (LinearSolverCSparse::solve, solvers/csparse/linear_solver_csparse.h:106-142).
// ---------------------------------------------------------------------------------------------------------
template <typename MatrixType>
class LinearSolverHip : public LinearSolver<MatrixType> {
 public:
  explicit LinearSolverHip(int blockDim = MatrixType::RowsAtCompileTime, int device = 0) : _ls(0) {
    if (g2ohip_ls_create(&_ls, blockDim, device) != G2OHIP_OK) {
      std::cerr << "LinearSolverHip: " << g2ohip_last_error() << std::endl;
      _ls = 0;
    }
  }
  virtual ~LinearSolverHip() { if (_ls) g2ohip_ls_destroy(_ls); }

  // drop the symbolic factorisation: the pattern may change until the next init() (linear_solver.h:50)
  virtual bool init() { return _ls && g2ohip_ls_init(_ls) == G2OHIP_OK; }

  // A symmetric, upper blocks only; x, b caller-allocated of length A.rows(); false iff not positive definite
  virtual bool solve(const SparseBlockMatrix<MatrixType>& A, double* x, double* b) {
    if (!_ls) return false;
    const int nb = (int)A.blockCols().size();
    exportUpper(A);
    const int rc = g2ohip_ls_solve(_ls, nb, _colptr.data(), _rowidx.data(), _values.data(), x, b);
    G2OBatchStatistics* gs = G2OBatchStatistics::globalStats();
    if (gs) {
      g2ohip_stats st;
      if (g2ohip_ls_get_stats(_ls, &st) == G2OHIP_OK) {   // the numeric factorisation alone, as LinearSolverCSparse times it
        gs->timeSymbolicDecomposition = st.timeSymbolicDecomposition;   // (linear_solver_csparse.h:122-139)
        gs->timeNumericDecomposition = st.timeNumericDecomposition;
        gs->choleskyNNZ = st.choleskyNNZ;
      }
    }
    if (rc != G2OHIP_OK && rc != G2OHIP_NOT_PD) std::cerr << "LinearSolverHip::solve: " << g2ohip_last_error() << std::endl;
    return rc == G2OHIP_OK;
  }

  // the diagonal blocks of A^-1 (linear_solver.h:64; LinearSolverCSparse::solveBlocks, linear_solver_csparse.h:144-187):
  // allocated here when the caller passes a null pointer, one bd x bd array per block row (symmetric: layout-free)
  virtual bool solveBlocks(double**& blocks, const SparseBlockMatrix<MatrixType>& A) {
    if (!_ls) return false;
    const int nb = (int)A.blockCols().size();
    if (nb == 0) return true;
    exportUpper(A);
    const int bd = A.rows() / nb;
    std::vector<int32_t> d(nb);
    for (int i = 0; i < nb; ++i) d[i] = i;
    std::vector<double> out((size_t)nb * bd * bd);
    const int rc = g2ohip_ls_solve_pattern(_ls, nb, _colptr.data(), _rowidx.data(), _values.data(), nb, d.data(), d.data(), out.data());
    if (rc != G2OHIP_OK) {
      if (rc != G2OHIP_NOT_PD) std::cerr << "LinearSolverHip::solveBlocks: " << g2ohip_last_error() << std::endl;
      return false;
    }
    if (!blocks) {
      blocks = new double*[A.rows()];                   // (the reference sizes the pointer array by rows, :160)
      for (int i = 0; i < nb; ++i) blocks[i] = new double[bd * bd];
    }
    for (int i = 0; i < nb; ++i) std::memcpy(blocks[i], &out[(size_t)i * bd * bd], sizeof(double) * bd * bd);
    return true;
  }

  // blocks of A^-1 (linear_solver.h:71; LinearSolverCSparse::solvePattern, linear_solver_csparse.h:190-221): factorise,
  // then every requested block on the pattern of the factor from one sparse-inverse pass
  virtual bool solvePattern(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices,
                            const SparseBlockMatrix<MatrixType>& A) {
    if (!_ls) return false;
    const int nb = (int)A.blockCols().size();
    exportUpper(A);
    const int bd = nb > 0 ? A.rows() / nb : 0;
    std::vector<int32_t> r(blockIndices.size()), c(blockIndices.size());
    for (size_t i = 0; i < blockIndices.size(); ++i) {
      r[i] = blockIndices[i].first;
      c[i] = blockIndices[i].second;
    }
    std::vector<double> out(blockIndices.size() * bd * bd);
    const int rc = g2ohip_ls_solve_pattern(_ls, nb, _colptr.data(), _rowidx.data(), _values.data(), (int)blockIndices.size(), r.data(), c.data(),
                                           out.data());
    if (rc != G2OHIP_OK) {
      if (rc != G2OHIP_NOT_PD) std::cerr << "LinearSolverHip::solvePattern: " << g2ohip_last_error() << std::endl;
      return false;
    }
    spinv = SparseBlockMatrix<MatrixXd>(&A.rowBlockIndices()[0], &A.rowBlockIndices()[0], nb, nb, true);   // marginal_covariance_cholesky.cpp:156-160
    for (size_t i = 0; i < blockIndices.size(); ++i) {
      MatrixXd* blk = spinv.block(r[i], c[i], true);
      std::memcpy(blk->data(), &out[i * bd * bd], sizeof(double) * bd * bd);
    }
    return true;
  }

 private:
  // block-CCS export of the upper triangle: column pointers, row block indices, dense blocks
  void exportUpper(const SparseBlockMatrix<MatrixType>& A) {
    const int nb = (int)A.blockCols().size();
    _colptr.assign(1, 0);
    _rowidx.clear();
    _values.clear();
    for (int c = 0; c < nb; ++c) {
      const typename SparseBlockMatrix<MatrixType>::IntBlockMap& column = A.blockCols()[c];
      for (typename SparseBlockMatrix<MatrixType>::IntBlockMap::const_iterator it = column.begin(); it != column.end(); ++it) {
        if (it->first > c) break;
        const MatrixType* blk = it->second;
        _rowidx.push_back(it->first);
        _values.insert(_values.end(), blk->data(), blk->data() + blk->rows() * blk->cols());
      }
      _colptr.push_back((int32_t)_rowidx.size());
    }
  }
  g2ohip_linear_solver* _ls;
  std::vector<int32_t> _colptr, _rowidx;
  std::vector<double> _values;
};

}  // namespace g2o

#endif
