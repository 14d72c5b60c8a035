// Device-resident drivers for the g2o plugin: OptimizationAlgorithmLevenberg / GaussNewton whose solve() keeps the WHOLE
// iteration on the MI355X when every active edge sits on one of BlockSolverHip's device front ends (EdgeProjectXYZ2UV bundle
// adjustment, EdgeSE2 / EdgeSE3 pose graphs), and is g2o's own host loop otherwise.
//
// Why: behind g2o's host loop the device works 2 ms of a 277 ms Levenberg-Marquardt iteration at the metric configuration
// (INTEGRATION.md): computeActiveErrors twice, activeRobustChi2 twice, update, push / pop and the estimate gather walk 5 M
// edges and 1.1 M vertices on one CPU core around every solve.  Those are members of SparseOptimizer, not of the Solver seam;
// the seam ONE level up -- OptimizationAlgorithm::solve (optimization_algorithm.h:46-110), found through the same factory --
// is where a plugin may replace them.  Here the estimates go up once per optimize() (iteration 0), every trial runs on the
// device (g2ohip_ba_* / g2ohip_pg_*: errors, chi2, oplus, the estimate stack; one host synchronisation per trial:
// g2ohip_solve_async / g2ohip_trial_stats) and the accepted estimates come back into the vertices at the end of each
// solve(), so that everything g2o or its user does between iterations (verbose output, actions, computeActiveErrors,
// save) sees what it would have seen.  The decisions are those of optimization_algorithm_levenberg.cpp:57-146 and
// optimization_algorithm_gauss_newton.cpp:50-93, taken on the same numbers.
//   G2OHIP_ADAPTER_DEVICE_LOOP=0   always g2o's host loop (A/B)
//   G2OHIP_ADAPTER_LOOKAHEAD=0     the LM driver does not queue the next iteration's first trial before the write-back (A/B; see
//                                  lookAhead below: same numbers, the device idles during the write-back)
//   G2OHIP_ADAPTER_WRITEBACK=0     estimates are NOT written back after every iteration (timing experiments only: the
//                                  vertices then keep their initial estimates)
// Not mirrored from the host loop: the edges' _error members are not refreshed (computeActiveErrors() does that on demand)
// and the vertices' own backup stacks are not touched (push / pop happen on the device).
#ifndef G2O_HIP_ALGORITHM_H
#define G2O_HIP_ALGORITHM_H

#include <cstdlib>
#include <iostream>
#include <limits>

#include "g2o/core/batch_stats.h"
#include "g2o/core/optimization_algorithm_gauss_newton.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/sparse_optimizer.h"
#include "g2o/stuff/timeutil.h"
#include "g2o_hip_solver.h"

namespace g2o {

namespace hip_detail {
inline bool envOff(const char* name) {
  const char* v = std::getenv(name);
  return v && v[0] == '0';
}
// iteration 0 of a device-resident run: is the structure built, and may the device loop take the graph?
inline bool canDeviceLoop(Solver* solver, HipDeviceGraph* dev) { return solver && dev && dev->deviceResident() && !envOff("G2OHIP_ADAPTER_DEVICE_LOOP"); }
// Decided where the structure may have changed (iteration 0, online); in between only taken back: a structure rebuilt under the
// algorithm (updateInitialization followed by solve(iteration > 0)) that is no longer all on the device must not stay on the loop
inline bool decideDeviceLoop(const char* who, Solver* solver, HipDeviceGraph* dev, bool resident, bool decide) {
  const bool can = canDeviceLoop(solver, dev);
  if (!decide) return resident && can;
  if (std::getenv("G2OHIP_ADAPTER_VERBOSE"))
    std::cerr << who << ": " << (can ? "device-resident iteration (errors, chi2, oplus and the estimate stack on the GPU)"
                                     : "g2o's host loop (not every active edge is on a device front end, or G2OHIP_ADAPTER_DEVICE_LOOP=0)")
              << std::endl;
  return can;
}
}  // namespace hip_detail

class OptimizationAlgorithmLevenbergHip : public OptimizationAlgorithmLevenberg {
 public:
  explicit OptimizationAlgorithmLevenbergHip(Solver* solver)
      : OptimizationAlgorithmLevenberg(solver), _dev(dynamic_cast<HipDeviceGraph*>(solver)), _resident(false),
        _writeBack(!hip_detail::envOff("G2OHIP_ADAPTER_WRITEBACK")), _fetched(false), _accepted(false), _lookChi(0.) {}

  //! did the last solve() run on the device?
  bool deviceLoopActive() const { return _resident; }

  virtual SolverResult solve(int iteration, bool online = false) {
    if (iteration == 0 && !online) {                     // levenberg.cpp:62-68
      if (!_solver->buildStructure()) {
        std::cerr << "OptimizationAlgorithmLevenbergHip::solve: Failure while building CCS structure" << std::endl;
        return OptimizationAlgorithm::Fail;
      }
    }
    _resident = hip_detail::decideDeviceLoop("OptimizationAlgorithmLevenbergHip", _solver, _dev, _resident, iteration == 0 || online);
    // (`online` only guards buildStructure in the host loop, levenberg.cpp:62: the structure exists by now)
    if (!_resident) return OptimizationAlgorithmLevenberg::solve(iteration, true);

    double t = get_monotonic_time();
    G2OBatchStatistics* globalStats = G2OBatchStatistics::globalStats();
    // The previous solve() of this optimize() may have queued this iteration's head (see lookAhead): errors, chi2, buildSystem
    // and the first trial at _currentLambda are in flight or done.  Anything but the next iteration of the same run drops it.
    const bool consume = _dev->devLookAheadPending() && iteration > 0 && !online && _dev->devEstimatesValid();
    if (!consume) _dev->devDropLookAhead();
    double currentChi = 0.;
    if (consume) {
      _dev->devSetLookAheadPending(false);              // (from here on the solver's entry points are this driver's own calls)
      currentChi = _lookChi;
    } else {
      // a new optimize() (the caller may have changed the vertices), online growth, or a structure rebuilt under us
      if (iteration == 0 || online || !_dev->devEstimatesValid()) {
        if (!_dev->devSetEstimates()) return OptimizationAlgorithm::Fail;
      }
      if (!_dev->devLinearize(true)) return OptimizationAlgorithm::Fail;       // computeActiveErrors + what buildSystem linearises
      if (!_dev->devChi2(currentChi)) return OptimizationAlgorithm::Fail;      // activeRobustChi2
      if (globalStats) {
        globalStats->timeResiduals = get_monotonic_time() - t;
        t = get_monotonic_time();
      }
      if (!_dev->devBuildSystem()) return OptimizationAlgorithm::Fail;
      if (globalStats) globalStats->timeQuadraticForm = get_monotonic_time() - t;

      if (iteration == 0) {                                // computeLambdaInit, levenberg.cpp:149-163
        double maxDiagonal = 0.;
        if (userLambdaInit() > 0) {
          _currentLambda = userLambdaInit();
        } else {
          if (!_dev->devMaxDiagonal(maxDiagonal)) return OptimizationAlgorithm::Fail;
          _currentLambda = _tau * maxDiagonal;
        }
        _ni = 2;
      }
      _fetched = false;
    }
    double tempChi = currentChi;

    double rho = 0;
    int& qmax = _levenbergIterations;
    qmax = 0;
    _accepted = false;
    bool queued = consume;                               // the first trial is already on the device
    do {
      double scale = 0.;
      int ok2 = -1;
      if (globalStats) {
        globalStats->levenbergIterations++;
        t = get_monotonic_time();
      }
      if (queued) {
        queued = false;
        ok2 = _dev->devTrialStats(_currentLambda, tempChi, scale);
      } else {
        if (_fetched) {                                    // (the read-back of a rejected trial: its buffers are about to be reused)
          _dev->devFetchCancel();
          _fetched = false;
        }
        if (!_dev->devPush()) return OptimizationAlgorithm::Fail;
        // solve, update, restoreDiagonal, computeActiveErrors queued back to back; status, chi2 and computeScale in ONE read-back
        if (queueTrial(/*fetch=*/true)) ok2 = _dev->devTrialStats(_currentLambda, tempChi, scale);
      }
      if (ok2 == 2) {                                    // (a dependency-driven launch gave up waiting: the trial again, synchronously)
        if (_fetched) _dev->devFetchCancel();
        _fetched = false;
        if (!_dev->devPop() || !_dev->devPush()) return OptimizationAlgorithm::Fail;
        _solver->setLambda(_currentLambda, true);
        ok2 = _dev->devSolve();
        if (ok2 >= 0 && _dev->devUpdate()) {
          _solver->restoreDiagonal();
          if (!_dev->devLinearize(false) || !_dev->devChi2(tempChi)) ok2 = -1;
          else if (ok2 == 1 && !_dev->devComputeScale(_currentLambda, scale)) ok2 = -1;
        } else {
          ok2 = -1;
        }
      }
      if (ok2 < 0) {
        _dev->devPop();
        finish(globalStats, false, currentChi);
        return OptimizationAlgorithm::Fail;
      }
      if (globalStats) globalStats->timeLinearSolution += get_monotonic_time() - t;

      if (!ok2) {
        tempChi = std::numeric_limits<double>::max();
        scale = 0.;
      }
      rho = (currentChi - tempChi);
      scale += 1e-3;                                     // make sure it's non-zero :)   (levenberg.cpp:121)
      rho /= scale;

      if (rho > 0 && tempChi - tempChi == 0.) {          // last step was good (and chi2 finite)
        double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
        alpha = (alpha < _goodStepUpperScale) ? alpha : _goodStepUpperScale;
        const double scaleFactor = (_goodStepLowerScale > alpha) ? _goodStepLowerScale : alpha;
        _currentLambda *= scaleFactor;
        _ni = 2;
        currentChi = tempChi;
        _accepted = true;
        if (!_dev->devDiscardTop()) return OptimizationAlgorithm::Fail;
      } else {
        _currentLambda *= _ni;
        _ni *= 2;
        if (!_dev->devPop()) return OptimizationAlgorithm::Fail;   // restore the last state before trying to optimize
      }
      qmax++;
    } while (rho < 0 && qmax < maxTrialsAfterFailure() && !_optimizer->terminate());

    const bool goesOn = !(qmax == maxTrialsAfterFailure() || rho == 0);
    // (iteration 0 never looks ahead: a caller that runs optimize(1) in a loop would pay for a trial it always drops)
    if (!finish(globalStats, goesOn && iteration > 0 && !online, currentChi)) return OptimizationAlgorithm::Fail;
    if (!goesOn) return Terminate;
    return OK;
  }

 private:
  // One trial behind devPush(), queued without a synchronisation: setLambda, solve, update, (the read-back of the trial's
  // estimates starts next to its error evaluation), restoreDiagonal, computeActiveErrors.  levenberg.cpp:98-117.
  bool queueTrialSolve(bool fetch) {
    _solver->setLambda(_currentLambda, true);
    if (!_dev->devSolveAsync() || !_dev->devUpdate()) return false;
    if (fetch && _writeBack) _fetched = _dev->devFetchBegin();
    _solver->restoreDiagonal();
    return true;
  }
  bool queueTrial(bool fetch) { return queueTrialSolve(fetch) && _dev->devLinearize(false); }
  // The head of the NEXT solve(), queued before the accepted estimates are written into the vertices: the same calls in the same
  // order as solve() makes them (devLinearize / devChi2 find the trial's evaluation still valid: no kernel, no synchronisation),
  // so the numbers are those of the run without look-ahead.  The read-back of THIS trial's estimates has to wait until the
  // write-back has emptied the host buffers: begun by finish().  Two halves: up to the trial's update, and its error evaluation
  // with the sums -- the hybrid loop's host edges evaluate their errors on the host, behind a synchronisation: there the
  // write-back goes in between.
  bool lookAheadSolve(double& chi) {
    _dev->devSetQueueing(true);
    const bool ok = _dev->devLinearize(true) && _dev->devChi2(chi) && _dev->devBuildSystem() && _dev->devPush() && queueTrialSolve(/*fetch=*/false);
    _dev->devSetQueueing(false);
    return ok;
  }
  bool lookAheadErrors() {
    _dev->devSetQueueing(true);                          // (hybrid loop: the host edges' trial errors must leave no trace in the vertices)
    const bool ok = _dev->devLinearize(false) && _dev->devTrialStatsBegin(_currentLambda);
    _dev->devSetQueueing(false);
    return ok;
  }
  // the accepted estimates into the vertices (what SparseOptimizer::update / pop left there in the host loop)
  // (the loop ends with an accepted trial or with the estimates popped back to what the vertices already hold: only an accepted
  // trial has anything to write; its read-back has been in flight since its update)
  bool finish(G2OBatchStatistics* globalStats, bool mayLookAhead, double currentChi) {
    const double t = get_monotonic_time();
    bool ok = true;
    if (_fetched && !_accepted) {
      _dev->devFetchCancel();
    } else if (_writeBack && _accepted) {
      const bool ahead = mayLookAhead && _fetched && _dev->devCanLookAhead();
      const bool hybrid = ahead && _dev->devHybrid();
      if (ahead) {
        double chi = 0.;
        if (!lookAheadSolve(chi) || (!hybrid && !lookAheadErrors())) {
          (void)_dev->devPop();                          // (a device error half-way: the estimate stack must not keep the trial's level)
          if (_fetched) _dev->devFetchCancel();          // ... nor the accepted trial's read-back stay in flight
          _fetched = false;
          return false;
        }
        _lookChi = chi;
        (void)currentChi;                                // (== chi: the accepted trial's sum, cached by the library)
      }
      ok = _fetched ? _dev->devFetchEnd() : _dev->devGetEstimates();
      _fetched = false;
      if (ahead) {
        if (hybrid && !lookAheadErrors()) {
          (void)_dev->devPop();
          _fetched = false;
          return false;
        }
        _fetched = _dev->devFetchBegin();                // the queued trial's estimates, behind its update
        _dev->devSetLookAheadPending(true);
      }
      if (globalStats) globalStats->timeUpdate = get_monotonic_time() - t;
      return ok;
    }
    _fetched = false;
    if (globalStats) globalStats->timeUpdate = get_monotonic_time() - t;
    return ok;
  }
  HipDeviceGraph* _dev;
  bool _resident, _writeBack;
  bool _fetched, _accepted;   // the current trial's estimates are on their way to the host; a trial of this iteration was accepted
  double _lookChi;            // chi2 at the estimates the queued look-ahead trial starts from
};

class OptimizationAlgorithmGaussNewtonHip : public OptimizationAlgorithmGaussNewton {
 public:
  explicit OptimizationAlgorithmGaussNewtonHip(Solver* solver)
      : OptimizationAlgorithmGaussNewton(solver), _dev(dynamic_cast<HipDeviceGraph*>(solver)), _resident(false),
        _writeBack(!hip_detail::envOff("G2OHIP_ADAPTER_WRITEBACK")), _fetched(false) {}

  bool deviceLoopActive() const { return _resident; }

  virtual SolverResult solve(int iteration, bool online = false) {
    // (gauss_newton.cpp:57-71 evaluates the errors before it builds the structure -- for max-mixture components; none of the
    // edge types a device front end takes reads them there, and the host loop below evaluates them again)
    if (iteration == 0 && !online) {
      if (!_solver->buildStructure()) {
        std::cerr << "OptimizationAlgorithmGaussNewtonHip::solve: Failure while building CCS structure" << std::endl;
        return OptimizationAlgorithm::Fail;
      }
    }
    _resident = hip_detail::decideDeviceLoop("OptimizationAlgorithmGaussNewtonHip", _solver, _dev, _resident, iteration == 0 || online);
    if (!_resident) return OptimizationAlgorithmGaussNewton::solve(iteration, true);

    double t = get_monotonic_time();
    G2OBatchStatistics* globalStats = G2OBatchStatistics::globalStats();
    // (look-ahead as in the Levenberg-Marquardt driver: the previous solve() of this run may have queued this iteration --
    // errors, buildSystem, push, solve, update, errors, the status read-back -- before it wrote its estimates into the vertices)
    const bool consume = _dev->devLookAheadPending() && iteration > 0 && !online && _dev->devEstimatesValid();
    if (!consume) _dev->devDropLookAhead();
    int ok = -1;
    bool fetched = false;
    if (consume) {
      // (BatchStatistics on this path: the errors / buildSystem / solve of this iteration ran inside the previous solve()'s
      // write-back; timeResiduals and timeQuadraticForm are reported as 0 -- not left over from an earlier iteration -- and
      // timeLinearSolution is the wait for the queued solve's status.  G2OHIP_ADAPTER_LOOKAHEAD=0 restores the reference's split.)
      if (globalStats) globalStats->timeResiduals = globalStats->timeQuadraticForm = 0.;
      _dev->devSetLookAheadPending(false);
      double chi = 0., scale = 0.;
      ok = _dev->devTrialStats(0., chi, scale);          // (the status of the queued solve; the sums ride along unused)
      if (ok == 1) {
        if (!_dev->devDiscardTop()) return OptimizationAlgorithm::Fail;
        fetched = _fetched;
      } else {
        if (_fetched) _dev->devFetchCancel();
        if (!_dev->devPop()) return OptimizationAlgorithm::Fail;     // the increment of a factorisation that broke down is not applied
        if (ok == 2 || ok == 0) {                          // (a dependency-driven launch gave up waiting, or the undamped factorisation broke down:
          ok = _dev->devSolve();                           //  again, synchronously -- devSolve repeats a singular undamped solve once with a tiny lambda)
          if (ok == 1 && !_dev->devUpdate()) return OptimizationAlgorithm::Fail;
        }
      }
      _fetched = false;
      if (globalStats) {
        globalStats->timeLinearSolution = get_monotonic_time() - t;
        t = get_monotonic_time();
      }
    } else {
      if (iteration == 0 || online || !_dev->devEstimatesValid()) {
        if (!_dev->devSetEstimates()) return OptimizationAlgorithm::Fail;
      }
      if (!_dev->devLinearize(true)) return OptimizationAlgorithm::Fail;
      if (globalStats) {
        globalStats->timeResiduals = get_monotonic_time() - t;
        t = get_monotonic_time();
      }
      if (!_dev->devBuildSystem()) return OptimizationAlgorithm::Fail;
      if (globalStats) {
        globalStats->timeQuadraticForm = get_monotonic_time() - t;
        t = get_monotonic_time();
      }
      ok = _dev->devSolve();
      if (globalStats) {
        globalStats->timeLinearSolution = get_monotonic_time() - t;
        t = get_monotonic_time();
      }
      // (the host loop applies x() even after a failed solve and then reports Fail, gauss_newton.cpp:86-92; the increment of a
      // factorisation that broke down is not applied here)
      if (ok == 1 && !_dev->devUpdate()) return OptimizationAlgorithm::Fail;
    }
    if (ok < 0) return OptimizationAlgorithm::Fail;
    if (ok == 1 && _writeBack) {
      if (!fetched) fetched = _dev->devFetchBegin();
      // (iteration 0 never looks ahead: optimize(1) in a loop would always drop the queued iteration)
      const bool ahead = fetched && iteration > 0 && !online && _dev->devCanLookAhead();
      if (ahead) {
        _dev->devSetQueueing(true);
        const bool q = _dev->devLinearize(true) && _dev->devBuildSystem() && _dev->devPush() && _dev->devSolveAsync() && _dev->devUpdate() &&
                       _dev->devLinearize(false) && _dev->devTrialStatsBegin(0.);
        _dev->devSetQueueing(false);
        if (!q) {
          (void)_dev->devPop();                            // (a device error half-way: no level may stay on the estimate stack)
          return OptimizationAlgorithm::Fail;
        }
      }
      if (!(fetched ? _dev->devFetchEnd() : _dev->devGetEstimates())) return OptimizationAlgorithm::Fail;
      if (ahead) {
        _fetched = _dev->devFetchBegin();                  // the queued iteration's estimates, behind its update
        _dev->devSetLookAheadPending(true);
      }
    }
    if (globalStats) globalStats->timeUpdate = get_monotonic_time() - t;
    return ok == 1 ? OK : Fail;
  }

 private:
  HipDeviceGraph* _dev;
  bool _resident, _writeBack;
  bool _fetched;              // the queued iteration's estimates are on their way to the host
};

}  // namespace g2o

#endif
