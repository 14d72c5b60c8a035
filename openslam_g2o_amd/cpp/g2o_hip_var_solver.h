// "<method>_var_hip": the solver names g2o registers for BlockSolverX (block_solver.h:186, solver_csparse.cpp:54-59) -- the
// variable-block-size solver most g2o applications ask for by default -- over the fixed-shape device solvers.
//
// BlockSolverX keeps Eigen::Dynamic blocks so that ONE solver object serves any graph.  The device path is compiled for the
// three shapes the reference instantiates (3-2, 6-3, 7-3); BlockSolverHipVar looks at the graph when the algorithm initialises
// the solver (Solver::init, called by OptimizationAlgorithmWithHessian::init at the start of every optimize() -- after
// initializeOptimization(), so indexMapping() is final), picks the shape from the dimension of the non-marginalised and of
// the marginalised vertices and forwards every call to a BlockSolverHip<p, l> of that shape.  A graph whose vertices do not
// all fit ONE of the three shapes (truly variable block sizes) is refused with a message: that is BlockSolverX's own domain
// and stays on the host solvers.
#ifndef G2O_HIP_VAR_SOLVER_H
#define G2O_HIP_VAR_SOLVER_H

#include <iostream>

#include "g2o_hip_solver.h"

namespace g2o {

class BlockSolverHipVar : public BlockSolverBase, public HipDeviceGraph {
 public:
  BlockSolverHipVar() : _inner(0), _dev(0), _p(0), _l(0), _schurFlag(true), _debugFlag(false) {}
  virtual ~BlockSolverHipVar() {
    _x = _b = 0;                                         // (the inner solver's arrays: ~Solver must not free them)
    delete _inner;
  }
  int poseDim() const { return _p; }
  int landmarkDim() const { return _l; }

  virtual bool init(SparseOptimizer* optimizer, bool online = false) {
    _optimizer = optimizer;
    int p = -1, l = -1;
    bool mixed = false;
    for (size_t i = 0; i < optimizer->indexMapping().size(); ++i) {
      const OptimizableGraph::Vertex* v = optimizer->indexMapping()[i];
      int& d = v->marginalized() ? l : p;
      if (d < 0) d = v->dimension();
      else if (d != v->dimension()) mixed = true;
    }
    if (l < 0) l = p == 3 ? 2 : 3;                         // (no marginalised vertex: the landmark side of the shape is unused)
    if (mixed || !((p == 3 && l == 2) || (p == 6 && l == 3) || (p == 7 && l == 3))) {
      std::cerr << "BlockSolverHipVar: the graph does not have ONE pose and ONE landmark dimension among 3-2 / 6-3 / 7-3 (found "
                << p << "-" << l << (mixed ? ", mixed" : "") << "): variable block sizes stay on g2o's BlockSolverX" << std::endl;
      return false;
    }
    if (!_inner || p != _p || l != _l) {
      _x = _b = 0;
      delete _inner;
      _inner = p == 3 ? static_cast<BlockSolverBase*>(new BlockSolverHip<3, 2>())
                      : (p == 6 ? static_cast<BlockSolverBase*>(new BlockSolverHip<6, 3>()) : static_cast<BlockSolverBase*>(new BlockSolverHip<7, 3>()));
      _dev = dynamic_cast<HipDeviceGraph*>(_inner);
      _p = p;
      _l = l;
      _inner->setSchur(_schurFlag);
      _inner->setWriteDebug(_debugFlag);
    }
    _inner->setLevenberg(_isLevenberg);
    const bool ok = _inner->init(optimizer, online);
    adopt();
    return ok;
  }
  virtual bool buildStructure(bool zeroBlocks = false) {
    if (!_inner) return false;
    const bool ok = _inner->buildStructure(zeroBlocks);
    adopt();
    return ok;
  }
  virtual bool updateStructure(const std::vector<HyperGraph::Vertex*>& vset, const HyperGraph::EdgeSet& edges) {
    if (!_inner) return false;
    const bool ok = _inner->updateStructure(vset, edges);
    adopt();
    return ok;
  }
  virtual bool buildSystem() { return _inner && _inner->buildSystem(); }
  virtual bool solve() { return _inner && _inner->solve(); }
  virtual bool computeMarginals(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices) {
    return _inner && _inner->computeMarginals(spinv, blockIndices);
  }
  virtual bool setLambda(double lambda, bool backup = false) { return _inner && _inner->setLambda(lambda, backup); }
  virtual void restoreDiagonal() { if (_inner) _inner->restoreDiagonal(); }
  virtual bool supportsSchur() { return true; }
  virtual bool schur() { return _inner ? _inner->schur() : _schurFlag; }
  virtual void setSchur(bool s) {
    _schurFlag = s;
    if (_inner) _inner->setSchur(s);
  }
  virtual void setWriteDebug(bool b) {
    _debugFlag = b;
    if (_inner) _inner->setWriteDebug(b);
  }
  virtual bool writeDebug() const { return _inner ? _inner->writeDebug() : _debugFlag; }
  virtual bool saveHessian(const std::string& fileName) const { return _inner && _inner->saveHessian(fileName); }
  virtual void multiplyHessian(double* dest, const double* src) const { if (_inner) _inner->multiplyHessian(dest, src); }

  // ---- HipDeviceGraph: the device-resident drivers see the inner solver's front ends
  virtual bool devFetchBegin() { return _dev && _dev->devFetchBegin(); }
  virtual bool devFetchEnd() { return _dev && _dev->devFetchEnd(); }
  virtual void devFetchCancel() { if (_dev) _dev->devFetchCancel(); }
  virtual bool deviceResident() const { return _dev && _dev->deviceResident(); }
  virtual bool devEstimatesValid() const { return _dev && _dev->devEstimatesValid(); }
  virtual bool devSetEstimates() { return _dev && _dev->devSetEstimates(); }
  virtual bool devGetEstimates() { return _dev && _dev->devGetEstimates(); }
  virtual bool devLinearize(bool jacobians) { return _dev && _dev->devLinearize(jacobians); }
  virtual bool devChi2(double& chi2) { return _dev && _dev->devChi2(chi2); }
  virtual bool devBuildSystem() { return _dev && _dev->devBuildSystem(); }
  virtual bool devMaxDiagonal(double& d) { return _dev && _dev->devMaxDiagonal(d); }
  virtual bool devComputeScale(double lambda, double& scale) { return _dev && _dev->devComputeScale(lambda, scale); }
  virtual int devSolve() { return _dev ? _dev->devSolve() : -1; }
  virtual bool devSolveAsync() { return _dev && _dev->devSolveAsync(); }
  virtual int devTrialStats(double lambda, double& chi2, double& scale) { return _dev ? _dev->devTrialStats(lambda, chi2, scale) : -1; }
  virtual bool devUpdate() { return _dev && _dev->devUpdate(); }
  virtual bool devPush() { return _dev && _dev->devPush(); }
  virtual bool devPop() { return _dev && _dev->devPop(); }
  virtual bool devDiscardTop() { return _dev && _dev->devDiscardTop(); }
  virtual bool devCanLookAhead() const { return _dev && _dev->devCanLookAhead(); }
  virtual bool devHybrid() const { return _dev && _dev->devHybrid(); }
  virtual bool devTrialStatsBegin(double lambda) { return _dev && _dev->devTrialStatsBegin(lambda); }
  virtual bool devLookAheadPending() const { return _dev && _dev->devLookAheadPending(); }
  virtual void devSetLookAheadPending(bool on) { if (_dev) _dev->devSetLookAheadPending(on); }
  virtual void devDropLookAhead() { if (_dev) _dev->devDropLookAhead(); }
  virtual void devSetQueueing(bool on) { if (_dev) _dev->devSetQueueing(on); }

 private:
  // x() / b() of Solver are not virtual: this object's pointers follow the inner solver's arrays
  void adopt() {
    _x = _inner->x();
    _b = _inner->b();
    _xSize = _inner->vectorSize();
  }
  BlockSolverBase* _inner;
  HipDeviceGraph* _dev;
  int _p, _l;
  bool _schurFlag, _debugFlag;
};

}  // namespace g2o

#endif
