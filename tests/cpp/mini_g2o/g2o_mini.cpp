// TEST INFRASTRUCTURE ONLY (see g2o_mini.h): the non-template part of the test host -- the optimisation-algorithm
// factory singleton the plugin registers with, SparseOptimizer's bookkeeping, the Gauss-Newton and Levenberg-Marquardt
// outer loops.  Built into libg2o_mini_core.so, which both the host program and the plugin (libg2o_solver_hip.so) link,
// the way g2o's libg2o_core.so sits between the g2o CLI and its solver plugins.
#include "g2o_mini.h"

#include <time.h>

namespace Eigen {
Matrix3d Quaterniond::toRotationMatrix() const {
  Matrix3d R;
  const double tx = 2 * x_, ty = 2 * y_, tz = 2 * z_;
  const double twx = tx * w_, twy = ty * w_, twz = tz * w_, txx = tx * x_, txy = ty * x_, txz = tz * x_, tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
  R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
  R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
  return R;
}
Quaterniond::Quaterniond(const Matrix3d& R) {
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  double q[4];   // x, y, z, w
  if (tr > 0) {
    double t = std::sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R(2, 1) - R(1, 2)) * t; q[1] = (R(0, 2) - R(2, 0)) * t; q[2] = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R(k, j) - R(j, k)) * t; q[j] = (R(j, i) + R(i, j)) * t; q[k] = (R(k, i) + R(i, k)) * t;
  }
  x_ = q[0]; y_ = q[1]; z_ = q[2]; w_ = q[3];
}
}  // namespace Eigen

namespace g2o {

double get_monotonic_time() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static G2OBatchStatistics* g_stats = 0;
G2OBatchStatistics* G2OBatchStatistics::globalStats() { return g_stats; }
void G2OBatchStatistics::setGlobalStats(G2OBatchStatistics* b) { g_stats = b; }

OptimizableGraph::~OptimizableGraph() {
  for (size_t k = 0; k < _edges.size(); ++k) delete _edges[k];
  for (std::map<int, Vertex*>::iterator it = _vertices.begin(); it != _vertices.end(); ++it) delete it->second;
}

SparseOptimizer::~SparseOptimizer() { delete _algorithm; }   // sparse_optimizer.cpp:56-59

void SparseOptimizer::setAlgorithm(OptimizationAlgorithm* algorithm) {
  delete _algorithm;
  _algorithm = algorithm;
  if (_algorithm) _algorithm->setOptimizer(this);
}

// poses (not marginalized) first, then the marginalized vertices, each in id order; fixed vertices get index -1
bool SparseOptimizer::initializeOptimization(int) {
  _ivMap.clear();
  _activeEdges = _edges;
  for (int pass = 0; pass < 2; ++pass)
    for (std::map<int, Vertex*>::iterator it = _vertices.begin(); it != _vertices.end(); ++it) {
      Vertex* v = it->second;
      if (v->fixed()) { v->setHessianIndex(-1); continue; }
      if ((pass == 1) != v->marginalized()) continue;
      v->setHessianIndex((int)_ivMap.size());
      _ivMap.push_back(v);
    }
  int maxDim = 1, maxErr = 1;
  for (size_t i = 0; i < _ivMap.size(); ++i) maxDim = std::max(maxDim, _ivMap[i]->dimension());
  for (std::map<int, Vertex*>::iterator it = _vertices.begin(); it != _vertices.end(); ++it) maxDim = std::max(maxDim, it->second->dimension());
  for (size_t k = 0; k < _activeEdges.size(); ++k) maxErr = std::max(maxErr, _activeEdges[k]->dimension());
  int maxVerts = 2;                                      // (jacobian_workspace.cpp:50-70: sized by the widest active edge)
  for (size_t k = 0; k < _activeEdges.size(); ++k) maxVerts = std::max(maxVerts, (int)_activeEdges[k]->vertices().size());
  _jacobianWorkspace.allocate(maxVerts, maxDim * maxErr);
  return true;
}

// sparse_optimizer.cpp:445-479: the new edges become active, the new free vertices take the next hessian indices, the
// algorithm (-> its Solver) grows the structure
bool SparseOptimizer::updateInitialization(HyperGraph::VertexSet& vset, HyperGraph::EdgeSet& eset) {
  std::vector<HyperGraph::Vertex*> newVertices;
  for (HyperGraph::EdgeSet::iterator it = eset.begin(); it != eset.end(); ++it) _activeEdges.push_back(static_cast<OptimizableGraph::Edge*>(*it));
  for (HyperGraph::VertexSet::iterator it = vset.begin(); it != vset.end(); ++it) {
    OptimizableGraph::Vertex* v = static_cast<OptimizableGraph::Vertex*>(*it);
    if (v->fixed()) { v->setHessianIndex(-1); continue; }
    if (v->marginalized()) return false;                 // (the reference aborts)
    v->setHessianIndex((int)_ivMap.size());
    _ivMap.push_back(v);
    newVertices.push_back(v);
  }
  return _algorithm && _algorithm->updateStructure(newVertices, eset);
}

void SparseOptimizer::computeActiveErrors() {
  for (size_t k = 0; k < _activeEdges.size(); ++k) _activeEdges[k]->computeError();
}

double SparseOptimizer::activeRobustChi2() const {
  double chi = 0.0;
  double rho[3];
  for (size_t k = 0; k < _activeEdges.size(); ++k) {
    const OptimizableGraph::Edge* e = _activeEdges[k];
    if (e->robustKernel()) {
      e->robustKernel()->robustify(e->chi2(), rho);
      chi += rho[0];
    } else {
      chi += e->chi2();
    }
  }
  return chi;
}

void SparseOptimizer::update(const double* update) {
  for (size_t i = 0; i < _ivMap.size(); ++i) {
    _ivMap[i]->oplus(update);
    update += _ivMap[i]->dimension();
  }
}
void SparseOptimizer::push() { for (size_t i = 0; i < _ivMap.size(); ++i) _ivMap[i]->push(); }
void SparseOptimizer::pop() { for (size_t i = 0; i < _ivMap.size(); ++i) _ivMap[i]->pop(); }
void SparseOptimizer::discardTop() { for (size_t i = 0; i < _ivMap.size(); ++i) _ivMap[i]->discardTop(); }

int SparseOptimizer::optimize(int iterations, bool online) {
  if (_ivMap.empty() || !_algorithm) return -1;
  if (!_algorithm->init(online)) return -1;
  int done = 0;
  for (int i = 0; i < iterations; ++i) {
    const OptimizationAlgorithm::SolverResult r = _algorithm->solve(i, online);
    if (r == OptimizationAlgorithm::Fail) return done;
    ++done;
    if (_verbose) {
      computeActiveErrors();
      std::cerr << "iteration= " << i << "\t chi2= " << activeRobustChi2() << "\t edges= " << _activeEdges.size() << std::endl;
    }
    if (r == OptimizationAlgorithm::Terminate) break;
  }
  return done;
}

// optimization_algorithm_with_hessian.cpp:50-73: any marginalized vertex switches the Schur complement on
bool OptimizationAlgorithmWithHessian::init(bool online) {
  bool useSchur = false;
  for (size_t i = 0; i < _optimizer->indexMapping().size(); ++i)
    if (_optimizer->indexMapping()[i]->marginalized()) { useSchur = true; break; }
  if (useSchur) {
    if (_solver->supportsSchur()) _solver->setSchur(true);
  } else if (_solver->supportsSchur()) {
    _solver->setSchur(false);
  }
  return _solver->init(_optimizer, online);
}

// optimization_algorithm_gauss_newton.cpp:50-93
OptimizationAlgorithm::SolverResult OptimizationAlgorithmGaussNewton::solve(int iteration, bool online) {
  if (iteration == 0 && !online && !_solver->buildStructure()) return Fail;   // gauss_newton.cpp:65
  _optimizer->computeActiveErrors();
  _solver->buildSystem();
  if (!_solver->solve()) return Fail;
  _optimizer->update(_solver->x());
  return OK;
}

// optimization_algorithm_levenberg.cpp:149-163
double OptimizationAlgorithmLevenberg::computeLambdaInit() const {
  if (_userLambdaInit > 0) return _userLambdaInit;
  double maxDiagonal = 0.;
  for (size_t k = 0; k < _optimizer->indexMapping().size(); ++k) {
    OptimizableGraph::Vertex* v = _optimizer->indexMapping()[k];
    for (int j = 0; j < v->dimension(); ++j) maxDiagonal = std::max(std::fabs(v->hessian(j, j)), maxDiagonal);
  }
  return _tau * maxDiagonal;
}
// :165-172
double OptimizationAlgorithmLevenberg::computeScale() const {
  double scale = 0.;
  for (size_t j = 0; j < _solver->vectorSize(); ++j) scale += _solver->x()[j] * (_currentLambda * _solver->x()[j] + _solver->b()[j]);
  return scale;
}
// :57-146
OptimizationAlgorithm::SolverResult OptimizationAlgorithmLevenberg::solve(int iteration, bool online) {
  if (iteration == 0 && !online && !_solver->buildStructure()) return Fail;   // levenberg.cpp:62
  double t = get_monotonic_time();                       // the BatchStatistics fields as optimization_algorithm_levenberg.cpp:70-113 fills them
  _optimizer->computeActiveErrors();
  G2OBatchStatistics* globalStats = G2OBatchStatistics::globalStats();
  if (globalStats) {
    globalStats->timeResiduals = get_monotonic_time() - t;
    t = get_monotonic_time();
  }
  double currentChi = _optimizer->activeRobustChi2();
  double tempChi = currentChi;
  _solver->buildSystem();
  if (globalStats) globalStats->timeQuadraticForm = get_monotonic_time() - t;
  if (iteration == 0) {
    _currentLambda = computeLambdaInit();
    _ni = 2;
  }
  double rho = 0;
  int qmax = 0;
  do {
    _optimizer->push();
    if (globalStats) {
      globalStats->levenbergIterations++;
      t = get_monotonic_time();
    }
    _solver->setLambda(_currentLambda, true);
    const bool ok2 = _solver->solve();
    if (globalStats) {
      globalStats->timeLinearSolution += get_monotonic_time() - t;
      t = get_monotonic_time();
    }
    _optimizer->update(_solver->x());
    if (globalStats) globalStats->timeUpdate = get_monotonic_time() - t;
    _solver->restoreDiagonal();
    _optimizer->computeActiveErrors();
    tempChi = _optimizer->activeRobustChi2();
    if (!ok2) tempChi = DBL_MAX;
    rho = currentChi - tempChi;
    double scale = ok2 ? computeScale() : 0.;
    scale += 1e-3;
    rho /= scale;
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow(2 * rho - 1, 3);
      alpha = std::min(alpha, _goodStepUpperScale);
      const double scaleFactor = std::max(_goodStepLowerScale, alpha);
      _currentLambda *= scaleFactor;
      _ni = 2;
      currentChi = tempChi;
      _optimizer->discardTop();
    } else {
      _currentLambda *= _ni;
      _ni *= 2;
      _optimizer->pop();
    }
    qmax++;
  } while (rho < 0 && qmax < _maxTrialsAfterFailure);
  _levenbergIterations = qmax;
  if (qmax == _maxTrialsAfterFailure || rho == 0) return Terminate;
  return OK;
}

OptimizationAlgorithmFactory* OptimizationAlgorithmFactory::instance() {
  static OptimizationAlgorithmFactory* f = new OptimizationAlgorithmFactory();
  return f;
}
void OptimizationAlgorithmFactory::registerSolver(AbstractOptimizationAlgorithmCreator* c) {
  for (CreatorList::iterator it = _creator.begin(); it != _creator.end(); ++it)
    if ((*it)->property().name == c->property().name) {
      _creator.erase(it);
      break;
    }
  _creator.push_back(c);
}
void OptimizationAlgorithmFactory::unregisterSolver(AbstractOptimizationAlgorithmCreator* c) {
  for (CreatorList::iterator it = _creator.begin(); it != _creator.end(); ++it)
    if (*it == c) {
      _creator.erase(it);
      delete c;
      return;
    }
}
OptimizationAlgorithm* OptimizationAlgorithmFactory::construct(const std::string& tag, OptimizationAlgorithmProperty& solverProperty) const {
  for (CreatorList::const_iterator it = _creator.begin(); it != _creator.end(); ++it)
    if ((*it)->property().name == tag) {
      solverProperty = (*it)->property();
      return (*it)->construct();
    }
  return 0;
}
void OptimizationAlgorithmFactory::listSolvers(std::ostream& os) const {
  for (CreatorList::const_iterator it = _creator.begin(); it != _creator.end(); ++it) os << (*it)->property().name << "\t" << (*it)->property().desc << std::endl;
}

}  // namespace g2o
