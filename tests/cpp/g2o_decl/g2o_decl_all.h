// TEST INFRASTRUCTURE ONLY.  Declarations (no implementation) of the g2o members the adapter
// openslam_g2o_amd/cpp/g2o_hip_solver.h + solver_hip.cpp touches, written from the reference's public interface so that
// the adapter can be syntax- and type-checked (-fsyntax-only) in a container without g2o / Eigen.  Each declaration
// cites the reference header it mirrors (paths relative to /root/reference).  Nothing here is linked or shipped.
#ifndef G2O_DECL_ALL_H
#define G2O_DECL_ALL_H
#include <cstddef>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace Eigen {                                       // the two Eigen members the adapter uses: data(), rows()/cols()
template <typename S, int R, int C>
struct Matrix {
  enum { RowsAtCompileTime = R, ColsAtCompileTime = C };
  S* data();
  const S* data() const;
  int rows() const;
  int cols() const;
};
typedef Matrix<double, -1, -1> MatrixXd;
}  // namespace Eigen

namespace g2o {
using Eigen::MatrixXd;                                  // g2o/core/eigen_types.h

double get_monotonic_time();                            // g2o/stuff/timeutil.h:88

struct G2OBatchStatistics {                             // g2o/core/batch_stats.h:40-77
  int iteration, numVertices, numEdges;
  double chi2, timeResiduals, timeLinearize, timeQuadraticForm;
  int levenbergIterations;
  double timeSchurComplement, timeSymbolicDecomposition, timeNumericDecomposition, timeLinearSolution, timeLinearSolver;
  int iterationsLinearSolver;
  double timeUpdate, timeIteration, timeMarginals;
  size_t hessianDimension, hessianPoseDimension, hessianLandmarkDimension, choleskyNNZ;
  static G2OBatchStatistics* globalStats();
};

class JacobianWorkspace {                               // g2o/core/jacobian_workspace.h:51-95
 public:
  double* workspaceForVertex(int vertexIndex);
};

class RobustKernel {                                    // g2o/core/robust_kernel.h:53-78
 public:
  virtual ~RobustKernel();
  double delta() const;
};
class RobustKernelHuber : public RobustKernel {};       // g2o/core/robust_kernel_impl.h:77
class RobustKernelPseudoHuber : public RobustKernel {}; // :94
class RobustKernelCauchy : public RobustKernel {};      // :108
class RobustKernelSaturated : public RobustKernel {};   // :119
class RobustKernelDCS : public RobustKernel {};         // :132

class HyperGraph {                                      // g2o/core/hyper_graph.h
 public:
  class Vertex { public: virtual ~Vertex(); };
  class Edge {
   public:
    virtual ~Edge();
    const std::vector<Vertex*>& vertices() const;       // hyper_graph.h:140
    Vertex* vertex(size_t i);                            // hyper_graph.h:147-148
  };
  typedef std::set<Edge*> EdgeSet;                       // hyper_graph.h:90
};

class OptimizableGraph : public HyperGraph {            // g2o/core/optimizable_graph.h
 public:
  class Vertex : public HyperGraph::Vertex {
   public:
    virtual void mapHessianMemory(double* d) = 0;       // :161
    int hessianIndex() const;                            // :299
    bool fixed() const;                                  // :306
    bool marginalized() const;                           // :311
    int dimension() const;                               // :316
    void setColInHessian(int c);                         // :322
  };
  class Edge : public HyperGraph::Edge {
   public:
    RobustKernel* robustKernel() const;                  // :416
    virtual const double* errorData() const = 0;         // :423
    virtual const double* informationData() const = 0;   // :427
    virtual void linearizeOplus(JacobianWorkspace& jacobianWorkspace) = 0;   // :455
    int dimension() const;                               // :473
  };
  typedef std::vector<OptimizableGraph::Vertex*> VertexContainer;   // :121
  typedef std::vector<OptimizableGraph::Edge*> EdgeContainer;       // :123
  JacobianWorkspace& jacobianWorkspace();               // :660
};

class SparseOptimizer : public OptimizableGraph {       // g2o/core/sparse_optimizer.h
 public:
  const VertexContainer& indexMapping() const;          // :192
  const EdgeContainer& activeEdges() const;             // :196
};

template <class MatrixType>
class SparseBlockMatrix {                               // g2o/core/sparse_block_matrix.h:61-220
 public:
  typedef MatrixType SparseMatrixBlock;
  typedef std::map<int, SparseMatrixBlock*> IntBlockMap; // :73
  int cols() const;                                      // :69
  int rows() const;                                      // :71
  SparseBlockMatrix();                                   // :88
  SparseBlockMatrix(const int* rbi, const int* cbi, int rb, int cb, bool hasStorage = true);   // :81
  SparseMatrixBlock* block(int r, int c, bool alloc = false);   // :97
  const std::vector<IntBlockMap>& blockCols() const;     // :178
  const std::vector<int>& rowBlockIndices() const;       // :182
};

class Solver {                                          // g2o/core/solver.h:44-149
 public:
  Solver();
  virtual ~Solver();
  virtual bool init(SparseOptimizer* optimizer, bool online = false) = 0;
  virtual bool buildStructure(bool zeroBlocks = false) = 0;
  virtual bool updateStructure(const std::vector<HyperGraph::Vertex*>& vset, const HyperGraph::EdgeSet& edges) = 0;
  virtual bool buildSystem() = 0;
  virtual bool solve() = 0;
  virtual bool computeMarginals(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices) = 0;
  virtual bool setLambda(double lambda, bool backup = false) = 0;
  virtual void restoreDiagonal() = 0;
  double* x();
  double* b();
  size_t vectorSize() const;
  virtual bool supportsSchur();
  virtual bool schur() = 0;
  virtual void setSchur(bool s) = 0;
  virtual void setWriteDebug(bool) = 0;
  virtual bool writeDebug() const = 0;
  virtual bool saveHessian(const std::string&) const = 0;

 protected:
  SparseOptimizer* _optimizer;
  double* _x;
  double* _b;
  size_t _xSize, _maxXSize;
  bool _isLevenberg;
  size_t _additionalVectorSpace;
  void resizeVector(size_t sx);
};

class BlockSolverBase : public Solver {                 // g2o/core/block_solver.h:83-91
 public:
  virtual ~BlockSolverBase();
  virtual void multiplyHessian(double* dest, const double* src) const = 0;
};

template <typename MatrixType>
class LinearSolver {                                    // g2o/core/linear_solver.h:40-81
 public:
  LinearSolver();
  virtual ~LinearSolver();
  virtual bool init() = 0;
  virtual bool solve(const SparseBlockMatrix<MatrixType>& A, double* x, double* b) = 0;
  virtual bool solveBlocks(double**& blocks, const SparseBlockMatrix<MatrixType>& A);   // :64
  virtual bool solvePattern(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices,
                            const SparseBlockMatrix<MatrixType>& A);   // :71
};

template <int _PoseDim, int _LandmarkDim>
struct BlockSolverTraits {                              // g2o/core/block_solver.h:43-57
  static const int PoseDim = _PoseDim;
  static const int LandmarkDim = _LandmarkDim;
  typedef Eigen::Matrix<double, PoseDim, PoseDim> PoseMatrixType;
  typedef LinearSolver<PoseMatrixType> LinearSolverType;
};

template <typename Traits>
class BlockSolver : public BlockSolverBase {            // g2o/core/block_solver.h:98-178
 public:
  typedef typename Traits::PoseMatrixType PoseMatrixType;
  typedef typename Traits::LinearSolverType LinearSolverType;
  BlockSolver(LinearSolverType* linearSolver);           // :116 (takes ownership)
  virtual bool init(SparseOptimizer* optimizer, bool online = false);
  virtual bool buildStructure(bool zeroBlocks = false);
  virtual bool updateStructure(const std::vector<HyperGraph::Vertex*>& vset, const HyperGraph::EdgeSet& edges);
  virtual bool buildSystem();
  virtual bool solve();
  virtual bool computeMarginals(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices);
  virtual bool setLambda(double lambda, bool backup = false);
  virtual void restoreDiagonal();
  virtual bool schur();
  virtual void setSchur(bool s);
  virtual void setWriteDebug(bool);
  virtual bool writeDebug() const;
  virtual bool saveHessian(const std::string&) const;
  virtual void multiplyHessian(double* dest, const double* src) const;
};

class OptimizationAlgorithm { public: virtual ~OptimizationAlgorithm(); };   // g2o/core/optimization_algorithm.h
class OptimizationAlgorithmGaussNewton : public OptimizationAlgorithm {      // optimization_algorithm_gauss_newton.h:46
 public: explicit OptimizationAlgorithmGaussNewton(Solver* solver);
};
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {        // optimization_algorithm_levenberg.h:45
 public: explicit OptimizationAlgorithmLevenberg(Solver* solver);
};
class OptimizationAlgorithmDogleg : public OptimizationAlgorithm {           // optimization_algorithm_dogleg.h:57
 public: explicit OptimizationAlgorithmDogleg(BlockSolverBase* solver);
};

struct OptimizationAlgorithmProperty {                  // g2o/core/optimization_algorithm_property.h:39-55
  std::string name, desc, type;
  bool requiresMarginalize;
  int poseDim, landmarkDim;
  OptimizationAlgorithmProperty(const std::string& name_, const std::string& desc_, const std::string& type_,
                                bool requiresMarginalize_, int poseDim_, int landmarkDim_);
};

class AbstractOptimizationAlgorithmCreator {            // g2o/core/optimization_algorithm_factory.h:55-66
 public:
  AbstractOptimizationAlgorithmCreator(const OptimizationAlgorithmProperty& p);
  virtual ~AbstractOptimizationAlgorithmCreator();
  virtual OptimizationAlgorithm* construct() = 0;
  const OptimizationAlgorithmProperty& property() const;
};

class RegisterOptimizationAlgorithmProxy {              // optimization_algorithm_factory.h:120-141
 public:
  RegisterOptimizationAlgorithmProxy(AbstractOptimizationAlgorithmCreator* c);
  ~RegisterOptimizationAlgorithmProxy();
};

}  // namespace g2o

// optimization_algorithm_factory.h:153-162 (non-MSVC branch)
#define G2O_REGISTER_OPTIMIZATION_LIBRARY(libraryname) \
  extern "C" void g2o_optimization_library_##libraryname(void) {}
#define G2O_REGISTER_OPTIMIZATION_ALGORITHM(optimizername, instance)    \
  extern "C" void g2o_optimization_algorithm_##optimizername(void) {}   \
  static g2o::RegisterOptimizationAlgorithmProxy g_optimization_algorithm_proxy_##optimizername(instance);

#endif
