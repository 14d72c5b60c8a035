// g2o plugin: registers the MI355X solvers with g2o's OptimizationAlgorithmFactory so that
//   g2o -solver lm_fix6_3_hip ...      (wide seam: BlockSolverHip)
//   g2o -solver lm_fix6_3_hipls ...    (narrow seam: g2o's BlockSolver + LinearSolverHip)
//   g2o -solver lm_fix6_3_hipdev ...   (BlockSolverHip under the device-resident Levenberg / Gauss-Newton loop of
//                                       g2o_hip_algorithm.h: the whole iteration on the GPU where the graph allows it)
// work once the shared object -- its file name has to match *_solver_*.so, e.g. libg2o_solver_hip.so -- sits in
// G2O_SOLVERS_DIR or next to the CLI library, or is passed with -solverlib
// (/root/reference/g2o/apps/g2o_cli/g2o_common.cpp:81-167, dl_wrapper.cpp:118).
// Follows the contract of /root/reference/g2o/solvers/csparse/solver_csparse.cpp:34-97 and
// /root/reference/g2o/core/optimization_algorithm_factory.h:120-162.
// Build (in a g2o tree):  g++ -shared -fPIC solver_hip.cpp -I<g2o> -I<eigen3> -I<repo>/include -I<repo>/openslam_g2o_amd/cpp
//                         -L<repo>/openslam_g2o_amd/lib -lg2ohip -lg2o_core -lg2o_stuff -o libg2o_solver_hip.so
#include <string>

#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_dogleg.h"
#include "g2o/core/optimization_algorithm_factory.h"
#include "g2o/core/optimization_algorithm_gauss_newton.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o_hip_algorithm.h"
#include "g2o_hip_solver.h"
#include "g2o_hip_var_solver.h"

namespace g2o {

namespace {

template <int p, int l>
Solver* allocWide() { return new BlockSolverHip<p, l>(); }

// g2o's own BlockSolver (CPU assembly + Schur complement) over the device Cholesky
template <int p, int l>
Solver* allocNarrow() {
  typedef BlockSolver<BlockSolverTraits<p, l> > SolverType;
  return new SolverType(new LinearSolverHip<typename SolverType::PoseMatrixType>(p));
}

// "<method>_<shape>_<hip|hipls|hipdev>": method gn | lm | dl (hipdev: gn | lm), shape fix3_2 | fix6_3 | fix7_3 | var (hip, hipdev)
OptimizationAlgorithm* createSolver(const std::string& fullSolverName) {
  const std::string method = fullSolverName.substr(0, 2);
  const std::string::size_type us = fullSolverName.rfind('_');
  const std::string shape = fullSolverName.substr(3, us - 3), seam = fullSolverName.substr(us + 1);
  const bool narrow = seam == "hipls";
  Solver* s = 0;
  if (shape == "var") s = narrow ? 0 : new BlockSolverHipVar();   // (the narrow seam's variable shape is g2o's own BlockSolverX over a LinearSolver)
  else if (shape == "fix3_2") s = narrow ? allocNarrow<3, 2>() : allocWide<3, 2>();
  else if (shape == "fix6_3") s = narrow ? allocNarrow<6, 3>() : allocWide<6, 3>();
  else if (shape == "fix7_3") s = narrow ? allocNarrow<7, 3>() : allocWide<7, 3>();
  if (!s) return 0;
  if (seam == "hipdev") {
    if (method == "gn") return new OptimizationAlgorithmGaussNewtonHip(s);
    if (method == "lm") return new OptimizationAlgorithmLevenbergHip(s);
    delete s;
    return 0;
  }
  if (method == "gn") return new OptimizationAlgorithmGaussNewton(s);
  if (method == "lm") return new OptimizationAlgorithmLevenberg(s);
  if (method == "dl") return new OptimizationAlgorithmDogleg(dynamic_cast<BlockSolverBase*>(s));
  delete s;
  return 0;
}

class HipSolverCreator : public AbstractOptimizationAlgorithmCreator {
 public:
  explicit HipSolverCreator(const OptimizationAlgorithmProperty& prop) : AbstractOptimizationAlgorithmCreator(prop) {}
  virtual OptimizationAlgorithm* construct() { return createSolver(property().name); }
};

}  // namespace

G2O_REGISTER_OPTIMIZATION_LIBRARY(hip);

#define G2OHIP_REGISTER(name, desc, p, l) \
  G2O_REGISTER_OPTIMIZATION_ALGORITHM(name, new HipSolverCreator(OptimizationAlgorithmProperty(#name, desc, "MI355X HIP", true, p, l)))

G2OHIP_REGISTER(gn_fix3_2_hip, "Gauss-Newton: multifrontal block Cholesky on MI355X (fixed blocksize)", 3, 2);
G2OHIP_REGISTER(gn_fix6_3_hip, "Gauss-Newton: multifrontal block Cholesky on MI355X (fixed blocksize)", 6, 3);
G2OHIP_REGISTER(gn_fix7_3_hip, "Gauss-Newton: multifrontal block Cholesky on MI355X (fixed blocksize)", 7, 3);
G2OHIP_REGISTER(lm_fix3_2_hip, "Levenberg: multifrontal block Cholesky on MI355X (fixed blocksize)", 3, 2);
G2OHIP_REGISTER(lm_fix6_3_hip, "Levenberg: multifrontal block Cholesky on MI355X (fixed blocksize)", 6, 3);
G2OHIP_REGISTER(lm_fix7_3_hip, "Levenberg: multifrontal block Cholesky on MI355X (fixed blocksize)", 7, 3);
G2OHIP_REGISTER(dl_fix3_2_hip, "Dogleg: multifrontal block Cholesky on MI355X (fixed blocksize)", 3, 2);
G2OHIP_REGISTER(dl_fix6_3_hip, "Dogleg: multifrontal block Cholesky on MI355X (fixed blocksize)", 6, 3);
G2OHIP_REGISTER(dl_fix7_3_hip, "Dogleg: multifrontal block Cholesky on MI355X (fixed blocksize)", 7, 3);
// the narrow seam under every method x shape the CSparse plugin registers with a fixed block size (solver_csparse.cpp:117-140)
G2OHIP_REGISTER(gn_fix3_2_hipls, "Gauss-Newton: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 3, 2);
G2OHIP_REGISTER(gn_fix6_3_hipls, "Gauss-Newton: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 6, 3);
G2OHIP_REGISTER(gn_fix7_3_hipls, "Gauss-Newton: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 7, 3);
G2OHIP_REGISTER(lm_fix3_2_hipls, "Levenberg: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 3, 2);
G2OHIP_REGISTER(lm_fix6_3_hipls, "Levenberg: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 6, 3);
G2OHIP_REGISTER(lm_fix7_3_hipls, "Levenberg: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 7, 3);
G2OHIP_REGISTER(dl_fix3_2_hipls, "Dogleg: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 3, 2);
G2OHIP_REGISTER(dl_fix6_3_hipls, "Dogleg: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 6, 3);
G2OHIP_REGISTER(dl_fix7_3_hipls, "Dogleg: g2o block solver over the MI355X Cholesky (LinearSolver seam)", 7, 3);
// the wide seam under the device-resident drivers (g2o_hip_algorithm.h): errors, chi2, oplus and the estimate stack on the device too
G2OHIP_REGISTER(gn_fix3_2_hipdev, "Gauss-Newton: device-resident iteration on MI355X (fixed blocksize)", 3, 2);
G2OHIP_REGISTER(gn_fix6_3_hipdev, "Gauss-Newton: device-resident iteration on MI355X (fixed blocksize)", 6, 3);
G2OHIP_REGISTER(gn_fix7_3_hipdev, "Gauss-Newton: device-resident iteration on MI355X (fixed blocksize)", 7, 3);
G2OHIP_REGISTER(lm_fix3_2_hipdev, "Levenberg: device-resident iteration on MI355X (fixed blocksize)", 3, 2);
G2OHIP_REGISTER(lm_fix6_3_hipdev, "Levenberg: device-resident iteration on MI355X (fixed blocksize)", 6, 3);
G2OHIP_REGISTER(lm_fix7_3_hipdev, "Levenberg: device-resident iteration on MI355X (fixed blocksize)", 7, 3);

// the names of the variable-block-size solver (solver_csparse.cpp:54-59 registers gn_var / lm_var / dl_var): the shape is read off the
// graph at Solver::init (g2o_hip_var_solver.h)
G2OHIP_REGISTER(gn_var_hip, "Gauss-Newton: multifrontal block Cholesky on MI355X (shape 3-2 / 6-3 / 7-3 read off the graph)", -1, -1);
G2OHIP_REGISTER(lm_var_hip, "Levenberg: multifrontal block Cholesky on MI355X (shape 3-2 / 6-3 / 7-3 read off the graph)", -1, -1);
G2OHIP_REGISTER(dl_var_hip, "Dogleg: multifrontal block Cholesky on MI355X (shape 3-2 / 6-3 / 7-3 read off the graph)", -1, -1);
G2OHIP_REGISTER(gn_var_hipdev, "Gauss-Newton: device-resident iteration on MI355X (shape read off the graph)", -1, -1);
G2OHIP_REGISTER(lm_var_hipdev, "Levenberg: device-resident iteration on MI355X (shape read off the graph)", -1, -1);

}  // namespace g2o
