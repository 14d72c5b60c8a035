// C++ consumer of the C ABI (no Python, no torch): a small random bundle-adjustment-shaped
// problem is solved through g2o_hip::HipBlockSolver_6_3 and verified by the residual of the
// damped normal equations computed with multiplyHessian.  Built and run by
// tests/test_gpu_cabi_cpp.py on the GPU box.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hip_block_solver.hpp"

static double urand(unsigned long long& s) {
  s = s * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)((s >> 11) & 0xFFFFFFFFFFFFFULL) / 4503599627370496.0 - 0.5;
}

int main() {
  const int nP = 50, nL = 400, K = 4, E = nL * K;
  std::vector<int32_t> v0(E), v1(E);
  std::vector<double> J0(E * 6), J1(E * 12), Om(E * 4), err(E * 2);
  unsigned long long seed = 12345;
  for (int j = 0; j < nL; ++j)
    for (int k = 0; k < K; ++k) {
      int e = j * K + k;
      v0[e] = nP + j;
      int pose = (j * nP / nL + k) % nP;
      v1[e] = (pose == 0) ? -1 : pose;  // pose 0 plays the fixed gauge vertex
      if (v1[e] > 0) v1[e] -= 0;
    }
  for (auto& v : J0) v = urand(seed);
  for (auto& v : J1) v = urand(seed);
  for (int e = 0; e < E; ++e) {
    Om[4 * e] = 1.0 + 0.1 * e / E;
    Om[4 * e + 3] = 2.0;
    Om[4 * e + 1] = Om[4 * e + 2] = 0.25;
    err[2 * e] = urand(seed);
    err[2 * e + 1] = urand(seed);
  }
  g2o_hip::HipBlockSolver_6_3 solver(0);
  if (!solver.valid()) return 2;
  int set = solver.addEdgeSet(2, E, v0.data(), v1.data());
  if (set < 0) return 3;
  if (!solver.buildStructure(nP, nL, true)) return 4;
  if (!solver.setEdgeData(set, J0.data(), J1.data(), Om.data(), err.data())) return 5;
  if (!solver.buildSystem()) return 6;
  const double lambda = 1e-3 * solver.maxDiagonal();
  solver.setLambda(lambda, true);
  if (!solver.solve()) return 7;
  std::vector<double> r(solver.vectorSize(), 0.0);
  solver.multiplyHessian(r.data(), solver.x());  // H already carries lambda
  double rmax = 0, bmax = 0;
  for (size_t i = 0; i < r.size(); ++i) {
    rmax = std::fmax(rmax, std::fabs(r[i] - solver.b()[i]));
    bmax = std::fmax(bmax, std::fabs(solver.b()[i]));
  }
  solver.restoreDiagonal();
  // indefinite system must be reported as solve() == false
  solver.setLambda(-100.0 * solver.maxDiagonal(), true);
  bool bad = solver.solve();
  solver.restoreDiagonal();
  // marginals: block (3,3) of the inverse of the (damped: some landmarks here have a single observation) reduced
  // system is symmetric with a positive diagonal
  solver.setLambda(lambda, true);
  const int32_t mr[1] = {3}, mc[1] = {3};
  double M[36];
  bool mok = solver.computeMarginals(1, mr, mc, M);
  for (int i = 0; i < 6 && mok; ++i) {
    if (!(M[i * 7] > 0)) mok = false;
    for (int j = 0; j < i; ++j)
      if (std::fabs(M[i + 6 * j] - M[j + 6 * i]) > 1e-9 * std::fabs(M[i * 7])) mok = false;
  }
  // PCG on the same damped system agrees with the direct solve
  bool pok = solver.solve();
  std::vector<double> xd(solver.x(), solver.x() + solver.vectorSize());
  pok = pok && solver.usePCG(true, 1e-24) && solver.solve();
  double dmax = 0, xmax = 0;
  for (size_t i = 0; i < xd.size(); ++i) {
    dmax = std::fmax(dmax, std::fabs(xd[i] - solver.x()[i]));
    xmax = std::fmax(xmax, std::fabs(xd[i]));
  }
  solver.usePCG(false);
  solver.restoreDiagonal();
  std::printf("residual %.3e chi2 %.6e notpd_detected %d marginals %d pcg_vs_direct %.2e\n", rmax / bmax, solver.chi2(), bad ? 0 : 1,
              mok ? 1 : 0, dmax / xmax);
  return (rmax <= 1e-11 * bmax && !bad && mok && pok && dmax <= 1e-7 * xmax) ? 0 : 1;
}
