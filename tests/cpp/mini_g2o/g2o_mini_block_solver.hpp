// TEST INFRASTRUCTURE ONLY (see g2o_mini.h): the CPU block solver of the test host -- what stands around the
// LinearSolver on the narrow seam.  Own code over plain vectors; follows the CONTRACT of g2o::BlockSolver
// (/root/reference/g2o/core/block_solver.hpp:142-295 structure, :501-560 buildSystem, :563-604 damping, :353-486 solve
// with the Schur complement, :489-499 computeMarginals), not its data structures.
#ifndef G2O_MINI_BLOCK_SOLVER_HPP
#define G2O_MINI_BLOCK_SOLVER_HPP

namespace g2o {

namespace mini {
// C (ra x cb) += alpha * A' (ra x n)' ... small dense helpers on column-major arrays
inline void atb(const double* A, int n, int ra, const double* B, int cb, double alpha, double* C) {   // C[ra x cb] += alpha A'[ra x n] B[n x cb]
  for (int j = 0; j < cb; ++j)
    for (int i = 0; i < ra; ++i) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += A[k + n * i] * B[k + n * j];
      C[i + ra * j] += alpha * s;
    }
}
inline void ab(const double* A, int ra, int n, const double* B, int cb, double* C) {   // C[ra x cb] = A[ra x n] B[n x cb]
  for (int j = 0; j < cb; ++j)
    for (int i = 0; i < ra; ++i) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += A[i + ra * k] * B[k + n * j];
      C[i + ra * j] = s;
    }
}
inline bool invert_spd(const double* A, int n, double* Ainv) {   // Gauss-Jordan with partial pivoting (n <= 3 here)
  std::vector<double> M((size_t)n * 2 * n, 0.0);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) M[i * 2 * n + j] = A[i + n * j];
    M[i * 2 * n + n + i] = 1.0;
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(M[r * 2 * n + c]) > std::fabs(M[piv * 2 * n + c])) piv = r;
    if (M[piv * 2 * n + c] == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < 2 * n; ++j) std::swap(M[c * 2 * n + j], M[piv * 2 * n + j]);
    const double d = 1.0 / M[c * 2 * n + c];
    for (int j = 0; j < 2 * n; ++j) M[c * 2 * n + j] *= d;
    for (int r = 0; r < n; ++r)
      if (r != c) {
        const double f = M[r * 2 * n + c];
        if (f != 0.0)
          for (int j = 0; j < 2 * n; ++j) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Ainv[i + n * j] = M[i * 2 * n + n + j];
  return true;
}
}  // namespace mini

template <typename Traits>
bool BlockSolver<Traits>::buildStructure(bool) {
  const int p = PoseDim, l = LandmarkDim;
  _nP = _nL = 0;
  for (size_t i = 0; i < _optimizer->indexMapping().size(); ++i) (_optimizer->indexMapping()[i]->marginalized() && _doSchur ? _nL : _nP)++;
  _Hpp_diag.assign(_nP, PoseMatrixType());
  _Hll.assign(_nL, LandmarkMatrixType());
  _Hpp_off.clear();
  _obs.clear();
  _obsIndex.clear();
  _lmObs.assign(_nL, std::vector<int>());
  // the vertices' diagonal blocks are mapped into a mirror (computeLambdaInit reads them, levenberg.cpp:149-163)
  _mirror.assign((size_t)_nP * p * p + (size_t)_nL * l * l, 0.0);
  {
    size_t off = 0;
    int col = 0;
    for (size_t i = 0; i < _optimizer->indexMapping().size(); ++i) {
      OptimizableGraph::Vertex* v = _optimizer->indexMapping()[i];
      v->setColInHessian(col);
      v->mapHessianMemory(&_mirror[off]);
      col += v->dimension();
      off += (size_t)v->dimension() * v->dimension();
    }
  }
  for (size_t k = 0; k < _optimizer->activeEdges().size(); ++k) {
    OptimizableGraph::Edge* e = _optimizer->activeEdges()[k];
    for (size_t sa = 0; sa < e->vertices().size(); ++sa)   // every vertex pair of the edge (block_solver.hpp:206-254)
      for (size_t sb = sa + 1; sb < e->vertices().size(); ++sb) {
        OptimizableGraph::Vertex* a = static_cast<OptimizableGraph::Vertex*>(e->vertex((int)sa));
        OptimizableGraph::Vertex* b = static_cast<OptimizableGraph::Vertex*>(e->vertex((int)sb));
        const int ia = a->hessianIndex(), ib = b->hessianIndex();
        if (ia < 0 || ib < 0) continue;
        const bool la = ia >= _nP, lb = ib >= _nP;
        if (!la && !lb) {
          if (ia != ib) _Hpp_off[std::make_pair(std::min(ia, ib), std::max(ia, ib))] = PoseMatrixType();
        } else if (la != lb) {
          const int pose = la ? ib : ia, lm = (la ? ia : ib) - _nP;
          if (!_obsIndex.count(std::make_pair(pose, lm))) {
            _obsIndex[std::make_pair(pose, lm)] = (int)_obs.size();
            Obs o;
            o.pose = pose;
            o.lm = lm;
            _obs.push_back(o);
          }
        }
      }
  }
  for (size_t q = 0; q < _obs.size(); ++q) _lmObs[_obs[q].lm].push_back((int)q);
  for (int j = 0; j < _nL; ++j) std::sort(_lmObs[j].begin(), _lmObs[j].end(), [&](int x, int y) { return _obs[x].pose < _obs[y].pose; });
  // pattern of the reduced system: Hpp plus the pose pairs that share a landmark (block_solver.hpp:262-290)
  std::vector<int> bi(_nP);
  for (int i = 0; i < _nP; ++i) bi[i] = (i + 1) * p;
  delete _Hschur;
  _Hschur = new SparseBlockMatrix<PoseMatrixType>(_nP ? &bi[0] : 0, _nP ? &bi[0] : 0, _nP, _nP, true);
  for (int i = 0; i < _nP; ++i) _Hschur->block(i, i, true);
  for (typename std::map<std::pair<int, int>, PoseMatrixType>::iterator it = _Hpp_off.begin(); it != _Hpp_off.end(); ++it)
    _Hschur->block(it->first.first, it->first.second, true);
  for (int j = 0; j < _nL; ++j)
    for (size_t a = 0; a < _lmObs[j].size(); ++a)
      for (size_t b = a; b < _lmObs[j].size(); ++b) _Hschur->block(_obs[_lmObs[j][a]].pose, _obs[_lmObs[j][b]].pose, true);
  resizeVector((size_t)_nP * p + (size_t)_nL * l);
  return true;
}

template <typename Traits>
bool BlockSolver<Traits>::buildSystem() {
  const int p = PoseDim, l = LandmarkDim;
  for (int i = 0; i < _nP; ++i) _Hpp_diag[i].setZero();
  for (int j = 0; j < _nL; ++j) _Hll[j].setZero();
  for (typename std::map<std::pair<int, int>, PoseMatrixType>::iterator it = _Hpp_off.begin(); it != _Hpp_off.end(); ++it) it->second.setZero();
  for (size_t q = 0; q < _obs.size(); ++q) _obs[q].B.setZero();
  std::memset(_b, 0, sizeof(double) * _xSize);
  JacobianWorkspace& ws = _optimizer->jacobianWorkspace();
  std::vector<double> OJ;
  for (size_t k = 0; k < _optimizer->activeEdges().size(); ++k) {
    OptimizableGraph::Edge* e = _optimizer->activeEdges()[k];
    e->linearizeOplus(ws);                              // block_solver.hpp:531
    const int d = e->dimension();
    const double* err = e->errorData();
    const double* Om = e->informationData();
    double w = 1.0;                                     // robust weight rho' (base_edge.h:96-102, base_binary_edge.hpp:99)
    if (e->robustKernel()) {
      double rho[3];
      e->robustKernel()->robustify(e->chi2(), rho);
      w = rho[1];
    }
    const int nv = (int)e->vertices().size();
    int idx[8], dim[8];
    if (nv > 8) return false;
    for (int s = 0; s < nv; ++s) {
      OptimizableGraph::Vertex* v = static_cast<OptimizableGraph::Vertex*>(e->vertex(s));
      idx[s] = v->hessianIndex();
      dim[s] = v->dimension();
    }
    std::vector<double> oe(d, 0.0);                     // w Omega e
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) oe[i] += w * Om[i + d * j] * err[j];
    for (int s = 0; s < nv; ++s) {
      if (idx[s] < 0) continue;
      const double* J = ws.workspaceForVertex(s);       // d x dim[s], column-major
      const bool lm = idx[s] >= _nP;
      double* bseg = _b + (lm ? (size_t)_nP * p + (size_t)(idx[s] - _nP) * l : (size_t)idx[s] * p);
      for (int c = 0; c < dim[s]; ++c) {
        double t = 0.0;
        for (int i = 0; i < d; ++i) t += J[i + d * c] * oe[i];
        bseg[c] -= t;
      }
      OJ.assign((size_t)d * dim[s], 0.0);               // w Omega J
      for (int c = 0; c < dim[s]; ++c)
        for (int i = 0; i < d; ++i) {
          double t = 0.0;
          for (int j = 0; j < d; ++j) t += w * Om[i + d * j] * J[j + d * c];
          OJ[i + d * c] = t;
        }
      double* H = lm ? _Hll[idx[s] - _nP].data() : _Hpp_diag[idx[s]].data();
      mini::atb(J, d, dim[s], OJ.data(), dim[s], 1.0, H);
      for (int t2 = s + 1; t2 < nv; ++t2) {             // off-diagonal block of the pair (s, t2)
        if (idx[t2] < 0) continue;
        const double* J2 = ws.workspaceForVertex(t2);
        const bool lm2 = idx[t2] >= _nP;
        if (!lm && !lm2) {
          if (idx[s] == idx[t2]) continue;
          const bool swap = idx[s] > idx[t2];           // stored block (min, max) = J_min' W J_max
          PoseMatrixType& Hb = _Hpp_off[std::make_pair(std::min(idx[s], idx[t2]), std::max(idx[s], idx[t2]))];
          if (swap) {
            mini::atb(J2, d, p, OJ.data(), p, 1.0, Hb.data());   // J2' (W J)
          } else {
            std::vector<double> OJ2((size_t)d * p, 0.0);          // W J2
            for (int c = 0; c < p; ++c)
              for (int i = 0; i < d; ++i) {
                double t = 0.0;
                for (int j = 0; j < d; ++j) t += w * Om[i + d * j] * J2[j + d * c];
                OJ2[i + d * c] = t;
              }
            mini::atb(J, d, p, OJ2.data(), p, 1.0, Hb.data());
          }
        } else if (lm != lm2) {
          const double* Jp = lm ? J2 : J;               // Hpl(pose, landmark) = Jp' W Jl
          const double* Jl = lm ? J : J2;
          std::vector<double> OJl((size_t)d * l, 0.0);
          for (int c = 0; c < l; ++c)
            for (int i = 0; i < d; ++i) {
              double t = 0.0;
              for (int j = 0; j < d; ++j) t += w * Om[i + d * j] * Jl[j + d * c];
              OJl[i + d * c] = t;
            }
          const int pose = lm ? idx[t2] : idx[s], lmi = (lm ? idx[s] : idx[t2]) - _nP;
          mini::atb(Jp, d, p, OJl.data(), l, 1.0, _obs[_obsIndex[std::make_pair(pose, lmi)]].B.data());
        }
      }
    }
  }
  // mirror of the diagonal blocks for the vertices (what v->hessian(j, j) reads)
  {
    size_t off = 0;
    for (int i = 0; i < _nP; ++i, off += (size_t)p * p) std::memcpy(&_mirror[off], _Hpp_diag[i].data(), sizeof(double) * p * p);
    for (int j = 0; j < _nL; ++j, off += (size_t)l * l) std::memcpy(&_mirror[off], _Hll[j].data(), sizeof(double) * l * l);
  }
  return true;
}

template <typename Traits>
bool BlockSolver<Traits>::setLambda(double lambda, bool backup) {
  const int p = PoseDim, l = LandmarkDim;
  if (backup) {
    _diagBackup.clear();
    for (int i = 0; i < _nP; ++i) for (int c = 0; c < p; ++c) _diagBackup.push_back(_Hpp_diag[i](c, c));
    for (int j = 0; j < _nL; ++j) for (int c = 0; c < l; ++c) _diagBackup.push_back(_Hll[j](c, c));
  }
  for (int i = 0; i < _nP; ++i) for (int c = 0; c < p; ++c) _Hpp_diag[i](c, c) += lambda;
  for (int j = 0; j < _nL; ++j) for (int c = 0; c < l; ++c) _Hll[j](c, c) += lambda;
  return true;
}

template <typename Traits>
void BlockSolver<Traits>::restoreDiagonal() {
  const int p = PoseDim, l = LandmarkDim;
  size_t k = 0;
  for (int i = 0; i < _nP; ++i) for (int c = 0; c < p; ++c) _Hpp_diag[i](c, c) = _diagBackup[k++];
  for (int j = 0; j < _nL; ++j) for (int c = 0; c < l; ++c) _Hll[j](c, c) = _diagBackup[k++];
}

template <typename Traits>
bool BlockSolver<Traits>::solve() {
  const int p = PoseDim, l = LandmarkDim;
  // reduced system: Hschur = Hpp - sum_j B_j Dinv_j B_j', bschur = b_p - sum_j B_j Dinv_j b_lj
  _Hschur->clear();
  for (int i = 0; i < _nP; ++i) *_Hschur->block(i, i) = _Hpp_diag[i];
  for (typename std::map<std::pair<int, int>, PoseMatrixType>::iterator it = _Hpp_off.begin(); it != _Hpp_off.end(); ++it)
    *_Hschur->block(it->first.first, it->first.second) = it->second;
  std::vector<double> bs(_b, _b + (size_t)_nP * p);
  std::vector<LandmarkMatrixType> Dinv(_nL);
  for (int j = 0; j < _nL; ++j) {
    if (!mini::invert_spd(_Hll[j].data(), l, Dinv[j].data())) return false;
    const double* bl = _b + (size_t)_nP * p + (size_t)j * l;
    double db[LandmarkDim];
    for (int r = 0; r < l; ++r) {
      db[r] = 0.0;
      for (int c = 0; c < l; ++c) db[r] += Dinv[j](r, c) * bl[c];
    }
    for (size_t a = 0; a < _lmObs[j].size(); ++a) {
      const Obs& oa = _obs[_lmObs[j][a]];
      double BD[PoseDim * LandmarkDim];
      mini::ab(oa.B.data(), p, l, Dinv[j].data(), l, BD);
      for (int r = 0; r < p; ++r)
        for (int c = 0; c < l; ++c) bs[(size_t)oa.pose * p + r] -= oa.B(r, c) * db[c];
      for (size_t b2 = a; b2 < _lmObs[j].size(); ++b2) {
        const Obs& ob = _obs[_lmObs[j][b2]];
        PoseMatrixType* H = _Hschur->block(oa.pose, ob.pose);
        for (int r = 0; r < p; ++r)
          for (int c = 0; c < p; ++c) {
            double s = 0.0;
            for (int k = 0; k < l; ++k) s += BD[r + p * k] * ob.B(c, k);
            (*H)(r, c) -= s;
          }
      }
    }
  }
  if (_nP > 0 && !_linearSolver->solve(*_Hschur, _x, bs.data())) return false;
  // back-substitution: x_l = Dinv (b_l - B' x_p)
  for (int j = 0; j < _nL; ++j) {
    double cl[LandmarkDim];
    const double* bl = _b + (size_t)_nP * p + (size_t)j * l;
    for (int c = 0; c < l; ++c) cl[c] = bl[c];
    for (size_t a = 0; a < _lmObs[j].size(); ++a) {
      const Obs& oa = _obs[_lmObs[j][a]];
      for (int c = 0; c < l; ++c)
        for (int r = 0; r < p; ++r) cl[c] -= oa.B(r, c) * _x[(size_t)oa.pose * p + r];
    }
    double* xl = _x + (size_t)_nP * p + (size_t)j * l;
    for (int r = 0; r < l; ++r) {
      xl[r] = 0.0;
      for (int c = 0; c < l; ++c) xl[r] += Dinv[j](r, c) * cl[c];
    }
  }
  return true;
}

// block_solver.hpp:489-499: solvePattern on Hpp (NOT the Schur complement), on the same LinearSolver, without init()
template <typename Traits>
bool BlockSolver<Traits>::computeMarginals(SparseBlockMatrix<MatrixXd>& spinv, const std::vector<std::pair<int, int> >& blockIndices) {
  const int p = PoseDim;
  std::vector<int> bi(_nP);
  for (int i = 0; i < _nP; ++i) bi[i] = (i + 1) * p;
  SparseBlockMatrix<PoseMatrixType> Hpp(_nP ? &bi[0] : 0, _nP ? &bi[0] : 0, _nP, _nP, true);
  for (int i = 0; i < _nP; ++i) *Hpp.block(i, i, true) = _Hpp_diag[i];
  for (typename std::map<std::pair<int, int>, PoseMatrixType>::iterator it = _Hpp_off.begin(); it != _Hpp_off.end(); ++it)
    *Hpp.block(it->first.first, it->first.second, true) = it->second;
  return _linearSolver->solvePattern(spinv, blockIndices, Hpp);
}

template <typename Traits>
void BlockSolver<Traits>::multiplyHessian(double* dest, const double* src) const {
  const int p = PoseDim;
  for (int i = 0; i < _nP; ++i)
    for (int r = 0; r < p; ++r)
      for (int c = 0; c < p; ++c) dest[(size_t)i * p + r] += _Hpp_diag[i](r, c) * src[(size_t)i * p + c];
  for (typename std::map<std::pair<int, int>, PoseMatrixType>::const_iterator it = _Hpp_off.begin(); it != _Hpp_off.end(); ++it) {
    const int a = it->first.first, b = it->first.second;
    for (int r = 0; r < p; ++r)
      for (int c = 0; c < p; ++c) {
        dest[(size_t)a * p + r] += it->second(r, c) * src[(size_t)b * p + c];
        dest[(size_t)b * p + c] += it->second(r, c) * src[(size_t)a * p + r];
      }
  }
}

}  // namespace g2o
#endif
