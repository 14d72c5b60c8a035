// BlockSolver on MI355X: structure build (host, once) + the per-iteration kernels
// K1-K8, K13, K14 of SURVEY.md section 2.3.  See block_solver.h for the layout.
#include "block_solver.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>

namespace g2ohip {

// =====================================================================================
// Device helpers
// =====================================================================================
namespace {

constexpr int kThreads = 256;

// Sum over groups of G consecutive lanes (G = 1, 2, 4, 8, 16) with DPP lane permutations: no LDS
// round trip (ds_bpermute) per step.  Every lane of the group ends up with the total.
template <int CTRL>
__device__ __forceinline__ double dpp_permute(double v) {
  // (every control used here -- quad_perm, row_half_mirror, row_mirror, row_ror -- reads a valid lane on every lane: with
  // bound_ctrl the "old" operand is dead and the compiler drops the two v_mov that zeroed it in front of every pair of DPP moves;
  // the destination loop's epilogue of the Schur tiles was 168 moves for 42 additions per destination)
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi2, lo2);
}
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if (G >= 2) v += dpp_permute<0xB1>(v);    // quad_perm [1,0,3,2]
  if (G >= 4) v += dpp_permute<0x4E>(v);    // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_permute<0x141>(v);   // row_half_mirror: lane i <-> 7-i inside each 8-lane half row
  if (G >= 16) v += dpp_permute<0x140>(v);  // row_mirror: lane i <-> 15-i inside each 16-lane row
  return v;
}

// N doubles through 16-byte global loads (addresses are 8-byte aligned at least; gfx950 global loads
// only need dword alignment)
typedef double dbl2_u __attribute__((ext_vector_type(2), aligned(8)));
template <int N>
__device__ __forceinline__ void load_vec(const double* __restrict__ p, double (&v)[N]) {
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) {
    const dbl2_u t = *reinterpret_cast<const dbl2_u*>(p + i);
    v[i] = t.x;
    v[i + 1] = t.y;
  }
  if (N & 1) v[N - 1] = p[N - 1];
}
template <int N>
__device__ __forceinline__ void store_vec(double* __restrict__ p, const double (&v)[N]) {
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) {
    dbl2_u t;
    t.x = v[i];
    t.y = v[i + 1];
    *reinterpret_cast<dbl2_u*>(p + i) = t;
  }
  if (N & 1) p[N - 1] = v[N - 1];
}
// N doubles out of LDS with 16-byte reads when the block size keeps every block 16-byte aligned
typedef double dbl2_a __attribute__((ext_vector_type(2), aligned(16)));
template <int N>
__device__ __forceinline__ void lds_block(const double* p, double (&v)[N]) {
  if (N % 2 == 0) {
#pragma unroll
    for (int i = 0; i < N; i += 2) {
      const dbl2_a t = *reinterpret_cast<const dbl2_a*>(p + i);
      v[i] = t.x;
      v[i + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = p[i];
  }
}
// blockIdx -> logical block so that consecutive logical blocks share an XCD (and its L2):
// the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md), speed only.
__device__ __forceinline__ int xcd_swizzle(int b, int nblocks) {
  const int per = nblocks >> 3;
  return (b < (per << 3)) ? (b & 7) * per + (b >> 3) : b;
}

// rho'(e2) and rho(e2) of the robust kernels of g2o/core/robust_kernel_impl.cpp (the weight base_binary_edge.hpp:92-112
// puts on the information matrix, and the value activeRobustChi2 sums): 1 Huber (:65-78), 2 PseudoHuber (:80-89),
// 3 Cauchy (:91-99), 4 Saturated (:101-113), 5 DCS (:116-126, delta is phi); 0 = none
__device__ __forceinline__ double robust_weight(int kind, double delta, double e2) {
  const double dsqr = delta * delta;
  switch (kind) {
    case 1: return (e2 <= dsqr) ? 1.0 : delta / sqrt(e2);
    case 2: return 1.0 / sqrt(e2 / dsqr + 1.0);
    case 3: return 1.0 / (e2 / dsqr + 1.0);
    case 4: return (e2 <= dsqr) ? 1.0 : 0.0;
    case 5: {
      double sc = (2.0 * delta) / (delta + e2);
      if (sc >= 1.0) sc = 1.0;
      return sc * sc;
    }
    default: return 1.0;
  }
}
__device__ __forceinline__ double robust_rho(int kind, double delta, double e2) {
  const double dsqr = delta * delta;
  switch (kind) {
    case 1: return (e2 <= dsqr) ? e2 : 2.0 * sqrt(e2) * delta - dsqr;
    case 2: return 2.0 * dsqr * (sqrt(e2 / dsqr + 1.0) - 1.0);
    case 3: return dsqr * log(e2 / dsqr + 1.0);
    case 4: return (e2 <= dsqr) ? e2 : dsqr;
    case 5: {
      double sc = (2.0 * delta) / (delta + e2);
      if (sc >= 1.0) sc = 1.0;
      return sc * e2 * sc;
    }
    default: return e2;
  }
}

// Edge classes of the BA front end (g2ohip_ba_set_edges_classes): observations that differ in camera intrinsics
// (EdgeProjectXYZ2UV::_cam, g2o/types/sba/types_six_dof_expmap.h:133-153) or in their robust kernel
// (OptimizableGraph::Edge::robustKernel, optimizable_graph.h:436-443) share ONE edge set; the class of an observation
// rides in the top byte of its camera index in every device copy, ctab[5 c] = (f, cx, cy, kernel kind, delta).  CLS = false
// (one class, the common case): the five values stay kernel arguments (scalar registers), nothing is decoded.
template <bool CLS>
__device__ __forceinline__ int ba_edge_class(int camword, const double* __restrict__ ctab, double& f, double& cx, double& cy, int& kind,
                                             double& delta) {
  if constexpr (CLS) {
    const double* t = ctab + 5 * (size_t)((unsigned)camword >> 24);
    f = t[0]; cx = t[1]; cy = t[2]; kind = (int)t[3]; delta = t[4];
    return camword & 0xffffff;
  } else {
    return camword;
  }
}

// ---------------------------------------------------------------------------------
// K2, vertex part: H_vv (+)= sum_e J' (rho' Omega) J ;  b_v (+)= sum_e J' (-rho' Omega e)
// base_binary_edge.hpp:54-120 / base_unary_edge.hpp:42-72, destination-major: G lanes share
// one vertex and walk its contributor list (edge << 1 | side), then a butterfly reduction.
// ---------------------------------------------------------------------------------
template <int D, int DV, int G>
__global__ void __launch_bounds__(kThreads) assemble_vertex_kernel(int nV, const int* __restrict__ vptr, const int* __restrict__ vent,
                                       const double* __restrict__ J0, const double* __restrict__ J1,
                                       const double* __restrict__ omega, const double* __restrict__ err, int kind, double delta,
                                       double* __restrict__ H, const int* __restrict__ diag_blk, double* __restrict__ b,
                                       int accumulate, const double* __restrict__ rk) {
  // rk != nullptr: every edge of the set brings its own robust kernel, rk[2 e] = kind, rk[2 e + 1] = delta (set_robust_kernel_per_edge)
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = gt / G, g = gt % G;
  const bool active = v < nV;
  double Hacc[DV * DV];
  double bacc[DV];
#pragma unroll
  for (int i = 0; i < DV * DV; ++i) Hacc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < DV; ++i) bacc[i] = 0.0;
  const int k0 = active ? vptr[v] : 0, k1 = active ? vptr[v + 1] : 0;
  for (int k = k0 + g; k < k1; k += G) {
    const int ent = vent[k];
    const size_t e = (size_t)(ent >> 1);
    const double* Jp = ((ent & 1) ? J1 : J0) + e * (D * DV);
    const double* Op = omega + e * (D * D);
    const double* rp = err + e * D;
    double J[D * DV], O[D * D], r[D], Or[D];
    load_vec<D * DV>(Jp, J);
    load_vec<D * D>(Op, O);
    load_vec<D>(rp, r);
    double e2 = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < D; ++j) t += O[i + D * j] * r[j];
      Or[i] = t;
      e2 += r[i] * t;
    }
    const double w = rk ? robust_weight((int)rk[2 * e], rk[2 * e + 1], e2) : robust_weight(kind, delta, e2);
    // b += J' (-w O r)
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) t += J[i + D * c] * Or[i];
      bacc[c] -= w * t;
    }
    // H += J' (w O) J, column by column
#pragma unroll
    for (int c = 0; c < DV; ++c) {
      double OJ[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) t += O[i + D * j] * J[j + D * c];
        OJ[i] = w * t;
      }
#pragma unroll
      for (int a = 0; a < DV; ++a) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) t += J[i + D * a] * OJ[i];
        Hacc[a + DV * c] += t;
      }
    }
  }
  if (G > 1) {
#pragma unroll
    for (int i = 0; i < DV * DV; ++i) Hacc[i] = group_sum<G>(Hacc[i]);
#pragma unroll
    for (int i = 0; i < DV; ++i) bacc[i] = group_sum<G>(bacc[i]);
  }
  if (!active) return;
  const size_t blk = diag_blk ? (size_t)diag_blk[v] : (size_t)v;
  double* Hd = H + blk * (DV * DV);
  double* bd = b + (size_t)v * DV;
#pragma unroll
  for (int i = 0; i < DV * DV; ++i)
    if (G == 1 || (i % G) == g) Hd[i] = accumulate ? Hd[i] + Hacc[i] : Hacc[i];
#pragma unroll
  for (int i = 0; i < DV; ++i)
    if (G == 1 || (i % G) == g) bd[i] = accumulate ? bd[i] + bacc[i] : bacc[i];
}

// ---------------------------------------------------------------------------------
// K2, off-diagonal part: H_ij (+)= A' (rho' Omega) B, or B' (rho' Omega) A when the block is
// stored transposed (block_solver.hpp:221-250).  One thread per destination block.
// ---------------------------------------------------------------------------------
template <int D, int DR, int DC>
__global__ void __launch_bounds__(kThreads) assemble_offdiag_kernel(int nDst, const int* __restrict__ dst, const int* __restrict__ ptr,
                                        const int* __restrict__ ent, const double* __restrict__ J0, const double* __restrict__ J1,
                                        const double* __restrict__ omega, const double* __restrict__ err, int kind, double delta,
                                        double* __restrict__ H, int accumulate, const double* __restrict__ rk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nDst) return;
  double acc[DR * DC];
#pragma unroll
  for (int i = 0; i < DR * DC; ++i) acc[i] = 0.0;
  for (int k = ptr[t]; k < ptr[t + 1]; ++k) {
    const int en = ent[k];
    const size_t e = (size_t)(en >> 1);
    const bool tr = en & 1;
    const double* Lp = tr ? (J1 + e * (D * DR)) : (J0 + e * (D * DR));
    const double* Rp = tr ? (J0 + e * (D * DC)) : (J1 + e * (D * DC));
    const double* Op = omega + e * (D * D);
    double O[D * D], Lm[D * DR], Rm[D * DC];
    load_vec<D * D>(Op, O);
    load_vec<D * DR>(Lp, Lm);
    load_vec<D * DC>(Rp, Rm);
    double w = 1.0;
    if (rk) {   // (per-edge kernels)
      kind = (int)rk[2 * e];
      delta = rk[2 * e + 1];
    }
    if (kind != 0) {
      const double* rp = err + e * D;
      double e2 = 0.0;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) s += O[i + D * j] * rp[j];
        e2 += rp[i] * s;
      }
      w = robust_weight(kind, delta, e2);
    }
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      double OR[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) s += O[i + D * j] * Rm[j + D * c];
        OR[i] = w * s;
      }
#pragma unroll
      for (int a = 0; a < DR; ++a) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) s += Lm[i + D * a] * OR[i];
        acc[a + DR * c] += s;
      }
    }
  }
  double* Hd = H + (size_t)dst[t] * (DR * DC);
  if (accumulate) {
    double old[DR * DC];
    load_vec<DR * DC>(Hd, old);
#pragma unroll
    for (int i = 0; i < DR * DC; ++i) acc[i] += old[i];
  }
  store_vec<DR * DC>(Hd, acc);
}

// chi2 partial sums: sum_e rho(e' Omega e)  (sparse_optimizer.cpp:100-114)
template <int D>
__global__ void __launch_bounds__(kThreads) chi2_kernel(int n, const double* __restrict__ omega, const double* __restrict__ err, int kind, double delta,
                            double* __restrict__ partial, const double* __restrict__ rk) {
  __shared__ double sh[kThreads];
  double s = 0.0;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)n; e += (size_t)gridDim.x * blockDim.x) {
    const double* O = omega + e * (D * D);
    const double* r = err + e * D;
    double e2 = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < D; ++j) t += O[i + D * j] * r[j];
      e2 += r[i] * t;
    }
    s += rk ? robust_rho((int)rk[2 * e], rk[2 * e + 1], e2) : robust_rho(kind, delta, e2);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ void __launch_bounds__(kThreads) scale_partial_kernel(size_t n, const double* __restrict__ x, const double* __restrict__ b, double lambda,
                                     double* __restrict__ partial) {
  __shared__ double sh[kThreads];
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += x[i] * (lambda * x[i] + b[i]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// max |diag| partials over nV blocks of dimension DV
__global__ void __launch_bounds__(kThreads) maxdiag_partial_kernel(int nV, int dv, const double* __restrict__ H, const int* __restrict__ diag_blk,
                                       double* __restrict__ partial) {
  __shared__ double sh[kThreads];
  double s = 0.0;
  const size_t total = (size_t)nV * dv;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t v = i / dv, j = i % dv;
    const size_t blk = diag_blk ? (size_t)diag_blk[v] : v;
    s = fmax(s, fabs(H[blk * dv * dv + j + dv * j]));
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// [n] 2 x 2 identity matrices (EdgeProjectXYZ2UV sets whose information matrices were declared the identity)
__global__ void __launch_bounds__(kThreads) identity2_kernel(size_t n, double* __restrict__ om) {
  const size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k >= n) return;
  dbl2_u a, b;
  a.x = 1.0; a.y = 0.0;
  b.x = 0.0; b.y = 1.0;
  reinterpret_cast<dbl2_u*>(om + 4 * k)[0] = a;
  reinterpret_cast<dbl2_u*>(om + 4 * k)[1] = b;
}

// K4: setLambda / restoreDiagonal (block_solver.hpp:563-604)
__global__ void __launch_bounds__(kThreads) lambda_kernel(int nV, int dv, double* __restrict__ H, const int* __restrict__ diag_blk, double* __restrict__ backup,
                              double lambda, int do_backup, int restore, const unsigned char* __restrict__ mask = nullptr) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)nV * dv) return;
  const size_t v = i / dv, j = i % dv;
  const size_t blk = diag_blk ? (size_t)diag_blk[v] : v;
  double* d = H + blk * dv * dv + j + dv * j;
  if (restore) {
    *d = backup[i];
    return;
  }
  if (do_backup) backup[i] = *d;
  if (mask && !mask[v]) return;   // multi-GPU: exactly one rank damps each pose block (the sum is formed later)
  *d += lambda;
}

// K6: Dinv = D^-1 (closed cofactor form like Eigen's fixed-size inverse, block_solver.hpp:389),
// db = Dinv * b_l (:391-395)
template <int LD>
__device__ __forceinline__ void small_inverse(const double* D, double* R) {
  if (LD == 1) {
    R[0] = 1.0 / D[0];
  } else if (LD == 2) {
    const double det = D[0] * D[3] - D[2] * D[1];
    const double id = 1.0 / det;
    R[0] = D[3] * id;
    R[1] = -D[1] * id;
    R[2] = -D[2] * id;
    R[3] = D[0] * id;
  } else if (LD == 3) {
#define M_(i, j) D[(i) + 3 * (j)]
    const double c00 = M_(1, 1) * M_(2, 2) - M_(1, 2) * M_(2, 1);
    const double c10 = M_(1, 2) * M_(2, 0) - M_(1, 0) * M_(2, 2);
    const double c20 = M_(1, 0) * M_(2, 1) - M_(1, 1) * M_(2, 0);
    const double det = M_(0, 0) * c00 + M_(0, 1) * c10 + M_(0, 2) * c20;
    const double id = 1.0 / det;
    R[0] = c00 * id;
    R[1] = c10 * id;
    R[2] = c20 * id;
    R[3] = (M_(0, 2) * M_(2, 1) - M_(0, 1) * M_(2, 2)) * id;
    R[4] = (M_(0, 0) * M_(2, 2) - M_(0, 2) * M_(2, 0)) * id;
    R[5] = (M_(0, 1) * M_(2, 0) - M_(0, 0) * M_(2, 1)) * id;
    R[6] = (M_(0, 1) * M_(1, 2) - M_(0, 2) * M_(1, 1)) * id;
    R[7] = (M_(0, 2) * M_(1, 0) - M_(0, 0) * M_(1, 2)) * id;
    R[8] = (M_(0, 0) * M_(1, 1) - M_(0, 1) * M_(1, 0)) * id;
#undef M_
  }
}
template <int LD>
__global__ void __launch_bounds__(kThreads) landmark_inverse_kernel(int nL, const double* __restrict__ Hll, const double* __restrict__ bl,
                                        double* __restrict__ Dinv, double* __restrict__ db, const double* __restrict__ lam) {
  const int lm = blockIdx.x * blockDim.x + threadIdx.x;
  if (lm >= nL) return;
  const double lambda = lam[1];   // virtual damping of the landmark blocks (set_lambda does not touch Hll)
  double D[LD * LD], R[LD * LD];
#pragma unroll
  for (int i = 0; i < LD * LD; ++i) D[i] = Hll[(size_t)lm * LD * LD + i];
#pragma unroll
  for (int i = 0; i < LD; ++i) D[i * (LD + 1)] += lambda;
  small_inverse<LD>(D, R);
#pragma unroll
  for (int i = 0; i < LD * LD; ++i) Dinv[(size_t)lm * LD * LD + i] = R[i];
  (void)bl;
  (void)db;
}

// K5+K7+K8, pass 1: Schur outer products over a TILE of consecutive landmarks
//   partial(i1,i2) = sum_{lm in tile} (B_i1 Dinv) B_i2'     (block_solver.hpp:400-431)
//   partial_rhs(i) = sum_{lm in tile} B_i Dinv b_lm         (block_solver.hpp:412)
// A tile's Hpl columns are contiguous in HBM: they are staged ONCE into LDS with coalesced 16-byte
// loads (Hpl is read once per iteration instead of once per pose pair), together with Dinv, b_l and the
// tile's contributor entries.  Each destination block touched by the tile is owned by G lanes that walk
// its entry list out of LDS (no atomics, fixed order) and write one partial block to HBM.
// Shared by the two tile kernels: (optional) landmark inversion on the staged blocks, the symmetric split of Dinv, then the
// destination loop.
//
// Symmetric split.  Every entry of a destination is (B_s1 Dinv) B_s2' with the two Hpl blocks of ONE landmark, so with
// Dinv = C Sg C' (C lower triangular, Sg = diag(+-1): the L D L' factors of the 3x3 block, |D|^(1/2) folded into C) and
//   V_s = B_s C        (each staged block is transformed ONCE, in place)
// an entry is V_s1 Sg V_s2' and the right-hand side term is V_s (Sg C' b_l).  The destination loop then needs no product with
// Dinv per entry: PD*LD fewer multiply-adds per lane and entry (a third of them) and LD*LD fewer LDS reads, at no extra LDS
// for the blocks.  Sg is the identity unless the damped landmark block is not positive definite (negative lambda,
// block_solver.hpp:563-604 allows it): a per-tile flag makes the loop apply the signs then.
template <int LD>
struct TileSplit {
  static constexpr int CP = (LD * LD + LD + 1) & ~1;   // doubles per landmark: C (column major, lower part used) | signs
};
// Where element (row a, column kk) of a transformed block V = B C sits inside its LDS slot of PD * LD doubles.  The destination
// loop reads the rows of ONE half (GC = 2: rows 0-2 or 3-5) of V_s1 per lane and all of V_s2.  With the plain column-major
// layout a half is three 24-byte runs at a 48-byte stride -- four ds_read2_b64 and a ds_read_b64, 34 LDS-array cycles per
// wave-instruction group, as many as the 144 bytes of V_s2 (nine ds_read_b128, 36) -- and the loop keeps the LDS array busier
// than the vector units.  For 6 x 3 blocks split in two the halves are stored one after the other, [rr + 3 kk] each, the second
// one rotated by one double so that eight of a half's nine doubles are 16-byte aligned in BOTH halves: four ds_read_b128 and
// one ds_read_b64 (18 cycles) with the same register assignment on both kinds of lanes.
template <int PD, int LD, int GC>
struct VSlot {
  static constexpr bool PERM = PD == 6 && LD == 3 && GC == 2;
  __host__ __device__ static constexpr int off(int a, int kk) {
    return !PERM ? a + PD * kk : (a < 3 ? a + 3 * kk : ((a - 3) + 3 * kk == 8 ? 9 : 10 + (a - 3) + 3 * kk));
  }
};
// R (symmetric, its lower triangle is read) = Lm diag(d) Lm' -> C = Lm |d|^(1/2) (column major, zeros above the diagonal),
// sg = sign(d), u = Sg C' b
// sqrt(a) and 1 / sqrt(a), a > 0: v_rsq_f64 seed, two Goldschmidt steps, one Newton correction (the recipe of the f64 sqrt
// expansion, which then also has the reciprocal root: ~1-2 ulp each).  A pivot of the split needs sqrt |d| and 1 / d = sign(d) /
// sqrt|d|^2: one of these instead of a division (~25 instructions) and a square root (~20) per pivot -- the list heads of the
// tile kernel run this with a fifth of their lanes active, the whole wave pays
__device__ __forceinline__ void bs_sqrt_and_rsqrt(double a, double& s, double& r) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double dd = fma(-g, g, a);
  g = fma(dd, h, g);
  s = g;
  r = h + h;
}
template <int LD>
__device__ __forceinline__ bool landmark_split(const double* R, const double* b, double* C, double* sg, double* u) {
  double Lm[LD * LD], d[LD], sqd[LD];
#pragma unroll
  for (int c = 0; c < LD; ++c) {
    double dc = R[c + LD * c];
#pragma unroll
    for (int k = 0; k < c; ++k) dc -= Lm[c + LD * k] * Lm[c + LD * k] * d[k];
    d[c] = dc;
    double rsd;
    bs_sqrt_and_rsqrt(fabs(dc), sqd[c], rsd);
    const double idc = copysign(rsd * rsd, dc);
#pragma unroll
    for (int r = c + 1; r < LD; ++r) {
      double v = R[r + LD * c];
#pragma unroll
      for (int k = 0; k < c; ++k) v -= Lm[r + LD * k] * Lm[c + LD * k] * d[k];
      Lm[r + LD * c] = v * idc;
    }
    Lm[c + LD * c] = 1.0;
  }
  bool neg = false;
#pragma unroll
  for (int c = 0; c < LD; ++c) {
    const double sq = sqd[c];
    sg[c] = d[c] < 0.0 ? -1.0 : 1.0;
    neg = neg || d[c] < 0.0;
    double uu = 0.0;
#pragma unroll
    for (int r = 0; r < LD; ++r) {
      const double cv = r >= c ? Lm[r + LD * c] * sq : 0.0;
      C[r + LD * c] = cv;
      if (r >= c) uu += cv * b[r];
    }
    u[c] = sg[c] * uu;
  }
  return neg;
}
// V = B C in place (C lower triangular)
template <int PD, int LD, int GC>
__device__ __forceinline__ void block_times_split(double* Bslot, const double* C) {
  double B[PD * LD];
  lds_block<PD * LD>(Bslot, B);
#pragma unroll
  for (int c = 0; c < LD; ++c)
#pragma unroll
    for (int r = 0; r < PD; ++r) {
      double v = B[r + PD * c] * C[c + LD * c];
#pragma unroll
      for (int k = c + 1; k < LD; ++k) v = fma(B[r + PD * k], C[k + LD * c], v);
      Bslot[VSlot<PD, LD, GC>::off(r, c)] = v;   // (every element of B is in registers by now)
    }
}
__device__ __forceinline__ bool tile_flag(const int* f) {
  int v = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) v |= f[w];
  return v != 0;
}
template <int PD, int LD, int GC>
__device__ __forceinline__ void schur_tile_prepare(double* Bs, double* Ds, double* bsm, double* Cs, int* neg_flag, int l0, int nlm, int ntot,
                                                   int slot_lm_first, const unsigned short* __restrict__ slot_lm, int q0, int nslots,
                                                   double* __restrict__ Dinv, const double* __restrict__ Hll,
                                                   const double* __restrict__ lam, int tid, int NT) {
  constexpr int CP = TileSplit<LD>::CP;
  const double lambda = Hll ? lam[1] : 0.0;
  for (int j = tid; j < nlm; j += NT) {
    double D[LD * LD], R[LD * LD];
#pragma unroll
    for (int i = 0; i < LD * LD; ++i) D[i] = Ds[j * (LD * LD) + i];
    if (Hll) {
#pragma unroll
      for (int i = 0; i < LD; ++i) D[i * (LD + 1)] += lambda;
      small_inverse<LD>(D, R);
#pragma unroll
      for (int i = 0; i < LD * LD; ++i) Ds[j * (LD * LD) + i] = R[i];
    } else {
#pragma unroll
      for (int i = 0; i < LD * LD; ++i) R[i] = D[i];
    }
    double C[LD * LD], sg[LD], u[LD], b[LD];
#pragma unroll
    for (int c = 0; c < LD; ++c) b[c] = bsm[j * LD + c];
    const bool neg = landmark_split<LD>(R, b, C, sg, u);
#pragma unroll
    for (int i = 0; i < LD * LD; ++i) Cs[j * CP + i] = C[i];
#pragma unroll
    for (int c = 0; c < LD; ++c) {
      Cs[j * CP + LD * LD + c] = sg[c];
      bsm[j * LD + c] = u[c];
    }
    if (neg) *neg_flag = 1;
  }
  __syncthreads();
  if (Hll) {
    double* dstD = Dinv + (size_t)l0 * LD * LD;
    for (int i = tid; i < nlm * LD * LD; i += NT) dstD[i] = Ds[i];
  }
  // V_s = B_s C, one lane per staged block
  for (int sb = 0; sb < ntot; sb += NT) {
    const int s_ = sb + tid;
    if (s_ < ntot) {
      int lm = 0;
      if (nlm > 1) lm = sb == 0 ? slot_lm_first : (int)slot_lm[q0 + s_];   // (a tile with a second block range holds one landmark)
      double C[LD * LD];
#pragma unroll
      for (int i = 0; i < LD * LD; ++i) C[i] = Cs[lm * CP + i];
      block_times_split<PD, LD, GC>(Bs + s_ * (PD * LD), C);
    }
  }
  __syncthreads();
  (void)nslots;
}

// the destination loop (no barrier inside)
template <int PD, int LD, int G, class EL>
__device__ __forceinline__ void schur_tile_dests(const double* Bs, const double* Cs, const double* bsm, bool neg, const int* ep,
                                                 const int* dptr, const int* ddiag, const EL* el, int td0, int td1, int e0,
                                                 double* __restrict__ Pd, double* __restrict__ Pr, int tid, int NT) {
  constexpr int PL = PD * LD;
  // G lanes per destination block = GC row parts x GE entry parts: a lane owns NR = PD/GC ROWS of the block
  // (PD*PD/GC accumulator registers: what keeps 3 workgroups on a CU) and walks every GE-th entry.  With the row
  // split a lane only needs its own rows of V_s1: the kernel is bound by VALU issue.  The GE partial sums are combined with DPP
  // (fixed order: deterministic).  Partial block layout in Pd: [row part][column][row inside the part].
  constexpr int GC = (PD % 2 == 0 && G >= 2) ? 2 : 1, GE = G / GC, NR = PD / GC;
  static_assert(GE == 1 || GE == 2 || GE == 4 || GE == 8 || (GC == 1 && GE == 16), "unsupported lane group");
  // GE <= 4 with the row split: the entry parts are the two LOW lane bits (gc is bit 2), so both steps of the sum are quad
  // permutations -- one DPP move per half register instead of two for the lane ^ 4 step
  constexpr bool QUAD = GC == 2 && (GE == 2 || GE == 4);
  auto ge_sum = [](double v) {   // sum over the GE lanes that share gc
    if (GC == 1) return group_sum<GE>(v);
    if (QUAD) {
      v += dpp_permute<0xB1>(v);                                     // lane ^ 1
      if (GE == 4) v += dpp_permute<0x4E>(v);                        // lane ^ 2
      return v;
    }
    v += dpp_permute<0x4E>(v);                                       // lane ^ 2
    if (GE >= 4) v += dpp_permute<0x141>(dpp_permute<0x1B>(v));      // lane ^ 4 = (lane ^ 3) ^ 7
    if (GE >= 8) v += dpp_permute<0x128>(v);                         // row_ror:8 = lane ^ 8 inside a 16-lane row
    return v;
  };
  // G = 8 (two row halves x four entry parts): the two quads of a destination are the lanes l and l ^ 12 -- the hardware serves a
  // wave's ds_read_b128 in four groups of sixteen lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS), and
  // both halves of a destination read the SAME V_s2: in one group the identical addresses are served once, so a group holds 8 distinct
  // 16-byte pieces instead of 16 and the bank conflicts between the (arbitrary) slots go down
  constexpr bool PAIR12 = QUAD && GE == 4;
  const int quad = (tid >> 2) & 3;
  const int grp = PAIR12 ? (tid >> 4) * 2 + ((quad ^ (quad >> 1)) & 1) : tid / G, g = tid % G, ngroups = NT / G;
  const int gc = PAIR12 ? quad >> 1 : (QUAD ? g / GE : g % GC), ge = PAIR12 ? (tid & 3) : (QUAD ? g % GE : g / GC);
  for (int ld = td0 + grp; ld < td1; ld += ngroups) {
    double acc[NR * PD], cacc[NR];   // acc[rr + NR * c]: row gc*NR + rr, column c
#pragma unroll
    for (int i = 0; i < NR * PD; ++i) acc[i] = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) cacc[r] = 0.0;
    const bool diag = ddiag[ld - td0] != 0;
    const int k0 = dptr[ld - td0] - e0, k1 = dptr[ld - td0 + 1] - e0;
    // (the entry word of the NEXT iteration is requested one iteration ahead: entry -> slot addresses -> blocks is a chain of two
    // LDS round trips per iteration otherwise; reading past the list ends inside the tile's own LDS tables and is never used)
    int pk_next = ep[k0 + ge];
    for (int k = k0 + ge; k < k1; k += GE) {
      const int pk = pk_next, lm = el[k];
      pk_next = ep[k + GE];
      const int s1 = pk & 0xffff, s2 = (pk >> 16) & 0xffff;
      double W[NR * LD];   // W[rr + NR * c] = V_s1(gc*NR + rr, c)   (V = B C: schur_tile_prepare)
      typedef VSlot<PD, LD, GC> VS;
      if constexpr (VS::PERM) {   // this lane's half: eight doubles 16-byte aligned, the ninth next to them (VSlot)
        const double* sp = Bs + s1 * PL;
        double h8[8];
        lds_block<8>(sp + (gc ? 10 : 0), h8);
#pragma unroll
        for (int i = 0; i < 8; ++i) W[i] = h8[i];
        W[8] = sp[gc ? 9 : 8];
      } else {
        const double* B1p = Bs + s1 * PL + gc * NR;
#pragma unroll
        for (int kk = 0; kk < LD; ++kk)
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) W[rr + NR * kk] = B1p[rr + PD * kk];
      }
      if (diag) {   // s1 == s2: rhs contribution B (Dinv b_l) = V (Sg C' b_l), this lane's rows
#pragma unroll
        for (int c = 0; c < LD; ++c) {
          const double bv = bsm[lm * LD + c];
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) cacc[rr] += W[rr + NR * c] * bv;
        }
      }
      if (neg) {   // (tile with a landmark block that is not positive definite: V_s1 Sg)
        constexpr int CP = TileSplit<LD>::CP;
#pragma unroll
        for (int c = 0; c < LD; ++c) {
          const double sv = Cs[lm * CP + LD * LD + c];
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) W[rr + NR * c] *= sv;
        }
      }
      // this lane's rows of W * B_s2'
      double B2[PL];
      lds_block<PL>(Bs + s2 * PL, B2);
      // (accumulated term by term: three fused multiply-adds per element and entry, no separate add -- the kernel is bound by
      // instruction issue)
#pragma unroll
      for (int c = 0; c < PD; ++c)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
          double v = acc[rr + NR * c];
#pragma unroll
          for (int kk = 0; kk < LD; ++kk) v = fma(W[rr + NR * kk], B2[VS::off(c, kk)], v);
          acc[rr + NR * c] = v;
        }
    }
    if (GE > 1) {   // lanes with the same gc are GC apart
#pragma unroll
      for (int i = 0; i < NR * PD; ++i) acc[i] = ge_sum(acc[i]);
#pragma unroll
      for (int r = 0; r < NR; ++r) cacc[r] = ge_sum(cacc[r]);
    }
    double* out = Pd + (size_t)ld * PD * PD + gc * NR * PD;
    if constexpr ((NR * PD) % 2 == 0) {   // 16-byte pieces, dealt round-robin to the GE lanes (all hold the sum)
      dbl2_u* out2 = reinterpret_cast<dbl2_u*>(out);
#pragma unroll
      for (int u = 0; u < NR * PD / 2; ++u)
        if (GE == 1 || (u % GE) == ge) {
          dbl2_u v;
          v.x = acc[2 * u];
          v.y = acc[2 * u + 1];
          out2[u] = v;
        }
    } else {
#pragma unroll
      for (int i = 0; i < NR * PD; ++i)
        if (GE == 1 || (i % GE) == ge) out[i] = acc[i];
    }
    if (diag) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
        if (GE == 1 || (r % GE) == ge) Pr[(size_t)ld * PD + gc * NR + r] = cacc[r];
    }
  }
}

template <int PD, int LD, int G>
#ifndef G2OHIP_SCHUR_OCC
#define G2OHIP_SCHUR_OCC 4   // 110 VGPRs after the row split: 4 workgroups of 39 KB tiles per CU
#endif
__global__ void __launch_bounds__(kThreads, G2OHIP_SCHUR_OCC) schur_tile_kernel(const int* __restrict__ tile_lm0, const int* __restrict__ tile_td0,
                                                            const int* __restrict__ pl_colptr, const double* __restrict__ Hpl,
                                                            double* __restrict__ Dinv, const double* __restrict__ bl,
                                                            const int* __restrict__ td_diag, const int* __restrict__ td_ptr,
                                                            const int* __restrict__ te_pack, const unsigned short* __restrict__ te_lm,
                                                            double* __restrict__ Pd,
                                                            double* __restrict__ Pr, const double* __restrict__ Hll,
                                                            const double* __restrict__ lam, const int2* __restrict__ tile_q2,
                                                            const unsigned short* __restrict__ slot_lm) {
  // Hll != nullptr: the landmark inversion (block_solver.hpp:386-389, with the virtual damping) is done here on
  // the staged blocks -- the tile reads Hll instead of Dinv and writes Dinv (back-substitution needs it) on the way.
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int PL = PD * LD;
  const int t = xcd_swizzle(blockIdx.x, gridDim.x);
  const int* tm = tile_lm0 + (size_t)t * 8;   // packed tile record (see build_structure)
  const int l0 = tm[0], l1 = tm[1], q0 = tm[2], nslots = tm[3], nlm = l1 - l0;
  const int td0 = tm[4], td1 = tm[5], e0 = tm[6], ne = tm[7];
  const int2 t2 = tile_q2[t];   // second block range (tiles of a split landmark; count 0 otherwise), staged behind the first
  double* Bs = smem;
  double* Ds = Bs + (((nslots + t2.y) * PL + 1) & ~1);
  double* bsm = Ds + ((nlm * LD * LD + 1) & ~1);
  double* Cs = bsm + ((nlm * LD + 1) & ~1);           // symmetric split of Dinv (schur_tile_prepare)
  int* ep = reinterpret_cast<int*>(Cs + nlm * TileSplit<LD>::CP);
  int* dptr = ep + ne;                                // td_ptr[td0 .. td1]
  int* ddiag = dptr + (td1 - td0 + 1);                // destination is a diagonal block?
  int* neg_flag = ddiag + (td1 - td0);
  unsigned short* el = reinterpret_cast<unsigned short*>(neg_flag + kThreads / 64);
  const int tid = threadIdx.x, NT = blockDim.x;
  if ((tid & 63) == 0) neg_flag[tid >> 6] = 0;   // one flag per wave: cleared and set in the wave's own program order
  const int slot_lm_first = slot_lm[q0 + min(tid, nslots - 1)];
  {
    // All global loads of the tile are issued before the first LDS store (one HBM round trip in total);
    // anything beyond the unrolled part (only with an enlarged tile budget) goes through stage_copy.
    const dbl2_u* srcB = reinterpret_cast<const dbl2_u*>(Hpl + (size_t)q0 * PL);
    dbl2_u* dstB = reinterpret_cast<dbl2_u*>(Bs);
    const int n2 = (nslots * PL) >> 1;
    const double* srcD = (Hll ? Hll : Dinv) + (size_t)l0 * LD * LD;
    const double* srcb = bl + (size_t)l0 * LD;
    const int nD = nlm * LD * LD, nb = nlm * LD, ndp = td1 - td0 + 1;
    constexpr int UB = 12, UD = 3, UE = 4;
    dbl2_u vB[UB];
    double vD[UD], vb;
    int vE[UE], vP, vG;
    unsigned short vL[UE];
    // branch-free loads (indices clamped into range) so that the compiler emits them back to back
    // without intermediate s_waitcnt; the LDS stores below are predicated instead
#pragma unroll
    for (int u = 0; u < UB; ++u) vB[u] = srcB[min(tid + u * NT, n2 - 1)];
#pragma unroll
    for (int u = 0; u < UD; ++u) vD[u] = srcD[min(tid + u * NT, nD - 1)];
    vb = srcb[min(tid, nb - 1)];
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const int i = min(tid + u * NT, ne - 1);
      vE[u] = te_pack[e0 + i];
      vL[u] = te_lm[e0 + i];
    }
    vP = td_ptr[td0 + min(tid, ndp - 1)];
    vG = td_diag[td0 + min(tid, max(ndp - 2, 0))];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int i = tid + u * NT;
      if (i < n2) dstB[i] = vB[u];
    }
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const int i = tid + u * NT;
      if (i < nD) Ds[i] = vD[u];
    }
    if (tid < nb) bsm[tid] = vb;
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const int i = tid + u * NT;
      if (i < ne) {
        ep[i] = vE[u];
        el[i] = vL[u];
      }
    }
    if (tid < ndp) dptr[tid] = vP;
    if (tid < ndp - 1) ddiag[tid] = vG;
    // remainders
    if (n2 > UB * NT) stage_copy<4>(dstB + UB * NT, srcB + UB * NT, n2 - UB * NT, tid, NT);
    if (((nslots * PL) & 1) && tid == 0) Bs[nslots * PL - 1] = Hpl[(size_t)q0 * PL + nslots * PL - 1];
    for (int i = tid; i < t2.y * PL; i += NT) Bs[nslots * PL + i] = Hpl[(size_t)t2.x * PL + i];
    if (nD > UD * NT) stage_copy<4>(Ds + UD * NT, srcD + UD * NT, nD - UD * NT, tid, NT);
    if (nb > NT) stage_copy<2>(bsm + NT, srcb + NT, nb - NT, tid, NT);
    if (ne > UE * NT) {
      stage_copy<4>(ep + UE * NT, te_pack + e0 + UE * NT, ne - UE * NT, tid, NT);
      stage_copy<4>(el + UE * NT, te_lm + e0 + UE * NT, ne - UE * NT, tid, NT);
    }
    if (ndp > NT) stage_copy<2>(dptr + NT, td_ptr + td0 + NT, ndp - NT, tid, NT);
    if (ndp - 1 > NT) stage_copy<2>(ddiag + NT, td_diag + td0 + NT, ndp - 1 - NT, tid, NT);
  }
  __syncthreads();
  schur_tile_prepare<PD, LD, ((PD % 2 == 0 && G >= 2) ? 2 : 1)>(Bs, Ds, bsm, Cs, neg_flag, l0, nlm, nslots + t2.y, slot_lm_first, slot_lm, q0, nslots, Dinv, Hll, lam, tid, NT);
  schur_tile_dests<PD, LD, G>(Bs, Cs, bsm, tile_flag(neg_flag), ep, dptr, ddiag, el, td0, td1, e0, Pd, Pr, tid, NT);
}

// K5+K7+K8, pass 2: Hschur(d) = Hpp(d) - sum_tiles partial(d) (fixed tile order), bschur = b_p - sum partial_rhs
template <int PD, int NRP>
__global__ void __launch_bounds__(kThreads) schur_reduce_kernel(int nDst, const int* __restrict__ rd_ptr, const int* __restrict__ rd_slot,
                                                              const int* __restrict__ hs_src, const double* __restrict__ Hpp,
                                                              const double* __restrict__ Pd, double* __restrict__ Hs,
                                                              const int* __restrict__ hs_diag, const double* __restrict__ Pr,
                                                              const double* __restrict__ b, double* __restrict__ bschur,
                                                              const double* __restrict__ lam, const unsigned char* __restrict__ lam_mask,
                                                              const int* __restrict__ active, int n_slots) {
  constexpr int BB = PD * PD;
  const size_t ta = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (ta >= (size_t)nDst * BB) return;
  // multi-GPU: only the blocks this rank produces or consumes (the others stay zero)
  const int e = (int)(ta % BB), d = active ? active[ta / BB] : (int)(ta / BB);
  const size_t t = (size_t)d * BB + e;
  const int src = hs_src[d];
  const int pose = hs_diag[d];
  double v = src >= 0 ? Hpp[(size_t)src * BB + e] : 0.0;
  // virtual damping of the pose blocks: (Hpp + lambda I) first, like the materialised setLambda did
  if (pose >= 0 && e % (PD + 1) == 0 && (!lam_mask || lam_mask[pose])) v += lam[0];
  const int k0 = rd_ptr[d], k1 = rd_ptr[d + 1];
  // partial blocks are stored [row part][column][row inside the part] (see schur_tile_dests)
  const int pe = (e % PD) / NRP * (NRP * PD) + (e % PD) % NRP + NRP * (e / PD);
  // a block has one to three partials almost always: their slot ids, then their values, are requested together
  // (two dependent memory round trips instead of one per partial); same subtraction order as the plain loop
  const int n = k1 - k0, kc = min(k0, n_slots - 1);
  const int s0 = rd_slot[kc], s1 = rd_slot[min(kc + 1, n_slots - 1)], s2 = rd_slot[min(kc + 2, n_slots - 1)];
  const double p0 = Pd[(size_t)s0 * BB + pe], p1 = Pd[(size_t)s1 * BB + pe], p2 = Pd[(size_t)s2 * BB + pe];
  if (n > 0) v -= p0;
  if (n > 1) v -= p1;
  if (n > 2) v -= p2;
  // long lists (loop closures: a pose pair shares landmarks of many tiles): eight partials per pair of round trips
  for (int k = k0 + 3; k < k1; k += 8) {
    int sl[8];
    double pv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sl[j] = rd_slot[min(k + j, k1 - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = Pd[(size_t)sl[j] * BB + pe];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k + j < k1) v -= pv[j];
  }
  Hs[t] = v;
  if (pose >= 0 && e < PD) {
    double r = b[(size_t)pose * PD + e];
    for (int k = k0; k < k1; k += 8) {
      int sl[8];
      double pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) sl[j] = rd_slot[min(k + j, k1 - 1)];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = Pr[(size_t)sl[j] * PD + e];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k + j < k1) r -= pv[j];
    }
    bschur[(size_t)pose * PD + e] = r;
  }
}

// the right-hand side half of the reduction alone (the matrix half is folded into the factorisation's front
// assembly: SparseCholesky::set_virtual_blocks): bschur = b_p - sum partial_rhs, same order as schur_reduce_kernel
template <int PD>
__global__ void __launch_bounds__(kThreads) schur_rhs_kernel(int nP, const int* __restrict__ pose_diag, const int* __restrict__ rd_ptr,
                                                           const int* __restrict__ rd_slot, const double* __restrict__ Pr,
                                                           const double* __restrict__ b, double* __restrict__ bschur,
                                                           const int* __restrict__ iperm, double* __restrict__ xp, int* __restrict__ status) {
  // iperm / xp: the right-hand side also goes out in elimination order (what SparseCholesky::solve_begin would produce), status:
  // the factorisation's status word is cleared here -- two launches less in front of the factorisation
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (status && t == 0) *status = 0;
  if (t >= nP * PD) return;
  const int pose = t / PD, e = t - pose * PD;
  const int d = pose_diag[pose];
  double r = b[t];
  for (int k = rd_ptr[d], k1 = rd_ptr[d + 1]; k < k1; k += 8) {   // (eight partials per pair of round trips, list order kept)
    int sl[8];
    double pv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sl[j] = rd_slot[min(k + j, k1 - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = Pr[(size_t)sl[j] * PD + e];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k + j < k1) r -= pv[j];
  }
  bschur[t] = r;
  if (xp) xp[(size_t)iperm[pose] * PD + e] = r;
}

// K13: x_l = Dinv (b_l - Hpl' x_p)   (block_solver.hpp:459-483)
template <int PD, int LD>
__global__ void __launch_bounds__(kThreads) back_substitute_kernel(int nL, const int* __restrict__ pl_colptr, const int* __restrict__ pl_row,
                                       const double* __restrict__ Hpl, const double* __restrict__ Dinv,
                                       const double* __restrict__ bl, const double* __restrict__ xp, double* __restrict__ xl) {
  const int lm = blockIdx.x * blockDim.x + threadIdx.x;
  if (lm >= nL) return;
  double c[LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) c[j] = bl[(size_t)lm * LD + j];
  for (int q = pl_colptr[lm]; q < pl_colptr[lm + 1]; ++q) {
    const double* B = Hpl + (size_t)q * PD * LD;
    const double* xs = xp + (size_t)pl_row[q] * PD;
    double xv[PD], Bm[PD * LD];
    load_vec<PD>(xs, xv);
    load_vec<PD * LD>(B, Bm);
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      double t = 0.0;
#pragma unroll
      for (int r = 0; r < PD; ++r) t += Bm[r + PD * j] * xv[r];
      c[j] -= t;
    }
  }
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < LD; ++j) t += Dinv[(size_t)lm * LD * LD + i + LD * j] * c[j];
    xl[(size_t)lm * LD + i] = t;
  }
}

// K14: dest += H src over the full system, destination-major like the assembly (no atomics: the result does not depend on
// the launch's timing; Dogleg and the residual checks read it).  Hpp part: BlockPCG's gather-form symmetric product;
// Hpl part: thread = scalar row of a pose over the pose-major list of its Hpl blocks / thread = landmark.
__global__ void __launch_bounds__(kThreads) add_vec_kernel(size_t n, const double* __restrict__ a, double* __restrict__ y) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) y[i] += a[i];
}
__global__ void __launch_bounds__(kThreads) spmv_pl_pose_kernel(int nP, int p, int l, size_t sizeP, const int* __restrict__ pm_ptr,
                                                              const int* __restrict__ pm_q, const int* __restrict__ pm_lm,
                                                              const double* __restrict__ Hpl, const double* __restrict__ src,
                                                              double* __restrict__ dst) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= (size_t)nP * p) return;
  const int i = (int)(t / p), r = (int)(t - (size_t)i * p);
  double acc = 0.0;
  for (int k = pm_ptr[i]; k < pm_ptr[i + 1]; ++k) {   // ascending landmark order: fixed
    const double* B = Hpl + (size_t)pm_q[k] * p * l + r;
    const double* sv = src + sizeP + (size_t)pm_lm[k] * l;
    for (int c = 0; c < l; ++c) acc += B[(size_t)p * c] * sv[c];
  }
  dst[t] += acc;
}
__global__ void __launch_bounds__(kThreads) spmv_pl_landmark_kernel(int nL, int p, int l, size_t sizeP, const int* __restrict__ colptr,
                                                                  const int* __restrict__ row, const double* __restrict__ Hpl,
                                                                  const double* __restrict__ Hll, const double* __restrict__ src,
                                                                  double* __restrict__ dst) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= (size_t)nL * l) return;
  const int lm = (int)(t / l), j = (int)(t - (size_t)lm * l);
  double acc = 0.0;
  for (int q = colptr[lm]; q < colptr[lm + 1]; ++q) {
    const double* B = Hpl + (size_t)q * p * l + (size_t)p * j;
    const double* sv = src + (size_t)row[q] * p;
    for (int i = 0; i < p; ++i) acc += B[i] * sv[i];
  }
  const double* D = Hll + (size_t)lm * l * l;
  for (int c = 0; c < l; ++c) acc += D[j + l * c] * src[sizeP + (size_t)lm * l + c];
  dst[sizeP + t] += acc;
}

// ---------------------------------------------------------------------------------
// Device-side input producers for EdgeProjectXYZ2UV:
//   computeError      g2o/types/sba/types_six_dof_expmap.h:139-147 (obs - cam_map(T.map(X)))
//   linearizeOplus    g2o/types/sba/types_six_dof_expmap.cpp:288-326
// cams: [n][12] = R (column-major) | t, world -> camera.  Jacobians are written in the layout
// g2ohip_set_edge_data documents (2x3 point block = J0, 2x6 pose block = J1, column-major).
// ---------------------------------------------------------------------------------
__device__ __forceinline__ double ba_linearize_edge(int e, const double* __restrict__ cams, const double* __restrict__ pts,
                                                    const int* __restrict__ cam_v, const int* __restrict__ pt_v,
                                                    const double* __restrict__ meas, double f, double cx, double cy,
                                                    double* __restrict__ Jpt, double* __restrict__ Jcam, double* __restrict__ err,
                                                    int want_jac, const double* __restrict__ omega, int ident, int kind, double delta,
                                                    bool want_rho, const double* __restrict__ ctab) {
  double T[12], X[3], z2[2];
  int cw = cam_v[e];
  if (ctab) cw = ba_edge_class<true>(cw, ctab, f, cx, cy, kind, delta);
  load_vec<12>(cams + (size_t)cw * 12, T);
  const double* Xp = pts + (size_t)pt_v[e] * 3;
  X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2];
  load_vec<2>(meas + (size_t)e * 2, z2);
  const double x = T[0] * X[0] + T[3] * X[1] + T[6] * X[2] + T[9];
  const double y = T[1] * X[0] + T[4] * X[1] + T[7] * X[2] + T[10];
  const double z = T[2] * X[0] + T[5] * X[1] + T[8] * X[2] + T[11];
  double r[2] = {z2[0] - (x / z * f + cx), z2[1] - (y / z * f + cy)};
  store_vec<2>(err + (size_t)e * 2, r);
  double rho = 0.0;
  if (want_rho) {   // e' Omega e as chi2_kernel<2> forms it
    double e2;
    if (ident) {
      e2 = r[0] * r[0] + r[1] * r[1];
    } else {
      const double* O = omega + (size_t)e * 4;
      e2 = r[0] * (O[0] * r[0] + O[2] * r[1]) + r[1] * (O[1] * r[0] + O[3] * r[1]);
    }
    rho = robust_rho(kind, delta, e2);
  }
  if (!want_jac) return rho;
  const double z_2 = z * z;
  const double tmp[6] = {f, 0.0, -x / z * f, 0.0, f, -y / z * f};   // row-major 2x3
  double A[6], B[12];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m) t += tmp[rr * 3 + m] * T[m + 3 * c];
      A[rr + 2 * c] = -1.0 / z * t;
    }
  B[0 + 2 * 0] = x * y / z_2 * f;        B[0 + 2 * 1] = -(1.0 + (x * x / z_2)) * f; B[0 + 2 * 2] = y / z * f;
  B[0 + 2 * 3] = -1.0 / z * f;           B[0 + 2 * 4] = 0.0;                        B[0 + 2 * 5] = x / z_2 * f;
  B[1 + 2 * 0] = (1.0 + y * y / z_2) * f; B[1 + 2 * 1] = -x * y / z_2 * f;          B[1 + 2 * 2] = -x / z * f;
  B[1 + 2 * 3] = 0.0;                    B[1 + 2 * 4] = -1.0 / z * f;               B[1 + 2 * 5] = y / z_2 * f;
  store_vec<6>(Jpt + (size_t)e * 6, A);
  store_vec<12>(Jcam + (size_t)e * 12, B);
  return rho;
}

__global__ void __launch_bounds__(kThreads) ba_linearize_kernel(int n, const double* __restrict__ cams, const double* __restrict__ pts,
                                                              const int* __restrict__ cam_v, const int* __restrict__ pt_v,
                                                              const double* __restrict__ meas, double f, double cx, double cy,
                                                              double* __restrict__ Jpt, double* __restrict__ Jcam,
                                                              double* __restrict__ err, int want_jac,
                                                              const double* __restrict__ omega, int ident, int kind, double delta,
                                                              double* __restrict__ chi_part, const double* __restrict__ ctab) {
  // chi_part != nullptr: the robustified chi2 of the edges (sparse_optimizer.cpp:100-114) rides along -- one partial sum
  // per workgroup, folded to 1 024 by fold_partials_kernel (fixed order); saves the pass of chi2_kernel over the errors
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  double rho = 0.0;
  if (e < n) rho = ba_linearize_edge(e, cams, pts, cam_v, pt_v, meas, f, cx, cy, Jpt, Jcam, err, want_jac, omega, ident, kind, delta, chi_part != nullptr, ctab);
  if (chi_part) {
    __shared__ double sh[kThreads];
    sh[threadIdx.x] = rho;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) chi_part[blockIdx.x] = sh[0];
  }
}
__global__ void __launch_bounds__(1024) fold_partials_kernel(const double* __restrict__ src, int n, double* __restrict__ dst) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += src[i];
  dst[threadIdx.x] = s;
}

// VertexSE3Expmap::oplusImpl: estimate <- SE3Quat::exp(update) * estimate
// (g2o/types/sba/types_six_dof_expmap.h:101-104, g2o/types/slam3d/se3quat.h:223-257); update = (omega, upsilon)
__global__ void __launch_bounds__(kThreads) ba_update_cams_kernel(int nc, double* __restrict__ cams, const int* __restrict__ hidx,
                                                                const double* __restrict__ xp) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nc) return;
  const int h = hidx[v];
  if (h < 0) return;
  const double* u = xp + (size_t)h * 6;
  double* T = cams + (size_t)v * 12;
  const double wx = u[0], wy = u[1], wz = u[2];
  const double theta = sqrt(wx * wx + wy * wy + wz * wz);
  const double Om[9] = {0.0, wz, -wy, -wz, 0.0, wx, wy, -wx, 0.0};   // column-major skew(omega)
  double Om2[9], R[9], V[9];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m) t += Om[r + 3 * m] * Om[m + 3 * c];
      Om2[r + 3 * c] = t;
    }
  double a, b, cc;
  const bool small = theta < 0.00001;
  if (!small) {
    a = sin(theta) / theta;
    b = (1.0 - cos(theta)) / (theta * theta);
    cc = (theta - sin(theta)) / (theta * theta * theta);
  } else {
    a = 1.0; b = 1.0; cc = 1.0;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double I = (i % 4 == 0) ? 1.0 : 0.0;
    R[i] = I + a * Om[i] + b * Om2[i];
    V[i] = small ? R[i] : I + b * Om[i] + cc * Om2[i];
  }
  double Rn[9], tn[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double t = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m) t += R[r + 3 * m] * T[m + 3 * c];
      Rn[r + 3 * c] = t;
    }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    tn[r] = R[r] * T[9] + R[r + 3] * T[10] + R[r + 6] * T[11] + V[r] * u[3] + V[r + 3] * u[4] + V[r + 6] * u[5];
#pragma unroll
  for (int i = 0; i < 9; ++i) T[i] = Rn[i];
  T[9] = tn[0]; T[10] = tn[1]; T[11] = tn[2];
}
// VertexSBAPointXYZ::oplusImpl (g2o/types/sba/types_sba.h:151-155)
__global__ void __launch_bounds__(kThreads) ba_update_pts_kernel(int np, double* __restrict__ pts, const int* __restrict__ hidx,
                                                               const double* __restrict__ xl) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np * 3) return;
  const int v = t / 3, c = t % 3;
  const int h = hidx[v];
  if (h >= 0) pts[t] += xl[(size_t)h * 3 + c];
}

// Shared by the fused assembly kernels: projection error, Jacobians and robust weight of one edge.
struct BaEdgeLin {
  double A[6], B[12], r[2], O[4], w;   // A: 2x3 point block, B: 2x6 pose block (column-major), r: error
};
__device__ __forceinline__ void ba_edge_linearize(const double* __restrict__ T, const double* __restrict__ X, const double* __restrict__ z2,
                                                  const double* __restrict__ Op, double f, double cx, double cy, int kind, double delta,
                                                  bool want_A, BaEdgeLin& L) {
  const double x = T[0] * X[0] + T[3] * X[1] + T[6] * X[2] + T[9];
  const double y = T[1] * X[0] + T[4] * X[1] + T[7] * X[2] + T[10];
  const double z = T[2] * X[0] + T[5] * X[1] + T[8] * X[2] + T[11];
  const double iz = 1.0 / z;
  L.r[0] = z2[0] - (x * iz * f + cx);
  L.r[1] = z2[1] - (y * iz * f + cy);
  L.O[0] = Op[0]; L.O[1] = Op[1]; L.O[2] = Op[2]; L.O[3] = Op[3];
  const double Or0 = L.O[0] * L.r[0] + L.O[2] * L.r[1], Or1 = L.O[1] * L.r[0] + L.O[3] * L.r[1];
  L.w = robust_weight(kind, delta, L.r[0] * Or0 + L.r[1] * Or1);
  // one reciprocal instead of a dozen fp64 divisions (each ~30 instructions): x/z -> x*iz etc.
  const double fz = f * iz, xz = x * iz, yz = y * iz;
  if (want_A) {
    const double tmp[6] = {f, 0.0, -xz * f, 0.0, f, -yz * f};
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m) t += tmp[rr * 3 + m] * T[m + 3 * c];
        L.A[rr + 2 * c] = -iz * t;
      }
  }
  L.B[0 + 2 * 0] = xz * yz * f;          L.B[0 + 2 * 1] = -(1.0 + xz * xz) * f; L.B[0 + 2 * 2] = yz * f;
  L.B[0 + 2 * 3] = -fz;                  L.B[0 + 2 * 4] = 0.0;                  L.B[0 + 2 * 5] = xz * fz;
  L.B[1 + 2 * 0] = (1.0 + yz * yz) * f;  L.B[1 + 2 * 1] = -xz * yz * f;         L.B[1 + 2 * 2] = -xz * f;
  L.B[1 + 2 * 3] = 0.0;                  L.B[1 + 2 * 4] = -fz;                  L.B[1 + 2 * 5] = yz * fz;
}

// One observation's contribution to its landmark: H += A' (w Omega) A, b -= A' (w Omega) r (base_binary_edge.hpp:86-118,
// landmark half); OA = (w Omega) A is handed back for the Hpl block.  Shared by the stand-alone landmark assembly and by
// the Schur tiles that assemble their landmarks themselves: the same operations in the same order.
__device__ __forceinline__ void ba_lm_accumulate(const BaEdgeLin& L, double (&H)[9], double (&b)[3], double (&OA)[6]) {
  const double w = L.w;
  const double O00 = w * L.O[0], O10 = w * L.O[1], O01 = w * L.O[2], O11 = w * L.O[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    OA[0 + 2 * c] = O00 * L.A[0 + 2 * c] + O01 * L.A[1 + 2 * c];
    OA[1 + 2 * c] = O10 * L.A[0 + 2 * c] + O11 * L.A[1 + 2 * c];
  }
  const double Or0 = O00 * L.r[0] + O01 * L.r[1], Or1 = O10 * L.r[0] + O11 * L.r[1];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    b[c] -= L.A[0 + 2 * c] * Or0 + L.A[1 + 2 * c] * Or1;
#pragma unroll
    for (int a = 0; a < 3; ++a) H[a + 3 * c] += L.A[0 + 2 * a] * OA[0 + 2 * c] + L.A[1 + 2 * a] * OA[1 + 2 * c];
  }
}
// Hpl(pose, lm) = B' (w Omega) A, 6 x 3 column-major
__device__ __forceinline__ void ba_hpl_block(const BaEdgeLin& L, const double (&OA)[6], double (&blk)[18]) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int a = 0; a < 6; ++a) blk[a + 6 * c] = L.B[0 + 2 * a] * OA[0 + 2 * c] + L.B[1 + 2 * a] * OA[1 + 2 * c];
}

// Fused buildSystem for EdgeProjectXYZ2UV graphs, landmark side (K1-K2 of SURVEY.md 2.3 with the
// reference's per-edge linearizeOplus + constructQuadraticForm, block_solver.hpp:529-532): one thread
// per landmark walks its observations, evaluates error and Jacobians on the fly (no Jacobian arrays in
// HBM), accumulates Hll / b_l and writes each Hpl block exactly once.
#ifndef G2OHIP_ASMLM_OCC
#define G2OHIP_ASMLM_OCC 4   // 126 VGPRs, 36 KB LDS: 4 workgroups per CU
#endif
template <int G, bool CLS = false>
__global__ void __launch_bounds__(kThreads, G2OHIP_ASMLM_OCC) ba_assemble_landmarks_kernel(
    int nL, const int* __restrict__ vptr, const int* __restrict__ vent, const double* __restrict__ cams, const double* __restrict__ pts,
    const int* __restrict__ cam_lm, const int* __restrict__ pt_lm, const double* __restrict__ meas_lm, const double* __restrict__ omega_lm,
    const int* __restrict__ hpl_lm, double f, double cx, double cy, int kind, double delta, double* __restrict__ Hll,
    double* __restrict__ bl, double* __restrict__ Hpl, double* __restrict__ err, int ident, const int* __restrict__ pl_colptr,
    int write_hpl, const double* __restrict__ ctab) {
  // write_hpl == 0: nobody reads Hpl this iteration (the Schur tiles and the back-substitution re-evaluate the Jacobians,
  // ba_schur_tile_kernel / ba_back_substitute_kernel): only Hll, b_l and the errors are produced -- 0.72 GB less to write
  // at the metric configuration.
  // The Hpl blocks of a wave's landmarks are contiguous in HBM (block-CCS by landmark column).  Written by
  // their lanes directly they would be 16-byte pieces at a 144-byte stride (64 memory transactions per
  // store instruction); instead the wave collects them in LDS and streams them out fully coalesced.
  __shared__ __attribute__((aligned(16))) double stage[kThreads / 64][64 * 18];
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int lm = gt / G, g = gt % G;
  const bool active = lm < nL;
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int lm_w0 = min((gt - lane) / G, nL), lm_w1 = min(lm_w0 + 64 / G, nL);   // landmarks of this wave
  const int q_base = pl_colptr[lm_w0], nslots = pl_colptr[lm_w1] - q_base;          // wave-uniform
  const bool staged = nslots <= 64;
  double H[9], b[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) H[i] = 0.0;
  b[0] = b[1] = b[2] = 0.0;
  const int k0 = active ? vptr[lm] : 0, k1 = active ? vptr[lm + 1] : 0;
  for (int k = k0 + g; k < k1; k += G) {
    const int e = vent[k] >> 1;   // (only for the error store; everything read is in observation-list order)
    double T[12], X[3], z2[2], Op[4];
    const double* Xp = pts + (size_t)pt_lm[k] * 3;
    X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2];
    load_vec<12>(cams + (size_t)ba_edge_class<CLS>(cam_lm[k], ctab, f, cx, cy, kind, delta) * 12, T);
    load_vec<2>(meas_lm + (size_t)k * 2, z2);
    if (ident) {   // information().setIdentity() declared for the whole set: no per-edge read
      Op[0] = Op[3] = 1.0;
      Op[1] = Op[2] = 0.0;
    } else {
      load_vec<4>(omega_lm + (size_t)k * 4, Op);
    }
    BaEdgeLin L;
    ba_edge_linearize(T, X, z2, Op, f, cx, cy, kind, delta, true, L);
    store_vec<2>(err + (size_t)e * 2, L.r);
    double OA[6];
    ba_lm_accumulate(L, H, b, OA);
    const int q = write_hpl ? hpl_lm[k] : -1;
    if (q >= 0) {   // Hpl(pose, lm) = B' (w Omega) A  (written transposed, block_solver.hpp:240-244)
      double blk[18];
      ba_hpl_block(L, OA, blk);
      if (staged) store_vec<18>(&stage[wave][(q - q_base) * 18], blk);
      else store_vec<18>(Hpl + (size_t)q * 18, blk);
    }
  }
  if (staged && write_hpl) {   // (same wave wrote and reads: LDS operations of a wave complete in order)
    __builtin_amdgcn_wave_barrier();
    const dbl2_u* src = reinterpret_cast<const dbl2_u*>(&stage[wave][0]);
    dbl2_u* dst = reinterpret_cast<dbl2_u*>(Hpl + (size_t)q_base * 18);
    for (int t = lane; t < nslots * 9; t += 64) dst[t] = src[t];
  }
  if (G > 1) {
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = group_sum<G>(H[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = group_sum<G>(b[i]);
  }
  if (!active) return;
#pragma unroll
  for (int i = 0; i < 9; ++i)
    if (G == 1 || (i % G) == g) Hll[(size_t)lm * 9 + i] = H[i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (G == 1 || (i % G) == g) bl[(size_t)lm * 3 + i] = b[i];
}

// The same tile pass for the fused EdgeProjectXYZ2UV path WITHOUT reading Hpl: the tile's Hpl blocks are evaluated from
// the estimates the system was built from, straight into the LDS stage (one lane per block: projection, the two Jacobians,
// the robust weight, B' (w Omega) A -- ~150 flops against 144 bytes of HBM read per block).  Hpl then never exists in HBM
// unless somebody asks for it (BlockSolver::ensure_hpl).  Block q of Hpl <-> one observation (ba_set_edges checked that),
// whose inputs are kept in block order (cam_q, pt_q, meas_q, omega_q).
// FLL: the tile also ASSEMBLES its landmarks (Hll, b_l, the errors: what ba_assemble_landmarks_kernel produces) while it
// evaluates their observations for the Hpl blocks -- build_system then launches nothing for the landmark side and every
// observation is evaluated once per iteration instead of twice.  Hll and b_l go to the LDS stage and to HBM (readers: the
// back-substitution, b, computeScale, multiplyHessian); they are summed in observation-list order, the stand-alone kernel
// sums per lane group: the two agree to rounding, not bit for bit.
template <int G, bool FLL, bool CLS = false>
__global__ void __launch_bounds__(kThreads, G2OHIP_SCHUR_OCC) ba_schur_tile_kernel(
    const int* __restrict__ tile_lm0, const double* __restrict__ cams, const double* __restrict__ pts, const int* __restrict__ cam_q,
    const int* __restrict__ pt_q, const double* __restrict__ meas_q, const double* __restrict__ omega_q, double f, double cx, double cy,
    int kind, double delta, int ident, double* __restrict__ Dinv, double* bl, const int* __restrict__ td_diag,
    const int* __restrict__ td_ptr, const int* __restrict__ te_pack, const unsigned short* __restrict__ te_lm, double* __restrict__ Pd,
    double* __restrict__ Pr, double* Hll, const double* __restrict__ lam, const int4* __restrict__ ll_rec,
    const int4* __restrict__ tile_ll, const int* __restrict__ ll_edge, double* __restrict__ err, const int2* __restrict__ tile_q2,
    int store_hll, const unsigned short* __restrict__ slot_lm, const double* __restrict__ ctab) {
  // (FLL: meas_q / omega_q are the slot-major copies ll_meas / ll_omega; cam_q / pt_q are unused; store_hll = 0: Hll stays in
  // the LDS stage -- the solve path reads Dinv and b_l only --, err = nullptr: the errors are not written either)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int PD = 6, LD = 3, PL = PD * LD;
  const int t = xcd_swizzle(blockIdx.x, gridDim.x);
  const int* tm = tile_lm0 + (size_t)t * 8;   // packed tile record (see build_structure)
  const int l0 = tm[0], l1 = tm[1], q0 = tm[2], nslots = tm[3], nlm = l1 - l0;
  const int td0 = tm[4], td1 = tm[5], e0 = tm[6], ne = tm[7];
  const int2 t2 = FLL ? make_int2(0, 0) : tile_q2[t];   // second block range of a split landmark's tile (never with FLL)
  const int ntot = nslots + t2.y;
  double* Bs = smem;
  double* Ds = Bs + ((ntot * PL + 1) & ~1);
  double* bsm = Ds + ((nlm * LD * LD + 1) & ~1);
  double* Cs = bsm + ((nlm * LD + 1) & ~1);
  int* ep = reinterpret_cast<int*>(Cs + nlm * TileSplit<LD>::CP);
  int* dptr = ep + ne;
  int* ddiag = dptr + (td1 - td0 + 1);
  int* neg_flag = ddiag + (td1 - td0);
  unsigned short* el = reinterpret_cast<unsigned short*>(neg_flag + kThreads / 64);
  const int tid = threadIdx.x, NT = blockDim.x;
  if ((tid & 63) == 0) neg_flag[tid >> 6] = 0;   // one flag per wave: cleared and set in the wave's own program order
  const int slot_lm_first = FLL ? 0 : (int)slot_lm[q0 + min(tid, nslots - 1)];
  {
    const double* srcD = Hll + (size_t)l0 * LD * LD;
    const double* srcb = bl + (size_t)l0 * LD;
    const int nD = nlm * LD * LD, nb = nlm * LD, ndp = td1 - td0 + 1;
    constexpr int UD = 3, UE = 4;
    double vD[UD], vb;
    int vE[UE], vP, vG;
    unsigned short vL[UE];
    // the observation of this lane's block (first pass), requested with everything else the tile stages
    const int s0 = min(tid, nslots - 1);
    int cq = 0, pq = 0;
    if constexpr (!FLL) {
      cq = cam_q[q0 + s0];
      pq = pt_q[q0 + s0];
#pragma unroll
      for (int u = 0; u < UD; ++u) vD[u] = srcD[min(tid + u * NT, nD - 1)];
      vb = srcb[min(tid, nb - 1)];
    }
    auto request_lists = [&]() {
#pragma unroll
      for (int u = 0; u < UE; ++u) {
        const int i = min(tid + u * NT, ne - 1);
        vE[u] = te_pack[e0 + i];
        vL[u] = te_lm[e0 + i];
      }
      vP = td_ptr[td0 + min(tid, ndp - 1)];
      vG = td_diag[td0 + min(tid, max(ndp - 2, 0))];
    };
    request_lists();
    if constexpr (FLL) {
      // One lane per OBSERVATION (slot table of the tile: a landmark's observations sit in consecutive lanes of one
      // wavefront), then the contributions of a list are summed into its first lane by doubling (ds_bpermute).
      const int4 tl = tile_ll[t];   // first slot, slots (a multiple of 64), first observation, longest list
      for (int sb = 0; sb < tl.y; sb += NT) {
        const int si = sb + tid;
        const size_t sg = (size_t)tl.x + min(si, tl.y - 1);
        int4 rc = ll_rec[sg];
        if (si >= tl.y) rc.x = (int)0x80000fffu;
        const int ent = rc.x;
        const int krel = ent & 0xfff, K = (ent >> 12) & 0xff, lmi = (ent >> 20) & 0x7ff;   // (lmi: position in the list on the other lanes)
        const bool has = krel != 0xfff, head = ent >= 0;
        const int pos = head ? 0 : lmi;
        double v[9];   // lower triangle of H (6: the block is symmetric) | b (3)
        double Bk[12], OAk[6];   // pose Jacobian and (w Omega) A of this lane's observation, kept for its V block
#pragma unroll
        for (int i = 0; i < 12; ++i) Bk[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) OAk[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = 0.0;
        {
          const int ci = ba_edge_class<CLS>(rc.y, ctab, f, cx, cy, kind, delta), pi = rc.z, q = rc.w, e = ll_edge[sg];
          double T[12], X[3], z2[2], Op[4];
          const double* Xp = pts + (size_t)pi * 3;
          X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2];
          load_vec<12>(cams + (size_t)ci * 12, T);
          load_vec<2>(meas_q + sg * 2, z2);
          if (ident) {
            Op[0] = Op[3] = 1.0;
            Op[1] = Op[2] = 0.0;
          } else {
            load_vec<4>(omega_q + sg * 4, Op);
          }
          if (has && !(store_hll & 8)) {
            BaEdgeLin L;
            ba_edge_linearize(T, X, z2, Op, f, cx, cy, kind, delta, true, L);
            if (err) store_vec<2>(err + (size_t)e * 2, L.r);
            double H[9], b[3], OA[6];
#pragma unroll
            for (int i = 0; i < 9; ++i) H[i] = 0.0;
            b[0] = b[1] = b[2] = 0.0;
            ba_lm_accumulate(L, H, b, OA);
            v[0] = H[0]; v[1] = H[1]; v[2] = H[2]; v[3] = H[4]; v[4] = H[5]; v[5] = H[8];   // (column-major lower triangle)
#pragma unroll
            for (int i = 0; i < 3; ++i) v[6 + i] = b[i];
            if (q >= 0) {   // (the Hpl block B' (w Omega) A is never formed: V = B' ((w Omega) A C) once C is known, below)
#pragma unroll
              for (int i = 0; i < 12; ++i) Bk[i] = L.B[i];
#pragma unroll
              for (int i = 0; i < 6; ++i) OAk[i] = OA[i];
            }
          }
        }
        // suffix sums inside a list by doubling: after the step with distance d a lane holds the sum of up to 2 d list
        // entries from its own on (fixed tree: deterministic); as many steps as the longest list of this WAVE needs
        for (int d = 1; __any(pos + d < K); d <<= 1) {
          const double take = pos + d < K ? 1.0 : 0.0;
          double o[9];   // (all nine exchanges in flight at once, then branch-free adds)
#pragma unroll
          for (int i = 0; i < 9; ++i) o[i] = __shfl_down(v[i], d);
          // v + o or v, as ONE fused multiply-add per value (o * 1 is exact, so the sum is the plain one bit for bit) instead of two
          // selects and an add: the vector units are what this phase is short of.  (A non-finite term of a neighbouring list would
          // leak through 0 * o -- such a system is lost anyway: its right-hand side is not finite.)
#pragma unroll
          for (int i = 0; i < 9; ++i) v[i] = fma(o[i], take, v[i]);
        }
        // The list head inverts the damped block (block_solver.hpp:386-389) and splits the inverse (schur_tile_prepare's
        // comment); the lanes of the list fetch C from it and turn their staged block into V = B C on the spot: no pass over
        // the blocks and no barrier for it.
        double C[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) C[i] = 0.0;
        if (head) {
          const int lm = l0 + lmi;
          double* bd = bl + (size_t)lm * 3;
          double Hf[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
          if (store_hll & 1) {
            double* hd = Hll + (size_t)lm * 9;
#pragma unroll
            for (int i = 0; i < 9; ++i) hd[i] = Hf[i];
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) bd[i] = v[6 + i];
          double R[9], sg[3], u[3];
          const double lambda = lam[1];
          Hf[0] += lambda; Hf[4] += lambda; Hf[8] += lambda;
          small_inverse<3>(Hf, R);
          store_vec<9>(Dinv + (size_t)lm * 9, R);
          if (landmark_split<3>(R, v + 6, C, sg, u)) neg_flag[tid >> 6] = 1;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            Cs[lmi * TileSplit<3>::CP + 9 + i] = sg[i];
            bsm[lmi * 3 + i] = u[i];
          }
        }
        {
          const int src = (int)(threadIdx.x & 63) - pos;
          C[0] = __shfl(C[0], src); C[1] = __shfl(C[1], src); C[2] = __shfl(C[2], src);
          C[4] = __shfl(C[4], src); C[5] = __shfl(C[5], src); C[8] = __shfl(C[8], src);
          if (has && rc.w >= 0) {
            double OAC[6], V[18];   // (w Omega) A C: 2 x 3, C lower triangular
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                double t = OAk[r + 2 * c] * C[c + 3 * c];
#pragma unroll
                for (int k = c + 1; k < 3; ++k) t = fma(OAk[r + 2 * k], C[k + 3 * c], t);
                OAC[r + 2 * c] = t;
              }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
              for (int a = 0; a < 6; ++a)   // (slot layout of the destination loop: VSlot)
                V[VSlot<PD, LD, (G >= 2 ? 2 : 1)>::off(a, c)] = Bk[0 + 2 * a] * OAC[0 + 2 * c] + Bk[1 + 2 * a] * OAC[1 + 2 * c];
            store_vec<18>(Bs + (rc.w - q0) * 18, V);
          }
        }
      }
    } else {
      for (int sb = 0; sb < ntot; sb += NT) {
        const int s_ = sb + tid;
        const int sc = min(s_, ntot - 1);
        const int qq = sc < nslots ? q0 + sc : t2.x + (sc - nslots);   // Hpl block of this slot
        const bool pre = sb == 0 && s_ < nslots;
        const int c_ = ba_edge_class<CLS>(pre ? cq : cam_q[qq], ctab, f, cx, cy, kind, delta), p_ = pre ? pq : pt_q[qq];
        double T[12], X[3], z2[2], Op[4];
        load_vec<12>(cams + (size_t)c_ * 12, T);
        const double* Xp = pts + (size_t)p_ * 3;
        X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2];
        load_vec<2>(meas_q + (size_t)qq * 2, z2);
        if (ident) {
          Op[0] = Op[3] = 1.0;
          Op[1] = Op[2] = 0.0;
        } else {
          load_vec<4>(omega_q + (size_t)qq * 4, Op);
        }
        BaEdgeLin L;
        ba_edge_linearize(T, X, z2, Op, f, cx, cy, kind, delta, true, L);
        const double w = L.w;
        const double O00 = w * L.O[0], O10 = w * L.O[1], O01 = w * L.O[2], O11 = w * L.O[3];
        double OA[6], blk[18];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          OA[0 + 2 * c] = O00 * L.A[0 + 2 * c] + O01 * L.A[1 + 2 * c];
          OA[1 + 2 * c] = O10 * L.A[0 + 2 * c] + O11 * L.A[1 + 2 * c];
        }
        ba_hpl_block(L, OA, blk);
        if (s_ < ntot) store_vec<18>(Bs + s_ * 18, blk);
      }
#pragma unroll
      for (int u = 0; u < UD; ++u) {
        const int i = tid + u * NT;
        if (i < nD) Ds[i] = vD[u];
      }
      if (tid < nb) bsm[tid] = vb;
    }
#pragma unroll
    for (int u = 0; u < UE; ++u) {
      const int i = tid + u * NT;
      if (i < ne) {
        ep[i] = vE[u];
        el[i] = vL[u];
      }
    }
    if (tid < ndp) dptr[tid] = vP;
    if (tid < ndp - 1) ddiag[tid] = vG;
    if constexpr (!FLL) {
      if (nD > UD * NT) stage_copy<4>(Ds + UD * NT, srcD + UD * NT, nD - UD * NT, tid, NT);
      if (nb > NT) stage_copy<2>(bsm + NT, srcb + NT, nb - NT, tid, NT);
    }
    if (ne > UE * NT) {
      stage_copy<4>(ep + UE * NT, te_pack + e0 + UE * NT, ne - UE * NT, tid, NT);
      stage_copy<4>(el + UE * NT, te_lm + e0 + UE * NT, ne - UE * NT, tid, NT);
    }
    if (ndp > NT) stage_copy<2>(dptr + NT, td_ptr + td0 + NT, ndp - NT, tid, NT);
    if (ndp - 1 > NT) stage_copy<2>(ddiag + NT, td_diag + td0 + NT, ndp - 1 - NT, tid, NT);
  }
  __syncthreads();
  if (!FLL) schur_tile_prepare<PD, LD, ((PD % 2 == 0 && G >= 2) ? 2 : 1)>(Bs, Ds, bsm, Cs, neg_flag, l0, nlm, ntot, slot_lm_first, slot_lm, q0, nslots, Dinv, Hll, lam, tid, NT);
  if (!(store_hll & 2)) schur_tile_dests<PD, LD, G>(Bs, Cs, bsm, tile_flag(neg_flag), ep, dptr, ddiag, el, td0, td1, e0, Pd, Pr, tid, NT);
}

// K13 for the fused EdgeProjectXYZ2UV path: x_l = Dinv (b_l - Hpl' x_p) WITHOUT reading Hpl (block_solver.hpp:459-483).
// Hpl(pose, lm)' x_p = A' (w Omega) (B x_p): the two Jacobians of an observation are re-evaluated from the estimates the
// system was built from (12 + 3 + 2 doubles, the poses out of L2) instead of streaming an 18-double block per
// observation -- the kernel moves a sixth of the bytes.  G lanes per landmark like the assembly.
template <int G, bool CLS = false>
__global__ void __launch_bounds__(kThreads) ba_back_substitute_kernel(
    int nL, const int* __restrict__ vptr, const double* __restrict__ cams, const double* __restrict__ pts,
    const int* __restrict__ cam_lm, const int* __restrict__ pt_lm, const double* __restrict__ meas_lm, const double* __restrict__ omega_lm,
    const int* __restrict__ row_lm, double f, double cx, double cy, int kind, double delta, int ident,
    const double* __restrict__ Dinv, const double* __restrict__ bl, const double* __restrict__ xp, double* __restrict__ xl,
    const double* __restrict__ ctab) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int lm = gt / G, g = gt % G;
  const bool active = lm < nL;
  double c[3] = {0.0, 0.0, 0.0};
  const int k0 = active ? vptr[lm] : 0, k1 = active ? vptr[lm + 1] : 0;
  for (int k = k0 + g; k < k1; k += G) {
    const int row = row_lm[k];
    if (row < 0) continue;   // fixed pose: no Hpl block
    double T[12], X[3], z2[2], Op[4], xv[6];
    const double* Xp = pts + (size_t)pt_lm[k] * 3;
    X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2];
    load_vec<12>(cams + (size_t)ba_edge_class<CLS>(cam_lm[k], ctab, f, cx, cy, kind, delta) * 12, T);
    load_vec<2>(meas_lm + (size_t)k * 2, z2);
    load_vec<6>(xp + (size_t)row * 6, xv);
    if (ident) {
      Op[0] = Op[3] = 1.0;
      Op[1] = Op[2] = 0.0;
    } else {
      load_vec<4>(omega_lm + (size_t)k * 4, Op);
    }
    BaEdgeLin L;
    ba_edge_linearize(T, X, z2, Op, f, cx, cy, kind, delta, true, L);
    double t0 = 0.0, t1 = 0.0;   // B x_p
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      t0 += L.B[0 + 2 * a] * xv[a];
      t1 += L.B[1 + 2 * a] * xv[a];
    }
    const double w = L.w;
    const double u0 = w * (L.O[0] * t0 + L.O[2] * t1), u1 = w * (L.O[1] * t0 + L.O[3] * t1);
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] -= L.A[0 + 2 * j] * u0 + L.A[1 + 2 * j] * u1;
  }
  if (G > 1) {
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] = group_sum<G>(c[j]);
  }
  if (!active || g >= 3) return;
  double cl[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) cl[j] = c[j] + bl[(size_t)lm * 3 + j];
  const double* D = Dinv + (size_t)lm * 9;
  if (G == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) xl[(size_t)lm * 3 + i] = D[i] * cl[0] + D[i + 3] * cl[1] + D[i + 6] * cl[2];
  } else {
    xl[(size_t)lm * 3 + g] = D[g] * cl[0] + D[g + 3] * cl[1] + D[g + 6] * cl[2];
  }
}

// The same back-substitution over the lane slots of the Schur tiles (one lane per observation, 92 % of the lanes busy
// where eight lanes per landmark keep five of eight): the terms of a list are summed into its first lane by doubling.
template <bool CLS>
__global__ void __launch_bounds__(kThreads) ba_back_substitute_slots_kernel(
    const int* __restrict__ tile_lm0, const int4* __restrict__ tile_ll, const int4* __restrict__ ll_rec, const int* __restrict__ ll_row,
    const double* __restrict__ ll_meas, const double* __restrict__ ll_omega, const double* __restrict__ cams,
    const double* __restrict__ pts, double f, double cx, double cy, int kind, double delta, int ident, const double* __restrict__ Dinv,
    const double* __restrict__ bl, const double* __restrict__ xp, double* __restrict__ xl, const double* __restrict__ ctab) {
  const int t = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
  const int l0 = tile_lm0[(size_t)t * 8];
  const int4 tl = tile_ll[t];   // first slot, slots (a multiple of 64), first observation, longest list
  for (int sb = 0; sb < tl.y; sb += NT) {
    const int si = sb + tid;
    const size_t sg = (size_t)tl.x + min(si, tl.y - 1);
    int4 rc = ll_rec[sg];
    if (si >= tl.y) rc.x = (int)0x80000fffu;
    const int ent = rc.x;
    const int krel = ent & 0xfff, K = (ent >> 12) & 0xff, lmi = (ent >> 20) & 0x7ff;
    const bool head = ent >= 0;
    const int pos = head ? 0 : lmi;
    const int row = krel != 0xfff ? ll_row[sg] : -1;
    double c[3] = {0.0, 0.0, 0.0};
    {
      double T[12], X[3], z2[2], Op[4], xv[6];
      const double* Xp = pts + (size_t)rc.z * 3;
      X[0] = Xp[0]; X[1] = Xp[1]; X[2] = Xp[2];
      load_vec<12>(cams + (size_t)ba_edge_class<CLS>(rc.y, ctab, f, cx, cy, kind, delta) * 12, T);
      load_vec<2>(ll_meas + sg * 2, z2);
      load_vec<6>(xp + (size_t)max(row, 0) * 6, xv);
      if (ident) {
        Op[0] = Op[3] = 1.0;
        Op[1] = Op[2] = 0.0;
      } else {
        load_vec<4>(ll_omega + sg * 4, Op);
      }
      if (row >= 0) {
        BaEdgeLin L;
        ba_edge_linearize(T, X, z2, Op, f, cx, cy, kind, delta, true, L);
        double t0 = 0.0, t1 = 0.0;   // B x_p
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          t0 += L.B[0 + 2 * a] * xv[a];
          t1 += L.B[1 + 2 * a] * xv[a];
        }
        const double w = L.w;
        const double u0 = w * (L.O[0] * t0 + L.O[2] * t1), u1 = w * (L.O[1] * t0 + L.O[3] * t1);
#pragma unroll
        for (int j = 0; j < 3; ++j) c[j] = -(L.A[0 + 2 * j] * u0 + L.A[1 + 2 * j] * u1);
      }
    }
    for (int d = 1; d < tl.w; d <<= 1) {
      const bool take = pos + d < K;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double o = __shfl_down(c[i], d);
        if (take) c[i] += o;
      }
    }
    if (head) {
      const int lm = l0 + lmi;
      double cl[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) cl[j] = c[j] + bl[(size_t)lm * 3 + j];
      const double* D = Dinv + (size_t)lm * 9;
#pragma unroll
      for (int i = 0; i < 3; ++i) xl[(size_t)lm * 3 + i] = D[i] * cl[0] + D[i + 3] * cl[1] + D[i + 6] * cl[2];
    }
  }
}

// Pose side: G lanes per free pose walk its observations, re-evaluate the pose Jacobian on the fly
// and reduce Hpp_ii / b_i with DPP butterflies.
#ifndef G2OHIP_ASMP_OCC
#define G2OHIP_ASMP_OCC 2
#endif
template <int G, bool CLS = false>
__global__ void __launch_bounds__(kThreads, G2OHIP_ASMP_OCC) ba_assemble_poses_kernel(
    int nP, const int* __restrict__ vptr, const double* __restrict__ cams, const double* __restrict__ pts,
    const int* __restrict__ cam_pm, const int* __restrict__ pt_pm, const double* __restrict__ meas_pm, const double* __restrict__ omega_pm,
    double f, double cx, double cy, int kind, double delta, double* __restrict__ Hpp, const int* __restrict__ diag_blk,
    double* __restrict__ bp, int accumulate, int ident, const int* __restrict__ act, const double* __restrict__ ctab) {
  // act != nullptr: nP poses of a list (a rank of a sharded job holds observations of a fraction of the poses; the blocks
  // of the others are zero from build_structure on and nobody writes them)
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  const int vi_ = gt / G, g = gt % G;
  const bool active = vi_ < nP;
  const int v = (act && active) ? act[vi_] : vi_;
  // the pose block B' (w Omega) B is symmetric: its upper triangle (21 values, (a, c) with a <= c at c (c + 1) / 2 + a) is
  // accumulated and summed over the lane group, the store mirrors it -- 15 accumulators (30 registers), 30 multiply-adds per
  // observation and a third of the butterfly less
  double H[21], b[6];
#pragma unroll
  for (int i = 0; i < 21; ++i) H[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = 0.0;
  const int k0 = active ? vptr[v] : 0, k1 = active ? vptr[v + 1] : 0;
  double T[12];
  // Software pipeline over this lane's observations (pose-major arrays: streaming reads; the point is a dependent
  // gather): the point index is requested two observations ahead, the point, the measurement and the information
  // matrix one ahead -- with ~2 waves per SIMD nothing else hides the two round trips per observation.
  const int kf = k0 + g;
  if (kf < k1) load_vec<12>(cams + (size_t)(cam_pm[kf] & (CLS ? 0xffffff : -1)) * 12, T);
  int pt1 = kf < k1 ? pt_pm[kf] : 0, pt2 = kf + G < k1 ? pt_pm[kf + G] : 0;
  double Xn[3] = {0.0, 0.0, 0.0}, z2n[2] = {0.0, 0.0}, Opn[4] = {1.0, 0.0, 0.0, 1.0};
  if (kf < k1) {
    const double* Xp = pts + (size_t)pt1 * 3;
    Xn[0] = Xp[0]; Xn[1] = Xp[1]; Xn[2] = Xp[2];
    load_vec<2>(meas_pm + (size_t)kf * 2, z2n);
    if (!ident) load_vec<4>(omega_pm + (size_t)kf * 4, Opn);
  }
  for (int k = kf; k < k1; k += G) {
    double X[3], z2[2], Op[4];
    X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
    z2[0] = z2n[0]; z2[1] = z2n[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) Op[i] = Opn[i];
    {
      const int kn = k + G, kn2 = k + 2 * G;
      const int pt3 = kn2 < k1 ? pt_pm[kn2] : 0;
      if (kn < k1) {
        const double* Xp = pts + (size_t)pt2 * 3;
        Xn[0] = Xp[0]; Xn[1] = Xp[1]; Xn[2] = Xp[2];
        load_vec<2>(meas_pm + (size_t)kn * 2, z2n);
        if (!ident) load_vec<4>(omega_pm + (size_t)kn * 4, Opn);
      }
      pt2 = pt3;
    }
    if constexpr (CLS) (void)ba_edge_class<true>(cam_pm[k], ctab, f, cx, cy, kind, delta);   // (a pose's observations share the camera, not the class)
    BaEdgeLin L;
    ba_edge_linearize(T, X, z2, Op, f, cx, cy, kind, delta, false, L);
    const double w = L.w;
    const double O00 = w * L.O[0], O10 = w * L.O[1], O01 = w * L.O[2], O11 = w * L.O[3];
    const double Or0 = O00 * L.r[0] + O01 * L.r[1], Or1 = O10 * L.r[0] + O11 * L.r[1];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double OB0 = O00 * L.B[0 + 2 * c] + O01 * L.B[1 + 2 * c], OB1 = O10 * L.B[0 + 2 * c] + O11 * L.B[1 + 2 * c];
      b[c] -= L.B[0 + 2 * c] * Or0 + L.B[1 + 2 * c] * Or1;
#pragma unroll
      for (int a = 0; a <= c; ++a) H[c * (c + 1) / 2 + a] += L.B[0 + 2 * a] * OB0 + L.B[1 + 2 * a] * OB1;
    }
  }
  if (G > 1) {
#pragma unroll
    for (int i = 0; i < 21; ++i) H[i] = group_sum<G>(H[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = group_sum<G>(b[i]);
  }
  if (!active) return;
  double* Hd = Hpp + (size_t)diag_blk[v] * 36;
  double* bd = bp + (size_t)v * 6;
#pragma unroll
  for (int i = 0; i < 36; ++i)
    if (G == 1 || (i % G) == g) {
      const int r_ = i % 6, c_ = i / 6, lo = r_ < c_ ? r_ : c_, hi = r_ < c_ ? c_ : r_;
      const double hv = H[hi * (hi + 1) / 2 + lo];
      Hd[i] = accumulate ? Hd[i] + hv : hv;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (G == 1 || (i % G) == g) bd[i] = accumulate ? bd[i] + b[i] : b[i];
}


// ---------------------------------------------------------------------------------------------------
// Pose-graph front end: EdgeSE2 / EdgeSE3 error + Jacobian producers and the vertex updates on the
// device, so that a Gauss-Newton / LM iteration of a pose graph needs no host round trip.
//   EdgeSE2::computeError / linearizeOplus     g2o/types/slam2d/edge_se2.h:51-57, edge_se2.cpp:76-99
//   VertexSE2::oplusImpl                        g2o/types/slam2d/vertex_se2.h:55-59, se2.h:92-98
//   EdgeSE3::computeError / linearizeOplus      g2o/types/slam3d/edge_se3.cpp:48-75 (analytic Jacobian of
//                                               isometry3d_gradients.h:39-126, dq_dR of dquat2mat.cpp)
//   VertexSE3::oplusImpl                        g2o/types/slam3d/vertex_se3.h:107-116 (fromVectorMQT)
// SE2 estimates are (x, y, theta); SE3 estimates / measurements are isometries T[12] = R (column-major) | t.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double pg_normalize_theta(double theta) {
  if (theta >= -M_PI && theta < M_PI) return theta;
  const double multiplier = floor(theta / (2 * M_PI));
  theta = theta - multiplier * 2 * M_PI;
  if (theta >= M_PI) theta -= 2 * M_PI;
  if (theta < -M_PI) theta += 2 * M_PI;
  return theta;
}
__device__ __forceinline__ void pg_se2_inverse(const double* a, double* r) {
  const double th = pg_normalize_theta(-a[2]), c = cos(th), s = sin(th);
  r[0] = c * (-a[0]) - s * (-a[1]);
  r[1] = s * (-a[0]) + c * (-a[1]);
  r[2] = th;
}
__device__ __forceinline__ void pg_se2_mul(const double* a, const double* b, double* r) {
  const double c = cos(a[2]), s = sin(a[2]);
  const double x = a[0] + c * b[0] - s * b[1], y = a[1] + s * b[0] + c * b[1];
  r[0] = x;
  r[1] = y;
  r[2] = pg_normalize_theta(a[2] + b[2]);
}
__global__ void __launch_bounds__(kThreads) pg_se2_linearize_kernel(int n, const double* __restrict__ poses, const int* __restrict__ vi,
                                                                  const int* __restrict__ vj, const double* __restrict__ meas,
                                                                  double* __restrict__ J0, double* __restrict__ J1,
                                                                  double* __restrict__ err, int jac) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const double* xi = poses + 3 * (size_t)vi[k];
  const double* xj = poses + 3 * (size_t)vj[k];
  double invm[3], invi[3], t1[3], delta[3];
  pg_se2_inverse(meas + 3 * (size_t)k, invm);
  pg_se2_inverse(xi, invi);
  pg_se2_mul(invi, xj, t1);
  pg_se2_mul(invm, t1, delta);
  err[3 * (size_t)k] = delta[0];
  err[3 * (size_t)k + 1] = delta[1];
  err[3 * (size_t)k + 2] = delta[2];
  if (!jac) return;
  const double thetai = xi[2], dtx = xj[0] - xi[0], dty = xj[1] - xi[1];
  const double si = sin(thetai), ci = cos(thetai);
  const double A[9] = {-ci, -si, -si * dtx + ci * dty, si, -ci, -ci * dtx - si * dty, 0, 0, -1};   // row-major
  const double B[9] = {ci, si, 0, -si, ci, 0, 0, 0, 1};
  const double cz = cos(invm[2]), sz = sin(invm[2]);
  const double Z[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double* o0 = J0 + 9 * (size_t)k;
  double* o1 = J1 + 9 * (size_t)k;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double a = 0, b = 0;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        a += Z[r * 3 + m] * A[m * 3 + c];
        b += Z[r * 3 + m] * B[m * 3 + c];
      }
      o0[r + 3 * c] = a;
      o1[r + 3 * c] = b;
    }
}
__global__ void __launch_bounds__(kThreads) pg_se2_update_kernel(int nv, double* __restrict__ poses, const int* __restrict__ hidx,
                                                               const double* __restrict__ x) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv || hidx[v] < 0) return;
  const double* u = x + 3 * (size_t)hidx[v];
  double* p = poses + 3 * (size_t)v;
  p[0] += u[0];
  p[1] += u[1];
  p[2] = pg_normalize_theta(p[2] + u[2]);
}

#define PG_R(T, a, b) (T)[(a) + 3 * (b)]
__device__ __forceinline__ void pg_iso_mul(const double* A, const double* B, double* C) {
  double R[9], t[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      double s = 0;
#pragma unroll
      for (int m = 0; m < 3; ++m) s += PG_R(A, r, m) * PG_R(B, m, c);
      R[r + 3 * c] = s;
    }
#pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = PG_R(A, r, 0) * B[9] + PG_R(A, r, 1) * B[10] + PG_R(A, r, 2) * B[11] + A[9 + r];
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = R[i];
  C[9] = t[0];
  C[10] = t[1];
  C[11] = t[2];
}
__device__ __forceinline__ void pg_iso_inv(const double* A, double* C) {
  double R[9], t[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) R[r + 3 * c] = PG_R(A, c, r);
#pragma unroll
  for (int r = 0; r < 3; ++r) t[r] = -(R[r] * A[9] + R[r + 3] * A[10] + R[r + 6] * A[11]);
#pragma unroll
  for (int i = 0; i < 9; ++i) C[i] = R[i];
  C[9] = t[0];
  C[10] = t[1];
  C[11] = t[2];
}
// Quaterniond(w,x,y,z).toRotationMatrix() (no normalisation, like Eigen)
__device__ __forceinline__ void pg_quat_to_R(double w, double x, double y, double z, double* R) {
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
  PG_R(R, 0, 0) = 1 - (tyy + tzz); PG_R(R, 0, 1) = txy - twz; PG_R(R, 0, 2) = txz + twy;
  PG_R(R, 1, 0) = txy + twz; PG_R(R, 1, 1) = 1 - (txx + tzz); PG_R(R, 1, 2) = tyz - twx;
  PG_R(R, 2, 0) = txz - twy; PG_R(R, 2, 1) = tyz + twx; PG_R(R, 2, 2) = 1 - (txx + tyy);
}
// Quaterniond(R) (four cases), normalised, w >= 0 (isometry3d_mappings.cpp:38-44); q = (x, y, z, w)
__device__ void pg_R_to_quat(const double* R, double* q) {
  const double tr = PG_R(R, 0, 0) + PG_R(R, 1, 1) + PG_R(R, 2, 2);
  if (tr > 0) {
    double t = sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (PG_R(R, 2, 1) - PG_R(R, 1, 2)) * t;
    q[1] = (PG_R(R, 0, 2) - PG_R(R, 2, 0)) * t;
    q[2] = (PG_R(R, 1, 0) - PG_R(R, 0, 1)) * t;
  } else {
    int i = 0;
    if (PG_R(R, 1, 1) > PG_R(R, 0, 0)) i = 1;
    if (PG_R(R, 2, 2) > PG_R(R, i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(PG_R(R, i, i) - PG_R(R, j, j) - PG_R(R, k, k) + 1.0);
    double qq[4];
    qq[i] = 0.5 * t;
    t = 0.5 / t;
    qq[3] = (PG_R(R, k, j) - PG_R(R, j, k)) * t;
    qq[j] = (PG_R(R, j, i) + PG_R(R, i, j)) * t;
    qq[k] = (PG_R(R, k, i) + PG_R(R, i, k)) * t;
    q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2]; q[3] = qq[3];
  }
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] /= nrm;
  if (q[3] < 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = -q[i];
  }
}
// d(qx,qy,qz)/d vec(R) (3 x 9, vec column-major; D row-major): partials of the case formulas of dquat2mat.cpp:9-43
__device__ void pg_dq_dR(const double* R, double* D) {
  for (int i = 0; i < 27; ++i) D[i] = 0;
#define PG_COL(a, b) ((a) + 3 * (b))
  const double r00 = PG_R(R, 0, 0), r11 = PG_R(R, 1, 1), r22 = PG_R(R, 2, 2);
  const double tr = r00 + r11 + r22;
  double qw;
  if (tr > 0) {
    const double w = 0.5 * sqrt(tr + 1.0);
    qw = w;
    const double num[3] = {PG_R(R, 2, 1) - PG_R(R, 1, 2), PG_R(R, 0, 2) - PG_R(R, 2, 0), PG_R(R, 1, 0) - PG_R(R, 0, 1)};
    const int pa[3] = {2, 0, 1}, pb[3] = {1, 2, 0};
    for (int c = 0; c < 3; ++c) {
      const double dd = -num[c] / (32.0 * w * w * w);
      D[c * 9 + PG_COL(0, 0)] = dd;
      D[c * 9 + PG_COL(1, 1)] = dd;
      D[c * 9 + PG_COL(2, 2)] = dd;
      D[c * 9 + PG_COL(pa[c], pb[c])] = 0.25 / w;
      D[c * 9 + PG_COL(pb[c], pa[c])] = -0.25 / w;
    }
  } else {
    int i = 0;
    if ((r00 > r11) & (r00 > r22)) i = 0;
    else if (r11 > r22) i = 1;
    else i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    const double s = 0.5 * sqrt(1.0 + PG_R(R, i, i) - PG_R(R, j, j) - PG_R(R, k, k));
    qw = (PG_R(R, k, j) - PG_R(R, j, k)) / (4.0 * s);
    D[i * 9 + PG_COL(i, i)] = 1.0 / (8.0 * s);
    D[i * 9 + PG_COL(j, j)] = -1.0 / (8.0 * s);
    D[i * 9 + PG_COL(k, k)] = -1.0 / (8.0 * s);
    const int other[2] = {j, k};
    for (int o = 0; o < 2; ++o) {
      const int c = other[o];
      const double num = PG_R(R, c, i) + PG_R(R, i, c);
      D[c * 9 + PG_COL(c, i)] += 0.25 / s;
      D[c * 9 + PG_COL(i, c)] += 0.25 / s;
      const double dd = num / (32.0 * s * s * s);
      D[c * 9 + PG_COL(i, i)] += -dd;
      D[c * 9 + PG_COL(j, j)] += dd;
      D[c * 9 + PG_COL(k, k)] += dd;
    }
  }
  if (qw <= 0)
    for (int i = 0; i < 27; ++i) D[i] = -D[i];
#undef PG_COL
}
__global__ void __launch_bounds__(kThreads) pg_se3_linearize_kernel(int n, const double* __restrict__ poses, const int* __restrict__ vi,
                                                                  const int* __restrict__ vj, const double* __restrict__ meas,
                                                                  double* __restrict__ J0, double* __restrict__ J1,
                                                                  double* __restrict__ err, int jac) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const double* Xi = poses + 12 * (size_t)vi[e];
  const double* Xj = poses + 12 * (size_t)vj[e];
  const double* Z = meas + 12 * (size_t)e;
  double A[12], Xii[12], B[12], E[12], q[4];
  pg_iso_inv(Z, A);
  pg_iso_inv(Xi, Xii);
  pg_iso_mul(Xii, Xj, B);
  pg_iso_mul(A, B, E);
  pg_R_to_quat(E, q);
  double* r = err + 6 * (size_t)e;
  r[0] = E[9]; r[1] = E[10]; r[2] = E[11]; r[3] = q[0]; r[4] = q[1]; r[5] = q[2];
  if (!jac) return;
  double* Ji = J0 + 36 * (size_t)e;   // column-major 6x6
  double* Jj = J1 + 36 * (size_t)e;
  for (int i = 0; i < 36; ++i) Ji[i] = Jj[i] = 0;
  const double* Ra = A;
  const double* Rab = E;
  const double* Rbc = B;
  const double* tbc = B + 9;
  for (int c = 0; c < 3; ++c)
    for (int rr = 0; rr < 3; ++rr) {
      Ji[rr + 6 * c] = -PG_R(Ra, rr, c);
      Jj[rr + 6 * c] = PG_R(Rab, rr, c);
    }
  {  // dte/dqi = Ra * skewT(tbc) (doubled components); dte/dqj = 0
    const double x = 2 * tbc[0], y = 2 * tbc[1], z = 2 * tbc[2];
    double S[9];
    PG_R(S, 0, 0) = 0; PG_R(S, 0, 1) = -z; PG_R(S, 0, 2) = y;
    PG_R(S, 1, 0) = z; PG_R(S, 1, 1) = 0; PG_R(S, 1, 2) = -x;
    PG_R(S, 2, 0) = -y; PG_R(S, 2, 1) = x; PG_R(S, 2, 2) = 0;
    for (int c = 0; c < 3; ++c)
      for (int rr = 0; rr < 3; ++rr) {
        double s = 0;
        for (int m = 0; m < 3; ++m) s += PG_R(Ra, rr, m) * PG_R(S, m, c);
        Ji[rr + 6 * (3 + c)] = s;
      }
  }
  double D[27];
  pg_dq_dR(E, D);
  {  // dre/dqi
    const double r11 = 2 * PG_R(Rbc, 0, 0), r12 = 2 * PG_R(Rbc, 0, 1), r13 = 2 * PG_R(Rbc, 0, 2), r21 = 2 * PG_R(Rbc, 1, 0),
                 r22 = 2 * PG_R(Rbc, 1, 1), r23 = 2 * PG_R(Rbc, 1, 2), r31 = 2 * PG_R(Rbc, 2, 0), r32 = 2 * PG_R(Rbc, 2, 1),
                 r33 = 2 * PG_R(Rbc, 2, 2);
    const double S[3][9] = {{0, 0, 0, r31, r32, r33, -r21, -r22, -r23}, {-r31, -r32, -r33, 0, 0, 0, r11, r12, r13},
                            {r21, r22, r23, -r11, -r12, -r13, 0, 0, 0}};   // row-wise
    for (int a = 0; a < 3; ++a) {
      double M[9];
      for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) {
          double s = 0;
          for (int m = 0; m < 3; ++m) s += PG_R(Ra, rr, m) * S[a][m * 3 + c];
          M[rr + 3 * c] = s;
        }
      for (int rr = 0; rr < 3; ++rr) {
        double s = 0;
        for (int m = 0; m < 9; ++m) s += D[rr * 9 + m] * M[m];
        Ji[(3 + rr) + 6 * (3 + a)] = s;
      }
    }
  }
  {  // dre/dqj (Rc = I)
    const double S[3][9] = {{0, 0, 0, 0, 0, -2, 0, 2, 0}, {0, 0, 2, 0, 0, 0, -2, 0, 0}, {0, -2, 0, 2, 0, 0, 0, 0, 0}};
    for (int a = 0; a < 3; ++a) {
      double M[9];
      for (int c = 0; c < 3; ++c)
        for (int rr = 0; rr < 3; ++rr) {
          double s = 0;
          for (int m = 0; m < 3; ++m) s += PG_R(Rab, rr, m) * S[a][m * 3 + c];
          M[rr + 3 * c] = s;
        }
      for (int rr = 0; rr < 3; ++rr) {
        double s = 0;
        for (int m = 0; m < 9; ++m) s += D[rr * 9 + m] * M[m];
        Jj[(3 + rr) + 6 * (3 + a)] = s;
      }
    }
  }
}
// estimate <- estimate * fromVectorMQT(update)
__global__ void __launch_bounds__(kThreads) pg_se3_update_kernel(int nv, double* __restrict__ poses, const int* __restrict__ hidx,
                                                               const double* __restrict__ x) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv || hidx[v] < 0) return;
  const double* u = x + 6 * (size_t)hidx[v];
  double* T = poses + 12 * (size_t)v;
  double inc[12];
  const double w = 1 - (u[3] * u[3] + u[4] * u[4] + u[5] * u[5]);
  if (w < 0) {
    for (int i = 0; i < 9; ++i) inc[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    pg_quat_to_R(sqrt(w), u[3], u[4], u[5], inc);
  }
  inc[9] = u[0]; inc[10] = u[1]; inc[11] = u[2];
  double Tl[12], out[12];
  for (int i = 0; i < 12; ++i) Tl[i] = T[i];
  pg_iso_mul(Tl, inc, out);
  for (int i = 0; i < 12; ++i) T[i] = out[i];
}
#undef PG_R

inline int grid_for(size_t n, int threads = kThreads) { return (int)((n + threads - 1) / threads); }

// ---- dispatch tables ----------------------------------------------------------------
template <int D, int DV>
void launch_vertex(int G, int nV, const int* vptr, const int* vent, const EdgeSet& es, double* H, const int* diag_blk, double* b,
                   int accumulate, hipStream_t st) {
  if (nV == 0) return;
#define G2OHIP_LV(GG)                                                                                                    \
  hipLaunchKernelGGL((assemble_vertex_kernel<D, DV, GG>), dim3(grid_for((size_t)nV * GG)), dim3(kThreads), 0, st, nV, vptr, \
                     vent, es.J0, es.J1, es.omega, es.err, es.kernel_kind, es.delta, H, diag_blk, b, accumulate, es.rk.p)
  if (G <= 1)
    G2OHIP_LV(1);
  else if (G <= 4)
    G2OHIP_LV(4);
  else
    G2OHIP_LV(8);
#undef G2OHIP_LV
}

void dispatch_vertex(int D, int DV, int G, int nV, const int* vptr, const int* vent, const EdgeSet& es, double* H,
                     const int* diag_blk, double* b, int accumulate, hipStream_t st) {
#define G2OHIP_CASE(d_, v_) \
  if (D == d_ && DV == v_) return launch_vertex<d_, v_>(G, nV, vptr, vent, es, H, diag_blk, b, accumulate, st)
  G2OHIP_CASE(2, 2);
  G2OHIP_CASE(2, 3);
  G2OHIP_CASE(2, 6);
  G2OHIP_CASE(3, 2);
  G2OHIP_CASE(3, 3);
  G2OHIP_CASE(3, 6);
  G2OHIP_CASE(6, 6);
  G2OHIP_CASE(7, 7);
  G2OHIP_CASE(1, 3);
  G2OHIP_CASE(1, 6);
  G2OHIP_CASE(1, 2);   // bearing-only observations of 2D points (EdgeSE2PointBearing)
  G2OHIP_CASE(2, 7);   // projections seen from similarity poses (EdgeSim3ProjectXYZ, BlockSolver_7_3)
  G2OHIP_CASE(3, 7);
#undef G2OHIP_CASE
  throw ArgFailure("unsupported (error_dim, vertex_dim) = (" + std::to_string(D) + "," + std::to_string(DV) + ")");
}

void dispatch_offdiag(int D, int DR, int DC, int nDst, const int* dst, const int* ptr, const int* ent, const EdgeSet& es, double* H,
                      int accumulate, hipStream_t st) {
  if (nDst == 0) return;
#define G2OHIP_CASE(d_, r_, c_)                                                                                         \
  if (D == d_ && DR == r_ && DC == c_) {                                                                                \
    hipLaunchKernelGGL((assemble_offdiag_kernel<d_, r_, c_>), dim3(grid_for(nDst)), dim3(kThreads), 0, st, nDst, dst, ptr, \
                       ent, es.J0, es.J1, es.omega, es.err, es.kernel_kind, es.delta, H, accumulate, es.rk.p);          \
    return;                                                                                                             \
  }
  G2OHIP_CASE(2, 3, 2);
  G2OHIP_CASE(2, 6, 3);
  G2OHIP_CASE(3, 3, 3);
  G2OHIP_CASE(3, 6, 3);
  G2OHIP_CASE(6, 6, 6);
  G2OHIP_CASE(7, 7, 7);
  G2OHIP_CASE(2, 3, 3);
  G2OHIP_CASE(2, 6, 6);
  G2OHIP_CASE(3, 6, 6);
  G2OHIP_CASE(1, 3, 2);
  G2OHIP_CASE(1, 6, 3);
  G2OHIP_CASE(2, 7, 3);
  G2OHIP_CASE(3, 7, 3);
#undef G2OHIP_CASE
  throw ArgFailure("unsupported off-diagonal block shape (d,rows,cols) = (" + std::to_string(D) + "," + std::to_string(DR) + "," +
                   std::to_string(DC) + ")");
}

int pick_group(double avg) { return avg <= 2.0 ? 1 : (avg <= 16.0 ? 4 : 8); }

// find row r in column c of a sorted block-CCS pattern
inline int find_block(const std::vector<int>& colptr, const std::vector<int>& row, int c, int r) {
  auto b = row.begin() + colptr[c], e = row.begin() + colptr[c + 1];
  auto it = std::lower_bound(b, e, r);
  if (it == e || *it != r) return -1;
  return (int)(it - row.begin());
}

void keys_to_ccs(std::vector<long long>& keys, long long N, int ncols, std::vector<int>& colptr, std::vector<int>& row) {
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  colptr.assign(ncols + 1, 0);
  row.resize(keys.size());
  for (size_t i = 0; i < keys.size(); ++i) {
    colptr[(int)(keys[i] / N) + 1]++;
    row[i] = (int)(keys[i] % N);
  }
  for (int c = 0; c < ncols; ++c) colptr[c + 1] += colptr[c];
}

// The same from a GENERATOR of (column, row) pairs that is run twice (count, then fill): no array of keys and no global sort --
// the 15 pose pairs of each of a million landmarks are bucketed by column and made unique there with a stamp per row (the Schur
// pattern of the metric configuration: 0.75 s as one std::sort of 15.5 M keys, 0.1 s this way).  Rows ascending per column.
template <class Gen>
void pairs_to_ccs(Gen&& gen, int nrows, int ncols, std::vector<int>& colptr, std::vector<int>& row) {
  std::vector<size_t> start((size_t)ncols + 1, 0);
  gen([&](int c, int) { ++start[(size_t)c + 1]; });
  for (int c = 0; c < ncols; ++c) start[c + 1] += start[c];
  std::vector<int> bucket(start[ncols]);
  {
    std::vector<size_t> w(start.begin(), start.end() - 1);
    gen([&](int c, int r) { bucket[w[c]++] = r; });
  }
  std::vector<int> stamp((size_t)std::max(nrows, 1), -1);
  colptr.assign((size_t)ncols + 1, 0);
  row.clear();
  for (int c = 0; c < ncols; ++c) {
    const size_t b0 = row.size();
    for (size_t k = start[c]; k < start[c + 1]; ++k) {
      const int r = bucket[k];
      if (stamp[r] != c) {
        stamp[r] = c;
        row.push_back(r);
      }
    }
    std::sort(row.begin() + b0, row.end());
    colptr[c + 1] = (int)row.size();
  }
}

// group (dest, payload) pairs by dest into CSR over [0, ndst)
void group_by(int ndst, const std::vector<int>& dest, const std::vector<int>& payload, std::vector<int>& ptr, std::vector<int>& ent) {
  ptr.assign(ndst + 1, 0);
  for (int d : dest) ptr[d + 1]++;
  for (int k = 0; k < ndst; ++k) ptr[k + 1] += ptr[k];
  ent.resize(dest.size());
  std::vector<int> w(ptr.begin(), ptr.end() - 1);
  for (size_t k = 0; k < dest.size(); ++k) ent[w[dest[k]]++] = payload[k];  // stable: edge order kept per destination
}

}  // namespace

// One-time per process: dynamic-LDS limit of the fused BA tile kernels.  Called when a BA front end is bound (set-up time) so
// that the first solve of a process does not pay for it, and again at the launch site as a no-op.
static void prepare_ba_tile_kernels() {
  static bool attr = false;
  if (!attr) {
#define G2OHIP_BA_TILE_ATTR(GG)                                                                                                    \
  (void)hipFuncSetAttribute((const void*)ba_schur_tile_kernel<GG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
  (void)hipFuncSetAttribute((const void*)ba_schur_tile_kernel<GG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    (void)hipFuncSetAttribute((const void*)ba_schur_tile_kernel<8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ba_schur_tile_kernel<8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ba_schur_tile_kernel<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ba_schur_tile_kernel<1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    G2OHIP_BA_TILE_ATTR(1);
    G2OHIP_BA_TILE_ATTR(2);
    G2OHIP_BA_TILE_ATTR(4);
    G2OHIP_BA_TILE_ATTR(8);
    G2OHIP_BA_TILE_ATTR(16);
#undef G2OHIP_BA_TILE_ATTR
    attr = true;
  }
}

// =====================================================================================
// BlockSolver
// =====================================================================================
BlockSolver::BlockSolver(int p, int l, int device) : p_(p), l_(l), device_(device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw HipFailure("no HIP device available: libg2ohip has no CPU fallback");
  if (device < 0 || device >= count) throw ArgFailure("bad device ordinal");
  G2OHIP_HIP_CHECK(hipSetDevice(device));
  G2OHIP_HIP_CHECK(hipStreamCreate(&st_));
  own_stream_ = true;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) num_cus_ = prop.multiProcessorCount;
    // The code objects are gfx950 only, and the in-launch hand-offs (relaxed agent-scope counters around sc1 data, in-order
    // workgroup dispatch) are reasoned for that target: refuse anything else instead of failing at the first launch.
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      throw HipFailure(std::string("libg2ohip is built for gfx950 (MI355X); device reports ") + prop.gcnArchName);
  }
  if (!(p == 3 || p == 6 || p == 7)) throw ArgFailure("pose_dim must be 3, 6 or 7");
  if (!(l == 2 || l == 3 || l == 0)) throw ArgFailure("landmark_dim must be 2 or 3");
}

BlockSolver::~BlockSolver() {
  invalidate_graphs();
  if (fetch_st_) {
    (void)hipStreamSynchronize(fetch_st_);
    (void)hipStreamDestroy(fetch_st_);
    (void)hipEventDestroy(fetch_fork_);
    for (int k = 0; k < kFetchMaxPieces; ++k) (void)hipEventDestroy(fetch_ev_[k]);
  }
  if (trial_ev_) {
    (void)hipEventSynchronize(trial_ev_);
    (void)hipEventDestroy(trial_ev_);
  }
  if (side_) (void)hipStreamDestroy(side_);
  if (side_fork_) (void)hipEventDestroy(side_fork_);
  if (side_join_) (void)hipEventDestroy(side_join_);
  if (own_stream_ && st_) (void)hipStreamDestroy(st_);
  if (h_trial_) (void)hipHostFree(h_trial_);
}

void BlockSolver::set_stream(hipStream_t st) {
  invalidate_graphs();
  if (own_stream_ && st_) (void)hipStreamDestroy(st_);
  own_stream_ = false;
  st_ = st;
}

void BlockSolver::init() {
  invalidate_graphs();
  // block_solver.hpp:606-620: numeric + symbolic state is rebuilt on the next buildStructure/solve
  if (chol_) chol_->reset();
  system_built_ = false;
}

// A new graph behind the same handle (a second optimize() of a g2o optimizer calls buildStructure again, the online
// growth re-registers everything): every edge set, the extra pattern of the reduced system and the bindings of the device
// front ends are dropped.
void BlockSolver::clear_edge_sets() {
  invalidate_graphs();
  if (chol_) chol_->reset();
  sets_.clear();
  extra_hs_.clear();
  ba_ = BaFrontEnd();
  pg_ = PgFrontEnd();
  structured_ = false;
  system_built_ = false;
}

int BlockSolver::add_edge_set(int d, int n, const int* v0, const int* v1) {
  if (d <= 0 || d > 7 || n < 0 || !v0) throw ArgFailure("add_edge_set: bad arguments");
  auto es = std::make_unique<EdgeSet>();
  es->d = d;
  es->n = n;
  es->unary = (v1 == nullptr);
  es->v0.assign(v0, v0 + n);
  if (v1) es->v1.assign(v1, v1 + n);
  else es->v1.assign(n, -1);
  sets_.push_back(std::move(es));
  structured_ = false;
  return (int)sets_.size() - 1;
}

// An edge set that stands for ONE PAIR of vertices of n-ary edges (BaseMultiEdge, base_multi_edge.hpp:170-222: H_ii and b_i once
// per vertex, H_ij once per pair i < j, chi2 once per edge): the caller registers a binary set per pair and switches off what
// another pair of the same edges already contributes.  Bits: 1 / 2 = no diagonal block and no right-hand side for vertex 0 /
// vertex 1, 4 = not counted in chi2.
void BlockSolver::set_edge_set_parts(int set, int parts) {
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("set_edge_set_parts: bad edge set id");
  if (parts < 0 || parts > 7) throw ArgFailure("set_edge_set_parts: parts is a combination of the bits 1, 2, 4");
  if (sets_[set]->unary && (parts & 2)) throw ArgFailure("set_edge_set_parts: a unary set has no vertex 1");
  if (sets_[set]->parts != parts) {
    sets_[set]->parts = parts;
    structured_ = false;
    chi2_valid_ = false;
  }
}

// Solver::updateStructure (block_solver.hpp:297-351): online growth of a system WITHOUT Schur complement -- new pose
// vertices at the end of the index mapping, new edges appended to an existing edge set.  The reference allocates the new
// blocks in Hpp and leaves the symbolic factorisation to the linear solver's next solve; here the structure (contributor
// lists, pattern, ordering, symbolic analysis) is rebuilt from the enlarged topology.  Returns false where the reference
// aborts (marginalised vertices, :313-316).
bool BlockSolver::update_structure(int new_poses, int set, int n, const int* v0, const int* v1) {
  require_structure();
  if (new_poses < 0 || n < 0 || (n > 0 && !v0)) throw ArgFailure("update_structure: bad arguments");
  if (schur_ || nL_ > 0) return false;
  if (n > 0) {
    if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("update_structure: no such edge set");
    EdgeSet& es = *sets_[set];
    if (es.unary != (v1 == nullptr)) throw ArgFailure("update_structure: the set's edges have a different number of vertices");
    const int nP_new = nP_ + new_poses;
    for (int k = 0; k < n; ++k)
      if (v0[k] >= nP_new || (v1 && v1[k] >= nP_new) || v0[k] < -1 || (v1 && v1[k] < -1)) throw ArgFailure("update_structure: vertex index out of range");
    es.v0.insert(es.v0.end(), v0, v0 + n);
    if (v1) es.v1.insert(es.v1.end(), v1, v1 + n);
    else es.v1.insert(es.v1.end(), n, -1);
    es.n += n;
    es.has_data = es.has_err = false;          // the per-edge arrays are sized for the old edge count: set_edge_data again
    es.J0 = es.J1 = es.omega = es.err = nullptr;
    es.external = false;
    // a device front end bound to the set holds vi / vj / measurements and the own_* arrays for the OLD edge count:
    // drop the binding (pg_set_edges / ba_set_edges again after growth), pg_linearize refuses until then
    if (set == pg_.set) pg_ = PgFrontEnd();
    if (set == ba_.set) ba_.set = -1;
    // per-edge robust kernels cover the old edge count: the new edges get "none" (kind 0) until set_robust_kernel_per_edge is
    // called again -- the array is extended on the device so that no kernel reads past its end
    if (es.rk.p) {
      DevBuf<double> grown;
      grown.alloc((size_t)es.n * 2);
      G2OHIP_HIP_CHECK(hipMemsetAsync(grown.p, 0, (size_t)es.n * 2 * sizeof(double), st_));
      G2OHIP_HIP_CHECK(hipMemcpyAsync(grown.p, es.rk.p, (size_t)(es.n - n) * 2 * sizeof(double), hipMemcpyDeviceToDevice, st_));
      G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
      es.rk = std::move(grown);
    }
  }
  build_structure(nP_ + new_poses, 0, false);
  return true;
}

void BlockSolver::add_schur_pattern(int n, const int* rows, const int* cols) {
  if (n < 0 || (n > 0 && (!rows || !cols))) throw ArgFailure("add_schur_pattern: bad arguments");
  for (int k = 0; k < n; ++k) {
    if (rows[k] < 0 || cols[k] < rows[k]) throw ArgFailure("add_schur_pattern: need 0 <= row <= col");
    extra_hs_.emplace_back(rows[k], cols[k]);
  }
  structured_ = false;
}

void BlockSolver::require_structure() const {
  if (!structured_) throw StateFailure("call g2ohip_build_structure first");
}

void BlockSolver::build_structure(int nP, int nL, bool do_schur) {
  invalidate_graphs();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  nP_ = nP;
  nL_ = nL;
  schur_ = do_schur && nL > 0;
  if (nP <= 0) throw ArgFailure("build_structure: need at least one free pose");
  if (nL > 0 && l_ == 0) throw ArgFailure("landmarks present but landmark_dim == 0");
  const int p = p_, l = l_;
  auto is_lm = [&](int v) { return v >= nP; };
  // G2OHIP_SETUP_TIMING=1: where the host side of the set-up goes (stderr, seconds per section)
  const bool lapt = getenv("G2OHIP_SETUP_TIMING") != nullptr;
  auto lap_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double lap_t = lap_now();
  auto lap = [&](const char* what) {
    if (!lapt) return;
    const double t = lap_now();
    fprintf(stderr, "build_structure: %-34s %.3f s\n", what, t - lap_t);
    lap_t = t;
  };
  std::unique_ptr<SparseCholesky> new_chol;   // (Schur mode: analysed on a thread of its own, below)
  std::thread analysis_thread;
  std::exception_ptr analysis_error;
  struct Joiner {
    std::thread& t;
    ~Joiner() { if (t.joinable()) t.join(); }   // (an exception on this thread must not leave the other one running into freed vectors)
  } analysis_joiner{analysis_thread};
  std::thread lists_thread;
  std::exception_ptr lists_error;
  Joiner lists_joiner{lists_thread};
  // ---- validate sets, vertex classes
  for (auto& esp : sets_) {
    EdgeSet& es = *esp;
    int c0 = -1, c1 = -1;
    for (int k = 0; k < es.n; ++k) {
      int a = es.v0[k], b = es.v1[k];
      if (a >= nP + nL || b >= nP + nL || a < -1 || b < -1) throw ArgFailure("edge index out of range");
      if (a >= 0) {
        int c = is_lm(a);
        if (c0 >= 0 && c0 != c) throw ArgFailure("edge set mixes vertex classes on side 0");
        c0 = c;
      }
      if (b >= 0) {
        int c = is_lm(b);
        if (c1 >= 0 && c1 != c) throw ArgFailure("edge set mixes vertex classes on side 1");
        c1 = c;
      }
      if (a >= 0 && b >= 0 && is_lm(a) && is_lm(b)) throw ArgFailure("landmark-landmark edges are not supported (block_solver.hpp:383)");
    }
    es.dim0 = (c0 == 1) ? l : p;
    es.dim1 = es.unary ? 0 : ((c1 == 1) ? l : p);
  }
  lap("validate");
  // ---- Hpp / Hpl patterns (block_solver.hpp:178-254)
  std::vector<long long> kpp, kpl;
  kpp.reserve(nP);
  for (int i = 0; i < nP; ++i) kpp.push_back((long long)i * nP + i);
  for (auto& esp : sets_) {
    EdgeSet& es = *esp;
    for (int k = 0; k < es.n; ++k) {
      int a = es.v0[k], b = es.v1[k];
      if (a < 0 || b < 0) continue;
      if (!is_lm(a) && !is_lm(b)) {
        int r = std::min(a, b), c = std::max(a, b);
        kpp.push_back((long long)c * nP + r);
      } else {
        int pose = is_lm(a) ? b : a, lm = (is_lm(a) ? a : b) - nP;
        kpl.push_back((long long)lm * nP + pose);
      }
    }
  }
  keys_to_ccs(kpp, nP, nP, pp_colptr, pp_row);
  if (nL > 0) keys_to_ccs(kpl, nP, nL, pl_colptr, pl_row);
  else {
    pl_colptr.assign(1, 0);
    pl_row.clear();
  }
  kpp.clear();
  kpp.shrink_to_fit();
  kpl.clear();
  kpl.shrink_to_fit();
  const int pp_nnzb = (int)pp_row.size(), pl_nnzb = (int)pl_row.size();
  pp_diag.resize(nP);
  for (int c = 0; c < nP; ++c) pp_diag[c] = find_block(pp_colptr, pp_row, c, c);
  lap("Hpp / Hpl patterns");
  // ---- contributor lists per set: they need the two patterns and nothing else, and nothing below needs them -- on a thread of
  // their own next to the Schur pattern and the tiles' set-up (option setup_overlap)
  const bool lists_threaded = setup_overlap && schur_;
  auto build_lists = [&, pp_nnzb, pl_nnzb, lists_threaded] {
    double lt = lap_now();
    auto llap = [&](const char* what) {
      if (!lapt) return;
      const double t = lap_now();
      fprintf(stderr, "build_structure: %-34s %.3f s%s\n", what, t - lt, lists_threaded ? " (on a thread of its own)" : "");
      lt = t;
    };
    // ---- contributor lists per set
    bool seen_pose = false, seen_lm = false, seen_op = false, seen_ol = false;
    std::vector<int> offdiag_blocks;
    offdiag_blocks.reserve(pp_nnzb - nP);
    for (int c = 0; c < nP; ++c)
      for (int q = pp_colptr[c]; q < pp_colptr[c + 1]; ++q)
        if (pp_row[q] != c) offdiag_blocks.push_back(q);
    for (auto& esp : sets_) {
      EdgeSet& es = *esp;
      std::vector<int> dp, pp_, dl, pl_, dop, pop, dol, pol;
      {
        // (5 M push_backs each and as many block searches at the metric configuration: contiguous chunks of the edge list on the
        // host threads, every chunk into its own lists, concatenated in chunk order -- the edge order of the sequential loop)
        struct Part { std::vector<int> v[8]; };
        const size_t nch = std::max<size_t>(1, std::min<size_t>((size_t)host_threads(), ((size_t)es.n + 65535) / 65536));
        std::vector<Part> parts(nch);
        host_parallel_chunks((size_t)es.n, nch, [&](size_t c, size_t kb, size_t ke) {
          std::vector<int>&dp = parts[c].v[0], &pp_ = parts[c].v[1], &dl = parts[c].v[2], &pl_ = parts[c].v[3], &dop = parts[c].v[4],
                          &pop = parts[c].v[5], &dol = parts[c].v[6], &pol = parts[c].v[7];
          for (int i = 0; i < 8; ++i) parts[c].v[i].reserve(ke - kb);
          for (int k = (int)kb; k < (int)ke; ++k) {
            int a = es.v0[k], b = es.v1[k];
            if (a >= 0 && !(es.parts & 1)) {   // (parts: the diagonal block / right-hand side of this side come from another set)
              if (is_lm(a)) { dl.push_back(a - nP); pl_.push_back(k << 1); }
              else { dp.push_back(a); pp_.push_back(k << 1); }
            }
            if (b >= 0 && !(es.parts & 2)) {
              if (is_lm(b)) { dl.push_back(b - nP); pl_.push_back((k << 1) | 1); }
              else { dp.push_back(b); pp_.push_back((k << 1) | 1); }
            }
            if (a >= 0 && b >= 0) {
              if (!is_lm(a) && !is_lm(b)) {
                int tr = a > b;
                int q = find_block(pp_colptr, pp_row, std::max(a, b), std::min(a, b));
                dop.push_back(q);
                pop.push_back((k << 1) | tr);
              } else {
                int pose = is_lm(a) ? b : a, lm = (is_lm(a) ? a : b) - nP;
                int tr = is_lm(a) ? 1 : 0;  // vertex 0 marginalized -> write transposed (block_solver.hpp:240-244)
                dol.push_back(find_block(pl_colptr, pl_row, lm, pose));
                pol.push_back((k << 1) | tr);
              }
            }
          }
        });
        std::vector<int>* outs[8] = {&dp, &pp_, &dl, &pl_, &dop, &pop, &dol, &pol};
        for (int i = 0; i < 8; ++i) {
          if (nch == 1) {
            outs[i]->swap(parts[0].v[i]);
            continue;
          }
          size_t tot = 0;
          for (size_t c = 0; c < nch; ++c) tot += parts[c].v[i].size();
          outs[i]->reserve(tot);
          for (size_t c = 0; c < nch; ++c) outs[i]->insert(outs[i]->end(), parts[c].v[i].begin(), parts[c].v[i].end());
        }
      }
      llap("contributor lists: edge walk");
      es.touches_pose = !dp.empty();
      es.touches_lm = !dl.empty();
      std::vector<int> ptr, ent;
      if (es.touches_pose) {
        group_by(nP, dp, pp_, ptr, ent);
        es.vp_ptr.upload(ptr, st_);
        es.vp_ent.upload(ent, st_);
        {
          std::vector<int> act;
          for (int v = 0; v < nP; ++v)
            if (ptr[v + 1] > ptr[v]) act.push_back(v);
          es.n_vp_act = (int)act.size() < nP ? (int)act.size() : 0;   // (0: every pose has entries, no list needed)
          if (es.n_vp_act > 0) {   // the poses WITHOUT entries follow the active ones (their blocks are re-zeroed per build_system)
            for (int v = 0; v < nP; ++v)
              if (ptr[v + 1] == ptr[v]) act.push_back(v);
            es.vp_act.upload(act, st_);
          }
        }
        es.h_vp_ent = ent;
        es.n_vp_ent = (long)ent.size();
        es.first_pose = !seen_pose;
        seen_pose = true;
      }
      if (es.touches_lm) {
        group_by(nL, dl, pl_, ptr, ent);
        es.vl_ptr.upload(ptr, st_);
        es.vl_ent.upload(ent, st_);
        es.h_vl_ent = ent;
        es.h_vl_ptr = ptr;
        es.n_vl_ent = (long)ent.size();
        es.first_lm = !seen_lm;
        seen_lm = true;
      }
      llap("contributor lists: per vertex");
      if (!dop.empty()) {
        es.first_op = !seen_op;
        seen_op = true;
        // destination list: all off-diagonal blocks for the first pose-pose set (so every block is written),
        // only the touched ones afterwards
        std::vector<int> dst_list;
        if (es.first_op) dst_list = offdiag_blocks;
        else {
          dst_list = dop;
          std::sort(dst_list.begin(), dst_list.end());
          dst_list.erase(std::unique(dst_list.begin(), dst_list.end()), dst_list.end());
        }
        std::vector<int> local(dop.size());
        host_parallel_for(dop.size(), [&](size_t b_, size_t e_) {
          for (size_t k = b_; k < e_; ++k) local[k] = (int)(std::lower_bound(dst_list.begin(), dst_list.end(), dop[k]) - dst_list.begin());
        });
        group_by((int)dst_list.size(), local, pop, ptr, ent);
        es.n_op = (int)dst_list.size();
        es.op_dst.upload(dst_list, st_);
        es.op_ptr.upload(ptr, st_);
        es.op_ent.upload(ent, st_);
      }
      if (!dol.empty()) {
        es.first_ol = !seen_ol;
        seen_ol = true;
        std::vector<int> dst_list;
        if (es.first_ol) {
          dst_list.resize(pl_nnzb);
          std::iota(dst_list.begin(), dst_list.end(), 0);
        } else {
          dst_list = dol;
          std::sort(dst_list.begin(), dst_list.end());
          dst_list.erase(std::unique(dst_list.begin(), dst_list.end()), dst_list.end());
        }
        std::vector<int> local(dol.size());
        if (es.first_ol) {
          local = dol;   // (the destination list is every block in order: the position of a block is the block)
        } else {
          host_parallel_for(dol.size(), [&](size_t b_, size_t e_) {
            for (size_t k = b_; k < e_; ++k) local[k] = (int)(std::lower_bound(dst_list.begin(), dst_list.end(), dol[k]) - dst_list.begin());
          });
        }
        group_by((int)dst_list.size(), local, pol, ptr, ent);
        es.n_ol = (int)dst_list.size();
        es.ol_dst.upload(dst_list, st_);
        es.ol_ptr.upload(ptr, st_);
        es.ol_ent.upload(ent, st_);
      }
      es.has_data = false;
    }
    llap("contributor lists: per off-diagonal block");
  };
  if (lists_threaded) {
    lists_thread = std::thread([this, build_lists, &lists_error] {   // (the closure by value: it is declared after the joiner)
      try {
        G2OHIP_HIP_CHECK(hipSetDevice(device_));
        build_lists();
      } catch (...) {
        lists_error = std::current_exception();
      }
    });
  } else {
    build_lists();
    lap_t = lap_now();
  }
  // ---- Schur structure (block_solver.hpp:256-292)
  n_sc_ = 0;
  if (schur_) {
    for (auto& rc : extra_hs_)
      if (rc.second >= nP || rc.second < 0 || rc.first < 0 || rc.first >= nP) throw ArgFailure("add_schur_pattern: block index out of range");
    pairs_to_ccs(
        [&](auto&& emit) {   // (column, row): the pattern of Hpp, every pose pair of every landmark, the caller's extra blocks
          for (int c = 0; c < nP; ++c)
            for (int q = pp_colptr[c]; q < pp_colptr[c + 1]; ++q) emit(c, pp_row[q]);
          for (int c = 0; c < nL; ++c)
            for (int q1 = pl_colptr[c]; q1 < pl_colptr[c + 1]; ++q1)
              for (int q2 = q1; q2 < pl_colptr[c + 1]; ++q2) emit(pl_row[q2], pl_row[q1]);
          for (auto& rc : extra_hs_) emit(rc.second, rc.first);
        },
        nP, nP, hs_colptr, hs_row);
    lap("Schur pattern");
    // The symbolic analysis of the reduced system (nested dissection, supernodes, frontal matrices, launch plans: 0.08 s at the
    // metric configuration) needs nothing but this pattern: it runs on a thread of its own next to the Schur tiles' set-up below.
    new_chol = std::make_unique<SparseCholesky>(p);
    new_chol->opt = chol_opt;
    if (setup_overlap) {
      analysis_thread = std::thread([&] {
        try {
          G2OHIP_HIP_CHECK(hipSetDevice(device_));
          new_chol->analyze(nP, hs_colptr.data(), hs_row.data(), st_);
        } catch (...) {
          analysis_error = std::current_exception();
        }
      });
    }
    const int hs_nnzb = (int)hs_row.size();
    rd_ptr_h_.clear();
    rd_slot_h_.clear();
    std::vector<int> hs_src(hs_nnzb, -1);
    for (int c = 0; c < nP; ++c)
      for (int q = pp_colptr[c]; q < pp_colptr[c + 1]; ++q) hs_src[find_block(hs_colptr, hs_row, c, pp_row[q])] = q;
    {
      std::vector<int> hs_diag(hs_nnzb, -1), pose_diag(std::max(nP, 1), 0);
      for (int c = 0; c < nP; ++c) {
        pose_diag[c] = find_block(hs_colptr, hs_row, c, c);
        hs_diag[pose_diag[c]] = c;
      }
      d_hs_diag.upload(hs_diag, st_);
      d_pose_diag.upload(pose_diag, st_);
      hs_diag_h_ = hs_diag;
    }
    d_hs_src.upload(hs_src, st_);
    hs_src_h_ = hs_src;
    // ---- landmark-range tiles for the Schur outer products (schur_tile_kernel)
    {
      // (DP: Dinv and its symmetric split C | signs, schur_tile_prepare)
      const size_t PL = (size_t)p * l, DP = (((size_t)l * l + 1) & ~(size_t)1) + (((size_t)l * l + l + 1) & ~(size_t)1), BP = ((size_t)l + 1) & ~(size_t)1;
      std::vector<int> tile_lm0, tile_td0, td_dest, td_ptr, te_pack;
      std::vector<unsigned short> slot_lm(pl_row.size() + 1, 0);   // landmark of an Hpl block, relative to its tile (0 for a split landmark)
      std::vector<unsigned short> te_lm;
      std::vector<int> rd_cnt(hs_nnzb, 0);
      std::vector<int> hs_row_is_diag(hs_nnzb, 0);
      for (int c = 0; c < nP; ++c) hs_row_is_diag[find_block(hs_colptr, hs_row, c, c)] = 1;
      tile_lm0.push_back(0);
      tile_td0.push_back(0);
      td_ptr.push_back(0);
      size_t max_lds = 0;
      int lm = 0;
      struct Ent { int dest, pack; unsigned short lml; };
      // per tile: landmark range, first block range, optional SECOND block range (split landmarks, below)
      std::vector<int> t_l0, t_l1, t_q0, t_ns, t_q1, t_n1;
      // Three passes: (A) the tile boundaries -- a sequential greedy walk over the landmarks that only counts --, (B) the entry
      // lists of the tiles on the host threads (15 M entries, a block search and a sort key each, at the metric configuration:
      // a tile is independent of the others), (C) the concatenation in tile order.
      struct TileDesc { int l0, l1, q0, ns, q1, n1; bool split, same; int a0, b0, na, nb2; };
      std::vector<TileDesc> descs;
      n_split_tiles_ = 0;
      while (lm < nL) {
        // greedy tile: as many landmarks as fit the LDS budget
        int l0 = lm;
        size_t bytes = 0;
        {
          // A landmark seen by so many poses that its blocks and pair list alone exceed the budget (K(K+1)/2 entries:
          // ~64 poses at 39 KB) is SPLIT: its observation list is cut into chunks, one tile per pair of chunks (A <= B)
          // stages the blocks of A and of B (two block ranges) and carries the pairs (a in A, b in B).  Every tile
          // inverts the landmark's block itself (the same bits) and the diagonal chunk pairs carry the right-hand side.
          const size_t K = pl_colptr[lm + 1] - pl_colptr[lm];
          const size_t alone = K * PL * 8 + DP * 8 + BP * 8 + K * (K + 1) / 2 * (6 + 8);
          if (alone > schur_tile_bytes && K > 2) {
            size_t C = 1;
            while ((2 * (C + 1)) * PL * 8 + DP * 8 + BP * 8 + (C + 1) * (C + 1) * (6 + 8) <= schur_tile_bytes) ++C;
            C = std::max<size_t>(C, 1);
            if (C >= 32768) throw ArgFailure("Schur tile budget too large for 16-bit slot indices");
            const int qb = pl_colptr[lm];
            for (size_t a0 = 0; a0 < K; a0 += C)
              for (size_t b0 = a0; b0 < K; b0 += C) {
                const int na = (int)std::min(C, K - a0), nb2 = (int)std::min(C, K - b0);
                const bool same = a0 == b0;
                const size_t nents = same ? (size_t)na * (na + 1) / 2 : (size_t)na * nb2;
                const size_t tb = (size_t)(na + (same ? 0 : nb2)) * PL * 8 + DP * 8 + BP * 8 + nents * (6 + 8);
                max_lds = std::max(max_lds, tb + 64);
                descs.push_back(TileDesc{lm, lm + 1, qb + (int)a0, na, same ? 0 : qb + (int)b0, same ? 0 : nb2, true, same, (int)a0, (int)b0, na, nb2});
                ++n_split_tiles_;
              }
            ++lm;
            continue;
          }
        }
        while (lm < nL) {
          size_t K = pl_colptr[lm + 1] - pl_colptr[lm];
          size_t add = K * PL * 8 + DP * 8 + BP * 8 + K * (K + 1) / 2 * (6 + 8);   // blocks, Dinv, b_l, entries (+ per-destination metadata bound)
          bool ok16 = (size_t)(pl_colptr[lm + 1] - pl_colptr[l0]) < 65536 && (lm + 1 - l0) < 65536;
          if (lm > l0 && (bytes + add > schur_tile_bytes || !ok16)) break;
          bytes += add;
          ++lm;
        }
        if ((size_t)(pl_colptr[lm] - pl_colptr[l0]) >= 65536) throw ArgFailure("a landmark is observed by >= 65536 poses: unsupported");
        max_lds = std::max(max_lds, bytes + 64);
        const int q0 = pl_colptr[l0];
        descs.push_back(TileDesc{l0, lm, q0, pl_colptr[lm] - q0, 0, 0, false, false, 0, 0, 0, 0});
        tile_lm0.push_back(lm);
      }
      struct TileOut {
        std::vector<int> dest, cnt, pack;   // destinations of the tile (longest run first), entries per destination, entries
        std::vector<unsigned short> lml;
      };
      std::vector<TileOut> outs(descs.size());
      const bool sort_dests = schur_sort_dests;
      host_parallel_for(descs.size(), [&](size_t tb_, size_t te_) {
        std::vector<Ent> ents;
        std::vector<int> order;
        std::vector<std::pair<int, int>> runs;   // (begin, end) into order
        for (size_t ti = tb_; ti < te_; ++ti) {
          const TileDesc& td = descs[ti];
          ents.clear();
          if (td.split) {
            const int qb = td.q0 - td.a0;
            for (int a = 0; a < td.na; ++a)
              for (int b = td.same ? a : 0; b < td.nb2; ++b) {
                const int qa = qb + td.a0 + a, qbb = qb + td.b0 + b;
                Ent e;
                e.dest = find_block(hs_colptr, hs_row, pl_row[qbb], pl_row[qa]);
                e.pack = a | ((td.same ? b : td.na + b) << 16);
                e.lml = 0;
                ents.push_back(e);
              }
          } else {
            for (int c = td.l0; c < td.l1; ++c)
              for (int q1 = pl_colptr[c]; q1 < pl_colptr[c + 1]; ++q1)
                for (int q2 = q1; q2 < pl_colptr[c + 1]; ++q2) {
                  Ent e;
                  e.dest = find_block(hs_colptr, hs_row, pl_row[q2], pl_row[q1]);
                  e.pack = (q1 - td.q0) | ((q2 - td.q0) << 16);
                  e.lml = (unsigned short)(c - td.l0);
                  ents.push_back(e);
                }
            for (int c = td.l0; c < td.l1; ++c)
              for (int q = pl_colptr[c]; q < pl_colptr[c + 1]; ++q) slot_lm[q] = (unsigned short)(c - td.l0);
          }
          order.resize(ents.size());
          std::iota(order.begin(), order.end(), 0);
          std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ents[a].dest < ents[b].dest; });  // landmark order kept per dest
          // runs of equal destination, longest first: the lane groups of one wave run in lockstep, so a wave should
          // hold destinations with similar entry counts (the order of the partials of a destination over the tiles,
          // and the entry order inside a destination, are unaffected: results do not change)
          runs.clear();
          for (size_t k = 0; k < order.size();) {
            size_t k2 = k + 1;
            while (k2 < order.size() && ents[order[k2]].dest == ents[order[k]].dest) ++k2;
            runs.emplace_back((int)k, (int)k2);
            k = k2;
          }
          if (sort_dests)
            std::stable_sort(runs.begin(), runs.end(), [](const std::pair<int, int>& x, const std::pair<int, int>& y) {
              return (x.second - x.first) > (y.second - y.first);
            });
          TileOut& o = outs[ti];
          o.dest.reserve(runs.size());
          o.cnt.reserve(runs.size());
          o.pack.reserve(ents.size());
          o.lml.reserve(ents.size());
          for (const auto& run : runs) {
            o.dest.push_back(ents[order[run.first]].dest);
            o.cnt.push_back(run.second - run.first);
            for (int k = run.first; k < run.second; ++k) {
              o.pack.push_back(ents[order[k]].pack);
              o.lml.push_back(ents[order[k]].lml);
            }
          }
        }
      }, 8);
      {
        size_t ntd = 0, nte = 0;
        for (const TileOut& o : outs) {
          ntd += o.dest.size();
          nte += o.pack.size();
        }
        td_dest.reserve(ntd);
        td_ptr.reserve(ntd + 1);
        te_pack.reserve(nte);
        te_lm.reserve(nte);
      }
      for (size_t ti = 0; ti < descs.size(); ++ti) {
        const TileDesc& td = descs[ti];
        TileOut& o = outs[ti];
        for (size_t r = 0; r < o.dest.size(); ++r) {
          td_dest.push_back(o.dest[r]);
          rd_cnt[o.dest[r]]++;
          td_ptr.push_back(td_ptr.back() + o.cnt[r]);
        }
        te_pack.insert(te_pack.end(), o.pack.begin(), o.pack.end());
        te_lm.insert(te_lm.end(), o.lml.begin(), o.lml.end());
        t_l0.push_back(td.l0); t_l1.push_back(td.l1); t_q0.push_back(td.q0); t_ns.push_back(td.ns); t_q1.push_back(td.q1); t_n1.push_back(td.n1);
        tile_td0.push_back((int)td_dest.size());
        std::vector<int>().swap(o.dest);   // (release as we go)
        std::vector<int>().swap(o.pack);
        std::vector<unsigned short>().swap(o.lml);
      }
      n_tiles_ = (int)t_l0.size();
      tile_lm0_h_ = tile_lm0;   // (tile boundaries: only meaningful while no landmark is split, n_split_tiles_ == 0)
      tiles_cover_all_ = true;
      n_td_ = (long)td_dest.size();
      n_sc_ = (long)te_pack.size();
      schur_lds_bytes_ = max_lds;
      if (getenv("G2OHIP_PLAN_DUMP"))
        fprintf(stderr, "schur tiles: %d tiles (LDS %zu B of %zu), %ld partial blocks for %d destinations = %.3f per destination\n", n_tiles_, (size_t)max_lds,
                (size_t)schur_tile_bytes, n_td_, hs_nnzb, hs_nnzb > 0 ? (double)n_td_ / hs_nnzb : 0.0);
      // per destination: its partial slots in tile order
      std::vector<int> rd_ptr(hs_nnzb + 1, 0), rd_slot(td_dest.size());
      for (int d = 0; d < hs_nnzb; ++d) rd_ptr[d + 1] = rd_ptr[d] + rd_cnt[d];
      rd_cnt_h_ = rd_cnt;
      {
        std::vector<int> w(rd_ptr.begin(), rd_ptr.end() - 1);
        for (size_t k = 0; k < td_dest.size(); ++k) rd_slot[w[td_dest[k]]++] = (int)k;
      }
      {
        // one 32-byte record per tile (l0, l1, first Hpl slot, slots, td0, td1, first entry, entries): a single
        // scalar load instead of three dependent ones
        std::vector<int> meta((size_t)n_tiles_ * 8);
        std::vector<int2> q2((size_t)std::max(n_tiles_, 1), make_int2(0, 0));
        for (int t = 0; t < n_tiles_; ++t) {
          int* m = &meta[(size_t)t * 8];
          m[0] = t_l0[t];
          m[1] = t_l1[t];
          m[2] = t_q0[t];
          m[3] = t_ns[t];
          q2[t] = make_int2(t_q1[t], t_n1[t]);   // second block range of a split landmark's tile (count 0: none)
          m[4] = tile_td0[t];
          m[5] = tile_td0[t + 1];
          m[6] = td_ptr[m[4]];
          m[7] = td_ptr[m[5]] - m[6];
        }
        d_tile_lm0.upload(meta, st_);
        d_tile_q2.upload(q2, st_);
      }
      d_tile_td0.upload(tile_td0, st_);
      {
        std::vector<int> td_diag(td_dest.size());
        for (size_t k = 0; k < td_dest.size(); ++k) td_diag[k] = hs_row_is_diag[td_dest[k]];
        d_td_diag.upload(td_diag, st_);
      }
      d_td_ptr.upload(td_ptr, st_);
      d_te_pack.upload(te_pack, st_);
      d_te_lm.upload(te_lm, st_);
      d_slot_lm.upload(slot_lm, st_);
      d_rd_ptr.upload(rd_ptr, st_);
      d_rd_slot.upload(rd_slot, st_);
      rd_ptr_h_ = rd_ptr;
      rd_slot_h_ = rd_slot;
      d_Pd.alloc((size_t)(std::max<long>(n_td_, 1) + 1) * p * p);   // + one all-zero block (slot n_td_): "no partial"
      d_Pd.zero(st_);
      d_Pr.alloc((size_t)std::max<long>(n_td_, 1) * p);
    }
    d_Hschur.alloc((size_t)hs_nnzb * p * p);
    d_Dinv.alloc((size_t)nL * l * l);
    d_db.alloc((size_t)nL * l);
    d_bschur.alloc((size_t)nP * p);
  } else {
    hs_colptr.clear();
    hs_row.clear();
  }
  // landmark id per Hpl block
  if (nL > 0) {
    std::vector<int> pl_lm(pl_nnzb);
    for (int c = 0; c < nL; ++c)
      for (int q = pl_colptr[c]; q < pl_colptr[c + 1]; ++q) pl_lm[q] = c;
    d_pl_lm.upload(pl_lm, st_);
    d_pl_colptr.upload(pl_colptr, st_);
    d_pl_row.upload(pl_row, st_);
    d_Hpl.alloc((size_t)pl_nnzb * p * l);
    d_Hll.alloc((size_t)nL * l * l);
    d_bkL.alloc((size_t)nL * l);
  }
  d_pp_diag.upload(pp_diag, st_);
  d_pp_colptr.upload(pp_colptr, st_);
  d_pp_row.upload(pp_row, st_);
  // multi-rank + Schur: room behind Hpp for the reduced-system blocks that are summed over the ranks (block d of Hschur at
  // pp_nnzb + d): the virtual source of the factorisation reads them there (exchange_setup, sharded_virtual)
  hpp_blocks_ = (size_t)pp_nnzb;
  d_Hpp.alloc(((size_t)pp_nnzb + ((schur_ && chol_opt.world > 1) ? hs_row.size() : 0)) * p * p);
  d_bkP.alloc((size_t)nP * p);
  d_b.alloc(vector_size());
  d_x.alloc(vector_size());
  d_red.alloc(4096);
  G2OHIP_HIP_CHECK(hipMemsetAsync(d_Hpp.p, 0, (size_t)pp_nnzb * p * p * sizeof(double), st_));
  if (nL > 0) {
    G2OHIP_HIP_CHECK(hipMemsetAsync(d_Hpl.p, 0, (size_t)pl_nnzb * p * l * sizeof(double), st_));
    G2OHIP_HIP_CHECK(hipMemsetAsync(d_Hll.p, 0, (size_t)nL * l * l * sizeof(double), st_));
  }
  G2OHIP_HIP_CHECK(hipMemsetAsync(d_b.p, 0, vector_size() * sizeof(double), st_));
  G2OHIP_HIP_CHECK(hipMemsetAsync(d_x.p, 0, vector_size() * sizeof(double), st_));
  lap("Schur tiles + uploads");
  if (lists_thread.joinable()) {
    lists_thread.join();
    if (lists_error) std::rethrow_exception(lists_error);
    lap("contributor lists (what the tiles' set-up did not hide)");
  }
  // ---- symbolic factorisation of the system the linear solver will see
  pcg_.reset();
  pcg_mf_.reset();
  pcg_hpp_.reset();
  mh_hpp_.reset();
  mf_ready_ = false;
  if (analysis_thread.joinable()) {
    analysis_thread.join();
    if (analysis_error) std::rethrow_exception(analysis_error);
    chol_ = std::move(new_chol);
  } else {
    chol_ = new_chol ? std::move(new_chol) : std::make_unique<SparseCholesky>(p);
    chol_->opt = chol_opt;
    if (schur_) chol_->analyze(nP, hs_colptr.data(), hs_row.data(), st_);
    else chol_->analyze(nP, pp_colptr.data(), pp_row.data(), st_);
  }
  // (the exchange of a sharded job was set up for the previous structure: its index lists address the old pattern and a merged
  // payload (sharded_merge) lives behind the previous Cholesky's exchange buffer -- gone with it)
  ex_ = Exchange();
  selftest_done_ = false;
  trial_.begun = false;   // (a read-back queued for the previous structure is dropped)
  lap("symbolic analysis (what the tiles' set-up did not hide)");
  n_active_ = -1;
  hschur_valid_ = true;
  {
    const double zero2[2] = {0.0, 0.0};
    d_lam.upload(zero2, 2, st_);
    lam_pose_ = lam_lm_ = 0.0;
    lam_mask_h_.clear();
  }
  if (schur_ && chol_opt.world == 1 && n_tiles_ > 0 && !rd_ptr_h_.empty()) {
    // lets solve() skip the reduction pass: the fronts are assembled from Hpp and the tiles' partial blocks
    SparseCholesky::VirtualBlocks vb;
    vb.base_idx = hs_src_h_.data();
    vb.is_diag = hs_diag_h_.data();
    vb.part_ptr = rd_ptr_h_.data();
    vb.part_slot = rd_slot_h_.data();
    vb.d_part_slot = d_rd_slot.p;
    vb.base = d_Hpp.p;
    vb.parts = d_Pd.p;
    vb.lam = d_lam.p;
    vb.zero_slot = (int)std::max<long>(n_td_, 1);
    vb.split = false;   // (set per solve: launch_schur_reduce)
    chol_->set_virtual_blocks(vb, st_);
  }
  if (chol_opt.world > 1) {
    // the rank that consumes a pose's diagonal block adds lambda to it (rank 0 for the shared ones)
    const std::vector<int>& cons = chol_->symbolic().block_consumer;
    const std::vector<int>& cp = schur_ ? hs_colptr : pp_colptr;
    std::vector<unsigned char> m(nP);
    for (int c = 0; c < nP; ++c) {
      const int o = cons[cp[c + 1] - 1];
      m[c] = (o == chol_opt.rank) || (o < 0 && chol_opt.rank == 0);
    }
    d_lam_mask.upload(m, st_);
    lam_mask_h_ = m;
    if (schur_) {
      // reduced-system blocks this rank has to form: the ones it consumes (own or shared fronts) and the ones
      // it contributes to; everything else stays zero (cleared below once)
      std::vector<int> act;
      for (int d = 0; d < (int)cons.size(); ++d)
        if (cons[d] == chol_opt.rank || cons[d] < 0 || (d < (int)rd_cnt_h_.size() && rd_cnt_h_[d] > 0)) act.push_back(d);
      n_active_ = (int)act.size();
      if (act.empty()) act.push_back(0);
      d_active.upload(act, st_);
      G2OHIP_HIP_CHECK(hipMemsetAsync(d_Hschur.p, 0, hs_row.size() * (size_t)p * p * sizeof(double), st_));
      G2OHIP_HIP_CHECK(hipMemsetAsync(d_bschur.p, 0, (size_t)nP * p * sizeof(double), st_));
    }
  }
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
  structured_ = true;
  system_built_ = false;
}

void BlockSolver::set_edge_data(int set, const double* J0, const double* J1, const double* omega, const double* err, bool on_device) {
  require_structure();
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("bad edge set id");
  EdgeSet& es = *sets_[set];
  if (!J0 || !omega || !err || (!es.unary && !J1)) throw ArgFailure("set_edge_data: null array");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  // Host arrays handed over again for a set that already has its own device copies (a g2o adapter's generic group, every
  // iteration): the device addresses stay, only what the kernels READ changes -- the captured launch sequences and the cached
  // evaluations of the device front ends (other sets) stay valid; chi2 does not.
  const double *pJ0 = es.J0, *pJ1 = es.J1, *pO = es.omega, *pE = es.err;
  const bool had = es.has_data && !es.external;
  es.external = on_device;
  if (on_device) {
    es.J0 = J0;
    es.J1 = J1;
    es.omega = omega;
    es.err = err;
  } else {
    const size_t n = (size_t)es.n;
    es.own_J0.upload(J0, n * es.d * es.dim0, st_);
    if (!es.unary) es.own_J1.upload(J1, n * es.d * es.dim1, st_);
    es.own_omega.upload(omega, n * es.d * es.d, st_);
    es.own_err.upload(err, n * es.d, st_);
    es.J0 = es.own_J0.p;
    es.J1 = es.unary ? nullptr : es.own_J1.p;
    es.omega = es.own_omega.p;
    es.err = es.own_err.p;
  }
  // (a set bound to a device front end holds that front end's evaluations: host arrays uploaded into it replace them, the cached
  // err_valid / jac_valid flags must go -- the full invalidation)
  const bool front_end_set = set == ba_.set || set == pg_.set;
  if (had && !on_device && !front_end_set && pJ0 == es.J0 && pJ1 == es.J1 && pO == es.omega && pE == es.err) chi2_valid_ = false;
  else invalidate_graphs();
  es.has_data = true;
  es.has_err = true;
}

// The errors of a set alone (computeActiveErrors at trial estimates: the Jacobians and information matrices stay what the system
// was built from).  Host array, the set's own device copy.
void BlockSolver::set_edge_errors(int set, const double* err) {
  require_structure();
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("bad edge set id");
  EdgeSet& es = *sets_[set];
  if (!err) throw ArgFailure("set_edge_errors: null array");
  if (!es.has_data || es.external || !es.own_err.p) throw StateFailure("set_edge_errors: the set has no host-supplied edge data (set_edge_data first)");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  es.own_err.upload(err, (size_t)es.n * es.d, st_);
  chi2_valid_ = false;
  es.has_err = true;
}

void BlockSolver::set_robust_kernel(int set, int kind, double delta) {
  invalidate_graphs();
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("bad edge set id");
  if (kind < 0 || kind > 5) throw ArgFailure("unsupported robust kernel");
  if (kind != 0 && !(delta > 0.0)) throw ArgFailure("robust kernel: delta must be positive");
  if (set == ba_.set && ba_.n_classes > 1)
    throw StateFailure("set_robust_kernel: the set is bound to the BA front end with edge classes, which carry their own kernels (ba_set_edges_classes)");
  sets_[set]->kernel_kind = kind;
  sets_[set]->delta = delta;
  sets_[set]->rk.release();   // (a set-level kernel replaces per-edge ones)
}

// Every edge its own robust kernel (in g2o the kernel is a member of the EDGE, optimizable_graph.h:436-443: a pose graph with
// kernels on its loop closures only is ONE homogeneous set of EdgeSE2 / EdgeSE3 here).  kind == nullptr: back to the set-level kernel.
void BlockSolver::set_robust_kernel_per_edge(int set, const int* kind, const double* delta) {
  invalidate_graphs();
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("bad edge set id");
  EdgeSet& es = *sets_[set];
  if (!kind) {
    es.rk.release();
    return;
  }
  if (!delta) throw ArgFailure("set_robust_kernel_per_edge: null delta array");
  if (set == ba_.set && ba_.set >= 0)
    throw StateFailure("set_robust_kernel_per_edge: the set is bound to the BA front end; per-edge kernels go through ba_set_edges_classes there");
  std::vector<double> h((size_t)es.n * 2);
  for (int k = 0; k < es.n; ++k) {
    if (kind[k] < 0 || kind[k] > 5) throw ArgFailure("unsupported robust kernel for edge " + std::to_string(k));
    if (kind[k] != 0 && !(delta[k] > 0.0)) throw ArgFailure("robust kernel: delta must be positive (edge " + std::to_string(k) + ")");
    h[2 * (size_t)k] = kind[k];
    h[2 * (size_t)k + 1] = delta[k];
  }
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (h.empty()) h.assign(2, 0.0);
  es.rk.upload(h, st_);
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
}

void BlockSolver::invalidate_graphs() {
  // (called by every setter that changes what the kernels read: also drops the cached evaluations)
  chi2_valid_ = false;
  ba_.err_valid = ba_.jac_valid = false;
  pg_.err_valid = pg_.jac_valid = false;
  drop_graph_segments();
}

void BlockSolver::drop_graph_segments() {   // (the captured launch sequences only: what the kernels read has not changed)
  for (GraphSeg& sg : segs_) {
    if (sg.e) (void)hipGraphExecDestroy(sg.e);
    if (sg.g) (void)hipGraphDestroy(sg.g);
    sg = GraphSeg();
  }
}

template <class F>
void BlockSolver::run_seg(int id, F&& body) {
  if (!use_graph || st_ == nullptr || in_outer_seg_) {
    body();
    return;
  }
  struct Outer {   // the whole-solve segment makes the segments inside it transparent
    bool& f; bool on;
    Outer(bool& f_, bool on_) : f(f_), on(on_) { if (on) f = true; }
    ~Outer() { if (on) f = false; }
  } outer(in_outer_seg_, id == kSegShardedAll && segs_[id].state >= 1);
  GraphSeg& sg = segs_[id];
  if (sg.state == 0) {   // first use: one-time initialisation (function attributes, lazy analysis) runs outside a capture
    body();
    sg.state = 1;
    return;
  }
  if (sg.state == 1) {
    // A capture that the runtime refuses (another library touching the stream, an unsupported node) is not an
    // error of the solve: the segment then runs as plain launches, for good.
    bool ok = hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      try {
        body();
      } catch (...) {
        hipGraph_t dead = nullptr;
        (void)hipStreamEndCapture(st_, &dead);
        if (dead) (void)hipGraphDestroy(dead);
        throw;
      }
      hipGraph_t g = nullptr;
      hipGraphExec_t e = nullptr;
      ok = hipStreamEndCapture(st_, &g) == hipSuccess && g != nullptr;
      if (ok) ok = hipGraphInstantiate(&e, g, nullptr, nullptr, 0) == hipSuccess;
      if (ok) {
        sg.g = g;
        sg.e = e;
        sg.state = 2;
      } else if (g) {
        (void)hipGraphDestroy(g);
      }
    }
    if (!ok) {
      (void)hipGetLastError();
      fprintf(stderr, "g2ohip: hipGraph capture of a launch sequence failed; continuing with plain launches\n");
      use_graph = false;
      body();
      return;
    }
  }
  G2OHIP_HIP_CHECK(hipGraphLaunch(sg.e, st_));
}

void BlockSolver::build_system() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  for (auto& esp : sets_)
    if (esp->n > 0 && !esp->has_data) throw StateFailure("build_system: edge data missing for a set");
  build_system_impl();   // a handful of kernels: launched plainly
  system_built_ = true;
}

// lanes per landmark of the fused BA assembly: one observation per lane where possible
int BlockSolver::ba_lm_group() const {
  const double avgK = (double)sets_[ba_.set]->n_vl_ent / std::max(1, nL_);
  return avgK <= 1.5 ? 1 : (avgK <= 4.0 ? 4 : 8);
}

// landmark side of the fused BA assembly: Hll, b_l, errors and -- when somebody will read it -- Hpl
void BlockSolver::launch_ba_landmarks(bool write_hpl) {
  EdgeSet& es = *sets_[ba_.set];
  const size_t sizeP = (size_t)nP_ * p_;
  const int GL = ba_lm_group();
#define G2OHIP_BA_LM_(GG, CC)                                                                                                    \
  hipLaunchKernelGGL((ba_assemble_landmarks_kernel<GG, CC>), dim3(grid_for((size_t)nL_ * GG)), dim3(kThreads), 0, st_, nL_, es.vl_ptr.p, \
                     es.vl_ent.p, ba_.cams.p, ba_.pts.p, ba_.cam_lm.p, ba_.pt_lm.p, ba_.meas_lm.p, ba_.omega_lm.p, ba_.hpl_lm.p, ba_.f, \
                     ba_.cx, ba_.cy, es.kernel_kind, es.delta, d_Hll.p, d_b.p + sizeP, d_Hpl.p, es.own_err.p, ba_.omega_identity ? 1 : 0, \
                     d_pl_colptr.p, write_hpl ? 1 : 0, ba_.ctab.p)
#define G2OHIP_BA_LM(GG) G2OHIP_BA_LM_(GG, false)
  if (ba_.n_classes > 1) {   // (edge classes: one lane-group width is instantiated)
    if (GL == 1) G2OHIP_BA_LM_(1, true);
    else G2OHIP_BA_LM_(8, true);
  } else if (GL == 1) G2OHIP_BA_LM(1);
  else if (GL == 4) G2OHIP_BA_LM(4);
  else G2OHIP_BA_LM(8);
#undef G2OHIP_BA_LM
#undef G2OHIP_BA_LM_
}

// May the fused BA assembly leave Hpl unwritten?  Only while its two readers on the solve path (Schur tiles,
// back-substitution) re-evaluate the Jacobians: direct solver, tiled Schur pass that covers every landmark.
bool BlockSolver::ba_skip_hpl_ok() const {
  return ba_skip_hpl && schur_ && p_ == 6 && l_ == 3 && linear_solver == 0 && n_tiles_ > 0 && tiles_cover_all_ && fuse_landmark_inverse &&
         ba_recompute_backsub && ba_.set >= 0 && ba_fused && ba_.fused_ok && ba_.cam_q.p != nullptr && [&] {
           for (size_t i = 0; i < sets_.size(); ++i)
             if ((int)i != ba_.set && sets_[i]->n > 0 && sets_[i]->touches_lm) return false;
           return true;
         }();
}

// May build_system leave the whole landmark side (Hll, b_l, the errors) to the Schur tiles of the solve?  They evaluate
// every observation of their landmarks anyway (ba_schur_tile_kernel<G, true>).
bool BlockSolver::ba_fuse_ll_ok() const { return ba_fuse_landmarks && ba_.ll_slots_ok && ba_skip_hpl_ok(); }

// somebody reads Hll, b_l or the errors between a build_system that left them to the solve and that solve
// (maxDiagonal of the first LM iteration, b(), multiplyHessian, chi2 without a linearisation)
void BlockSolver::ensure_ll() {
  if (ll_valid_ && !ll_hbm_partial_) return;
  if (!ba_recompute_ok())
    throw StateFailure("the landmark blocks were left to solve() by the last build_system and the estimates (or the robust kernel) have "
                       "changed since: call build_system again, or set option ba_fuse_landmarks = 0");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  launch_ba_landmarks(false);
  ll_valid_ = true;
  ll_hbm_partial_ = false;
  G2OHIP_HIP_CHECK(hipGetLastError());
}

// readers of b_l alone (b(), computeScale): the Schur tiles of the solve always leave it in memory
void BlockSolver::ensure_bl() {
  if (!ll_valid_) ensure_ll();
}

// somebody reads Hpl (copy_values, multiplyHessian, the matrix-free operator, a solve path without the fused kernels)
// after a build_system that skipped it
void BlockSolver::ensure_hpl() {
  if (hpl_valid_) return;
  if (!ba_recompute_ok())
    throw StateFailure("Hpl was not materialised by the last build_system and the estimates (or the robust kernel) have changed since: "
                       "call build_system again, or set option ba_skip_hpl = 0");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  launch_ba_landmarks(true);   // the same kernel with the Hpl stores on (Hll, b_l and the errors come out identical)
  hpl_valid_ = ll_valid_ = true;
  ll_hbm_partial_ = false;
  G2OHIP_HIP_CHECK(hipGetLastError());
}

// A rank of a sharded job assembles only the poses it has observations of; the diagonal blocks and b segments of the others
// must read as zero.  They are zero after build_structure and no kernel of the library writes them, but the compact pose
// kernel no longer heals them the way the full one did (round-3 advisor finding): re-zero them per build_system.
__global__ void zero_inactive_poses_kernel(int n, int p, const int* __restrict__ list, const int* __restrict__ pp_diag,
                                           double* __restrict__ Hpp, double* __restrict__ b) {
  const int per = p * p + p;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * per) return;
  const int v = list[t / per], e = (int)(t % per);
  if (e < p * p) Hpp[(size_t)pp_diag[v] * p * p + e] = 0.0;
  else b[(size_t)v * p + (e - p * p)] = 0.0;
}

// pose side of the fused BA assembly (Hpp diagonal blocks, b_p) on stream sp
void BlockSolver::launch_ba_poses(hipStream_t sp) {
  EdgeSet& es = *sets_[ba_.set];
  const int G = pick_group((double)es.n_vp_ent / std::max(1, nP_));
  // (a rank of a sharded job: only the poses it has observations of -- the only pose-side set, so nothing else ever
  // writes the blocks of the others)
  const bool compact = es.n_vp_act > 0 && sets_.size() == 1 && chol_opt.world > 1;
  const int nPk = compact ? es.n_vp_act : nP_;
  const int* pact = compact ? es.vp_act.p : (const int*)nullptr;
#define G2OHIP_BA_POSE_(GG, CC)                                                                                                  \
  hipLaunchKernelGGL((ba_assemble_poses_kernel<GG, CC>), dim3(grid_for((size_t)nPk * GG)), dim3(kThreads), 0, sp, nPk, es.vp_ptr.p,  \
                     ba_.cams.p, ba_.pts.p, ba_.cam_pm.p, ba_.pt_pm.p, ba_.meas_pm.p, ba_.omega_pm.p, ba_.f, ba_.cx, ba_.cy,       \
                     es.kernel_kind, es.delta, d_Hpp.p, d_pp_diag.p, d_b.p, es.first_pose ? 0 : 1, ba_.omega_identity ? 1 : 0,     \
                     pact, ba_.ctab.p)
#define G2OHIP_BA_POSE(GG) G2OHIP_BA_POSE_(GG, false)
  if (compact && es.first_pose)
    hipLaunchKernelGGL(zero_inactive_poses_kernel, dim3(grid_for((size_t)(nP_ - nPk) * (p_ * p_ + p_))), dim3(kThreads), 0, sp, nP_ - nPk, p_,
                       es.vp_act.p + nPk, d_pp_diag.p, d_Hpp.p, d_b.p);
  if (es.touches_pose) {
    static const int g_env = getenv("G2OHIP_POSE_GROUP") ? atoi(getenv("G2OHIP_POSE_GROUP")) : 0;   // (experiments)
    const int Gp = g_env > 0 ? g_env : G;
    if (ba_.n_classes > 1) G2OHIP_BA_POSE_(8, true);   // (edge classes: one lane-group width is instantiated)
    else if (Gp <= 1) G2OHIP_BA_POSE(1);
    else if (Gp <= 4) G2OHIP_BA_POSE(4);
    else if (Gp <= 8) G2OHIP_BA_POSE(8);
    else G2OHIP_BA_POSE(16);
  }
#undef G2OHIP_BA_POSE
#undef G2OHIP_BA_POSE_
}

void BlockSolver::ensure_side() {
  if (side_) return;
  G2OHIP_HIP_CHECK(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
  G2OHIP_HIP_CHECK(hipEventCreateWithFlags(&side_fork_, hipEventDisableTiming));
  G2OHIP_HIP_CHECK(hipEventCreateWithFlags(&side_join_, hipEventDisableTiming));
}

void BlockSolver::build_system_impl() {
  if (profiling) tq_.start(st_);
  ba_.sys_version = -1;   // (set again by the fused BA branch)
  hpl_valid_ = ll_valid_ = true;
  ll_hbm_partial_ = false;
  const size_t sizeP = (size_t)nP_ * p_;
  bool any_pose = false, any_lm = false;
  int set_index = -1;
  for (auto& esp : sets_) {
    EdgeSet& es = *esp;
    ++set_index;
    if (es.n == 0) continue;
    if (set_index == ba_.set && ba_.n_classes > 1 && !(ba_fused && ba_.fused_ok && ba_.n_cams > 0))
      throw StateFailure("build_system: a BA edge set with edge classes needs the fused path (option ba_fused) and its estimates");
    if (set_index == ba_.set && ba_fused && ba_.fused_ok && ba_.n_cams > 0) {
      // fused EdgeProjectXYZ2UV path: errors + Jacobians evaluated inside the assembly kernels
      ba_.sys_version = ba_.est_version;
      ba_.sys_kind = es.kernel_kind;
      ba_.sys_delta = es.delta;
      // The two sides write disjoint arrays (Hll / b_l / Hpl and Hpp / b_p): the pose side runs on a side stream
      // next to the landmark side unless one of them is being timed on its own.
      const bool overlap = overlap_assembly && es.touches_pose && !prof.timing(KernelProf::kAsmLandmark) && !prof.timing(KernelProf::kAsmPose);
      if (overlap) {
        ensure_side();
        G2OHIP_HIP_CHECK(hipEventRecord(side_fork_, st_));             // fork before either kernel is queued
        G2OHIP_HIP_CHECK(hipStreamWaitEvent(side_, side_fork_, 0));
      }
      hpl_valid_ = !ba_skip_hpl_ok();
      ll_valid_ = !ba_fuse_ll_ok();   // false: the Schur tiles of the solve assemble the landmark side
      if (ll_valid_) {
        prof.begin(KernelProf::kAsmLandmark, st_);
        launch_ba_landmarks(hpl_valid_);
        prof.end(KernelProf::kAsmLandmark, st_);
      }
      prof.begin(KernelProf::kAsmPose, st_);
      launch_ba_poses(overlap ? side_ : st_);
      if (overlap) {
        G2OHIP_HIP_CHECK(hipEventRecord(side_join_, side_));
        G2OHIP_HIP_CHECK(hipStreamWaitEvent(st_, side_join_, 0));
      }
      prof.end(KernelProf::kAsmPose, st_);
      continue;
    }
    if (es.touches_pose) {
      int G = pick_group((double)es.n_vp_ent / std::max(1, nP_));
      prof.begin(KernelProf::kAsmPose, st_);
      dispatch_vertex(es.d, p_, G, nP_, es.vp_ptr.p, es.vp_ent.p, es, d_Hpp.p, d_pp_diag.p, d_b.p, es.first_pose ? 0 : 1, st_);
      prof.end(KernelProf::kAsmPose, st_);
      any_pose = true;
    }
    if (es.touches_lm) {
      int G = pick_group((double)es.n_vl_ent / std::max(1, nL_));
      prof.begin(KernelProf::kAsmLandmark, st_);
      dispatch_vertex(es.d, l_, G, nL_, es.vl_ptr.p, es.vl_ent.p, es, d_Hll.p, nullptr, d_b.p + sizeP, es.first_lm ? 0 : 1, st_);
      prof.end(KernelProf::kAsmLandmark, st_);
      any_lm = true;
    }
    if (es.n_op > 0) {
      prof.begin(KernelProf::kAsmOffPP, st_);
      dispatch_offdiag(es.d, p_, p_, es.n_op, es.op_dst.p, es.op_ptr.p, es.op_ent.p, es, d_Hpp.p, es.first_op ? 0 : 1, st_);
      prof.end(KernelProf::kAsmOffPP, st_);
    }
    if (es.n_ol > 0) {
      prof.begin(KernelProf::kAsmOffPL, st_);
      dispatch_offdiag(es.d, p_, l_, es.n_ol, es.ol_dst.p, es.ol_ptr.p, es.ol_ent.p, es, d_Hpl.p, es.first_ol ? 0 : 1, st_);
      prof.end(KernelProf::kAsmOffPL, st_);
    }
  }
  // classes nobody touches stay zero (cleared once in build_structure)
  (void)any_pose;
  (void)any_lm;
  G2OHIP_HIP_CHECK(hipGetLastError());
  if (profiling) {
    tq_.stop(st_);
    times.quadratic = tq_.seconds();
  }
}

double BlockSolver::reduce_sum_finish(int nblocks) {
  std::vector<double> h(nblocks);
  d_red.download(h.data(), nblocks, st_);
  double s = 0.0;
  for (double v : h) s += v;  // fixed order: deterministic
  return s;
}

double BlockSolver::chi2() {
  require_structure();
  if (chi2_valid_) return chi2_value_;   // same errors, same kernels as the last evaluation (LM asks twice per accepted step)
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (ba_.set >= 0 && ba_.n_classes > 1 && !ba_.err_valid) ba_linearize(false);   // (edge classes: the per-class kernels live in the BA kernels only)
  if ((!ll_valid_ || ll_hbm_partial_) && !ba_.err_valid) ensure_ll();   // errors come from a linearisation or from the landmark side of the assembly
  double total = 0.0;
  for (auto& esp : sets_) {
    EdgeSet& es = *esp;
    if (es.n == 0 || (es.parts & 4)) continue;   // (parts bit 4: the edges of this set are counted by another one)
    if (!es.has_err) throw StateFailure("chi2: edge data missing");
    int nblocks = std::min(1024, grid_for(es.n));
    if (ba_.err_valid && esp.get() == sets_[ba_.set].get()) {   // partial sums left by ba_linearize (same values trial_stats reads)
      std::vector<double> h(1024);
      G2OHIP_HIP_CHECK(hipMemcpyAsync(h.data(), d_red_multi.p + (size_t)(ba_.set + 1) * 1024, 1024 * sizeof(double), hipMemcpyDeviceToHost, st_));
      G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
      double sset = 0.0;
      for (double v : h) sset += v;
      total += sset;
      continue;
    }
#define G2OHIP_CHI(d_)                                                                                                      \
  case d_:                                                                                                                  \
    hipLaunchKernelGGL((chi2_kernel<d_>), dim3(nblocks), dim3(kThreads), 0, st_, es.n, es.omega, es.err, es.kernel_kind, es.delta, \
                       d_red.p, es.rk.p);                                                                                   \
    break
    switch (es.d) {
      G2OHIP_CHI(1);
      G2OHIP_CHI(2);
      G2OHIP_CHI(3);
      G2OHIP_CHI(6);
      G2OHIP_CHI(7);
      default:
        throw ArgFailure("chi2: unsupported error dimension");
    }
#undef G2OHIP_CHI
    total += reduce_sum_finish(nblocks);
  }
  chi2_value_ = total;
  chi2_valid_ = true;
  for (auto& esp : sets_)
    if (esp->external) chi2_valid_ = false;   // caller-owned arrays: nothing tells the solver when they change
  return total;
}

void BlockSolver::set_lambda(double lambda, bool backup) { set_lambda_split(lambda, lambda, backup); }

// With the Schur complement the damping is VIRTUAL: lambda lives in two device scalars that
// landmark_inverse (Hll + lambda I) and schur_reduce (Hpp + lambda I) read; Hpp / Hll stay untouched, so
// there is nothing to back up or restore (the reference walks every diagonal block twice per trial,
// block_solver.hpp:563-604).  Readers of the matrices (copy_values, multiply_hessian, max_diagonal) add it.
__global__ void set_pair_kernel(double* __restrict__ p, double a, double b) {
  p[0] = a;
  p[1] = b;
}

void BlockSolver::set_lambda_split(double lambda_pose, double lambda_landmark, bool backup) {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (schur_) {
    prof.begin(KernelProf::kLambda, st_);
    hipLaunchKernelGGL(set_pair_kernel, dim3(1), dim3(1), 0, st_, d_lam.p, lambda_pose, lambda_landmark);
    prof.end(KernelProf::kLambda, st_);
    lam_pose_ = lambda_pose;
    lam_lm_ = lambda_landmark;
    G2OHIP_HIP_CHECK(hipGetLastError());
    return;
  }
  prof.begin(KernelProf::kLambda, st_);
  hipLaunchKernelGGL(lambda_kernel, dim3(grid_for((size_t)nP_ * p_)), dim3(kThreads), 0, st_, nP_, p_, d_Hpp.p, d_pp_diag.p, d_bkP.p,
                     lambda_pose, backup ? 1 : 0, 0, chol_opt.world > 1 ? d_lam_mask.p : (const unsigned char*)nullptr);
  if (nL_ > 0)
    hipLaunchKernelGGL(lambda_kernel, dim3(grid_for((size_t)nL_ * l_)), dim3(kThreads), 0, st_, nL_, l_, d_Hll.p, (const int*)nullptr,
                       d_bkL.p, lambda_landmark, backup ? 1 : 0, 0);
  prof.end(KernelProf::kLambda, st_);
  G2OHIP_HIP_CHECK(hipGetLastError());
}

void BlockSolver::restore_diagonal() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (schur_) {
    hipLaunchKernelGGL(set_pair_kernel, dim3(1), dim3(1), 0, st_, d_lam.p, 0.0, 0.0);
    lam_pose_ = lam_lm_ = 0.0;
    G2OHIP_HIP_CHECK(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(lambda_kernel, dim3(grid_for((size_t)nP_ * p_)), dim3(kThreads), 0, st_, nP_, p_, d_Hpp.p, d_pp_diag.p, d_bkP.p,
                     0.0, 0, 1);
  if (nL_ > 0)
    hipLaunchKernelGGL(lambda_kernel, dim3(grid_for((size_t)nL_ * l_)), dim3(kThreads), 0, st_, nL_, l_, d_Hll.p, (const int*)nullptr,
                       d_bkL.p, 0.0, 0, 1);
  G2OHIP_HIP_CHECK(hipGetLastError());
}

double BlockSolver::max_diagonal() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ensure_ll();
  double m = 0.0;
  {
    int nblocks = std::min(1024, grid_for((size_t)nP_ * p_));
    hipLaunchKernelGGL(maxdiag_partial_kernel, dim3(nblocks), dim3(kThreads), 0, st_, nP_, p_, d_Hpp.p, d_pp_diag.p, d_red.p);
    std::vector<double> h(nblocks);
    d_red.download(h.data(), nblocks, st_);
    for (double v : h) m = std::max(m, v + lam_pose_);
  }
  if (nL_ > 0) {
    int nblocks = std::min(1024, grid_for((size_t)nL_ * l_));
    hipLaunchKernelGGL(maxdiag_partial_kernel, dim3(nblocks), dim3(kThreads), 0, st_, nL_, l_, d_Hll.p, (const int*)nullptr, d_red.p);
    std::vector<double> h(nblocks);
    d_red.download(h.data(), nblocks, st_);
    for (double v : h) m = std::max(m, v + lam_lm_);
  }
  return m;
}

// scalar diagonal of H, poses then landmarks, in hessian-index order (what OptimizationAlgorithmLevenberg::computeLambdaInit
// reads through v->hessian(j, j), optimization_algorithm_levenberg.cpp:149-163); includes the current damping
__global__ void gather_diag_kernel(int nv, int dim, const double* __restrict__ H, const int* __restrict__ diag_block, double add,
                                   double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nv * dim) return;
  const int v = t / dim, j = t - v * dim;
  const size_t blk = diag_block ? (size_t)diag_block[v] : (size_t)v;
  out[t] = H[blk * dim * dim + (size_t)j * (dim + 1)] + add;
}

void BlockSolver::copy_diagonal(double* host) {
  require_structure();
  if (!host) throw ArgFailure("copy_diagonal: null pointer");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ensure_ll();
  const size_t n = vector_size();
  DevBuf<double> tmp;
  tmp.alloc(n);
  hipLaunchKernelGGL(gather_diag_kernel, dim3(grid_for((size_t)nP_ * p_)), dim3(kThreads), 0, st_, nP_, p_, d_Hpp.p, d_pp_diag.p,
                     schur_ ? lam_pose_ : 0.0, tmp.p);
  if (nL_ > 0)
    hipLaunchKernelGGL(gather_diag_kernel, dim3(grid_for((size_t)nL_ * l_)), dim3(kThreads), 0, st_, nL_, l_, d_Hll.p, (const int*)nullptr,
                       schur_ ? lam_lm_ : 0.0, tmp.p + (size_t)nP_ * p_);
  tmp.download(host, n, st_);
}

double BlockSolver::compute_scale(double lambda) {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ensure_bl();
  int nblocks = std::min(1024, grid_for(vector_size()));
  hipLaunchKernelGGL(scale_partial_kernel, dim3(nblocks), dim3(kThreads), 0, st_, vector_size(), d_x.p, d_b.p, lambda, d_red.p);
  return reduce_sum_finish(nblocks);
}

void BlockSolver::solve_schur() {
  require_structure();
  if (!schur_) return;
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  const bool sv = sv_ready_ && sharded_virtual && chol_opt.world > 1 && linear_solver == 0 && fuse_schur_reduce;
  if (sv != sv_now_) drop_graph_segments();   // (the factor segments are specific to the source of the matrix)
  sv_now_ = sv;
  solve_schur_impl(!sv);
  if (sv && ex_.nbb > 0) launch_boundary_reduce();
}

// sharded_virtual: the boundary blocks of the reduced system (and the right-hand side of their diagonal ones) from this rank's
// Hpp and partial blocks, into the region behind Hpp -- schur_reduce_kernel over the boundary list
void BlockSolver::launch_boundary_reduce() {
  const int G = schur_group > 0 ? schur_group : pick_group((double)n_sc_ / std::max<long>(1, n_td_));
  const bool split = (p_ % 2 == 0) && G >= 2;
  double* Hs = d_Hpp.p + hpp_blocks_ * (size_t)p_ * p_;
  const int n_red = ex_.nbb;
#define G2OHIP_RED(P_, NRP_)                                                                                                  \
  hipLaunchKernelGGL((schur_reduce_kernel<P_, NRP_>), dim3(grid_for((size_t)n_red * P_ * P_)), dim3(kThreads), 0, st_, n_red,     \
                     d_rd_ptr.p, d_rd_slot.p, d_hs_src.p, d_Hpp.p, d_Pd.p, Hs, d_hs_diag.p, d_Pr.p, d_b.p, d_bschur.p, d_lam.p,  \
                     d_lam_mask.p, ex_.bblock.p, (int)std::max<long>(1, n_td_))
  switch (p_) {
    case 3: G2OHIP_RED(3, 3); break;
    case 6: if (split) G2OHIP_RED(6, 3); else G2OHIP_RED(6, 6); break;
    case 7: G2OHIP_RED(7, 7); break;
    default: throw ArgFailure("unsupported pose dimension for Schur");
  }
#undef G2OHIP_RED
  G2OHIP_HIP_CHECK(hipGetLastError());
}

void BlockSolver::solve_schur_impl(bool want_matrix) {
  if (profiling) ts_.start(st_);
  const size_t sizeP = (size_t)nP_ * p_;
  const int G = schur_group > 0 ? schur_group : pick_group((double)n_sc_ / std::max<long>(1, n_td_));
  // every landmark lies in exactly one tile, so the tiles can invert the landmark blocks themselves
  const bool fuse_inv = fuse_landmark_inverse && n_tiles_ > 0 && tiles_cover_all_;
  // fused EdgeProjectXYZ2UV system built from the current estimates: the tiles evaluate their Hpl blocks themselves
  const bool ba_tiles = fuse_inv && p_ == 6 && l_ == 3 && ba_.cam_q.p != nullptr && ba_recompute_ok();
  if (!ba_tiles) {   // the generic tiles read Hpl, Hll and b_l from memory: both must be there (a fused solve with
    ensure_hpl();    // ba_store_ll = 0 leaves Hll partial, and with ba_skip_hpl = 0 ensure_hpl alone returns at once)
    ensure_ll();
  }
  if (ba_tiles) {
    EdgeSet& es = *sets_[ba_.set];
    prepare_ba_tile_kernels();
    prof.begin(KernelProf::kSchurBlocks, st_);
    const bool fll = !ll_valid_ || ll_hbm_partial_;   // the tiles assemble Hll / b_l / errors themselves (again, if the last ones kept Hll on chip)
    // ... and write to HBM only what the solve path reads (Dinv, b_l) unless option ba_store_ll asks for Hll and the errors
    // too: 150 MB less to write at the metric configuration; a later reader (maxDiagonal, multiplyHessian, chi2 without a
    // linearisation, inspection) has them recomputed by ensure_ll()
    const bool store_ll = ba_store_ll != 0;
    static const int schur_abl = getenv("G2OHIP_SCHUR_ABL") ? atoi(getenv("G2OHIP_SCHUR_ABL")) & ~1 : 0;   // (timing experiments only)
#define G2OHIP_BA_TILE(GG) G2OHIP_BA_TILE_(GG, false)
#define G2OHIP_BA_TILE_(GG, CC)                                                                                                    \
  do {                                                                                                                             \
    if (fll)                                                                                                                       \
      hipLaunchKernelGGL((ba_schur_tile_kernel<GG, true, CC>), dim3(n_tiles_), dim3(kThreads), schur_lds_bytes_, st_, d_tile_lm0.p,        \
                         ba_.cams.p, ba_.pts.p, (const int*)nullptr, (const int*)nullptr, ba_.ll_meas.p, ba_.ll_omega.p, ba_.f, ba_.cx,  \
                         ba_.cy,                                                                                                     \
                         es.kernel_kind, es.delta, ba_.omega_identity ? 1 : 0, d_Dinv.p, d_b.p + sizeP, d_td_diag.p, d_td_ptr.p,     \
                         d_te_pack.p, d_te_lm.p, d_Pd.p, d_Pr.p, d_Hll.p, d_lam.p, ba_.ll_rec.p, ba_.tile_ll.p, ba_.ll_edge.p,       \
                         (ba_.err_valid || !store_ll) ? (double*)nullptr : es.own_err.p, /* (errors of these estimates already there) */ \
                         d_tile_q2.p, (store_ll ? 1 : 0) | schur_abl, d_slot_lm.p, ba_.ctab.p);                                            \
    else                                                                                                                           \
      hipLaunchKernelGGL((ba_schur_tile_kernel<GG, false, CC>), dim3(n_tiles_), dim3(kThreads), schur_lds_bytes_, st_, d_tile_lm0.p,       \
                         ba_.cams.p, ba_.pts.p, ba_.cam_q.p, ba_.pt_q.p, ba_.meas_q.p, ba_.omega_q.p, ba_.f, ba_.cx, ba_.cy,           \
                         es.kernel_kind, es.delta, ba_.omega_identity ? 1 : 0, d_Dinv.p, d_b.p + sizeP, d_td_diag.p, d_td_ptr.p,     \
                         d_te_pack.p, d_te_lm.p, d_Pd.p, d_Pr.p, d_Hll.p, d_lam.p, (const int4*)nullptr, (const int4*)nullptr,        \
                         (const int*)nullptr, (double*)nullptr, d_tile_q2.p, 1, d_slot_lm.p, ba_.ctab.p);                                  \
  } while (0)
    if (ba_.n_classes > 1) {   // edge classes: two lane-group widths are instantiated -- the ones on either side of the row split of the
      if (G <= 1) G2OHIP_BA_TILE_(1, true);   // partial blocks (launch_schur_reduce and the factorisation read the layout G implies)
      else G2OHIP_BA_TILE_(8, true);
    } else if (G <= 1) G2OHIP_BA_TILE(1);
    else if (G <= 2) G2OHIP_BA_TILE(2);
    else if (G <= 4) G2OHIP_BA_TILE(4);
    else if (G <= 8) G2OHIP_BA_TILE(8);
    else G2OHIP_BA_TILE(16);
    ll_valid_ = true;
    ll_hbm_partial_ = fll && !store_ll;
#undef G2OHIP_BA_TILE
#undef G2OHIP_BA_TILE_
    prof.end(KernelProf::kSchurBlocks, st_);
  } else
#define G2OHIP_TILE_ARGS d_tile_lm0.p, d_tile_td0.p, d_pl_colptr.p, d_Hpl.p, d_Dinv.p, d_b.p + sizeP, d_td_diag.p, d_td_ptr.p, d_te_pack.p, \
                         d_te_lm.p, d_Pd.p, d_Pr.p, fuse_inv ? d_Hll.p : (const double*)nullptr, d_lam.p, d_tile_q2.p, d_slot_lm.p
#define G2OHIP_SCHUR(P_, L_)                                                                                                   \
  if (p_ == P_ && l_ == L_) {                                                                                                  \
    if (!fuse_inv) {                                                                                                           \
      prof.begin(KernelProf::kLmInverse, st_);                                                                                 \
      hipLaunchKernelGGL((landmark_inverse_kernel<L_>), dim3(grid_for(nL_)), dim3(kThreads), 0, st_, nL_, d_Hll.p, d_b.p + sizeP, \
                         d_Dinv.p, d_db.p, d_lam.p);                                                                           \
      prof.end(KernelProf::kLmInverse, st_);                                                                                   \
    }                                                                                                                          \
    prof.begin(KernelProf::kSchurBlocks, st_);                                                                                 \
    if (n_tiles_ > 0) {                                                                                                        \
      static bool attr = false;                                                                                                \
      if (!attr) {                                                                                                             \
        (void)hipFuncSetAttribute((const void*)schur_tile_kernel<P_, L_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)hipFuncSetAttribute((const void*)schur_tile_kernel<P_, L_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)hipFuncSetAttribute((const void*)schur_tile_kernel<P_, L_, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)hipFuncSetAttribute((const void*)schur_tile_kernel<P_, L_, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)hipFuncSetAttribute((const void*)schur_tile_kernel<P_, L_, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        attr = true;                                                                                                           \
      }                                                                                                                        \
      if (G <= 1)                                                                                                              \
        hipLaunchKernelGGL((schur_tile_kernel<P_, L_, 1>), dim3(n_tiles_), dim3(kThreads), schur_lds_bytes_, st_, G2OHIP_TILE_ARGS); \
      else if (G <= 2)                                                                                                         \
        hipLaunchKernelGGL((schur_tile_kernel<P_, L_, 2>), dim3(n_tiles_), dim3(kThreads), schur_lds_bytes_, st_, G2OHIP_TILE_ARGS); \
      else if (G <= 4)                                                                                                         \
        hipLaunchKernelGGL((schur_tile_kernel<P_, L_, 4>), dim3(n_tiles_), dim3(kThreads), schur_lds_bytes_, st_, G2OHIP_TILE_ARGS); \
      else if (G <= 8)                                                                                                         \
        hipLaunchKernelGGL((schur_tile_kernel<P_, L_, 8>), dim3(n_tiles_), dim3(kThreads), schur_lds_bytes_, st_, G2OHIP_TILE_ARGS); \
      else                                                                                                                     \
        hipLaunchKernelGGL((schur_tile_kernel<P_, L_, 16>), dim3(n_tiles_), dim3(kThreads), schur_lds_bytes_, st_, G2OHIP_TILE_ARGS); \
    }                                                                                                                          \
    prof.end(KernelProf::kSchurBlocks, st_);                                                                                   \
  } else
  G2OHIP_SCHUR(6, 3)
  G2OHIP_SCHUR(3, 2)
  G2OHIP_SCHUR(7, 3)
  G2OHIP_SCHUR(6, 2)
  G2OHIP_SCHUR(3, 3) { throw ArgFailure("unsupported (pose_dim, landmark_dim) for Schur"); }
#undef G2OHIP_SCHUR
#undef G2OHIP_TILE_ARGS
  prof.begin(KernelProf::kSchurRhs, st_);
  launch_schur_reduce(want_matrix);
  prof.end(KernelProf::kSchurRhs, st_);
  G2OHIP_HIP_CHECK(hipGetLastError());
  if (profiling) {
    ts_.stop(st_);
    times.schur = ts_.seconds();
  }
}

// Pass 2 of the Schur complement.  matrix: Hschur and bschur (schur_reduce_kernel); otherwise only bschur -- the
// factorisation then reads Hpp and the partial blocks itself (virtual_reduced_ok) and Hschur stays stale until
// ensure_hschur().
void BlockSolver::launch_schur_reduce(bool matrix) {
  const int hs_nnzb = (int)hs_row.size();
  const int G = schur_group > 0 ? schur_group : pick_group((double)n_sc_ / std::max<long>(1, n_td_));
  const bool split = (p_ % 2 == 0) && G >= 2;   // rows per lane part of the tile kernel's partial layout
  chol_->set_virtual_split(split);
  if (!matrix) {
    hschur_valid_ = false;
    // one GPU, direct solver: the kernel writes the permuted right-hand side and clears the status word for the factorisation
    // that follows (solve_reduced_device then starts with the band chains: no permute_in launch, no memset)
    const bool pre = rhs_prefill && chol_opt.world <= 1 && linear_solver == 0;
    size_t xpn = 0;
    rhs_prefilled_ = pre;
#define G2OHIP_RHS(P_)                                                                                                        \
  case P_:                                                                                                                    \
    hipLaunchKernelGGL((schur_rhs_kernel<P_>), dim3(grid_for((size_t)nP_ * P_)), dim3(kThreads), 0, st_, nP_, d_pose_diag.p, d_rd_ptr.p, \
                       d_rd_slot.p, d_Pr.p, d_b.p, d_bschur.p, pre ? chol_->inverse_permutation_device() : (const int*)nullptr,        \
                       pre ? chol_->permuted_solution(&xpn) : (double*)nullptr, pre ? chol_->status_word_device() : (int*)nullptr);      \
    break
    switch (p_) {
      G2OHIP_RHS(3);
      G2OHIP_RHS(6);
      G2OHIP_RHS(7);
      default: throw ArgFailure("unsupported pose dimension for Schur");
    }
#undef G2OHIP_RHS
    return;
  }
  hschur_valid_ = true;
  const int n_red = n_active_ >= 0 ? n_active_ : hs_nnzb;
  if (n_red <= 0) return;
#define G2OHIP_RED(P_, NRP_)                                                                                                  \
  hipLaunchKernelGGL((schur_reduce_kernel<P_, NRP_>), dim3(grid_for((size_t)n_red * P_ * P_)), dim3(kThreads), 0, st_, n_red,     \
                     d_rd_ptr.p, d_rd_slot.p, d_hs_src.p, d_Hpp.p, d_Pd.p, d_Hschur.p, d_hs_diag.p, d_Pr.p, d_b.p, d_bschur.p, \
                     d_lam.p, chol_opt.world > 1 ? d_lam_mask.p : (const unsigned char*)nullptr,                              \
                     n_active_ >= 0 ? d_active.p : (const int*)nullptr, (int)std::max<long>(1, n_td_))
  switch (p_) {
    case 3: G2OHIP_RED(3, 3); break;
    case 6: if (split) G2OHIP_RED(6, 3); else G2OHIP_RED(6, 6); break;
    case 7: G2OHIP_RED(7, 7); break;
    default: throw ArgFailure("unsupported pose dimension for Schur");
  }
#undef G2OHIP_RED
}

// somebody reads Hschur (copy_values, PCG, marginals, multi-GPU exchange) after a solve() that skipped it
void BlockSolver::ensure_hschur() {
  if (!schur_ || hschur_valid_) return;
  launch_schur_reduce(true);
  G2OHIP_HIP_CHECK(hipGetLastError());
}

// may solve() leave the reduction to the factorisation?  (one GPU, direct solver, tiled Schur pass)
bool BlockSolver::virtual_reduced_ok() {
  // (a block of the reduced system that collects partial blocks from many tiles -- loop closures: pose pairs share
  // landmarks all over the landmark order -- is summed faster by the one fully parallel reduction pass than inside the
  // front assembly of its tree level: 6.4 against 7.5 ms per iteration on the 10 000-pose loop-closure graph)
  return schur_ && fuse_schur_reduce && chol_opt.world == 1 && linear_solver == 0 && n_tiles_ > 0 && n_active_ < 0 && !rd_ptr_h_.empty() &&
         (double)n_td_ <= fuse_reduce_max_partials * (double)std::max<size_t>(hs_row.size(), 1);
}

int BlockSolver::solve_reduced() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ensure_hschur();
  if (virt_now_) invalidate_graphs();
  virt_now_ = false;
  return solve_reduced_impl();
}

int BlockSolver::solve_reduced_impl() {
  if (!chol_->analyzed()) {
    if (schur_) chol_->analyze(nP_, hs_colptr.data(), hs_row.data(), st_);
    else chol_->analyze(nP_, pp_colptr.data(), pp_row.data(), st_);
  }
  if (linear_solver == 1) {   // LinearSolverPCG on the reduced system
    if (!pcg_) pcg_ = std::make_unique<BlockPCG>(p_);
    if (!pcg_->analyzed()) {
      if (schur_) pcg_->analyze(nP_, hs_colptr.data(), hs_row.data(), st_);
      else pcg_->analyze(nP_, pp_colptr.data(), pp_row.data(), st_);
    }
    pcg_->opt = pcg_opt;
    if (profiling) tn_.start(st_);
    prof.begin(KernelProf::kCholFactor, st_);
    const bool ok = pcg_->solve(schur_ ? d_Hschur.p : d_Hpp.p, schur_ ? d_bschur.p : d_b.p, d_x.p, st_);
    prof.end(KernelProf::kCholFactor, st_);
    if (profiling) {
      tn_.stop(st_);
      times.numeric = tn_.seconds();
      times.linsolve = 0.0;
    }
    pcg_iterations = pcg_->last_iterations();
    return ok ? 0 : 1;
  }
  solve_reduced_device();
  bool bad = chol_->failed(st_);   // synchronises
  if (bad && chol_->dependency_stall() && ++dependency_fallbacks) {   // (safety net of the dependency-driven launches: repeat with one launch per level)
    invalidate_graphs();
    solve_reduced_device();
    bad = chol_->failed(st_);
  }
  if (profiling) {
    times.numeric = tn_.seconds();
    times.linsolve = tl_.seconds();
  }
  return bad ? 1 : 0;
}

void BlockSolver::solve_reduced_device() {
  // factorisation with the forward sweep fused into it, then the backward sweep
  const double* Hred = schur_ ? (virt_now_ ? (const double*)nullptr : d_Hschur.p) : d_Hpp.p;   // nullptr: virtual source
  const double* bred = schur_ ? d_bschur.p : d_b.p;
  // The two launch-bound sequences (one launch per tree level) are hipGraph segments; the timing events sit
  // between the segments, so per-slot times stay available while the graphs replay.
  if (profiling) tn_.start(st_);
  if (chol_->has_band_chains(0) && prof.enabled && prof.only < 0) {
    // every kernel slot is being timed: the factorisation is two kernels (the band chains, the tree levels above them);
    // plain launches with the events right around the band chains' launch -- its slot is that kernel alone, the other
    // slot everything else of the sequence (right-hand side in, the other levels)
    prof.begin(KernelProf::kCholFactor, st_);
    chol_->band_hook = [this](int after) {
      if (!after) {
        prof.end(KernelProf::kCholFactor, st_);
        prof.begin(KernelProf::kCholBand, st_);
      } else {
        prof.end(KernelProf::kCholBand, st_);
        prof.begin(KernelProf::kCholFactor, st_, /*cont=*/true);
      }
    };
    rhs_prefilled_ = false;   // (every slot timed: the plain sequence)
    chol_->solve_begin(bred, st_);
    chol_->factor_phase(Hred, 0, st_, true);
    chol_->factor_phase(Hred, 1, st_, true);
    chol_->band_hook = nullptr;
    prof.end(KernelProf::kCholFactor, st_);
  } else {
    prof.begin(KernelProf::kCholFactor, st_);
    const bool pre = rhs_prefilled_ && schur_ && virt_now_;   // (launch_schur_reduce has written the permuted right-hand side and cleared the status word)
    rhs_prefilled_ = false;
    run_seg(pre ? kSegFactorPre : kSegFactor, [&] {
      if (!pre) chol_->solve_begin(bred, st_);
      chol_->skip_status_clear = pre;
      chol_->factor_phase(Hred, 0, st_, true);
      chol_->factor_phase(Hred, 1, st_, true);
    });
    prof.end(KernelProf::kCholFactor, st_);
  }
  if (profiling) {
    tn_.stop(st_);
    tl_.start(st_);
  }
  prof.begin(KernelProf::kCholSolve, st_);
  run_seg(kSegBackward, [&] {
    chol_->solve_backward_phase(1, st_);
    chol_->solve_backward_phase(0, st_);
    chol_->solve_end(d_x.p, st_);
  });
  prof.end(KernelProf::kCholSolve, st_);
  if (profiling) tl_.stop(st_);
}

void BlockSolver::set_partition(int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) throw ArgFailure("set_partition: bad rank/world");
  invalidate_graphs();
  chol_opt.rank = rank;
  chol_opt.world = world;
  structured_ = false;
}

void BlockSolver::solve_reduced_local() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  solve_reduced_local_impl();
}

void BlockSolver::solve_reduced_local_impl() {
  prof.begin(KernelProf::kCholFactor, st_);
  run_seg(kSegLocal, [&] {
    chol_->solve_begin(schur_ ? d_bschur.p : d_b.p, st_);
    chol_->factor_phase(schur_ ? (sv_now_ ? (const double*)nullptr : d_Hschur.p) : d_Hpp.p, 0, st_, true);   // forward sweep fused in (nullptr: virtual source)
    chol_->pack_exchange(st_);
  });
  prof.end(KernelProf::kCholFactor, st_);
  G2OHIP_HIP_CHECK(hipGetLastError());
}

void BlockSolver::solve_reduced_shared() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  solve_reduced_shared_impl();
}

void BlockSolver::solve_reduced_shared_impl() {
  prof.begin(KernelProf::kCholFactor, st_);
  run_seg(kSegShared, [&] {
    chol_->unpack_exchange(st_);
    chol_->factor_phase(schur_ ? (sv_now_ ? (const double*)nullptr : d_Hschur.p) : d_Hpp.p, 1, st_, true);
  });
  prof.end(KernelProf::kCholFactor, st_);
  prof.begin(KernelProf::kCholSolve, st_);
  run_seg(kSegSharedBack, [&] {
    chol_->solve_backward_phase(1, st_);
    chol_->solve_backward_phase(0, st_);
    if (mask_solution) chol_->mask_solution(st_);
  });
  prof.end(KernelProf::kCholSolve, st_);
  G2OHIP_HIP_CHECK(hipGetLastError());
}

int BlockSolver::solve_reduced_finish() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  chol_->solve_end(d_x.p, st_);
  const bool bad = chol_->failed(st_);
  if (bad && chol_->dependency_stall() && ++dependency_fallbacks) invalidate_graphs();   // this solve is reported failed; the next one runs level by level
  return bad ? 1 : 0;
}

// ---- halo exchange helpers (see block_solver.h) ------------------------------------------------------
__global__ void exchange_boundary_kernel(int nbb, int nbp, int pp, int p, const int* __restrict__ bblock, const int* __restrict__ bpose,
                                         const double* __restrict__ hkeep, const double* __restrict__ bkeep, double* __restrict__ Hs,
                                         double* __restrict__ bs, double* __restrict__ buf, int unpack) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n1 = nbb * pp;
  if (t >= n1 + nbp * p) return;
  double* target;
  double keep;
  if (t < n1) {
    target = Hs + (size_t)bblock[t / pp] * pp + t % pp;
    keep = hkeep[t / pp];
  } else {
    const int u = t - n1;
    target = bs + (size_t)bpose[u / p] * p + u % p;
    keep = bkeep[u / p];
  }
  if (unpack) *target = buf[t] * keep;   // a rank keeps only what it consumes (a stale sum would be added again)
  else buf[t] = *target;
}
__global__ void exchange_halo_kernel(int nh, int p, const int* __restrict__ halo, const double* __restrict__ hmine, double* __restrict__ x,
                                     double* __restrict__ buf, const int* __restrict__ status, int unpack) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > nh * p) return;
  if (t == nh * p) {
    // (summed over the ranks: not positive definite counts 1, a dependency-driven launch that gave up waiting 1024 --
    // the solve is to be REPEATED then, not reported failed)
    if (!unpack) buf[t] = *status == 2 ? 1024.0 : (*status != 0 ? 1.0 : 0.0);
    return;
  }
  double* target = x + (size_t)halo[t / p] * p + t % p;
  if (unpack) *target = buf[t];
  else buf[t] = *target * hmine[t / p];
}

void BlockSolver::exchange_setup(int nbb, const int* bblock, const double* hkeep, int nbp, const int* bpose, const double* bkeep, int nh,
                                 const int* halo, const double* hmine) {
  require_structure();
  if (nbb < 0 || nbp < 0 || nh < 0) throw ArgFailure("exchange_setup: negative count");
  drop_graph_segments();   // (buffers move: the captured launch sequences hold their addresses; what the kernels read has not changed)
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ex_.nbb = nbb;
  ex_.nbp = nbp;
  ex_.nh = nh;
  ex_.valid = true;
  auto up_i = [&](DevBuf<int>& d, const int* h, int n) {
    std::vector<int> v(h, h + n);
    if (v.empty()) v.push_back(0);
    d.upload(v, st_);
  };
  auto up_d = [&](DevBuf<double>& d, const double* h, int n) {
    std::vector<double> v(h, h + n);
    if (v.empty()) v.push_back(0.0);
    d.upload(v, st_);
  };
  up_i(ex_.bblock, bblock, nbb);
  up_i(ex_.bpose, bpose, nbp);
  up_i(ex_.halo, halo, nh);
  up_d(ex_.hkeep, hkeep, nbb);
  up_d(ex_.bkeep, bkeep, nbp);
  up_d(ex_.hmine, hmine, nh);
  ex_.buf1.alloc((size_t)std::max(1, nbb * p_ * p_ + nbp * p_));
  ex_.buf1.zero(st_);
  // Two collectives instead of three: the first all-reduce (boundary blocks of the reduced system, boundary b_p) can wait for
  // the second (update matrices / vectors of the subtree roots) when nothing a rank factorises on its OWN reads the summed
  // values -- every boundary block is assembled by a front of the shared top, every boundary pose is eliminated there.  With
  // landmarks dealt by their first OWNED observing pose that holds whenever no landmark spans two subtrees (a band: always).
  ex_.merged = false;
  ex_.tail = nullptr;
  if (sharded_merge && chol_ && chol_opt.world > 1 && schur_) {
    const CholSymbolic& S = chol_->symbolic();
    bool ok = (int)S.block_consumer.size() == (int)hs_row.size() && (int)S.pose_owner.size() == nP_;
    for (int k = 0; k < nbb && ok; ++k) ok = bblock[k] >= 0 && bblock[k] < (int)S.block_consumer.size() && S.block_consumer[bblock[k]] < 0;
    for (int k = 0; k < nbp && ok; ++k) ok = bpose[k] >= 0 && bpose[k] < nP_ && S.pose_owner[bpose[k]] < 0;
    if (ok && nbb * p_ * p_ + nbp * p_ > 0) {
      ex_.tail = chol_->reserve_exchange_tail((size_t)nbb * p_ * p_ + (size_t)nbp * p_);
      ex_.merged = true;
    }
  }
  ex_.buf3.alloc((size_t)nh * p_ + 1);
  ex_.buf3.zero(st_);
  // Virtual source on a rank (sharded_virtual): as on one GPU the fronts are assembled from Hpp and the tiles' partial blocks
  // and Hschur is not written -- except the boundary blocks, whose value is a sum over ranks: those are reduced locally
  // into the region behind Hpp, summed there by the exchange and read from there by the factorisation (no partials, no
  // damping of their own: the sum holds both).
  sv_ready_ = false;
  if (sharded_virtual && schur_ && chol_opt.world > 1 && n_tiles_ > 0 && !rd_ptr_h_.empty() && fuse_schur_reduce && linear_solver == 0 &&
      d_Hpp.n >= (hpp_blocks_ + hs_row.size()) * (size_t)p_ * p_ &&
      (double)n_td_ <= fuse_reduce_max_partials * (double)std::max<size_t>(hs_row.size(), 1)) {
    const int nb = (int)hs_row.size();
    std::vector<char> isb(nb, 0);
    for (int k = 0; k < nbb; ++k) {
      if (bblock[k] < 0 || bblock[k] >= nb) throw ArgFailure("exchange_setup: block index out of range");
      isb[bblock[k]] = 1;
    }
    sv_base_.assign(nb, -1);
    sv_diag_.assign(nb, -1);
    sv_ptr_.assign(nb + 1, 0);
    sv_slot_.clear();
    for (int d = 0; d < nb; ++d) {
      if (isb[d]) {
        sv_base_[d] = (int)hpp_blocks_ + d;
      } else {
        sv_base_[d] = hs_src_h_[d];
        sv_diag_[d] = hs_diag_h_[d];
        for (int k = rd_ptr_h_[d]; k < rd_ptr_h_[d + 1]; ++k) sv_slot_.push_back(rd_slot_h_[k]);
      }
      sv_ptr_[d + 1] = (int)sv_slot_.size();
    }
    if (sv_slot_.empty()) sv_slot_.push_back(0);
    d_sv_slot.upload(sv_slot_, st_);
    SparseCholesky::VirtualBlocks vb;
    vb.base_idx = sv_base_.data();
    vb.is_diag = sv_diag_.data();
    vb.part_ptr = sv_ptr_.data();
    vb.part_slot = sv_slot_.data();
    vb.d_part_slot = d_sv_slot.p;
    vb.base = d_Hpp.p;
    vb.parts = d_Pd.p;
    vb.lam = d_lam.p;
    vb.zero_slot = (int)std::max<long>(n_td_, 1);
    vb.split = false;   // (set per solve: launch_schur_reduce)
    chol_->set_virtual_blocks(vb, st_);
    G2OHIP_HIP_CHECK(hipMemsetAsync(d_Hpp.p + hpp_blocks_ * (size_t)p_ * p_, 0, hs_row.size() * (size_t)p_ * p_ * sizeof(double), st_));
    sv_ready_ = true;
    drop_graph_segments();
  }
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
}

void BlockSolver::exchange_pack(int which) {
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (which == 1) {
    const int n = ex_.nbb * p_ * p_ + ex_.nbp * p_;
    if (n > 0)
      hipLaunchKernelGGL(exchange_boundary_kernel, dim3(grid_for(n)), dim3(kThreads), 0, st_, ex_.nbb, ex_.nbp, p_ * p_, p_, ex_.bblock.p,
                         ex_.bpose.p, ex_.hkeep.p, ex_.bkeep.p, sv_now_ ? d_Hpp.p + hpp_blocks_ * (size_t)p_ * p_ : d_Hschur.p, d_bschur.p,
                         ex_.merged ? ex_.tail : ex_.buf1.p, 0);
  } else if (which == 3) {
    hipLaunchKernelGGL(exchange_halo_kernel, dim3(grid_for(ex_.nh * p_ + 1)), dim3(kThreads), 0, st_, ex_.nh, p_, ex_.halo.p, ex_.hmine.p,
                       d_x.p, ex_.buf3.p, chol_->status_device(), 0);
  } else {
    throw ArgFailure("exchange_pack: which must be 1 or 3");
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
}

void BlockSolver::exchange_unpack(int which) {
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (which == 1) {
    const int n = ex_.nbb * p_ * p_ + ex_.nbp * p_;
    if (n > 0)
      hipLaunchKernelGGL(exchange_boundary_kernel, dim3(grid_for(n)), dim3(kThreads), 0, st_, ex_.nbb, ex_.nbp, p_ * p_, p_, ex_.bblock.p,
                         ex_.bpose.p, ex_.hkeep.p, ex_.bkeep.p, sv_now_ ? d_Hpp.p + hpp_blocks_ * (size_t)p_ * p_ : d_Hschur.p, d_bschur.p,
                         ex_.merged ? ex_.tail : ex_.buf1.p, 1);
  } else if (which == 3) {
    hipLaunchKernelGGL(exchange_halo_kernel, dim3(grid_for(ex_.nh * p_ + 1)), dim3(kThreads), 0, st_, ex_.nh, p_, ex_.halo.p, ex_.hmine.p,
                       d_x.p, ex_.buf3.p, chol_->status_device(), 1);
  } else {
    throw ArgFailure("exchange_unpack: which must be 1 or 3");
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
}

void BlockSolver::solve_reduced_finish_async() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  chol_->solve_end(d_x.p, st_);
}

int BlockSolver::exchange_status() {
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  double flag = 0.0;
  G2OHIP_HIP_CHECK(hipMemcpyAsync(&flag, ex_.buf3.p + (size_t)ex_.nh * p_, sizeof(double), hipMemcpyDeviceToHost, st_));
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
  if (flag != 0.0) {   // some rank failed; the local cleanup (dependency counters, stall fallback) runs where it applies
    if (chol_->failed(st_) && chol_->dependency_stall() && ++dependency_fallbacks) invalidate_graphs();
    // a stall anywhere (safety net of the dependency-driven launches): every rank sees the same sum and repeats the solve;
    // the rank that stalled runs one launch per level from now on.  Distinct from "not positive definite".
    return flag >= 1024.0 ? 2 : 1;
  }
  return 0;
}

// ---- sharded solve / LM scalars with the collectives inside the library ------------------------------------------
// x'(lambda x + b) of this rank: b_p holds this rank's contributions only (they sum to b_p over the ranks), x_p is valid
// for the poses this rank's landmarks observe (own, shared, halo) -- exactly where its b_p can be non-zero; lambda x^2
// is counted once per pose, by the rank that damps it (mask), and for every local landmark.
__global__ void __launch_bounds__(kThreads) scale_sharded_partial_kernel(size_t nPose, size_t n, const double* __restrict__ x,
                                                                         const double* __restrict__ b, double lambda, int pd,
                                                                         const unsigned char* __restrict__ mask,
                                                                         double* __restrict__ partial) {
  __shared__ double sh[kThreads];
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double bi = b[i];
    if (i < nPose) {
      const double xi = (bi != 0.0 || (mask && mask[i / pd])) ? x[i] : 0.0;   // (entries of foreign poses are never read)
      s += xi * bi;
      if (!mask || mask[i / pd]) s += lambda * xi * xi;
    } else {
      s += x[i] * (lambda * x[i] + bi);
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(kThreads) max_partial_kernel(size_t n, const double* __restrict__ v, double* __restrict__ partial) {
  __shared__ double sh[kThreads];
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s = fmax(s, fabs(v[i]));
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

void BlockSolver::comm_init_rccl(int rank, int world, const char* id128) {
  G2OHIP_HIP_CHECK(hipSetDevice(device_));   // the communicator binds to the current device
  comm.init_rccl(rank, world, id128);
}
void BlockSolver::comm_init_peer(int rank, int world, HostAllReduceFn fn, void* ctx, size_t slot_doubles) {
  G2OHIP_HIP_CHECK(hipSetDevice(device_));   // the mailbox lives on the solver's device
  comm.init_peer(rank, world, fn, ctx, slot_doubles);
}
void BlockSolver::comm_all_reduce(double* dev, size_t n, int op) {
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  comm.all_reduce(dev, n, op, st_);
}

int BlockSolver::solve_sharded_once() {
  require_structure();
  if (!schur_) throw StateFailure("solve_sharded: the sharded path needs the Schur complement");
  if (comm.kind() == Comm::kNone && chol_opt.world > 1 && !comm_emulate)
    throw StateFailure("solve_sharded: no communicator (g2ohip_comm_init_*)");
  if (chol_opt.world > 1 && !ex_.valid)
    throw StateFailure("solve_sharded: g2ohip_exchange_setup has not been called for the current structure (build_structure drops the exchange)");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  auto whole_solve = [&] {
  // (1) local Schur pass; boundary blocks of the reduced system + boundary right-hand sides summed over the ranks
  solve_schur();
  sharded_collectives = 0;
  const size_t n1 = (size_t)ex_.nbb * p_ * p_ + (size_t)ex_.nbp * p_;
  const bool merged = ex_.merged && n1 > 0;
  if (n1 > 0) {
    prof.begin(KernelProf::kExBoundary, st_);   // (pack + all-reduce + unpack: what the exchange costs the solve)
    exchange_pack(1);                           // (merged: into the tail of the subtree-root buffer; summed with it below)
    if (!merged) {
      comm.all_reduce(ex_.buf1.p, n1, 0, st_);
      ++sharded_collectives;
      exchange_unpack(1);
    }
    prof.end(KernelProf::kExBoundary, st_);
  }
  // (2) own subtrees: factor + forward sweep; update matrices / vectors of the subtree roots summed (separator-sized)
  solve_reduced_local();
  {
    size_t n = 0;
    double* xb = chol_->exchange_buffer(&n);
    prof.begin(KernelProf::kExRoots, st_);
    comm.all_reduce(xb, n + (merged ? n1 : 0), 0, st_);
    ++sharded_collectives;
    if (merged) {
      exchange_unpack(1);
      chol_->solve_begin(d_bschur.p, st_);   // the permuted right-hand side again: b of the shared poses is complete only now (the
                                             // forward sweep of the own subtrees has read its part and left it untouched)
    }
    prof.end(KernelProf::kExRoots, st_);
  }
  // (3) shared top of the tree (redundant), backward sweep down the own subtrees; halo x_p + failure flags summed
  solve_reduced_shared();
  solve_reduced_finish_async();
  prof.begin(KernelProf::kExHalo, st_);
  exchange_pack(3);
  comm.all_reduce(ex_.buf3.p, (size_t)ex_.nh * p_ + 1, 0, st_);
  ++sharded_collectives;
  exchange_unpack(3);
  prof.end(KernelProf::kExHalo, st_);
  solve_back_substitute();           // (harmless after a failed factorisation: the caller discards x)
  };
  // One hipGraph for the whole sequence when nothing in it has to cross the host: a rank's kernels are short at N = 8
  // (0.4 ms in total) and the launch boundaries between the phases were a third of its time.  The events of the kernel
  // timers cannot live inside a captured graph: with profiling on the phases run as before.
  const bool one_graph = sharded_graph > 0 && use_graph && !prof.enabled && !profiling &&
                         ((comm.kind() == Comm::kNone) || comm.kind() == Comm::kPeer || (comm.kind() == Comm::kRccl && sharded_graph >= 2));
  if (getenv("G2OHIP_GRAPH_DEBUG")) {
    static int once = 0;
    if (once++ < 8) fprintf(stderr, "solve_sharded: one_graph %d (sharded_graph %d use_graph %d prof %d profiling %d comm %d) state %d\n", (int)one_graph, sharded_graph,
                            (int)use_graph, (int)prof.enabled, (int)profiling, (int)comm.kind(), segs_[kSegShardedAll].state);
  }
  if (one_graph) {
    run_seg(kSegShardedAll, whole_solve);
  } else {
    whole_solve();
  }
  const int rc = exchange_status();   // (synchronises)
  comm.poll_error();
  return rc;
}

int BlockSolver::solve_sharded_repeat() {
  int rc = solve_sharded_once();
  for (int again = 0; rc == 2 && again < 2; ++again) rc = solve_sharded_once();   // (every rank takes the same branch: the flag is a sum)
  return rc == 0 ? 0 : 1;
}

// Start-up self-test of the sharded schedule (option sharded_selftest, default on; the first solve_sharded of a structure with a
// real communicator): the two-collective merge (sharded_merge) and the solve captured as one hipGraph with the collectives
// inside (sharded_graph) are the variants that no multi-GPU hardware has run before a job meets them.  The first solve is
// therefore done TWICE -- as configured, and with the three-collective, uncaptured reference schedule -- and the two
// solutions are compared on every rank (the verdict is a max over the ranks, so all of them take the same branch).  If they
// differ beyond rounding (or one of them is not finite) the job continues on the reference schedule with a note on stderr
// instead of failing on plumbing; otherwise the configured variant stays.  Costs one extra solve per structure.
int BlockSolver::solve_sharded() {
  const bool real_comm = comm.kind() != Comm::kNone && chol_opt.world > 1 && !comm_emulate;
  const bool variant = ex_.merged || (sharded_graph > 0 && use_graph);
  if (!sharded_selftest || selftest_done_ || !real_comm || !variant) return solve_sharded_repeat();
  selftest_done_ = true;
  const int rc1 = solve_sharded_repeat();
  const size_t n = vector_size();
  std::vector<double> x1(n), x0(n);
  d_x.download(x1.data(), n, st_);
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
  if (selftest_break && comm.rank() == 0)
    for (size_t i = 0; i < n; ++i) x1[i] *= 2.0;
  const bool merged_was = ex_.merged;
  const int graph_was = sharded_graph;
  ex_.merged = false;
  sharded_graph = 0;
  drop_graph_segments();
  const int rc0 = solve_sharded_repeat();
  d_x.download(x0.data(), n, st_);
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
  double scale = 0.0, diff = 0.0;
  bool finite = true;
  for (size_t i = 0; i < n; ++i) {
    finite = finite && std::isfinite(x0[i]) && std::isfinite(x1[i]);
    scale = std::max(scale, std::fabs(x0[i]));
    diff = std::max(diff, std::fabs(x0[i] - x1[i]));
  }
  // (both solves failed the same way -- not positive definite -- is agreement: x is meaningless then)
  double bad = (rc0 != rc1 || (rc0 == 0 && (!finite || diff > 1e-6 * std::max(scale, 1e-300)))) ? 1.0 : 0.0;
  comm.all_reduce_host(&bad, 1, 1, st_);
  if (bad > 0.0) {
    if (comm.rank() == 0)
      fprintf(stderr, "g2ohip: the sharded solve's start-up self-test failed (merged collectives %d, one-graph capture %d: |dx| %.3e against %.3e, "
              "status %d / %d): continuing on the three-collective, uncaptured schedule (sharded_merge = 0, sharded_graph = 0)\n",
              (int)merged_was, graph_was, diff, scale, rc1, rc0);
    selftest_fallback_ = true;   // (exchange_setup keeps the merge off from now on)
    sharded_merge = 0;
  } else {
    ex_.merged = merged_was;
    sharded_graph = graph_was;
    drop_graph_segments();
  }
  return rc0;
}

double BlockSolver::chi2_sharded() {
  double v = chi2();
  comm.all_reduce_host(&v, 1, 0, st_);
  return v;
}

double BlockSolver::compute_scale_sharded(double lambda) {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ensure_bl();
  const int nblocks = std::min(1024, grid_for(vector_size()));
  hipLaunchKernelGGL(scale_sharded_partial_kernel, dim3(nblocks), dim3(kThreads), 0, st_, (size_t)nP_ * p_, vector_size(), d_x.p, d_b.p,
                     lambda, p_, chol_opt.world > 1 ? d_lam_mask.p : (const unsigned char*)nullptr, d_red.p);
  double v = reduce_sum_finish(nblocks);
  comm.all_reduce_host(&v, 1, 0, st_);
  return v;
}

double BlockSolver::max_diagonal_sharded() {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ensure_ll();
  // pose diagonal: every rank holds its own contributions -> sum over the ranks first, then the maximum
  const size_t np = (size_t)nP_ * p_;
  DevBuf<double> diag;
  diag.alloc(np);
  hipLaunchKernelGGL(gather_diag_kernel, dim3(grid_for(np)), dim3(kThreads), 0, st_, nP_, p_, d_Hpp.p, d_pp_diag.p, 0.0, diag.p);
  comm.all_reduce(diag.p, np, 0, st_);
  double m = 0.0;
  {
    const int nblocks = std::min(1024, grid_for(np));
    hipLaunchKernelGGL(max_partial_kernel, dim3(nblocks), dim3(kThreads), 0, st_, np, diag.p, d_red.p);
    std::vector<double> h(nblocks);
    d_red.download(h.data(), nblocks, st_);
    for (double v : h) m = std::max(m, v + lam_pose_);
  }
  double ml = 0.0;
  if (nL_ > 0) {
    const int nblocks = std::min(1024, grid_for((size_t)nL_ * l_));
    hipLaunchKernelGGL(maxdiag_partial_kernel, dim3(nblocks), dim3(kThreads), 0, st_, nL_, l_, d_Hll.p, (const int*)nullptr, d_red.p);
    std::vector<double> h(nblocks);
    d_red.download(h.data(), nblocks, st_);
    for (double v : h) ml = std::max(ml, v + lam_lm_);
  }
  comm.all_reduce_host(&ml, 1, 1, st_);
  return std::max(m, ml);
}

void BlockSolver::partition_info(int* pose_owner, int* block_consumer) {
  require_structure();
  const CholSymbolic& S = chol_->symbolic();
  if (pose_owner) std::copy(S.pose_owner.begin(), S.pose_owner.end(), pose_owner);
  if (block_consumer) std::copy(S.block_consumer.begin(), S.block_consumer.end(), block_consumer);
}

void BlockSolver::solve_back_substitute() {
  require_structure();
  if (!schur_) return;
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  solve_back_substitute_impl();
}

void BlockSolver::solve_back_substitute_impl() {
  if (profiling) tb_.start(st_);
  const size_t sizeP = (size_t)nP_ * p_;
  prof.begin(KernelProf::kBackSub, st_);
  if (ba_recompute_ok()) {
    // fused EdgeProjectXYZ2UV system, estimates unchanged since build_system: Hpl' x_p from the Jacobians, Hpl is not read
    EdgeSet& es = *sets_[ba_.set];
    const int GL = ba_lm_group();
    if (ba_fuse_landmarks && ba_.ll_slots_ok && n_tiles_ > 0) {
#define G2OHIP_BA_BACK_SLOTS(CC)                                                                                                   \
  hipLaunchKernelGGL((ba_back_substitute_slots_kernel<CC>), dim3(n_tiles_), dim3(kThreads), 0, st_, d_tile_lm0.p, ba_.tile_ll.p,    \
                     ba_.ll_rec.p, ba_.ll_row.p, ba_.ll_meas.p, ba_.ll_omega.p, ba_.cams.p, ba_.pts.p, ba_.f, ba_.cx, ba_.cy,       \
                     es.kernel_kind, es.delta, ba_.omega_identity ? 1 : 0, d_Dinv.p, d_b.p + sizeP, d_x.p, d_x.p + sizeP, ba_.ctab.p)
      if (ba_.n_classes > 1) G2OHIP_BA_BACK_SLOTS(true);
      else G2OHIP_BA_BACK_SLOTS(false);
#undef G2OHIP_BA_BACK_SLOTS
    } else
#define G2OHIP_BA_BACK(GG) G2OHIP_BA_BACK_(GG, false)
#define G2OHIP_BA_BACK_(GG, CC)                                                                                                    \
  hipLaunchKernelGGL((ba_back_substitute_kernel<GG, CC>), dim3(grid_for((size_t)nL_ * GG)), dim3(kThreads), 0, st_, nL_, es.vl_ptr.p, \
                     ba_.cams.p, ba_.pts.p, ba_.cam_lm.p, ba_.pt_lm.p, ba_.meas_lm.p, ba_.omega_lm.p, ba_.row_lm.p, ba_.f,          \
                     ba_.cx, ba_.cy, es.kernel_kind, es.delta, ba_.omega_identity ? 1 : 0, d_Dinv.p, d_b.p + sizeP, d_x.p,           \
                     d_x.p + sizeP, ba_.ctab.p)
    if (ba_.n_classes > 1) G2OHIP_BA_BACK_(8, true);
    else if (GL == 1) G2OHIP_BA_BACK(1);
    else if (GL == 4) G2OHIP_BA_BACK(4);
    else G2OHIP_BA_BACK(8);
#undef G2OHIP_BA_BACK
#undef G2OHIP_BA_BACK_
  } else if ((ensure_hpl(), false)) {
  } else
#define G2OHIP_BACK(P_, L_)                                                                                                    \
  if (p_ == P_ && l_ == L_)                                                                                                    \
    hipLaunchKernelGGL((back_substitute_kernel<P_, L_>), dim3(grid_for(nL_)), dim3(kThreads), 0, st_, nL_, d_pl_colptr.p,        \
                       d_pl_row.p, d_Hpl.p, d_Dinv.p, d_b.p + sizeP, d_x.p, d_x.p + sizeP);                                    \
  else
  G2OHIP_BACK(6, 3)
  G2OHIP_BACK(3, 2)
  G2OHIP_BACK(7, 3)
  G2OHIP_BACK(6, 2)
  G2OHIP_BACK(3, 3) { throw ArgFailure("unsupported (pose_dim, landmark_dim)"); }
#undef G2OHIP_BACK
  prof.end(KernelProf::kBackSub, st_);
  G2OHIP_HIP_CHECK(hipGetLastError());
  if (profiling) {
    tb_.stop(st_);
    times.backsub = tb_.seconds();
  }
}

// ---- matrix-free PCG on the reduced system (linear_solver 2) -------------------------------------------
// out_i = init_i (+ lam d_i) + sign * sum_{blocks q of pose i} Hpl_q t[lm(q)]      (thread = scalar row of a pose)
template <int PD, int LD>
__global__ void __launch_bounds__(kThreads) mf_pose_kernel(int nP, const int* __restrict__ pm_ptr, const int* __restrict__ pm_q,
                                                         const int* __restrict__ pm_lm, const double* __restrict__ Hpl,
                                                         const double* __restrict__ tl, const double* init,
                                                         const double* __restrict__ dvec, const double* __restrict__ lam, double sign,
                                                         double* out) {   // (init may be out)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nP * PD) return;
  const int i = t / PD, r = t - i * PD;
  double acc = 0.0;
  for (int k = pm_ptr[i]; k < pm_ptr[i + 1]; ++k) {
    const double* B = Hpl + (size_t)pm_q[k] * PD * LD + r;
    const double* tv = tl + (size_t)pm_lm[k] * LD;
#pragma unroll
    for (int c = 0; c < LD; ++c) acc += B[PD * c] * tv[c];
  }
  double v = init[t];
  if (dvec) v += lam[0] * dvec[t];
  out[t] = v + sign * acc;
}
// diagonal blocks of the reduced system: Hpp_ii + lam I - sum_q B_q Dinv B_q'   (thread = element of a block)
template <int PD, int LD>
__global__ void __launch_bounds__(kThreads) mf_diag_kernel(int nP, const int* __restrict__ pm_ptr, const int* __restrict__ pm_q,
                                                         const int* __restrict__ pm_lm, const double* __restrict__ Hpl,
                                                         const double* __restrict__ Dinv, const double* __restrict__ Hpp,
                                                         const int* __restrict__ pp_diag, const double* __restrict__ lam,
                                                         double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nP * PD * PD) return;
  const int i = t / (PD * PD), e = t - i * PD * PD, r = e % PD, c = e / PD;
  double acc = 0.0;
  for (int k = pm_ptr[i]; k < pm_ptr[i + 1]; ++k) {
    const double* B = Hpl + (size_t)pm_q[k] * PD * LD;
    const double* D = Dinv + (size_t)pm_lm[k] * LD * LD;
#pragma unroll
    for (int b = 0; b < LD; ++b) {
      double w = 0.0;
#pragma unroll
      for (int a = 0; a < LD; ++a) w += B[r + PD * a] * D[a + LD * b];
      acc += w * B[c + PD * b];
    }
  }
  out[t] = Hpp[(size_t)pp_diag[i] * PD * PD + e] + (r == c ? lam[0] : 0.0) - acc;
}

void BlockSolver::mf_prepare_lists() {
  if (!schur_) throw StateFailure("the matrix-free reduced operator needs the Schur complement mode");
  if (mf_ready_) return;
  mf_ready_ = true;
  const size_t sizeP = (size_t)nP_ * p_;
  // pose-major view of the Hpl pattern (stored landmark-major)
  std::vector<int> ptr(nP_ + 1, 0), qq(pl_row.size()), lm(pl_row.size());
  for (int r : pl_row) ptr[r + 1]++;
  for (int i = 0; i < nP_; ++i) ptr[i + 1] += ptr[i];
  std::vector<int> w(ptr.begin(), ptr.end() - 1);
  for (int l = 0; l < nL_; ++l)
    for (int q = pl_colptr[l]; q < pl_colptr[l + 1]; ++q) {
      const int k = w[pl_row[q]]++;
      qq[k] = q;
      lm[k] = l;
    }
  if (qq.empty()) { qq.push_back(0); lm.push_back(0); }
  d_pm_ptr.upload(ptr, st_);
  d_pm_q.upload(qq, st_);
  d_pm_lm.upload(lm, st_);
  d_mf_l.alloc((size_t)std::max(nL_, 1) * l_);
  d_mf_diag.alloc(std::max(sizeP * p_, (size_t)1));
  d_mf_zero.alloc(std::max(sizeP, (size_t)nL_ * l_) + 1);
  d_mf_zero.zero(st_);
  pcg_hpp_ = std::make_unique<BlockPCG>(p_);
  pcg_hpp_->analyze(nP_, pp_colptr.data(), pp_row.data(), st_);
}

#define G2OHIP_MF_DISPATCH(BODY)                                                                              \
  if (p_ == 6 && l_ == 3) { constexpr int P_ = 6, L_ = 3; BODY }                                              \
  else if (p_ == 3 && l_ == 2) { constexpr int P_ = 3, L_ = 2; BODY }                                         \
  else if (p_ == 7 && l_ == 3) { constexpr int P_ = 7, L_ = 3; BODY }                                         \
  else throw ArgFailure("unsupported (pose_dim, landmark_dim) for the matrix-free reduced operator");

// Dinv = (Hll + lam_l I)^-1, bschur = b_p - Hpl Dinv b_l, diagonal blocks of the reduced system (device array 107)
void BlockSolver::schur_operator_prepare() {
  require_structure();
  if (!system_built_) throw StateFailure("schur_operator_prepare before build_system");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  mf_prepare_lists();
  ensure_hpl();
  const size_t sizeP = (size_t)nP_ * p_;
  const int gl = grid_for(nL_), gp = grid_for(sizeP);
  G2OHIP_MF_DISPATCH(
    const int gd = grid_for(sizeP * P_);
    hipLaunchKernelGGL((landmark_inverse_kernel<L_>), dim3(gl), dim3(kThreads), 0, st_, nL_, d_Hll.p, d_b.p + sizeP, d_Dinv.p, d_db.p,
                       d_lam.p);
    hipLaunchKernelGGL((back_substitute_kernel<P_, L_>), dim3(gl), dim3(kThreads), 0, st_, nL_, d_pl_colptr.p, d_pl_row.p, d_Hpl.p,
                       d_Dinv.p, d_b.p + sizeP, d_mf_zero.p, d_mf_l.p);   // Dinv b_l
    hipLaunchKernelGGL((mf_pose_kernel<P_, L_>), dim3(gp), dim3(kThreads), 0, st_, nP_, d_pm_ptr.p, d_pm_q.p, d_pm_lm.p, d_Hpl.p,
                       d_mf_l.p, d_b.p, (const double*)nullptr, d_lam.p, -1.0, d_bschur.p);
    hipLaunchKernelGGL((mf_diag_kernel<P_, L_>), dim3(gd), dim3(kThreads), 0, st_, nP_, d_pm_ptr.p, d_pm_q.p, d_pm_lm.p, d_Hpl.p,
                       d_Dinv.p, d_Hpp.p, d_pp_diag.p, d_lam.p, d_mf_diag.p);
  )
  hschur_valid_ = false;
  G2OHIP_HIP_CHECK(hipGetLastError());
}

// dout = (Hpp + lam_p I - Hpl Dinv Hpl') din on device vectors of nP * p doubles (after schur_operator_prepare)
void BlockSolver::schur_operator_apply(const double* din, double* dout) {
  if (!mf_ready_) throw StateFailure("schur_operator_apply before schur_operator_prepare");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  const size_t sizeP = (size_t)nP_ * p_;
  const int gl = grid_for(nL_), gp = grid_for(sizeP);
  pcg_hpp_->multiply(d_Hpp.p, din, dout, st_);
  G2OHIP_MF_DISPATCH(
    hipLaunchKernelGGL((back_substitute_kernel<P_, L_>), dim3(gl), dim3(kThreads), 0, st_, nL_, d_pl_colptr.p, d_pl_row.p, d_Hpl.p,
                       d_Dinv.p, d_mf_zero.p, din, d_mf_l.p);   // -Dinv Hpl' d
    hipLaunchKernelGGL((mf_pose_kernel<P_, L_>), dim3(gp), dim3(kThreads), 0, st_, nP_, d_pm_ptr.p, d_pm_q.p, d_pm_lm.p, d_Hpl.p,
                       d_mf_l.p, dout, din, d_lam.p, 1.0, dout);
  )
}
#undef G2OHIP_MF_DISPATCH

int BlockSolver::solve_matrix_free() {
  if (chol_opt.world > 1) throw StateFailure("linear_solver 2 inside the library: one GPU (the sharded form lives in distributed.py)");
  mf_prepare_lists();
  if (!pcg_mf_) {
    pcg_mf_ = std::make_unique<BlockPCG>(p_);
    pcg_mf_->analyze_operator(nP_, st_);
  }
  pcg_mf_->opt = pcg_opt;
  if (profiling) tn_.start(st_);
  prof.begin(KernelProf::kCholFactor, st_);
  schur_operator_prepare();
  auto apply = [&](const double* din, double* dout) { schur_operator_apply(din, dout); };
  const bool ok = pcg_mf_->solve_operator(d_mf_diag.p, apply, d_bschur.p, d_x.p, st_);
  prof.end(KernelProf::kCholFactor, st_);
  if (profiling) {
    tn_.stop(st_);
    times.numeric = tn_.seconds();
    times.linsolve = 0.0;
  }
  pcg_iterations = pcg_mf_->last_iterations();
  return ok ? 0 : 1;
}

int BlockSolver::solve() {
  if (!system_built_) throw StateFailure("solve before build_system");
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (linear_solver == 2) {   // matrix-free PCG: neither the Schur tiles nor the reduction run
    const int rc = solve_matrix_free();
    if (rc != 0) return rc;
    solve_back_substitute();
    return 0;
  }
  const bool virt = virtual_reduced_ok();
  if (virt != virt_now_) invalidate_graphs();   // the captured factor segment is specific to the source of the matrix
  virt_now_ = virt;
  if (schur_) solve_schur_impl(!virt);
  int rc = solve_reduced_impl();
  if (rc != 0) return rc;
  solve_back_substitute();
  return 0;
}

void BlockSolver::solve_async() {
  if (!system_built_) throw StateFailure("solve before build_system");
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (linear_solver != 0) {   // (the iterative solver synchronises anyway: keep its status for trial_stats)
    sync_status_ = solve();
    return;
  }
  const bool virt = virtual_reduced_ok();
  if (virt != virt_now_) invalidate_graphs();
  virt_now_ = virt;
  if (schur_) solve_schur_impl(!virt);
  solve_reduced_device();
  if (schur_) solve_back_substitute_impl();
  deferred_status_ = true;
}

// The queued half of trial_stats(): the reduction kernels, the read-back of their partial sums and of the status word of an
// asynchronous solve into pinned memory, and an event behind them.  A caller that has host work of its own (the adapter's
// look-ahead: the write-back of the accepted estimates into the vertices while the device runs the NEXT iteration's first
// trial) calls this, does that work, and calls trial_stats() -- which then only waits for the event and sums.
void BlockSolver::trial_stats_begin(double lambda) {
  if (trial_.begun) throw StateFailure("trial_stats_begin: the previous one has not been read (trial_stats)");
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  constexpr int kMaxBlocks = 1024;
  const size_t nsets = sets_.size();
  if (d_red_multi.n < (nsets + 1) * kMaxBlocks) d_red_multi.alloc((nsets + 1) * kMaxBlocks);
  std::vector<int>& nblk = trial_.nblk;
  nblk.assign(nsets + 1, 0);
  // slot 0: computeScale (optimization_algorithm_levenberg.cpp:165-172)
  nblk[0] = std::min(kMaxBlocks, grid_for(vector_size()));
  hipLaunchKernelGGL(scale_partial_kernel, dim3(nblk[0]), dim3(kThreads), 0, st_, vector_size(), d_x.p, d_b.p, lambda, d_red_multi.p);
  // slots 1..: chi2 of every edge set (same kernels as chi2())
  const bool need_chi = !chi2_valid_;
  if (need_chi && ba_.set >= 0 && ba_.n_classes > 1 && !ba_.err_valid) ba_linearize(false);
  for (size_t k = 0; k < nsets && need_chi; ++k) {
    EdgeSet& es = *sets_[k];
    if (es.n == 0 || (es.parts & 4)) continue;
    if (!es.has_err) throw StateFailure("trial_stats: edge data missing");
    nblk[k + 1] = std::min(kMaxBlocks, grid_for(es.n));
    if (ba_.err_valid && (int)k == ba_.set) {   // already there: ba_linearize left this set's partial sums in its slot
      nblk[k + 1] = kMaxBlocks;
      continue;
    }
    double* red = d_red_multi.p + (k + 1) * kMaxBlocks;
#define G2OHIP_CHI(d_)                                                                                                      \
  case d_:                                                                                                                  \
    hipLaunchKernelGGL((chi2_kernel<d_>), dim3(nblk[k + 1]), dim3(kThreads), 0, st_, es.n, es.omega, es.err, es.kernel_kind, es.delta, red, es.rk.p); \
    break
    switch (es.d) {
      G2OHIP_CHI(1);
      G2OHIP_CHI(2);
      G2OHIP_CHI(3);
      G2OHIP_CHI(6);
      G2OHIP_CHI(7);
      default:
        throw ArgFailure("chi2: unsupported error dimension");
    }
#undef G2OHIP_CHI
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
  // read-back into pinned memory: the partial sums and (asynchronous solve) the status word of the factorisation are two
  // copies behind each other and ONE synchronisation (a pageable destination made each copy a host round trip of its own)
  const size_t hn = (nsets + 1) * kMaxBlocks;
  if (h_trial_n_ < hn + 1) {
    if (h_trial_) (void)hipHostFree(h_trial_);
    h_trial_ = nullptr;
    G2OHIP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_trial_), (hn + 1) * sizeof(double), hipHostMallocDefault));
    h_trial_n_ = hn + 1;
  }
  double* h = h_trial_;
  size_t copy_n = 0;   // only the slots in use
  for (size_t k = 0; k <= nsets; ++k)
    if (nblk[k] > 0) copy_n = (k + 1) * kMaxBlocks;
  G2OHIP_HIP_CHECK(hipMemcpyAsync(h, d_red_multi.p, copy_n * sizeof(double), hipMemcpyDeviceToHost, st_));
  int* hstat = reinterpret_cast<int*>(h + hn);
  trial_.mode = 0;
  trial_.bad = false;
  if (sync_status_ >= 0) {
    trial_.bad = sync_status_ != 0;
    sync_status_ = -1;
  } else if (deferred_status_) {
    chol_->status_async(hstat, st_);
    deferred_status_ = false;
    trial_.mode = 1;
  }
  if (!trial_ev_) G2OHIP_HIP_CHECK(hipEventCreateWithFlags(&trial_ev_, hipEventDisableTiming));
  G2OHIP_HIP_CHECK(hipEventRecord(trial_ev_, st_));
  trial_.need_chi = need_chi;
  trial_.hn = hn;
  trial_.begun = true;
}

void BlockSolver::trial_stats(double lambda, int* ok, double* chi2_out, double* scale_out) {
  if (!trial_.begun) trial_stats_begin(lambda);
  constexpr int kMaxBlocks = 1024;
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  trial_.begun = false;
  G2OHIP_HIP_CHECK(hipEventSynchronize(trial_ev_));
  const size_t nsets = trial_.nblk.size() - 1;
  const std::vector<int>& nblk = trial_.nblk;
  const bool need_chi = trial_.need_chi;
  const double* h = h_trial_;
  bool bad = trial_.bad, stalled = false;
  if (trial_.mode == 1) {
    const int* hstat = reinterpret_cast<const int*>(h + trial_.hn);
    bad = chol_->failed_with(*hstat, st_);
    if (bad && chol_->dependency_stall() && ++dependency_fallbacks) {
      invalidate_graphs();   // the next solve runs level by level
      stalled = true;        // NOT "not positive definite": the caller repeats the trial (solve() does that itself)
    }
  }
  auto sum = [&](size_t slot) {
    double s = 0.0;
    for (int i = 0; i < nblk[slot]; ++i) s += h[slot * kMaxBlocks + i];   // fixed order: deterministic
    return s;
  };
  *scale_out = sum(0);
  if (need_chi) {
    double total = 0.0;
    for (size_t k = 0; k < nsets; ++k) total += sum(k + 1);
    chi2_value_ = total;
    chi2_valid_ = true;
    for (auto& esp : sets_)
      if (esp->external) chi2_valid_ = false;
  }
  *chi2_out = chi2_value_;
  *ok = stalled ? 2 : (bad ? 0 : 1);
}

void BlockSolver::multiply_hessian(double* dest_host, const double* src_host) {
  require_structure();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  const size_t n = vector_size();
  ensure_hpl();
  DevBuf<double> src, dst, tmp;
  src.upload(src_host, n, st_);
  dst.upload(dest_host, n, st_);
  const size_t sizeP = (size_t)nP_ * p_;
  if (!mh_hpp_) {   // gather-form lists, once per structure
    mh_hpp_ = std::make_unique<BlockPCG>(p_);
    mh_hpp_->analyze(nP_, pp_colptr.data(), pp_row.data(), st_);
    if (nL_ > 0) {
      std::vector<int> ptr(nP_ + 1, 0), qq(pl_row.size()), lm(pl_row.size());
      for (int r : pl_row) ptr[r + 1]++;
      for (int i = 0; i < nP_; ++i) ptr[i + 1] += ptr[i];
      std::vector<int> w(ptr.begin(), ptr.end() - 1);
      for (int l = 0; l < nL_; ++l)
        for (int q = pl_colptr[l]; q < pl_colptr[l + 1]; ++q) {
          const int k = w[pl_row[q]]++;
          qq[k] = q;
          lm[k] = l;
        }
      if (qq.empty()) { qq.push_back(0); lm.push_back(0); }
      d_mh_ptr.upload(ptr, st_);
      d_mh_q.upload(qq, st_);
      d_mh_lm.upload(lm, st_);
    }
  }
  if (sizeP > 0) {
    tmp.alloc(sizeP);
    mh_hpp_->multiply(d_Hpp.p, src.p, tmp.p, st_);
    hipLaunchKernelGGL(add_vec_kernel, dim3(grid_for(sizeP)), dim3(kThreads), 0, st_, sizeP, tmp.p, dst.p);
  }
  if (nL_ > 0) {
    if (sizeP > 0)
      hipLaunchKernelGGL(spmv_pl_pose_kernel, dim3(grid_for(sizeP)), dim3(kThreads), 0, st_, nP_, p_, l_, sizeP, d_mh_ptr.p, d_mh_q.p,
                         d_mh_lm.p, d_Hpl.p, src.p, dst.p);
    hipLaunchKernelGGL(spmv_pl_landmark_kernel, dim3(grid_for((size_t)nL_ * l_)), dim3(kThreads), 0, st_, nL_, p_, l_, sizeP,
                       d_pl_colptr.p, d_pl_row.p, d_Hpl.p, d_Hll.p, src.p, dst.p);
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
  dst.download(dest_host, n, st_);
  if (lam_pose_ != 0.0 || lam_lm_ != 0.0) {   // virtual damping
    const size_t np = (size_t)nP_ * p_;
    for (size_t i = 0; i < n; ++i) {
      if (i < np && !lam_mask_h_.empty() && !lam_mask_h_[i / p_]) continue;
      dest_host[i] += (i < np ? lam_pose_ : lam_lm_) * src_host[i];
    }
  }
}

void BlockSolver::set_x(const double* h) {
  require_structure();
  if (!h) throw ArgFailure("set_x: null vector");
  d_x.upload(h, vector_size(), st_);
}
void BlockSolver::copy_x(double* h) {
  require_structure();
  d_x.download(h, vector_size(), st_);
}
void BlockSolver::copy_b(double* h) {
  require_structure();
  ensure_bl();
  d_b.download(h, vector_size(), st_);
}
void BlockSolver::sync() { G2OHIP_HIP_CHECK(hipStreamSynchronize(st_)); }

int BlockSolver::nnzb(int which) const {
  switch (which) {
    case 0: return (int)pp_row.size();
    case 1: return (int)pl_row.size();
    case 2: return nL_;
    case 3: return (int)hs_row.size();
    case 4: return nL_;
    default: throw ArgFailure("bad matrix selector");
  }
}
void BlockSolver::get_pattern(int which, int* colptr, int* rowidx) const {
  const std::vector<int>*cp, *ri;
  switch (which) {
    case 0: cp = &pp_colptr; ri = &pp_row; break;
    case 1: cp = &pl_colptr; ri = &pl_row; break;
    case 3: cp = &hs_colptr; ri = &hs_row; break;
    default: throw ArgFailure("pattern available for HPP, HPL, HSCHUR");
  }
  std::copy(cp->begin(), cp->end(), colptr);
  std::copy(ri->begin(), ri->end(), rowidx);
}
void BlockSolver::copy_values(int which, double* h) {
  require_structure();
  switch (which) {
    case 0:
      d_Hpp.download(h, pp_row.size() * p_ * p_, st_);
      if (lam_pose_ != 0.0)
        for (int c = 0; c < nP_; ++c) {
          if (!lam_mask_h_.empty() && !lam_mask_h_[c]) continue;
          double* blk = h + (size_t)(pp_colptr[c + 1] - 1) * p_ * p_;   // the diagonal block closes its column
          for (int i = 0; i < p_; ++i) blk[i * (p_ + 1)] += lam_pose_;
        }
      break;
    case 1: ensure_hpl(); d_Hpl.download(h, pl_row.size() * p_ * l_, st_); break;
    case 2:
      ensure_ll();
      d_Hll.download(h, (size_t)nL_ * l_ * l_, st_);
      if (lam_lm_ != 0.0)
        for (size_t j = 0; j < (size_t)nL_; ++j)
          for (int i = 0; i < l_; ++i) h[j * l_ * l_ + i * (l_ + 1)] += lam_lm_;
      break;
    case 3: ensure_hschur(); d_Hschur.download(h, hs_row.size() * p_ * p_, st_); break;
    case 4: d_Dinv.download(h, (size_t)nL_ * l_ * l_, st_); break;
    default: throw ArgFailure("bad matrix selector");
  }
}
// ---- bundle-adjustment front end -------------------------------------------------------------
void BlockSolver::ba_set_edges(int set, const int* cam_vertex, const int* point_vertex, const double* meas, const double* info,
                               double f, double cx, double cy) {
  ba_set_edges_classes(set, cam_vertex, point_vertex, meas, info, f, cx, cy, 1, nullptr, nullptr);
}

// ---- permuted copies of the per-observation inputs, gathered ON THE DEVICE from the arrays in edge order (ba_set_edges): the
// host used to fill and upload each of them (0.6 GB over PCIe at the metric configuration)
// per Hpl block (block order; every block has one observation): scatter by the edge's block
__global__ void ba_gather_blocks_kernel(size_t n, const int* __restrict__ edge_hpl, const int* __restrict__ cam_v, const int* __restrict__ pt_v,
                                        const double* __restrict__ meas, const double* __restrict__ omega, int* __restrict__ cam_q,
                                        int* __restrict__ pt_q, double* __restrict__ meas_q, double* __restrict__ omega_q) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int q = edge_hpl[k];
  if (q < 0) return;
  cam_q[q] = cam_v[k];
  pt_q[q] = pt_v[k];
  meas_q[2 * (size_t)q] = meas[2 * k];
  meas_q[2 * (size_t)q + 1] = meas[2 * k + 1];
  if (omega_q)
    for (int i = 0; i < 4; ++i) omega_q[4 * (size_t)q + i] = omega[4 * k + i];
}
// observation-list order of a vertex side (ent = edge << 1 | side); hpl / row only for the landmark side
__global__ void ba_gather_lists_kernel(size_t n, const int* __restrict__ ent, const int* __restrict__ cam_v, const int* __restrict__ pt_v,
                                       const double* __restrict__ meas, const double* __restrict__ omega, const int* __restrict__ edge_hpl,
                                       const int* __restrict__ v1, int* __restrict__ cam_o, int* __restrict__ pt_o,
                                       double* __restrict__ meas_o, double* __restrict__ omega_o, int* __restrict__ hpl_o,
                                       int* __restrict__ row_o) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const size_t e = (size_t)(ent[k] >> 1);
  cam_o[k] = cam_v[e];
  pt_o[k] = pt_v[e];
  meas_o[2 * k] = meas[2 * e];
  meas_o[2 * k + 1] = meas[2 * e + 1];
  if (omega_o)
    for (int i = 0; i < 4; ++i) omega_o[4 * k + i] = omega[4 * e + i];
  if (hpl_o) {
    const int h = edge_hpl[e];
    hpl_o[k] = h;
    row_o[k] = h >= 0 ? v1[e] : -1;
  }
}
// lane slots of the tiles: what a lane reads about its observation, slot-major (one workgroup per tile)
__global__ void ba_gather_slots_kernel(const int4* __restrict__ tile_ll, const int* __restrict__ slots, const int* __restrict__ vl_ent,
                                       const int* __restrict__ cam_lm, const int* __restrict__ pt_lm, const int* __restrict__ hpl_lm,
                                       const int* __restrict__ row_lm, const double* __restrict__ meas_lm, const double* __restrict__ omega_lm,
                                       int4* __restrict__ rec, int* __restrict__ edge, int* __restrict__ srow, double* __restrict__ ms,
                                       double* __restrict__ os) {
  const int4 tl = tile_ll[blockIdx.x];
  for (int i = threadIdx.x; i < tl.y; i += blockDim.x) {
    const size_t sidx = (size_t)tl.x + i;
    const int w = slots[sidx];
    int4 r = make_int4(w, 0, 0, -1);
    int ed = 0, row = -1;
    double m0 = 0.0, m1 = 0.0, o[4] = {0.0, 0.0, 0.0, 0.0};
    if ((w & 0xfff) != 0xfff) {
      const size_t k = (size_t)tl.z + (w & 0xfff);
      r = make_int4(w, cam_lm[k], pt_lm[k], hpl_lm[k]);
      ed = vl_ent[k] >> 1;
      row = row_lm[k];
      m0 = meas_lm[2 * k];
      m1 = meas_lm[2 * k + 1];
      if (os)
        for (int j = 0; j < 4; ++j) o[j] = omega_lm[4 * k + j];
    }
    rec[sidx] = r;
    edge[sidx] = ed;
    srow[sidx] = row;
    ms[2 * sidx] = m0;
    ms[2 * sidx + 1] = m1;
    if (os)
      for (int j = 0; j < 4; ++j) os[4 * sidx + j] = o[j];
  }
}

// Edge classes: class_params[5 c] = (focal length, principal point x, y, robust kernel kind, delta) of class c, edge_class[k]
// the class of edge k.  One class (n_classes == 1, both arrays may be null): the intrinsics are the three scalars and the
// robust kernel is the edge set's (set_robust_kernel), exactly ba_set_edges.  More: class 0 also takes its values from the
// table; set_robust_kernel on the set is refused while the classes are bound (the table owns the kernels).
void BlockSolver::ba_set_edges_classes(int set, const int* cam_vertex, const int* point_vertex, const double* meas, const double* info,
                                       double f, double cx, double cy, int n_classes, const double* class_params, const int* edge_class) {
  invalidate_graphs();
  require_structure();
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("bad edge set id");
  EdgeSet& es = *sets_[set];
  if (es.d != 2 || es.unary || es.dim0 != 3 || es.dim1 != 6 || p_ != 6 || l_ != 3)
    throw ArgFailure("ba_set_edges: the set must be EdgeProjectXYZ2UV-shaped (d=2, vertex0 = 3-dof point, vertex1 = 6-dof pose)");
  if (!cam_vertex || !point_vertex || !meas) throw ArgFailure("ba_set_edges: null array");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  const size_t n = (size_t)es.n;
  const bool lapt = getenv("G2OHIP_SETUP_TIMING") != nullptr;
  auto lap_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double lap_t = lap_now();
  auto lap = [&](const char* what) {
    if (!lapt) return;
    const double t = lap_now();
    fprintf(stderr, "ba_set_edges: %-37s %.3f s\n", what, t - lap_t);
    lap_t = t;
  };
  if (n_classes < 1 || n_classes > 128) throw ArgFailure("ba_set_edges_classes: 1 to 128 edge classes");
  if (n_classes > 1 && (!class_params || !edge_class)) throw ArgFailure("ba_set_edges_classes: null class table");
  // Everything that can refuse the binding is checked BEFORE the front end or the edge set is changed: a failed call leaves both
  // as they were.
  if (es.rk.p)   // (symmetric with set_robust_kernel_per_edge, which refuses a bound set: the fused kernels never read es.rk)
    throw StateFailure("ba_set_edges: the set carries per-edge robust kernels (set_robust_kernel_per_edge); on the BA front end they go through ba_set_edges_classes -- clear them first");
  std::vector<int> camc;   // camera index | class << 24: what every device copy of the camera index carries
  if (n_classes > 1) {
    for (int c = 0; c < n_classes; ++c) {
      const double kd = class_params[5 * c + 3];
      if (!(kd == 0.0 || kd == 1.0 || kd == 2.0 || kd == 3.0 || kd == 4.0 || kd == 5.0)) throw ArgFailure("ba_set_edges_classes: robust kernel kind of a class must be 0..5");
      if (kd != 0.0 && !(class_params[5 * c + 4] > 0.0)) throw ArgFailure("ba_set_edges_classes: robust kernel delta of class " + std::to_string(c) + " must be positive");
    }
    camc.resize(n);
    for (size_t k = 0; k < n; ++k) {
      if (edge_class[k] < 0 || edge_class[k] >= n_classes) throw ArgFailure("ba_set_edges_classes: class of edge " + std::to_string(k) + " outside the table");
      if (cam_vertex[k] < 0 || cam_vertex[k] >= (1 << 24)) throw ArgFailure("ba_set_edges_classes: camera indices are limited to 2^24 with edge classes");
      camc[k] = cam_vertex[k] | (edge_class[k] << 24);
    }
  }
  ba_validate_edges(es, cam_vertex, point_vertex, n);
  // Hpl block written by each edge; the fused assembly requires one observation per (pose, landmark) pair
  std::vector<int> edge_hpl(n, -1);
  bool unique = es.first_lm && es.first_ol;
  {
    std::vector<char> seen(pl_row.size(), 0);
    for (size_t k = 0; k < n; ++k)
      if (es.v0[k] < 0) unique = false;   // a fixed landmark: its edges would be skipped by the landmark-major kernel
    host_parallel_for(n, [&](size_t b_, size_t e_) {   // (the searches in parallel, the duplicate check in edge order)
      for (size_t k = b_; k < e_; ++k) {
        const int a = es.v0[k], b = es.v1[k];
        if (a >= 0 && b >= 0) edge_hpl[k] = find_block(pl_colptr, pl_row, a - nP_, b);
      }
    });
    for (size_t k = 0; k < n; ++k) {
      const int q = edge_hpl[k];
      if (q < 0) continue;
      if (seen[q]) unique = false;
      seen[q] = 1;
    }
    if (n_classes > 1 && !unique)
      throw ArgFailure("ba_set_edges_classes: edge classes need the fused path (one observation per (pose, landmark) pair, no fixed landmark)");
  }
  lap("validation + Hpl block per edge");
  ba_.set = set;
  ba_.n_edges = es.n;
  ba_.f = f; ba_.cx = cx; ba_.cy = cy;
  ba_.n_classes = n_classes;
  if (n_classes > 1) {
    ba_.h_ctab.assign(class_params, class_params + 5 * (size_t)n_classes);
    ba_.ctab.upload(ba_.h_ctab, st_);
    ba_.f = class_params[0]; ba_.cx = class_params[1]; ba_.cy = class_params[2];
    es.kernel_kind = (int)class_params[3];
    es.delta = class_params[4];
  } else {
    ba_.h_ctab.clear();
  }
  ba_.h_cam_v.assign(cam_vertex, cam_vertex + n);
  ba_.h_pt_v.assign(point_vertex, point_vertex + n);
  if (n_classes > 1) cam_vertex = camc.data();   // (validated plain; from here on the indices carry the class)
  ba_.err_valid = ba_.jac_valid = false;
  chi2_valid_ = false;
  es.has_err = false;      // errors of the previous edge data are gone
  ba_.cam_v.upload(cam_vertex, n, st_);
  ba_.pt_v.upload(point_vertex, n, st_);
  ba_.meas.upload(meas, n * 2, st_);
  ba_.omega_identity = info == nullptr;
  if (!info) {   // information().setIdentity(): written on the device (160 MB less to fill and to send at the metric configuration)
    es.own_omega.alloc(n * 4);
    if (n) hipLaunchKernelGGL(identity2_kernel, dim3(grid_for(n)), dim3(kThreads), 0, st_, n, es.own_omega.p);
  } else {
    es.own_omega.upload(info, n * 4, st_);
  }
  lap("uploads in edge order");
  const double* d_omega = ba_.omega_identity ? (const double*)nullptr : es.own_omega.p;   // (identity: not read, no permuted copies)
  auto zero_alloc = [&](auto& buf, size_t count) {
    buf.alloc(std::max<size_t>(count, 1));
    buf.zero(st_);
  };
  {
    ba_.fused_ok = unique;
    ba_.edge_hpl.upload(edge_hpl, st_);
    if (unique) {   // the observation behind every Hpl block, in block order (ba_schur_tile_kernel); blocks without one stay zero
      const size_t nq = std::max<size_t>(pl_row.size(), 1);
      zero_alloc(ba_.cam_q, nq);
      zero_alloc(ba_.pt_q, nq);
      zero_alloc(ba_.meas_q, nq * 2);
      if (d_omega) zero_alloc(ba_.omega_q, nq * 4);
      if (n)
        hipLaunchKernelGGL(ba_gather_blocks_kernel, dim3(grid_for(n)), dim3(kThreads), 0, st_, n, ba_.edge_hpl.p, ba_.cam_v.p, ba_.pt_v.p,
                           ba_.meas.p, d_omega, ba_.cam_q.p, ba_.pt_q.p, ba_.meas_q.p, d_omega ? ba_.omega_q.p : (double*)nullptr);
    }
  }
  lap("copies in Hpl block order");
  {
    // pose-major copies of what the pose-side assembly reads per observation (it walks a pose's observation list:
    // through the edge id these would be 16-byte gathers at an 80-byte stride), and the same for the landmark side
    // (observation-list order of the landmarks) with the Hpl block and its pose block row
    const size_t nk = es.h_vp_ent.size(), nl = es.h_vl_ent.size();
    DevBuf<int> d_v1;
    d_v1.upload(es.v1, st_);
    ba_.meas_pm.alloc(std::max<size_t>(nk * 2, 1));
    ba_.pt_pm.alloc(std::max<size_t>(nk, 1));
    ba_.cam_pm.alloc(std::max<size_t>(nk, 1));
    if (d_omega) ba_.omega_pm.alloc(std::max<size_t>(nk * 4, 1));
    if (nk)
      hipLaunchKernelGGL(ba_gather_lists_kernel, dim3(grid_for(nk)), dim3(kThreads), 0, st_, nk, es.vp_ent.p, ba_.cam_v.p, ba_.pt_v.p, ba_.meas.p,
                         d_omega, (const int*)nullptr, (const int*)nullptr, ba_.cam_pm.p, ba_.pt_pm.p, ba_.meas_pm.p,
                         d_omega ? ba_.omega_pm.p : (double*)nullptr, (int*)nullptr, (int*)nullptr);
    ba_.meas_lm.alloc(std::max<size_t>(nl * 2, 1));
    ba_.cam_lm.alloc(std::max<size_t>(nl, 1));
    ba_.pt_lm.alloc(std::max<size_t>(nl, 1));
    ba_.hpl_lm.alloc(std::max<size_t>(nl, 1));
    ba_.row_lm.alloc(std::max<size_t>(nl, 1));
    if (d_omega) ba_.omega_lm.alloc(std::max<size_t>(nl * 4, 1));
    if (nl)
      hipLaunchKernelGGL(ba_gather_lists_kernel, dim3(grid_for(nl)), dim3(kThreads), 0, st_, nl, es.vl_ent.p, ba_.cam_v.p, ba_.pt_v.p, ba_.meas.p,
                         d_omega, ba_.edge_hpl.p, d_v1.p, ba_.cam_lm.p, ba_.pt_lm.p, ba_.meas_lm.p,
                         d_omega ? ba_.omega_lm.p : (double*)nullptr, ba_.hpl_lm.p, ba_.row_lm.p);
    G2OHIP_HIP_CHECK(hipGetLastError());
    lap("pose-major + landmark-major copies");
    // lane slots of the tiles that assemble their own landmarks
    ba_.ll_slots_ok = false;
    if (schur_ && n_tiles_ > 0 && tiles_cover_all_ && n_split_tiles_ == 0 && (int)tile_lm0_h_.size() == n_tiles_ + 1 &&
        (int)es.h_vl_ptr.size() == nL_ + 1) {
      constexpr int kIdleSlot = (int)0x80000fffu;   // no observation, not a first lane, list length 0
      // two passes over the tiles on the host threads: slots per tile (a landmark never straddles a wavefront), then the words
      std::vector<int4> tl((size_t)n_tiles_);
      std::vector<int> cnt((size_t)n_tiles_, 0);
      std::atomic<bool> okf(true);
      auto walk = [&](int t, int* out, int& kmax) {   // out == nullptr: count only; returns the slots of tile t (-1: does not fit the word)
        const int l0 = tile_lm0_h_[t], l1 = tile_lm0_h_[t + 1];
        const int kb = es.h_vl_ptr[l0];
        int used = 0, total = 0;
        kmax = 0;
        for (int lm = l0; lm < l1; ++lm) {
          const int K = es.h_vl_ptr[lm + 1] - es.h_vl_ptr[lm], lmi = lm - l0;
          if (K > 64 || lmi >= 0x7ff || es.h_vl_ptr[lm + 1] - kb >= 0xfff) return -1;
          const int need = std::max(K, 1);
          if (used + need > 64) {
            if (out)
              for (int i = used; i < 64; ++i) out[total + i - used] = kIdleSlot;
            total += 64 - used;
            used = 0;
          }
          if (out) {
            if (K == 0) out[total] = 0xfff | (lmi << 20);
            for (int j = 0; j < K; ++j)
              out[total + j] = (es.h_vl_ptr[lm] + j - kb) | (K << 12) | (j == 0 ? (lmi << 20) : (int)(0x80000000u | ((unsigned)j << 20)));
          }
          total += need;
          used = (used + need) & 63;
          kmax = std::max(kmax, K);
        }
        if (used) {
          if (out)
            for (int i = used; i < 64; ++i) out[total + i - used] = kIdleSlot;
          total += 64 - used;
        }
        return total;
      };
      host_parallel_for((size_t)n_tiles_, [&](size_t t0_, size_t t1_) {
        for (size_t t = t0_; t < t1_; ++t) {
          int kmax = 0;
          const int c = walk((int)t, nullptr, kmax);
          if (c < 0) okf.store(false);
          cnt[t] = std::max(c, 0);
        }
      }, 64);
      size_t ns = 0;
      for (int t = 0; t < n_tiles_; ++t) {
        tl[t].x = (int)ns;
        ns += (size_t)cnt[t];
      }
      if (okf.load() && ns < ((size_t)1 << 31)) {
        std::vector<int> slots(std::max<size_t>(ns, 1), kIdleSlot);
        host_parallel_for((size_t)n_tiles_, [&](size_t t0_, size_t t1_) {   // (a tile's slots are its own)
          for (size_t t = t0_; t < t1_; ++t) {
            int kmax = 0;
            walk((int)t, slots.data() + tl[t].x, kmax);
            tl[t] = make_int4(tl[t].x, cnt[t], es.h_vl_ptr[tile_lm0_h_[t]], kmax);
          }
        }, 64);
        // slot-major copies of what a lane reads about its observation (one memory round trip after the tile record), gathered
        // on the device from the landmark-major arrays
        const size_t nsa = slots.size();
        DevBuf<int> d_slots;
        d_slots.upload(slots, st_);
        ba_.tile_ll.upload(tl, st_);
        ba_.ll_rec.alloc(nsa);
        ba_.ll_edge.alloc(nsa);
        ba_.ll_row.alloc(nsa);
        ba_.ll_meas.alloc(nsa * 2);
        if (d_omega) ba_.ll_omega.alloc(nsa * 4);
        if (ns == 0) {   // (no tile has a slot: the one idle record)
          const int4 idle = make_int4(kIdleSlot, 0, 0, -1);
          const int none = -1;
          ba_.ll_rec.upload(&idle, 1, st_);
          ba_.ll_row.upload(&none, 1, st_);
          ba_.ll_edge.zero(st_);
          ba_.ll_meas.zero(st_);
        } else {
          hipLaunchKernelGGL(ba_gather_slots_kernel, dim3(n_tiles_), dim3(kThreads), 0, st_, ba_.tile_ll.p, d_slots.p, es.vl_ent.p, ba_.cam_lm.p,
                             ba_.pt_lm.p, ba_.hpl_lm.p, ba_.row_lm.p, ba_.meas_lm.p, d_omega ? ba_.omega_lm.p : (const double*)nullptr,
                             ba_.ll_rec.p, ba_.ll_edge.p, ba_.ll_row.p, ba_.ll_meas.p, d_omega ? ba_.ll_omega.p : (double*)nullptr);
        }
        G2OHIP_HIP_CHECK(hipGetLastError());
        G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));   // (d_slots, d_v1 die with this scope)
        ba_.ll_slots_ok = true;
      }
    }
    lap("lane slots of the tiles");
    G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
  }
  lap("pose-major / landmark-major uploads");
  // (the Jacobian arrays -- 144 bytes per observation, 0.72 GB at the metric configuration -- only where something reads them: the
  // fused kernels evaluate the Jacobians themselves; otherwise ba_linearize allocates them when it is first asked for them)
  if (!(ba_fused && ba_.fused_ok)) {
    es.own_J0.alloc(n * 6);
    es.own_J1.alloc(n * 12);
  }
  es.own_err.alloc(n * 2);
  es.J0 = es.own_J0.p; es.J1 = es.own_J1.p; es.omega = es.own_omega.p; es.err = es.own_err.p;
  es.has_data = false;   // becomes valid with the first ba_linearize
  prepare_ba_tile_kernels();
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
  lap("buffers, kernel attributes, sync");
}

// Edge -> estimate indices against the estimate tables and against the edge set's hessian indices: a wrong index would be
// an out-of-bounds device read in the linearisation kernels, a mismatch a silently inconsistent system.  Runs as soon as
// both ba_set_edges and ba_set_estimates have been called (in either order).
// May the back-substitution re-evaluate the Jacobians instead of reading Hpl?  The system must come from the fused BA
// assembly of the CURRENT estimates (an LM trial solves before it updates; a rejected trial pops back to them), with the
// same robust kernel, and no other edge set may contribute pose-landmark blocks.
bool BlockSolver::ba_recompute_ok() const {
  if (!ba_recompute_backsub || !schur_ || p_ != 6 || l_ != 3 || ba_.set < 0 || !ba_fused || !ba_.fused_ok || ba_.n_cams <= 0) return false;
  if (ba_.sys_version < 0 || ba_.sys_version != ba_.est_version) return false;
  const EdgeSet& bs = *sets_[ba_.set];
  if (bs.kernel_kind != ba_.sys_kind || bs.delta != ba_.sys_delta) return false;
  for (size_t i = 0; i < sets_.size(); ++i)
    if ((int)i != ba_.set && sets_[i]->n > 0 && sets_[i]->touches_lm) return false;
  return true;
}

void BlockSolver::ba_validate() {
  if (ba_.set < 0 || ba_.h_cam_v.empty() || ba_.h_cam_hidx.empty()) return;
  ba_validate_edges(*sets_[ba_.set], ba_.h_cam_v.data(), ba_.h_pt_v.data(), ba_.h_cam_v.size());
}
// the edges' (camera, point) indices against the estimate tables and the edge set's vertices (nothing to check before the
// estimates are there: ba_set_estimates validates then)
void BlockSolver::ba_validate_edges(const EdgeSet& es, const int* cam_v, const int* pt_v, size_t n) const {
  if (n == 0 || ba_.h_cam_hidx.empty()) return;
  const int nc = (int)ba_.h_cam_hidx.size(), np = (int)ba_.h_pt_hidx.size();
  for (size_t k = 0; k < n; ++k) {
    const int c = cam_v[k], q = pt_v[k];
    if (c < 0 || c >= nc) throw ArgFailure("ba: camera index " + std::to_string(c) + " of edge " + std::to_string(k) + " outside the estimate table");
    if (q < 0 || q >= np) throw ArgFailure("ba: point index " + std::to_string(q) + " of edge " + std::to_string(k) + " outside the estimate table");
    const int hc = ba_.h_cam_hidx[c], hp = ba_.h_pt_hidx[q];
    if (hc >= nP_ || hp >= nL_) throw ArgFailure("ba: hessian index of an estimate outside the structure");
    const int v1 = hc < 0 ? -1 : hc, v0 = hp < 0 ? -1 : nP_ + hp;
    if (es.v1[k] != v1 || es.v0[k] != v0)
      throw ArgFailure("ba: edge " + std::to_string(k) + " connects vertices (" + std::to_string(es.v0[k]) + ", " + std::to_string(es.v1[k]) +
                       ") in the edge set but estimates with hessian indices (" + std::to_string(v0) + ", " + std::to_string(v1) + ")");
  }
}

void BlockSolver::pg_validate() {
  if (pg_.set < 0 || pg_.h_vi.empty() || pg_.h_hidx.empty()) return;
  const EdgeSet& es = *sets_[pg_.set];
  const int nv = (int)pg_.h_hidx.size();
  for (size_t k = 0; k < pg_.h_vi.size(); ++k) {
    const int a = pg_.h_vi[k], b = pg_.h_vj[k];
    if (a < 0 || a >= nv || b < 0 || b >= nv) throw ArgFailure("pg: vertex index of edge " + std::to_string(k) + " outside the estimate table");
    const int ha = pg_.h_hidx[a], hb = pg_.h_hidx[b];
    if (ha >= nP_ || hb >= nP_) throw ArgFailure("pg: hessian index of an estimate outside the structure");
    if (es.v0[k] != (ha < 0 ? -1 : ha) || es.v1[k] != (hb < 0 ? -1 : hb))
      throw ArgFailure("pg: edge " + std::to_string(k) + ": the estimates' hessian indices differ from the edge set's");
  }
}

void BlockSolver::ba_set_estimates(int n_cams, const double* cams, const int* cam_hidx, int n_points, const double* points,
                                   const int* point_hidx) {
  if (n_cams <= 0 || n_points <= 0 || !cams || !points || !cam_hidx || !point_hidx) throw ArgFailure("ba_set_estimates: bad arguments");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (fetch_pieces_ > 0) {   // (a read-back still in flight targets the caller's buffers -- possibly the very ones handed over here)
    G2OHIP_HIP_CHECK(hipStreamSynchronize(fetch_st_));
    fetch_pieces_ = 0;
  }
  // The caller of every iteration (the g2o adapter: setEstimate of all vertices before buildSystem) hands over the same
  // tables with new values: only the estimates move -- no index check, no re-upload of the index mapping, and the captured
  // launch sequences stay (the device addresses are the same).
  if (n_cams == ba_.n_cams && n_points == ba_.n_points && ba_.cams.p && ba_.pts.p && (int)ba_.h_cam_hidx.size() == n_cams &&
      (int)ba_.h_pt_hidx.size() == n_points && std::memcmp(ba_.h_cam_hidx.data(), cam_hidx, sizeof(int) * (size_t)n_cams) == 0 &&
      std::memcmp(ba_.h_pt_hidx.data(), point_hidx, sizeof(int) * (size_t)n_points) == 0) {
    ++ba_.est_version;
    ba_.err_valid = ba_.jac_valid = false;
    chi2_valid_ = false;
    ba_.has_backup = false;
    ba_.cams.upload(cams, (size_t)n_cams * 12, st_);
    ba_.pts.upload(points, (size_t)n_points * 3, st_);
    G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
    return;
  }
  invalidate_graphs();
  ba_.n_cams = n_cams;
  ba_.n_points = n_points;
  ++ba_.est_version;
  ba_.h_cam_hidx.assign(cam_hidx, cam_hidx + n_cams);
  ba_.h_pt_hidx.assign(point_hidx, point_hidx + n_points);
  ba_validate();
  ba_.cams.upload(cams, (size_t)n_cams * 12, st_);
  ba_.pts.upload(points, (size_t)n_points * 3, st_);
  ba_.cam_hidx.upload(cam_hidx, n_cams, st_);
  ba_.pt_hidx.upload(point_hidx, n_points, st_);
  ba_.cams_bak.alloc((size_t)n_cams * 12);
  ba_.pts_bak.alloc((size_t)n_points * 3);
  ba_.has_backup = false;
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
}

void BlockSolver::ba_get_estimates(double* cams, double* points) {
  if (ba_.n_cams <= 0) throw StateFailure("ba_get_estimates before ba_set_estimates");
  if (cams) ba_.cams.download(cams, (size_t)ba_.n_cams * 12, st_);
  if (points) ba_.pts.download(points, (size_t)ba_.n_points * 3, st_);
}

// The estimates of SELECTED vertices (what a caller with a few host-side edges needs of a trial: the adapter's hybrid loop reads the
// handful of cameras / points its host-linearised edges touch here instead of waiting for the whole read-back): one gather
// kernel, one small copy, one synchronisation.
__global__ void ba_gather_estimates_kernel(int nc, int np, const int* __restrict__ idx, const double* __restrict__ cams,
                                           const double* __restrict__ pts, double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < nc * 12) out[t] = cams[(size_t)idx[t / 12] * 12 + t % 12];
  else if (t < nc * 12 + np * 3) {
    const int u = t - nc * 12;
    out[t] = pts[(size_t)idx[nc + u / 3] * 3 + u % 3];
  }
}

void BlockSolver::ba_get_estimates_of(int n_cams, const int* cam_idx, double* cams, int n_points, const int* point_idx, double* points) {
  if (ba_.n_cams <= 0) throw StateFailure("ba_get_estimates_of before ba_set_estimates");
  if (n_cams < 0 || n_points < 0 || (n_cams > 0 && (!cam_idx || !cams)) || (n_points > 0 && (!point_idx || !points)))
    throw ArgFailure("ba_get_estimates_of: bad arguments");
  for (int i = 0; i < n_cams; ++i)
    if (cam_idx[i] < 0 || cam_idx[i] >= ba_.n_cams) throw ArgFailure("ba_get_estimates_of: camera index out of range");
  for (int i = 0; i < n_points; ++i)
    if (point_idx[i] < 0 || point_idx[i] >= ba_.n_points) throw ArgFailure("ba_get_estimates_of: point index out of range");
  const size_t n = (size_t)n_cams * 12 + (size_t)n_points * 3;
  if (n == 0) return;
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  std::vector<int> idx((size_t)n_cams + n_points);
  std::copy(cam_idx, cam_idx + n_cams, idx.begin());
  std::copy(point_idx, point_idx + n_points, idx.begin() + n_cams);
  ba_.sel_idx.upload(idx, st_);
  ba_.sel_out.alloc(n);
  hipLaunchKernelGGL(ba_gather_estimates_kernel, dim3(grid_for(n)), dim3(kThreads), 0, st_, n_cams, n_points, ba_.sel_idx.p, ba_.cams.p,
                     ba_.pts.p, ba_.sel_out.p);
  G2OHIP_HIP_CHECK(hipGetLastError());
  std::vector<double> h(n);
  ba_.sel_out.download(h.data(), n, st_);   // (synchronises)
  std::copy(h.begin(), h.begin() + (size_t)n_cams * 12, cams);
  std::copy(h.begin() + (size_t)n_cams * 12, h.end(), points);
}

// The same read-back started ASYNCHRONOUSLY behind everything queued on the solver's stream so far (the caller: right after
// ba_update of an LM trial), on a copy stream of its own, in pieces with an event each: piece 0 = the cameras, pieces 1 .. n = the
// points in n equal ranges.  ba_fetch_wait(k) returns once piece k is in the caller's buffer, so the caller writes piece k into
// its vertices while piece k + 1 is still crossing PCIe -- and the whole copy runs next to the error evaluation of the trial.
// Anything that WRITES the estimates afterwards (update, pop, set_estimates) makes the solver's stream wait for the copy first.
void BlockSolver::ba_fetch_begin(double* cams, double* points, int point_pieces) {
  if (ba_.n_cams <= 0) throw StateFailure("ba_fetch_begin before ba_set_estimates");
  if (!cams || !points || point_pieces < 1 || point_pieces > kFetchMaxPieces - 1) throw ArgFailure("ba_fetch_begin: bad arguments");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  if (!fetch_st_) {
    G2OHIP_HIP_CHECK(hipStreamCreateWithFlags(&fetch_st_, hipStreamNonBlocking));
    G2OHIP_HIP_CHECK(hipEventCreateWithFlags(&fetch_fork_, hipEventDisableTiming));
    for (int k = 0; k < kFetchMaxPieces; ++k) G2OHIP_HIP_CHECK(hipEventCreateWithFlags(&fetch_ev_[k], hipEventDisableTiming));
  }
  G2OHIP_HIP_CHECK(hipEventRecord(fetch_fork_, st_));
  G2OHIP_HIP_CHECK(hipStreamWaitEvent(fetch_st_, fetch_fork_, 0));
  G2OHIP_HIP_CHECK(hipMemcpyAsync(cams, ba_.cams.p, (size_t)ba_.n_cams * 12 * sizeof(double), hipMemcpyDeviceToHost, fetch_st_));
  G2OHIP_HIP_CHECK(hipEventRecord(fetch_ev_[0], fetch_st_));
  const size_t np = (size_t)ba_.n_points, step = (np + point_pieces - 1) / point_pieces;
  for (int k = 0; k < point_pieces; ++k) {
    const size_t b = std::min(np, k * step), e = std::min(np, b + step);
    if (e > b) G2OHIP_HIP_CHECK(hipMemcpyAsync(points + 3 * b, ba_.pts.p + 3 * b, (e - b) * 3 * sizeof(double), hipMemcpyDeviceToHost, fetch_st_));
    G2OHIP_HIP_CHECK(hipEventRecord(fetch_ev_[1 + k], fetch_st_));
  }
  fetch_pieces_ = 1 + point_pieces;
}
void BlockSolver::ba_fetch_wait(int piece) {
  if (fetch_pieces_ <= 0) throw StateFailure("ba_fetch_wait without ba_fetch_begin");
  if (piece < 0 || piece >= fetch_pieces_) throw ArgFailure("ba_fetch_wait: no such piece");
  G2OHIP_HIP_CHECK(hipEventSynchronize(fetch_ev_[piece]));
}
// (the solver's stream is about to overwrite the estimates: a read-back in flight has to see the old ones)
void BlockSolver::ba_fetch_fence() {
  if (fetch_pieces_ > 0) G2OHIP_HIP_CHECK(hipStreamWaitEvent(st_, fetch_ev_[fetch_pieces_ - 1], 0));
}

void BlockSolver::ba_linearize(bool jacobians) {
  if (ba_.set < 0 || ba_.n_cams <= 0) throw StateFailure("ba_linearize: call ba_set_edges and ba_set_estimates first");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  EdgeSet& es = *sets_[ba_.set];
  const bool fused = ba_fused && ba_.fused_ok;
  // fused mode: build_system re-evaluates the Jacobians itself, only the errors (for chi2) are produced here
  const bool need_jac = jacobians && !fused;
  if (ba_.err_valid && (!need_jac || ba_.jac_valid)) {   // the estimates have not moved since the last evaluation
    if (jacobians) es.has_data = true;
    return;
  }
  if (need_jac && (es.own_J0.n < (size_t)es.n * 6 || es.own_J1.n < (size_t)es.n * 12)) {   // (left out by ba_set_edges on the fused path)
    es.own_J0.alloc((size_t)es.n * 6);
    es.own_J1.alloc((size_t)es.n * 12);
    es.J0 = es.own_J0.p;
    es.J1 = es.own_J1.p;
    invalidate_graphs();   // (clears err_valid / jac_valid: before they are set for the evaluation below)
  }
  ba_.err_valid = true;
  if (need_jac) ba_.jac_valid = true;
  chi2_valid_ = false;
  if (profiling) tfe_.start(st_);
  // the chi2 of these errors rides along: 1 024 partial sums in this set's slot of the trial read-back buffer (chi2() and
  // trial_stats() take them from there while err_valid holds)
  constexpr int kMaxBlocks = 1024;
  const int lgrid = grid_for(es.n);
  if (d_red_multi.n < (sets_.size() + 1) * kMaxBlocks) d_red_multi.alloc((sets_.size() + 1) * kMaxBlocks);
  if (ba_.chi_part.n < (size_t)lgrid) ba_.chi_part.alloc(lgrid);
  hipLaunchKernelGGL(ba_linearize_kernel, dim3(lgrid), dim3(kThreads), 0, st_, es.n, ba_.cams.p, ba_.pts.p, ba_.cam_v.p,
                     ba_.pt_v.p, ba_.meas.p, ba_.f, ba_.cx, ba_.cy, es.own_J0.p, es.own_J1.p, es.own_err.p, (jacobians && !fused) ? 1 : 0,
                     es.omega, ba_.omega_identity ? 1 : 0, es.kernel_kind, es.delta, ba_.chi_part.p,
                     ba_.n_classes > 1 ? ba_.ctab.p : (const double*)nullptr);
  hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(1024), 0, st_, ba_.chi_part.p, lgrid, d_red_multi.p + (size_t)(ba_.set + 1) * kMaxBlocks);
  if (profiling) {
    tfe_.stop(st_);
    (jacobians ? times.linearize : times.residuals) = tfe_.seconds();
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
  es.has_err = true;
  if (jacobians) es.has_data = true;
}

void BlockSolver::ba_update() {
  require_structure();
  if (ba_.n_cams <= 0) throw StateFailure("ba_update before ba_set_estimates");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  ba_.err_valid = ba_.jac_valid = false;
  ++ba_.est_version;
  ba_fetch_fence();
  if (profiling) tfe_.start(st_);
  hipLaunchKernelGGL(ba_update_cams_kernel, dim3(grid_for(ba_.n_cams)), dim3(kThreads), 0, st_, ba_.n_cams, ba_.cams.p, ba_.cam_hidx.p,
                     d_x.p);
  hipLaunchKernelGGL(ba_update_pts_kernel, dim3(grid_for((size_t)ba_.n_points * 3)), dim3(kThreads), 0, st_, ba_.n_points, ba_.pts.p,
                     ba_.pt_hidx.p, d_x.p + (size_t)nP_ * p_);
  if (profiling) {
    tfe_.stop(st_);
    times.update = tfe_.seconds();
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
}

// estimate stack of depth one: what the LM trial loop needs (push / pop / discardTop,
// optimization_algorithm_levenberg.cpp:96,135,139; base_vertex.h:96-99)
void BlockSolver::ba_push() {
  if (ba_.n_cams <= 0) throw StateFailure("ba_push before ba_set_estimates");
  if (ba_.has_backup) throw StateFailure("ba_push: the estimate stack holds one level");
  G2OHIP_HIP_CHECK(hipMemcpyAsync(ba_.cams_bak.p, ba_.cams.p, (size_t)ba_.n_cams * 12 * sizeof(double), hipMemcpyDeviceToDevice, st_));
  G2OHIP_HIP_CHECK(hipMemcpyAsync(ba_.pts_bak.p, ba_.pts.p, (size_t)ba_.n_points * 3 * sizeof(double), hipMemcpyDeviceToDevice, st_));
  ba_.has_backup = true;
  ba_.bak_version = ba_.est_version;
}
void BlockSolver::ba_pop() {
  if (!ba_.has_backup) throw StateFailure("ba_pop without push");
  ba_.err_valid = ba_.jac_valid = false;
  ba_fetch_fence();
  G2OHIP_HIP_CHECK(hipMemcpyAsync(ba_.cams.p, ba_.cams_bak.p, (size_t)ba_.n_cams * 12 * sizeof(double), hipMemcpyDeviceToDevice, st_));
  G2OHIP_HIP_CHECK(hipMemcpyAsync(ba_.pts.p, ba_.pts_bak.p, (size_t)ba_.n_points * 3 * sizeof(double), hipMemcpyDeviceToDevice, st_));
  ba_.has_backup = false;
  ba_.est_version = ba_.bak_version;
}
void BlockSolver::ba_discard_top() {
  if (!ba_.has_backup) throw StateFailure("ba_discard_top without push");
  ba_.has_backup = false;
}


// ---- pose-graph front end (EdgeSE2 / EdgeSE3) ---------------------------------------------------------
void BlockSolver::pg_set_edges(int set, int type, const int* vi, const int* vj, const double* meas, const double* info) {
  invalidate_graphs();
  require_structure();
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("bad edge set id");
  if (type != 1 && type != 2) throw ArgFailure("pg_set_edges: type must be 1 (EdgeSE2) or 2 (EdgeSE3)");
  EdgeSet& es = *sets_[set];
  const int d = type == 1 ? 3 : 6;
  if (es.unary || es.d != d || es.dim0 != d || es.dim1 != d) throw ArgFailure("pg_set_edges: the set must be a binary pose-pose set of matching dimension");
  if (!vi || !vj || !meas || !info) throw ArgFailure("pg_set_edges: null array");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  const size_t n = (size_t)es.n, ms = type == 1 ? 3 : 12;
  pg_.set = set;
  pg_.type = type;
  pg_.h_vi.assign(vi, vi + n);
  pg_.h_vj.assign(vj, vj + n);
  pg_.set = set;
  pg_validate();
  pg_.vi.upload(vi, n, st_);
  pg_.vj.upload(vj, n, st_);
  pg_.meas.upload(meas, n * ms, st_);
  es.own_omega.upload(info, n * d * d, st_);
  es.own_J0.alloc(n * d * d);
  es.own_J1.alloc(n * d * d);
  es.own_err.alloc(n * d);
  es.J0 = es.own_J0.p; es.J1 = es.own_J1.p; es.omega = es.own_omega.p; es.err = es.own_err.p;
  es.has_data = false;
  es.has_err = false;
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
}

void BlockSolver::pg_set_estimates(int nv, const double* poses, const int* hidx) {
  if (pg_.type == 0) throw StateFailure("pg_set_estimates: call pg_set_edges first");
  if (nv <= 0 || !poses || !hidx) throw ArgFailure("pg_set_estimates: bad arguments");
  pg_.err_valid = pg_.jac_valid = false;
  chi2_valid_ = false;
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  const size_t ps = pg_.type == 1 ? 3 : 12;
  if (nv == pg_.nv && pg_.poses.p && (int)pg_.h_hidx.size() == nv && std::memcmp(pg_.h_hidx.data(), hidx, sizeof(int) * (size_t)nv) == 0) {
    pg_.has_backup = false;   // (same tables, new values: see ba_set_estimates)
    pg_.poses.upload(poses, (size_t)nv * ps, st_);
    G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
    return;
  }
  pg_.nv = nv;
  pg_.h_hidx.assign(hidx, hidx + nv);
  pg_validate();
  pg_.poses.upload(poses, (size_t)nv * ps, st_);
  pg_.hidx.upload(hidx, (size_t)nv, st_);
  pg_.poses_bak.alloc((size_t)nv * ps);
  pg_.has_backup = false;
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
}

void BlockSolver::pg_get_estimates(double* poses) {
  if (pg_.nv <= 0) throw StateFailure("pg_get_estimates before pg_set_estimates");
  pg_.poses.download(poses, (size_t)pg_.nv * (pg_.type == 1 ? 3 : 12), st_);
}

void BlockSolver::pg_linearize(bool jacobians) {
  if (pg_.set < 0 || pg_.nv <= 0) throw StateFailure("pg_linearize: call pg_set_edges and pg_set_estimates first");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  EdgeSet& es = *sets_[pg_.set];
  if ((int)pg_.h_vi.size() != es.n) throw StateFailure("pg_linearize: the edge set has grown since pg_set_edges (g2ohip_update_structure): call pg_set_edges again");
  if (pg_.err_valid && (!jacobians || pg_.jac_valid)) {   // the estimates have not moved since the last evaluation
    if (jacobians) es.has_data = true;
    return;
  }
  pg_.err_valid = true;
  if (jacobians) pg_.jac_valid = true;
  chi2_valid_ = false;
  if (pg_.type == 1)
    hipLaunchKernelGGL(pg_se2_linearize_kernel, dim3(grid_for(es.n)), dim3(kThreads), 0, st_, es.n, pg_.poses.p, pg_.vi.p, pg_.vj.p,
                       pg_.meas.p, es.own_J0.p, es.own_J1.p, es.own_err.p, jacobians ? 1 : 0);
  else
    hipLaunchKernelGGL(pg_se3_linearize_kernel, dim3(grid_for(es.n)), dim3(kThreads), 0, st_, es.n, pg_.poses.p, pg_.vi.p, pg_.vj.p,
                       pg_.meas.p, es.own_J0.p, es.own_J1.p, es.own_err.p, jacobians ? 1 : 0);
  G2OHIP_HIP_CHECK(hipGetLastError());
  es.has_err = true;
  if (jacobians) es.has_data = true;
}

void BlockSolver::pg_update() {
  require_structure();
  if (pg_.nv <= 0) throw StateFailure("pg_update before pg_set_estimates");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  pg_.err_valid = pg_.jac_valid = false;
  if (pg_.type == 1)
    hipLaunchKernelGGL(pg_se2_update_kernel, dim3(grid_for(pg_.nv)), dim3(kThreads), 0, st_, pg_.nv, pg_.poses.p, pg_.hidx.p, d_x.p);
  else
    hipLaunchKernelGGL(pg_se3_update_kernel, dim3(grid_for(pg_.nv)), dim3(kThreads), 0, st_, pg_.nv, pg_.poses.p, pg_.hidx.p, d_x.p);
  G2OHIP_HIP_CHECK(hipGetLastError());
}

void BlockSolver::pg_push() {
  if (pg_.nv <= 0) throw StateFailure("pg_push before pg_set_estimates");
  if (pg_.has_backup) throw StateFailure("pg_push: the estimate stack holds one level");
  G2OHIP_HIP_CHECK(hipMemcpyAsync(pg_.poses_bak.p, pg_.poses.p, (size_t)pg_.nv * (pg_.type == 1 ? 3 : 12) * sizeof(double),
                                  hipMemcpyDeviceToDevice, st_));
  pg_.has_backup = true;
}
void BlockSolver::pg_pop() {
  if (!pg_.has_backup) throw StateFailure("pg_pop without push");
  pg_.err_valid = pg_.jac_valid = false;
  G2OHIP_HIP_CHECK(hipMemcpyAsync(pg_.poses.p, pg_.poses_bak.p, (size_t)pg_.nv * (pg_.type == 1 ? 3 : 12) * sizeof(double),
                                  hipMemcpyDeviceToDevice, st_));
  pg_.has_backup = false;
}
void BlockSolver::pg_discard_top() {
  if (!pg_.has_backup) throw StateFailure("pg_discard_top without push");
  pg_.has_backup = false;
}

// Blocks (rows[i], cols[i]) of the inverse of the system the linear solver factorises (Hpp without Schur, the reduced
// pose system with it: the pose marginals with the landmarks integrated out).  One factorisation, then a pair of
// triangular sweeps per requested scalar column.  Replaces BlockSolver::computeMarginals -> LinearSolver::solvePattern
// (block_solver.hpp:489-498, linear_solver.h:63-69, marginal_covariance_cholesky.cpp:71-220 computes the same entries
// by recursion on the factor).
__global__ void set_unit_kernel(double* __restrict__ v, size_t n, size_t k) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) v[i] = (i == k) ? 1.0 : 0.0;
}
// Hpp (+ lambda on the diagonal of the diagonal blocks) laid out on the pattern of Hschur, which the factorisation was
// analysed for (a superset of Hpp's pattern): the matrix BlockSolver::computeMarginals hands to solvePattern
__global__ void hpp_on_schur_pattern_kernel(size_t n, int bb, int pd, const int* __restrict__ hs_src, const int* __restrict__ hs_diag,
                                            const double* __restrict__ Hpp, const double* __restrict__ lam, double* __restrict__ Hs) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int d = (int)(t / bb), e = (int)(t % bb);
  const int src = hs_src[d];
  double v = src >= 0 ? Hpp[(size_t)src * bb + e] : 0.0;
  if (hs_diag[d] >= 0 && e % (pd + 1) == 0) v += lam[0];
  Hs[t] = v;
}

// block i of the output <- block at off[i] of the sparse-inverse slab (leading dimension ld[i], transposed if tr[i])
__global__ void gather_inverse_blocks_kernel(int n, int p, const long long* __restrict__ off, const int* __restrict__ ld,
                                             const int* __restrict__ tr, const double* __restrict__ Z, double* __restrict__ out) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= (size_t)n * p * p) return;
  const int b = (int)(t / (p * p)), e = (int)(t % (p * p)), i = e % p, j = e / p;
  if (off[b] < 0) return;
  out[t] = tr[b] ? Z[off[b] + j + (long long)ld[b] * i] : Z[off[b] + i + (long long)ld[b] * j];
}

int BlockSolver::compute_marginals(int n, const int* rows, const int* cols, double* out) {
  require_structure();
  if (!system_built_) throw StateFailure("compute_marginals before build_system");
  if (n < 0 || (n > 0 && (!rows || !cols || !out))) throw ArgFailure("compute_marginals: bad arguments");
  for (int i = 0; i < n; ++i)
    if (rows[i] < 0 || rows[i] >= nP_ || cols[i] < 0 || cols[i] >= nP_) throw ArgFailure("compute_marginals: block index out of range");
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  // Reference behaviour (block_solver.hpp:489-493): solvePattern(spinv, blockIndices, *_Hpp) -- the inverse of Hpp
  // ALONE, also when the Schur complement is on.  Option "marginals_reduced" = 1 inverts the reduced pose system
  // instead (the pose marginals with the landmarks integrated out).
  if (schur_ && marginals_reduced) {
    solve_schur_impl(true);   // (re)forms the reduced system with the current damping
  } else if (schur_) {
    const size_t ne = hs_row.size() * (size_t)p_ * p_;
    hipLaunchKernelGGL(hpp_on_schur_pattern_kernel, dim3(grid_for(ne)), dim3(kThreads), 0, st_, ne, p_ * p_, p_, d_hs_src.p,
                       d_hs_diag.p, d_Hpp.p, d_lam.p, d_Hschur.p);
    hschur_valid_ = false;    // d_Hschur no longer holds the reduced system
  }
  const double* H = schur_ ? d_Hschur.p : d_Hpp.p;
  chol_->factor(H, st_);
  if (chol_->failed(st_)) {
    // a dependency-driven launch that gave up waiting is not "not positive definite": the factorisation has switched to
    // one launch per level -- once more (as g2ohip_solve does)
    if (!(chol_->dependency_stall() && ++dependency_fallbacks)) return 1;
    invalidate_graphs();
    chol_->factor(H, st_);
    if (chol_->failed(st_)) return 1;
  }
  // Blocks inside the pattern of the factor: one top-down pass over the frontal matrices gives ALL of them (sparse
  // inverse); what lies outside the pattern (fill-free pairs of distant poses) falls back to a pair of triangular
  // sweeps per requested column.
  std::vector<char> done(n, 0);
  if (marginals_recursion && chol_opt.world == 1 && n > 0) {
    chol_->sparse_inverse(st_);
    std::vector<long long> off(n, -1);
    std::vector<int> ldv(n, 0), trv(n, 0);
    int found = 0;
    for (int i = 0; i < n; ++i) {
      bool tr = false;
      if (chol_->inverse_block(rows[i], cols[i], &off[i], &ldv[i], &tr)) {
        trv[i] = tr ? 1 : 0;
        done[i] = 1;
        ++found;
      } else {
        off[i] = -1;
      }
    }
    if (found > 0) {
      DevBuf<long long> d_off;
      DevBuf<int> d_ld, d_tr;
      DevBuf<double> d_out;
      d_off.upload(off, st_);
      d_ld.upload(ldv, st_);
      d_tr.upload(trv, st_);
      d_out.alloc((size_t)n * p_ * p_);
      hipLaunchKernelGGL(gather_inverse_blocks_kernel, dim3(grid_for((size_t)n * p_ * p_)), dim3(kThreads), 0, st_, n, p_, d_off.p, d_ld.p,
                         d_tr.p, chol_->inverse_slab(), d_out.p);
      std::vector<double> ho((size_t)n * p_ * p_);
      d_out.download(ho.data(), ho.size(), st_);
      for (int i = 0; i < n; ++i)
        if (done[i]) std::copy(ho.begin() + (size_t)i * p_ * p_, ho.begin() + (size_t)(i + 1) * p_ * p_, out + (size_t)i * p_ * p_);
    }
    if (found == n) return 0;
  }
  const size_t np = (size_t)nP_ * p_;
  DevBuf<double> rhs, sol;
  rhs.alloc(np);
  sol.alloc(np);
  std::vector<double> h(np);
  std::vector<int> order;
  for (int i = 0; i < n; ++i)
    if (!done[i]) order.push_back(i);
  n = (int)order.size();
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cols[a] < cols[b]; });
  for (int i = 0; i < n;) {
    const int c = cols[order[i]];
    int i1 = i;
    while (i1 < n && cols[order[i1]] == c) ++i1;
    for (int k = 0; k < p_; ++k) {
      hipLaunchKernelGGL(set_unit_kernel, dim3(grid_for(np)), dim3(kThreads), 0, st_, rhs.p, np, (size_t)c * p_ + k);
      chol_->solve(rhs.p, sol.p, st_);
      sol.download(h.data(), np, st_);
      for (int j = i; j < i1; ++j) {
        const int r = rows[order[j]];
        double* blk = out + (size_t)order[j] * p_ * p_;
        for (int rr = 0; rr < p_; ++rr) blk[rr + p_ * k] = h[(size_t)r * p_ + rr];
      }
    }
    i = i1;
  }
  return 0;
}

void BlockSolver::copy_edge_data(int set, double* J0, double* J1, double* err) {
  if (set < 0 || set >= (int)sets_.size()) throw ArgFailure("bad edge set id");
  EdgeSet& es = *sets_[set];
  if (!es.has_err) throw StateFailure("copy_edge_data: the set has no edge data yet");
  if (set == ba_.set && (!ll_valid_ || ll_hbm_partial_) && !ba_.err_valid) ensure_ll();
  G2OHIP_HIP_CHECK(hipSetDevice(device_));
  const size_t n = (size_t)es.n;
  auto pull = [&](double* dst, const double* src, size_t cnt) {
    if (!dst || !src || cnt == 0) return;
    G2OHIP_HIP_CHECK(hipMemcpyAsync(dst, src, cnt * sizeof(double), hipMemcpyDeviceToHost, st_));
  };
  if (es.has_data && !(set == ba_.set && !ba_.jac_valid)) {
    pull(J0, es.J0, n * es.d * es.dim0);
    if (!es.unary) pull(J1, es.J1, n * es.d * es.dim1);
  } else if (J0 || J1) {
    throw StateFailure("copy_edge_data: Jacobians were not produced (fused assembly or error-only linearize)");
  }
  pull(err, es.err, n * es.d);
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st_));
}

void BlockSolver::device_array(int which, double** ptr, size_t* count) {
  require_structure();
  switch (which) {
    case 0: *ptr = d_Hpp.p; *count = pp_row.size() * p_ * p_; break;
    case 1: ensure_hpl(); *ptr = d_Hpl.p; *count = pl_row.size() * p_ * l_; break;
    case 2: ensure_ll(); *ptr = d_Hll.p; *count = (size_t)nL_ * l_ * l_; break;
    case 3: ensure_hschur(); *ptr = d_Hschur.p; *count = hs_row.size() * p_ * p_; break;
    case 100: *ptr = d_bschur.p; *count = (size_t)nP_ * p_; break;
    case 101: *ptr = d_x.p; *count = vector_size(); break;
    case 102: *ptr = d_b.p; *count = vector_size(); break;
    case 103: *ptr = chol_->exchange_buffer(count); break;      // subtree-root update matrices + vectors
    case 104: *ptr = chol_->permuted_solution(count); break;   // x_p in elimination order (masked before the all-reduce)
    case 105: *ptr = ex_.merged ? ex_.tail : ex_.buf1.p; *count = (size_t)std::max(1, ex_.nbb * p_ * p_ + ex_.nbp * p_); break;   // exchange_setup buffers (what exchange_pack(1) fills)
    case 106: *ptr = ex_.buf3.p; *count = (size_t)ex_.nh * p_ + 1; break;
    case 107: *ptr = d_mf_diag.p; *count = mf_ready_ ? (size_t)nP_ * p_ * p_ : 0; break;   // schur_operator_prepare: diagonal blocks
    default: throw ArgFailure("bad array selector");
  }
}

}  // namespace g2ohip
