// Block-Jacobi preconditioned conjugate gradients on the device (see block_pcg.h).
#include "block_pcg.h"

#include <cmath>
#include <vector>

namespace g2ohip {
namespace {

constexpr int kT = 256;
// d_scal layout
enum { S_DN = 0, S_DQ, S_D0, S_BA, S_DONE, S_ITERS, S_BAD, S_COUNT };

// inverse of a symmetric positive definite BS x BS block through its Cholesky factor (static register
// indexing; the reference calls Eigen's inverse() on the diagonal blocks, linear_solver_pcg.hpp:96).
// Returns false when a pivot is not positive.
template <int BS>
__device__ __forceinline__ bool spd_inverse(const double* A /* column-major */, double* J) {
  double L[BS][BS];
#pragma unroll
  for (int c = 0; c < BS; ++c)
#pragma unroll
    for (int r = 0; r < BS; ++r) L[r][c] = (r >= c) ? A[r + BS * c] : 0.0;
  bool ok = true;
#pragma unroll
  for (int c = 0; c < BS; ++c) {
    double d = L[c][c];
    if (!(d > 0.0)) {
      ok = false;
      d = 1.0;
    }
    const double s = sqrt(d), inv = 1.0 / s;
    L[c][c] = s;
#pragma unroll
    for (int i = c + 1; i < BS; ++i) L[i][c] *= inv;
#pragma unroll
    for (int j = c + 1; j < BS; ++j)
#pragma unroll
      for (int i = j; i < BS; ++i) L[i][j] -= L[i][c] * L[j][c];
  }
  // M = L^-1 (lower triangular), column by column
  double M[BS][BS];
#pragma unroll
  for (int c = 0; c < BS; ++c) {
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      if (r < c) {
        M[r][c] = 0.0;
      } else if (r == c) {
        M[r][c] = 1.0 / L[c][c];
      } else {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < BS; ++k)
          if (k >= c && k < r) s += L[r][k] * M[k][c];
        M[r][c] = -s / L[r][r];
      }
    }
  }
  // J = M' M
#pragma unroll
  for (int c = 0; c < BS; ++c)
#pragma unroll
    for (int r = 0; r < BS; ++r) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < BS; ++k)
        if (k >= r && k >= c) s += M[k][r] * M[k][c];
      J[r + BS * c] = s;
    }
  return ok;
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  // fixed-order tree over the 256 threads of the workgroup (deterministic)
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int off = kT / 2; off > 0; off >>= 1) {
    if (tid < off) sh[tid] += sh[tid + off];
    __syncthreads();
  }
  return sh[0];
}

template <int BS>
__global__ void __launch_bounds__(kT) pcg_init_kernel(int nb, const double* __restrict__ A, const int* __restrict__ diag,
                                                     const double* __restrict__ b, double* __restrict__ J, double* __restrict__ r,
                                                     double* __restrict__ d, double* __restrict__ x, double* __restrict__ part,
                                                     double* __restrict__ scal) {
  __shared__ double sh[kT];
  const int i = blockIdx.x * kT + threadIdx.x;
  double dot = 0.0;
  if (i < nb) {
    double Ji[BS * BS];
    const bool ok = spd_inverse<BS>(A + (size_t)diag[i] * BS * BS, Ji);
    if (!ok) scal[S_BAD] = 1.0;
#pragma unroll
    for (int k = 0; k < BS * BS; ++k) J[(size_t)i * BS * BS + k] = Ji[k];
    double ri[BS];
#pragma unroll
    for (int k = 0; k < BS; ++k) {
      ri[k] = b[(size_t)i * BS + k];
      r[(size_t)i * BS + k] = ri[k];
      x[(size_t)i * BS + k] = 0.0;
    }
#pragma unroll
    for (int rr = 0; rr < BS; ++rr) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < BS; ++c) t += Ji[rr + BS * c] * ri[c];
      d[(size_t)i * BS + rr] = t;
      dot += ri[rr] * t;
    }
  }
  const double s = block_sum(dot, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// which: 0 = after init (dn, d0), 1 = d'q, 2 = new r'Jr (beta, iteration count, stopping test)
__global__ void __launch_bounds__(kT) pcg_reduce_kernel(int n, const double* __restrict__ part, double* __restrict__ scal, int which,
                                                       double tol, int absolute, double prev_residual, int max_iter) {
  __shared__ double sh[kT];
  if (which != 0 && scal[S_DONE] != 0.0) return;
  double v = 0.0;
  for (int k = threadIdx.x; k < n; k += kT) v += part[k];
  const double s = block_sum(v, sh);
  if (threadIdx.x != 0) return;
  if (which == 0) {
    double d0 = tol * s;
    if (absolute && prev_residual > 0.0 && prev_residual > d0) d0 = prev_residual;
    scal[S_DN] = s;
    scal[S_D0] = d0;
    scal[S_ITERS] = 0.0;
    scal[S_DONE] = (!(s == s) || s <= d0 || max_iter <= 0) ? 1.0 : 0.0;
    if (!(s == s)) scal[S_BAD] = 1.0;
  } else if (which == 1) {
    scal[S_DQ] = s;
  } else {
    const double dold = scal[S_DN];
    scal[S_BA] = s / dold;
    scal[S_DN] = s;
    const double it = scal[S_ITERS] + 1.0;
    scal[S_ITERS] = it;
    if (!(s == s)) {
      scal[S_BAD] = 1.0;
      scal[S_DONE] = 1.0;
    } else if (s <= scal[S_D0] || it >= (double)max_iter) {
      scal[S_DONE] = 1.0;
    }
  }
}

// q = A d (symmetric, upper blocks stored): gather over the entry list of every block row, then d'q
template <int BS>
__global__ void __launch_bounds__(kT) pcg_spmv_kernel(int nb, const double* __restrict__ A, const int* __restrict__ diag,
                                                     const int* __restrict__ ent_ptr, const int* __restrict__ ent,
                                                     const int* __restrict__ other, const double* __restrict__ d,
                                                     double* __restrict__ q, double* __restrict__ part, const double* __restrict__ scal) {
  __shared__ double sh[kT];
  if (scal[S_DONE] != 0.0) return;
  const int i = blockIdx.x * kT + threadIdx.x;
  double dot = 0.0;
  if (i < nb) {
    double y[BS], xi[BS];
#pragma unroll
    for (int k = 0; k < BS; ++k) {
      xi[k] = d[(size_t)i * BS + k];
      y[k] = 0.0;
    }
    {
      const double* D = A + (size_t)diag[i] * BS * BS;   // multDiag (:181-188): the full diagonal block
#pragma unroll
      for (int c = 0; c < BS; ++c)
#pragma unroll
        for (int rr = 0; rr < BS; ++rr) y[rr] += D[rr + BS * c] * xi[c];
    }
    for (int e = ent_ptr[i]; e < ent_ptr[i + 1]; ++e) {
      const int pk = ent[e];
      const double* B = A + (size_t)(pk >> 1) * BS * BS;
      const double* xo = d + (size_t)other[e] * BS;
      double xv[BS];
#pragma unroll
      for (int k = 0; k < BS; ++k) xv[k] = xo[k];
      if (pk & 1) {   // y_i += B' x_other
#pragma unroll
        for (int c = 0; c < BS; ++c)
#pragma unroll
          for (int rr = 0; rr < BS; ++rr) y[c] += B[rr + BS * c] * xv[rr];
      } else {        // y_i += B x_other
#pragma unroll
        for (int c = 0; c < BS; ++c)
#pragma unroll
          for (int rr = 0; rr < BS; ++rr) y[rr] += B[rr + BS * c] * xv[c];
      }
    }
#pragma unroll
    for (int k = 0; k < BS; ++k) {
      q[(size_t)i * BS + k] = y[k];
      dot += xi[k] * y[k];
    }
  }
  const double s = block_sum(dot, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// x += a d; r -= a q; s = J r; partial r's
template <int BS>
__global__ void __launch_bounds__(kT) pcg_update_kernel(int nb, const double* __restrict__ J, double* __restrict__ x, double* __restrict__ r,
                                                       const double* __restrict__ d, const double* __restrict__ q, double* __restrict__ s,
                                                       double* __restrict__ part, const double* __restrict__ scal) {
  __shared__ double sh[kT];
  if (scal[S_DONE] != 0.0) return;
  const double a = scal[S_DN] / scal[S_DQ];
  const int i = blockIdx.x * kT + threadIdx.x;
  double dot = 0.0;
  if (i < nb) {
    double ri[BS];
#pragma unroll
    for (int k = 0; k < BS; ++k) {
      const size_t t = (size_t)i * BS + k;
      x[t] += a * d[t];
      ri[k] = r[t] - a * q[t];
      r[t] = ri[k];
    }
    const double* Ji = J + (size_t)i * BS * BS;
#pragma unroll
    for (int rr = 0; rr < BS; ++rr) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < BS; ++c) t += Ji[rr + BS * c] * ri[c];
      s[(size_t)i * BS + rr] = t;
      dot += ri[rr] * t;
    }
  }
  const double v = block_sum(dot, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = v;
}

__global__ void __launch_bounds__(kT) pcg_direction_kernel(size_t n, const double* __restrict__ s, double* __restrict__ d,
                                                          const double* __restrict__ scal) {
  if (scal[S_DONE] != 0.0) return;
  const size_t t = blockIdx.x * (size_t)kT + threadIdx.x;
  if (t < n) d[t] = s[t] + scal[S_BA] * d[t];
}

}  // namespace

void BlockPCG::analyze(int nb, const int* colptr, const int* rowidx, hipStream_t st) {
  if (nb <= 0) throw ArgFailure("BlockPCG: empty matrix");
  if (bs_ != 3 && bs_ != 6 && bs_ != 7) throw ArgFailure("BlockPCG: unsupported block size (3, 6, 7)");
  std::vector<int> diag(nb, -1), cnt(nb + 1, 0);
  for (int c = 0; c < nb; ++c)
    for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
      const int r = rowidx[q];
      if (r == c) {
        diag[c] = q;
      } else {
        cnt[r + 1]++;
        cnt[c + 1]++;
      }
    }
  for (int c = 0; c < nb; ++c)
    if (diag[c] < 0) throw ArgFailure("BlockPCG: a diagonal block is missing from the pattern");
  for (int i = 0; i < nb; ++i) cnt[i + 1] += cnt[i];
  std::vector<int> ent(cnt[nb]), other(cnt[nb]), w(cnt.begin(), cnt.end() - 1);
  for (int c = 0; c < nb; ++c)
    for (int q = colptr[c]; q < colptr[c + 1]; ++q) {
      const int r = rowidx[q];
      if (r == c) continue;
      ent[w[r]] = q << 1;          // y_r += A_q d_c
      other[w[r]++] = c;
      ent[w[c]] = (q << 1) | 1;    // y_c += A_q' d_r
      other[w[c]++] = r;
    }
  d_diag.upload(diag, st);
  d_ent_ptr.upload(cnt, st);
  if (ent.empty()) {
    ent.push_back(0);
    other.push_back(0);
  }
  d_ent.upload(ent, st);
  d_ent_other.upload(other, st);
  const size_t n = (size_t)nb * bs_;
  d_J.alloc((size_t)nb * bs_ * bs_);
  d_r.alloc(n);
  d_d.alloc(n);
  d_q.alloc(n);
  d_s.alloc(n);
  n_part_ = (nb + kT - 1) / kT;
  d_part.alloc(n_part_);
  d_scal.alloc(S_COUNT);
  nb_ = nb;
  residual_ = -1.0;
  G2OHIP_HIP_CHECK(hipStreamSynchronize(st));
}

// partial sums of d'q per block of kT block rows (what pcg_spmv_kernel leaves in `part`)
template <int BS>
__global__ void __launch_bounds__(kT) pcg_dot_kernel(int nb, const double* __restrict__ d, const double* __restrict__ q,
                                                    double* __restrict__ part, const double* __restrict__ scal) {
  __shared__ double sh[kT];
  if (scal[S_DONE] != 0.0) return;
  const int i = blockIdx.x * kT + threadIdx.x;
  double dot = 0.0;
  if (i < nb) {
#pragma unroll
    for (int k = 0; k < BS; ++k) dot += d[(size_t)i * BS + k] * q[(size_t)i * BS + k];
  }
  const double s = block_sum(dot, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

void BlockPCG::analyze_operator(int nb, hipStream_t st) {
  if (bs_ != 3 && bs_ != 6 && bs_ != 7) throw ArgFailure("BlockPCG: unsupported block size (3, 6, 7)");
  nb_ = nb;
  std::vector<int> diag(std::max(nb, 1));
  for (int i = 0; i < nb; ++i) diag[i] = i;   // diagonal block i of the array handed to solve_operator
  d_diag.upload(diag, st);
  const size_t n = (size_t)nb * bs_;
  d_J.alloc((size_t)nb * bs_ * bs_);
  d_r.alloc(n);
  d_d.alloc(n);
  d_q.alloc(n);
  d_s.alloc(n);
  n_part_ = (nb + kT - 1) / kT;
  d_part.alloc(n_part_);
  d_scal.alloc(S_COUNT);
  residual_ = -1.0;
}

void BlockPCG::multiply(const double* dA, const double* d_in, double* d_out, hipStream_t st) {
  if (nb_ <= 0 || !d_ent_ptr.p) throw StateFailure("BlockPCG::multiply before analyze");
  if (!d_zero_scal.p) {
    d_zero_scal.alloc(S_COUNT);
    d_zero_scal.zero(st);
  }
  switch (bs_) {
#define G2OHIP_MUL(BS_)                                                                                                          \
  case BS_:                                                                                                                      \
    hipLaunchKernelGGL((pcg_spmv_kernel<BS_>), dim3(n_part_), dim3(kT), 0, st, nb_, dA, d_diag.p, d_ent_ptr.p, d_ent.p, d_ent_other.p, \
                       d_in, d_out, d_part.p, d_zero_scal.p);                                                                    \
    break
    G2OHIP_MUL(3);
    G2OHIP_MUL(6);
    G2OHIP_MUL(7);
#undef G2OHIP_MUL
    default: throw ArgFailure("BlockPCG: unsupported block size (3, 6, 7)");
  }
  G2OHIP_HIP_CHECK(hipGetLastError());
}

bool BlockPCG::solve_operator(const double* d_diag_blocks, const std::function<void(const double*, double*)>& apply,
                              const double* d_b, double* d_x, hipStream_t st) {
  if (nb_ <= 0) throw StateFailure("BlockPCG::solve_operator before analyze_operator");
  const int nb = nb_, grid = n_part_;
  const size_t n = (size_t)nb * bs_;
  const int max_iter = opt.max_iter < 0 ? (int)n : opt.max_iter;
  G2OHIP_HIP_CHECK(hipMemsetAsync(d_scal.p, 0, S_COUNT * sizeof(double), st));
#define G2OHIP_PCGOP(BS_)                                                                                                        \
  {                                                                                                                              \
    hipLaunchKernelGGL((pcg_init_kernel<BS_>), dim3(grid), dim3(kT), 0, st, nb, d_diag_blocks, d_diag.p, d_b, d_J.p, d_r.p, d_d.p, \
                       d_x, d_part.p, d_scal.p);                                                                                 \
    hipLaunchKernelGGL(pcg_reduce_kernel, dim3(1), dim3(kT), 0, st, grid, d_part.p, d_scal.p, 0, opt.tolerance,                   \
                       opt.absolute_tolerance ? 1 : 0, residual_, max_iter);                                                     \
    double h[S_COUNT] = {0};                                                                                                     \
    int launched = 0;                                                                                                            \
    for (;;) {                                                                                                                   \
      const int chunk = std::max(1, std::min(opt.check_every, max_iter - launched));                                             \
      for (int k = 0; k < chunk; ++k) {                                                                                          \
        apply(d_d.p, d_q.p);                                                                                                     \
        hipLaunchKernelGGL((pcg_dot_kernel<BS_>), dim3(grid), dim3(kT), 0, st, nb, d_d.p, d_q.p, d_part.p, d_scal.p);             \
        hipLaunchKernelGGL(pcg_reduce_kernel, dim3(1), dim3(kT), 0, st, grid, d_part.p, d_scal.p, 1, 0.0, 0, 0.0, max_iter);     \
        hipLaunchKernelGGL((pcg_update_kernel<BS_>), dim3(grid), dim3(kT), 0, st, nb, d_J.p, d_x, d_r.p, d_d.p, d_q.p, d_s.p,      \
                           d_part.p, d_scal.p);                                                                                  \
        hipLaunchKernelGGL(pcg_reduce_kernel, dim3(1), dim3(kT), 0, st, grid, d_part.p, d_scal.p, 2, 0.0, 0, 0.0, max_iter);     \
        hipLaunchKernelGGL(pcg_direction_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, st, n, d_s.p, d_d.p, d_scal.p); \
      }                                                                                                                          \
      launched += chunk;                                                                                                         \
      G2OHIP_HIP_CHECK(hipGetLastError());                                                                                       \
      d_scal.download(h, S_COUNT, st);                                                                                           \
      if (h[S_DONE] != 0.0 || launched >= max_iter) break;                                                                       \
    }                                                                                                                            \
    iters_ = (int)h[S_ITERS];                                                                                                    \
    residual_ = 0.5 * h[S_DN];                                                                                                   \
    return h[S_BAD] == 0.0;                                                                                                      \
  }
  switch (bs_) {
    case 3: G2OHIP_PCGOP(3)
    case 6: G2OHIP_PCGOP(6)
    case 7: G2OHIP_PCGOP(7)
    default: throw ArgFailure("BlockPCG: unsupported block size (3, 6, 7)");
  }
#undef G2OHIP_PCGOP
}

bool BlockPCG::solve(const double* dA, const double* d_b, double* d_x, hipStream_t st) {
  if (nb_ <= 0) throw StateFailure("BlockPCG::solve before analyze");
  const int nb = nb_, grid = n_part_;
  const size_t n = (size_t)nb * bs_;
  const int max_iter = opt.max_iter < 0 ? (int)n : opt.max_iter;
  G2OHIP_HIP_CHECK(hipMemsetAsync(d_scal.p, 0, S_COUNT * sizeof(double), st));
#define G2OHIP_PCG(BS_)                                                                                                          \
  {                                                                                                                              \
    hipLaunchKernelGGL((pcg_init_kernel<BS_>), dim3(grid), dim3(kT), 0, st, nb, dA, d_diag.p, d_b, d_J.p, d_r.p, d_d.p, d_x,       \
                       d_part.p, d_scal.p);                                                                                      \
    hipLaunchKernelGGL(pcg_reduce_kernel, dim3(1), dim3(kT), 0, st, grid, d_part.p, d_scal.p, 0, opt.tolerance,                   \
                       opt.absolute_tolerance ? 1 : 0, residual_, max_iter);                                                     \
    double h[S_COUNT] = {0};                                                                                                     \
    int launched = 0;                                                                                                            \
    for (;;) {                                                                                                                   \
      const int chunk = std::max(1, std::min(opt.check_every, max_iter - launched));                                             \
      for (int k = 0; k < chunk; ++k) {                                                                                          \
        hipLaunchKernelGGL((pcg_spmv_kernel<BS_>), dim3(grid), dim3(kT), 0, st, nb, dA, d_diag.p, d_ent_ptr.p, d_ent.p,            \
                           d_ent_other.p, d_d.p, d_q.p, d_part.p, d_scal.p);                                                     \
        hipLaunchKernelGGL(pcg_reduce_kernel, dim3(1), dim3(kT), 0, st, grid, d_part.p, d_scal.p, 1, 0.0, 0, 0.0, max_iter);     \
        hipLaunchKernelGGL((pcg_update_kernel<BS_>), dim3(grid), dim3(kT), 0, st, nb, d_J.p, d_x, d_r.p, d_d.p, d_q.p, d_s.p,      \
                           d_part.p, d_scal.p);                                                                                  \
        hipLaunchKernelGGL(pcg_reduce_kernel, dim3(1), dim3(kT), 0, st, grid, d_part.p, d_scal.p, 2, 0.0, 0, 0.0, max_iter);     \
        hipLaunchKernelGGL(pcg_direction_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, st, n, d_s.p, d_d.p, d_scal.p); \
      }                                                                                                                          \
      launched += chunk;                                                                                                         \
      G2OHIP_HIP_CHECK(hipGetLastError());                                                                                       \
      d_scal.download(h, S_COUNT, st);                                                                                           \
      if (h[S_DONE] != 0.0 || launched >= max_iter) break;                                                                       \
    }                                                                                                                            \
    iters_ = (int)h[S_ITERS];                                                                                                    \
    residual_ = 0.5 * h[S_DN];                                                                                                   \
    return h[S_BAD] == 0.0;                                                                                                      \
  }
  switch (bs_) {
    case 3: G2OHIP_PCG(3)
    case 6: G2OHIP_PCG(6)
    case 7: G2OHIP_PCG(7)
    default: throw ArgFailure("BlockPCG: unsupported block size (3, 6, 7)");
  }
#undef G2OHIP_PCG
}

}  // namespace g2ohip
