// Probe: the per-landmark Schur outer product  S_l = (B_l Dinv_l) B_l'  (B_l = the K stacked 6 x 3 Hpl blocks of a
// landmark, block_solver.hpp:400-431) formulated for the vector ALU and for the matrix cores of gfx950.
//
//   valu : the destination-major form of schur_tile_dests stripped of its list handling -- one lane per (landmark, pose
//          pair): V = B_i Dinv (54 FMAs), acc += V B_j' (108 FMAs), operands from LDS;
//   mfma : one wave per landmark step -- the stacked B (6 K <= 32 rows, 3 -> 4 columns) as A / B operands of
//          v_mfma_f64_16x16x4_f64, three upper 16 x 16 tiles per landmark, accumulated in the result registers
//          (flush = 0: as if every landmark of the run had the same K poses -- the best case for the matrix cores)
//          or scattered to per-pair blocks in LDS with ds_add_f64 after every landmark (flush = 1: the general case,
//          pose sets change from landmark to landmark).
//
// Both run the same number of landmarks per workgroup from an LDS-resident tile (no HBM traffic in the timed loop) on
// every CU, so the figure is the arithmetic / operand-delivery rate of the formulation, not a memory rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/schur_mfma_probe.hip -o /tmp/schur_probe && /tmp/schur_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int K = 5;             // observations per landmark (the metric configuration)
constexpr int NLM = 47;          // landmarks of a tile
constexpr int PAIRS = K * (K + 1) / 2;

__global__ void __launch_bounds__(256) valu_kernel(const double* __restrict__ Bg, const double* __restrict__ Dg, double* __restrict__ out,
                                                 int reps) {
  __shared__ double Bs[NLM * K * 18];
  __shared__ double Ds[NLM * 9];
  for (int i = threadIdx.x; i < NLM * K * 18; i += 256) Bs[i] = Bg[i];
  for (int i = threadIdx.x; i < NLM * 9; i += 256) Ds[i] = Dg[i];
  __syncthreads();
  // lane <-> pose pair (a <= b) of the K x K block grid; the lanes of a workgroup walk the landmarks
  const int pair = threadIdx.x % 16, grp = threadIdx.x / 16;
  int a = 0, b = 0;
  for (int p = 0, i = 0; i < K; ++i)
    for (int j = i; j < K; ++j, ++p)
      if (p == pair) { a = i; b = j; }
  double acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
  if (pair < PAIRS)
    for (int r = 0; r < reps; ++r)
      for (int l = grp; l < NLM; l += 16) {
        const double* Bi = Bs + (l * K + a) * 18;
        const double* Bj = Bs + (l * K + b) * 18;
        const double* D = Ds + l * 9;
        double V[18];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) V[rr + 6 * c] = Bi[rr] * D[3 * c] + Bi[rr + 6] * D[1 + 3 * c] + Bi[rr + 12] * D[2 + 3 * c];
#pragma unroll
        for (int c = 0; c < 6; ++c)
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) acc[rr + 6 * c] += V[rr] * Bj[c] + V[rr + 6] * Bj[c + 6] + V[rr + 12] * Bj[c + 12];
      }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 36; ++i) s += acc[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FLUSH>
__global__ void __launch_bounds__(256) mfma_kernel(const double* __restrict__ Bg, const double* __restrict__ Dg, double* __restrict__ out,
                                                 int reps) {
  __shared__ double Bs[NLM * 32 * 4];     // stacked B of a landmark, rows padded to 32, columns to 4: [k][row]
  __shared__ double Vs[NLM * 32 * 4];     // B Dinv, same layout
  __shared__ double Sacc[4][PAIRS * 36];  // per wave: the pair blocks (flush target)
  for (int i = threadIdx.x; i < NLM * 128; i += 256) {
    const int l = i / 128, k = (i % 128) / 32, row = i % 32;
    double bv = 0.0, vv = 0.0;
    if (k < 3 && row < 6 * K) {
      const int o = row / 6, rr = row % 6;
      const double* Bi = Bg + (l * K + o) * 18;
      bv = Bi[rr + 6 * k];
      const double* D = Dg + l * 9;
      vv = Bi[rr] * D[3 * k] + Bi[rr + 6] * D[1 + 3 * k] + Bi[rr + 12] * D[2 + 3 * k];
    }
    Bs[i] = bv;
    Vs[i] = vv;
  }
  for (int i = threadIdx.x; i < 4 * PAIRS * 36; i += 256) (&Sacc[0][0])[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x % 64, w = threadIdx.x / 64;
  const int lr = lane % 16, lk = lane / 16;
  d4 t00 = {0, 0, 0, 0}, t01 = {0, 0, 0, 0}, t11 = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (int l = w; l < NLM; l += 4) {
      // operands: A[i][k] in lane i + 16 k = V(16 t + i, k); B[j][k] in lane j + 16 k = B(16 t + j, k)
      const double a0 = Vs[l * 128 + lk * 32 + lr], a1 = Vs[l * 128 + lk * 32 + 16 + lr];
      const double b0 = Bs[l * 128 + lk * 32 + lr], b1 = Bs[l * 128 + lk * 32 + 16 + lr];
      if (FLUSH) { t00 = d4{0, 0, 0, 0}; t01 = t00; t11 = t00; }
      t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, t00, 0, 0, 0);
      t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, t01, 0, 0, 0);
      t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, t11, 0, 0, 0);
      if (FLUSH) {
        // element (row, col) of the 32 x 32 result -> block (row / 6, col / 6), upper block pairs only
        auto put = [&](const d4& t, int r0, int c0) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int row = r0 + lk + 4 * v, col = c0 + lr;
            const int oa = row / 6, ob = col / 6;
            if (ob < K && oa <= ob) {
              const int p = oa * K - oa * (oa - 1) / 2 + (ob - oa);
              (void)__hip_atomic_fetch_add(&Sacc[w][p * 36 + (row % 6) + 6 * (col % 6)], t[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        };
        put(t00, 0, 0);
        put(t01, 0, 16);
        put(t11, 16, 16);
      }
    }
  double s = 0.0;
  for (int v = 0; v < 4; ++v) s += t00[v] + t01[v] + t11[v];
  if (FLUSH) s += Sacc[w][lane];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  std::vector<double> hB(NLM * K * 18), hD(NLM * 9);
  srand(1);
  for (auto& v : hB) v = rand() / (double)RAND_MAX - 0.5;
  for (auto& v : hD) v = rand() / (double)RAND_MAX - 0.5;
  double *dB, *dD, *dout;
  const int blocks = 256 * 4;
  hipMalloc(&dB, hB.size() * 8); hipMalloc(&dD, hD.size() * 8); hipMalloc(&dout, (size_t)blocks * 256 * 8);
  hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dD, hD.data(), hD.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  auto time_it = [&](const char* name, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double lm = (double)blocks * NLM * reps;
    printf("%-28s %8.3f ms   %7.2f landmarks / ns (whole GPU)   %6.1f ns per landmark and workgroup\n", name, ms, lm / (ms * 1e6),
           ms * 1e6 / (NLM * (double)reps) );
  };
  time_it("valu (lane per pose pair)", [&] { valu_kernel<<<blocks, 256>>>(dB, dD, dout, reps); });
  time_it("mfma, accumulate in place", [&] { mfma_kernel<0><<<blocks, 256>>>(dB, dD, dout, reps); });
  time_it("mfma, scatter per landmark", [&] { mfma_kernel<1><<<blocks, 256>>>(dB, dD, dout, reps); });
  std::vector<double> ho(256);
  hipMemcpy(ho.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
  printf("(checksum %g)\n", ho[0] + ho[17]);
  return 0;
}
