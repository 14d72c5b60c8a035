// Probe: the pivot-block step ("k-block") of wave_front.inc / band_chain.inc in isolation -- a 5 x 5 upper tile window in
// accumulator registers, NW waves per workgroup, steps of four k-blocks on tile row 0 followed by the band kernel's
// register shift, no HBM traffic and no assembly.  Reports the time per k-block for a lone workgroup and for a full GPU
// (every CU loaded to the occupancy the variant allows), so that a change of the step's structure can be judged
// without the rest of the solver:
//   var 0: as shipped -- publish row panel, barrier, pivot block by every lane, scaled panel of row tile t by wave
//          t mod NW -> LDS, barrier, rank-4 update from LDS
//   var 1: every wave computes the scaled panels it needs itself (redundant MFMAs, the matrix pipe is ~10 % busy): the
//          result register IS the operand of the update -- one barrier per k-block, no Lb traffic
//   hipcc --offload-arch=gfx950 -O3 tools/probe/kblock_probe.hip -o /tmp/kblock_probe && /tmp/kblock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double neg(double v) { return __hiloint2double(__double2hiint(v) ^ (int)0x80000000u, __double2loint(v)); }
__device__ __forceinline__ void sqrt_and_rsqrt(double d, double& s, double& r) {
  const double y = __builtin_amdgcn_rsq(d);
  double g = d * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double dd = fma(-g, g, d);
  g = fma(dd, h, g);
  s = g;
  r = h + h;
}
constexpr int slot(int ti, int tj, int kind) { return ti | (tj << 4) | (kind << 8); }
constexpr int kPad = slot(5, 5, 3);
template <int NW> struct Slots;
template <> struct Slots<4> {
  static constexpr int OWN = 4;
  static constexpr int tab[4][4] = {{slot(0, 0, 1), slot(1, 1, 1), slot(2, 2, 2), slot(3, 3, 0)}, {slot(0, 3, 1), slot(1, 3, 1), slot(2, 3, 2), slot(3, 4, 0)},
                                    {slot(0, 4, 1), slot(1, 4, 1), slot(2, 4, 2), slot(4, 4, 0)}, {slot(0, 1, 1), slot(1, 2, 2), slot(0, 2, 2), kPad}};
};
template <> struct Slots<3> {
  static constexpr int OWN = 5;
  static constexpr int tab[3][5] = {{slot(0, 0, 1), slot(1, 1, 1), slot(2, 2, 2), slot(3, 3, 0), slot(0, 2, 2)},
                                    {slot(0, 3, 1), slot(1, 3, 1), slot(2, 3, 2), slot(0, 1, 1), slot(1, 2, 2)},
                                    {slot(0, 4, 1), slot(1, 4, 1), slot(2, 4, 2), slot(3, 4, 0), slot(4, 4, 0)}};
};
template <> struct Slots<2> {
  static constexpr int OWN = 8;
  static constexpr int tab[2][8] = {{slot(0, 0, 1), slot(1, 1, 1), slot(2, 2, 2), slot(0, 3, 1), slot(1, 3, 1), slot(2, 3, 2), slot(3, 3, 0), slot(0, 2, 2)},
                                    {slot(0, 1, 1), slot(1, 2, 2), slot(0, 4, 1), slot(1, 4, 1), slot(2, 4, 2), slot(3, 4, 0), slot(4, 4, 0), kPad}};
};
template <> struct Slots<1> {
  static constexpr int OWN = 15;
  static constexpr int tab[1][15] = {{slot(0, 0, 1), slot(1, 1, 1), slot(2, 2, 2), slot(0, 3, 1), slot(1, 3, 1), slot(2, 3, 2), slot(0, 4, 1), slot(1, 4, 1),
                                      slot(2, 4, 2), slot(0, 1, 1), slot(1, 2, 2), slot(0, 2, 2), slot(3, 3, 0), slot(3, 4, 0), slot(4, 4, 0)}};
};

template <int NW, int VAR>
__global__ void __launch_bounds__(64 * NW, 4) kblock_kernel(int nsteps, double* __restrict__ out, long long* __restrict__ cyc, double* __restrict__ Lg) {
  __shared__ __attribute__((aligned(16))) double Rb[5 * 64];
  __shared__ __attribute__((aligned(16))) double Lb[5 * 64];
  constexpr int OWN = Slots<NW>::OWN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int oti[OWN], otj[OWN], okind[OWN];
#pragma unroll
  for (int k = 0; k < OWN; ++k) {
    int v = kPad;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww)
      if (w == ww) v = Slots<NW>::tab[ww][k];
    oti[k] = v & 15;
    otj[k] = (v >> 4) & 15;
    okind[k] = (v >> 8) & 15;
  }
  d4 S[OWN];
  const int lr = lane & 15, lk = lane >> 4;
  auto fresh = [&](int ti, int tj, int step) {
    d4 r;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = lk + 4 * v, j = lr;
      r[v] = (ti == tj && i == j) ? 1000.0 + step : 0.01 * (1 + ((i * 7 + j * 3 + ti + tj) & 7));
      if (ti == tj && i != j) r[v] = 0.01 * (1 + (((i < j ? i : j) * 7 + (i < j ? j : i) * 3) & 7));   // symmetric
    }
    return r;
  };
#pragma unroll
  for (int k = 0; k < OWN; ++k) S[k] = fresh(oti[k], otj[k], 0);
  const long long t0 = wall_clock64();
  bool bad = false;
  for (int step = 0; step < nsteps; ++step) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int k0 = 4 * kb;
#pragma unroll
      for (int k = 0; k < OWN; ++k)
        if (oti[k] == 0) Rb[otj[k] * 64 + lane] = S[k][kb];
      __syncthreads();
      double D[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) D[a][b] = Rb[(k0 + b) + 16 * a];
      double aop = 0.0, sqsel = 0.0, rssel = 0.0;
      {
        double Ld[4][4], rs[4], sq[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          double d = D[jj][jj];
          if (!(d > 0.0)) {
            bad = true;
            d = 1.0;
          }
          sqrt_and_rsqrt(d, sq[jj], rs[jj]);
#pragma unroll
          for (int i = jj + 1; i < 4; ++i) Ld[i][jj] = D[jj][i] * rs[jj];
#pragma unroll
          for (int c = jj + 1; c < 4; ++c)
#pragma unroll
            for (int i = c; i < 4; ++i) D[c][i] -= Ld[i][jj] * Ld[c][jj];
        }
        double W[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          W[jj][jj] = rs[jj];
#pragma unroll
          for (int i = jj + 1; i < 4; ++i) {
            double s_ = 0.0;
#pragma unroll
            for (int k = jj; k < i; ++k) s_ += Ld[i][k] * W[k][jj];
            W[i][jj] = -rs[i] * s_;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k <= i; ++k)
            if (lr == i && lk == k) aop = W[i][k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lk == k) {
            sqsel = sq[k];
            rssel = rs[k];
          }
      }
      if constexpr (VAR == 0) {
#pragma unroll
        for (int t = 0; t < 5; ++t)
          if ((t % NW) == w) {
            const d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Rb[t * 64 + lane], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
            Lb[t * 64 + lane] = r[0];
            if (Lg) Lg[((size_t)blockIdx.x * 5 + t) * 64 + lane] = r[0] + sqsel + rssel;
          }
        __syncthreads();
        double la[OWN], lb[OWN];
#pragma unroll
        for (int k = 0; k < OWN; ++k) {
          la[k] = Lb[(oti[k] < 5 ? oti[k] : 4) * 64 + lane];
          lb[k] = Lb[(otj[k] < 5 ? otj[k] : 4) * 64 + lane];
        }
#pragma unroll
        for (int k = 0; k < OWN; ++k)
          if (okind[k] != 3) S[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(neg(la[k]), lb[k], S[k], 0, 0, 0);
      } else {
        double lp[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, Rb[t * 64 + lane], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
          lp[t] = r[0];
          if (Lg && (t % NW) == w) Lg[((size_t)blockIdx.x * 5 + t) * 64 + lane] = r[0] + sqsel + rssel;
        }
#pragma unroll
        for (int k = 0; k < OWN; ++k)
          if (okind[k] != 3) {
            double la = lp[0], lb = lp[0];
#pragma unroll
            for (int t = 1; t < 5; ++t) {
              if (oti[k] == t) la = lp[t];
              if (otj[k] == t) lb = lp[t];
            }
            S[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(neg(la), lb, S[k], 0, 0, 0);
          }
        if constexpr (VAR == 1) __syncthreads();   // (Rb is rewritten by the next publish)
      }
    }
#pragma unroll
    for (int k = 0; k < OWN; ++k) {
      if (okind[k] == 1 && k + 1 < OWN) S[k] = S[k + 1];
      if (okind[k] == 2) S[k] = fresh(oti[k], otj[k], step + 1);
    }
  }
  const long long t1 = wall_clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  double acc = bad ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < OWN; ++k) acc += S[k][0] + S[k][1] + S[k][2] + S[k][3];
  out[(size_t)blockIdx.x * 64 * NW + tid] = acc;
}


// One wave per window, everything at compile time: the row panel of a tile IS the B operand (no LDS), the pivot block
// comes over v_readlane, no barrier, no redundant pivot arithmetic.
constexpr int kT1[15][2] = {{0, 0}, {1, 1}, {2, 2}, {0, 3}, {1, 3}, {2, 3}, {0, 4}, {1, 4}, {2, 4}, {0, 1}, {1, 2}, {0, 2}, {3, 3}, {3, 4}, {4, 4}};
constexpr int kK1[15] = {1, 1, 2, 1, 1, 2, 1, 1, 2, 1, 2, 2, 0, 0, 0};
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__global__ void __launch_bounds__(64, 2) kblock_one_wave(int nsteps, double* __restrict__ out, long long* __restrict__ cyc, double* __restrict__ Lg) {
  const int lane = threadIdx.x;
  const int lr = lane & 15, lk = lane >> 4;
  d4 S[15];
  auto fresh = [&](int ti, int tj, int step) {
    d4 r;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = lk + 4 * v, j = lr;
      r[v] = (ti == tj && i == j) ? 1000.0 + step : 0.01 * (1 + ((i * 7 + j * 3 + ti + tj) & 7));
      if (ti == tj && i != j) r[v] = 0.01 * (1 + (((i < j ? i : j) * 7 + (i < j ? j : i) * 3) & 7));
    }
    return r;
  };
#pragma unroll
  for (int k = 0; k < 15; ++k) S[k] = fresh(kT1[k][0], kT1[k][1], 0);
  const long long t0 = wall_clock64();
  bool bad = false;
  for (int step = 0; step < nsteps; ++step) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int k0 = 4 * kb;
      // pivot block from the diagonal tile's row register: D[a][b] = F(k0 + a, k0 + b) sits in lane (k0 + b) + 16 a
      double D[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) D[a][b] = readlane_f64(S[0][kb], (k0 + b) + 16 * a);
      double aop = 0.0, sqsel = 0.0, rssel = 0.0;
      {
        double Ld[4][4], rs[4], sq[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          double d = D[jj][jj];
          if (!(d > 0.0)) {
            bad = true;
            d = 1.0;
          }
          sqrt_and_rsqrt(d, sq[jj], rs[jj]);
#pragma unroll
          for (int i = jj + 1; i < 4; ++i) Ld[i][jj] = D[jj][i] * rs[jj];
#pragma unroll
          for (int c = jj + 1; c < 4; ++c)
#pragma unroll
            for (int i = c; i < 4; ++i) D[c][i] -= Ld[i][jj] * Ld[c][jj];
        }
        double W[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          W[jj][jj] = rs[jj];
#pragma unroll
          for (int i = jj + 1; i < 4; ++i) {
            double s_ = 0.0;
#pragma unroll
            for (int k = jj; k < i; ++k) s_ += Ld[i][k] * W[k][jj];
            W[i][jj] = -rs[i] * s_;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k <= i; ++k)
            if (lr == i && lk == k) aop = W[i][k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (lk == k) {
            sqsel = sq[k];
            rssel = rs[k];
          }
      }
      // scaled panels: B operand = the row register of tile (0, t); slots of tile row 0: (0,0)=0, (0,1)=9, (0,2)=11, (0,3)=3, (0,4)=6
      constexpr int row0[5] = {0, 9, 11, 3, 6};
      double lp[5], ln[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, S[row0[t]][kb], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        lp[t] = r[0];
        ln[t] = neg(r[0]);
        if (Lg) Lg[((size_t)blockIdx.x * 5 + t) * 64 + lane] = r[0] + sqsel + rssel;
      }
#pragma unroll
      for (int k = 0; k < 15; ++k) S[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(ln[kT1[k][0]], lp[kT1[k][1]], S[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) {
      if (kK1[k] == 1) S[k] = S[k + 1];
      if (kK1[k] == 2) S[k] = fresh(kT1[k][0], kT1[k][1], step + 1);
    }
  }
  const long long t1 = wall_clock64();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  double acc = bad ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 15; ++k) acc += S[k][0] + S[k][1] + S[k][2] + S[k][3];
  out[(size_t)blockIdx.x * 64 + lane] = acc;
}

void run_one(int nsteps) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kblock_one_wave, 64, 0);
  for (int mode = 0; mode < 4; ++mode) {
    const int per_cu = mode == 0 ? 0 : (mode == 1 ? 4 : (mode == 2 ? 8 : occ));
    const int grid = mode == 0 ? 1 : prop.multiProcessorCount * per_cu;
    double *out, *Lg;
    long long* cyc;
    hipMalloc(&out, (size_t)grid * 64 * sizeof(double));
    hipMalloc(&cyc, (size_t)grid * sizeof(long long));
    hipMalloc(&Lg, (size_t)grid * 5 * 64 * sizeof(double));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kblock_one_wave<<<grid, 64>>>(nsteps, out, cyc, Lg);
    hipEventRecord(a);
    kblock_one_wave<<<grid, 64>>>(nsteps, out, cyc, Lg);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (long long v : h) s += (double)v;
    double ho = 0;
    hipMemcpy(&ho, out, sizeof(double), hipMemcpyDeviceToHost);
    printf("one wave, compile-time tiles   occ %d grid %5d: %.3f us per k-block (in-kernel stamps), kernel %.3f ms -> %.2f k-blocks/us over the GPU, check %.6e\n", occ, grid,
           s / grid * 0.01 / (4.0 * nsteps), ms, (double)grid * 4.0 * nsteps / (ms * 1e3), ho);
  }
}

template <int NW, int VAR>
void run(const char* name, int nsteps, bool stores) {
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kblock_kernel<NW, VAR>, 64 * NW, 0);
  const int ncu = prop.multiProcessorCount;
  for (int mode = 0; mode < 2; ++mode) {
    const int grid = mode == 0 ? 1 : ncu * occ;
    double* out;
    long long* cyc;
    double* Lg = nullptr;
    hipMalloc(&out, (size_t)grid * 64 * NW * sizeof(double));
    hipMalloc(&cyc, (size_t)grid * sizeof(long long));
    if (stores) hipMalloc(&Lg, (size_t)grid * 5 * 64 * sizeof(double));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kblock_kernel<NW, VAR><<<grid, 64 * NW>>>(nsteps, out, cyc, Lg);
    hipEventRecord(a);
    kblock_kernel<NW, VAR><<<grid, 64 * NW>>>(nsteps, out, cyc, Lg);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (long long v : h) s += (double)v;
    std::vector<double> ho((size_t)grid * 64 * NW);
    hipMemcpy(ho.data(), out, ho.size() * sizeof(double), hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz
    printf("%-28s waves %d occ %d grid %5d: %.3f us per k-block (in-kernel stamps), kernel %.3f ms -> %.2f k-blocks/us over the GPU, check %.6e\n", name, NW, occ,
           grid, s / grid * 0.01 / (4.0 * nsteps), ms, (double)grid * 4.0 * nsteps / (ms * 1e3), ho[0]);
    hipFree(out);
    hipFree(cyc);
    if (Lg) hipFree(Lg);
  }
}

int main(int argc, char** argv) {
  const int nsteps = argc > 1 ? atoi(argv[1]) : 64;
  run_one(nsteps);
  run<4, 0>("two barriers, Lb via LDS", nsteps, true);
  run<3, 0>("two barriers, Lb via LDS", nsteps, true);
  run<2, 0>("two barriers, Lb via LDS", nsteps, true);
  run<4, 1>("own panels, one barrier", nsteps, true);
  run<3, 1>("own panels, one barrier", nsteps, true);
  run<2, 1>("own panels, one barrier", nsteps, true);
  run<1, 2>("one wave, no barrier", nsteps, true);
  run<4, 0>("two barriers, no L stores", nsteps, false);
  run<4, 1>("own panels, no L stores", nsteps, false);
  return 0;
}
