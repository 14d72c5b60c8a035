// Issue / dependent-latency probe for the instruction kinds the one-wave pivot-block factorisation is made of (gfx950).
// One wavefront, wall_clock64 (100 MHz) around N repetitions; prints ns per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_latency tools/probe/issue_latency.hip && /tmp/issue_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double rl(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#define REP 64
template <int MODE>
__global__ void __launch_bounds__(64) probe(double* out, long long* t, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-3, b = 1.0 + 1e-9 * threadIdx.x, c = 0.5;
  double r[8];
  __shared__ double lds[64];
  lds[threadIdx.x] = seed + threadIdx.x;
  __syncthreads();
  for (int k = 0; k < 8; ++k) r[k] = a + k;
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < REP; ++k) {
      if (MODE == 0) a = fma(a, b, c);                                   // dependent fma chain
      if (MODE == 1) r[k & 7] = fma(r[k & 7], b, c);                     // 8 independent chains
      if (MODE == 2) a = fma(-b, rl(a, k & 31), a);                      // dependent: readlane x2 -> fma -> readlane ...
      if (MODE == 3) r[k & 7] = fma(-b, rl(a, k & 31), r[k & 7]);        // independent updates with a readlane multiplier
      if (MODE == 4) a = __builtin_amdgcn_rsq(a) + 1.5;                  // dependent rsq (+ add)
      if (MODE == 5) {                                                   // the pivot chain: readlane, rsq + 2 Goldschmidt steps, scale
        const double d = rl(a, k & 15);
        const double y = __builtin_amdgcn_rsq(d);
        double g = d * y, h = 0.5 * y;
        double e = fma(-h, g, 0.5);
        g = fma(g, e, g);
        h = fma(h, e, h);
        e = fma(-h, g, 0.5);
        h = fma(h, e, h);
        a = fma(a, h + h, 2.0);
      }
      if (MODE == 6) r[k & 7] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r[k & 7]), k & 31), __double2loint(r[k & 7]));   // readlane throughput (one per op), scalar -> vector move
      if (MODE == 7) a = a * b;                                           // dependent mul
      if (MODE == 8) {                                                    // multiplier moved into a VECTOR register first
        double sv = rl(a, k & 31);
        asm volatile("" : "+v"(sv));
        r[k & 7] = fma(-b, sv, r[k & 7]);
      }
      if (MODE == 9) r[k & 7] = fma(-b, __shfl(a, k & 31), r[k & 7]);     // ds_bpermute x2
      if (MODE == 10) r[k & 7] = fma(-b, lds[k & 31], r[k & 7]);          // LDS broadcast read
      if (MODE == 11) {                                                   // two multipliers requested ahead of their use (manual pipelining)
        const double s0 = rl(a, k & 31), s1 = rl(a, (k + 1) & 31), s2 = rl(a, (k + 2) & 31), s3 = rl(a, (k + 3) & 31);
        r[0] = fma(-b, s0, r[0]);
        r[1] = fma(-b, s1, r[1]);
        r[2] = fma(-b, s2, r[2]);
        r[3] = fma(-b, s3, r[3]);
      }
      asm volatile("" : "+v"(a));
    }
  }
  long long t1 = wall_clock64();
  double s = a;
  for (int k = 0; k < 8; ++k) s += r[k];
  out[threadIdx.x + 64 * blockIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

int main() {
  double* out;
  long long* t;
  hipMalloc(&out, 64 * 4096 * 8);
  hipMalloc(&t, 4096 * 8);
  const char* names[12] = {"dependent v_fma_f64", "independent v_fma_f64 (8 chains)", "dependent readlane x2 + fma", "independent readlane x2 + fma",
                          "dependent v_rsq_f64 + add", "pivot chain (readlane, rsq, 2 Goldschmidt steps, scale): per column", "readlane + v_mov per op",
                          "dependent v_mul_f64", "independent readlane x2 -> v_mov x2 -> fma", "independent ds_bpermute x2 + fma", "independent LDS broadcast read + fma",
                          "four readlane pairs, then four fma (per group of 4)"};
  const int iters = 200;
  for (int grid : {1}) {
    printf("== %d workgroup(s) of one wave\n", grid);
    for (int m = 0; m < 12; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (m) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 3: hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 4: hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 5: hipLaunchKernelGGL(probe<5>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 6: hipLaunchKernelGGL(probe<6>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 7: hipLaunchKernelGGL(probe<7>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 8: hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 9: hipLaunchKernelGGL(probe<9>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 10: hipLaunchKernelGGL(probe<10>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
          case 11: hipLaunchKernelGGL(probe<11>, dim3(grid), dim3(64), 0, 0, out, t, iters, 1.0); break;
        }
        hipDeviceSynchronize();
      }
      std::vector<long long> h(grid);
      hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost);
      double mx = 0;
      for (long long v : h) mx = v > mx ? v : mx;
      printf("%-75s %7.2f ns per op\n", names[m], mx * 10.0 / (iters * REP));
    }
  }
  return 0;
}
