// Probe: do v_mfma_f64_16x16x4_f64 and fp64 VALU fused multiply-adds of DIFFERENT waves on one SIMD run side by side on gfx950,
// or do they share the fp64 datapath?  (What a Schur tile with its destination products on the matrix cores could hide
// behind the observation evaluation of the other waves -- DESIGN.md section 6.)
// Workgroups of 512 threads: waves 0-3 and waves 4-7 land on SIMDs 0-3 in turn, so every SIMD holds one wave of each half.
//   mode 0: waves 0-3 issue MFMAs, waves 4-7 idle        mode 1: waves 0-3 idle, waves 4-7 issue FMAs
//   mode 2: both at once                                 mode 3: every wave interleaves 1 MFMA with NF independent FMAs
//   mode 4: all eight waves MFMA                         mode 5: all eight waves FMA
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_overlap tools/probe/mfma_valu_overlap.hip && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NF = 16;   // FMAs per MFMA in the work units below: 16 x 4 cycles = the 64 cycles of one MFMA

__device__ __forceinline__ void mfma_unit(d4 (&acc)[4], double a, double b) {
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
}
__device__ __forceinline__ void fma_unit(double (&r)[NF], double b, double c) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int k = 0; k < NF; ++k) r[k] = fma(r[k], b, c);
}

template <int MODE>
__global__ void __launch_bounds__(512) probe(double* out, int iters, double seed) {
  const int w = threadIdx.x >> 6;
  double a = seed + threadIdx.x * 1e-3, b = 1.0 + 1e-12 * threadIdx.x, c = 1e-9;
  d4 acc[4];
  double r[NF];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = d4{0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < NF; ++k) r[k] = a + k;
  const bool do_m = MODE == 0 ? w < 4 : MODE == 2 ? w < 4 : MODE == 4;
  const bool do_f = MODE == 1 ? w >= 4 : MODE == 2 ? w >= 4 : MODE == 5;
  if (MODE == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NF; ++q) r[q] = fma(r[q], b, c);
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) mfma_unit(acc, a, b);
  } else if (do_f) {
    for (int it = 0; it < iters; ++it) fma_unit(r, b, c);
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
#pragma unroll
  for (int k = 0; k < NF; ++k) s += r[k];
  if (s == 12345.678) out[threadIdx.x] = s;
}

template <int MODE>
static double run(double* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, d, 16, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* d;
  hipMalloc(&d, 4096);
  const int iters = 20000;
  const double t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters),
               t5 = run<5>(d, iters);
  // per unit: 4 MFMAs (4 x 64 cycles nominal) or 64 FMAs (64 x 4 cycles nominal)
  auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 / iters; };
  printf("one workgroup of 8 waves per CU (2 waves per SIMD), %d units per wave; a unit = 4 MFMA f64 16x16x4 or 64 fp64 FMAs (256 cycles nominal each)\n", iters);
  printf("mode 0  MFMA waves alone (1 per SIMD)            %8.3f ms   %7.1f cycles per unit at 2.4 GHz\n", t0, cyc(t0));
  printf("mode 1  FMA waves alone (1 per SIMD)             %8.3f ms   %7.1f\n", t1, cyc(t1));
  printf("mode 2  MFMA wave + FMA wave on every SIMD       %8.3f ms   %7.1f   (sum of the two alone %.3f, max %.3f)\n", t2, cyc(t2), t0 + t1,
         t0 > t1 ? t0 : t1);
  printf("mode 3  one wave kind: 1 MFMA + 16 FMAs x 4      %8.3f ms   %7.1f   (2 waves per SIMD, each doing both units)\n", t3, cyc(t3));
  printf("mode 4  MFMA on both waves of a SIMD             %8.3f ms   %7.1f\n", t4, cyc(t4));
  printf("mode 5  FMA on both waves of a SIMD              %8.3f ms   %7.1f\n", t5, cyc(t5));
  return 0;
}
