// Empirical operand / result layout of v_mfma_f64_16x16x4_f64 on gfx950 (one wave): prints, per (lane, reg), which
// element (i, j) of D = A * B^T the result register holds, for A[i][k] loaded by lane (i + 16 k), B[j][k] by lane (j + 16 k).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16], b = B[(l % 16) * 4 + l / 16];
  double4_t acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[l * 4 + v] = acc[v];
}
int main() {
  double hA[64], hB[64], hD[256], ref[16][16];
  for (int i = 0; i < 64; ++i) { hA[i] = std::sin(1.0 + i); hB[i] = std::cos(2.0 + 3 * i); }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[j * 4 + k]; ref[i][j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int okA = 1, okB = 1;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const double d = hD[l * 4 + v];
    int fi = -1, fj = -1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (std::fabs(ref[i][j] - d) < 1e-13) { fi = i; fj = j; }
    if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> (%d,%d)\n", l, v, fi, fj);
    if (!(fi == 4 * (l / 16) + v && fj == l % 16)) okA = 0;
    if (!(fi == (l / 16) + 4 * v && fj == l % 16)) okB = 0;
  }
  printf("layout i=4*(lane/16)+reg, j=lane%%16: %d ; layout i=lane/16+4*reg: %d\n", okA, okB);
  return 0;
}
