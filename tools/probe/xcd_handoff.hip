// Hand-off of a 9.4 KB update matrix between two workgroups through memory: producer stores + acknowledgement + flag, consumer polls
// the flag and loads -- on the SAME XCD (blockIdx p and p + 8 under round-robin dispatch) and on DIFFERENT XCDs (p and p + 1), with
// device-coherent (sc1) and XCD-local (sc0 / plain) accesses.  Prints the XCC_ID of every workgroup, the consumer's load time after
// the flag, the producer's store + acknowledgement time, and whether the data was the expected one (stale lines would show).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_handoff tools/probe/xcd_handoff.hip && /tmp/xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 1176, IT = 200;   // doubles per hand-off (packed lower triangle of a 48 x 48 update matrix), repetitions
template <int SCOPE_ST, int SCOPE_LD>   // 0: plain, 1: workgroup (sc0), 2: agent (sc1)
__global__ void k(double* buf, int* flags, int* back, long long* out, int* xcc, int partner_delta) {
  const int b = blockIdx.x, tid = threadIdx.x;
  int id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (tid == 0) xcc[b] = id & 0xf;
  const bool producer = b == 0, consumer = b == partner_delta;
  if (!producer && !consumer) return;
  double* data = buf;
  long long tsum = 0, bad = 0;
  for (int it = 1; it <= IT; ++it) {
    if (producer) {
      if (tid == 0) while (__hip_atomic_load(back, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it - 1) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      const long long t0 = wall_clock64();
      for (int i = tid; i < N; i += blockDim.x) {
        const double v = it * 1000.0 + i;
        if (SCOPE_ST == 2) __hip_atomic_store(data + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE_ST == 1) __hip_atomic_store(data + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else data[i] = v;
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) {
        tsum += wall_clock64() - t0;
        __hip_atomic_store(flags, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (tid == 0) while (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it) __builtin_amdgcn_s_sleep(1);
      __syncthreads();
      const long long t0 = wall_clock64();
      double s = 0.0;
      for (int i = tid; i < N; i += blockDim.x) {
        double v;
        if (SCOPE_LD == 2) v = __hip_atomic_load(data + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE_LD == 1) v = __hip_atomic_load(data + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else v = data[i];
        if (v != it * 1000.0 + i) s += 1.0;
      }
      __shared__ double sh[256];
      sh[tid] = s;
      __syncthreads();
      if (tid == 0) {
        double a = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) a += sh[i];
        tsum += wall_clock64() - t0;
        bad += (long long)a;
        __hip_atomic_store(back, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    out[producer ? 0 : 1] = tsum;
    if (consumer) out[2] = bad;
  }
}
template <int S, int L>
void run(const char* what, int delta) {
  double* buf; int *flags, *back, *xcc; long long* out;
  hipMalloc(&buf, N * 8); hipMalloc(&flags, 4); hipMalloc(&back, 4); hipMalloc(&xcc, 64 * 4); hipMalloc(&out, 3 * 8);
  hipMemset(flags, 0, 4); hipMemset(back, 0, 4); hipMemset(out, 0, 24); hipMemset(buf, 0, N * 8);
  hipLaunchKernelGGL((k<S, L>), dim3(16), dim3(256), 0, 0, buf, flags, back, out, xcc, delta);
  hipDeviceSynchronize();
  long long h[3]; int hx[16];
  hipMemcpy(h, out, 24, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost);
  printf("%-44s producer XCC %d consumer (block %2d) XCC %d | store+ack %.2f us  load %.2f us  mismatching doubles %lld of %d\n", what, hx[0], delta, hx[delta],
         h[0] * 0.01 / IT, h[1] * 0.01 / IT, h[2], N * IT);
  hipFree(buf); hipFree(flags); hipFree(back); hipFree(xcc); hipFree(out);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<2, 2>("other XCD:  st sc1, ld sc1 (product)", 1);
    run<2, 2>("same XCD:   st sc1, ld sc1", 8);
    run<2, 1>("same XCD:   st sc1, ld sc0", 8);
    run<1, 1>("same XCD:   st sc0, ld sc0", 8);
    run<0, 1>("same XCD:   st plain, ld sc0", 8);
    run<2, 1>("other XCD:  st sc1, ld sc0 (expected stale?)", 1);
    run<0, 1>("other XCD:  st plain, ld sc0 (expected stale)", 1);
  }
  return 0;
}
