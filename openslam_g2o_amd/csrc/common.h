// Shared helpers for libg2ohip: error plumbing and device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace g2ohip {

std::string& last_error_ref();
inline void set_error(const std::string& s) { last_error_ref() = s; }

struct HipFailure : std::runtime_error {
  explicit HipFailure(const std::string& s) : std::runtime_error(s) {}
};
struct ArgFailure : std::runtime_error {
  explicit ArgFailure(const std::string& s) : std::runtime_error(s) {}
};
struct StateFailure : std::runtime_error {
  explicit StateFailure(const std::string& s) : std::runtime_error(s) {}
};

#define G2OHIP_HIP_CHECK(expr)                                                                       \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      throw ::g2ohip::HipFailure(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                                 ":" + std::to_string(__LINE__) + ")");                              \
  } while (0)

// fn(begin, end) over [0, n) on up to G2OHIP_HOST_THREADS (default 8, at most the hardware's) host threads: the one-time set-up
// loops over the edges / observations of a graph (5 M at the metric configuration).  Chunks are contiguous and in order, so a
// caller that gives every chunk its own output and concatenates them keeps the sequential order.  No HIP calls inside fn.
inline int host_threads() {
  static const int nt = [] {
    const char* e = std::getenv("G2OHIP_HOST_THREADS");
    int v = e ? std::atoi(e) : 8;
    const unsigned hc = std::thread::hardware_concurrency();
    if (hc > 0 && v > (int)hc) v = (int)hc;
    return v < 1 ? 1 : v;
  }();
  return nt;
}
template <class Fn>
inline void host_parallel_chunks(size_t n, size_t nchunks, Fn fn) {   // fn(chunk, begin, end)
  if (nchunks < 1) nchunks = 1;
  const size_t step = (n + nchunks - 1) / nchunks;
  if (nchunks == 1 || n == 0) {
    fn((size_t)0, (size_t)0, n);
    return;
  }
  std::vector<std::thread> pool;
  std::vector<std::exception_ptr> err(nchunks);
  auto body = [&](size_t c) {
    const size_t b = c * step < n ? c * step : n, e = b + step < n ? b + step : n;
    try {
      fn(c, b, e);
    } catch (...) {
      err[c] = std::current_exception();
    }
  };
  for (size_t c = 1; c < nchunks; ++c) pool.emplace_back(body, c);
  body(0);
  for (auto& t : pool) t.join();
  for (auto& e : err)
    if (e) std::rethrow_exception(e);
}
template <class Fn>
inline void host_parallel_for(size_t n, Fn fn, size_t grain = (size_t)1 << 15) {   // fn(begin, end)
  const size_t nt = (size_t)host_threads(), want = (n + grain - 1) / grain;
  host_parallel_chunks(n, want < nt ? want : nt, [&](size_t, size_t b, size_t e) { fn(b, e); });
}

// When set, DevBuf operations are skipped: lets the host-only symbolic analysis (partition queries, CPU
// tests) share the code path of the device build without touching HIP.
inline bool& host_only_flag() {
  static thread_local bool f = false;
  return f;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      n = o.n;
      o.p = nullptr;
      o.n = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (host_only_flag()) return;
    if (count <= n && p) return;
    release();
    if (count == 0) count = 1;
    G2OHIP_HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
    n = count;
  }
  void upload(const T* h, size_t count, hipStream_t st) {
    if (host_only_flag()) return;
    alloc(count);
    if (count) G2OHIP_HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void upload(const std::vector<T>& v, hipStream_t st) {
    upload(v.data(), v.size(), st);
    // pageable host memory: the copy is staged before return, v may die afterwards
  }
  void download(T* h, size_t count, hipStream_t st) const {
    if (count) G2OHIP_HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, st));
    G2OHIP_HIP_CHECK(hipStreamSynchronize(st));
  }
  void zero(hipStream_t st) {
    if (host_only_flag()) return;
    if (p && n) G2OHIP_HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), st));
  }
};

struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;
  bool armed = false;
  void ensure() {
    if (!a) {
      G2OHIP_HIP_CHECK(hipEventCreate(&a));
      G2OHIP_HIP_CHECK(hipEventCreate(&b));
    }
  }
  void start(hipStream_t st) {
    ensure();
    G2OHIP_HIP_CHECK(hipEventRecord(a, st));
  }
  void stop(hipStream_t st) {
    G2OHIP_HIP_CHECK(hipEventRecord(b, st));
    armed = true;
  }
  double seconds() {
    if (!armed) return 0.0;
    float ms = 0;
    G2OHIP_HIP_CHECK(hipEventSynchronize(b));
    G2OHIP_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return 1e-3 * ms;
  }
  ~EventTimer() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};


// global -> LDS copy with U independent loads in flight per thread.  A plain `for (i...) lds[i] = g[i]`
// loop is compiled as load / wait / store per iteration and pays one full HBM latency per element.
template <int U, class T>
__device__ __forceinline__ void stage_copy(T* __restrict__ dst, const T* __restrict__ src, int n, int tid, int nthreads) {
  for (int base = tid; base < n; base += U * nthreads) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * nthreads;
      if (i < n) v[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * nthreads;
      if (i < n) dst[i] = v[u];
    }
  }
}

// Per-kernel timing with HIP events on the launch stream (enabled by g2ohip_set_profiling).
struct KernelProf {
  enum Slot { kAsmPose = 0, kAsmLandmark, kAsmOffPP, kAsmOffPL, kLmInverse, kSchurBlocks, kSchurRhs, kCholFactor, kCholSolve,
              kBackSub, kLambda, kExBoundary, kExRoots, kExHalo, kCholBand, kNumSlots };
  static const char* name(int s) {
    static const char* n[] = {"assemble_vertex(pose)", "assemble_vertex(landmark)", "assemble_offdiag(Hpp)", "assemble_offdiag(Hpl)",
                              "landmark_inverse", "schur_tiles", "schur_reduce", "chol_factor(all levels)", "chol_solve(all levels)",
                              "back_substitute", "set_lambda/restore", "exchange(boundary blocks + b_p)", "exchange(subtree roots)",
                              "exchange(halo x_p + status)", "chol_factor(band chains)"};
    return (s >= 0 && s < kNumSlots) ? n[s] : "?";
  }
  bool enabled = false;
  int only = -1;   // >= 0: time this slot only (two events per iteration instead of ~20: the records are not free)
  struct Pair { hipEvent_t a, b; bool cont; };   // cont: the slot was interrupted by another one and goes on (not a new launch)
  std::vector<Pair> pending[kNumSlots];
  std::vector<hipEvent_t> pool;
  double total[kNumSlots] = {0};
  long launches[kNumSlots] = {0};
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; G2OHIP_HIP_CHECK(hipEventCreate(&e)); return e;
  }
  bool timing(int slot) const { return enabled && (only < 0 || slot == only); }
  void begin(int slot, hipStream_t st, bool cont = false) {
    if (!enabled || (only >= 0 && slot != only)) return;
    Pair p{get(), get(), cont};
    G2OHIP_HIP_CHECK(hipEventRecord(p.a, st));
    pending[slot].push_back(p);
  }
  void end(int slot, hipStream_t st) {
    if (!enabled || (only >= 0 && slot != only)) return;
    G2OHIP_HIP_CHECK(hipEventRecord(pending[slot].back().b, st));
  }
  // synchronises the recorded events and folds them into total/launches
  void collect() {
    for (int s = 0; s < kNumSlots; ++s) {
      for (Pair& p : pending[s]) {
        float ms = 0;
        G2OHIP_HIP_CHECK(hipEventSynchronize(p.b));
        G2OHIP_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        total[s] += 1e-3 * ms;
        if (!p.cont) launches[s]++;
        pool.push_back(p.a);
        pool.push_back(p.b);
      }
      pending[s].clear();
    }
  }
  void reset() {
    collect();
    for (int s = 0; s < kNumSlots; ++s) { total[s] = 0; launches[s] = 0; }
  }
  ~KernelProf() {
    for (int s = 0; s < kNumSlots; ++s) for (Pair& p : pending[s]) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  }
};

}  // namespace g2ohip
