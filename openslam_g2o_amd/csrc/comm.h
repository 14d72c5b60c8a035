// Collectives inside libg2ohip: one communicator per solver handle.
//   * RCCL over xGMI (one process per GPU): ncclCommInitRank / ncclAllReduce on the solver's stream.  The library is bound
//     at run time (dlopen of librccl.so.1, or the copy already loaded into the process -- PyTorch ships its own), so
//     libg2ohip.so keeps loading on a box without RCCL and a C++ consumer needs nothing but -lg2ohip.
//   * host callback: the caller supplies an all-reduce over host memory (MPI, gloo, a test harness); device buffers are
//     staged through pinned memory.  For several ranks sharing one GPU (RCCL refuses that) and for boxes without peer
//     access.  Not a performance path.
// The sharded solve (BlockSolver::solve_sharded) needs three latency-sized all-reduces per solve.
#pragma once
#include <dlfcn.h>

#include <cstring>

#include "common.h"

namespace g2ohip {

typedef int (*HostAllReduceFn)(void* ctx, double* host_buffer, size_t count, int op);   // in place; op 0 = sum, 1 = max

class Comm {
 public:
  enum Kind { kNone = 0, kRccl = 1, kHost = 2 };
  ~Comm() { destroy(); }
  Kind kind() const { return kind_; }
  int rank() const { return rank_; }
  int world() const { return world_; }

  static void unique_id(char* id128) {
    Api& a = api();
    unique_id_t id;
    check(a.get_unique_id(&id), "ncclGetUniqueId");
    std::memcpy(id128, id.internal, 128);
  }
  void init_rccl(int rank, int world, const char* id128) {
    destroy();
    Api& a = api();
    unique_id_t id;
    std::memcpy(id.internal, id128, 128);
    check(a.comm_init_rank(&comm_, world, id, rank), "ncclCommInitRank");
    kind_ = kRccl;
    rank_ = rank;
    world_ = world;
  }
  void init_host(int rank, int world, HostAllReduceFn fn, void* ctx) {
    destroy();
    if (!fn) throw ArgFailure("comm_init_host: null callback");
    fn_ = fn;
    ctx_ = ctx;
    kind_ = kHost;
    rank_ = rank;
    world_ = world;
  }
  void destroy() {
    if (kind_ == kRccl && comm_) (void)api().comm_destroy(comm_);
    comm_ = nullptr;
    if (stage_) (void)hipHostFree(stage_);
    stage_ = nullptr;
    stage_n_ = 0;
    kind_ = kNone;
    rank_ = 0;
    world_ = 1;
  }
  // in-place all-reduce of a device buffer, asynchronous on st for RCCL, synchronous for the host callback
  void all_reduce(double* dev, size_t n, int op, hipStream_t st) {
    if (n == 0 || kind_ == kNone) return;
    if (kind_ == kRccl) {
      check(api().all_reduce(dev, dev, n, /*ncclDouble*/ 8, op == 1 ? /*ncclMax*/ 2 : /*ncclSum*/ 0, comm_, st), "ncclAllReduce");
      return;
    }
    ensure_stage(n);
    G2OHIP_HIP_CHECK(hipMemcpyAsync(stage_, dev, n * sizeof(double), hipMemcpyDeviceToHost, st));
    G2OHIP_HIP_CHECK(hipStreamSynchronize(st));
    if (fn_(ctx_, stage_, n, op) != 0) throw StateFailure("host all-reduce callback failed");
    G2OHIP_HIP_CHECK(hipMemcpyAsync(dev, stage_, n * sizeof(double), hipMemcpyHostToDevice, st));
  }
  // a few host scalars (chi2, computeScale, the maximal diagonal entry); synchronises
  void all_reduce_host(double* host, size_t n, int op, hipStream_t st) {
    if (n == 0 || kind_ == kNone) return;
    if (kind_ == kHost) {
      if (fn_(ctx_, host, n, op) != 0) throw StateFailure("host all-reduce callback failed");
      return;
    }
    scal_.alloc(n);
    G2OHIP_HIP_CHECK(hipMemcpyAsync(scal_.p, host, n * sizeof(double), hipMemcpyHostToDevice, st));
    all_reduce(scal_.p, n, op, st);
    G2OHIP_HIP_CHECK(hipMemcpyAsync(host, scal_.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
    G2OHIP_HIP_CHECK(hipStreamSynchronize(st));
  }

 private:
  struct unique_id_t { char internal[128]; };
  struct Api {
    int (*get_unique_id)(unique_id_t*) = nullptr;
    int (*comm_init_rank)(void**, int, unique_id_t, int) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
  };
  static Api& api() {
    static Api a;
    static bool tried = false;
    if (!tried) {
      tried = true;
      // the copy already in the process first (one RCCL per process: PyTorch loads its own), then the system library
      void* h = dlopen(nullptr, RTLD_NOW | RTLD_GLOBAL);
      if (!h || !dlsym(h, "ncclAllReduce")) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (h) {
        a.get_unique_id = reinterpret_cast<int (*)(unique_id_t*)>(dlsym(h, "ncclGetUniqueId"));
        a.comm_init_rank = reinterpret_cast<int (*)(void**, int, unique_id_t, int)>(dlsym(h, "ncclCommInitRank"));
        a.comm_destroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
        a.all_reduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
        a.error_string = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
      }
    }
    if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_reduce)
      throw StateFailure("RCCL is not available in this process (librccl.so.1 could not be loaded)");
    return a;
  }
  static void check(int rc, const char* what) {
    if (rc == 0) return;
    Api& a = api();
    throw StateFailure(std::string(what) + ": " + (a.error_string ? a.error_string(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
  }
  void ensure_stage(size_t n) {
    if (n <= stage_n_) return;
    if (stage_) (void)hipHostFree(stage_);
    stage_ = nullptr;
    G2OHIP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&stage_), n * sizeof(double), hipHostMallocDefault));
    stage_n_ = n;
  }
  Kind kind_ = kNone;
  int rank_ = 0, world_ = 1;
  void* comm_ = nullptr;
  HostAllReduceFn fn_ = nullptr;
  void* ctx_ = nullptr;
  double* stage_ = nullptr;
  size_t stage_n_ = 0;
  DevBuf<double> scal_;
};

}  // namespace g2ohip
