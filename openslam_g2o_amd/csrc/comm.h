// Collectives inside libg2ohip: one communicator per solver handle.
//   * RCCL over xGMI (one process per GPU): ncclCommInitRank / ncclAllReduce on the solver's stream.  The library is bound
//     at run time (dlopen of librccl.so.1, or the copy already loaded into the process -- PyTorch ships its own), so
//     libg2ohip.so keeps loading on a box without RCCL and a C++ consumer needs nothing but -lg2ohip.
//   * host callback: the caller supplies an all-reduce over host memory (MPI, gloo, a test harness); device buffers are
//     staged through pinned memory.  For several ranks sharing one GPU (RCCL refuses that) and for boxes without peer
//     access.  Not a performance path.
//   * peer mailboxes (opt-in): every rank owns a device buffer with one slot per rank and parity, exported with
//     hipIpcGetMemHandle and mapped by the others; an all-reduce is ONE kernel that stores the rank's payload into its slot of
//     every peer's mailbox (over xGMI: plain stores, a system-scope fence, then the slot's sequence number) and ONE kernel that
//     waits for the sequence numbers of all slots and adds them in rank order (every rank forms the same sums).  Two launches and
//     one xGMI latency instead of a ring of 2 (N - 1) steps for the latency-sized payloads of the sharded solve (94 KB).
//     The handles travel through the caller's host all-reduce once; the host scalars keep using it.  Exercised with several
//     processes on ONE GPU only (tests/test_gpu_multirank.py): the coherence of a mailbox polled by its owner while a peer
//     writes it over xGMI can only be established on a multi-GPU node -- hence opt-in, RCCL stays the default.
// The sharded solve (BlockSolver::solve_sharded) needs three latency-sized all-reduces per solve.
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace g2ohip {

typedef int (*HostAllReduceFn)(void* ctx, double* host_buffer, size_t count, int op);   // in place; op 0 = sum, 1 = max

namespace {
constexpr int kPeerMaxWorld = 16;
constexpr int kPeerHeader = 64;   // doubles in front of the slots: 2 x world sequence numbers (64 bit), padded to 512 B
struct PeerTable { double* box[kPeerMaxWorld]; };

// layout of a rank's mailbox: [sequence numbers: parity x world][slots: parity x world x cap doubles]
__device__ __forceinline__ unsigned long long* peer_seq(double* box, int parity, int world, int r) {
  return reinterpret_cast<unsigned long long*>(box) + (size_t)parity * world + r;
}

// blockIdx.y = peer: this rank's payload into its slot of the peer's mailbox; the last workgroup of a peer publishes the sequence number
// The sequence number of a call lives in device memory (*dev_seq + 1; bumped by the last workgroup of the call's peer_sum_kernel), not
// in a kernel argument: the two launches can be replayed from a captured graph.
__global__ void __launch_bounds__(256) peer_put_kernel(PeerTable T, const double* __restrict__ src, size_t n, size_t cap, int world, int rank,
                                                       const unsigned long long* __restrict__ dev_seq, unsigned int* __restrict__ cnt) {
  const unsigned long long seq = __hip_atomic_load(dev_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int parity = (int)(seq & 1);
  const int q = blockIdx.y;
  double* dst = T.box[q] + kPeerHeader + ((size_t)parity * world + rank) * cap;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
  __threadfence_system();   // this thread's stores have reached the peer
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = __hip_atomic_fetch_add(cnt + q, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (done == gridDim.x) {
      __hip_atomic_store(cnt + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(peer_seq(T.box[q], parity, world, rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// waits until every rank's slot carries `seq`, then dst = slot 0 (+|max) slot 1 ... in rank order (bit-identical on all ranks).
// The wait is bounded (ticks of the 100 MHz wall clock): a peer that never arrives raises *err instead of hanging the GPU.
__global__ void __launch_bounds__(256) peer_sum_kernel(double* box, double* __restrict__ dst, size_t n, size_t cap, int world, unsigned long long* dev_seq,
                                                       unsigned int* __restrict__ cnt, int op, long long tick_limit, int* err) {
  const unsigned long long seq = __hip_atomic_load(dev_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const int parity = (int)(seq & 1);
  if ((int)threadIdx.x < world) {
    const unsigned long long* fl = peer_seq(box, parity, world, threadIdx.x);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      __builtin_amdgcn_s_sleep(16);
      if (wall_clock64() - t0 > tick_limit) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  __threadfence_system();
  const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(box + kPeerHeader + (size_t)parity * world * cap);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    // system-scope loads: the slots were written by other devices (or processes), not through this device's caches
    double acc = __longlong_as_double((long long)__hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    for (int q = 1; q < world; ++q) {
      const double v = __longlong_as_double((long long)__hip_atomic_load(slots + (size_t)q * cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
      acc = op == 1 ? fmax(acc, v) : acc + v;
    }
    dst[i] = acc;
  }
  __syncthreads();   // (every thread of the workgroup has read the sequence number)
  if (threadIdx.x == 0) {
    const unsigned int done = __hip_atomic_fetch_add(cnt + kPeerMaxWorld, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (done == gridDim.x) {   // the last workgroup of the call: the next call's number
      __hip_atomic_store(cnt + kPeerMaxWorld, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dev_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
}  // namespace

class Comm {
 public:
  enum Kind { kNone = 0, kRccl = 1, kHost = 2, kPeer = 3 };
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
  // Peer mailboxes: `cap` doubles per slot (larger payloads go in pieces); fn = the host all-reduce the handles (and later the
  // host scalars) travel through.  Collective: every rank of the world calls it, with the same cap.
  void init_peer(int rank, int world, HostAllReduceFn fn, void* ctx, size_t cap) {
    destroy();
    if (!fn) throw ArgFailure("comm_init_peer: null callback");
    if (world < 1 || world > kPeerMaxWorld) throw ArgFailure("comm_init_peer: 1 <= world <= 16");
    if (cap < 1024) cap = 1024;
    fn_ = fn;
    ctx_ = ctx;
    rank_ = rank;
    world_ = world;
    peer_cap_ = cap;
    kind_ = kPeer;   // from here on destroy() releases whatever has been set up so far
    // A rank whose local set-up fails does not throw in front of a collective call (its peers would wait there for ever): it
    // remembers why, takes part in the handle exchange with an empty row and in the final barrier, which sums a failure flag --
    // every rank then releases what it holds and fails together.
    std::string why;
    hipIpcMemHandle_t mine;
    std::memset(&mine, 0, sizeof(mine));
    const size_t doubles = kPeerHeader + 2 * (size_t)world * cap;
    try {
      // uncached (fine-grained) device memory when the runtime exports it, plain device memory otherwise
      void* p = nullptr;
      bool have = false;
      if (hipExtMallocWithFlags(&p, doubles * sizeof(double), hipDeviceMallocUncached) == hipSuccess && p) {
        if (hipIpcGetMemHandle(&mine, p) == hipSuccess) have = true;
        else {
          (void)hipGetLastError();
          (void)hipFree(p);
          p = nullptr;
        }
      } else {
        (void)hipGetLastError();
      }
      if (!have) {
        G2OHIP_HIP_CHECK(hipMalloc(&p, doubles * sizeof(double)));
        peer_box_[rank] = static_cast<double*>(p);
        G2OHIP_HIP_CHECK(hipIpcGetMemHandle(&mine, p));
      }
      peer_box_[rank] = static_cast<double*>(p);
      if (std::getenv("G2OHIP_COMM_DEBUG")) fprintf(stderr, "comm_init_peer: rank %d mailbox %s, %zu KB\n", rank, have ? "uncached (fine-grained)" : "plain device memory", doubles / 128);
      G2OHIP_HIP_CHECK(hipMemset(p, 0, doubles * sizeof(double)));
      G2OHIP_HIP_CHECK(hipDeviceSynchronize());
      // [0, 16): finished workgroups of a put per peer; [16]: ... of a sum; behind them (8-byte aligned) the sequence number of the last call
      G2OHIP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&peer_cnt_), (kPeerMaxWorld + 4) * sizeof(unsigned int)));
      G2OHIP_HIP_CHECK(hipMemset(peer_cnt_, 0, (kPeerMaxWorld + 4) * sizeof(unsigned int)));
      G2OHIP_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&peer_err_), sizeof(int), hipHostMallocMapped));
      *peer_err_ = 0;
    } catch (const std::exception& e) {
      why = e.what();
    }
    // all-gather of the handles through the host all-reduce: one double per byte, a rank fills its own row (the call is also
    // the barrier behind which every mailbox is zeroed)
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t");
    std::vector<double> hb((size_t)world * 64, 0.0);
    const unsigned char* mb = reinterpret_cast<const unsigned char*>(&mine);
    for (int i = 0; i < 64; ++i) hb[(size_t)rank * 64 + i] = mb[i];
    if (world > 1 && fn_(ctx_, hb.data(), hb.size(), 0) != 0) {
      destroy();
      throw StateFailure("comm_init_peer: host all-reduce callback failed");
    }
    for (int q = 0; q < world && why.empty(); ++q) {
      if (q == rank) continue;
      hipIpcMemHandle_t h;
      unsigned char* hbq = reinterpret_cast<unsigned char*>(&h);
      for (int i = 0; i < 64; ++i) hbq[i] = (unsigned char)hb[(size_t)q * 64 + i];
      void* pq = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&pq, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        why = std::string("hipIpcOpenMemHandle (mailbox of rank ") + std::to_string(q) + "): " + hipGetErrorString(e);
        break;
      }
      peer_box_[q] = static_cast<double*>(pq);
    }
    // nobody stores into a mailbox before every rank has mapped them all (a peer's first put may otherwise race a late memset-free
    // start-up elsewhere: cheap insurance); the same call tells every rank whether any of them failed
    double bad = why.empty() ? 0.0 : 1.0;
    if (world > 1 && fn_(ctx_, &bad, 1, 0) != 0) {
      destroy();
      throw StateFailure("comm_init_peer: host all-reduce callback failed");
    }
    if (bad > 0.0) {
      destroy();
      throw StateFailure("comm_init_peer: " + (why.empty() ? std::string("another rank could not set up its mailboxes") : why));
    }
  }
  // (after a synchronisation of the stream the collectives ran on) a peer never arrived within the wait limit
  void poll_error() {
    if (kind_ == kPeer && peer_err_ && *peer_err_) {
      *peer_err_ = 0;
      throw StateFailure("peer exchange: a rank did not deliver its payload within the wait limit");
    }
  }
  void set_peer_wait_seconds(double s) { peer_wait_ticks_ = (long long)(s * 1e8); }
  void destroy() {
    if (kind_ == kRccl && comm_) (void)api().comm_destroy(comm_);
    comm_ = nullptr;
    if (kind_ == kPeer) {
      (void)hipDeviceSynchronize();
      for (int q = 0; q < world_; ++q) {
        if (!peer_box_[q]) continue;
        if (q == rank_) (void)hipFree(peer_box_[q]);
        else (void)hipIpcCloseMemHandle(peer_box_[q]);
        peer_box_[q] = nullptr;
      }
      if (peer_cnt_) (void)hipFree(peer_cnt_);
      if (peer_err_) (void)hipHostFree(peer_err_);
      peer_cnt_ = nullptr;
      peer_err_ = nullptr;
    }
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
    if (kind_ == kPeer) {
      PeerTable T;
      for (int q = 0; q < kPeerMaxWorld; ++q) T.box[q] = q < world_ ? peer_box_[q] : nullptr;
      for (size_t off = 0; off < n; off += peer_cap_) {
        const size_t m = std::min(peer_cap_, n - off);
        unsigned long long* dev_seq = reinterpret_cast<unsigned long long*>(peer_cnt_ + kPeerMaxWorld + 2);
        const int nb = (int)std::min<size_t>(64, (m + 1023) / 1024);
        hipLaunchKernelGGL(peer_put_kernel, dim3(nb, world_), dim3(256), 0, st, T, dev + off, m, peer_cap_, world_, rank_, dev_seq, peer_cnt_);
        hipLaunchKernelGGL(peer_sum_kernel, dim3(nb), dim3(256), 0, st, peer_box_[rank_], dev + off, m, peer_cap_, world_, dev_seq, peer_cnt_, op,
                           peer_wait_ticks_, peer_err_);
      }
      G2OHIP_HIP_CHECK(hipGetLastError());
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
    if (kind_ == kHost || kind_ == kPeer) {
      if (kind_ == kPeer) {   // (the callers have synchronised the stream or are about to: report a missed delivery here)
        G2OHIP_HIP_CHECK(hipStreamSynchronize(st));
        poll_error();
      }
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
  double* peer_box_[kPeerMaxWorld] = {nullptr};   // [rank_] = own mailbox, the others mapped through hipIpcOpenMemHandle
  size_t peer_cap_ = 0;
  unsigned int* peer_cnt_ = nullptr;
  int* peer_err_ = nullptr;
  long long peer_wait_ticks_ = 5LL * 100000000LL;   // 5 s of the 100 MHz wall clock
};

}  // namespace g2ohip
