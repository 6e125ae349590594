// Multi-GPU store: the model row-sharded by key range over the ranks of one node, one process per
// GPU, behind the C ABI (include/difacto_hip.h, "sharded store").  Replaces the ps-lite Push / Pull
// of src/store (include/difacto/store.h:53-93): every rank is a worker (it trains its own
// minibatches) and a server (it owns a contiguous range of the reversed keys — what ReverseBytes is
// for, include/difacto/base.h:29-38).  Because the Localizer emits a minibatch's keys in ascending
// order (src/data/localizer.cc:28-48) every owner's keys are one contiguous slice: send buffers need
// no permutation.
//
// One dfh_shard_step = the batch executor of SGDLearner::IterateData (sgd_learner.cc:131-178) with
// the store calls turned into one exchange each way:
//     counts   every rank learns how many keys each peer sends it (+ "I still have data")
//     L        the keys THIS rank owns never leave it: k_lookup on its own table (row ids, Push(kFeaCount))
//     K        the other keys (+ epoch-0 counts) --alltoallv--> owners
//     R        owners: resolve keys -> rows once, Push(kFeaCount) per source, Pull (one gather)
//     RW       rows --alltoallv--> workers          fixed stride dfh_row_stride(V_dim)
//     F        FMLoss::Predict / Evaluate on a mixed source: own keys read the table in place, the others
//              the pulled rows (k_forward<MIXED>); CalcGrad in two launches of k_backward_all: the others'
//              keys into gradient rows, the own keys straight into the fused in-place update
//     G        gradients --alltoallv--> owners
//     P        owners: Push(kGradient) of the other sources, applied in ascending rank order in ONE launch
// A key's pushes are applied owner first, then the other sources in ascending rank order — one legal
// execution of the reference's asynchronous Push protocol, the same on every run.  With one rank
// nothing is exchanged and nothing waits for the host: the step is dfh_sgd_step.
// Zero staleness: every minibatch reads the model all earlier ones have updated.  No all-reduce:
// the traffic is key-routed rows, an all-to-all that uses every xGMI link of a GPU at once.
//
// Transport (dfh_comm): RCCL ncclSend / ncclRecv grouped per exchange, on the context's stream —
// the library is loaded at run time (dlopen) so that a process which already carries an RCCL (a
// Python host with torch) shares it, and a host without multi-GPU needs never loads one; or a host
// callback (exchange staged through host memory) for process groups RCCL cannot serve — tests with
// several ranks sharing one GPU.
#ifndef DFH_SHARD_HIP_
#define DFH_SHARD_HIP_
#include <dlfcn.h>
#include <link.h>

#include <chrono>
#include <deque>
#include <thread>

namespace dfh {

// ---- the slice of RCCL's C API this file binds (rccl/rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220,
// ncclCommDestroy :260, ncclGetErrorString :339, ncclSend :700, ncclRecv :722, ncclGroupStart/End :923-933)
struct RcclId { char internal[128]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*GetVersion)(int*) = nullptr;
  std::string path;        // the file ncclSend was bound from (dladdr)
  bool was_loaded = false; // the process already carried this library (e.g. the copy inside torch/lib)
};

// a library of the process whose file name contains "librccl": a host that already carries RCCL (PyTorch bundles
// one) must share it — two RCCLs in one process would each run their own bootstrap and proxy threads
inline int rccl_phdr_cb(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
    *static_cast<std::string*>(out) = info->dlpi_name;
    return 1;
  }
  return 0;
}

inline RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // (1) whatever RCCL the process has already mapped, by its own path; (2) a loaded library under the usual
    // sonames; (3) load one
    std::string loaded;
    dl_iterate_phdr(rccl_phdr_cb, &loaded);
    if (!loaded.empty()) api.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      if (api.lib) break;
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    }
    api.was_loaded = api.lib != nullptr;
    for (const char* n : names) {
      if (api.lib) break;
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.lib) return;
#define DFH_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, name))
    DFH_SYM(GetUniqueId, "ncclGetUniqueId");
    DFH_SYM(CommInitRank, "ncclCommInitRank");
    DFH_SYM(CommDestroy, "ncclCommDestroy");
    DFH_SYM(GetErrorString, "ncclGetErrorString");
    DFH_SYM(Send, "ncclSend");
    DFH_SYM(Recv, "ncclRecv");
    DFH_SYM(GroupStart, "ncclGroupStart");
    DFH_SYM(GroupEnd, "ncclGroupEnd");
    DFH_SYM(GetVersion, "ncclGetVersion");
#undef DFH_SYM
    Dl_info di;
    if (api.Send && dladdr(reinterpret_cast<void*>(api.Send), &di) && di.dli_fname) api.path = di.dli_fname;
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd) {
      dlclose(api.lib);
      api.lib = nullptr;
    }
  });
  return api.lib ? &api : nullptr;
}

#define DFH_RCCL(call)                                                                                  \
  do {                                                                                                  \
    int r__ = (call);                                                                                   \
    if (r__ != 0) {                                                                                     \
      RcclApi* a__ = rccl_api();                                                                        \
      ::dfh::set_error(std::string(#call) + ": " + ((a__ && a__->GetErrorString) ? a__->GetErrorString(r__) : "RCCL error")); \
      return DFH_ERR_HIP;                                                                               \
    }                                                                                                   \
  } while (0)

// per-destination {keys I send you, do I still have data} -> device words for the counts exchange
__global__ void k_shard_counts(const int64_t* __restrict__ bounds, int world, int64_t has_data, int64_t* __restrict__ out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < world) {
    out[2 * d] = bounds ? bounds[d + 1] - bounds[d] : 0;
    out[2 * d + 1] = has_data;
  }
}

// ---- loop-back transport (dfh_comm_create_loopback): the copies of one exchange in ONE launch
struct LoopSegs {
  const char* src[64];
  char* dst[64];            // NULL: a send — the bytes are read and dropped
  unsigned long long bytes[64];
  int n;
};

// block b walks the segments in 16 B (or, for a misaligned segment, 4 B) units, grid-stride.  A "send" is read in full
// (one xor per unit, stored only if it hits a value it cannot hit: the loads stay) — on real wires the send buffer is
// read out of this GPU's HBM once and nothing is written here.  Thread 0 of block 0 stamps the wall clock for k_loop_wait.
__global__ void __launch_bounds__(256) k_loop_copy(LoopSegs g, unsigned long long* __restrict__ stamp, unsigned* __restrict__ sink) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && stamp) *stamp = wall_clock64();
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (int i = 0; i < g.n; ++i) {
    const char* s = g.src[i];
    char* d = g.dst[i];
    const size_t nb = g.bytes[i];
    const bool wide = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | nb) & 15) == 0;
    if (wide) {
      const uint4* s4 = reinterpret_cast<const uint4*>(s);
      uint4* d4 = reinterpret_cast<uint4*>(d);
      for (size_t j = tid; j < nb / 16; j += nth) {
        const uint4 v = s4[j];
        if (d) d4[j] = v; else acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    } else {
      const unsigned* s1 = reinterpret_cast<const unsigned*>(s);
      unsigned* d1 = reinterpret_cast<unsigned*>(d);
      for (size_t j = tid; j < nb / 4; j += nth) {
        const unsigned v = s1[j];
        if (d) d1[j] = v; else acc ^= v;
      }
    }
  }
  if (acc == 0x9E3779B9u && sink) *sink = acc;
}

// holds the stream until `ticks` wall-clock ticks after the stamp k_loop_copy left: the modelled wire time of the exchange
__global__ void k_loop_wait(const unsigned long long* __restrict__ stamp, unsigned long long ticks) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = *stamp;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  }
}

}  // namespace dfh

using namespace dfh;

// the list of the very hot keys' parts for the update's split role (SplitOut): filled by the launches of a training step that see
// the minibatch's keys — the own keys' lookup (ranks offset by own_lo) and the others' row words
static inline SplitOut shard_split_out(const dfh_ctx* ctx, dfh_batch* b, int is_train, uint32_t u_base) {
  return (b && is_train && ctx->upd_kernel && ctx->upd_split) ? SplitOut{b->d_split_ent, b->d_U + 2, u_base} : SplitOut{nullptr, nullptr, 0u};
}

struct dfh_comm {
  dfh_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  void* rccl = nullptr;                 // ncclComm_t, or
  dfh_alltoallv_fn fn = nullptr;        // host callback
  void* user = nullptr;
  char* h_send = nullptr;               // pinned staging of the callback transport
  char* h_recv = nullptr;
  size_t h_cap = 0;
  // payload this rank handed to / took from OTHER ranks since the last reset, and the message groups it took
  uint64_t bytes_sent = 0, bytes_recv = 0, groups = 0;
  // loop-back transport (measurement): fed sources per kind of exchange, the wire model
  bool loopback = false;
  std::deque<const void*> feed[DFH_XCHG_KINDS];
  const void* sticky[DFH_XCHG_KINDS] = {nullptr};
  unsigned long long* d_stamp = nullptr;   // [0]: wall clock at the start of the current exchange's copies; [1]: sink
  double link_gbps = 0, latency_us = 0;
  double wall_khz = 100000.0;              // hipDeviceAttributeWallClockRate
  double wire_us_sum = 0;                  // modelled wire time of all exchanges since the last stats reset
};

struct dfh_shard {
  dfh_table* t = nullptr;
  dfh_comm* c = nullptr;
  uint64_t* d_splits = nullptr;         // [world-1] first keys of shards 1.. (NULL: uniform ranges)
  std::vector<uint64_t> h_splits;       // the same on the host, uniform ranges spelled out
  int64_t* d_bounds = nullptr;          // [world+1]
  int64_t* d_cnt = nullptr;             // [2][2*world]: what I send every peer / what every peer sends me
  int64_t* h_cnt = nullptr;             // pinned copy
  // Two sets of exchange buffers: the sync step uses set 0; the overlapped step (two minibatches in flight)
  // alternates, one set per minibatch under way.
  // owner side, sized to the keys received in a step
  uint64_t* r_keys[2] = {nullptr, nullptr};
  float* r_cnt[2] = {nullptr, nullptr};
  uint32_t* r_rowid[2] = {nullptr, nullptr};
  float* r_rows[2] = {nullptr, nullptr};   // rows out, then gradients in
  size_t r_cap[2] = {0, 0};
  // worker side, sized to the batch's unique keys
  float* w_rows[2] = {nullptr, nullptr};
  size_t w_cap[2] = {0, 0};
  float* w_grads[2] = {nullptr, nullptr};
  size_t g_cap[2] = {0, 0};
  uint64_t steps = 0;
  // counts of the FOLLOWING step, exchanged inside the current one (dfh_shard_prefetch_counts)
  dfh_batch* next_b = nullptr;
  bool next_armed = false;            // the next dfh_shard_step issues the counts exchange of next_b
  bool counts_ready = false;          // h_cnt holds the counts of counts_for; cnt_ev marks their arrival
  dfh_batch* counts_for = nullptr;
  hipEvent_t cnt_ev = nullptr;
  // ---- overlapped exchange (dfh_shard_set_exchange(s, 1)): two minibatches in flight
  int exchange = 0;                   // 0: sync (one minibatch at a time, zero staleness), 1: overlap (staleness <= 1)
  hipStream_t cs = nullptr;           // the collectives' own stream (the sync step exchanges on the context's stream)
  hipEvent_t ev_k[2] = {nullptr, nullptr}, ev_r[2] = {nullptr, nullptr}, ev_rw[2] = {nullptr, nullptr};
  hipEvent_t ev_f = nullptr, ev_g = nullptr;
  // P (or the release) of the minibatch that last used exchange slot q has been queued and recorded: the next K into
  // r_keys[q] / r_cnt[q] waits for it (P reads r_keys / r_rowid / r_rows on the main stream; K writes on the collectives')
  hipEvent_t ev_p[2] = {nullptr, nullptr};
  bool p_pending[2] = {false, false};
  struct Flight {                     // one minibatch on its way through the stages
    dfh_batch* b = nullptr;           // NULL: this rank has no minibatch in that step (it still serves its shard)
    bool described = false;           // the counts exchange is through: the sizes below are valid
    bool pulled = false;              // K, R, RW are queued: the other owners' rows are on their way to w_rows[slot]
    bool have = false;
    std::vector<size_t> send, recv, seg, off;
    size_t nrecv = 0, U = 0, active = 0;
    uint32_t own_lo = 0, own_hi = 0xFFFFFFFFu;
    bool any_own = false, any_remote = false;
    int slot = 0;
    bool listed = false;              // R left the key lists of the per-key owner side: P runs over them
  } fl[2];
  int cur = 0;                        // fl[cur]: the minibatch the next step trains; fl[cur ^ 1]: the one after
  // ---- per-stage timing (dfh_shard_set_timing): counts, L, K, R, RW, F, G, P
  bool timing = false;
  struct Span { int id; hipEvent_t a, b; };
  std::vector<Span> spans;
  std::vector<hipEvent_t> ev_pool;
  double stage_ms[DFH_SHARD_STAGES] = {0};
  uint64_t stage_steps = 0;
};

namespace {

// one part of an exchange: per peer p, send_b[p] bytes from d_send + send_off[p] and recv_b[p] bytes into
// d_recv + recv_off[p] (offsets NULL: contiguous in peer order)
struct XPart {
  const void* d_send;
  const size_t* send_b;
  const size_t* send_off;
  void* d_recv;
  const size_t* recv_b;
  const size_t* recv_off;
};

// alltoallv of device buffers; nparts parts would share ONE message group (callers pass one part: one send and
// one receive per peer and group)
int comm_exchange(dfh_comm* c, const XPart* parts, int nparts, hipStream_t on = nullptr, int kind = DFH_XCHG_OTHER) {
  hipStream_t s = on ? on : c->ctx->stream;
  const int W = c->world;
  for (int i = 0; i < nparts; ++i)
    for (int p = 0; p < W; ++p)
      if (p != c->rank) {
        c->bytes_sent += parts[i].send_b[p];
        c->bytes_recv += parts[i].recv_b[p];
      }
  ++c->groups;
  if (c->loopback) {
    // every message of the exchange as a device copy of its exact size, one launch per part
    for (int i = 0; i < nparts; ++i) {
      const XPart& x = parts[i];
      const char* fed = nullptr;
      if (!c->feed[kind].empty()) {
        fed = static_cast<const char*>(c->feed[kind].front());
        c->feed[kind].pop_front();
      } else if (c->sticky[kind]) {
        fed = static_cast<const char*>(c->sticky[kind]);
      }
      LoopSegs g;
      g.n = 0;
      size_t so = 0, ro = 0, total = 0, largest = 0;
      for (int p = 0; p < W; ++p) {
        const size_t sp = x.send_off ? x.send_off[p] : so, rp = x.recv_off ? x.recv_off[p] : ro;
        const size_t sb = x.send_b[p], rb = x.recv_b[p];
        if (p == c->rank) {  // the rank's own slot: a real self-copy
          const size_t nb = std::min(sb, rb);
          if (nb) {
            g.src[g.n] = static_cast<const char*>(x.d_send) + sp;
            g.dst[g.n] = static_cast<char*>(x.d_recv) + rp;
            g.bytes[g.n++] = nb;
          }
        } else {
          if (sb) {  // out: read and dropped
            g.src[g.n] = static_cast<const char*>(x.d_send) + sp;
            g.dst[g.n] = nullptr;
            g.bytes[g.n++] = sb;
          }
          if (rb) {  // in: from the fed source (laid out like the receive side), or an echo of what this rank sends
            g.src[g.n] = fed ? fed + rp : static_cast<const char*>(x.d_send) + sp;
            g.dst[g.n] = static_cast<char*>(x.d_recv) + rp;
            g.bytes[g.n++] = fed ? rb : std::min(sb, rb);
            if (!g.bytes[g.n - 1]) --g.n;
          }
          largest = std::max(largest, std::max(sb, rb));
        }
        total += sb + rb;
        so += sb;
        ro += rb;
      }
      if (!total) continue;
      const int blocks = (int)std::min<size_t>(1024, std::max<size_t>(1, total / (256 * 64)));
      hipLaunchKernelGGL(k_loop_copy, dim3(blocks), dim3(256), 0, s, g, c->d_stamp, reinterpret_cast<unsigned*>(c->d_stamp + 1));
      DFH_HIP(hipGetLastError());
      if (c->link_gbps > 0 && largest) {
        const double us = c->latency_us + (double)largest / (c->link_gbps * 1e3);
        c->wire_us_sum += us;
        hipLaunchKernelGGL(k_loop_wait, dim3(1), dim3(64), 0, s, c->d_stamp, (unsigned long long)(us * 1e-3 * c->wall_khz));
        DFH_HIP(hipGetLastError());
      }
    }
    return DFH_OK;
  }
  if (c->rccl) {
    RcclApi* a = rccl_api();
    bool any = false;
    for (int i = 0; i < nparts && !any; ++i)
      for (int p = 0; p < W; ++p) any = any || parts[i].send_b[p] || parts[i].recv_b[p];
    if (!any) return DFH_OK;
    DFH_RCCL(a->GroupStart());
    for (int i = 0; i < nparts; ++i) {
      const XPart& x = parts[i];
      size_t so = 0, ro = 0;
      for (int p = 0; p < W; ++p) {
        const size_t sp = x.send_off ? x.send_off[p] : so, rp = x.recv_off ? x.recv_off[p] : ro;
        if (x.send_b[p]) DFH_RCCL(a->Send(static_cast<const char*>(x.d_send) + sp, x.send_b[p], 0 /* ncclChar */, p, c->rccl, s));
        if (x.recv_b[p]) DFH_RCCL(a->Recv(static_cast<char*>(x.d_recv) + rp, x.recv_b[p], 0, p, c->rccl, s));
        so += x.send_b[p];
        ro += x.recv_b[p];
      }
    }
    DFH_RCCL(a->GroupEnd());
    return DFH_OK;
  }
  // host callback: every part staged contiguously through pinned memory
  for (int i = 0; i < nparts; ++i) {
    const XPart& x = parts[i];
    size_t st = 0, rt = 0;
    for (int p = 0; p < W; ++p) {
      st += x.send_b[p];
      rt += x.recv_b[p];
    }
    const size_t need = std::max(st, rt);
    if (need > c->h_cap) {
      if (c->h_send) DFH_HIP(hipHostFree(c->h_send));
      if (c->h_recv) DFH_HIP(hipHostFree(c->h_recv));
      c->h_send = c->h_recv = nullptr;
      c->h_cap = 0;
      const size_t cap = std::max<size_t>(need * 2, 1 << 16);
      DFH_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_send), cap, hipHostMallocDefault));
      DFH_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_recv), cap, hipHostMallocDefault));
      c->h_cap = cap;
    }
    size_t so = 0, ro = 0;
    for (int p = 0; p < W; ++p) {
      const size_t sp = x.send_off ? x.send_off[p] : so;
      if (x.send_b[p])
        DFH_HIP(hipMemcpyAsync(c->h_send + so, static_cast<const char*>(x.d_send) + sp, x.send_b[p], hipMemcpyDeviceToHost, s));
      so += x.send_b[p];
    }
    DFH_HIP(hipStreamSynchronize(s));
    if (c->fn(c->user, c->h_send, x.send_b, c->h_recv, x.recv_b) != 0) {
      set_error("dfh_comm: the host exchange callback failed");
      return DFH_ERR_HIP;
    }
    for (int p = 0; p < W; ++p) {
      const size_t rp = x.recv_off ? x.recv_off[p] : ro;
      if (x.recv_b[p])
        DFH_HIP(hipMemcpyAsync(static_cast<char*>(x.d_recv) + rp, c->h_recv + ro, x.recv_b[p], hipMemcpyHostToDevice, s));
      ro += x.recv_b[p];
    }
    // h_recv is reused by the next part: its copies must have left it
    if (i + 1 < nparts) DFH_HIP(hipStreamSynchronize(s));
  }
  return DFH_OK;
}

int comm_alltoallv(dfh_comm* c, const void* d_send, const size_t* send_b, void* d_recv, const size_t* recv_b, hipStream_t on = nullptr,
                   int kind = DFH_XCHG_OTHER) {
  XPart x{d_send, send_b, nullptr, d_recv, recv_b, nullptr};
  return comm_exchange(c, &x, 1, on, kind);
}

template <typename T>
int grow(T** p, size_t n, hipStream_t sync_stream) {
  if (*p) {
    DFH_HIP(hipStreamSynchronize(sync_stream));
    DFH_HIP(hipFree(*p));
    *p = nullptr;
  }
  DFH_HIP(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T)));
  return DFH_OK;
}

}  // namespace

extern "C" {

int dfh_comm_unique_id(void* id128) {
  DFH_ARG(id128, "dfh_comm_unique_id: NULL argument");
  RcclApi* a = rccl_api();
  if (!a) {
    set_error("RCCL is not available (librccl.so.1 could not be loaded)");
    return DFH_ERR_HIP;
  }
  RcclId id;
  DFH_RCCL(a->GetUniqueId(&id));
  memcpy(id128, id.internal, sizeof(id.internal));
  return DFH_OK;
}

int dfh_comm_create_rccl(dfh_ctx* ctx, int rank, int world, const void* id128, dfh_comm** out) {
  DFH_ARG(ctx && id128 && out && world >= 1 && world <= 32 && rank >= 0 && rank < world,
          "dfh_comm_create_rccl: bad argument (1 <= world <= 32)");
  RcclApi* a = rccl_api();
  if (!a) {
    set_error("RCCL is not available (librccl.so.1 could not be loaded)");
    return DFH_ERR_HIP;
  }
  DFH_HIP(hipSetDevice(ctx->device));
  RcclId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  void* comm = nullptr;
  DFH_RCCL(a->CommInitRank(&comm, world, id, rank));
  dfh_comm* c = new (std::nothrow) dfh_comm();
  DFH_ARG(c != nullptr, "out of host memory");
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->rccl = comm;
  *out = c;
  return DFH_OK;
}

int dfh_comm_create_callback(dfh_ctx* ctx, int rank, int world, dfh_alltoallv_fn fn, void* user, dfh_comm** out) {
  DFH_ARG(ctx && fn && out && world >= 1 && world <= 32 && rank >= 0 && rank < world,
          "dfh_comm_create_callback: bad argument (1 <= world <= 32)");
  dfh_comm* c = new (std::nothrow) dfh_comm();
  DFH_ARG(c != nullptr, "out of host memory");
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->fn = fn;
  c->user = user;
  *out = c;
  return DFH_OK;
}

int dfh_comm_create_loopback(dfh_ctx* ctx, int rank, int world, dfh_comm** out) {
  DFH_ARG(ctx && out && world >= 1 && world <= 32 && rank >= 0 && rank < world, "dfh_comm_create_loopback: bad argument (1 <= world <= 32)");
  DFH_HIP(hipSetDevice(ctx->device));
  dfh_comm* c = new (std::nothrow) dfh_comm();
  DFH_ARG(c != nullptr, "out of host memory");
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->loopback = true;
  if (hipMalloc(reinterpret_cast<void**>(&c->d_stamp), 2 * sizeof(unsigned long long)) != hipSuccess) {
    delete c;
    set_error("dfh_comm_create_loopback: hipMalloc failed");
    return DFH_ERR_HIP;
  }
  (void)hipMemset(c->d_stamp, 0, 2 * sizeof(unsigned long long));
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) == hipSuccess && khz > 0) c->wall_khz = khz;
  *out = c;
  return DFH_OK;
}

int dfh_comm_loopback_feed(dfh_comm* c, int kind, const void* d_src, int sticky) {
  DFH_ARG(c && c->loopback && kind >= 0 && kind < DFH_XCHG_KINDS, "dfh_comm_loopback_feed: a loop-back communicator and a DFH_XCHG_* kind");
  if (sticky) {
    c->sticky[kind] = d_src;
  } else {
    DFH_ARG(d_src, "dfh_comm_loopback_feed: NULL source");
    c->feed[kind].push_back(d_src);
  }
  return DFH_OK;
}

int dfh_comm_loopback_wire(dfh_comm* c, double link_gbps, double latency_us) {
  DFH_ARG(c && c->loopback && link_gbps >= 0 && latency_us >= 0, "dfh_comm_loopback_wire: a loop-back communicator, link_gbps >= 0, latency_us >= 0");
  c->link_gbps = link_gbps;
  c->latency_us = latency_us;
  return DFH_OK;
}

int dfh_comm_loopback_wire_time(dfh_comm* c, int reset, double* us) {
  DFH_ARG(c && c->loopback && us, "dfh_comm_loopback_wire_time: bad argument");
  *us = c->wire_us_sum;
  if (reset) c->wire_us_sum = 0;
  return DFH_OK;
}

int dfh_comm_destroy(dfh_comm* c) {
  if (!c) return DFH_OK;
  hipSetDevice(c->ctx->device);
  hipStreamSynchronize(c->ctx->stream);
  if (c->d_stamp) {
    hipDeviceSynchronize();  // exchanges of the collectives' stream may still read the stamp
    hipFree(c->d_stamp);
  }
  if (c->rccl) {
    RcclApi* a = rccl_api();
    if (a) a->CommDestroy(c->rccl);
  }
  if (c->h_send) hipHostFree(c->h_send);
  if (c->h_recv) hipHostFree(c->h_recv);
  delete c;
  return DFH_OK;
}

int dfh_comm_rank(dfh_comm* c) { return c ? c->rank : -1; }
int dfh_comm_world(dfh_comm* c) { return c ? c->world : 0; }

int dfh_comm_allreduce_sum(dfh_comm* c, double* vals, int n) {
  DFH_ARG(c && vals && n >= 1 && n <= 64, "dfh_comm_allreduce_sum: 1 <= n <= 64 host doubles");
  dfh_ctx* ctx = c->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int W = c->world;
  const size_t bytes = (size_t)n * sizeof(double);
  int rc = ensure_scratch(ctx, 2 * (size_t)W * bytes + 512);
  if (rc) return rc;
  Carver cv(ctx->scratch);
  double* d_s = cv.take<double>((size_t)W * n);
  double* d_r = cv.take<double>((size_t)W * n);
  std::vector<double> h((size_t)W * n);
  for (int p = 0; p < W; ++p) memcpy(&h[(size_t)p * n], vals, bytes);
  DFH_HIP(hipMemcpyAsync(d_s, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  std::vector<size_t> cnt(W, bytes);
  rc = comm_alltoallv(c, d_s, cnt.data(), d_r, cnt.data());
  if (rc) return rc;
  DFH_HIP(hipMemcpyAsync(h.data(), d_r, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int p = 0; p < W; ++p) s += h[(size_t)p * n + i];  // rank order: the same sum on every rank
    vals[i] = s;
  }
  return DFH_OK;
}

int dfh_comm_stats(dfh_comm* c, int reset, uint64_t* bytes_sent, uint64_t* bytes_recv, uint64_t* groups) {
  DFH_ARG(c, "dfh_comm_stats: NULL communicator");
  if (bytes_sent) *bytes_sent = c->bytes_sent;
  if (bytes_recv) *bytes_recv = c->bytes_recv;
  if (groups) *groups = c->groups;
  if (reset) c->bytes_sent = c->bytes_recv = c->groups = 0;
  return DFH_OK;
}

int dfh_comm_info(dfh_comm* c, char* buf, size_t n) {
  DFH_ARG(c && buf && n >= 1, "dfh_comm_info: bad argument");
  std::string t;
  if (c->rccl) {
    RcclApi* a = rccl_api();
    int v = 0;
    if (a && a->GetVersion) a->GetVersion(&v);
    t = "rccl " + std::to_string(v) + " from " + (a ? a->path : std::string("?")) +
        (a && a->was_loaded ? " (already loaded by the host process)" : " (loaded by libdifacto_hip)");
  } else if (c->loopback) {
    char w[160];
    snprintf(w, sizeof w, "loop-back transport (measurement: rank %d of %d alone on its GPU; wire model %s", c->rank, c->world,
             c->link_gbps > 0 ? "" : "off)");
    t = w;
    if (c->link_gbps > 0) {
      snprintf(w, sizeof w, "%.1f GB/s per link and direction + %.1f us per exchange)", c->link_gbps, c->latency_us);
      t += w;
    }
  } else {
    t = "host callback transport";
  }
  snprintf(buf, n, "%s", t.c_str());
  return DFH_OK;
}

// start-up self-check: every rank contributes rank + 1 to an all-to-all and must read world (world + 1) / 2 back.
// The exchange is queued and then POLLED: a peer that never joined (bad rendezvous, wrong world size, a dead rank)
// makes this return DFH_ERR_STATE after timeout_s seconds instead of hanging the job in its first step.
int dfh_comm_selfcheck(dfh_comm* c, double timeout_s) {
  DFH_ARG(c && timeout_s > 0, "dfh_comm_selfcheck: bad argument");
  if (c->loopback) return DFH_OK;  // there are no peers to hear from
  dfh_ctx* ctx = c->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int W = c->world;
  int rc = ensure_scratch(ctx, 2 * (size_t)W * sizeof(double) + 512);
  if (rc) return rc;
  Carver cv(ctx->scratch);
  double* d_s = cv.take<double>(W);
  double* d_r = cv.take<double>(W);
  std::vector<double> h(W, (double)(c->rank + 1));
  DFH_HIP(hipMemcpyAsync(d_s, h.data(), W * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  DFH_HIP(hipMemsetAsync(d_r, 0, W * sizeof(double), ctx->stream));
  std::vector<size_t> cnt(W, sizeof(double));
  rc = comm_alltoallv(c, d_s, cnt.data(), d_r, cnt.data());
  if (rc) return rc;
  DFH_HIP(hipMemcpyAsync(h.data(), d_r, W * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(ctx->stream);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) {
      set_error(std::string("dfh_comm_selfcheck: ") + hipGetErrorString(q));
      return DFH_ERR_HIP;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
      set_error("dfh_comm_selfcheck: the first exchange did not complete within " + std::to_string((int)timeout_s) +
                " s — a rank is missing or the rendezvous is wrong (rank " + std::to_string(c->rank) + " of " + std::to_string(W) + ")");
      return DFH_ERR_STATE;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
  double sum = 0;
  for (int p = 0; p < W; ++p) sum += h[p];
  if (sum != 0.5 * W * (W + 1)) {
    set_error("dfh_comm_selfcheck: the ranks' ids do not add up (got " + std::to_string(sum) + "): duplicate or missing ranks");
    return DFH_ERR_STATE;
  }
  return DFH_OK;
}

// What do the wires give?  `reps` grouped exchanges in which this rank sends bytes_per_peer to and receives bytes_per_peer
// from EVERY other rank (the shape of dfh_shard_step's K / RW / G exchanges: on a full xGMI mesh every link carries one
// message each way at once), timed with HIP events on the context's stream after two untimed ones.  *us_per_exchange: the
// average; bytes_per_peer / that = what one link gave per direction.  COLLECTIVE.  The projection of DESIGN 6a rests on an
// assumed link rate: with this the first run on a real node replaces the assumption by itself (bench.py --gpus N).
int dfh_comm_wire_probe(dfh_comm* c, size_t bytes_per_peer, int reps, double* us_per_exchange) {
  DFH_ARG(c && us_per_exchange && bytes_per_peer >= 1 && bytes_per_peer <= ((size_t)1 << 30) && reps >= 1 && reps <= 1000,
          "dfh_comm_wire_probe: 1 <= bytes_per_peer <= 1 GiB, 1 <= reps <= 1000");
  dfh_ctx* ctx = c->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int W = c->world;
  *us_per_exchange = 0;
  if (W < 2) return DFH_OK;
  char *d_s = nullptr, *d_r = nullptr;
  const size_t total = (size_t)W * bytes_per_peer;
  if (hipMalloc(reinterpret_cast<void**>(&d_s), total) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&d_r), total) != hipSuccess) {
    if (d_s) (void)hipFree(d_s);
    (void)hipGetLastError();
    set_error("dfh_comm_wire_probe: no device memory for the probe's buffers");
    return DFH_ERR_HIP;
  }
  std::vector<size_t> cnt(W, bytes_per_peer);
  cnt[c->rank] = 0;   // nothing to itself: the wires are what is measured
  const uint64_t bs = c->bytes_sent, br = c->bytes_recv, gr = c->groups;   // the probe is not part of the job's statistics
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = DFH_OK;
  auto fail = [&](hipError_t e) {
    if (e != hipSuccess && rc == DFH_OK) {
      set_error(std::string("dfh_comm_wire_probe: ") + hipGetErrorString(e));
      rc = DFH_ERR_HIP;
    }
  };
  fail(hipMemsetAsync(d_s, 1, total, ctx->stream));
  fail(hipEventCreate(&e0));
  fail(hipEventCreate(&e1));
  // (offsets spelled out: a peer's slot is W-indexed although this rank's own slot carries nothing)
  std::vector<size_t> off(W);
  for (int p = 0; p < W; ++p) off[p] = (size_t)p * bytes_per_peer;
  XPart x{d_s, cnt.data(), off.data(), d_r, cnt.data(), off.data()};
  for (int i = 0; i < 2 && rc == DFH_OK; ++i) rc = comm_exchange(c, &x, 1);
  const auto t0 = std::chrono::steady_clock::now();
  if (rc == DFH_OK) fail(hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps && rc == DFH_OK; ++i) rc = comm_exchange(c, &x, 1);
  if (rc == DFH_OK) fail(hipEventRecord(e1, ctx->stream));
  if (rc == DFH_OK) fail(hipStreamSynchronize(ctx->stream));
  if (rc == DFH_OK) {
    float ms = 0;
    fail(hipEventElapsedTime(&ms, e0, e1));
    // the host-callback transport exchanges on the host between two drains of the stream: its time is the host's
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    *us_per_exchange = (c->fn ? host_us : (double)ms * 1e3) / reps;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d_s);
  (void)hipFree(d_r);
  c->bytes_sent = bs;
  c->bytes_recv = br;
  c->groups = gr;
  return rc;
}

int dfh_comm_allgather(dfh_comm* c, const void* send, size_t bytes, void* recv) {
  DFH_ARG(c && send && recv && bytes >= 1 && bytes <= (1u << 24), "dfh_comm_allgather: 1 <= bytes <= 16 MiB of host memory");
  dfh_ctx* ctx = c->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  const int W = c->world;
  const size_t padded = (bytes + 15) & ~(size_t)15;
  int rc = ensure_scratch(ctx, (size_t)(W + 1) * padded + 512);
  if (rc) return rc;
  Carver cv(ctx->scratch);
  char* d_s = cv.take<char>(padded);
  char* d_r = cv.take<char>((size_t)W * padded);
  DFH_HIP(hipMemcpyAsync(d_s, send, bytes, hipMemcpyHostToDevice, ctx->stream));
  // the same bytes to every peer: W sends out of one buffer (offsets all zero), W receives side by side
  std::vector<size_t> cnt(W, bytes), zero(W, 0), roff(W);
  for (int p = 0; p < W; ++p) roff[p] = (size_t)p * padded;
  XPart x{d_s, cnt.data(), zero.data(), d_r, cnt.data(), roff.data()};
  rc = comm_exchange(c, &x, 1);
  if (rc) return rc;
  for (int p = 0; p < W; ++p)
    DFH_HIP(hipMemcpyAsync(static_cast<char*>(recv) + (size_t)p * bytes, d_r + roff[p], bytes, hipMemcpyDeviceToHost, ctx->stream));
  DFH_HIP(hipStreamSynchronize(ctx->stream));
  return DFH_OK;
}

int dfh_shard_balanced_splits(dfh_comm* c, const uint64_t* sample_keys, size_t n, uint64_t* splits) {
  DFH_ARG(c && splits && (n == 0 || sample_keys), "dfh_shard_balanced_splits: NULL argument");
  const int W = c->world;
  if (W == 1) return DFH_OK;
  // a fixed-size message per rank: {count, keys[S]} — S evenly spaced keys of the rank's sorted sample
  constexpr size_t S = 2048;
  std::vector<uint64_t> mine(sample_keys, sample_keys + n);
  std::sort(mine.begin(), mine.end());
  std::vector<uint64_t> msg(1 + S, 0), all((size_t)W * (1 + S));
  const size_t take = std::min(S, mine.size());
  msg[0] = take;
  for (size_t i = 0; i < take; ++i) msg[1 + i] = mine[(i * mine.size()) / take];
  int rc = dfh_comm_allgather(c, msg.data(), msg.size() * sizeof(uint64_t), all.data());
  if (rc) return rc;
  std::vector<uint64_t> uni;
  for (int p = 0; p < W; ++p) {
    const uint64_t* m = &all[(size_t)p * (1 + S)];
    DFH_ARG(m[0] <= S, "dfh_shard_balanced_splits: corrupt sample message");
    uni.insert(uni.end(), m + 1, m + 1 + m[0]);
  }
  std::sort(uni.begin(), uni.end());
  const uint64_t span = (~0ULL / (uint64_t)W) + 1;
  for (int d = 1; d < W; ++d) {
    // the first key of shard d: the d/W quantile of the union (ascending by construction); no sample at all: uniform
    splits[d - 1] = uni.empty() ? (uint64_t)d * span : uni[((size_t)d * uni.size()) / W];
  }
  return DFH_OK;
}

int dfh_shard_create(dfh_table* t, dfh_comm* c, const uint64_t* splits, dfh_shard** out) {
  DFH_ARG(t && c && out && t->ctx == c->ctx, "dfh_shard_create: table and communicator must share a context");
  if (!(t->v.p.init_mode == DFH_INIT_HASH || t->v.k == 0)) {
    set_error("the sharded store needs V_init = hash (the rand_r chain of the reference depends on the global order of allocations)");
    return DFH_ERR_STATE;
  }
  const int W = c->world;
  for (int d = 1; splits && d < W - 1; ++d) DFH_ARG(splits[d] >= splits[d - 1], "splits must be ascending");
  DFH_HIP(hipSetDevice(t->ctx->device));
  dfh_shard* s = new (std::nothrow) dfh_shard();
  DFH_ARG(s != nullptr, "out of host memory");
  s->t = t;
  s->c = c;
  hipStream_t st = t->ctx->stream;
  {
    const uint64_t span = W == 1 ? ~0ULL : (~0ULL / (uint64_t)W) + 1;
    for (int d = 1; d < W; ++d) s->h_splits.push_back(splits ? splits[d - 1] : (uint64_t)d * span);
  }
  if (splits && W > 1) {
    DFH_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_splits), (W - 1) * sizeof(uint64_t)));
    DFH_HIP(hipMemcpyAsync(s->d_splits, splits, (W - 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  }
  DFH_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_bounds), (W + 1) * sizeof(int64_t)));
  DFH_HIP(hipMalloc(reinterpret_cast<void**>(&s->d_cnt), 4 * (size_t)W * sizeof(int64_t)));
  DFH_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->h_cnt), 4 * (size_t)W * sizeof(int64_t), hipHostMallocDefault));
  DFH_HIP(hipEventCreateWithFlags(&s->cnt_ev, hipEventDisableTiming));
  DFH_HIP(hipStreamSynchronize(st));
  *out = s;
  return DFH_OK;
}

int dfh_shard_destroy(dfh_shard* s) {
  if (!s) return DFH_OK;
  hipSetDevice(s->t->ctx->device);
  sync_all(s->t->ctx);
  void* ptrs[] = {s->d_splits, s->d_bounds, s->d_cnt, s->r_keys[0], s->r_cnt[0], s->r_rowid[0], s->r_rows[0], s->w_rows[0], s->w_grads[0], s->w_grads[1],
                  s->r_keys[1], s->r_cnt[1], s->r_rowid[1], s->r_rows[1], s->w_rows[1]};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (s->h_cnt) hipHostFree(s->h_cnt);
  if (s->cnt_ev) hipEventDestroy(s->cnt_ev);
  for (hipEvent_t e : {s->ev_k[0], s->ev_k[1], s->ev_r[0], s->ev_r[1], s->ev_rw[0], s->ev_rw[1], s->ev_f, s->ev_g, s->ev_p[0], s->ev_p[1]})
    if (e) hipEventDestroy(e);
  for (auto& sp : s->spans) {
    hipEventDestroy(sp.a);
    hipEventDestroy(sp.b);
  }
  for (hipEvent_t e : s->ev_pool) hipEventDestroy(e);
  if (s->cs) {
    auto& ex = s->c->ctx->extra;
    ex.erase(std::remove(ex.begin(), ex.end(), s->cs), ex.end());
    hipStreamDestroy(s->cs);
  }
  delete s;
  return DFH_OK;
}

int dfh_shard_owned_range(dfh_shard* s, const uint64_t* splits, uint64_t* key_lo, uint64_t* key_hi) {
  DFH_ARG(s && key_lo && key_hi, "NULL argument");
  const int W = s->c->world, r = s->c->rank;
  const uint64_t span = W == 1 ? ~0ULL : (~0ULL / (uint64_t)W) + 1;
  *key_lo = r == 0 ? 0 : (splits ? splits[r - 1] : (uint64_t)r * span);
  *key_hi = r == W - 1 ? 0 : (splits ? splits[r] : (uint64_t)(r + 1) * span);  // 0: no upper bound
  return DFH_OK;
}

}  // extern "C"

namespace {
// queue, on the main stream: owner ranges of b's keys -> {keys for every owner, "I have a minibatch"} ->
// every peer -> pinned host memory
int queue_counts(dfh_shard* s, dfh_batch* b, hipStream_t on = nullptr) {
  dfh_comm* c = s->c;
  hipStream_t st = on ? on : c->ctx->stream;
  const int W = c->world;
  const bool have = b != nullptr && b->nnz > 0;
  if (have) {
    const uint64_t span = (~0ULL / (uint64_t)W) + 1;
    hipLaunchKernelGGL(k_key_ranges64, dim3((W + 256) / 256), dim3(256), 0, st, b->d_feaids, b->d_U, W, span, s->d_splits,
                       s->d_bounds, 0u);
  }
  hipLaunchKernelGGL(k_shard_counts, dim3(1), dim3(64), 0, st, have ? s->d_bounds : (const int64_t*)nullptr, W,
                     (int64_t)(b != nullptr ? 1 : 0), s->d_cnt);
  DFH_HIP(hipGetLastError());
  std::vector<size_t> cb(W, 2 * sizeof(int64_t));
  int rc = comm_alltoallv(c, s->d_cnt, cb.data(), s->d_cnt + 2 * W, cb.data(), st, DFH_XCHG_COUNTS);
  if (rc) return rc;
  DFH_HIP(hipMemcpyAsync(s->h_cnt, s->d_cnt, 4 * (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  return DFH_OK;
}

// brackets one stage of a step with two events on `st` when per-stage timing is on
struct StageScope {
  dfh_shard* s;
  int id;
  hipStream_t st;
  hipEvent_t a = nullptr, b = nullptr;
  static hipEvent_t get(dfh_shard* s) {
    if (!s->ev_pool.empty()) {
      hipEvent_t e = s->ev_pool.back();
      s->ev_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
  }
  StageScope(dfh_shard* sh, int stage, hipStream_t stream) : s(sh), id(stage), st(stream) {
    if (!s->timing) return;
    a = get(s);
    b = get(s);
    if (a && b) (void)hipEventRecord(a, st);
  }
  ~StageScope() {
    if (!a || !b) return;
    (void)hipEventRecord(b, st);
    s->spans.push_back({id, a, b});
  }
};

// sizes of a minibatch's exchange from the counts that have arrived in h_cnt
void flight_sizes(dfh_shard* s, dfh_shard::Flight& f, dfh_batch* b, int slot) {
  const int W = s->c->world, me = s->c->rank;
  f.b = b;
  f.have = b != nullptr && b->nnz > 0;
  f.send.assign(W, 0);
  f.recv.assign(W, 0);
  f.seg.assign(W + 1, 0);
  f.off.assign(W + 1, 0);
  f.active = 0;
  for (int p = 0; p < W; ++p) {
    f.send[p] = (size_t)s->h_cnt[2 * p];
    f.recv[p] = (size_t)s->h_cnt[2 * W + 2 * p];
    f.active += s->h_cnt[2 * W + 2 * p + 1] != 0 ? 1 : 0;
    f.off[p + 1] = f.off[p] + f.send[p];  // owner p's keys are ranks [off[p], off[p+1]) of the minibatch's key list
  }
  f.own_lo = (uint32_t)f.off[me];
  f.own_hi = (uint32_t)f.off[me + 1];
  f.send[me] = f.recv[me] = 0;  // the rank's own keys are not exchanged
  f.nrecv = 0;
  for (int p = 0; p < W; ++p) {
    f.nrecv += f.recv[p];
    f.seg[p + 1] = f.seg[p] + f.recv[p];
  }
  f.U = f.off[W];
  f.any_own = f.have && f.own_hi > f.own_lo;
  f.any_remote = f.have && (f.own_lo > 0 || (size_t)f.own_hi < f.U);
  f.described = true;
  f.pulled = false;
  f.slot = slot;
}

// the exchange buffers of a flight's slot (growing one waits for everything queued: rare)
int flight_bufs(dfh_shard* s, const dfh_shard::Flight& f, size_t stride) {
  const int q = f.slot;
  int rc;
  hipStream_t st = s->t->ctx->stream;
  if (f.nrecv > s->r_cap[q] || (f.any_remote && (f.U > s->w_cap[q] || f.U > s->g_cap[q]))) {
    if (s->cs) DFH_HIP(hipStreamSynchronize(s->cs));
    rc = sync_all(s->t->ctx);
    if (rc) return rc;
  }
  if (f.nrecv > s->r_cap[q]) {
    const size_t cap = f.nrecv + f.nrecv / 2 + 1024;
    if ((rc = grow(&s->r_keys[q], cap, st)) || (rc = grow(&s->r_cnt[q], cap, st)) || (rc = grow(&s->r_rowid[q], multi_words(cap, s->c->world), st)) ||
        (rc = grow(&s->r_rows[q], cap * stride, st)))
      return rc;
    s->r_cap[q] = cap;
  }
  if (f.any_remote && f.U > s->w_cap[q]) {
    const size_t cap = f.U + f.U / 2 + 1024;
    if ((rc = grow(&s->w_rows[q], cap * stride, st))) return rc;
    s->w_cap[q] = cap;
  }
  if (f.any_remote && f.U > s->g_cap[q]) {
    const size_t cap = f.U + f.U / 2 + 1024;
    if ((rc = grow(&s->w_grads[q], cap * stride, st))) return rc;
    s->g_cap[q] = cap;
  }
  return DFH_OK;
}

// ctx option "owner_per_key" (DFH_OWNER_PER_KEY=1 sets its default; measurement): the owner side per DISTINCT key in two launches (dfh_shard_count_pull_multi,
// dfh_shard_push_grad_listed) instead of per received entry in three.  Built in round 6 on VERDICT r5 #2, bit-identical
// (tests/test_gpu_parity.py "listed"), and NOT faster: 101 against 98 us for the owner side of an N = 8 step
// (profiles/r06o_owner_side_per_key.txt) — the default stays per entry.
bool owner_per_entry(const dfh_table* t) { return !t->ctx->owner_per_key; }

void flight_bytes(const dfh_shard::Flight& f, int W, size_t unit, std::vector<size_t>& sb, std::vector<size_t>& rb, std::vector<size_t>& so) {
  sb.resize(W);
  rb.resize(W);
  so.resize(W);
  for (int p = 0; p < W; ++p) {
    sb[p] = f.send[p] * unit;
    rb[p] = f.recv[p] * unit;
    so[p] = f.off[p] * unit;  // worker-side buffers are indexed by the key's rank u: owner p's slice starts at off[p]
  }
}

// K on the collectives' stream: the keys other ranks own (+ their epoch-0 counts) to their owners
int flight_K(dfh_shard* s, dfh_shard::Flight& f, int push_cnt) {
  dfh_comm* c = s->c;
  const int W = c->world;
  dfh_batch* b = f.b;
  StageScope ts(s, DFH_SHARD_STAGE_K, s->cs);
  if (b && b->ready_pending) DFH_HIP(hipStreamWaitEvent(s->cs, b->ev_ready, 0));  // its Localizer (preparation stream)
  if (s->p_pending[f.slot]) {  // the previous user of this slot's receive buffers has applied (or released) what it received
    DFH_HIP(hipStreamWaitEvent(s->cs, s->ev_p[f.slot], 0));
    s->p_pending[f.slot] = false;
  }
  std::vector<size_t> sb, rb, so;
  flight_bytes(f, W, sizeof(uint64_t), sb, rb, so);
  XPart xk{f.have ? b->d_feaids : nullptr, sb.data(), so.data(), s->r_keys[f.slot], rb.data(), nullptr};
  int rc = comm_exchange(c, &xk, 1, s->cs, DFH_XCHG_KEYS);
  if (rc) return rc;
  if (push_cnt) {
    if (f.have && !b->has_cnt) {
      hipLaunchKernelGGL(k_loc_counts, dim3(grid_for_threads(b->nnz, c->ctx)), dim3(256), 0, s->cs, b->d_col_ptr, b->d_U, b->d_feacnt);
      DFH_HIP(hipGetLastError());
      b->has_cnt = true;
    }
    flight_bytes(f, W, sizeof(float), sb, rb, so);
    XPart xc{f.have ? b->d_feacnt : nullptr, sb.data(), so.data(), s->r_cnt[f.slot], rb.data(), nullptr};
    rc = comm_exchange(c, &xc, 1, s->cs, DFH_XCHG_CNT);
    if (rc) return rc;
  }
  DFH_HIP(hipEventRecord(s->ev_k[f.slot], s->cs));
  return DFH_OK;
}

// R on the main stream (owners resolve the received keys once, count-push, pull), RW on the collectives' stream
int flight_R_RW(dfh_shard* s, dfh_shard::Flight& f, int push_cnt) {
  dfh_comm* c = s->c;
  dfh_table* t = s->t;
  const int W = c->world, q = f.slot;
  hipStream_t st = t->ctx->stream;
  const size_t stride = dfh_row_stride(t->v.k);
  int rc;
  {
    StageScope ts(s, DFH_SHARD_STAGE_R, st);
    DFH_HIP(hipStreamWaitEvent(st, s->ev_k[q], 0));
    if (f.nrecv) {
      rc = dfh_shard_resolve_multi(t, s->r_keys[q], f.seg.data(), W, q, s->r_rowid[q]);
      if (rc) return rc;
      f.listed = !owner_per_entry(t);
      if (!f.listed) {
        if (push_cnt) {
          rc = dfh_shard_push_count_multi(t, s->r_rowid[q], s->r_keys[q], f.seg.data(), W, q, s->r_cnt[q]);
          if (rc) return rc;
        }
        rc = dfh_shard_pull_resolved(t, s->r_rowid[q], f.nrecv, s->r_rows[q]);
      } else {
        rc = dfh_shard_count_pull_multi(t, s->r_rowid[q], s->r_keys[q], f.seg.data(), W, q, push_cnt ? s->r_cnt[q] : nullptr, s->r_rows[q]);
      }
      if (rc) return rc;
    }
    DFH_HIP(hipEventRecord(s->ev_r[q], st));
  }
  {
    StageScope ts(s, DFH_SHARD_STAGE_RW, s->cs);
    DFH_HIP(hipStreamWaitEvent(s->cs, s->ev_r[q], 0));
    std::vector<size_t> sb, rb, so;
    flight_bytes(f, W, stride * sizeof(float), sb, rb, so);
    XPart x{s->r_rows[q], rb.data(), nullptr, s->w_rows[q], sb.data(), so.data()};
    rc = comm_exchange(c, &x, 1, s->cs, DFH_XCHG_ROWS);
    if (rc) return rc;
    DFH_HIP(hipEventRecord(s->ev_rw[q], s->cs));
  }
  f.pulled = true;
  return DFH_OK;
}

int shard_step_overlap(dfh_shard* s, dfh_batch* b, int is_train, int push_cnt, int* any_active);
}  // namespace

extern "C" {

int dfh_shard_prefetch_counts(dfh_shard* s, dfh_batch* b_next) {
  DFH_ARG(s, "dfh_shard_prefetch_counts: NULL shard");
  DFH_ARG(!b_next || b_next->ctx == s->t->ctx, "batch and shard must share a context");
  if (b_next && !b_next->localized) {
    set_error("dfh_shard_prefetch_counts: batch is not localized (call dfh_localize first)");
    return DFH_ERR_STATE;
  }
  s->next_b = b_next;
  s->next_armed = s->c->world > 1;  // one rank: there is nothing to count
  return DFH_OK;
}

int dfh_shard_set_exchange(dfh_shard* s, int mode) {
  DFH_ARG(s && (mode == 0 || mode == 1), "dfh_shard_set_exchange: mode 0 (sync) or 1 (overlap)");
  if (s->fl[0].pulled || s->fl[1].pulled || s->counts_ready) {
    set_error("dfh_shard_set_exchange: a minibatch is under way (call it between epochs)");
    return DFH_ERR_STATE;
  }
  dfh_ctx* ctx = s->t->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  int rc = sync_all(ctx);
  if (rc) return rc;
  if (mode == 1 && !s->cs) {
    DFH_HIP(hipStreamCreateWithFlags(&s->cs, hipStreamNonBlocking));
    s->c->ctx->extra.push_back(s->cs);  // a growing table drains it before it re-allocates (table_grow)
    // these events order two streams of ONE device: no system-scope fence at the record
    const unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
    for (hipEvent_t* e : {&s->ev_k[0], &s->ev_k[1], &s->ev_r[0], &s->ev_r[1], &s->ev_rw[0], &s->ev_rw[1], &s->ev_f, &s->ev_g,
                          &s->ev_p[0], &s->ev_p[1]})
      DFH_HIP(hipEventCreateWithFlags(e, evf));
  }
  if (s->cs) DFH_HIP(hipStreamSynchronize(s->cs));
  s->exchange = mode;
  s->p_pending[0] = s->p_pending[1] = false;  // everything was drained above
  s->fl[0] = dfh_shard::Flight();
  s->fl[1] = dfh_shard::Flight();
  s->cur = 0;
  s->next_armed = false;
  return DFH_OK;
}

int dfh_shard_reserve(dfh_shard* s, size_t batch_keys, size_t recv_keys) {
  DFH_ARG(s, "dfh_shard_reserve: NULL shard");
  if (s->fl[0].pulled || s->fl[1].pulled || s->counts_ready) {
    set_error("dfh_shard_reserve: a minibatch is under way (call it before the first step or between epochs)");
    return DFH_ERR_STATE;
  }
  dfh_ctx* ctx = s->t->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  if (s->cs) DFH_HIP(hipStreamSynchronize(s->cs));
  int rc = sync_all(ctx);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  const size_t stride = dfh_row_stride(s->t->v.k);
  const int nslots = s->exchange == 1 ? 2 : 1;
  for (int q = 0; q < nslots; ++q) {
    if (recv_keys > s->r_cap[q]) {
      const size_t cap = recv_keys;
      if ((rc = grow(&s->r_keys[q], cap, st)) || (rc = grow(&s->r_cnt[q], cap, st)) ||
          (rc = grow(&s->r_rowid[q], multi_words(cap, s->c->world), st)) || (rc = grow(&s->r_rows[q], cap * stride, st)))
        return rc;
      s->r_cap[q] = cap;
    }
    if (s->c->world > 1 && batch_keys > s->w_cap[q]) {
      if ((rc = grow(&s->w_rows[q], batch_keys * stride, st))) return rc;
      s->w_cap[q] = batch_keys;
    }
    if (s->c->world > 1 && batch_keys > s->g_cap[q]) {
      if ((rc = grow(&s->w_grads[q], batch_keys * stride, st))) return rc;
      s->g_cap[q] = batch_keys;
    }
  }
  return DFH_OK;
}

int dfh_shard_set_timing(dfh_shard* s, int enable) {
  DFH_ARG(s, "dfh_shard_set_timing: NULL shard");
  s->timing = enable != 0;
  return DFH_OK;
}

int dfh_shard_get_timing(dfh_shard* s, int reset, double* ms, uint64_t* steps) {
  DFH_ARG(s && ms, "dfh_shard_get_timing: NULL argument");
  DFH_HIP(hipSetDevice(s->t->ctx->device));
  if (s->cs) DFH_HIP(hipStreamSynchronize(s->cs));
  int rc = sync_all(s->t->ctx);
  if (rc) return rc;
  for (auto& sp : s->spans) {
    float e = 0;
    if (hipEventElapsedTime(&e, sp.a, sp.b) == hipSuccess) s->stage_ms[sp.id] += e;
    s->ev_pool.push_back(sp.a);
    s->ev_pool.push_back(sp.b);
  }
  s->spans.clear();
  for (int i = 0; i < DFH_SHARD_STAGES; ++i) {
    ms[i] = s->stage_ms[i];
    if (reset) s->stage_ms[i] = 0;
  }
  if (steps) *steps = s->stage_steps;
  if (reset) s->stage_steps = 0;
  return DFH_OK;
}

int dfh_shard_step(dfh_shard* s, dfh_batch* b, int is_train, int push_cnt, int* any_active) {
  DFH_ARG(s, "dfh_shard_step: NULL shard");
  dfh_table* t = s->t;
  dfh_comm* c = s->c;
  dfh_ctx* ctx = t->ctx;
  DFH_ARG(!b || b->ctx == ctx, "batch and shard must share a context");
  if (b && !b->localized) {
    set_error("dfh_shard_step: batch is not localized (call dfh_localize first)");
    return DFH_ERR_STATE;
  }
  if (is_train) {
    if (int rca = require_aux(t, "dfh_shard_step(is_train)")) return rca;
  }
  if (s->exchange == 1 && c->world > 1) return shard_step_overlap(s, b, is_train, push_cnt, any_active);
  if (s->timing) ++s->stage_steps;
  DFH_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int W = c->world, me = c->rank;
  const int k = t->v.k, kp = t->v.kp;
  const size_t stride = dfh_row_stride(k);
  int rc;
  const bool have = b != nullptr && b->nnz > 0;
  if (b) {
    rc = main_begin(b);
    if (rc) return rc;
  }
  // ---- counts: {keys for every owner, "I have a minibatch"} -> every peer, then to the host (the step's one
  // host wait).  One rank: every key is its own, nothing to count, nothing to wait for.
  std::vector<size_t> send(W, 0), recv(W, 0), seg(W + 1, 0), off(W + 1, 0);
  size_t nrecv = 0, active = b != nullptr ? 1 : 0;
  uint32_t own_lo = 0, own_hi = 0xFFFFFFFFu;  // ranks [own_lo, own_hi) of the minibatch's unique keys are this rank's
  if (W > 1) {
    if (s->counts_ready && s->counts_for == b) {
      // exchanged inside the previous step: long arrived
      DFH_HIP(hipEventSynchronize(s->cnt_ev));
    } else {
      DFH_ARG(!s->counts_ready, "dfh_shard_step: the batch differs from the one announced to dfh_shard_prefetch_counts");
      {
        StageScope ts(s, DFH_SHARD_STAGE_COUNTS, st);
        rc = queue_counts(s, b);
        if (rc) return rc;
      }
      DFH_HIP(hipStreamSynchronize(st));
    }
    s->counts_ready = false;
    active = 0;
    for (int p = 0; p < W; ++p) {
      send[p] = (size_t)s->h_cnt[2 * p];
      recv[p] = (size_t)s->h_cnt[2 * W + 2 * p];
      active += s->h_cnt[2 * W + 2 * p + 1] != 0 ? 1 : 0;
      off[p + 1] = off[p] + send[p];   // = bounds[p + 1]: owner p's keys are ranks [off[p], off[p+1]) of the minibatch
    }
    own_lo = (uint32_t)off[me];
    own_hi = (uint32_t)off[me + 1];
    send[me] = recv[me] = 0;           // the rank's own keys are not exchanged
    for (int p = 0; p < W; ++p) {
      nrecv += recv[p];
      seg[p + 1] = seg[p] + recv[p];
    }
  }
  const size_t U = off[W];             // W > 1 only; with one rank the count stays on the device (d_U)
  const bool any_own = W == 1 ? have : (have && own_hi > own_lo);
  const bool any_remote = W > 1 && have && (own_lo > 0 || (size_t)own_hi < U);
  if (any_active) *any_active = active != 0 ? 1 : 0;
  ++s->steps;
  if (b) b->nrows_seen += (float)b->nrows;
  if (active == 0) {
    s->next_armed = false;  // the epoch is over: nothing follows
    return b ? main_end(b) : DFH_OK;
  }
  // ---- buffers
  if (nrecv > s->r_cap[0]) {
    const size_t cap = nrecv + nrecv / 2 + 1024;
    if ((rc = grow(&s->r_keys[0], cap, st)) || (rc = grow(&s->r_cnt[0], cap, st)) || (rc = grow(&s->r_rowid[0], multi_words(cap, s->c->world), st)) ||
        (rc = grow(&s->r_rows[0], cap * stride, st)))
      return rc;
    s->r_cap[0] = cap;
  }
  if (any_remote && U > s->w_cap[0]) {
    const size_t cap = U + U / 2 + 1024;
    if ((rc = grow(&s->w_rows[0], cap * stride, st))) return rc;
    s->w_cap[0] = cap;
  }
  if (any_remote && U > s->g_cap[0]) {
    const size_t cap = U + U / 2 + 1024;
    if ((rc = grow(&s->w_grads[0], cap * stride, st))) return rc;
    s->g_cap[0] = cap;
  }
  std::vector<size_t> sb(W), rb(W), so(W), sb2(W), rb2(W), so2(W);
  auto bytes = [&](size_t unit, std::vector<size_t>& sbv, std::vector<size_t>& rbv, std::vector<size_t>& sov) {
    for (int p = 0; p < W; ++p) {
      sbv[p] = send[p] * unit;
      rbv[p] = recv[p] * unit;
      sov[p] = off[p] * unit;          // worker-side buffers are indexed by the key's rank u: owner p's slice starts at off[p]
    }
  };
  const uint32_t n_own = own_hi - own_lo;  // meaningless for W == 1 (the kernels clamp to *d_U)
  // ---- L: this rank's own keys: rows + Push(kFeaCount) on its own table, {row, w} per key for the forward
  if (any_own) {
    if (int rcr = table_reserve(t, W == 1 ? b->nnz : n_own)) return rcr;
    StageScope ts(s, DFH_SHARD_STAGE_L, st);
    const bool counts = push_cnt != 0;
    hipLaunchKernelGGL(k_lookup_step, dim3(grid_for_threads(W == 1 ? b->nnz : n_own, ctx)), dim3(256), 0, st, t->v, b->d_feaids + own_lo,
                       W == 1 ? b->d_U : (const uint32_t*)nullptr, W == 1 ? 0u : n_own, b->d_urow + own_lo,
                       (counts && b->has_cnt) ? b->d_feacnt + own_lo : (const float*)nullptr, b->d_col_ptr + own_lo,
                       counts ? ((is_train && ctx->upd_kernel) ? 2 : 1) : 0, (uint32_t*)nullptr, 0, b->d_uw + own_lo, auc_pending(b),
                       shard_split_out(ctx, b, is_train, own_lo));
    b->auc_pending_n = 0;  // (the lookup's first block added up the AUC slots this batch object's previous step left)
    DFH_HIP(hipGetLastError());
  }
  // ---- K: the other keys (+ counts in epoch 0) to their owners.  Two message groups, one send and one receive
  // per peer each — the pattern every RCCL release serves (several sends to one peer inside a group are not)
  if (W > 1) {
    StageScope ts(s, DFH_SHARD_STAGE_K, st);
    bytes(sizeof(uint64_t), sb, rb, so);
    XPart xk{have ? b->d_feaids : nullptr, sb.data(), so.data(), s->r_keys[0], rb.data(), nullptr};
    rc = comm_exchange(c, &xk, 1, nullptr, DFH_XCHG_KEYS);
    if (rc) return rc;
    if (push_cnt) {
      if (have && !b->has_cnt) {
        hipLaunchKernelGGL(k_loc_counts, dim3(grid_for_threads(b->nnz, ctx)), dim3(256), 0, st, b->d_col_ptr, b->d_U, b->d_feacnt);
        DFH_HIP(hipGetLastError());
        b->has_cnt = true;
      }
      bytes(sizeof(float), sb2, rb2, so2);
      XPart xc{have ? b->d_feacnt : nullptr, sb2.data(), so2.data(), s->r_cnt[0], rb2.data(), nullptr};
      rc = comm_exchange(c, &xc, 1, nullptr, DFH_XCHG_CNT);
      if (rc) return rc;
    }
  }
  // ---- R: owners resolve once, count-push, pull (every source reads the same model version)
  const bool per_entry = owner_per_entry(t);
  if (nrecv) {
    StageScope ts(s, DFH_SHARD_STAGE_R, st);
    rc = dfh_shard_resolve_multi(t, s->r_keys[0], seg.data(), W, 0, s->r_rowid[0]);
    if (rc) return rc;
    if (per_entry) {
      if (push_cnt) {
        rc = dfh_shard_push_count_multi(t, s->r_rowid[0], s->r_keys[0], seg.data(), W, 0, s->r_cnt[0]);
        if (rc) return rc;
      }
      rc = dfh_shard_pull_resolved(t, s->r_rowid[0], nrecv, s->r_rows[0]);
    } else {
      rc = dfh_shard_count_pull_multi(t, s->r_rowid[0], s->r_keys[0], seg.data(), W, 0, push_cnt ? s->r_cnt[0] : nullptr, s->r_rows[0]);
    }
    if (rc) return rc;
  }
  // ---- RW: rows back to the workers, each owner's slice to its place among the minibatch's keys
  if (W > 1) {
    StageScope ts(s, DFH_SHARD_STAGE_RW, st);
    bytes(stride * sizeof(float), sb, rb, so);
    XPart x{s->r_rows[0], rb.data(), nullptr, s->w_rows[0], sb.data(), so.data()};
    rc = comm_exchange(c, &x, 1, nullptr, DFH_XCHG_ROWS);
    if (rc) return rc;
  }
  // ---- F: the worker's math: own keys on the table, the others on the pulled rows
  const KeyRange own{own_lo, own_hi, 0u}, others{own_lo, own_hi, 1u}, others_pen{own_lo, own_hi, 3u};
  if (b) {
    StageScope ts(s, DFH_SHARD_STAGE_F, st);
    rc = ensure_xv(b, kp);
    if (rc) return rc;
    const RowSrc tsrc = table_src(t, b->d_urow);
    if (any_remote) {
      hipLaunchKernelGGL(k_uw_remote, dim3(grid_for_threads(U, ctx)), dim3(256), 0, st, s->w_rows[0], stride, b->d_U, own_lo, own_hi,
                         b->d_uw, b->d_col_ptr, shard_split_out(ctx, b, is_train, 0u));
      DFH_HIP(hipGetLastError());
    }
    MixSrc mix{any_remote ? s->w_rows[0] + 4 : nullptr, stride};
    rc = launch_forward(b, tsrc, k, kp, b->d_uw, W > 1 ? &mix : nullptr);
    if (rc) return rc;
    // BinClassMetric::AUC of the minibatch: rides in the own keys' update launch of a training step (k_update_fused has idle
    // VALUs), a launch of its own otherwise
    // round 5: with keys of other ranks in the minibatch ONE launch of k_update_fused<MIXED> serves all keys — gradient rows for
    // the others' keys, the in-place update for the own ones (ctx option shard_mixed_update = 0: the two launches of round 4)
    const bool mixed = is_train && any_remote && ctx->upd_kernel != 0 && ctx->shard_mixed_update != 0;
    bool auc_rides = b->compute_auc && is_train && (any_own || mixed) && ctx->auc_in_update != 0 && ctx->upd_kernel != 0;
    if (b->compute_auc && !auc_rides) {
      rc = launch_auc(b);
      if (rc) return rc;
    }
    BatchView bv = batch_view(b);
    const int pgrid = have ? std::min(grid_for_waves(b->nnz, ctx), PROG_SLOTS) : 1;
    if (any_remote && !is_train) {  // EvaluatePenalty over the pulled weights (sgd_learner.cc:249-273)
      hipLaunchKernelGGL((k_penalty<1>), dim3(pgrid), dim3(256), 0, st, bv, packed_src(s->w_rows[0], k), t->v, k, kp, others);
      DFH_HIP(hipGetLastError());
    }
    if (mixed) {
      const bool with_auc = auc_rides && b->nrows <= AUC_PAIRS_MAX_N && UPD_THREADS == 256;
      rc = launch_update_fused(b, t->v, k, kp, b->d_need, b->d_uw, kAllKeys, push_cnt != 0, with_auc, s->w_rows[0], s->w_grads[0], stride);
      if (rc) return rc;
      if (auc_rides && !with_auc) {  // the minibatch is beyond the pair-counting size
        rc = launch_auc(b);
        if (rc) return rc;
      }
    } else {
      if (is_train && any_remote) {  // the gradient-row launch reads every pulled row anyway: it adds up their penalty too
        rc = launch_backward<false>(b, packed_src(s->w_rows[0], k), t->v, s->w_grads[0], stride, k, kp, nullptr, others_pen);
        if (rc) return rc;
      }
      if (is_train && any_own) {  // the fused in-place update accumulates the own keys' penalty itself
        const bool auc_wanted = auc_rides;
        rc = launch_backward<true>(b, tsrc, t->v, nullptr, 0, k, kp, b->d_need, own, b->d_uw, push_cnt != 0 && ctx->upd_kernel != 0,
                                   &auc_rides);
        if (rc) return rc;
        if (auc_wanted && !auc_rides) {  // the minibatch is beyond the pair-counting size
          rc = launch_auc(b);
          if (rc) return rc;
        }
      } else if (any_own) {
        hipLaunchKernelGGL((k_penalty<1>), dim3(pgrid), dim3(256), 0, st, bv, tsrc, t->v, k, kp, own);
        DFH_HIP(hipGetLastError());
      }
    }
  }
  // ---- the counts of the FOLLOWING step (dfh_shard_prefetch_counts): on their way to the host while this
  // step's gradients travel and are applied, so that the next call finds them there
  if (s->next_armed) {
    s->next_armed = false;
    dfh_batch* nb = s->next_b;
    if (nb) {
      rc = main_begin(nb);  // its Localizer (preparation stream) has to be through
      if (rc) return rc;
    }
    rc = queue_counts(s, nb);
    if (rc) return rc;
    DFH_HIP(hipEventRecord(s->cnt_ev, st));
    s->counts_ready = true;
    s->counts_for = nb;
  }
  // ---- G + P: gradients to the owners, applied source rank after source rank
  if (is_train && W > 1) {
    bytes(stride * sizeof(float), sb, rb, so);
    XPart x{s->w_grads[0], sb.data(), so.data(), s->r_rows[0], rb.data(), nullptr};
    {
      StageScope ts(s, DFH_SHARD_STAGE_G, st);
      rc = comm_exchange(c, &x, 1, nullptr, DFH_XCHG_GRADS);
      if (rc) return rc;
    }
    if (nrecv) {
      StageScope ts(s, DFH_SHARD_STAGE_P, st);
      rc = per_entry ? dfh_shard_push_grad_multi(t, s->r_rowid[0], s->r_keys[0], seg.data(), W, 0, s->r_rows[0])
                             : dfh_shard_push_grad_listed(t, s->r_rowid[0], s->r_keys[0], seg.data(), W, 0, s->r_rows[0]);
      if (rc) return rc;
    }
  } else if (nrecv) {
    rc = dfh_shard_release(t, s->r_rowid[0], nrecv, 0);
    if (rc) return rc;
  }
  return b ? main_end(b) : DFH_OK;
}

}  // extern "C"

namespace {
// ---- dfh_shard_step with two minibatches in flight (dfh_shard_set_exchange(s, 1)); world > 1.
// cur = the minibatch this call trains.  If it was announced to the previous call (dfh_shard_prefetch_counts) its
// counts, keys and rows were exchanged in there; otherwise (first step of an epoch) they are exchanged now.
int shard_step_overlap(dfh_shard* s, dfh_batch* b, int is_train, int push_cnt, int* any_active) {
  dfh_table* t = s->t;
  dfh_comm* c = s->c;
  dfh_ctx* ctx = t->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream, cs = s->cs;
  const int W = c->world;
  const int k = t->v.k, kp = t->v.kp;
  const size_t stride = dfh_row_stride(k);
  int rc;
  if (s->timing) ++s->stage_steps;
  static const bool trace = getenv("DFH_SHARD_TRACE") != nullptr;
  dfh_shard::Flight& cur = s->fl[s->cur];
  if (trace)
    fprintf(stderr, "[shard %d] step %llu: b=%p cur{slot %d described %d pulled %d b=%p} armed %d next_b=%p\n", c->rank,
            (unsigned long long)s->steps, (void*)b, s->cur, (int)cur.described, (int)cur.pulled, (void*)cur.b, (int)s->next_armed,
            (void*)s->next_b);
  if (!(cur.described && cur.b == b)) {
    // not announced: describe it now (its counts over the collectives' stream, one host wait)
    DFH_ARG(!cur.described || !cur.pulled, "dfh_shard_step: the batch differs from the one announced to dfh_shard_prefetch_counts");
    if (b && b->ready_pending) DFH_HIP(hipStreamWaitEvent(cs, b->ev_ready, 0));
    {
      StageScope ts(s, DFH_SHARD_STAGE_COUNTS, cs);
      rc = queue_counts(s, b, cs);
      if (rc) return rc;
    }
    DFH_HIP(hipStreamSynchronize(cs));
    flight_sizes(s, cur, b, s->cur);
  }
  if (any_active) *any_active = cur.active != 0 ? 1 : 0;
  ++s->steps;
  if (b) b->nrows_seen += (float)b->nrows;
  // the minibatch after this one, if the caller named it
  const bool look = s->next_armed;
  dfh_batch* nb = look ? s->next_b : nullptr;
  s->next_armed = false;
  dfh_shard::Flight& nxt = s->fl[s->cur ^ 1];
  nxt = dfh_shard::Flight();
  if (cur.active == 0) {
    // nobody has a minibatch: the epoch is over (nothing can have been announced behind it)
    cur = dfh_shard::Flight();
    return b ? main_end(b) : DFH_OK;
  }
  rc = flight_bufs(s, cur, stride);
  if (rc) return rc;
  if (b) {
    rc = main_begin(b);  // its Localizer (preparation stream) -> main stream
    if (rc) return rc;
  }
  // ---- L: this rank's own keys: rows + Push(kFeaCount) on its own table, {row, w} per key for the forward.  Runs
  // after everything the previous step applied: own keys are read with zero staleness.
  const uint32_t n_own = cur.own_hi - cur.own_lo;
  // the row words of the others' keys ride in this launch when their rows are already on their way (every step but an
  // epoch's first): one launch boundary less on the main stream
  const bool uw_folded = cur.any_own && cur.pulled && cur.any_remote && b != nullptr;
  if (cur.any_own) {
    if (int rcr = table_reserve(t, n_own)) return rcr;
    StageScope ts(s, DFH_SHARD_STAGE_L, st);
    const bool counts = push_cnt != 0;
    const float* cntp = (counts && b->has_cnt) ? b->d_feacnt + cur.own_lo : (const float*)nullptr;
    const int mode = counts ? ((is_train && ctx->upd_kernel) ? 2 : 1) : 0;
    if (uw_folded) {
      DFH_HIP(hipStreamWaitEvent(st, s->ev_rw[cur.slot], 0));  // the rows of the other owners have arrived
      const UwRemote m{s->w_rows[cur.slot], stride, b->d_U, cur.own_lo, cur.own_hi, b->d_uw, b->d_col_ptr, shard_split_out(ctx, b, is_train, 0u)};
      hipLaunchKernelGGL(k_lookup_uw_remote, dim3(grid_for_threads(cur.U, ctx)), dim3(256), 0, st, t->v, b->d_feaids + cur.own_lo, n_own,
                         b->d_urow + cur.own_lo, cntp, b->d_col_ptr + cur.own_lo, mode, b->d_uw + cur.own_lo, auc_pending(b), m,
                         shard_split_out(ctx, b, is_train, cur.own_lo));
    } else {
      hipLaunchKernelGGL(k_lookup_step, dim3(grid_for_threads(n_own, ctx)), dim3(256), 0, st, t->v, b->d_feaids + cur.own_lo,
                         (const uint32_t*)nullptr, n_own, b->d_urow + cur.own_lo, cntp, b->d_col_ptr + cur.own_lo, mode, (uint32_t*)nullptr,
                         0, b->d_uw + cur.own_lo, auc_pending(b), shard_split_out(ctx, b, is_train, cur.own_lo));
    }
    b->auc_pending_n = 0;
    DFH_HIP(hipGetLastError());
  }
  if (!cur.pulled) {  // pipeline fill: K, R, RW of this very minibatch, in the sync step's order (after L)
    rc = flight_K(s, cur, push_cnt);
    if (rc) return rc;
    rc = flight_R_RW(s, cur, push_cnt);
    if (rc) return rc;
  }
  // ---- counts of the next minibatch: on their way while F is queued and runs
  if (look) {
    if (nb && nb->ready_pending) DFH_HIP(hipStreamWaitEvent(cs, nb->ev_ready, 0));
    StageScope ts(s, DFH_SHARD_STAGE_COUNTS, cs);
    rc = queue_counts(s, nb, cs);
    if (rc) return rc;
    DFH_HIP(hipEventRecord(s->cnt_ev, cs));
  }
  // ---- F: the worker's math: own keys on the table, the others on the pulled rows
  const KeyRange own{cur.own_lo, cur.own_hi, 0u}, others{cur.own_lo, cur.own_hi, 1u}, others_pen{cur.own_lo, cur.own_hi, 3u};
  const int q = cur.slot;
  // BinClassMetric::AUC of the minibatch: rides in the own keys' update launch of a training step (k_update_fused has idle
  // VALUs), a launch of its own otherwise
  const bool mixed = b && is_train && cur.any_remote && ctx->upd_kernel != 0 && ctx->shard_mixed_update != 0;  // (see the sync step)
  bool auc_rides = b && b->compute_auc && is_train && (cur.any_own || mixed) && ctx->auc_in_update != 0 && ctx->upd_kernel != 0;
  if (b) {
    StageScope ts(s, DFH_SHARD_STAGE_F, st);
    rc = ensure_xv(b, kp);
    if (rc) return rc;
    if (!uw_folded) DFH_HIP(hipStreamWaitEvent(st, s->ev_rw[q], 0));  // the rows of the other owners have arrived
    const RowSrc tsrc = table_src(t, b->d_urow);
    if (cur.any_remote && !uw_folded) {
      hipLaunchKernelGGL(k_uw_remote, dim3(grid_for_threads(cur.U, ctx)), dim3(256), 0, st, s->w_rows[q], stride, b->d_U, cur.own_lo,
                         cur.own_hi, b->d_uw, b->d_col_ptr, shard_split_out(ctx, b, is_train, 0u));
      DFH_HIP(hipGetLastError());
    }
    MixSrc mix{cur.any_remote ? s->w_rows[q] + 4 : nullptr, stride};
    rc = launch_forward(b, tsrc, k, kp, b->d_uw, &mix);
    if (rc) return rc;
    if (b->compute_auc && !auc_rides) {
      rc = launch_auc(b);
      if (rc) return rc;
    }
    BatchView bv = batch_view(b);
    const int pgrid = cur.have ? std::min(grid_for_waves(b->nnz, ctx), PROG_SLOTS) : 1;
    if (cur.any_remote && !is_train) {  // EvaluatePenalty over the pulled weights (sgd_learner.cc:249-273)
      hipLaunchKernelGGL((k_penalty<1>), dim3(pgrid), dim3(256), 0, st, bv, packed_src(s->w_rows[q], k), t->v, k, kp, others);
      DFH_HIP(hipGetLastError());
    }
    if (mixed) {  // gradient rows of the others' keys AND the own keys' in-place update, one launch
      const bool with_auc = auc_rides && b->nrows <= AUC_PAIRS_MAX_N && UPD_THREADS == 256;
      rc = launch_update_fused(b, t->v, k, kp, b->d_need, b->d_uw, kAllKeys, push_cnt != 0, with_auc, s->w_rows[q], s->w_grads[q], stride);
      if (rc) return rc;
      if (auc_rides && !with_auc) {
        rc = launch_auc(b);
        if (rc) return rc;
      }
    } else if (is_train && cur.any_remote) {  // the gradient-row launch reads every pulled row anyway: it adds up their penalty too
      rc = launch_backward<false>(b, packed_src(s->w_rows[q], k), t->v, s->w_grads[q], stride, k, kp, nullptr, others_pen);
      if (rc) return rc;
    }
  }
  if (is_train) DFH_HIP(hipEventRecord(s->ev_f, st));  // the gradient rows are complete: G may start ...
  if (b && !mixed) {  // ... while the own keys are updated in place
    StageScope ts(s, DFH_SHARD_STAGE_F, st);
    const RowSrc tsrc = table_src(t, b->d_urow);
    if (is_train && cur.any_own) {  // the fused in-place update accumulates the own keys' penalty itself
      const bool auc_wanted = auc_rides;
      rc = launch_backward<true>(b, tsrc, t->v, nullptr, 0, k, kp, b->d_need, own, b->d_uw, push_cnt != 0 && ctx->upd_kernel != 0,
                                 &auc_rides);
      if (rc) return rc;
      if (auc_wanted && !auc_rides) {  // the minibatch is beyond the pair-counting size
        rc = launch_auc(b);
        if (rc) return rc;
      }
    } else if (cur.any_own) {
      BatchView bv = batch_view(b);
      const int pgrid = cur.have ? std::min(grid_for_waves(b->nnz, ctx), PROG_SLOTS) : 1;
      hipLaunchKernelGGL((k_penalty<1>), dim3(pgrid), dim3(256), 0, st, bv, tsrc, t->v, k, kp, own);
      DFH_HIP(hipGetLastError());
    }
  }
  // ---- the next minibatch's sizes (the one host wait; the device is busy with F), then its keys
  bool ahead = false;
  if (look) {
    DFH_HIP(hipEventSynchronize(s->cnt_ev));
    flight_sizes(s, nxt, nb, s->cur ^ 1);
    ahead = nxt.active != 0;
    if (trace)
      fprintf(stderr, "[shard %d]   next: active %zu U %zu nrecv %zu own [%u, %u) -> ahead %d\n", c->rank, nxt.active, nxt.U, nxt.nrecv,
              nxt.own_lo, nxt.own_hi, (int)ahead);
    if (ahead) {
      rc = flight_bufs(s, nxt, stride);
      if (rc) return rc;
      rc = flight_K(s, nxt, push_cnt);  // small; travels while F computes
      if (rc) return rc;
    }
  }
  // ---- G: gradient rows to the owners (into the buffer their rows came from)
  std::vector<size_t> sb, rb, so;
  if (is_train) {
    StageScope ts(s, DFH_SHARD_STAGE_G, cs);
    DFH_HIP(hipStreamWaitEvent(cs, s->ev_f, 0));
    flight_bytes(cur, W, stride * sizeof(float), sb, rb, so);
    XPart x{s->w_grads[q], sb.data(), so.data(), s->r_rows[q], rb.data(), nullptr};
    rc = comm_exchange(c, &x, 1, cs, DFH_XCHG_GRADS);
    if (rc) return rc;
    DFH_HIP(hipEventRecord(s->ev_g, cs));
  }
  // ---- R, RW of the next minibatch: the owners pull while this step's gradients travel — before they are applied
  // (staleness 1 for the rows of other owners) — and the rows travel while they are applied
  if (ahead) {
    rc = flight_R_RW(s, nxt, push_cnt);
    if (rc) return rc;
  }
  // ---- P: the other sources' gradients, applied source rank after source rank
  if (is_train) {
    StageScope ts(s, DFH_SHARD_STAGE_P, st);
    DFH_HIP(hipStreamWaitEvent(st, s->ev_g, 0));
    if (cur.nrecv) {
      rc = !cur.listed ? dfh_shard_push_grad_multi(t, s->r_rowid[q], s->r_keys[q], cur.seg.data(), W, q, s->r_rows[q])
                             : dfh_shard_push_grad_listed(t, s->r_rowid[q], s->r_keys[q], cur.seg.data(), W, q, s->r_rows[q]);
      if (rc) return rc;
    }
  } else if (cur.nrecv) {
    rc = dfh_shard_release(t, s->r_rowid[q], cur.nrecv, q);
    if (rc) return rc;
  }
  if (cur.nrecv) {  // r_keys[q] / r_rowid[q] / r_rows[q] are free again once this point of the main stream is reached
    DFH_HIP(hipEventRecord(s->ev_p[q], st));
    s->p_pending[q] = true;
  }
  cur = dfh_shard::Flight();
  if (look) s->cur ^= 1;  // the announced minibatch (described, maybe pulled) is the next call's
  return b ? main_end(b) : DFH_OK;
}
}  // namespace

namespace {
// ---- the literal, call-by-call Store::Pull / Push on the sharded model (dfh_shard_pull_host / dfh_shard_push_host)
struct HostCall {
  std::vector<size_t> send, recv, seg, soff;  // keys per owner / per source, prefix sums
  size_t nrecv = 0;
};

// who owns what of `keys` (ascending), and how much every rank sends this one: the W x W count matrix is gathered
int host_call_sizes(dfh_shard* s, const uint64_t* keys, size_t n, HostCall* h) {
  dfh_comm* c = s->c;
  const int W = c->world, me = c->rank;
  for (size_t i = 1; i < n; ++i) DFH_ARG(keys[i] > keys[i - 1], "keys must be strictly ascending (a Localizer's output)");
  h->send.assign(W, 0);
  h->soff.assign(W + 1, 0);
  for (int p = 0; p < W; ++p) {
    const uint64_t* e = p + 1 < W ? std::lower_bound(keys, keys + n, s->h_splits[p]) : keys + n;
    h->soff[p + 1] = (size_t)(e - keys);
    h->send[p] = h->soff[p + 1] - h->soff[p];
  }
  std::vector<uint64_t> mine(h->send.begin(), h->send.end()), all((size_t)W * W);
  int rc = dfh_comm_allgather(c, mine.data(), (size_t)W * sizeof(uint64_t), all.data());
  if (rc) return rc;
  h->recv.assign(W, 0);
  h->seg.assign(W + 1, 0);
  for (int p = 0; p < W; ++p) {
    h->recv[p] = (size_t)all[(size_t)p * W + me];
    h->seg[p + 1] = h->seg[p] + h->recv[p];
  }
  h->nrecv = h->seg[W];
  return DFH_OK;
}

int host_call_bufs(dfh_shard* s, size_t nrecv, size_t n, size_t stride) {
  hipStream_t st = s->t->ctx->stream;
  int rc;
  if (nrecv > s->r_cap[0]) {
    const size_t cap = nrecv + nrecv / 2 + 1024;
    if ((rc = grow(&s->r_keys[0], cap, st)) || (rc = grow(&s->r_cnt[0], cap, st)) || (rc = grow(&s->r_rowid[0], multi_words(cap, s->c->world), st)) ||
        (rc = grow(&s->r_rows[0], cap * stride, st)))
      return rc;
    s->r_cap[0] = cap;
  }
  if (n > s->w_cap[0]) {
    const size_t cap = n + n / 2 + 1024;
    if ((rc = grow(&s->w_rows[0], cap * stride, st))) return rc;
    s->w_cap[0] = cap;
  }
  return DFH_OK;
}

void bytes_of(const HostCall& h, int W, size_t unit, std::vector<size_t>& sb, std::vector<size_t>& rb) {
  sb.resize(W);
  rb.resize(W);
  for (int p = 0; p < W; ++p) {
    sb[p] = h.send[p] * unit;
    rb[p] = h.recv[p] * unit;
  }
}

int host_call_guard(dfh_shard* s, const char* who) {
  if (s->fl[0].pulled || s->fl[1].pulled || s->counts_ready) {
    set_error(std::string(who) + ": a minibatch of dfh_shard_step is under way");
    return DFH_ERR_STATE;
  }
  return DFH_OK;
}
}  // namespace

extern "C" {

int dfh_shard_pull_host(dfh_shard* s, const uint64_t* keys, size_t n, float* vals, size_t* nvals, int* lens, size_t* nlens) {
  DFH_ARG(s && nvals && nlens && (n == 0 || (keys && vals && lens)), "dfh_shard_pull_host: NULL argument");
  if (int g = host_call_guard(s, "dfh_shard_pull_host")) return g;
  dfh_table* t = s->t;
  dfh_comm* c = s->c;
  dfh_ctx* ctx = t->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int W = c->world, k = t->v.k;
  const size_t stride = dfh_row_stride(k);
  *nvals = 0;
  *nlens = k == 0 ? 0 : n;  // sgd_updater.cc:40
  if (int rck = check_keys(keys, n)) return rck;
  HostCall h;
  int rc = host_call_sizes(s, keys, n, &h);
  if (rc) return rc;
  rc = host_call_bufs(s, h.nrecv, n, stride);
  if (rc) return rc;
  rc = ensure_scratch(ctx, padded<uint64_t>(std::max<size_t>(n, 1)));
  if (rc) return rc;
  uint64_t* d_keys = static_cast<uint64_t*>(ctx->scratch);
  if (n) DFH_HIP(hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  std::vector<size_t> sb, rb;
  bytes_of(h, W, sizeof(uint64_t), sb, rb);
  XPart xk{d_keys, sb.data(), nullptr, s->r_keys[0], rb.data(), nullptr};
  rc = comm_exchange(c, &xk, 1);
  if (rc) return rc;
  if (h.nrecv) {  // owners: SGDUpdater::Get for every requester (sgd_updater.cc:32-56)
    rc = dfh_shard_resolve(t, s->r_keys[0], h.nrecv, s->r_rowid[0]);
    if (rc) return rc;
    rc = dfh_shard_pull_resolved(t, s->r_rowid[0], h.nrecv, s->r_rows[0]);
    if (rc) return rc;
  }
  bytes_of(h, W, stride * sizeof(float), sb, rb);
  XPart xr{s->r_rows[0], rb.data(), nullptr, s->w_rows[0], sb.data(), nullptr};
  rc = comm_exchange(c, &xr, 1);
  if (rc) return rc;
  std::vector<float> rows(std::max<size_t>(n, 1) * stride);
  if (n) DFH_HIP(hipMemcpyAsync(rows.data(), s->w_rows[0], n * stride * sizeof(float), hipMemcpyDeviceToHost, st));
  DFH_HIP(hipStreamSynchronize(st));
  rc = check_table_err(t);
  if (rc) return rc;
  size_t p = 0;
  for (size_t i = 0; i < n; ++i) {  // ragged layout of SGDUpdater::Get (sgd_updater.cc:46-53)
    const float* r = rows.data() + i * stride;
    vals[p++] = r[0];
    if (r[1] != 0.0f) {
      memcpy(vals + p, r + 4, sizeof(float) * (size_t)k);
      p += (size_t)k;
      lens[i] = k + 1;
    } else if (k != 0) {
      lens[i] = 1;
    }
  }
  *nvals = p;
  return DFH_OK;
}

int dfh_shard_push_host(dfh_shard* s, const uint64_t* keys, size_t n, int val_type, const float* vals, size_t nvals, const int* lens,
                        size_t nlens) {
  DFH_ARG(s && (n == 0 || (keys && vals)), "dfh_shard_push_host: NULL argument");
  DFH_ARG(val_type == DFH_FEA_COUNT || val_type == DFH_GRADIENT, "dfh_shard_push_host: unknown val_type (sgd_updater.cc:99)");
  if (int g = host_call_guard(s, "dfh_shard_push_host")) return g;
  dfh_table* t = s->t;
  dfh_comm* c = s->c;
  dfh_ctx* ctx = t->ctx;
  DFH_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int W = c->world, k = t->v.k;
  const size_t stride = dfh_row_stride(k);
  if (int rck = check_keys(keys, n)) return rck;
  const bool grad = val_type == DFH_GRADIENT;
  std::vector<float> rows;
  if (grad) {
    if (int rca = require_aux(t, "dfh_shard_push_host(kGradient)")) return rca;
    const bool w_only = nlens == 0;
    if (w_only) {
      DFH_ARG(nvals == n, "kGradient: CHECK_EQ(values.size(), size) (sgd_updater.cc:79)");
    } else {
      DFH_ARG(nlens == n && lens, "kGradient: CHECK_EQ(lens.size(), size) (sgd_updater.cc:81)");
    }
    rows.assign(std::max<size_t>(n, 1) * stride, 0.0f);
    size_t p = 0;
    for (size_t i = 0; i < n; ++i) {
      float* r = rows.data() + i * stride;
      DFH_ARG(p < nvals, "kGradient: values shorter than lens imply (sgd_updater.cc:96)");
      r[0] = vals[p++];
      if (!w_only && lens[i] > 1) {
        DFH_ARG(lens[i] == k + 1, "kGradient: CHECK_EQ(lens[i], V_dim+1) (sgd_updater.cc:91)");
        DFH_ARG(p + (size_t)k <= nvals, "kGradient: values shorter than lens imply (sgd_updater.cc:96)");
        r[1] = 1.0f;
        memcpy(r + 4, vals + p, sizeof(float) * (size_t)k);
        p += (size_t)k;
      }
    }
    DFH_ARG(p == nvals, "kGradient: CHECK_EQ(p, values.size()) (sgd_updater.cc:96)");
  } else {
    DFH_ARG(nvals == n, "kFeaCount: CHECK_EQ(fea_ids.size(), values.size()) (sgd_updater.cc:63)");
  }
  HostCall h;
  int rc = host_call_sizes(s, keys, n, &h);
  if (rc) return rc;
  rc = host_call_bufs(s, h.nrecv, n, stride);
  if (rc) return rc;
  rc = ensure_scratch(ctx, padded<uint64_t>(std::max<size_t>(n, 1)) + padded<float>(std::max<size_t>(n, 1)));
  if (rc) return rc;
  Carver cv(ctx->scratch);
  uint64_t* d_keys = cv.take<uint64_t>(std::max<size_t>(n, 1));
  float* d_cnt = cv.take<float>(std::max<size_t>(n, 1));
  if (n) DFH_HIP(hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  std::vector<size_t> sb, rb;
  bytes_of(h, W, sizeof(uint64_t), sb, rb);
  XPart xk{d_keys, sb.data(), nullptr, s->r_keys[0], rb.data(), nullptr};
  rc = comm_exchange(c, &xk, 1);
  if (rc) return rc;
  if (grad) {
    if (n) DFH_HIP(hipMemcpyAsync(s->w_rows[0], rows.data(), n * stride * sizeof(float), hipMemcpyHostToDevice, st));
    bytes_of(h, W, stride * sizeof(float), sb, rb);
    XPart xg{s->w_rows[0], sb.data(), nullptr, s->r_rows[0], rb.data(), nullptr};
    rc = comm_exchange(c, &xg, 1);
    if (rc) return rc;
    DFH_HIP(hipStreamSynchronize(st));  // `rows` leaves scope
  } else {
    if (n) DFH_HIP(hipMemcpyAsync(d_cnt, vals, n * sizeof(float), hipMemcpyHostToDevice, st));
    bytes_of(h, W, sizeof(float), sb, rb);
    XPart xc{d_cnt, sb.data(), nullptr, s->r_cnt[0], rb.data(), nullptr};
    rc = comm_exchange(c, &xc, 1);
    if (rc) return rc;
  }
  if (h.nrecv) {  // owners: SGDUpdater::Update, the sources one after the other in ascending rank order
    rc = dfh_shard_resolve_multi(t, s->r_keys[0], h.seg.data(), W, 0, s->r_rowid[0]);
    if (rc) return rc;
    if (grad) {
      rc = dfh_shard_push_grad_multi(t, s->r_rowid[0], s->r_keys[0], h.seg.data(), W, 0, s->r_rows[0]);
      if (rc) return rc;
    } else {
      rc = dfh_shard_push_count_multi(t, s->r_rowid[0], s->r_keys[0], h.seg.data(), W, 0, s->r_cnt[0]);
      if (rc) return rc;
      rc = dfh_shard_release(t, s->r_rowid[0], h.nrecv, 0);
      if (rc) return rc;
    }
  }
  DFH_HIP(hipStreamSynchronize(st));
  return check_table_err(t);
}

}  // extern "C"
#endif  // DFH_SHARD_HIP_
