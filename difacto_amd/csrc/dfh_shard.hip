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
};

inline RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
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
#undef DFH_SYM
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

}  // namespace dfh

using namespace dfh;

struct dfh_comm {
  dfh_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  void* rccl = nullptr;                 // ncclComm_t, or
  dfh_alltoallv_fn fn = nullptr;        // host callback
  void* user = nullptr;
  char* h_send = nullptr;               // pinned staging of the callback transport
  char* h_recv = nullptr;
  size_t h_cap = 0;
};

struct dfh_shard {
  dfh_table* t = nullptr;
  dfh_comm* c = nullptr;
  uint64_t* d_splits = nullptr;         // [world-1] first keys of shards 1.. (NULL: uniform ranges)
  int64_t* d_bounds = nullptr;          // [world+1]
  int64_t* d_cnt = nullptr;             // [2][2*world]: what I send every peer / what every peer sends me
  int64_t* h_cnt = nullptr;             // pinned copy
  // owner side, sized to the keys received in a step
  uint64_t* r_keys = nullptr;
  float* r_cnt = nullptr;
  uint32_t* r_rowid = nullptr;
  float* r_rows = nullptr;              // rows out, then gradients in
  size_t r_cap = 0;
  // worker side, sized to the batch's unique keys
  float* w_rows = nullptr;
  float* w_grads = nullptr;
  size_t w_cap = 0;
  uint64_t steps = 0;
  // counts of the FOLLOWING step, exchanged inside the current one (dfh_shard_prefetch_counts)
  dfh_batch* next_b = nullptr;
  bool next_armed = false;            // the next dfh_shard_step issues the counts exchange of next_b
  bool counts_ready = false;          // h_cnt holds the counts of counts_for; cnt_ev marks their arrival
  dfh_batch* counts_for = nullptr;
  hipEvent_t cnt_ev = nullptr;
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
int comm_exchange(dfh_comm* c, const XPart* parts, int nparts) {
  hipStream_t s = c->ctx->stream;
  const int W = c->world;
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

int comm_alltoallv(dfh_comm* c, const void* d_send, const size_t* send_b, void* d_recv, const size_t* recv_b) {
  XPart x{d_send, send_b, nullptr, d_recv, recv_b, nullptr};
  return comm_exchange(c, &x, 1);
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

int dfh_comm_destroy(dfh_comm* c) {
  if (!c) return DFH_OK;
  hipSetDevice(c->ctx->device);
  hipStreamSynchronize(c->ctx->stream);
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
  void* ptrs[] = {s->d_splits, s->d_bounds, s->d_cnt, s->r_keys, s->r_cnt, s->r_rowid, s->r_rows, s->w_rows, s->w_grads};
  for (void* p : ptrs)
    if (p) hipFree(p);
  if (s->h_cnt) hipHostFree(s->h_cnt);
  if (s->cnt_ev) hipEventDestroy(s->cnt_ev);
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
int queue_counts(dfh_shard* s, dfh_batch* b) {
  dfh_comm* c = s->c;
  hipStream_t st = c->ctx->stream;
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
  int rc = comm_alltoallv(c, s->d_cnt, cb.data(), s->d_cnt + 2 * W, cb.data());
  if (rc) return rc;
  DFH_HIP(hipMemcpyAsync(s->h_cnt, s->d_cnt, 4 * (size_t)W * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  return DFH_OK;
}
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
      rc = queue_counts(s, b);
      if (rc) return rc;
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
  if (nrecv > s->r_cap) {
    const size_t cap = nrecv + nrecv / 2 + 1024;
    if ((rc = grow(&s->r_keys, cap, st)) || (rc = grow(&s->r_cnt, cap, st)) || (rc = grow(&s->r_rowid, cap, st)) ||
        (rc = grow(&s->r_rows, cap * stride, st)))
      return rc;
    s->r_cap = cap;
  }
  if (any_remote && U > s->w_cap) {
    const size_t cap = U + U / 2 + 1024;
    if ((rc = grow(&s->w_rows, cap * stride, st)) || (rc = grow(&s->w_grads, cap * stride, st))) return rc;
    s->w_cap = cap;
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
    const bool counts = push_cnt != 0;
    hipLaunchKernelGGL(k_lookup, dim3(grid_for_threads(W == 1 ? b->nnz : n_own, ctx)), dim3(256), 0, st, t->v, b->d_feaids + own_lo,
                       W == 1 ? b->d_U : (const uint32_t*)nullptr, W == 1 ? 0u : n_own, b->d_urow + own_lo,
                       (counts && b->has_cnt) ? b->d_feacnt + own_lo : (const float*)nullptr, b->d_col_ptr + own_lo, counts ? 1 : 0,
                       (uint32_t*)nullptr, 0, b->d_uw + own_lo);
    DFH_HIP(hipGetLastError());
  }
  // ---- K: the other keys (+ counts in epoch 0) to their owners.  Two message groups, one send and one receive
  // per peer each — the pattern every RCCL release serves (several sends to one peer inside a group are not)
  if (W > 1) {
    bytes(sizeof(uint64_t), sb, rb, so);
    XPart xk{have ? b->d_feaids : nullptr, sb.data(), so.data(), s->r_keys, rb.data(), nullptr};
    rc = comm_exchange(c, &xk, 1);
    if (rc) return rc;
    if (push_cnt) {
      if (have && !b->has_cnt) {
        hipLaunchKernelGGL(k_loc_counts, dim3(grid_for_threads(b->nnz, ctx)), dim3(256), 0, st, b->d_col_ptr, b->d_U, b->d_feacnt);
        DFH_HIP(hipGetLastError());
        b->has_cnt = true;
      }
      bytes(sizeof(float), sb2, rb2, so2);
      XPart xc{have ? b->d_feacnt : nullptr, sb2.data(), so2.data(), s->r_cnt, rb2.data(), nullptr};
      rc = comm_exchange(c, &xc, 1);
      if (rc) return rc;
    }
  }
  // ---- R: owners resolve once, count-push, pull (every source reads the same model version)
  if (nrecv) {
    rc = dfh_shard_resolve_multi(t, s->r_keys, seg.data(), W, 0, s->r_rowid);
    if (rc) return rc;
    if (push_cnt) {
      rc = dfh_shard_push_count_multi(t, s->r_rowid, s->r_keys, seg.data(), W, 0, s->r_cnt);
      if (rc) return rc;
    }
    rc = dfh_shard_pull_resolved(t, s->r_rowid, nrecv, s->r_rows);
    if (rc) return rc;
  }
  // ---- RW: rows back to the workers, each owner's slice to its place among the minibatch's keys
  if (W > 1) {
    bytes(stride * sizeof(float), sb, rb, so);
    XPart x{s->r_rows, rb.data(), nullptr, s->w_rows, sb.data(), so.data()};
    rc = comm_exchange(c, &x, 1);
    if (rc) return rc;
  }
  // ---- F: the worker's math: own keys on the table, the others on the pulled rows
  const KeyRange own{own_lo, own_hi, 0u}, others{own_lo, own_hi, 1u};
  if (b) {
    rc = ensure_xv(b, kp);
    if (rc) return rc;
    const RowSrc tsrc = table_src(t, b->d_urow);
    if (any_remote) {
      hipLaunchKernelGGL(k_uw_remote, dim3(grid_for_threads(U, ctx)), dim3(256), 0, st, s->w_rows, stride, b->d_U, own_lo, own_hi,
                         b->d_uw);
      DFH_HIP(hipGetLastError());
    }
    MixSrc mix{any_remote ? s->w_rows + 4 : nullptr, stride};
    rc = launch_forward(b, tsrc, k, kp, b->d_uw, W > 1 ? &mix : nullptr);
    if (rc) return rc;
    if (b->compute_auc) {
      rc = launch_auc(b);
      if (rc) return rc;
    }
    BatchView bv = batch_view(b);
    const int pgrid = have ? std::min(grid_for_waves(b->nnz, ctx), PROG_SLOTS) : 1;
    if (any_remote) {  // EvaluatePenalty over the pulled weights (sgd_learner.cc:249-273)
      hipLaunchKernelGGL((k_penalty<1>), dim3(pgrid), dim3(256), 0, st, bv, packed_src(s->w_rows, k), t->v, k, kp, others);
      DFH_HIP(hipGetLastError());
    }
    if (is_train && any_remote) {
      TableView dummy{};
      rc = launch_backward<false>(b, packed_src(s->w_rows, k), dummy, s->w_grads, stride, k, kp, nullptr, others);
      if (rc) return rc;
    }
    if (is_train && any_own) {  // the fused in-place update accumulates the own keys' penalty itself
      rc = launch_backward<true>(b, tsrc, t->v, nullptr, 0, k, kp, b->d_need, own, b->d_uw);
      if (rc) return rc;
    } else if (any_own) {
      hipLaunchKernelGGL((k_penalty<1>), dim3(pgrid), dim3(256), 0, st, bv, tsrc, t->v, k, kp, own);
      DFH_HIP(hipGetLastError());
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
    XPart x{s->w_grads, sb.data(), so.data(), s->r_rows, rb.data(), nullptr};
    rc = comm_exchange(c, &x, 1);
    if (rc) return rc;
    if (nrecv) {
      rc = dfh_shard_push_grad_multi(t, s->r_rowid, s->r_keys, seg.data(), W, 0, s->r_rows);
      if (rc) return rc;
    }
  } else if (nrecv) {
    rc = dfh_shard_release(t, s->r_rowid, nrecv, 0);
    if (rc) return rc;
  }
  return b ? main_end(b) : DFH_OK;
}

}  // extern "C"
#endif  // DFH_SHARD_HIP_
