// C ABI of libdifacto_hip (include/difacto_hip.h): handle management, host<->device
// staging for the literal Store/Loss calls, and kernel orchestration.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

#include <hip/hip_ext.h>
#include <rocprim/rocprim.hpp>

#include "dfh_kernels.hip"
#include "dfh_localize.hip"
#include "dfh_update.hip"
#include "dfh_riders.hip"

using namespace dfh;

namespace dfh {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace dfh

// --------------------------------------------------------------------- handles
struct dfh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // preparation streams (copies, Localizer, key lookup): with pipelining on, the batches being
  // prepared take them round-robin while an earlier batch trains on `stream`
  std::vector<hipStream_t> preps;
  std::vector<hipStream_t> extra;  // other streams that carry work on this context's tables (a shard's collectives stream): drained by sync_all
  unsigned nprep = 0;       // streams in use (0: pipelining off, everything on `stream`)
  unsigned next_prep = 0;
  bool pipeline = false;
  // monotonic scratch for the literal (host-pointer) calls
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  int num_cu = 256;
  // launch tuning (dfh_ctx_set_option); the defaults are the measured optima for C3 on MI355X
  int fwd_depth = 5;           // independent V-row loads a forward lane issues before consuming any
  int fwd_blocks = 0;          // cap on the forward grid (0: one wave per example)
  int bwd_small_blocks = 2048; // cap on the short-segment blocks of the backward/update launch
  // k_update_fused (fused update on the resident table): 1 = on (default), 0 = k_backward_all's fused form;
  // caps on the blocks of its four roles
  int upd_kernel = 1;
  // (C3 on MI355X, profiles/r03_update_roles.txt: alone, hot needs >= 1024 blocks to reach 17 us, mid 1024 for 16 us,
  // few 1024 for 20 us, the singles 37 us from 1280 blocks on; together 512 / 512 / 1024, list roles first, is the best
  // found: 58.7 us; more list blocks cost more in block dispatch than they save in chain length)
  int upd_hot_blocks = 512, upd_mid_blocks = 512, upd_few_blocks = 1024, upd_single_blocks = 4096;
  int upd_interleave = 0;      // n > 1: every n-th block of the launch is a list-role block; 0 / 1: list roles first
  int owner_per_key = 0;       // sharded store, owner side per distinct key in two launches (1) or per received entry in three (0); env DFH_OWNER_PER_KEY=1 sets the default
  int upd_split = 1;           // keys with more than HOT_SPLIT occurrences part by part, a block per part (0: the hot role walks them whole; A/B)
  int auc_in_update = 1;       // a training step's AUC as the first blocks of k_update_fused (1) or a launch of its own (0)
  int shard_mixed_update = 1;  // sharded step: one k_update_fused<MIXED> launch for own + others' keys (1) or round 4's two launches (0)
  int grow_initial_rows = 1 << 20;  // first allocation of a growing table (dfh_table_create with capacity_rows = 0)
  // cross-stream events without the system-scope fence (no L2 write-back / invalidate at the record): every
  // consumer of these events is a stream of this device
  int event_flags = 1;
  int prep_priority = -1;      // preparation streams: -1 lowest, 0 default, 1 highest stream priority
  // The single-queue step (dfh_riders.hip): no preparation stream.  dfh_localize on a batch object only NOTES the work
  // (pend, in call order); the stages of the sample sort then ride, one per launch and minibatch, as extra blocks of the
  // launches later steps make on the main stream — stage s in the launch of kind rider_slot[s] (0 = the step's lookup pass,
  // 1 = forward, 2 = update); rider_alone[s]: as a launch of its own just before that launch instead (A/B).  A consumer that
  // needs a minibatch before its stages have all found a carrier runs the rest as plain launches (flush_pending).
  int single_queue = 0;
  int rider_slot[4] = {2, 0, 1, 2};     // count, scatter, sort, emit
  int rider_alone[4] = {0, 0, 0, 0};
  int rider_period[3] = {1, 1, 1};      // L, F, U: one group of 8 rider blocks every n groups of 8 blocks (1: riders first — the best measured, profiles/r06a_*)
  int rider_start[3] = {0, 0, 0};       // L, F, U: per cent of the carrier's own blocks dispatched before the first rider group
  std::vector<dfh_batch*> pend;
  // The library's code object (8 MB, a few hundred kernel instantiations) is loaded by the runtime on the FIRST launch of any
  // of its kernels: 30-40 ms that used to fall into a job's first minibatch (build/difacto: the first dfh_batch_prepare_rows
  // took 32-42 ms, profiles/r05e_e2e_startup.txt).  A helper thread makes that first launch while the caller goes on to
  // allocate its model table; joined when the context is destroyed.
  std::thread warm;

  // optional per-kernel HIP-event timing (dfh_ctx_set_timing)
  uint32_t timing = 0;  // bit i: time kernel id i
  struct Span { int id; hipEvent_t a, b; int ride = 0; };  // ride: the events travel on the dispatch itself (main stream)
  std::vector<Span> spans;
  std::vector<hipEvent_t> pool;
  double t_ms[DFH_K_COUNT] = {0};
  uint64_t t_calls[DFH_K_COUNT] = {0};
};

namespace {
// brackets the launches of one logical kernel with HIP events on the ctx stream
struct TimeScope {
  dfh_ctx* c;
  int id;
  hipStream_t st;
  hipEvent_t a = nullptr, b = nullptr;
  static hipEvent_t get(dfh_ctx* c) {
    if (!c->pool.empty()) {
      hipEvent_t e = c->pool.back();
      c->pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  }
  TimeScope(dfh_ctx* ctx, int kid, hipStream_t s = nullptr) : c(ctx), id(kid), st(s ? s : ctx->stream) {
    if (kid >= DFH_K_COUNT || !((c->timing >> kid) & 1u)) return;
    a = get(c);
    b = get(c);
    if (a && b) (void)hipEventRecord(a, st);
  }
  ~TimeScope() {
    if (!a || !b) return;
    (void)hipEventRecord(b, st);
    c->spans.push_back({id, a, b});
  }
};
}  // namespace

struct dfh_table {
  dfh_ctx* ctx = nullptr;
  TableView v{};
  uint64_t hslots = 0;
  uint64_t bytes = 0;
  bool has_aux = true;  // SGDUpdater::has_aux_ (sgd_updater.h:80): false after loading a model saved without optimiser state
  // capacity_rows = 0 at creation: the table GROWS like the reference's unordered_map (sgd_updater.h:78).  rows_bound is a
  // host-side upper bound of the rows allocated so far (every launch that may insert adds the keys it carries); when it
  // reaches the capacity the true count is read back and, if the headroom is short, the arrays are re-allocated.
  bool auto_grow = false;
  uint64_t rows_bound = 0;
  uint64_t grows = 0;
  // REFRAND scratch (need/rank/urow per pushed key) for the literal + shard calls
  uint32_t* d_need = nullptr;
  uint32_t* d_rank = nullptr;
  uint32_t* d_urow = nullptr;
  uint32_t* d_total = nullptr;
  size_t aux_cap = 0;
};

struct dfh_batch {
  dfh_ctx* ctx = nullptr;
  size_t max_rows = 0, max_nnz = 0;
  size_t nrows = 0, nnz = 0;
  bool has_value = false;
  bool has_cnt = false;
  bool localized = false;
  // raw input: own buffers (o_*), or caller-owned device memory attached without a copy
  uint64_t* d_raw = nullptr;
  uint32_t* d_offset = nullptr;
  float* d_value = nullptr;
  float* d_label = nullptr;
  void* arena = nullptr;     // the one device allocation the object's arrays are carved from (dfh_batch_create)
  void* shared_arena = nullptr;  // SharedArena*: dfh_batch_create_many
  size_t arena_bytes = 0;
  uint64_t* o_raw = nullptr;
  uint32_t* o_offset = nullptr;
  float* o_value = nullptr;
  float* o_label = nullptr;
  // pinned staging of dfh_batch_load_host (one block: offsets | labels | ids | values), allocated on first use
  char* h_stage = nullptr;
  size_t stage_bytes = 0;          // allocated size of h_stage (ensure_stage: the row-description paths need ~160 KB, load_host the whole batch)
  hipEvent_t ev_staged = nullptr;  // the copies out of h_stage (or the kernels that read it in place) have completed
  bool staged_pending = false;
  double t_prof[6] = {0, 0, 0, 0, 0, 0};  // DFH_PROFILE_PREP: host seconds inside dfh_batch_prepare_rows, by section
  uint64_t n_prof = 0;
  char* d_stage_view = nullptr;    // h_stage as the device sees it (dfh_batch_prepare_rows: the gather reads it in place)
  bool defer_ready = false;        // inside a combined preparation call: the intermediate phases do not record ev_ready
  // localizer workspace
  uint64_t *d_keys = nullptr, *d_skeys = nullptr;   // bucket-major / sorted keys
  uint32_t *d_pos = nullptr, *d_spos = nullptr;     // row of every nnz position (d_pos) / sorted positions
  uint32_t *d_bpos = nullptr;                       // bucket-major positions
  uint32_t *d_head = nullptr, *d_uid = nullptr;     // library-sort path only
  void* d_temp = nullptr;
  size_t temp_bytes = 0;
  // sample-sort localizer: bootstrap samples, persistent splitters, per-call bucket bookkeeping
  uint64_t *d_smp_key = nullptr, *d_spl_key = nullptr, *d_first_key = nullptr, *d_last_key = nullptr;
  uint32_t *d_smp_rank = nullptr, *d_smp_pos = nullptr, *d_spl_pos = nullptr, *d_packed = nullptr, *d_run_off = nullptr,
           *d_bstart = nullptr, *d_btotal = nullptr, *d_nheads = nullptr, *d_lh = nullptr;
  size_t max_tiles = 0;
  int spl_P = 0;                   // number of buckets the stored splitters partition into (0: none yet)
  // localized view
  uint64_t* d_feaids = nullptr;
  float* d_feacnt = nullptr;
  uint32_t *d_col_ptr = nullptr, *d_index = nullptr, *d_s_row = nullptr;
  float* d_s_val = nullptr;
  uint32_t* d_U = nullptr;
  // step workspace
  // long-segment key lists for the backward pass (SegLists): list buckets = sort buckets
  uint2 *d_mid = nullptr, *d_hot = nullptr, *d_few = nullptr;  // [list buckets] {cnt, off}
  SegEnt *d_mid_ent = nullptr, *d_hot_ent = nullptr, *d_few_ent = nullptr;
  // the parts of the keys with more than HOT_SPLIT_MIN occurrences (SegLists::split_ent), their partial sums and tickets
  SegEnt* d_split_ent = nullptr;
  uint32_t* d_split_ticket = nullptr;
  float* d_split_part = nullptr;
  size_t split_cap = 0;
  uint32_t seg_nb = 0;                              // list buckets of the current localized view
  uint2* d_uw = nullptr;           // {table row, w} per unique key, written by the step's k_lookup
  uint32_t *d_urow = nullptr, *d_need = nullptr, *d_rank = nullptr, *d_total = nullptr;
  float *d_pred = nullptr, *d_slope = nullptr, *d_xv = nullptr;
  size_t xv_floats = 0;
  double* d_prog = nullptr;
  float nrows_seen = 0;
  // pipelining: prep-stream work -> ev_ready -> main-stream step -> ev_free -> next prep
  hipEvent_t ev_ready = nullptr, ev_free = nullptr;
  bool ready_pending = false, free_pending = false;
  hipStream_t prep = nullptr;      // stream of the current preparation phase (load .. localize .. lookup)
  bool compute_auc = false;        // dfh_sgd_step also accumulates BinClassMetric::AUC per batch
  uint32_t *d_auc_keys = nullptr, *d_auc_skeys = nullptr, *d_auc_lab = nullptr, *d_auc_slab = nullptr;
  uint32_t* d_auc_part = nullptr;  // auc_pairs_block: one count per unit + the positives per row tile
  uint32_t auc_pending_n = 0;      // examples of the AUC whose units have been queued but not finalised (0: none)
  bool force_radix = false;        // tests: take the library-sort path of dfh_localize
  bool force_sort_fallback = false;  // tests: k_ss_sort's global-memory path for every bucket
  dfh_table* looked_up = nullptr;  // dfh_batch_lookup already resolved urow against this table
  // device feed: a minibatch DESCRIBED by dfh_batch_prepare_rows whose rows are still in their row buffers: gathered by the
  // Localizer's count pass itself (k_loc_count_gather) or, where that pass cannot (first call of a size class, the large size
  // class, the library sort, a tile spanning too many rows, more than four buffers), by k_gather_rows_staged first
  struct GatherSeg { struct dfh_rowbuf* rb; size_t at, n; };
  std::vector<GatherSeg> gsegs;
  GatherSrc gsrc{};
  bool gather_pending = false, gather_fusable = false, gather_any_value = false, loc_fuse_gather = false;
  const uint32_t *g_rows = nullptr, *g_off = nullptr;   // the description as the device sees it (mapped host memory)
  const float* g_lab = nullptr;
  // single-queue step: the sample sort of the loaded minibatch as noted by dfh_localize (arguments of its four stages) and the
  // first stage that has not been queued yet (RID_STAGES: nothing pending)
  LocView loc_v{};
  EmitOut loc_o{};
  uint32_t loc_gsort = 0;
  bool loc_big = false;
  int pend_stage = RID_STAGES;
};

namespace {
int flush_pending(dfh_batch* b);  // single-queue step: queue what is left of the batch's noted Localizer as plain launches
void pend_remove(dfh_ctx* c, dfh_batch* b);
// stream of the batch's current preparation phase
inline hipStream_t prep_of(const dfh_batch* b) {
  const dfh_ctx* c = b->ctx;
  return (c->pipeline && b->prep) ? b->prep : c->stream;
}
// a new phase (a load / attach call) takes the next preparation stream
void phase_begin(dfh_batch* b) {
  dfh_ctx* c = b->ctx;
  // a new minibatch is loaded into an object whose noted Localizer has not been queued completely: finish it first (its
  // stages leave per-object state behind — bucket totals reset by the sort, the next call's splitters written by emit)
  if (b->pend_stage < RID_STAGES) (void)flush_pending(b);
  hipStream_t next = c->pipeline ? c->preps[c->next_prep++ % c->nprep] : nullptr;
  if (next && b->prep && next != b->prep && b->ready_pending) {
    // the previous phase's output was never consumed by a step: keep the two phases ordered
    (void)hipStreamWaitEvent(next, b->ev_ready, 0);
  }
  b->prep = next;
}
// prep-stream work on a batch must not overwrite buffers a queued step still reads
int prep_begin(dfh_batch* b) {
  dfh_ctx* c = b->ctx;
  if (c->pipeline && !b->prep) phase_begin(b);
  if (c->pipeline && b->free_pending) {
    DFH_HIP(hipStreamWaitEvent(prep_of(b), b->ev_free, 0));
    b->free_pending = false;
  }
  return DFH_OK;
}
int prep_end(dfh_batch* b) {
  dfh_ctx* c = b->ctx;
  if (c->pipeline && !b->defer_ready) {
    DFH_HIP(hipEventRecord(b->ev_ready, prep_of(b)));
    b->ready_pending = true;
  }
  return DFH_OK;
}
int main_begin(dfh_batch* b) {
  dfh_ctx* c = b->ctx;
  if (b->pend_stage < RID_STAGES) {  // consumed before every stage found a carrier launch (the first steps of a loop)
    int rc = flush_pending(b);
    if (rc) return rc;
  }
  if (b->ready_pending) {
    DFH_HIP(hipStreamWaitEvent(c->stream, b->ev_ready, 0));
    b->ready_pending = false;
  }
  return DFH_OK;
}
int main_end(dfh_batch* b) {
  dfh_ctx* c = b->ctx;
  if (c->pipeline) {
    DFH_HIP(hipEventRecord(b->ev_free, c->stream));
    b->free_pending = true;
  }
  return DFH_OK;
}
int sync_all(dfh_ctx* c) {
  for (hipStream_t p : c->preps) DFH_HIP(hipStreamSynchronize(p));
  for (hipStream_t p : c->extra) DFH_HIP(hipStreamSynchronize(p));
  DFH_HIP(hipStreamSynchronize(c->stream));
  return DFH_OK;
}
}  // namespace

namespace {

inline int grid_for_threads(size_t n, const dfh_ctx* c) {
  size_t b = (n + 255) / 256;
  size_t cap = (size_t)c->num_cu * 8;
  return (int)std::max<size_t>(1, std::min(b, cap));
}
inline int grid_for_waves(size_t nwaves, const dfh_ctx* c) {
  size_t b = (nwaves + 3) / 4;
  size_t cap = (size_t)c->num_cu * 8;
  return (int)std::max<size_t>(1, std::min(b, cap));
}

int ensure_scratch(dfh_ctx* c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return DFH_OK;
  if (c->scratch) {
    DFH_HIP(hipStreamSynchronize(c->stream));
    DFH_HIP(hipFree(c->scratch));
    c->scratch = nullptr;
    c->scratch_bytes = 0;
  }
  size_t want = std::max(bytes, (size_t)1 << 20);
  want = (want + 255) & ~(size_t)255;
  DFH_HIP(hipMalloc(&c->scratch, want));
  c->scratch_bytes = want;
  return DFH_OK;
}

// carve aligned pieces out of the scratch block
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(static_cast<char*>(b)) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};
template <typename T>
size_t padded(size_t n) { return ((n * sizeof(T)) + 255 + 256) & ~(size_t)255; }

int check_table_err(dfh_table* t) {
  uint32_t e = 0;
  DFH_HIP(hipMemcpyAsync(&e, t->v.err, sizeof(e), hipMemcpyDeviceToHost, t->ctx->stream));
  DFH_HIP(hipStreamSynchronize(t->ctx->stream));
  if (e & 1u) {
    set_error("model table is full (capacity_rows exceeded)");
    return DFH_ERR_CAPACITY;
  }
  if (e & 2u) {
    set_error("a source's key list repeats a key (dfh_shard_resolve_multi: the lists must be unique, as a Localizer emits them)");
    return DFH_ERR_ARG;
  }
  if (e & 4u) {
    set_error("gradient carries V for a key whose V is not allocated (reference CHECK(e.V != nullptr))");
    return DFH_ERR_ARG;
  }
  if (e & 8u) {
    set_error("key index: a claimed slot never received its row id (find_or_insert gave up waiting)");
    return DFH_ERR_STATE;
  }
  return DFH_OK;
}

int ensure_table_aux(dfh_table* t, size_t n) {
  if (n <= t->aux_cap) return DFH_OK;
  if (t->d_need) {
    DFH_HIP(hipStreamSynchronize(t->ctx->stream));
    DFH_HIP(hipFree(t->d_need));
    DFH_HIP(hipFree(t->d_rank));
    DFH_HIP(hipFree(t->d_urow));
  }
  size_t cap = std::max<size_t>(n, 1024);
  DFH_HIP(hipMalloc(&t->d_need, cap * sizeof(uint32_t)));
  DFH_HIP(hipMalloc(&t->d_rank, cap * sizeof(uint32_t)));
  DFH_HIP(hipMalloc(&t->d_urow, cap * sizeof(uint32_t)));
  t->aux_cap = cap;
  return DFH_OK;
}

// REFRAND: initialise the rows flagged in need[0..n) in key order, advance the chain
int refrand_flush(dfh_table* t, const uint64_t* d_keys, const uint32_t* d_n, uint32_t n_static, const uint32_t* d_urow,
                  const uint32_t* d_need, uint32_t* d_rank, uint32_t* d_total) {
  hipStream_t s = t->ctx->stream;
  TimeScope ts(t->ctx, DFH_K_MISC);
  hipLaunchKernelGGL(k_refrand_scan, dim3(1), dim3(1024), 0, s, d_need, d_n, n_static, d_rank, d_total);
  hipLaunchKernelGGL(k_refrand_init, dim3(grid_for_threads(n_static, t->ctx)), dim3(256), 0, s, t->v, d_keys, d_n,
                     n_static, d_urow, d_need, d_rank);
  hipLaunchKernelGGL(k_refrand_advance, dim3(1), dim3(64), 0, s, t->v, d_total);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

inline int lanes_for(int kp) {
  int nvec = kp / 4;
  int L = 1;
  while (L < nvec) L <<= 1;
  return L;
}

// SGDUpdater::Update(kGradient) starts with CHECK(has_aux_) << "no aux data" (sgd_updater.cc:75)
int require_aux(const dfh_table* t, const char* who) {
  if (t->has_aux) return DFH_OK;
  set_error(std::string(who) + ": no aux data — the model was loaded without optimiser state (reference CHECK(has_aux_), "
                               "sgd_updater.cc:75)");
  return DFH_ERR_STATE;
}

// the largest pow2 index with load factor <= 0.5
inline uint64_t hslots_for(uint64_t capacity_rows) {
  uint64_t H = 1024;
  while (H < 2 * capacity_rows) H <<= 1;
  return H;
}

// re-allocate an auto-growing table at new_cap rows: row ids are stable (rows are handed out by a counter and never move
// inside hdr / va), so the two row arrays are copied as they are and only the key index is rebuilt.  Every stream of the
// context is drained first: no launch holds the old pointers.
int table_grow(dfh_table* t, uint64_t new_cap) {
  dfh_ctx* c = t->ctx;
  TableView& v = t->v;
  int rc = sync_all(c);
  if (rc) return rc;
  uint32_t nrows = 0;
  DFH_HIP(hipMemcpy(&nrows, v.nrows, sizeof(nrows), hipMemcpyDeviceToHost));
  nrows = std::min<uint32_t>(nrows, v.capacity);
  const uint64_t H = hslots_for(new_cap);
  const size_t ht_b = H * sizeof(HEntry), hdr_b = new_cap * sizeof(RowHdr);
  const size_t row_b = (size_t)(2 * v.kp) * sizeof(float);
  const size_t va_b = std::max<size_t>(new_cap * row_b, 256);
  HEntry* ht2 = nullptr;
  RowHdr* hdr2 = nullptr;
  float* va2 = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&ht2, ht_b)) != hipSuccess || (e = hipMalloc(&hdr2, hdr_b)) != hipSuccess || (e = hipMalloc(&va2, va_b)) != hipSuccess) {
    if (ht2) (void)hipFree(ht2);
    if (hdr2) (void)hipFree(hdr2);
    if (va2) (void)hipFree(va2);
    (void)hipGetLastError();
    set_error("model table: cannot grow to " + std::to_string(new_cap) + " rows (" + hipGetErrorString(e) +
              "); set table capacity explicitly to what fits the HBM");
    return DFH_ERR_CAPACITY;
  }
  hipStream_t s = c->stream;
  DFH_HIP(hipMemsetAsync(ht2, 0xFF, ht_b, s));
  DFH_HIP(hipMemcpyAsync(hdr2, v.hdr, (size_t)nrows * sizeof(RowHdr), hipMemcpyDeviceToDevice, s));
  DFH_HIP(hipMemsetAsync(hdr2 + nrows, 0, hdr_b - (size_t)nrows * sizeof(RowHdr), s));
  if (row_b) {
    DFH_HIP(hipMemcpyAsync(va2, v.va, (size_t)nrows * row_b, hipMemcpyDeviceToDevice, s));
    DFH_HIP(hipMemsetAsync(reinterpret_cast<char*>(va2) + (size_t)nrows * row_b, 0, va_b - (size_t)nrows * row_b, s));
  } else {
    DFH_HIP(hipMemsetAsync(va2, 0, va_b, s));
  }
  hipLaunchKernelGGL(k_rehash, dim3(grid_for_threads(t->hslots, c)), dim3(256), 0, s, v.ht, t->hslots, ht2, H - 1);
  DFH_HIP(hipGetLastError());
  DFH_HIP(hipStreamSynchronize(s));
  DFH_HIP(hipFree(v.ht));
  DFH_HIP(hipFree(v.hdr));
  DFH_HIP(hipFree(v.va));
  v.ht = ht2;
  v.hdr = hdr2;
  v.va = va2;
  v.hmask = H - 1;
  v.capacity = (uint32_t)new_cap;
  t->hslots = H;
  t->bytes = ht_b + hdr_b + va_b;
  ++t->grows;
  return DFH_OK;
}

// before a launch that may insert up to n new keys into the index.  Tables of a fixed capacity: nothing (an overflow is
// flagged by the device and reported as DFH_ERR_CAPACITY).  Growing tables: keep rows_bound + n inside the capacity.
int table_reserve(dfh_table* t, uint64_t n) {
  if (!t->auto_grow) return DFH_OK;
  TableView& v = t->v;
  if (t->rows_bound + n <= v.capacity) {
    t->rows_bound += n;
    return DFH_OK;
  }
  // the bound is spent: read the true row count (one drain of the context's streams; rare, see the headroom below)
  int rc = sync_all(t->ctx);
  if (rc) return rc;
  uint32_t nrows = 0;
  DFH_HIP(hipMemcpy(&nrows, v.nrows, sizeof(nrows), hipMemcpyDeviceToHost));
  t->rows_bound = nrows;
  // grow when fewer than 32 launches of this size fit (at most 2^24 rows of slack for a bulk load): the drain above then
  // recurs every 32 launches at worst
  const uint64_t want = t->rows_bound + n + std::min<uint64_t>(31 * n, 1ull << 24);
  uint64_t cap = v.capacity;
  while (cap < (uint64_t)kRowMask && want > cap) cap = std::min<uint64_t>(2 * cap, (uint64_t)kRowMask);
  if (cap != v.capacity) {
    rc = table_grow(t, cap);
    if (rc == DFH_ERR_CAPACITY && t->rows_bound + n <= v.capacity) rc = DFH_OK;  // no room to grow yet, but this launch still fits
    if (rc == DFH_ERR_CAPACITY) {  // try the smallest capacity that holds this launch
      const uint64_t least = std::min<uint64_t>((uint64_t)kRowMask, t->rows_bound + 2 * n);
      if (least > v.capacity && least >= t->rows_bound + n) rc = table_grow(t, least);
    }
    if (rc) return rc;
  }
  if (t->rows_bound + n > v.capacity) {
    set_error("model table is full: " + std::to_string(t->rows_bound) + " rows + " + std::to_string(n) + " keys exceed 2^28 - 1 rows");
    return DFH_ERR_CAPACITY;
  }
  t->rows_bound += n;
  return DFH_OK;
}

RowSrc table_src(const dfh_table* t, const uint32_t* urow) {
  RowSrc s;
  s.wbase = reinterpret_cast<const float*>(t->v.hdr);
  s.wstride = sizeof(RowHdr) / sizeof(float);
  s.vbase = t->v.va;
  s.vstride = (size_t)2 * t->v.kp;
  s.urow = urow;
  s.flag_is_float = 0;
  return s;
}

RowSrc packed_src(const float* rows, int V_dim) {
  RowSrc s;
  size_t stride = dfh_row_stride(V_dim);
  s.wbase = rows;
  s.wstride = stride;
  s.vbase = rows + 4;
  s.vstride = stride;
  s.urow = nullptr;
  s.flag_is_float = 1;
  return s;
}

BatchView batch_view(const dfh_batch* b) {
  BatchView v;
  v.nrows = (uint32_t)b->nrows;
  v.nnz = (uint32_t)b->nnz;
  v.d_U = b->d_U;
  v.offset = b->d_offset;
  v.index = b->d_index;
  v.value = b->has_value ? b->d_value : nullptr;
  v.label = b->d_label;
  v.feaids = b->d_feaids;
  v.col_ptr = b->d_col_ptr;
  v.s_row = b->d_s_row;
  v.s_val = b->has_value ? b->d_s_val : nullptr;
  v.urow = b->d_urow;
  v.uw = nullptr;
  v.pred = b->d_pred;
  v.slope = b->d_slope;
  v.xv = b->d_xv;
  v.prog = b->d_prog;
  v.seg.mid = b->d_mid;
  v.seg.mid_ent = b->d_mid_ent;
  v.seg.hot = b->d_hot;
  v.seg.hot_ent = b->d_hot_ent;
  v.seg.few = b->d_few;
  v.seg.few_ent = b->d_few_ent;
  v.seg.split_ent = b->d_split_ent;
  v.seg.split_n = b->d_U + 2;   // the batch's device scalar block: [0] U, [1] REFRAND total, [2] entries of the split list
  return v;
}

int ensure_xv(dfh_batch* b, int kp) {
  size_t need = b->max_rows * (size_t)std::max(kp, 4);
  if (need <= b->xv_floats) return DFH_OK;
  if (b->d_xv) {
    int rc = sync_all(b->ctx);
    if (rc) return rc;
    DFH_HIP(hipFree(b->d_xv));
  }
  DFH_HIP(hipMalloc(&b->d_xv, need * sizeof(float)));
  b->xv_floats = need;
  return DFH_OK;
}

template <typename F>
int dispatch_L(int kp, F&& f) {
  switch (lanes_for(kp)) {
    case 1: f(std::integral_constant<int, 1>()); break;
    case 2: f(std::integral_constant<int, 2>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    case 8: f(std::integral_constant<int, 8>()); break;
    case 16: f(std::integral_constant<int, 16>()); break;
    case 32: f(std::integral_constant<int, 32>()); break;
    case 64: f(std::integral_constant<int, 64>()); break;
    default:
      set_error("V_dim > 256 is not supported by the wave-per-row kernels");
      return DFH_ERR_ARG;
  }
  return DFH_OK;
}

constexpr KeyRange kAllKeys{0u, 0xFFFFFFFFu, 0u};
constexpr size_t kSmallBatchPairs = 16384;  // at or below: no probe ahead on the preparation stream (launch-bound sizes)

// (forward declared: the rider helpers live next to the Localizer's host code further down)
void collect_riders(dfh_ctx* c, int slot, bool can_ride, uint32_t main_groups, RiderSet* rs);

int launch_forward(dfh_batch* b, const RowSrc& src, int k, int kp, const uint2* uw = nullptr, const MixSrc* mix = nullptr,
                   bool riders = false) {
  BatchView bv = batch_view(b);
  bv.uw = uw;
  // one wave per example, all resident at once where possible: the kernel is
  // bound by the latency of its dependent gathers, not by launch size
  int grid = (int)std::max<size_t>(1, std::min<size_t>((b->nrows + 3) / 4, PROG_SLOTS));
  hipStream_t s = b->ctx->stream;
  const int fwd_depth = b->ctx->fwd_depth;
  if (b->ctx->fwd_blocks > 0) grid = std::min(grid, b->ctx->fwd_blocks);
  // timing: the dispatch itself carries the two events (hipExtLaunchKernelGGL), so the span is the
  // kernel's own begin/end as the command processor stamps them — what a profiler reports — and no
  // marker packet drains the stream around it
  dfh_ctx* c = b->ctx;
  hipEvent_t ea = nullptr, eb = nullptr;
  if ((c->timing >> DFH_K_FORWARD) & 1u) {
    ea = TimeScope::get(c);
    eb = TimeScope::get(c);
  }
  MixSrc mx{nullptr, 0};
  if (mix) mx = *mix;
  // single-queue step: the stages of later minibatches' Localizer that belong into a forward launch ride in this one
  RiderSet rs;
  rs.n = 0;
  if (riders && c->single_queue) collect_riders(c, 1, !mix && fwd_depth == 5, ((uint32_t)grid + 7u) / 8u, &rs);
  if (rs.n) {
    const dim3 rgrid((((unsigned)grid + 7u) / 8u + rs.ngroups) * 8u);
    const size_t shm = std::max<size_t>(rider_smem(rs), 4 * sizeof(double));
    int rcr = dispatch_L(kp, [&](auto Lc) {
      constexpr int L = decltype(Lc)::value;
      if (ea && eb) hipExtLaunchKernelGGL((k_forward_riders<L, 5>), rgrid, dim3(256), shm, s, ea, eb, 0, bv, src, k, kp, (uint32_t)grid, rs);
      else hipLaunchKernelGGL((k_forward_riders<L, 5>), rgrid, dim3(256), shm, s, bv, src, k, kp, (uint32_t)grid, rs);
    });
    if (ea && eb) c->spans.push_back({DFH_K_FORWARD, ea, eb});
    if (rcr) return rcr;
    DFH_HIP(hipGetLastError());
    return DFH_OK;
  }
  int rc = dispatch_L(kp, [&](auto Lc) {
    constexpr int L = decltype(Lc)::value;
#define DFH_FWD1(D, M)                                                                                                       \
  if (ea && eb) hipExtLaunchKernelGGL((k_forward<L, D, M>), dim3(grid), dim3(256), 0, s, ea, eb, 0, bv, src, k, kp, mx); \
  else hipLaunchKernelGGL((k_forward<L, D, M>), dim3(grid), dim3(256), 0, s, bv, src, k, kp, mx)
#define DFH_FWD(D)                      \
  if (mix) { DFH_FWD1(D, true); }       \
  else { DFH_FWD1(D, false); }
    switch (fwd_depth) {
      case 4: DFH_FWD(4); break;
      case 10: DFH_FWD(10); break;
      case 8: DFH_FWD(8); break;
      default: DFH_FWD(5); break;
    }
#undef DFH_FWD1
#undef DFH_FWD
  });
  if (ea && eb) c->spans.push_back({DFH_K_FORWARD, ea, eb});
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// an AUC whose units have run (inside an update launch) but whose slots have not been added up yet
AucFin auc_pending(dfh_batch* b) { return AucFin{b->d_auc_part, b->auc_pending_n, b->d_prog + PROG_AUC * PROG_SLOTS}; }
int auc_flush_pending(dfh_batch* b) {
  if (!b->auc_pending_n) return DFH_OK;
  hipLaunchKernelGGL(k_auc_finalize, dim3(1), dim3(256), 0, b->ctx->stream, auc_pending(b));
  b->auc_pending_n = 0;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// BinClassMetric::AUC of the batch's predictions, accumulated into the progress block
int launch_auc(dfh_batch* b) {
  hipStream_t s = b->ctx->stream;
  const uint32_t n = (uint32_t)b->nrows;
  if (n <= AUC_PAIRS_MAX_N) {
    // minibatch-sized: pair counting (auc_pairs_block units), then one block adds the units' slots up
    int rc = auc_flush_pending(b);
    if (rc) return rc;
    hipLaunchKernelGGL(k_auc_pairs, dim3(auc_units(n)), dim3(256), 0, s, b->d_pred, b->d_label, n, b->d_auc_part);
    hipLaunchKernelGGL(k_auc_finalize, dim3(1), dim3(256), 0, s, AucFin{b->d_auc_part, n, b->d_prog + PROG_AUC * PROG_SLOTS});
    DFH_HIP(hipGetLastError());
    return DFH_OK;
  }
  hipLaunchKernelGGL(k_auc_keys, dim3(grid_for_threads(n, b->ctx)), dim3(256), 0, s, b->d_pred, b->d_label, n, b->d_auc_keys,
                     b->d_auc_lab);
  size_t tb = b->temp_bytes;
  // NB: d_temp is shared with the localizer's library-sort path, which only runs on the prep stream of
  // ANOTHER batch object; within one batch the step follows its own preparation
  DFH_HIP(rocprim::radix_sort_pairs(b->d_temp, tb, b->d_auc_keys, b->d_auc_skeys, b->d_auc_lab, b->d_auc_slab, (size_t)n, 0, 32, s));
  hipLaunchKernelGGL(k_auc_area, dim3(1), dim3(1024), 0, s, b->d_auc_slab, n, b->d_prog + PROG_AUC * PROG_SLOTS);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// argument block of k_update_fused for one localized minibatch
UpdArgs upd_args(dfh_batch* b, const TableView& tv, int k, int kp, uint32_t* need, const uint2* uw, KeyRange rg, bool add_cnt) {
  dfh_ctx* c = b->ctx;
  UpdArgs a;
  a.offset = b->d_offset;
  a.index = b->d_index;
  a.value = b->has_value ? b->d_value : nullptr;
  a.uw = uw;
  a.col_ptr = b->d_col_ptr;
  a.feaids = b->d_feaids;
  a.feacnt = (add_cnt && b->has_cnt) ? b->d_feacnt : nullptr;
  a.s_row = b->d_s_row;
  a.s_val = b->has_value ? b->d_s_val : nullptr;
  a.slope = b->d_slope;
  a.xv = b->d_xv;
  a.hdr = tv.hdr;
  a.va = tv.va;
  a.need_init = need;
  a.prog = b->d_prog;
  a.seg.mid = b->d_mid;
  a.seg.mid_ent = b->d_mid_ent;
  a.seg.hot = b->d_hot;
  a.seg.hot_ent = b->d_hot_ent;
  a.seg.few = b->d_few;
  a.seg.few_ent = b->d_few_ent;
  a.seg.split_ent = b->d_split_ent;
  a.seg.split_n = b->d_U + 2;
  a.split_part = b->d_split_part;
  a.split_ticket = b->d_split_ticket;
  a.nb_split = 0;
  a.nrows = (uint32_t)b->nrows;
  a.nlist = b->seg_nb;
  a.k = k;
  a.kp = kp;
  a.rg = rg;
  a.p = tv.p;
  a.ileave = (uint32_t)c->upd_interleave;
  a.nb_hot = a.nb_mid = a.nb_few = 0;
  a.auc_pred = nullptr;
  a.auc_label = nullptr;
  a.auc_part = nullptr;
  a.nb_auc = 0;
  a.rrows = nullptr;
  a.grows = nullptr;
  a.rstride = 0;
  return a;
}

// the fused update on the resident table (k_update_fused, dfh_update.hip): needs the {row | flags, w} words this
// step's k_lookup left per unique key
// rrows / grows (sharded store): the keys of other ranks (kRemoteRow in uw) read the rows their owners sent and leave
// gradient rows, in the same launch that updates this rank's own keys in place (k_update_fused<..., MIXED>)
int launch_update_fused(dfh_batch* b, const TableView& tv, int k, int kp, uint32_t* need, const uint2* uw, KeyRange rg, bool add_cnt,
                        bool with_auc = false, const float* rrows = nullptr, float* grows = nullptr, size_t rstride = 0,
                        bool riders = false) {
  dfh_ctx* c = b->ctx;
  hipStream_t s = c->stream;
  const int L = lanes_for(kp);
  UpdArgs a = upd_args(b, tv, k, kp, need, uw, rg, add_cnt);
  const bool mixed = rrows != nullptr && grows != nullptr;
  a.rrows = rrows;
  a.grows = grows;
  a.rstride = (uint32_t)rstride;
  if (with_auc) {  // the minibatch's AUC rides in this launch (dfh_sgd_step, compute_auc)
    a.auc_pred = b->d_pred;
    a.auc_label = b->d_label;
    int rcf = auc_flush_pending(b);  // the slots are about to be overwritten (normally closed by this step's k_lookup already)
    if (rcf) return rcf;
    a.auc_part = b->d_auc_part;
    a.nb_auc = (auc_units((uint32_t)b->nrows) + 7u) & ~7u;  // a multiple of 8: the roles behind keep their XCDs
    b->auc_pending_n = (uint32_t)b->nrows;  // finalised by the next launch that carries it (k_lookup of the next step, dfh_batch_progress)
  }
  // blocks per role (U and the list sizes live on the device; nnz bounds them): surplus blocks find their
  // list exhausted and leave at once
  const size_t nnz = b->nnz, G = 64 / (size_t)L;
  a.nb_hot = (uint32_t)std::max<size_t>(1, std::min<size_t>(nnz / (BWD_MID + 1) + 1, (size_t)c->upd_hot_blocks));
  a.nb_mid = (uint32_t)std::max<size_t>(1, std::min<size_t>((nnz / (BWD_SMALL + 1)) / UPD_NW + 1, (size_t)c->upd_mid_blocks));
  a.nb_few = (uint32_t)std::max<size_t>(1, std::min<size_t>((nnz / 2) / (UPD_NW * G) + 1, (size_t)c->upd_few_blocks));
  // singles block b on the XCD that ran the forward's block b (workgroups go round-robin over the 8 XCDs; the XV rows and
  // slopes of a block's four examples sit in that XCD's L2): the blocks before it add up to a multiple of 8.  (Measured in
  // round 4 by shifting the role 1 or 4 blocks: no difference, 86.0 against 85.9 M examples/sec — kept because it is free.)
  a.nb_few += (8u - (a.nb_hot + a.nb_mid + a.nb_few) % 8u) % 8u;
  // the parts of keys with more than HOT_SPLIT_MIN occurrences: taken by the hot role's blocks before their own lists
  a.nb_split = (c->upd_split && nnz > HOT_SPLIT_MIN) ? 1u : 0u;
  const size_t nb_single = std::max<size_t>(1, std::min<size_t>((b->nrows + UPD_NW - 1) / UPD_NW, (size_t)c->upd_single_blocks));
  hipEvent_t ea = nullptr, eb = nullptr;  // timing rides on the dispatch, like the forward's
  if ((c->timing >> DFH_K_BACKWARD) & 1u) {
    ea = TimeScope::get(c);
    eb = TimeScope::get(c);
  }
  const dim3 grid((unsigned)(a.nb_auc + a.nb_hot + a.nb_mid + a.nb_few + nb_single)), block(UPD_THREADS);
  // single-queue step: the stages of later minibatches' Localizer that belong into an update launch ride in this one
  RiderSet rs;
  rs.n = 0;
  if (riders && c->single_queue) collect_riders(c, 2, !mixed && UPD_THREADS == RID_THREADS, (grid.x + 7u) / 8u, &rs);
  if (rs.n) {
    const dim3 rgrid(((grid.x + 7u) / 8u + rs.ngroups) * 8u);
    const size_t shm = std::max<size_t>(rider_smem(rs), UPD_SMEM);
    int rcr = dispatch_L(kp, [&](auto Lc) {
      constexpr int LL = decltype(Lc)::value;
#define DFH_UPDR(EXACT, HV)                                                                                                 \
  if (ea && eb) hipExtLaunchKernelGGL((k_update_fused_riders<LL, EXACT, HV>), rgrid, block, shm, s, ea, eb, 0, a, grid.x, rs); \
  else hipLaunchKernelGGL((k_update_fused_riders<LL, EXACT, HV>), rgrid, block, shm, s, a, grid.x, rs)
      if (kp == 4 * LL) {
        if (b->has_value) { DFH_UPDR(true, true); } else { DFH_UPDR(true, false); }
      } else {
        if (b->has_value) { DFH_UPDR(false, true); } else { DFH_UPDR(false, false); }
      }
#undef DFH_UPDR
    });
    if (ea && eb) c->spans.push_back({DFH_K_BACKWARD, ea, eb});
    if (rcr) return rcr;
    DFH_HIP(hipGetLastError());
    return DFH_OK;
  }
  int rc = dispatch_L(kp, [&](auto Lc) {
    constexpr int LL = decltype(Lc)::value;
#define DFH_UPD(EXACT, HV)                                                                                         \
  if (mixed) {                                                                                                     \
    if (ea && eb) hipExtLaunchKernelGGL((k_update_fused<LL, EXACT, HV, true>), grid, block, 0, s, ea, eb, 0, a);    \
    else hipLaunchKernelGGL((k_update_fused<LL, EXACT, HV, true>), grid, block, 0, s, a);                           \
  } else {                                                                                                         \
    if (ea && eb) hipExtLaunchKernelGGL((k_update_fused<LL, EXACT, HV, false>), grid, block, 0, s, ea, eb, 0, a);   \
    else hipLaunchKernelGGL((k_update_fused<LL, EXACT, HV, false>), grid, block, 0, s, a);                          \
  }
    if (kp == 4 * LL) {
      if (b->has_value) { DFH_UPD(true, true); } else { DFH_UPD(true, false); }
    } else {
      if (b->has_value) { DFH_UPD(false, true); } else { DFH_UPD(false, false); }
    }
#undef DFH_UPD
  });
  if (ea && eb) c->spans.push_back({DFH_K_BACKWARD, ea, eb});
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

template <bool FUSED>
int launch_backward(dfh_batch* b, const RowSrc& src, const TableView& tv, float* grads, size_t gstride, int k, int kp,
                    uint32_t* need, KeyRange rg = kAllKeys, const uint2* uw = nullptr, bool add_cnt = false, bool* auc_rides = nullptr,
                    bool riders = false) {
  if (FUSED && src.urow && uw && b->ctx->upd_kernel) {
    // auc_rides: in: the caller wants the minibatch's AUC; out: this launch computed it
    const bool with_auc = auc_rides && *auc_rides && b->nrows <= AUC_PAIRS_MAX_N && UPD_THREADS == 256;
    if (auc_rides) *auc_rides = with_auc;
    return launch_update_fused(b, tv, k, kp, need, uw, rg, add_cnt, with_auc, nullptr, nullptr, 0, riders);
  }
  if (auc_rides) *auc_rides = false;
  if (riders) {  // no rider form of k_backward_all: the stages due in an update launch run alone
    RiderSet none;
    collect_riders(b->ctx, 2, false, 0, &none);
  }
  BatchView bv = batch_view(b);
  hipStream_t s = b->ctx->stream;
  const int L = lanes_for(kp);
  dfh_ctx* c = b->ctx;
  // U and the list sizes live on the device; nnz bounds them.  One launch of BWD_THREADS blocks:
  //   [0, nb_hot)            one hot key (> BWD_MID occurrences) per block and iteration
  //   [nb_hot, +nb_mid)      one mid key per wave and iteration
  //   the rest               short segments: (64/L) keys per wave and iteration, grid-stride
  // the long chains start first; surplus blocks find their list exhausted and leave at once
  constexpr size_t NWB = BWD_THREADS / 64;
  const size_t nb_hot = std::max<size_t>(1, std::min<size_t>(b->nnz / (BWD_MID + 1) + 1, 256));
#ifndef DFH_BWD_MID_BLOCKS
#define DFH_BWD_MID_BLOCKS 512
#endif
  const size_t nb_mid = std::max<size_t>(1, std::min<size_t>((b->nnz / (BWD_SMALL + 1)) / NWB + 1, DFH_BWD_MID_BLOCKS));
  const size_t small_cap = (size_t)c->bwd_small_blocks;
  const size_t keys_per_block = NWB * (64 / L);
  const size_t nb_small = std::max<size_t>(1, std::min<size_t>((b->nnz + keys_per_block - 1) / keys_per_block, small_cap));
  hipEvent_t ea = nullptr, eb = nullptr;  // timing rides on the dispatch, like the forward's
  if ((c->timing >> DFH_K_BACKWARD) & 1u) {
    ea = TimeScope::get(c);
    eb = TimeScope::get(c);
  }
  const dim3 grid((unsigned)(nb_hot + nb_mid + nb_small)), block(BWD_THREADS);
  const uint32_t nh = (uint32_t)nb_hot, nm = (uint32_t)nb_mid, nlist = b->seg_nb;
  int rc = dispatch_L(kp, [&](auto Lc) {
    constexpr int LL = decltype(Lc)::value;
    const bool lean = FUSED && src.urow;
#define DFH_BWD(LEAN, EXACT)                                                                                          \
  if (ea && eb)                                                                                                       \
    hipExtLaunchKernelGGL((k_backward_all<LL, FUSED, LEAN, EXACT>), grid, block, 0, s, ea, eb, 0, bv, src, tv, grads, \
                          gstride, k, kp, need, nh, nm, nlist, rg);                                                   \
  else                                                                                                                \
    hipLaunchKernelGGL((k_backward_all<LL, FUSED, LEAN, EXACT>), grid, block, 0, s, bv, src, tv, grads, gstride, k, kp, \
                       need, nh, nm, nlist, rg)
    if (lean && kp == 4 * LL) {
      DFH_BWD(FUSED, true);
    } else if (lean) {
      DFH_BWD(FUSED, false);
    } else {
      DFH_BWD(false, false);
    }
#undef DFH_BWD
  });
  if (ea && eb) c->spans.push_back({DFH_K_BACKWARD, ea, eb});
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

}  // namespace

// ===================================================================== C ABI
extern "C" {

const char* dfh_last_error(void) { return g_err.c_str(); }

void dfh_updater_param_default(dfh_updater_param* p, int V_dim) {
  // src/sgd/sgd_param.h:95-105
  p->l1 = 1.0f;
  p->l2 = 0.0f;
  p->V_l2 = 0.01f;
  p->lr = 0.01f;
  p->lr_beta = 1.0f;
  p->V_lr = 0.01f;
  p->V_lr_beta = 1.0f;
  p->V_init_scale = 0.01f;
  p->V_threshold = 10;
  p->V_dim = V_dim;
  p->seed = 0;
  p->init_mode = DFH_INIT_HASH;
}

uint64_t dfh_reverse_bytes(uint64_t x) { return reverse_bytes(x); }
uint64_t dfh_encode_fea_grp_id(uint64_t x, int gid, int nbits) { return (x << nbits) | (uint64_t)gid; }
size_t dfh_row_stride(int V_dim) { return 4 + (size_t)((V_dim + 3) / 4) * 4; }

// ------------------------------------------------------------------ context
int dfh_ctx_create(int device, void* stream, dfh_ctx** out) {
  DFH_ARG(out != nullptr, "dfh_ctx_create: out is NULL");
  int ndev = 0;
  DFH_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0 || device < 0 || device >= ndev) {
    set_error("dfh_ctx_create: no such HIP device (this library has no CPU fallback)");
    return DFH_ERR_HIP;
  }
  DFH_HIP(hipSetDevice(device));
  // DFH_SCHEDULE = spin | yield | blocking (experiment, tools/gpu_r04u.sh): how host threads wait for the device
  if (const char* e = getenv("DFH_SCHEDULE")) {
    const unsigned f = !strcmp(e, "spin") ? hipDeviceScheduleSpin : !strcmp(e, "yield") ? hipDeviceScheduleYield
                       : !strcmp(e, "blocking") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
    if (hipSetDeviceFlags(f) != hipSuccess) (void)hipGetLastError();
  }
  dfh_ctx* c = new (std::nothrow) dfh_ctx();
  DFH_ARG(c != nullptr, "out of host memory");
  c->device = device;
  if (const char* e = getenv("DFH_OWNER_PER_KEY")) c->owner_per_key = atoi(e) != 0;  // (measurement: bench.py --emulate-world)
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
      return DFH_ERR_HIP;
    }
    c->own_stream = true;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  if (!(getenv("DFH_WARM_LOAD") && atoi(getenv("DFH_WARM_LOAD")) == 0)) {   // (0: A/B)
    try {
    c->warm = std::thread([device] {
      if (hipSetDevice(device) != hipSuccess) return;
      hipStream_t s = nullptr;
      uint32_t* d = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&d), 256) == hipSuccess) {
        hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, s, d, 0u);
        (void)hipStreamSynchronize(s);
      }
      (void)hipGetLastError();
      if (d) (void)hipFree(d);
      if (s) (void)hipStreamDestroy(s);
    });
    } catch (...) {  // no thread to be had: the first launch loads the code object, as before round 5 (nothing crosses the C ABI)
    }
  }
  *out = c;
  return DFH_OK;
}

int dfh_ctx_destroy(dfh_ctx* c) {
  if (!c) return DFH_OK;
  if (c->warm.joinable()) c->warm.join();
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->scratch) hipFree(c->scratch);
  for (auto& sp : c->spans) {
    hipEventDestroy(sp.a);
    hipEventDestroy(sp.b);
  }
  for (auto e : c->pool) hipEventDestroy(e);
  for (hipStream_t p : c->preps) {
    hipStreamSynchronize(p);
    hipStreamDestroy(p);
  }
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return DFH_OK;
}

int dfh_ctx_sync(dfh_ctx* c) {
  DFH_ARG(c, "ctx is NULL");
  return sync_all(c);
}

int dfh_ctx_set_pipeline(dfh_ctx* c, int enable) {
  DFH_ARG(c, "ctx is NULL");
  DFH_ARG(enable >= 0 && enable <= 4, "dfh_ctx_set_pipeline: 0 (off) .. 4 preparation streams");
  if (c->single_queue && enable) {  // the single-queue step has no preparation stream; minibatches are still prepared ahead
    return DFH_OK;
  }
  int rc = sync_all(c);
  if (rc) return rc;
  while ((int)c->preps.size() < enable) {
    // lowest priority: the main stream (lookup -> forward -> backward) is the critical path of a
    // step; the preparation of later batches has a whole step (or two) of slack and fills the gaps
    // (measured: highest 72.9, normal 72.7, lowest 75.8 M examples/sec)
    int lo = 0, hi = 0;
    DFH_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const int prio = c->prep_priority < 0 ? lo : (c->prep_priority > 0 ? hi : 0);
    hipStream_t p = nullptr;
    DFH_HIP(hipStreamCreateWithPriority(&p, hipStreamNonBlocking, prio));
    c->preps.push_back(p);
  }
  c->nprep = (unsigned)enable;
  c->next_prep = 0;
  c->pipeline = enable != 0;
  return DFH_OK;
}
int dfh_ctx_set_option(dfh_ctx* c, const char* name, int value) {
  DFH_ARG(c && name, "dfh_ctx_set_option: NULL argument");
  const std::string n(name);
  if (n == "fwd_depth") {
    DFH_ARG(value == 4 || value == 5 || value == 8 || value == 10, "fwd_depth must be 4, 5, 8 or 10");
    c->fwd_depth = value;
  } else if (n == "fwd_blocks") {
    DFH_ARG(value >= 0 && value <= PROG_SLOTS, "fwd_blocks must be in [0, 16384] (0: one wave per example)");
    c->fwd_blocks = value;
  } else if (n == "bwd_small_blocks") {
    DFH_ARG(value >= 1 && value <= 65536, "bwd_small_blocks must be in [1, 65536]");
    c->bwd_small_blocks = value;
  } else if (n == "upd_kernel") {
    DFH_ARG(value == 0 || value == 1, "upd_kernel must be 0 (k_backward_all) or 1 (k_update_fused)");
    c->upd_kernel = value;
  } else if (n == "upd_hot_blocks" || n == "upd_mid_blocks" || n == "upd_few_blocks" || n == "upd_single_blocks") {
    DFH_ARG(value >= 1 && value <= 65536, "upd_*_blocks must be in [1, 65536]");
    (n == "upd_hot_blocks" ? c->upd_hot_blocks : n == "upd_mid_blocks" ? c->upd_mid_blocks : n == "upd_few_blocks" ? c->upd_few_blocks
                                                                                                        : c->upd_single_blocks) = value;
  } else if (n == "grow_initial_rows") {
    DFH_ARG(value >= 16 && value <= (1 << 28), "grow_initial_rows must be in [16, 2^28]");
    c->grow_initial_rows = value;
  } else if (n == "auc_in_update") {
    DFH_ARG(value == 0 || value == 1, "auc_in_update must be 0 (k_auc_pairs as its own launch) or 1 (a role of k_update_fused)");
    c->auc_in_update = value;
  } else if (n == "shard_mixed_update") {
    DFH_ARG(value == 0 || value == 1, "shard_mixed_update must be 0 or 1");
    c->shard_mixed_update = value;
  } else if (n == "owner_per_key") {
    DFH_ARG(value == 0 || value == 1, "owner_per_key must be 0 (count push, Pull, gradient push per received entry) or 1 (per distinct key)");
    c->owner_per_key = value;
  } else if (n == "upd_split") {
    DFH_ARG(value == 0 || value == 1, "upd_split must be 0 (one block per hot key, whatever its length) or 1 (a block per part of 1 024 occurrences)");
    c->upd_split = value;
  } else if (n == "upd_interleave") {
    DFH_ARG(value >= 0 && value <= 64, "upd_interleave must be in [0, 64]");
    c->upd_interleave = value;
  } else if (n == "event_flags") {
    DFH_ARG(value == 0 || value == 1, "event_flags must be 0 (default events) or 1 (no system-scope fence); set before batches are created");
    c->event_flags = value;
  } else if (n == "single_queue") {
    DFH_ARG(value == 0 || value == 1, "single_queue must be 0 (preparation streams) or 1 (the Localizer's stages ride in the step's launches)");
    if (value != c->single_queue) {
      for (dfh_batch* p : std::vector<dfh_batch*>(c->pend)) {
        int rcf = flush_pending(p);
        if (rcf) return rcf;
      }
      int rcs = sync_all(c);
      if (rcs) return rcs;
      if (value) {  // one queue: no preparation streams
        c->nprep = 0;
        c->next_prep = 0;
        c->pipeline = false;
      }
      c->single_queue = value;
    }
  } else if (n == "rider_slot_count" || n == "rider_slot_scatter" || n == "rider_slot_sort" || n == "rider_slot_emit") {
    // which launch of a step carries the stage: 0 lookup, 1 forward, 2 update; + 4: as a launch of its own just before it
    DFH_ARG(value >= 0 && value <= 6 && (value & 3) <= 2, "rider_slot_*: 0 lookup, 1 forward, 2 update (+ 4: alone, before that launch)");
    const int st = n == "rider_slot_count" ? 0 : n == "rider_slot_scatter" ? 1 : n == "rider_slot_sort" ? 2 : 3;
    c->rider_slot[st] = value & 3;
    c->rider_alone[st] = (value >> 2) & 1;
  } else if (n == "rider_start_lookup" || n == "rider_start_forward" || n == "rider_start_update") {
    DFH_ARG(value >= 0 && value <= 100, "rider_start_*: per cent of the carrier's own blocks dispatched before the first rider group");
    c->rider_start[n == "rider_start_lookup" ? 0 : n == "rider_start_forward" ? 1 : 2] = value;
  } else if (n == "rider_period_lookup" || n == "rider_period_forward" || n == "rider_period_update") {
    DFH_ARG(value >= 1 && value <= 4096, "rider_period_*: one group of 8 rider blocks every n groups (1: riders first)");
    c->rider_period[n == "rider_period_lookup" ? 0 : n == "rider_period_forward" ? 1 : 2] = value;
  } else if (n == "prep_priority") {
    DFH_ARG(value >= -1 && value <= 1, "prep_priority must be -1 (lowest), 0 (default) or 1 (highest)");
    DFH_ARG(c->preps.empty(), "prep_priority must be set before dfh_ctx_set_pipeline creates the streams");
    c->prep_priority = value;
  } else {
    set_error("dfh_ctx_set_option: unknown option " + n);
    return DFH_ERR_ARG;
  }
  return DFH_OK;
}
void* dfh_ctx_stream(dfh_ctx* c) { return c ? c->stream : nullptr; }
int dfh_ctx_device(dfh_ctx* c) { return c ? c->device : -1; }

int dfh_ctx_set_timing(dfh_ctx* c, int enable) {
  DFH_ARG(c, "ctx is NULL");
  c->timing = enable ? ~0u : 0u;
  return DFH_OK;
}

int dfh_ctx_set_timing_mask(dfh_ctx* c, uint32_t mask) {
  DFH_ARG(c, "ctx is NULL");
  c->timing = mask;
  return DFH_OK;
}

int dfh_ctx_get_timing(dfh_ctx* c, int reset, double* total_ms, uint64_t* calls) {
  DFH_ARG(c, "ctx is NULL");
  {
    int rc = sync_all(c);
    if (rc) return rc;
  }
  if (getenv("DFH_GAP_TRACE") && c->spans.size() > 1) {
    double g[DFH_K_COUNT][DFH_K_COUNT] = {};
    uint64_t n[DFH_K_COUNT][DFH_K_COUNT] = {};
    const dfh_ctx::Span* prev = nullptr;  // the main stream's launches only: lookup (its own events), forward, update
    for (const auto& sp : c->spans) {
      const bool main_launch = sp.ride || sp.id == DFH_K_FORWARD || sp.id == DFH_K_BACKWARD;
      if (!main_launch) continue;
      float ms = 0;
      if (prev && hipEventElapsedTime(&ms, prev->b, sp.a) == hipSuccess) {
        g[prev->id][sp.id] += ms;
        n[prev->id][sp.id] += 1;
      }
      prev = &sp;
    }
    for (int x = 0; x < DFH_K_COUNT; ++x)
      for (int y = 0; y < DFH_K_COUNT; ++y)
        if (n[x][y])
          fprintf(stderr, "gap trace: end of %s -> start of %s: %.2f us on average (%llu pairs)\n", dfh_kernel_name(x), dfh_kernel_name(y),
                  g[x][y] / n[x][y] * 1e3, (unsigned long long)n[x][y]);
  }
  for (auto& sp : c->spans) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      c->t_ms[sp.id] += ms;
      c->t_calls[sp.id] += 1;
    }
    c->pool.push_back(sp.a);
    c->pool.push_back(sp.b);
  }
  c->spans.clear();
  for (int i = 0; i < DFH_K_COUNT; ++i) {
    if (total_ms) total_ms[i] = c->t_ms[i];
    if (calls) calls[i] = c->t_calls[i];
    if (reset) {
      c->t_ms[i] = 0;
      c->t_calls[i] = 0;
    }
  }
  return DFH_OK;
}

const char* dfh_kernel_name(int id) {
  static const char* names[DFH_K_COUNT] = {"localize", "lookup", "forward", "backward", "pull_rows", "push_grad", "misc", "auc"};
  return (id >= 0 && id < DFH_K_COUNT) ? names[id] : "?";
}

int dfh_malloc(dfh_ctx* c, size_t bytes, void** dptr) {
  DFH_ARG(c && dptr, "dfh_malloc: NULL argument");
  DFH_HIP(hipSetDevice(c->device));
  DFH_HIP(hipMalloc(dptr, std::max<size_t>(bytes, 16)));
  return DFH_OK;
}
int dfh_free(dfh_ctx* c, void* dptr) {
  DFH_ARG(c, "ctx is NULL");
  if (dptr) DFH_HIP(hipFree(dptr));
  return DFH_OK;
}
int dfh_memcpy_h2d(dfh_ctx* c, void* dst, const void* src, size_t bytes) {
  DFH_ARG(c, "ctx is NULL");
  DFH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  DFH_HIP(hipStreamSynchronize(c->stream));
  return DFH_OK;
}
int dfh_memcpy_d2h(dfh_ctx* c, void* dst, const void* src, size_t bytes) {
  DFH_ARG(c, "ctx is NULL");
  DFH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  DFH_HIP(hipStreamSynchronize(c->stream));
  return DFH_OK;
}

// -------------------------------------------------------------------- table
int dfh_table_create(dfh_ctx* c, const dfh_updater_param* p, uint64_t capacity_rows, dfh_table** out) {
  DFH_ARG(c && p && out, "dfh_table_create: NULL argument");
  DFH_ARG(p->V_dim >= 0 && p->V_dim <= 10000, "V_dim out of range [0, 10000] (FMLossParam, fm_loss.h:25)");
  // the four top bits of a row word carry flags (kRemoteRow, kSingleRow, kCountLater, kHasV); 2^28 rows of V_dim 64 are 150 GB,
  // of V_dim 128 (C5: 1.25e8 rows per GPU) 296 GB: more than the HBM either way
  DFH_ARG(capacity_rows <= (uint64_t)kRowMask, "capacity_rows must be in [1, 2^28), or 0: a table that grows");
  const bool auto_grow = capacity_rows == 0;
  if (auto_grow) capacity_rows = (uint64_t)c->grow_initial_rows;  // the first allocation (2^20 rows: 0.6 GB at V_dim 64); doubled as the model grows
  DFH_ARG(p->lr > 0, "lr must be > 0");
  DFH_ARG(p->init_mode == DFH_INIT_HASH || p->init_mode == DFH_INIT_REFRAND, "bad init_mode");
  DFH_HIP(hipSetDevice(c->device));
  dfh_table* t = new (std::nothrow) dfh_table();
  DFH_ARG(t != nullptr, "out of host memory");
  t->ctx = c;
  t->auto_grow = auto_grow;
  TableView& v = t->v;
  v.p = *p;
  v.k = p->V_dim;
  v.kp = (p->V_dim + 3) / 4 * 4;
  v.capacity = (uint32_t)capacity_rows;
  const uint64_t H = hslots_for(capacity_rows);
  t->hslots = H;
  v.hmask = H - 1;
  size_t ht_b = H * sizeof(HEntry);
  size_t hdr_b = capacity_rows * sizeof(RowHdr);
  size_t va_b = std::max<size_t>(capacity_rows * (size_t)(2 * v.kp) * sizeof(float), 256);
  hipError_t e;
  if ((e = hipMalloc(&v.ht, ht_b)) != hipSuccess || (e = hipMalloc(&v.hdr, hdr_b)) != hipSuccess ||
      (e = hipMalloc(&v.va, va_b)) != hipSuccess || (e = hipMalloc(&v.nrows, 256)) != hipSuccess) {
    set_error(std::string("dfh_table_create: hipMalloc: ") + hipGetErrorString(e));
    if (v.ht) hipFree(v.ht);
    if (v.hdr) hipFree(v.hdr);
    if (v.va) hipFree(v.va);
    delete t;
    return DFH_ERR_HIP;
  }
  v.err = v.nrows + 1;
  v.rng_state = v.nrows + 2;
  t->d_total = v.nrows + 3;
  t->bytes = ht_b + hdr_b + va_b;
  DFH_HIP(hipMemsetAsync(v.ht, 0xFF, ht_b, c->stream));  // key = ~0 (empty), row = ~0
  DFH_HIP(hipMemsetAsync(v.hdr, 0, hdr_b, c->stream));
  DFH_HIP(hipMemsetAsync(v.va, 0, va_b, c->stream));
  DFH_HIP(hipMemsetAsync(v.nrows, 0, 256, c->stream));
  uint32_t seed = p->seed;
  DFH_HIP(hipMemcpyAsync(v.rng_state, &seed, sizeof(seed), hipMemcpyHostToDevice, c->stream));
  DFH_HIP(hipStreamSynchronize(c->stream));
  *out = t;
  return DFH_OK;
}

int dfh_table_destroy(dfh_table* t) {
  if (!t) return DFH_OK;
  hipSetDevice(t->ctx->device);
  sync_all(t->ctx);  // preparation streams may still probe the table
  hipFree(t->v.ht);
  hipFree(t->v.hdr);
  hipFree(t->v.va);
  hipFree(t->v.nrows);
  if (t->d_need) {
    hipFree(t->d_need);
    hipFree(t->d_rank);
    hipFree(t->d_urow);
  }
  delete t;
  return DFH_OK;
}

int dfh_table_size(dfh_table* t, uint64_t* nkeys) {
  DFH_ARG(t && nkeys, "NULL argument");
  uint32_t n = 0;
  {
    int rc = sync_all(t->ctx);  // a preparation-stream lookup may still be inserting keys
    if (rc) return rc;
  }
  DFH_HIP(hipMemcpyAsync(&n, t->v.nrows, sizeof(n), hipMemcpyDeviceToHost, t->ctx->stream));
  DFH_HIP(hipStreamSynchronize(t->ctx->stream));
  *nkeys = std::min<uint64_t>(n, t->v.capacity);
  return check_table_err(t);
}

int dfh_table_param(dfh_table* t, dfh_updater_param* out) {
  DFH_ARG(t && out, "NULL argument");
  *out = t->v.p;
  return DFH_OK;
}
uint64_t dfh_table_bytes(dfh_table* t) { return t ? t->bytes : 0; }

int dfh_table_capacity(dfh_table* t, uint64_t* capacity_rows, uint64_t* grows) {
  DFH_ARG(t, "NULL table");
  if (capacity_rows) *capacity_rows = t->v.capacity;
  if (grows) *grows = t->grows;
  return DFH_OK;
}

int dfh_table_set_has_aux(dfh_table* t, int has_aux) {
  DFH_ARG(t, "NULL table");
  t->has_aux = has_aux != 0;
  return DFH_OK;
}
int dfh_table_has_aux(dfh_table* t) { return (t && t->has_aux) ? 1 : 0; }

int dfh_table_warm_start(dfh_table* t, const uint64_t* d_keys, size_t n, float w0, float cnt0) {
  DFH_ARG(t && (n == 0 || d_keys), "dfh_table_warm_start: NULL argument");
  if (n == 0) return DFH_OK;
  if (int rcr = table_reserve(t, n)) return rcr;
  hipLaunchKernelGGL(k_warm_start, dim3(grid_for_waves(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_keys,
                     (uint64_t)n, w0, cnt0);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// ---- device-pointer (sharded) store calls
int dfh_shard_pull(dfh_table* t, const uint64_t* d_keys, size_t n, float* d_rows) {
  DFH_ARG(t && (n == 0 || (d_keys && d_rows)), "dfh_shard_pull: NULL argument");
  if (n == 0) return DFH_OK;
  if (int rcr = table_reserve(t, n)) return rcr;  // Get inserts an entry for every id ever pulled (sgd_updater.cc:44)
  TimeScope ts(t->ctx, DFH_K_PULL);
  hipLaunchKernelGGL(k_pull_rows, dim3(grid_for_waves(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_keys,
                     (uint32_t)n, d_rows, dfh_row_stride(t->v.k));
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

int dfh_shard_push_count(dfh_table* t, const uint64_t* d_keys, size_t n, const float* d_cnt) {
  DFH_ARG(t && (n == 0 || (d_keys && d_cnt)), "dfh_shard_push_count: NULL argument");
  if (n == 0) return DFH_OK;
  if (int rcr = table_reserve(t, n)) return rcr;
  bool refrand = t->v.p.init_mode == DFH_INIT_REFRAND && t->v.k > 0;
  if (refrand) {
    int rc = ensure_table_aux(t, n);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_lookup, dim3(grid_for_threads(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_keys,
                     (const uint32_t*)nullptr, (uint32_t)n, refrand ? t->d_urow : (uint32_t*)nullptr, d_cnt,
                     (const uint32_t*)nullptr, 1, refrand ? t->d_need : (uint32_t*)nullptr, 0, (uint2*)nullptr, AucFin{nullptr, 0u, nullptr});
  DFH_HIP(hipGetLastError());
  if (refrand) return refrand_flush(t, d_keys, nullptr, (uint32_t)n, t->d_urow, t->d_need, t->d_rank, t->d_total);
  return DFH_OK;
}

int dfh_shard_push_grad(dfh_table* t, const uint64_t* d_keys, size_t n, const float* d_grads) {
  DFH_ARG(t && (n == 0 || (d_keys && d_grads)), "dfh_shard_push_grad: NULL argument");
  if (int rca = require_aux(t, "dfh_shard_push_grad")) return rca;
  if (n == 0) return DFH_OK;
  if (int rcr = table_reserve(t, n)) return rcr;
  bool refrand = t->v.p.init_mode == DFH_INIT_REFRAND && t->v.k > 0;
  if (refrand) {
    int rc = ensure_table_aux(t, n);
    if (rc) return rc;
  }
  TimeScope* ts = new TimeScope(t->ctx, DFH_K_PUSH);
  hipLaunchKernelGGL(k_push_grad, dim3(grid_for_waves(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_keys,
                     (uint32_t)n, d_grads, dfh_row_stride(t->v.k), refrand ? t->d_need : (uint32_t*)nullptr,
                     refrand ? t->d_urow : (uint32_t*)nullptr);
  delete ts;
  DFH_HIP(hipGetLastError());
  if (refrand) return refrand_flush(t, d_keys, nullptr, (uint32_t)n, t->d_urow, t->d_need, t->d_rank, t->d_total);
  return DFH_OK;
}

// ---- resolved owner-side calls: probe once per step, then work on row ids
int dfh_shard_resolve(dfh_table* t, const uint64_t* d_keys, size_t n, uint32_t* d_rowid) {
  DFH_ARG(t && (n == 0 || (d_keys && d_rowid)), "dfh_shard_resolve: NULL argument");
  if (n == 0) return DFH_OK;
  if (int rcr = table_reserve(t, n)) return rcr;
  TimeScope ts(t->ctx, DFH_K_LOOKUP);
  hipLaunchKernelGGL(k_resolve, dim3(grid_for_threads(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_keys, (uint32_t)n,
                     d_rowid);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

int dfh_shard_pull_resolved(dfh_table* t, const uint32_t* d_rowid, size_t n, float* d_rows) {
  DFH_ARG(t && (n == 0 || (d_rowid && d_rows)), "dfh_shard_pull_resolved: NULL argument");
  if (n == 0) return DFH_OK;
  TimeScope ts(t->ctx, DFH_K_PULL);
  const size_t stride = dfh_row_stride(t->v.k);
  int rc = dispatch_L(std::max(t->v.kp, 4), [&](auto Lc) {
    constexpr int L = decltype(Lc)::value;
    const size_t blocks = std::min<size_t>((n * L + 255) / 256, (size_t)t->ctx->num_cu * 16);
    hipLaunchKernelGGL((k_pull_resolved<L>), dim3((unsigned)blocks), dim3(256), 0, t->ctx->stream, t->v, d_rowid, (uint32_t)n,
                       d_rows, stride);
  });
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

int dfh_shard_push_count_resolved(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, size_t n, const float* d_cnt) {
  DFH_ARG(t && (n == 0 || (d_rowid && d_keys && d_cnt)), "dfh_shard_push_count_resolved: NULL argument");
  if (!(t->v.p.init_mode == DFH_INIT_HASH || t->v.k == 0)) {
    set_error("resolved store calls need V_init = hash (order independent)");
    return DFH_ERR_STATE;
  }
  if (n == 0) return DFH_OK;
  hipLaunchKernelGGL(k_lookup, dim3(grid_for_threads(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_keys,
                     (const uint32_t*)nullptr, (uint32_t)n, const_cast<uint32_t*>(d_rowid), d_cnt, (const uint32_t*)nullptr, 1,
                     (uint32_t*)nullptr, 1, (uint2*)nullptr, AucFin{nullptr, 0u, nullptr});
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

int dfh_shard_push_grad_resolved(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, size_t n, const float* d_grads) {
  DFH_ARG(t && (n == 0 || (d_rowid && d_keys && d_grads)), "dfh_shard_push_grad_resolved: NULL argument");
  if (int rca = require_aux(t, "dfh_shard_push_grad_resolved")) return rca;
  if (!(t->v.p.init_mode == DFH_INIT_HASH || t->v.k == 0)) {
    set_error("resolved store calls need V_init = hash (order independent)");
    return DFH_ERR_STATE;
  }
  if (n == 0) return DFH_OK;
  TimeScope ts(t->ctx, DFH_K_PUSH);
  const size_t stride = dfh_row_stride(t->v.k);
  int rc = dispatch_L(std::max(t->v.kp, 4), [&](auto Lc) {
    constexpr int L = decltype(Lc)::value;
    const size_t blocks = std::min<size_t>((n * L + 255) / 256, (size_t)t->ctx->num_cu * 16);
    hipLaunchKernelGGL((k_push_grad_resolved<L>), dim3((unsigned)blocks), dim3(256), 0, t->ctx->stream, t->v, d_rowid, d_keys,
                       (uint32_t)n, d_grads, stride);
  });
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// ---- all source ranks of a step in one launch per operation
namespace {
int make_segoff(const size_t* seg, int nsrc, int mask_slot, SegOff* out) {
  DFH_ARG(seg && nsrc >= 1 && nsrc <= 32, "seg must hold nsrc+1 offsets, 1 <= nsrc <= 32");
  DFH_ARG(mask_slot == 0 || mask_slot == 1, "mask_slot must be 0 or 1");
  out->slot = mask_slot;
  DFH_ARG(seg[0] == 0, "seg[0] must be 0");
  for (int s = 0; s <= nsrc; ++s) {
    DFH_ARG(seg[s] < (1ULL << 27) - 1 && (s == 0 || seg[s] >= seg[s - 1]), "seg must be ascending offsets below 2^27");
    out->off[s] = (uint32_t)seg[s];
  }
  out->nsrc = nsrc;
  return DFH_OK;
}
bool hash_init_only(const dfh_table* t) { return t->v.p.init_mode == DFH_INIT_HASH || t->v.k == 0; }
}  // namespace

int dfh_shard_resolve_multi(dfh_table* t, const uint64_t* d_keys, const size_t* seg, int nsrc, int mask_slot,
                            uint32_t* d_rowid) {
  DFH_ARG(t, "NULL table");
  SegOff g;
  int rc = make_segoff(seg, nsrc, mask_slot, &g);
  if (rc) return rc;
  const size_t n = g.off[nsrc];
  DFH_ARG(n == 0 || (d_keys && d_rowid), "dfh_shard_resolve_multi: NULL argument");
  if (n == 0) return DFH_OK;
  if (int rcr = table_reserve(t, n)) return rcr;
  TimeScope ts(t->ctx, DFH_K_LOOKUP);
  hipLaunchKernelGGL(k_resolve_multi, dim3(grid_for_threads(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_keys, g, d_rowid);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

int dfh_shard_push_count_multi(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc,
                               int mask_slot, const float* d_cnt) {
  DFH_ARG(t, "NULL table");
  if (!hash_init_only(t)) {
    set_error("multi-source store calls need V_init = hash (order independent)");
    return DFH_ERR_STATE;
  }
  SegOff g;
  int rc = make_segoff(seg, nsrc, mask_slot, &g);
  if (rc) return rc;
  const size_t n = g.off[nsrc];
  DFH_ARG(n == 0 || (d_rowid && d_keys && d_cnt), "dfh_shard_push_count_multi: NULL argument");
  if (n == 0) return DFH_OK;
  TimeScope ts(t->ctx, DFH_K_LOOKUP);
  hipLaunchKernelGGL(k_push_count_multi, dim3(grid_for_threads(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_rowid, d_keys, g,
                     d_cnt);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

int dfh_shard_push_grad_multi(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc,
                              int mask_slot, const float* d_grads) {
  DFH_ARG(t, "NULL table");
  if (int rca = require_aux(t, "dfh_shard_push_grad_multi")) return rca;
  if (!hash_init_only(t)) {
    set_error("multi-source store calls need V_init = hash (order independent)");
    return DFH_ERR_STATE;
  }
  SegOff g;
  int rc = make_segoff(seg, nsrc, mask_slot, &g);
  if (rc) return rc;
  const size_t n = g.off[nsrc];
  DFH_ARG(n == 0 || (d_rowid && d_keys && d_grads), "dfh_shard_push_grad_multi: NULL argument");
  if (n == 0) return DFH_OK;
  TimeScope ts(t->ctx, DFH_K_PUSH);
  const size_t stride = dfh_row_stride(t->v.k);
  rc = dispatch_L(std::max(t->v.kp, 4), [&](auto Lc) {
    constexpr int L = decltype(Lc)::value;
    const size_t blocks = std::min<size_t>((n * L + 255) / 256, (size_t)t->ctx->num_cu * 16);   // (8 / 32 per CU: 61.0 / 58.6 against 57.4 us)
    hipLaunchKernelGGL((k_push_grad_multi<L>), dim3((unsigned)blocks), dim3(256), 0, t->ctx->stream, t->v, d_rowid, d_keys, g,
                       d_grads, stride);
  });
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// Push(kFeaCount) of all sources (d_cnt, or NULL: none this step) and Pull, per distinct key, in one launch; leaves the key
// lists dfh_shard_push_grad_listed runs over
int dfh_shard_count_pull_multi(dfh_table* t, uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc, int mask_slot,
                               const float* d_cnt, float* d_rows) {
  DFH_ARG(t, "NULL table");
  if (d_cnt && !hash_init_only(t)) {
    set_error("multi-source store calls need V_init = hash (order independent)");
    return DFH_ERR_STATE;
  }
  SegOff g;
  int rc = make_segoff(seg, nsrc, mask_slot, &g);
  if (rc) return rc;
  const size_t n = g.off[nsrc];
  DFH_ARG(n == 0 || (d_rowid && d_keys && d_rows), "dfh_shard_count_pull_multi: NULL argument");
  if (n == 0) return DFH_OK;
  TimeScope ts(t->ctx, DFH_K_PULL);
  const size_t stride = dfh_row_stride(t->v.k);
  const size_t nchunk = (n + MULTI_CH - 1) / MULTI_CH;
  rc = dispatch_L(std::max(t->v.kp, 4), [&](auto Lc) {
    constexpr int L = decltype(Lc)::value;
    const dim3 grid((unsigned)std::min<size_t>(nchunk, (size_t)t->ctx->num_cu * 16));
    if (d_cnt)
      hipLaunchKernelGGL((k_count_pull_multi<L, true>), grid, dim3(256), 0, t->ctx->stream, t->v, d_rowid, d_keys, g, d_cnt, d_rows, stride);
    else
      hipLaunchKernelGGL((k_count_pull_multi<L, false>), grid, dim3(256), 0, t->ctx->stream, t->v, d_rowid, d_keys, g, d_cnt, d_rows, stride);
  });
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// Push(kGradient) of all sources over the key lists dfh_shard_count_pull_multi left in d_rowid (same seg, same slot)
int dfh_shard_push_grad_listed(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc,
                               int mask_slot, const float* d_grads) {
  DFH_ARG(t, "NULL table");
  if (int rca = require_aux(t, "dfh_shard_push_grad_listed")) return rca;
  if (!hash_init_only(t)) {
    set_error("multi-source store calls need V_init = hash (order independent)");
    return DFH_ERR_STATE;
  }
  SegOff g;
  int rc = make_segoff(seg, nsrc, mask_slot, &g);
  if (rc) return rc;
  const size_t n = g.off[nsrc];
  DFH_ARG(n == 0 || (d_rowid && d_keys && d_grads), "dfh_shard_push_grad_listed: NULL argument");
  if (n == 0) return DFH_OK;
  TimeScope ts(t->ctx, DFH_K_PUSH);
  const size_t stride = dfh_row_stride(t->v.k);
  const size_t nchunk = (n + MULTI_CH - 1) / MULTI_CH;
  rc = dispatch_L(std::max(t->v.kp, 4), [&](auto Lc) {
    constexpr int L = decltype(Lc)::value;
    const dim3 grid((unsigned)std::min<size_t>(nchunk, (size_t)t->ctx->num_cu * 16));
    hipLaunchKernelGGL((k_push_grad_chunks<L>), grid, dim3(256), 0, t->ctx->stream, t->v, d_rowid, d_keys, g, d_grads, stride);
  });
  if (rc) return rc;
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

size_t dfh_shard_multi_words(size_t n, int nsrc) { return multi_words(n, nsrc < 1 ? 1 : nsrc); }

int dfh_shard_release(dfh_table* t, const uint32_t* d_rowid, size_t n, int mask_slot) {
  DFH_ARG(t && (n == 0 || d_rowid), "dfh_shard_release: NULL argument");
  DFH_ARG(mask_slot == 0 || mask_slot == 1, "mask_slot must be 0 or 1");
  if (n == 0) return DFH_OK;
  hipLaunchKernelGGL(k_release_rows, dim3(grid_for_threads(n, t->ctx)), dim3(256), 0, t->ctx->stream, t->v, d_rowid, (uint32_t)n,
                     mask_slot);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

#ifdef DFH_BWD_TRACE
// measurement build only (tools/): {role, start, end} of every block of the last k_backward_all launch
int dfh_debug_bwd_trace(unsigned long long* out, size_t n) {
  DFH_HIP(hipDeviceSynchronize());
  DFH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_trace), std::min<size_t>(n, 3 * 8192) * sizeof(unsigned long long)));
  return DFH_OK;
}
#endif

int dfh_table_check(dfh_table* t) {
  DFH_ARG(t, "NULL table");
  return check_table_err(t);
}

// ---- literal Store API (host pointers)
namespace {
// the literal calls hand every key to its own lane: a key listed twice would be updated by two
// lanes at once (the reference applies them one after the other; the Localizer never produces them)
int check_keys(const uint64_t* keys, size_t n) {
  bool ascending = true;
  for (size_t i = 0; i < n; ++i) {
    DFH_ARG(keys[i] != kEmptyKey, "key ~0 is reserved");
    if (i && keys[i] <= keys[i - 1]) ascending = false;
  }
  if (ascending) return DFH_OK;  // Localizer output: strictly ascending, hence unique
  std::vector<uint64_t> tmp(keys, keys + n);
  std::sort(tmp.begin(), tmp.end());
  DFH_ARG(std::adjacent_find(tmp.begin(), tmp.end()) == tmp.end(), "duplicate key inside one push/pull");
  return DFH_OK;
}
}  // namespace

int dfh_pull(dfh_table* t, const uint64_t* keys, size_t n, float* vals, size_t* nvals, int* lens, size_t* nlens) {
  DFH_ARG(t && nvals && nlens && (n == 0 || (keys && vals && lens)), "dfh_pull: NULL argument");
  const int k = t->v.k;
  *nvals = 0;
  *nlens = k == 0 ? 0 : n;  // sgd_updater.cc:40
  if (n == 0) return DFH_OK;
  if (int rck = check_keys(keys, n)) return rck;
  dfh_ctx* c = t->ctx;
  DFH_HIP(hipSetDevice(c->device));
  const size_t stride = dfh_row_stride(k);
  int rc = ensure_scratch(c, padded<uint64_t>(n) + padded<float>(n * stride));
  if (rc) return rc;
  Carver cv(c->scratch);
  uint64_t* d_keys = cv.take<uint64_t>(n);
  float* d_rows = cv.take<float>(n * stride);
  DFH_HIP(hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  rc = dfh_shard_pull(t, d_keys, n, d_rows);
  if (rc) return rc;
  std::vector<float> rows(n * stride);
  DFH_HIP(hipMemcpyAsync(rows.data(), d_rows, rows.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  DFH_HIP(hipStreamSynchronize(c->stream));
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

int dfh_push(dfh_table* t, const uint64_t* keys, size_t n, int val_type, const float* vals, size_t nvals,
             const int* lens, size_t nlens) {
  DFH_ARG(t && (n == 0 || keys), "dfh_push: NULL argument");
  DFH_ARG(val_type == DFH_FEA_COUNT || val_type == DFH_GRADIENT, "dfh_push: unknown val_type (sgd_updater.cc:99)");
  dfh_ctx* c = t->ctx;
  DFH_HIP(hipSetDevice(c->device));
  const int k = t->v.k;
  if (int rck = check_keys(keys, n)) return rck;
  if (val_type == DFH_FEA_COUNT) {
    DFH_ARG(nvals == n, "kFeaCount: CHECK_EQ(fea_ids.size(), values.size()) (sgd_updater.cc:63)");
    if (n == 0) return DFH_OK;
    int rc = ensure_scratch(c, padded<uint64_t>(n) + padded<float>(n));
    if (rc) return rc;
    Carver cv(c->scratch);
    uint64_t* d_keys = cv.take<uint64_t>(n);
    float* d_cnt = cv.take<float>(n);
    DFH_HIP(hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    DFH_HIP(hipMemcpyAsync(d_cnt, vals, n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    rc = dfh_shard_push_count(t, d_keys, n, d_cnt);
    if (rc) return rc;
    return check_table_err(t);
  }
  // kGradient (sgd_updater.cc:74-97)
  if (int rca = require_aux(t, "dfh_push(kGradient)")) return rca;
  const bool w_only = nlens == 0;
  if (w_only) {
    DFH_ARG(nvals == n, "kGradient: CHECK_EQ(values.size(), size) (sgd_updater.cc:79)");
  } else {
    DFH_ARG(nlens == n && lens, "kGradient: CHECK_EQ(lens.size(), size) (sgd_updater.cc:81)");
  }
  if (n == 0) return DFH_OK;
  const size_t stride = dfh_row_stride(k);
  std::vector<float> rows(n * stride, 0.0f);
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
  int rc = ensure_scratch(c, padded<uint64_t>(n) + padded<float>(n * stride));
  if (rc) return rc;
  Carver cv(c->scratch);
  uint64_t* d_keys = cv.take<uint64_t>(n);
  float* d_rows = cv.take<float>(n * stride);
  DFH_HIP(hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  DFH_HIP(hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  rc = dfh_shard_push_grad(t, d_keys, n, d_rows);
  if (rc) return rc;
  return check_table_err(t);
}

int dfh_table_export(dfh_table* t, uint64_t cap, uint64_t* keys, float* scal, int* has_V, float* V, uint64_t* n) {
  DFH_ARG(t && n, "NULL argument");
  dfh_ctx* c = t->ctx;
  DFH_HIP(hipSetDevice(c->device));
  uint64_t cnt = 0;
  int rc = dfh_table_size(t, &cnt);
  if (rc) return rc;
  *n = cnt;
  if (cap == 0 || cnt == 0) return DFH_OK;
  DFH_ARG(cap >= cnt && keys && scal && has_V, "dfh_table_export: buffers too small");
  const int k = t->v.k;
  size_t vfl = (size_t)cnt * 2 * (size_t)std::max(k, 1);
  rc = ensure_scratch(c, padded<uint64_t>(cnt) + padded<float>(cnt * 4) + padded<int>(cnt) + padded<float>(vfl) + 512);
  if (rc) return rc;
  Carver cv(c->scratch);
  uint64_t* d_keys = cv.take<uint64_t>(cnt);
  float* d_scal = cv.take<float>(cnt * 4);
  int* d_has = cv.take<int>(cnt);
  float* d_V = cv.take<float>(vfl);
  unsigned long long* d_counter = cv.take<unsigned long long>(1);
  DFH_HIP(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_export, dim3(grid_for_threads(t->hslots, c)), dim3(256), 0, c->stream, t->v, t->hslots, cnt,
                     d_keys, d_scal, d_has, d_V, d_counter);
  DFH_HIP(hipGetLastError());
  DFH_HIP(hipMemcpyAsync(keys, d_keys, cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  DFH_HIP(hipMemcpyAsync(scal, d_scal, cnt * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  DFH_HIP(hipMemcpyAsync(has_V, d_has, cnt * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (k > 0 && V) DFH_HIP(hipMemcpyAsync(V, d_V, (size_t)cnt * 2 * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  DFH_HIP(hipStreamSynchronize(c->stream));
  return DFH_OK;
}

int dfh_table_import(dfh_table* t, uint64_t n, const uint64_t* keys, const float* scal, const int* has_V, const float* V) {
  DFH_ARG(t && (n == 0 || (keys && scal && has_V)), "NULL argument");
  if (n == 0) return DFH_OK;
  dfh_ctx* c = t->ctx;
  DFH_HIP(hipSetDevice(c->device));
  const int k = t->v.k;
  DFH_ARG(k == 0 || V, "V is NULL");
  if (int rcr = table_reserve(t, n)) return rcr;
  size_t vfl = (size_t)n * 2 * (size_t)std::max(k, 1);
  int rc = ensure_scratch(c, padded<uint64_t>(n) + padded<float>(n * 4) + padded<int>(n) + padded<float>(vfl));
  if (rc) return rc;
  Carver cv(c->scratch);
  uint64_t* d_keys = cv.take<uint64_t>(n);
  float* d_scal = cv.take<float>(n * 4);
  int* d_has = cv.take<int>(n);
  float* d_V = cv.take<float>(vfl);
  DFH_HIP(hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
  DFH_HIP(hipMemcpyAsync(d_scal, scal, n * 4 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  DFH_HIP(hipMemcpyAsync(d_has, has_V, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  if (k > 0) DFH_HIP(hipMemcpyAsync(d_V, V, (size_t)n * 2 * k * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_import, dim3(grid_for_threads(n, c)), dim3(256), 0, c->stream, t->v, n, d_keys, d_scal, d_has, d_V);
  DFH_HIP(hipGetLastError());
  return check_table_err(t);
}

// --------------------------------------------------------- literal Loss API
namespace {
int to_u32_offsets(const size_t* offset, size_t nrows, std::vector<uint32_t>* out) {
  out->resize(nrows + 1);
  const size_t base = offset[0];
  for (size_t i = 0; i <= nrows; ++i) {
    size_t o = offset[i] - base;
    DFH_ARG(o < 0xFFFFFFFFULL, "batch has >= 2^32 nonzeros (localizer.cc:19 CHECK_LT)");
    (*out)[i] = (uint32_t)o;
  }
  return DFH_OK;
}
}  // namespace

// ---- model files (Updater::Save / Load, include/difacto/updater.h:40-47: stubs in the reference, so the
// format is ours and shared with the C++ host's DeviceSGDUpdater):
//   "DFHM", u32 version, i32 V_dim, i32 has_aux, u64 n, then n entries
//   {u64 key, f32 w, [f32 fea_cnt, sqrt_g, z if aux], i32 has_V, [V_dim f32 V, [V_dim f32 acc if aux]] if has_V}
int dfh_table_save(dfh_table* t, const char* path, int save_aux, uint64_t* n_saved) {
  DFH_ARG(t && path, "dfh_table_save: NULL argument");
  uint64_t n = 0;
  int rc = dfh_table_size(t, &n);
  if (rc) return rc;
  const int k = t->v.k;
  const size_t kk = (size_t)std::max(k, 1);
  const uint64_t cap = std::max<uint64_t>(n, 1);
  std::vector<uint64_t> keys(cap);
  std::vector<float> scal(cap * 4), V(cap * 2 * kk);
  std::vector<int> hasv(cap);
  uint64_t m = 0;
  rc = dfh_table_export(t, cap, keys.data(), scal.data(), hasv.data(), V.data(), &m);
  if (rc) return rc;
  FILE* f = fopen(path, "wb");
  if (!f) {
    set_error(std::string("dfh_table_save: cannot open ") + path);
    return DFH_ERR_ARG;
  }
  std::vector<char> iobuf(1 << 22);
  setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
  // entries with w == 0 and no V carry nothing a predictor needs: skipped unless aux is wanted
  uint64_t kept = 0;
  for (uint64_t i = 0; i < m; ++i) kept += (save_aux || scal[i * 4 + 1] != 0 || hasv[i]) ? 1 : 0;
  const uint32_t ver = 1;
  const int32_t kv = k, aux = save_aux ? 1 : 0;
  bool ok = fwrite("DFHM", 1, 4, f) == 4 && fwrite(&ver, 4, 1, f) == 1 && fwrite(&kv, 4, 1, f) == 1 &&
            fwrite(&aux, 4, 1, f) == 1 && fwrite(&kept, 8, 1, f) == 1;
  for (uint64_t i = 0; ok && i < m; ++i) {
    if (!(save_aux || scal[i * 4 + 1] != 0 || hasv[i])) continue;
    ok = ok && fwrite(&keys[i], 8, 1, f) == 1 && fwrite(&scal[i * 4 + 1], 4, 1, f) == 1;
    if (save_aux) {
      const float a3[3] = {scal[i * 4 + 0], scal[i * 4 + 2], scal[i * 4 + 3]};
      ok = ok && fwrite(a3, 4, 3, f) == 3;
    }
    const int32_t hv = hasv[i];
    ok = ok && fwrite(&hv, 4, 1, f) == 1;
    if (hv && k > 0) {
      ok = ok && fwrite(&V[i * 2 * k], 4, (size_t)k, f) == (size_t)k;
      if (save_aux) ok = ok && fwrite(&V[i * 2 * k + k], 4, (size_t)k, f) == (size_t)k;
    }
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok) {
    set_error(std::string("dfh_table_save: write error on ") + path);
    return DFH_ERR_ARG;
  }
  if (n_saved) *n_saved = kept;
  return DFH_OK;
}

int dfh_table_load(dfh_table* t, const char* path, uint64_t key_lo, uint64_t key_hi, int* has_aux, uint64_t* n_loaded) {
  DFH_ARG(t && path, "dfh_table_load: NULL argument");
  FILE* f = fopen(path, "rb");
  if (!f) {
    set_error(std::string("dfh_table_load: cannot open ") + path);
    return DFH_ERR_ARG;
  }
  std::vector<char> iobuf(1 << 22);
  setvbuf(f, iobuf.data(), _IOFBF, iobuf.size());
  char magic[4];
  uint32_t ver = 0;
  int32_t k = 0, aux = 0;
  uint64_t n = 0;
  bool ok = fread(magic, 1, 4, f) == 4 && !memcmp(magic, "DFHM", 4) && fread(&ver, 4, 1, f) == 1 && fread(&k, 4, 1, f) == 1 &&
            fread(&aux, 4, 1, f) == 1 && fread(&n, 8, 1, f) == 1;
  if (!ok || k != t->v.k) {
    fclose(f);
    set_error(!ok ? "dfh_table_load: not a difacto-hip model file" : "dfh_table_load: model V_dim differs from the table's");
    return DFH_ERR_ARG;
  }
  if (has_aux) *has_aux = aux != 0;
  const size_t kk = (size_t)std::max(k, 1);
  const size_t chunk = 1 << 20;
  std::vector<uint64_t> keys;
  std::vector<float> scal, V, row(2 * kk);
  std::vector<int> hasv;
  uint64_t loaded = 0;
  int rc = DFH_OK;
  auto flush = [&]() -> int {
    if (keys.empty()) return DFH_OK;
    int r = dfh_table_import(t, keys.size(), keys.data(), scal.data(), hasv.data(), V.data());
    loaded += keys.size();
    keys.clear(); scal.clear(); V.clear(); hasv.clear();
    return r;
  };
  for (uint64_t i = 0; ok && rc == DFH_OK && i < n; ++i) {
    uint64_t key;
    float w, a3[3] = {0.f, 0.f, 0.f};
    int32_t hv;
    ok = fread(&key, 8, 1, f) == 1 && fread(&w, 4, 1, f) == 1;
    if (ok && aux) ok = fread(a3, 4, 3, f) == 3;
    ok = ok && fread(&hv, 4, 1, f) == 1;
    std::fill(row.begin(), row.end(), 0.f);
    if (ok && hv && k > 0) {
      ok = fread(row.data(), 4, (size_t)k, f) == (size_t)k;
      if (ok && aux) ok = fread(row.data() + k, 4, (size_t)k, f) == (size_t)k;
    }
    if (!ok) break;
    if (key < key_lo || (key_hi != 0 && key >= key_hi)) continue;  // another shard's key
    keys.push_back(key);
    scal.push_back(a3[0]); scal.push_back(w); scal.push_back(a3[1]); scal.push_back(a3[2]);
    hasv.push_back(hv);
    V.insert(V.end(), row.begin(), row.begin() + 2 * kk);
    if (keys.size() == chunk) rc = flush();
  }
  fclose(f);
  if (!ok) {
    set_error(std::string("dfh_table_load: truncated model file ") + path);
    return DFH_ERR_ARG;
  }
  if (rc == DFH_OK) rc = flush();
  if (n_loaded) *n_loaded = loaded;
  if (rc == DFH_OK && !aux && loaded > 0) t->has_aux = false;  // entries now lack fea_cnt / FTRL / AdaGrad state
  return rc;
}

int dfh_fm_predict(dfh_ctx* c, int V_dim, size_t nrows, const size_t* offset, const uint32_t* index, const float* value,
                   const float* weights, size_t nweights, const int* w_pos, const int* V_pos, size_t npos, float* pred) {
  DFH_ARG(c && V_dim >= 0, "dfh_fm_predict: bad argument");
  if (nrows == 0) return DFH_OK;
  DFH_ARG(offset && pred && weights, "dfh_fm_predict: NULL argument");
  DFH_ARG(V_dim == 0 || (w_pos && V_pos), "dfh_fm_predict: V_dim > 0 needs w_pos and V_pos");
  DFH_HIP(hipSetDevice(c->device));
  std::vector<uint32_t> off32;
  int rc = to_u32_offsets(offset, nrows, &off32);
  if (rc) return rc;
  const size_t base = offset[0];
  const size_t nnz = off32[nrows];
  DFH_ARG(nnz == 0 || index, "dfh_fm_predict: index is NULL");
  const size_t ncols = w_pos ? npos : nweights;
  for (size_t j = 0; j < nnz; ++j) DFH_ARG(index[base + j] < ncols, "dfh_fm_predict: index out of range");
  if (w_pos) {  // SpMV/SpMM::CheckPos (spmv.h:194-202, spmm.h:172-180)
    for (size_t u = 0; u < npos; ++u) {
      DFH_ARG(w_pos[u] == -1 || (w_pos[u] >= 0 && (size_t)w_pos[u] < nweights), "w_pos out of range");
      if (V_dim > 0) DFH_ARG(V_pos[u] == -1 || (V_pos[u] >= 0 && (size_t)V_pos[u] + V_dim <= nweights), "V_pos out of range");
    }
  }
  rc = ensure_scratch(c, padded<uint32_t>(nrows + 1) + padded<uint32_t>(nnz) + padded<float>(nnz) + padded<float>(nweights) +
                             2 * padded<int>(npos) + padded<float>(nrows));
  if (rc) return rc;
  Carver cv(c->scratch);
  uint32_t* d_off = cv.take<uint32_t>(nrows + 1);
  uint32_t* d_idx = cv.take<uint32_t>(std::max<size_t>(nnz, 1));
  float* d_val = cv.take<float>(std::max<size_t>(nnz, 1));
  float* d_w = cv.take<float>(std::max<size_t>(nweights, 1));
  int* d_wp = cv.take<int>(std::max<size_t>(npos, 1));
  int* d_vp = cv.take<int>(std::max<size_t>(npos, 1));
  float* d_pred = cv.take<float>(nrows);
  hipStream_t s = c->stream;
  DFH_HIP(hipMemcpyAsync(d_off, off32.data(), (nrows + 1) * 4, hipMemcpyHostToDevice, s));
  if (nnz) DFH_HIP(hipMemcpyAsync(d_idx, index + base, nnz * 4, hipMemcpyHostToDevice, s));
  if (nnz && value) DFH_HIP(hipMemcpyAsync(d_val, value + base, nnz * 4, hipMemcpyHostToDevice, s));
  if (nweights) DFH_HIP(hipMemcpyAsync(d_w, weights, nweights * 4, hipMemcpyHostToDevice, s));
  if (w_pos) {
    DFH_HIP(hipMemcpyAsync(d_wp, w_pos, npos * 4, hipMemcpyHostToDevice, s));
    if (V_pos) DFH_HIP(hipMemcpyAsync(d_vp, V_pos, npos * 4, hipMemcpyHostToDevice, s));
  }
  DFH_HIP(hipMemcpyAsync(d_pred, pred, nrows * 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_predict_generic, dim3(grid_for_waves(nrows, c)), dim3(256), 0, s, (uint32_t)nrows, d_off, d_idx,
                     value ? d_val : (const float*)nullptr, d_w, w_pos ? d_wp : (const int*)nullptr,
                     (w_pos && V_pos) ? d_vp : (const int*)nullptr, V_dim, d_pred, (float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
  DFH_HIP(hipGetLastError());
  DFH_HIP(hipMemcpyAsync(pred, d_pred, nrows * 4, hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  return DFH_OK;
}

int dfh_fm_calcgrad(dfh_ctx* c, int V_dim, size_t nrows, const size_t* offset, const uint32_t* index, const float* value,
                    const float* label, const float* weights, size_t nweights, const int* w_pos, const int* V_pos,
                    size_t npos, const float* pred, float* grad) {
  DFH_ARG(c && V_dim >= 0, "dfh_fm_calcgrad: bad argument");
  if (nrows == 0) return DFH_OK;
  DFH_ARG(offset && pred && weights && grad && label, "dfh_fm_calcgrad: NULL argument");
  DFH_ARG(V_dim == 0 || (w_pos && V_pos), "dfh_fm_calcgrad: V_dim > 0 needs w_pos and V_pos");
  DFH_HIP(hipSetDevice(c->device));
  std::vector<uint32_t> off32;
  int rc = to_u32_offsets(offset, nrows, &off32);
  if (rc) return rc;
  const size_t base = offset[0];
  const size_t nnz = off32[nrows];
  const size_t ncols = w_pos ? npos : nweights;
  // column-major view (stable counting sort by column => ascending rows per column,
  // the order SpMV/SpMM::TransTimes accumulate in: spmv.h:152-168, spmm.h:137-156)
  std::vector<uint32_t> col_ptr(ncols + 1, 0), s_row(std::max<size_t>(nnz, 1));
  std::vector<float> s_val(value ? std::max<size_t>(nnz, 1) : 0);
  for (size_t j = 0; j < nnz; ++j) {
    DFH_ARG(index[base + j] < ncols, "dfh_fm_calcgrad: index out of range");
    ++col_ptr[index[base + j] + 1];
  }
  for (size_t u = 0; u < ncols; ++u) col_ptr[u + 1] += col_ptr[u];
  {
    std::vector<uint32_t> fill(col_ptr.begin(), col_ptr.end() - 1);
    for (size_t i = 0; i < nrows; ++i) {
      for (uint32_t j = off32[i]; j < off32[i + 1]; ++j) {
        uint32_t q = fill[index[base + j]]++;
        s_row[q] = (uint32_t)i;
        if (value) s_val[q] = value[base + j];
      }
    }
  }
  if (w_pos) {
    for (size_t u = 0; u < npos; ++u) {
      DFH_ARG(w_pos[u] == -1 || (w_pos[u] >= 0 && (size_t)w_pos[u] < nweights), "w_pos out of range");
      if (V_dim > 0) DFH_ARG(V_pos[u] == -1 || (V_pos[u] >= 0 && (size_t)V_pos[u] + V_dim <= nweights), "V_pos out of range");
    }
  }
  const size_t xvn = nrows * (size_t)std::max(V_dim, 1);
  rc = ensure_scratch(c, padded<uint32_t>(nrows + 1) + 2 * padded<uint32_t>(nnz) + 2 * padded<float>(nnz) +
                             2 * padded<float>(nweights) + 2 * padded<int>(npos) + 3 * padded<float>(nrows) +
                             padded<float>(xvn) + padded<uint32_t>(ncols + 1));
  if (rc) return rc;
  Carver cv(c->scratch);
  uint32_t* d_off = cv.take<uint32_t>(nrows + 1);
  uint32_t* d_idx = cv.take<uint32_t>(std::max<size_t>(nnz, 1));
  float* d_val = cv.take<float>(std::max<size_t>(nnz, 1));
  uint32_t* d_srow = cv.take<uint32_t>(std::max<size_t>(nnz, 1));
  float* d_sval = cv.take<float>(std::max<size_t>(nnz, 1));
  uint32_t* d_colptr = cv.take<uint32_t>(ncols + 1);
  float* d_w = cv.take<float>(std::max<size_t>(nweights, 1));
  float* d_grad = cv.take<float>(std::max<size_t>(nweights, 1));
  int* d_wp = cv.take<int>(std::max<size_t>(npos, 1));
  int* d_vp = cv.take<int>(std::max<size_t>(npos, 1));
  float* d_pred = cv.take<float>(nrows);
  float* d_label = cv.take<float>(nrows);
  float* d_slope = cv.take<float>(nrows);
  float* d_xv = cv.take<float>(xvn);
  hipStream_t s = c->stream;
  DFH_HIP(hipMemcpyAsync(d_off, off32.data(), (nrows + 1) * 4, hipMemcpyHostToDevice, s));
  if (nnz) {
    DFH_HIP(hipMemcpyAsync(d_idx, index + base, nnz * 4, hipMemcpyHostToDevice, s));
    DFH_HIP(hipMemcpyAsync(d_srow, s_row.data(), nnz * 4, hipMemcpyHostToDevice, s));
    if (value) {
      DFH_HIP(hipMemcpyAsync(d_val, value + base, nnz * 4, hipMemcpyHostToDevice, s));
      DFH_HIP(hipMemcpyAsync(d_sval, s_val.data(), nnz * 4, hipMemcpyHostToDevice, s));
    }
  }
  DFH_HIP(hipMemcpyAsync(d_colptr, col_ptr.data(), (ncols + 1) * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(d_w, weights, nweights * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(d_grad, grad, nweights * 4, hipMemcpyHostToDevice, s));
  if (w_pos) {
    DFH_HIP(hipMemcpyAsync(d_wp, w_pos, npos * 4, hipMemcpyHostToDevice, s));
    if (V_pos) DFH_HIP(hipMemcpyAsync(d_vp, V_pos, npos * 4, hipMemcpyHostToDevice, s));
  }
  DFH_HIP(hipMemcpyAsync(d_pred, pred, nrows * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(d_label, label, nrows * 4, hipMemcpyHostToDevice, s));
  const float* dv = value ? d_val : nullptr;
  const int* dwp = w_pos ? d_wp : nullptr;
  const int* dvp = (w_pos && V_pos) ? d_vp : nullptr;
  // pass A: XV = X*V (what Predict left in XV_, fm_loss.h:81-83) and the slope p (fm_loss.h:157-161)
  hipLaunchKernelGGL(k_predict_generic, dim3(grid_for_waves(nrows, c)), dim3(256), 0, s, (uint32_t)nrows, d_off, d_idx, dv,
                     d_w, dwp, dvp, V_dim, (float*)nullptr, V_dim > 0 ? d_xv : (float*)nullptr, d_label, d_pred, d_slope);
  // pass B: per-column accumulation
  hipLaunchKernelGGL(k_calcgrad_generic, dim3(grid_for_waves(ncols, c)), dim3(256), 0, s, (uint32_t)ncols, d_colptr,
                     d_srow, value ? d_sval : (const float*)nullptr, d_w, dwp, dvp, V_dim, d_slope, d_xv, d_grad);
  DFH_HIP(hipGetLastError());
  DFH_HIP(hipMemcpyAsync(grad, d_grad, nweights * 4, hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  return DFH_OK;
}

int dfh_loss_evaluate(dfh_ctx* c, const float* label, const float* pred, size_t n, float* objv) {
  DFH_ARG(c && objv, "NULL argument");
  *objv = 0;
  if (n == 0) return DFH_OK;
  DFH_ARG(label && pred, "NULL argument");
  DFH_HIP(hipSetDevice(c->device));
  int rc = ensure_scratch(c, 2 * padded<float>(n) + 512);
  if (rc) return rc;
  Carver cv(c->scratch);
  float* d_l = cv.take<float>(n);
  float* d_p = cv.take<float>(n);
  double* d_o = cv.take<double>(1);
  hipStream_t s = c->stream;
  DFH_HIP(hipMemcpyAsync(d_l, label, n * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(d_p, pred, n * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemsetAsync(d_o, 0, sizeof(double), s));
  hipLaunchKernelGGL(k_logloss, dim3(grid_for_threads(n, c)), dim3(256), 0, s, d_l, d_p, (uint32_t)n, d_o);
  DFH_HIP(hipGetLastError());
  double o = 0;
  DFH_HIP(hipMemcpyAsync(&o, d_o, sizeof(double), hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  *objv = (float)o;
  return DFH_OK;
}

// ------------------------------------------------------------------ batches
namespace {
// several batch objects carved from ONE device allocation (dfh_batch_create_many): the last one destroyed frees it
struct SharedArena {
  void* base = nullptr;
  int refs = 0;
};
}  // namespace

// shared != NULL: the object's arrays are carved at shared->base + at (sized by a first call with size_only = true)
static int batch_create_impl(dfh_ctx* c, size_t max_rows, size_t max_nnz, dfh_batch** out, SharedArena* shared, size_t at,
                             size_t* size_only, bool sync) {
  DFH_ARG(c && (out || size_only) && max_rows >= 1 && max_nnz >= 1, "dfh_batch_create: bad argument");
  DFH_ARG(max_nnz < 0xFFFFFFF0ULL && max_rows < 0xFFFFFFF0ULL, "batch too large for 32-bit positions");
  DFH_HIP(hipSetDevice(c->device));
  dfh_batch* b = new (std::nothrow) dfh_batch();
  DFH_ARG(b != nullptr, "out of host memory");
  b->ctx = c;
  b->max_rows = max_rows;
  b->max_nnz = max_nnz;
  const size_t N = max_nnz, B = max_rows;
  size_t sort_bytes = 0, scan_bytes = 0;
  rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                            (uint32_t*)nullptr, N, 0, 64, c->stream);
  rocprim::inclusive_scan(nullptr, scan_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, N, rocprim::plus<uint32_t>(), c->stream);
  size_t auc_bytes = 0;
  rocprim::radix_sort_pairs(nullptr, auc_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                            (uint32_t*)nullptr, B, 0, 32, c->stream);
  b->temp_bytes = std::max(std::max(sort_bytes, scan_bytes), auc_bytes) + 256;
  // ONE device allocation per batch object, carved below (256 B aligned pieces).  Until round 5 every array was its own
  // hipMalloc — 61 of them, ~40 us each: the worker loop's twelve objects cost a job ~35 ms of its ~65 ms start-up.  Two passes
  // over the same list: the first adds the sizes up, the second hands the pieces out.
  char* arena_at = nullptr;
  size_t arena_need = 0;
  bool sizing = true;
#define DFH_ALLOC(ptr, count, type)                                                        \
  do {                                                                                     \
    const size_t bytes__ = (((size_t)(count) * sizeof(type)) + 255) & ~(size_t)255;        \
    if (sizing) {                                                                          \
      arena_need += bytes__;                                                               \
    } else {                                                                               \
      (ptr) = reinterpret_cast<type*>(arena_at);                                           \
      arena_at += bytes__;                                                                 \
    }                                                                                      \
  } while (0)
  for (int pass = 0; pass < 2; ++pass) {
  sizing = pass == 0;
  if (!sizing) {
    if (size_only) {   // the caller only asked how much one object takes
      *size_only = arena_need;
      delete b;
      return DFH_OK;
    }
    if (shared) {
      b->shared_arena = shared;
      ++shared->refs;
      arena_at = static_cast<char*>(shared->base) + at;
    } else {
      hipError_t e__ = hipMalloc(reinterpret_cast<void**>(&b->arena), arena_need);
      if (e__ != hipSuccess) {
        set_error(std::string("dfh_batch_create: hipMalloc: ") + hipGetErrorString(e__));
        dfh_batch_destroy(b);
        return DFH_ERR_HIP;
      }
      arena_at = static_cast<char*>(b->arena);
    }
  }
  DFH_ALLOC(b->o_raw, N, uint64_t);
  DFH_ALLOC(b->o_offset, B + 1, uint32_t);
  DFH_ALLOC(b->o_value, N, float);
  DFH_ALLOC(b->o_label, B, float);
  b->d_raw = b->o_raw;
  b->d_offset = b->o_offset;
  b->d_value = b->o_value;
  b->d_label = b->o_label;
  DFH_ALLOC(b->d_keys, N, uint64_t);
  DFH_ALLOC(b->d_skeys, N, uint64_t);
  DFH_ALLOC(b->d_pos, N, uint32_t);
  DFH_ALLOC(b->d_spos, N, uint32_t);
  DFH_ALLOC(b->d_bpos, N, uint32_t);
  DFH_ALLOC(b->d_head, N, uint32_t);
  DFH_ALLOC(b->d_uid, N, uint32_t);
  DFH_ALLOC(b->d_temp, b->temp_bytes, char);
  b->max_tiles = (N + LOC_TILE - 1) / LOC_TILE;
  DFH_ALLOC(b->d_spl_key, LOC_BIG_BUCKETS, uint64_t);
  DFH_ALLOC(b->d_spl_pos, LOC_BIG_BUCKETS, uint32_t);
  DFH_ALLOC(b->d_smp_key, LOC_BIG_BUCKETS * LOC_OVERSAMPLE, uint64_t);
  DFH_ALLOC(b->d_smp_pos, LOC_BIG_BUCKETS * LOC_OVERSAMPLE, uint32_t);
  DFH_ALLOC(b->d_smp_rank, LOC_BIG_BUCKETS * LOC_OVERSAMPLE, uint32_t);
  DFH_ALLOC(b->d_first_key, LOC_BIG_BUCKETS, uint64_t);
  DFH_ALLOC(b->d_last_key, LOC_BIG_BUCKETS, uint64_t);
  DFH_ALLOC(b->d_packed, N, uint32_t);
  DFH_ALLOC(b->d_run_off, b->max_tiles * LOC_BIG_BUCKETS, uint32_t);
  DFH_ALLOC(b->d_bstart, LOC_BIG_BUCKETS + 1, uint32_t);
  DFH_ALLOC(b->d_btotal, LOC_XCDS * LOC_BIG_BUCKETS, uint32_t);
  DFH_ALLOC(b->d_nheads, LOC_BIG_BUCKETS, uint32_t);
  DFH_ALLOC(b->d_lh, LOC_BIG_BUCKETS, uint32_t);
  DFH_ALLOC(b->d_feaids, N, uint64_t);
  DFH_ALLOC(b->d_feacnt, N, float);
  DFH_ALLOC(b->d_col_ptr, N + 1, uint32_t);
  DFH_ALLOC(b->d_index, N, uint32_t);
  DFH_ALLOC(b->d_s_row, N, uint32_t);
  DFH_ALLOC(b->d_s_val, N, float);
  DFH_ALLOC(b->d_U, 64, uint32_t);
  DFH_ALLOC(b->d_urow, N, uint32_t);
  DFH_ALLOC(b->d_uw, N, uint2);
  // bucket q of the sample sort may list n_q / 9 + 2 mid and n_q / 257 + 2 hot keys (k_loc_emit)
  DFH_ALLOC(b->d_mid_ent, N / (BWD_SMALL + 1) + 2 * LOC_BIG_BUCKETS + 16, SegEnt);
  DFH_ALLOC(b->d_hot_ent, N / (BWD_MID + 1) + 2 * LOC_BIG_BUCKETS + 16, SegEnt);
  DFH_ALLOC(b->d_few_ent, N / 2 + 2 * LOC_BIG_BUCKETS + 16, SegEnt);
  b->split_cap = 2 * (N / HOT_SPLIT) + 16;   // >= the sum over the segments longer than HOT_SPLIT_MIN of ceil(len / HOT_SPLIT)
  DFH_ALLOC(b->d_split_ent, b->split_cap, SegEnt);
  DFH_ALLOC(b->d_split_ticket, b->split_cap, uint32_t);
  DFH_ALLOC(b->d_split_part, b->split_cap * UPD_SPLIT_STRIDE, float);
  DFH_ALLOC(b->d_mid, LOC_BIG_BUCKETS, uint2);
  DFH_ALLOC(b->d_hot, LOC_BIG_BUCKETS, uint2);
  DFH_ALLOC(b->d_few, LOC_BIG_BUCKETS, uint2);
  DFH_ALLOC(b->d_need, N, uint32_t);
  DFH_ALLOC(b->d_rank, N, uint32_t);
  DFH_ALLOC(b->d_pred, B, float);
  DFH_ALLOC(b->d_slope, B, float);
  DFH_ALLOC(b->d_prog, 2 * PROG_SLOTS + 64, double);
  DFH_ALLOC(b->d_auc_keys, B, uint32_t);
  DFH_ALLOC(b->d_auc_skeys, B, uint32_t);
  DFH_ALLOC(b->d_auc_lab, B, uint32_t);
  DFH_ALLOC(b->d_auc_slab, B, uint32_t);
  DFH_ALLOC(b->d_auc_part, AUC_PART_WORDS, uint32_t);
  }  // sizing pass, carving pass
#undef DFH_ALLOC
  b->arena_bytes = arena_need;
  b->d_total = b->d_U + 1;
  // ev_ready / ev_free order streams of ONE device: no system-scope fence (cache write-back + invalidate) at the record
  const unsigned evf = hipEventDisableTiming | (c->event_flags ? hipEventDisableSystemFence : 0u);
  // (a failure from here on gives the object — and its reference on a shared arena — back: ADVICE r5)
  hipError_t e2 = hipEventCreateWithFlags(&b->ev_ready, evf);
  if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&b->ev_free, evf);
  if (e2 == hipSuccess) e2 = hipMemsetAsync(b->d_prog, 0, (2 * PROG_SLOTS + 64) * sizeof(double), c->stream);
  if (e2 == hipSuccess) e2 = hipMemsetAsync(b->d_U, 0, 64 * sizeof(uint32_t), c->stream);
  // a key's ticket is back at zero when its last part has been applied (upd_split_role)
  if (e2 == hipSuccess) e2 = hipMemsetAsync(b->d_split_ticket, 0, b->split_cap * sizeof(uint32_t), c->stream);
  // k_loc_sort keeps the bucket totals zero between calls
  if (e2 == hipSuccess) e2 = hipMemsetAsync(b->d_btotal, 0, LOC_XCDS * LOC_BIG_BUCKETS * sizeof(uint32_t), c->stream);
  // row ids are written by the lookups of the keys a step resolves; anything else must never be used as one:
  // all-ones makes a stray use fault at once instead of reading some row
  if (e2 == hipSuccess) e2 = hipMemsetAsync(b->d_urow, 0xFF, N * sizeof(uint32_t), c->stream);
  if (e2 == hipSuccess) e2 = hipMemsetAsync(b->d_uw, 0xFF, N * sizeof(uint2), c->stream);
  if (e2 == hipSuccess && sync) e2 = hipStreamSynchronize(c->stream);
  if (e2 != hipSuccess) {
    set_error(std::string("dfh_batch_create: ") + hipGetErrorString(e2));
    dfh_batch_destroy(b);
    return DFH_ERR_HIP;
  }
  *out = b;
  return DFH_OK;
}

int dfh_batch_create(dfh_ctx* c, size_t max_rows, size_t max_nnz, dfh_batch** out) {
  return batch_create_impl(c, max_rows, max_nnz, out, nullptr, 0, nullptr, true);
}

int dfh_batch_create_many(dfh_ctx* c, int n, size_t max_rows, size_t max_nnz, dfh_batch** out) {
  DFH_ARG(c && out && n >= 1 && n <= 64, "dfh_batch_create_many: 1 <= n <= 64 objects");
  size_t each = 0;
  int rc = batch_create_impl(c, max_rows, max_nnz, nullptr, nullptr, 0, &each, false);
  if (rc) return rc;
  each = (each + 4095) & ~(size_t)4095;
  SharedArena* sa = new (std::nothrow) SharedArena();
  DFH_ARG(sa != nullptr, "out of host memory");
  DFH_HIP(hipSetDevice(c->device));
  if (hipMalloc(&sa->base, each * (size_t)n) != hipSuccess) {
    delete sa;
    set_error("dfh_batch_create_many: hipMalloc failed");
    return DFH_ERR_HIP;
  }
  for (int i = 0; i < n; ++i) out[i] = nullptr;
  sa->refs = 1;  // this call's own reference: the arena outlives every failure path below
  for (int i = 0; i < n && !rc; ++i) rc = batch_create_impl(c, max_rows, max_nnz, &out[i], sa, each * (size_t)i, nullptr, false);
  if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) {   // the objects' memsets
    set_error("dfh_batch_create_many: hipStreamSynchronize failed");
    rc = DFH_ERR_HIP;
  }
  if (rc) {
    for (int j = 0; j < n; ++j) {
      if (out[j]) dfh_batch_destroy(out[j]);
      out[j] = nullptr;
    }
  }
  if (--sa->refs == 0) {  // no object holds it (every creation failed)
    hipFree(sa->base);
    delete sa;
  }
  return rc;
}

int dfh_batch_destroy(dfh_batch* b) {
  if (b && b->n_prof)
    fprintf(stderr, "dfh_batch_prepare_rows x %llu: begin %.4f s, wait staged %.4f, write description %.4f, gather %.4f, localize %.4f, lookup + ready %.4f\n",
            (unsigned long long)b->n_prof, b->t_prof[0], b->t_prof[1], b->t_prof[2], b->t_prof[3], b->t_prof[4], b->t_prof[5]);
  if (!b) return DFH_OK;
  hipSetDevice(b->ctx->device);
  pend_remove(b->ctx, b);  // stages noted but never queued die with the object
  sync_all(b->ctx);
  if (b->ev_ready) hipEventDestroy(b->ev_ready);
  if (b->ev_free) hipEventDestroy(b->ev_free);
  if (b->ev_staged) hipEventDestroy(b->ev_staged);
  if (b->h_stage) hipHostFree(b->h_stage);
  if (b->d_xv) hipFree(b->d_xv);      // (grows with V_dim: its own allocation, ensure_xv)
  if (b->arena) hipFree(b->arena);   // every other device array of the object
  if (b->shared_arena) {             // ... or its share of an allocation made for several objects
    SharedArena* sa = static_cast<SharedArena*>(b->shared_arena);
    if (--sa->refs == 0) {
      hipFree(sa->base);
      delete sa;
    }
  }
  delete b;
  return DFH_OK;
}

// page-locked staging of a batch object, sized for what the calling path puts there: a minibatch DESCRIBED by row numbers
// needs offsets + labels + row numbers (~160 KB), dfh_batch_load_host the ids and values too (9 MB at C3's sizes — 2.4 ms
// of hipHostMalloc each, which the worker loop's twelve batch objects paid on their first minibatch before round 4)
static int ensure_stage(dfh_batch* b, size_t need) {
  if (b->h_stage && b->stage_bytes >= need) return DFH_OK;
  if (b->h_stage) {
    if (b->staged_pending) DFH_HIP(hipEventSynchronize(b->ev_staged));
    b->staged_pending = false;
    DFH_HIP(hipHostFree(b->h_stage));
    b->h_stage = nullptr;
    b->d_stage_view = nullptr;
  }
  DFH_HIP(hipHostMalloc(reinterpret_cast<void**>(&b->h_stage), need, hipHostMallocDefault));
  b->stage_bytes = need;
  if (!b->ev_staged) DFH_HIP(hipEventCreateWithFlags(&b->ev_staged, hipEventDisableTiming));
  return DFH_OK;
}

int dfh_batch_load_host(dfh_batch* b, size_t nrows, const size_t* offset, const uint64_t* index, const float* value,
                        const float* label) {
  DFH_ARG(b && offset && label, "dfh_batch_load_host: NULL argument");
  DFH_ARG(nrows >= 1 && nrows <= b->max_rows, "dfh_batch_load_host: nrows out of range");
  std::vector<uint32_t> off32;
  int rc = to_u32_offsets(offset, nrows, &off32);
  if (rc) return rc;
  const size_t base = offset[0], nnz = off32[nrows];
  DFH_ARG(nnz <= b->max_nnz, "dfh_batch_load_host: nnz exceeds max_nnz");
  DFH_ARG(nnz == 0 || index, "index is NULL");
  DFH_HIP(hipSetDevice(b->ctx->device));
  phase_begin(b);
  rc = prep_begin(b);
  if (rc) return rc;
  hipStream_t s = prep_of(b);
  b->d_raw = b->o_raw; b->d_offset = b->o_offset; b->d_value = b->o_value; b->d_label = b->o_label;
  // Pinned staging: the caller's (pageable) arrays are copied into page-locked memory owned by the
  // batch object and sent with asynchronous copies on the preparation stream — the call returns as
  // soon as the bytes are staged, the caller may reuse its arrays, and with two batch objects used
  // alternately the transfer of minibatch t+1 overlaps the training of minibatch t.
  const size_t o_off = 0, o_lab = (b->max_rows + 1) * 4, o_idx = ((o_lab + b->max_rows * 4 + 255) & ~(size_t)255),
               o_val = o_idx + b->max_nnz * 8, total = o_val + b->max_nnz * 4;
  rc = ensure_stage(b, total);
  if (rc) return rc;
  if (b->staged_pending) {  // the previous minibatch staged here has long left; this wait is a formality
    // (a query first: hipEventSynchronize costs ~100 us of host time even on an event that completed long ago)
    if (hipEventQuery(b->ev_staged) != hipSuccess) DFH_HIP(hipEventSynchronize(b->ev_staged));
    b->staged_pending = false;
  }
  memcpy(b->h_stage + o_off, off32.data(), (nrows + 1) * 4);
  memcpy(b->h_stage + o_lab, label, nrows * 4);
  if (nnz) memcpy(b->h_stage + o_idx, index + base, nnz * 8);
  if (nnz && value) memcpy(b->h_stage + o_val, value + base, nnz * 4);
  DFH_HIP(hipMemcpyAsync(b->d_offset, b->h_stage + o_off, (nrows + 1) * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(b->d_label, b->h_stage + o_lab, nrows * 4, hipMemcpyHostToDevice, s));
  if (nnz) DFH_HIP(hipMemcpyAsync(b->d_raw, b->h_stage + o_idx, nnz * 8, hipMemcpyHostToDevice, s));
  if (nnz && value) DFH_HIP(hipMemcpyAsync(b->d_value, b->h_stage + o_val, nnz * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipEventRecord(b->ev_staged, s));
  b->staged_pending = true;
  b->nrows = nrows;
  b->nnz = nnz;
  b->has_value = value != nullptr;
  b->has_cnt = false;
  b->localized = false;
  b->looked_up = nullptr;
  return DFH_OK;
}

int dfh_batch_load_device(dfh_batch* b, size_t nrows, size_t nnz, const uint32_t* d_offset, const uint64_t* d_index,
                          const float* d_value, const float* d_label) {
  DFH_ARG(b && d_offset && d_label && (nnz == 0 || d_index), "dfh_batch_load_device: NULL argument");
  DFH_ARG(nrows >= 1 && nrows <= b->max_rows && nnz <= b->max_nnz, "dfh_batch_load_device: shape out of range");
  phase_begin(b);
  {
    int rc = prep_begin(b);
    if (rc) return rc;
  }
  hipStream_t s = prep_of(b);
  b->d_raw = b->o_raw; b->d_offset = b->o_offset; b->d_value = b->o_value; b->d_label = b->o_label;
  DFH_HIP(hipMemcpyAsync(b->d_offset, d_offset, (nrows + 1) * 4, hipMemcpyDeviceToDevice, s));
  if (nnz) DFH_HIP(hipMemcpyAsync(b->d_raw, d_index, nnz * 8, hipMemcpyDeviceToDevice, s));
  if (nnz && d_value) DFH_HIP(hipMemcpyAsync(b->d_value, d_value, nnz * 4, hipMemcpyDeviceToDevice, s));
  DFH_HIP(hipMemcpyAsync(b->d_label, d_label, nrows * 4, hipMemcpyDeviceToDevice, s));
  b->nrows = nrows;
  b->nnz = nnz;
  b->has_value = d_value != nullptr;
  b->has_cnt = false;
  b->localized = false;
  b->looked_up = nullptr;
  return DFH_OK;
}

int dfh_batch_attach_device(dfh_batch* b, size_t nrows, size_t nnz, const uint32_t* d_offset, const uint64_t* d_index,
                            const float* d_value, const float* d_label) {
  DFH_ARG(b && d_offset && d_label && (nnz == 0 || d_index), "dfh_batch_attach_device: NULL argument");
  DFH_ARG(nrows >= 1 && nrows <= b->max_rows && nnz <= b->max_nnz, "dfh_batch_attach_device: shape out of range");
  phase_begin(b);
  {
    int rc = prep_begin(b);  // a queued step may still read the previously attached memory; nothing of ours is overwritten
    if (rc) return rc;
  }
  b->d_offset = const_cast<uint32_t*>(d_offset);
  b->d_raw = const_cast<uint64_t*>(d_index);
  b->d_value = d_value ? const_cast<float*>(d_value) : b->o_value;
  b->d_label = const_cast<float*>(d_label);
  b->nrows = nrows;
  b->nnz = nnz;
  b->has_value = d_value != nullptr;
  b->has_cnt = false;
  b->localized = false;
  b->looked_up = nullptr;
  return DFH_OK;
}


// ---------------------------------------------------------------------------------------
// Device feed: the shuffle buffer of BatchReader (src/reader/batch_reader.cc:38-52) lives in HBM; a minibatch is gathered
// out of it by row number on the device (the host sends 4 B per row instead of copying ~300 B per row twice).
// ---------------------------------------------------------------------------------------
struct dfh_rowbuf {
  dfh_ctx* ctx = nullptr;
  size_t max_rows = 0, max_nnz = 0, nrows = 0, nnz = 0;
  uint32_t* d_off = nullptr;   // [max_rows + 1]
  uint64_t* d_idx = nullptr;   // [max_nnz]
  float* d_val = nullptr;      // [max_nnz]
  bool has_value = false;
  hipStream_t up = nullptr;    // uploads: the feed thread's own stream
  hipEvent_t ev_loaded = nullptr;
  // one "gathered" event per stream that has gathered out of this buffer (the two batch objects of a worker loop gather on
  // different preparation streams: one shared event would only remember the LAST gather); `pending` marks the ones
  // recorded since the last upload.  Set by the thread that gathers, read by the thread that uploads.
  struct Used { hipStream_t stream; hipEvent_t ev; bool pending; };
  std::mutex mu;
  std::vector<Used> used;
  std::vector<hipStream_t> seen_loaded;  // streams ordered behind the current upload already (one wait per stream and upload)
  double t_prof[3] = {0, 0, 0};          // DFH_PROFILE_PREP: host seconds of dfh_rowbuf_load_host (offsets, queue, wait)
  uint64_t n_prof = 0, bytes_prof = 0;
  std::vector<uint32_t> off32;
};

namespace {
// one wave per row of the minibatch: row rows[q] of the buffer -> positions dst_off[q] .. of the minibatch's arrays
__global__ void __launch_bounds__(256) k_gather_rows(const uint32_t* __restrict__ src_off, const uint64_t* __restrict__ src_idx,
                                                     const float* __restrict__ src_val, const uint32_t* __restrict__ rows, uint32_t n,
                                                     const uint32_t* __restrict__ dst_off, uint64_t* __restrict__ dst_idx,
                                                     float* __restrict__ dst_val) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; q < n; q += nw) {
    const uint32_t r = rows[q];
    const uint32_t lo = src_off[r], len = src_off[r + 1] - lo, d0 = dst_off[q];
    for (uint32_t j = lane; j < len; j += 64u) {
      dst_idx[d0 + j] = src_idx[lo + j];
      if (dst_val) dst_val[d0 + j] = src_val ? src_val[lo + j] : 1.0f;   // a buffer without values holds ones
    }
  }
}
}  // namespace

int dfh_rowbuf_create(dfh_ctx* c, size_t max_rows, size_t max_nnz, dfh_rowbuf** out) {
  DFH_ARG(c && out && max_rows >= 1 && max_nnz >= 1, "dfh_rowbuf_create: bad argument");
  DFH_ARG(max_nnz < 0xFFFFFFF0ULL && max_rows < 0xFFFFFFF0ULL, "dfh_rowbuf_create: a row buffer holds fewer than 2^32 rows / nonzeros");
  DFH_HIP(hipSetDevice(c->device));
  dfh_rowbuf* rb = new (std::nothrow) dfh_rowbuf();
  if (!rb) {
    set_error("dfh_rowbuf_create: out of host memory");
    return DFH_ERR_HIP;
  }
  rb->ctx = c;
  rb->max_rows = max_rows;
  rb->max_nnz = max_nnz;
  hipError_t e;
  if ((e = hipMalloc(reinterpret_cast<void**>(&rb->d_off), (max_rows + 1) * sizeof(uint32_t))) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&rb->d_idx), max_nnz * sizeof(uint64_t))) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&rb->d_val), max_nnz * sizeof(float))) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&rb->up, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&rb->ev_loaded, hipEventDisableTiming)) != hipSuccess) {
    set_error(std::string("dfh_rowbuf_create: ") + hipGetErrorString(e));
    dfh_rowbuf_destroy(rb);
    return DFH_ERR_HIP;
  }
  *out = rb;
  return DFH_OK;
}

int dfh_rowbuf_destroy(dfh_rowbuf* rb) {
  if (!rb) return DFH_OK;
  if (rb->n_prof)
    fprintf(stderr, "dfh_rowbuf_load_host x %llu (%.1f MB): offsets %.4f s, queue copies %.4f, wait %.4f\n", (unsigned long long)rb->n_prof,
            rb->bytes_prof / 1e6, rb->t_prof[0], rb->t_prof[1], rb->t_prof[2]);
  hipSetDevice(rb->ctx->device);
  // the gathers out of this buffer, wherever they were queued (NOT sync_all: this may run beside the thread that drives
  // the context, and only this buffer's own consumers matter)
  for (auto& u : rb->used) {
    if (u.pending) hipEventSynchronize(u.ev);
    hipEventDestroy(u.ev);
  }
  if (rb->up) {
    hipStreamSynchronize(rb->up);
    hipStreamDestroy(rb->up);
  }
  if (rb->ev_loaded) hipEventDestroy(rb->ev_loaded);
  for (void* p : {(void*)rb->d_off, (void*)rb->d_idx, (void*)rb->d_val})
    if (p) hipFree(p);
  delete rb;
  return DFH_OK;
}

int dfh_rowbuf_load_host(dfh_rowbuf* rb, size_t nrows, const size_t* offset, const uint64_t* index, const float* value) {
  DFH_ARG(rb && offset && nrows >= 1 && nrows <= rb->max_rows, "dfh_rowbuf_load_host: bad argument / more rows than the buffer holds");
  const size_t base = offset[0], nnz = offset[nrows] - base;
  DFH_ARG(nnz <= rb->max_nnz, "dfh_rowbuf_load_host: more nonzeros than the buffer holds");
  DFH_ARG(nnz == 0 || index, "dfh_rowbuf_load_host: index is NULL");
  DFH_HIP(hipSetDevice(rb->ctx->device));
  {
    // every gather out of the previous contents, on whichever stream it was queued, precedes the copies below
    std::lock_guard<std::mutex> lk(rb->mu);
    for (auto& u : rb->used) {
      if (!u.pending) continue;
      DFH_HIP(hipStreamWaitEvent(rb->up, u.ev, 0));
      u.pending = false;
    }
    rb->seen_loaded.clear();
  }
  static const bool prof = getenv("DFH_PROFILE_PREP") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp = prof ? now() : 0;
  rb->off32.resize(nrows + 1);
  for (size_t i = 0; i <= nrows; ++i) {
    DFH_ARG(offset[i] >= base && (i == 0 || offset[i] >= offset[i - 1]), "dfh_rowbuf_load_host: offsets must not decrease");
    rb->off32[i] = (uint32_t)(offset[i] - base);
  }
  if (prof) { const double x = now(); rb->t_prof[0] += x - tp; tp = x; }
  DFH_HIP(hipMemcpyAsync(rb->d_off, rb->off32.data(), (nrows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, rb->up));
  if (nnz) DFH_HIP(hipMemcpyAsync(rb->d_idx, index + base, nnz * sizeof(uint64_t), hipMemcpyHostToDevice, rb->up));
  if (nnz && value) DFH_HIP(hipMemcpyAsync(rb->d_val, value + base, nnz * sizeof(float), hipMemcpyHostToDevice, rb->up));
  DFH_HIP(hipEventRecord(rb->ev_loaded, rb->up));
  if (prof) { const double x = now(); rb->t_prof[1] += x - tp; tp = x; }
  DFH_HIP(hipStreamSynchronize(rb->up));   // the caller's arrays are free again
  if (prof) { rb->t_prof[2] += now() - tp; ++rb->n_prof; rb->bytes_prof += nnz * (value ? 12 : 8) + nrows * 4; }
  rb->nrows = nrows;
  rb->nnz = nnz;
  rb->has_value = value != nullptr;
  return DFH_OK;
}

namespace {
__global__ void k_fill_f32(float* __restrict__ p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
}  // namespace

// dfh_rowbuf_load_host for a buffer that was never assembled on the host: `offset` [nrows + 1] are the buffer's own
// (cumulative) offsets, the ids / values arrive as `nslices` pieces that follow one another (slice g: nnz[g] ids at
// index[g], values at value[g] or NULL = all ones).  One copy per piece, straight out of the caller's arrays.
int dfh_rowbuf_load_host_slices(dfh_rowbuf* rb, size_t nrows, const size_t* offset, int nslices, const uint64_t* const* index,
                                const float* const* value, const size_t* nnz_of) {
  DFH_ARG(rb && offset && nrows >= 1 && nrows <= rb->max_rows, "dfh_rowbuf_load_host_slices: bad argument / more rows than the buffer holds");
  DFH_ARG(nslices >= 0 && (nslices == 0 || (index && value && nnz_of)), "dfh_rowbuf_load_host_slices: NULL slice arrays");
  const size_t base = offset[0], nnz = offset[nrows] - base;
  DFH_ARG(nnz <= rb->max_nnz, "dfh_rowbuf_load_host_slices: more nonzeros than the buffer holds");
  size_t total = 0;
  bool any_value = false;
  for (int g = 0; g < nslices; ++g) {
    DFH_ARG(nnz_of[g] == 0 || index[g], "dfh_rowbuf_load_host_slices: a slice without ids");
    total += nnz_of[g];
    any_value = any_value || (nnz_of[g] && value[g]);
  }
  DFH_ARG(total == nnz, "dfh_rowbuf_load_host_slices: the slices must hold the buffer's nonzeros");
  DFH_HIP(hipSetDevice(rb->ctx->device));
  {
    std::lock_guard<std::mutex> lk(rb->mu);
    for (auto& u : rb->used) {
      if (!u.pending) continue;
      DFH_HIP(hipStreamWaitEvent(rb->up, u.ev, 0));
      u.pending = false;
    }
    rb->seen_loaded.clear();
  }
  static const bool prof = getenv("DFH_PROFILE_PREP") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp = prof ? now() : 0;
  rb->off32.resize(nrows + 1);
  for (size_t i = 0; i <= nrows; ++i) {
    DFH_ARG(offset[i] >= base && (i == 0 || offset[i] >= offset[i - 1]), "dfh_rowbuf_load_host_slices: offsets must not decrease");
    rb->off32[i] = (uint32_t)(offset[i] - base);
  }
  if (prof) { const double x = now(); rb->t_prof[0] += x - tp; tp = x; }
  DFH_HIP(hipMemcpyAsync(rb->d_off, rb->off32.data(), (nrows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, rb->up));
  size_t at = 0;
  for (int g = 0; g < nslices; ++g) {
    const size_t n = nnz_of[g];
    if (!n) continue;
    DFH_HIP(hipMemcpyAsync(rb->d_idx + at, index[g], n * sizeof(uint64_t), hipMemcpyHostToDevice, rb->up));
    if (any_value) {
      if (value[g]) {
        DFH_HIP(hipMemcpyAsync(rb->d_val + at, value[g], n * sizeof(float), hipMemcpyHostToDevice, rb->up));
      } else {  // a block without a value array beside blocks with one: ones (compressed_row_block.h:36-44)
        hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, rb->up, rb->d_val + at, n, 1.0f);
      }
    }
    at += n;
  }
  DFH_HIP(hipGetLastError());
  DFH_HIP(hipEventRecord(rb->ev_loaded, rb->up));
  if (prof) { const double x = now(); rb->t_prof[1] += x - tp; tp = x; }
  DFH_HIP(hipStreamSynchronize(rb->up));   // the caller's arrays are free again
  if (prof) { rb->t_prof[2] += now() - tp; ++rb->n_prof; rb->bytes_prof += nnz * (any_value ? 12 : 8) + nrows * 4; }
  rb->nrows = nrows;
  rb->nnz = nnz;
  rb->has_value = any_value;
  return DFH_OK;
}

int dfh_batch_gather_rows(dfh_batch* b, size_t nrows, const size_t* offset, const float* label, int nseg, dfh_rowbuf* const* bufs,
                          const uint32_t* const* rows, const size_t* seg_rows) {
  DFH_ARG(b && offset && label && nseg >= 1 && bufs && rows && seg_rows, "dfh_batch_gather_rows: NULL argument");
  DFH_ARG(nrows >= 1 && nrows <= b->max_rows, "dfh_batch_gather_rows: nrows out of range");
  std::vector<uint32_t> off32;
  int rc = to_u32_offsets(offset, nrows, &off32);
  if (rc) return rc;
  const size_t nnz = off32[nrows];
  DFH_ARG(nnz <= b->max_nnz, "dfh_batch_gather_rows: nnz exceeds max_nnz");
  size_t total = 0;
  bool any_value = false;
  for (int g = 0; g < nseg; ++g) {
    DFH_ARG(bufs[g] && bufs[g]->ctx == b->ctx && (seg_rows[g] == 0 || rows[g]), "dfh_batch_gather_rows: bad segment");
    total += seg_rows[g];
    any_value = any_value || bufs[g]->has_value;
  }
  DFH_ARG(total == nrows, "dfh_batch_gather_rows: the segments must hold nrows rows");
  DFH_HIP(hipSetDevice(b->ctx->device));
  phase_begin(b);
  rc = prep_begin(b);
  if (rc) return rc;
  hipStream_t s = prep_of(b);
  b->d_raw = b->o_raw; b->d_offset = b->o_offset; b->d_value = b->o_value; b->d_label = b->o_label;
  // offsets, labels and row numbers through the batch's pinned staging (the row numbers where load_host puts the ids)
  const size_t o_off = 0, o_lab = (b->max_rows + 1) * 4, o_idx = ((o_lab + b->max_rows * 4 + 255) & ~(size_t)255),
               o_val = o_idx + b->max_nnz * 8, stage_total = o_val + b->max_nnz * 4;
  (void)stage_total;   // (the layout of dfh_batch_load_host; only the head of it is used here)
  rc = ensure_stage(b, o_idx + (b->max_rows + 1) * 4);
  if (rc) return rc;
  if (b->staged_pending) {
    // (a query first: hipEventSynchronize costs ~100 us of host time even on an event that completed long ago)
    if (hipEventQuery(b->ev_staged) != hipSuccess) DFH_HIP(hipEventSynchronize(b->ev_staged));
    b->staged_pending = false;
  }
  memcpy(b->h_stage + o_off, off32.data(), (nrows + 1) * 4);
  memcpy(b->h_stage + o_lab, label, nrows * 4);
  uint32_t* h_rows = reinterpret_cast<uint32_t*>(b->h_stage + o_idx);
  size_t at = 0;
  for (int g = 0; g < nseg; ++g) {
    for (size_t i = 0; i < seg_rows[g]; ++i) {
      DFH_ARG(rows[g][i] < bufs[g]->nrows, "dfh_batch_gather_rows: row number beyond the buffer");
      h_rows[at + i] = rows[g][i];
    }
    at += seg_rows[g];
  }
  uint32_t* d_rows = b->d_pos;   // scratch until the Localizer writes its row ids there (same stream, later)
  DFH_HIP(hipMemcpyAsync(b->d_offset, b->h_stage + o_off, (nrows + 1) * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(b->d_label, b->h_stage + o_lab, nrows * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(d_rows, h_rows, nrows * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipEventRecord(b->ev_staged, s));
  b->staged_pending = true;
  at = 0;
  for (int g = 0; g < nseg; ++g) {
    dfh_rowbuf* rb = bufs[g];
    if (seg_rows[g] == 0) continue;
    DFH_HIP(hipStreamWaitEvent(s, rb->ev_loaded, 0));
    const unsigned blocks = (unsigned)std::min<size_t>((seg_rows[g] + 3) / 4, 4096);
    hipLaunchKernelGGL(k_gather_rows, dim3(blocks), dim3(256), 0, s, rb->d_off, rb->d_idx, rb->has_value ? rb->d_val : (const float*)nullptr,
                       d_rows + at, (uint32_t)seg_rows[g], b->d_offset + at, b->d_raw, any_value ? b->d_value : (float*)nullptr);
    {
      std::lock_guard<std::mutex> lk(rb->mu);
      dfh_rowbuf::Used* u = nullptr;
      for (auto& x : rb->used)
        if (x.stream == s) u = &x;
      if (!u) {
        hipEvent_t ev = nullptr;
        DFH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        rb->used.push_back({s, ev, false});
        u = &rb->used.back();
      }
      DFH_HIP(hipEventRecord(u->ev, s));
      u->pending = true;
    }
    at += seg_rows[g];
  }
  DFH_HIP(hipGetLastError());
  b->nrows = nrows;
  b->nnz = nnz;
  b->has_value = any_value;
  b->has_cnt = false;
  b->localized = false;
  b->looked_up = nullptr;
  return DFH_OK;
}

namespace {
// dfh_batch_prepare_rows: the description of the minibatch (row numbers, offsets, labels) is read where the host wrote it —
// page-locked host memory mapped into the device's address space — 256 rows per block, coalesced, and passed on: the
// minibatch's own offsets / labels land in HBM by the same kernel that gathers its rows, no copy is queued.
__global__ void __launch_bounds__(256) k_gather_rows_staged(const uint32_t* __restrict__ src_off, const uint64_t* __restrict__ src_idx,
                                                            const float* __restrict__ src_val, const uint32_t* __restrict__ h_rows,
                                                            const uint32_t* __restrict__ h_off, const float* __restrict__ h_lab, uint32_t n,
                                                            uint32_t* __restrict__ dst_off, float* __restrict__ dst_lab,
                                                            uint64_t* __restrict__ dst_idx, float* __restrict__ dst_val, int write_end) {
  // GR rows per block and pass: few enough that a minibatch spreads over the whole chip (10 000 rows = 313 blocks; 256
  // rows per block left 216 of the 256 CUs idle and took 92 us), enough that the description is read in 128 B pieces
  constexpr uint32_t GR = 32;
  __shared__ uint32_t s_lo[GR], s_len[GR], s_d0[GR];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  for (uint32_t q0 = blockIdx.x * GR; q0 < n; q0 += gridDim.x * GR) {
    const uint32_t m = min(GR, n - q0);
    __syncthreads();
    if (threadIdx.x < m) {
      const uint32_t q = q0 + threadIdx.x;
      const uint32_t r = h_rows[q], o = h_off[q];
      const uint32_t lo = src_off[r];
      s_lo[threadIdx.x] = lo;
      s_len[threadIdx.x] = src_off[r + 1] - lo;
      s_d0[threadIdx.x] = o;
      dst_off[q] = o;
      dst_lab[q] = h_lab[q];
    }
    if (threadIdx.x == 255 && write_end && q0 + m == n) dst_off[n] = h_off[n];
    __syncthreads();
    for (uint32_t t = w; t < m; t += 4u) {   // 8 rows per wave, independent addresses: the copies overlap
      const uint32_t lo = s_lo[t], len = s_len[t], d0 = s_d0[t];
      for (uint32_t j = lane; j < len; j += 64u) {
        dst_idx[d0 + j] = src_idx[lo + j];
        if (dst_val) dst_val[d0 + j] = src_val ? src_val[lo + j] : 1.0f;   // a buffer without values holds ones
      }
    }
  }
}
}  // namespace

namespace {
// after the launch(es) that read a described minibatch's rows out of their buffers have been queued on s: the buffers may be
// refilled, the page-locked description rewritten, once those launches are through
int gather_queued(dfh_batch* b, hipStream_t s) {
  for (const auto& g : b->gsegs) {
    dfh_rowbuf* rb = g.rb;
    std::lock_guard<std::mutex> lk(rb->mu);
    dfh_rowbuf::Used* u = nullptr;
    for (auto& x : rb->used)
      if (x.stream == s) u = &x;
    if (!u) {
      hipEvent_t ev = nullptr;
      DFH_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      rb->used.push_back({s, ev, false});
      u = &rb->used.back();
    }
    DFH_HIP(hipEventRecord(u->ev, s));
    u->pending = true;
  }
  DFH_HIP(hipEventRecord(b->ev_staged, s));   // the page-locked block may be rewritten once the launches have read it
  b->staged_pending = true;
  b->gather_pending = false;
  return DFH_OK;
}
// the rows of a described minibatch by launches of their own (k_gather_rows_staged): where the count pass cannot gather
int gather_alone(dfh_batch* b, hipStream_t s) {
  for (const auto& g : b->gsegs) {
    dfh_rowbuf* rb = g.rb;
    const unsigned blocks = (unsigned)std::min<size_t>((g.n + 31) / 32, 2048);
    hipLaunchKernelGGL(k_gather_rows_staged, dim3(blocks), dim3(256), 0, s, rb->d_off, rb->d_idx,
                       rb->has_value ? rb->d_val : (const float*)nullptr, b->g_rows + g.at, b->g_off + g.at, b->g_lab + g.at, (uint32_t)g.n,
                       b->d_offset + g.at, b->d_label + g.at, b->d_raw, b->gather_any_value ? b->d_value : (float*)nullptr,
                       g.at + g.n == b->nrows ? 1 : 0);
  }
  DFH_HIP(hipGetLastError());
  return gather_queued(b, s);
}
}  // namespace

namespace {
// one stage (RiderKind) of the noted sample sort of b's minibatch as a launch of its own on stream s
void launch_loc_stage(dfh_batch* b, int stage, hipStream_t s, dfh_table* probe = nullptr) {
  const LocView& v = b->loc_v;
  switch (stage) {
    case RID_COUNT:
      if (b->loc_big) hipLaunchKernelGGL(k_loc_count<LOC_BIG_BUCKETS>, dim3(v.ntiles), dim3(LOC_TILE_THREADS), 0, s, v);
      else if (b->loc_fuse_gather) hipLaunchKernelGGL(k_loc_count_gather<LOC_MAX_BUCKETS>, dim3(v.ntiles), dim3(LOC_TILE_THREADS), 0, s, v, b->gsrc);
      else hipLaunchKernelGGL(k_loc_count<LOC_MAX_BUCKETS>, dim3(v.ntiles), dim3(LOC_TILE_THREADS), 0, s, v);
      break;
    case RID_SCATTER:
      if (b->loc_big) hipLaunchKernelGGL(k_loc_scatter<LOC_BIG_BUCKETS>, dim3(v.ntiles), dim3(LOC_TILE_THREADS), 0, s, v);
      else hipLaunchKernelGGL(k_loc_scatter<LOC_MAX_BUCKETS>, dim3(v.ntiles), dim3(LOC_TILE_THREADS), 0, s, v);
      break;
    case RID_SORT:
      hipLaunchKernelGGL(k_loc_sort, dim3(b->loc_gsort), dim3(LOC_SORT_THREADS), 0, s, v);
      break;
    default:
      if (probe) hipLaunchKernelGGL(k_loc_emit<true>, dim3(b->loc_gsort), dim3(LOC_EMIT_THREADS), 0, s, v, b->loc_o, probe->v, b->d_urow);
      else hipLaunchKernelGGL(k_loc_emit<false>, dim3(b->loc_gsort), dim3(LOC_EMIT_THREADS), 0, s, v, b->loc_o, TableView{}, (uint32_t*)nullptr);
      break;
  }
}

void pend_remove(dfh_ctx* c, dfh_batch* b) {
  auto it = std::find(c->pend.begin(), c->pend.end(), b);
  if (it != c->pend.end()) c->pend.erase(it);
}

int flush_pending(dfh_batch* b) {
  dfh_ctx* c = b->ctx;
  if (b->pend_stage >= RID_STAGES) return DFH_OK;
  {
    TimeScope ts(c, DFH_K_LOCALIZE);
    for (int st = b->pend_stage; st < RID_STAGES; ++st) launch_loc_stage(b, st, c->stream);
  }
  b->pend_stage = RID_STAGES;
  pend_remove(c, b);
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

// the riders of the next launch of kind `slot` (0 lookup, 1 forward, 2 update): for every minibatch with noted stages, in the
// order they were noted, its next stage if that stage belongs into this kind of launch.  A stage configured to run alone, or
// one too many for the launch (can_ride false: the launch has no rider form), is queued here as a launch of its own — just
// before the carrier — and the minibatch's following stage is considered at once.  main_groups: the carrier's own blocks / 8.
void collect_riders(dfh_ctx* c, int slot, bool can_ride, uint32_t main_groups, RiderSet* rs) {
  rs->n = 0;
  rs->ngroups = 0;
  rs->period = 1;
  rs->start = 0;
  rs->first[0] = 0;
  if (!c->single_queue || c->pend.empty()) return;
  for (size_t i = 0; i < c->pend.size(); ++i) {
    dfh_batch* p = c->pend[i];
    while (p->pend_stage < RID_STAGES && c->rider_slot[p->pend_stage] == slot) {
      const int st = p->pend_stage;
      if (c->rider_alone[st] || !can_ride || rs->n == MAX_RIDERS) {
        TimeScope ts(c, DFH_K_LOCALIZE);
        launch_loc_stage(p, st, c->stream);
        ++p->pend_stage;
        continue;
      }
      Rider& r = rs->r[rs->n];
      r.v = p->loc_v;
      r.o = p->loc_o;
      r.kind = (uint32_t)st;
      r.nblk = (st == RID_COUNT || st == RID_SCATTER) ? (uint32_t)p->loc_v.ntiles : p->loc_gsort;
      rs->first[rs->n + 1] = rs->first[rs->n] + ((r.nblk + 7u) & ~7u);
      ++rs->n;
      ++p->pend_stage;
      break;  // the minibatch's next stage needs this launch to have ended
    }
  }
  // emit (the stage the next step's lookup waits for) before the others: its blocks are dispatched first
  for (uint32_t j = 1; j < rs->n; ++j) {
    if (rs->r[j].kind == RID_EMIT && rs->r[0].kind != RID_EMIT) {
      std::swap(rs->r[0], rs->r[j]);
      uint32_t at = 0;
      for (uint32_t q = 0; q < rs->n; ++q) {
        rs->first[q] = at;
        at += (rs->r[q].nblk + 7u) & ~7u;
      }
      rs->first[rs->n] = at;
      break;
    }
  }
  c->pend.erase(std::remove_if(c->pend.begin(), c->pend.end(), [](dfh_batch* p) { return p->pend_stage >= RID_STAGES; }), c->pend.end());
  rs->ngroups = rs->first[rs->n] / 8u;
  for (uint32_t q = rs->n + 1; q <= (uint32_t)MAX_RIDERS; ++q) rs->first[q] = rs->first[rs->n];  // (run_rider: no rider beyond n - 1)
  if (rs->ngroups) {
    // every rider group needs a place: the last one sits at group (ngroups - 1) * period of main_groups + ngroups
    rs->start = (uint32_t)((uint64_t)main_groups * (uint64_t)c->rider_start[slot] / 100u);
    const uint32_t left = main_groups - rs->start;   // main groups among which the rider groups are dealt
    uint32_t per = (uint32_t)std::max(1, c->rider_period[slot]);
    if (rs->ngroups > 1) per = std::min(per, (left + rs->ngroups - 1u) / (rs->ngroups - 1u));
    rs->period = std::max(1u, per);
  }
}

// Localizer::Compact of the loaded minibatch; with a table also the key-index probe of dfh_batch_lookup, done by the
// emit pass itself (the thread that writes a unique key looks it up: one launch and one event record fewer per minibatch)
int localize_impl(dfh_batch* b, uint64_t max_index, dfh_table* probe) {
  dfh_ctx* c = b->ctx;
  DFH_HIP(hipSetDevice(c->device));
  const uint32_t N = (uint32_t)b->nnz;
  if (probe && N) {
    if (int rcr = table_reserve(probe, N)) return rcr;  // before anything of this phase is queued: growth drains the streams
  }
  {
    int rc = prep_begin(b);
    if (rc) return rc;
  }
  hipStream_t s = prep_of(b);
  b->looked_up = nullptr;
  if (b->pend_stage < RID_STAGES) {  // localized twice without a step in between
    int rc = flush_pending(b);
    if (rc) return rc;
  }
  if (N == 0) {
    if (b->gather_pending) {  // rows without a single feature: their offsets and labels still have to arrive
      int rcg = gather_alone(b, s);
      if (rcg) return rcg;
    }
    // reference would index an empty vector (localizer.cc:35); define: no keys
    DFH_HIP(hipMemsetAsync(b->d_U, 0, 64 * sizeof(uint32_t), s));  // U = 0
    DFH_HIP(hipMemsetAsync(b->d_col_ptr, 0, 4, s));
    b->seg_nb = 0;  // no long segments
    b->localized = true;
    return prep_end(b);
  }
  const int g = grid_for_threads(N, c);
  TimeScope* tsp = new TimeScope(c, DFH_K_LOCALIZE, s);
  // bucket count: the stored splitters' while the average bucket stays in a sane range (they
  // describe the data distribution, not this minibatch), else sized for this minibatch and
  // bootstrapped from a sample
  int P = 0;
  bool cold = true;
  if (b->spl_P > 0 && N / (uint32_t)b->spl_P >= (uint32_t)LOC_MIN_AVG && N / (uint32_t)b->spl_P <= (uint32_t)LOC_MAX_AVG) {
    P = b->spl_P;
    cold = false;
  } else {
    // the small size class (<= 1024 buckets, up to 700 pairs each) as long as it holds the minibatch, then the large one
    // (<= 4096 buckets: 2.9 M pairs); beyond that the library sort
    // small minibatches (the reference's batch-of-100 quick start) are a chain of launch latencies, not of bandwidth: half-size
    // buckets are one 64-run per wave and one merge round less in k_loc_sort (round 5, profiles/r05v_*)
#ifndef DFH_LOC_SMALL_AVG
#define DFH_LOC_SMALL_AVG 192
#endif
    const size_t avg_bucket = N <= 65536u ? (size_t)DFH_LOC_SMALL_AVG : (size_t)LOC_AVG_BUCKET;
    const size_t P_want = (N + avg_bucket - 1) / avg_bucket;
    if (P_want <= (size_t)LOC_MAX_BUCKETS) P = (int)std::max<size_t>(1, P_want);
    else if (N / LOC_MAX_BUCKETS <= (uint32_t)LOC_MAX_AVG) P = LOC_MAX_BUCKETS;
    else if (P_want <= (size_t)LOC_BIG_BUCKETS) P = (int)P_want;
    else if (N / LOC_BIG_BUCKETS <= (uint32_t)LOC_MAX_AVG) P = LOC_BIG_BUCKETS;
  }
  // a described minibatch (device feed): the count pass of the small size class gathers the rows itself; everywhere else —
  // a first call samples the raw ids before anything is counted, the large size class has no LDS to spare, the library sort
  // has no count pass, the single-queue step may run the count pass much later — the gather is queued here, first
  b->loc_fuse_gather = b->gather_pending && b->gather_fusable && !cold && P > 0 && P <= LOC_MAX_BUCKETS && !b->force_radix &&
                       !c->single_queue && !getenv("DFH_GATHER_ALONE");
  if (b->gather_pending && !b->loc_fuse_gather) {
    int rcg = gather_alone(b, s);
    if (rcg) return rcg;
  }
  if (P > 0 && !b->force_radix) {
    // hand-written sample sort (dfh_localize.hip)
    LocView& v = b->loc_v;
    v.raw = b->d_raw;
    v.n = N;
    v.max_index = max_index;
    v.P = P;
    const bool big = P > LOC_MAX_BUCKETS;
    v.bstride = big ? LOC_BIG_BUCKETS : LOC_MAX_BUCKETS;
    v.ntiles = (int)((N + LOC_TILE - 1) / LOC_TILE);
    v.force_global = b->force_sort_fallback ? 1 : 0;
    v.tb = N > 1 ? std::min(16, __builtin_clz(N - 1)) : 16;  // (N - 1) << tb fits 32 bits
    v.nrows = (uint32_t)b->nrows;
    v.offset = b->d_offset;
    v.rowid = b->d_pos;
    v.smp_key = b->d_smp_key;
    v.smp_pos = b->d_smp_pos;
    v.smp_rank = b->d_smp_rank;
    v.spl_key = b->d_spl_key;
    v.spl_pos = b->d_spl_pos;
    v.packed = b->d_packed;
    v.run_off = b->d_run_off;
    v.btotal = b->d_btotal;
    v.bstart = b->d_bstart;
    v.bkeys = b->d_keys;
    v.bpos = b->d_bpos;
    v.skeys = b->d_skeys;
    v.spos = b->d_spos;
    v.first_key = b->d_first_key;
    v.last_key = b->d_last_key;
    v.nheads = b->d_nheads;
    v.lh = b->d_lh;
    EmitOut& o = b->loc_o;
    o.value = b->has_value ? b->d_value : (const float*)nullptr;
    o.feaids = b->d_feaids;
    o.col_ptr = b->d_col_ptr;
    o.index = b->d_index;
    o.s_row = b->d_s_row;
    o.s_val = b->d_s_val;
    o.d_U = b->d_U;
    o.sl.mid = b->d_mid;
    o.sl.mid_ent = b->d_mid_ent;
    o.sl.hot = b->d_hot;
    o.sl.hot_ent = b->d_hot_ent;
    o.sl.few = b->d_few;
    o.sl.few_ent = b->d_few_ent;
    v.split_n = b->d_U + 2;
    b->loc_big = big;
#ifndef DFH_LOC_GRID_CAP
#define DFH_LOC_GRID_CAP 1024
#endif
    b->loc_gsort = (unsigned)std::min<int>(P, big ? LOC_BIG_BUCKETS : DFH_LOC_GRID_CAP);
    if (cold && P > 1) {
      const uint32_t S = (uint32_t)P * LOC_OVERSAMPLE;
      const uint32_t nt = (S + 255) / 256;
      hipLaunchKernelGGL(k_ss_sample, dim3(nt), dim3(256), 0, s, v);
      hipLaunchKernelGGL(k_ss_rank, dim3(nt * nt), dim3(256), 0, s, v);
      hipLaunchKernelGGL(k_loc_splitters, dim3(nt), dim3(256), 0, s, v);
    }
    // single-queue step: note the four stages; they ride in the launches of the steps that follow (dfh_riders.hip).  Not
    // for a first call (its splitters are being bootstrapped right here), the large size class (64 KB of LDS in count) or a
    // probe folded into emit
    if (c->single_queue && !cold && !big && !probe) {
      b->pend_stage = 0;
      c->pend.push_back(b);
    } else {
      for (int st = 0; st < RID_STAGES; ++st) {
        launch_loc_stage(b, st, s, probe);
        if (st == RID_COUNT && b->loc_fuse_gather) {  // the rows have been read once this launch is through
          int rcg = gather_queued(b, s);
          b->loc_fuse_gather = false;
          if (rcg) return rcg;
        }
      }
    }
    b->spl_P = P;  // k_loc_emit leaves this minibatch's exact P-quantiles as the next call's splitters
    b->seg_nb = (uint32_t)P;
  } else {
    // very large batches: library LSD radix sort
    hipLaunchKernelGGL(k_rdx_keys, dim3(g), dim3(256), 0, s, b->d_raw, N, max_index, b->d_keys, b->d_pos);
    size_t tb = b->temp_bytes;
    DFH_HIP(rocprim::radix_sort_pairs(b->d_temp, tb, b->d_keys, b->d_skeys, b->d_pos, b->d_spos, (size_t)N, 0, 64, s));
    hipLaunchKernelGGL(k_rdx_heads, dim3(g), dim3(256), 0, s, b->d_skeys, N, b->d_head);
    tb = b->temp_bytes;
    DFH_HIP(rocprim::inclusive_scan(b->d_temp, tb, b->d_head, b->d_uid, (size_t)N, rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(k_rdx_emit, dim3(g), dim3(256), 0, s, b->d_skeys, b->d_spos, b->d_head, b->d_uid, N, (uint32_t)b->nrows,
                       b->d_offset, b->has_value ? b->d_value : (const float*)nullptr, b->d_feaids, b->d_col_ptr, b->d_index,
                       b->d_s_row, b->d_s_val, b->d_U);
    // keys with long segments, for the backward pass: one list bucket
    hipLaunchKernelGGL(k_seg_lists_reset, dim3(1), dim3(64), 0, s, b->d_mid, b->d_hot, b->d_few, b->d_U + 2);
    hipLaunchKernelGGL(k_seg_lists, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((N + 1023) / 1024, 256))), dim3(1024), 0, s,
                       b->d_col_ptr, b->d_U, b->d_mid, b->d_hot, b->d_few, b->d_mid_ent, b->d_hot_ent, b->d_few_ent);
    b->seg_nb = 1;
    if (probe)  // the library sort's path has no emit pass to carry the probe
      hipLaunchKernelGGL(k_lookup, dim3(grid_for_threads(b->nnz, c)), dim3(256), 0, s, probe->v, b->d_feaids, b->d_U, 0u, b->d_urow,
                         (const float*)nullptr, b->d_col_ptr, 0, (uint32_t*)nullptr, 0, (uint2*)nullptr, AucFin{nullptr, 0u, nullptr});
  }
  delete tsp;
  DFH_HIP(hipGetLastError());
  b->localized = true;
  b->has_cnt = false;
  if (probe) b->looked_up = probe;
  return prep_end(b);
}
}  // namespace


int dfh_batch_prepare_rows(dfh_table* t, dfh_batch* b, size_t nrows, const size_t* offset, const float* label, int nseg,
                           dfh_rowbuf* const* bufs, const uint32_t* const* rows, const size_t* seg_rows, uint64_t max_index) {
  DFH_ARG(t && b && t->ctx == b->ctx && offset && label && nseg >= 1 && bufs && rows && seg_rows, "dfh_batch_prepare_rows: NULL argument");
  DFH_ARG(nrows >= 1 && nrows <= b->max_rows, "dfh_batch_prepare_rows: nrows out of range");
  DFH_ARG(max_index != 0, "max_index must be nonzero");
  const size_t base = offset[0], nnz = offset[nrows] - base;
  DFH_ARG(nnz <= b->max_nnz && nnz < 0xFFFFFFFFULL, "dfh_batch_prepare_rows: nnz exceeds max_nnz");
  size_t total = 0;
  bool any_value = false;
  for (int g = 0; g < nseg; ++g) {
    DFH_ARG(bufs[g] && bufs[g]->ctx == b->ctx && (seg_rows[g] == 0 || rows[g]), "dfh_batch_prepare_rows: bad segment");
    total += seg_rows[g];
    any_value = any_value || bufs[g]->has_value;
  }
  DFH_ARG(total == nrows, "dfh_batch_prepare_rows: the segments must hold nrows rows");
  dfh_ctx* c = b->ctx;
  static const bool prof = getenv("DFH_PROFILE_PREP") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp = prof ? now() : 0;
  auto lap = [&](int k) { if (prof) { const double x = now(); b->t_prof[k] += x - tp; tp = x; } };
  DFH_HIP(hipSetDevice(c->device));
  if (nnz) {
    int rcr = table_reserve(t, nnz);  // U <= nnz keys may be new; before anything of this phase is queued
    if (rcr) return rcr;
  }
  phase_begin(b);
  int rc = prep_begin(b);
  if (rc) return rc;
  hipStream_t s = prep_of(b);
  b->d_raw = b->o_raw; b->d_offset = b->o_offset; b->d_value = b->o_value; b->d_label = b->o_label;
  // the same page-locked block as dfh_batch_load_host / dfh_batch_gather_rows: offsets | labels | (ids ->) row numbers
  const size_t o_off = 0, o_lab = (b->max_rows + 1) * 4, o_idx = ((o_lab + b->max_rows * 4 + 255) & ~(size_t)255),
               o_val = o_idx + b->max_nnz * 8, stage_total = o_val + b->max_nnz * 4;
  (void)stage_total;   // (the layout of dfh_batch_load_host; only the head of it is used here)
  const size_t o_tile = o_idx + (b->max_rows + 1) * 4;   // the tiles' first rows (k_loc_count_gather), behind the row numbers
  rc = ensure_stage(b, o_tile + (b->max_tiles + 2) * 4);
  if (rc) return rc;
  if (!b->d_stage_view) DFH_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->d_stage_view), b->h_stage, 0));
  lap(0);  // set-up, prep_begin (wait for the batch object's previous step)
  if (b->staged_pending) {
    // (a query first: hipEventSynchronize costs ~100 us of host time even on an event that completed long ago)
    if (hipEventQuery(b->ev_staged) != hipSuccess) DFH_HIP(hipEventSynchronize(b->ev_staged));
    b->staged_pending = false;
  }
  lap(1);  // the previous description has been read
  uint32_t* h_off = reinterpret_cast<uint32_t*>(b->h_stage + o_off);
  for (size_t i = 0; i <= nrows; ++i) {
    DFH_ARG(offset[i] >= base && (i == 0 || offset[i] >= offset[i - 1]), "dfh_batch_prepare_rows: offsets must not decrease");
    h_off[i] = (uint32_t)(offset[i] - base);
  }
  memcpy(b->h_stage + o_lab, label, nrows * 4);
  uint32_t* h_rows = reinterpret_cast<uint32_t*>(b->h_stage + o_idx);
  size_t at = 0;
  for (int g = 0; g < nseg; ++g) {
    const uint32_t lim = (uint32_t)bufs[g]->nrows;
    const uint32_t* src = rows[g];
    uint32_t worst = 0;
    for (size_t i = 0; i < seg_rows[g]; ++i) {
      h_rows[at + i] = src[i];
      worst = std::max(worst, src[i]);
    }
    DFH_ARG(seg_rows[g] == 0 || worst < lim, "dfh_batch_prepare_rows: row number beyond the buffer");
    at += seg_rows[g];
  }
  lap(2);  // description written
  const uint32_t* v_off = reinterpret_cast<const uint32_t*>(b->d_stage_view + o_off);
  const float* v_lab = reinterpret_cast<const float*>(b->d_stage_view + o_lab);
  const uint32_t* v_rows = reinterpret_cast<const uint32_t*>(b->d_stage_view + o_idx);
  // the rows stay where they are for now: the Localizer's count pass gathers them as it reads them (k_loc_count_gather), or
  // localize_impl queues k_gather_rows_staged first where that pass cannot (see dfh_batch::gsegs)
  b->gsegs.clear();
  at = 0;
  for (int g = 0; g < nseg; ++g) {
    dfh_rowbuf* rb = bufs[g];
    if (seg_rows[g] == 0) continue;
    bool waited;
    {
      std::lock_guard<std::mutex> lk(rb->mu);
      waited = std::find(rb->seen_loaded.begin(), rb->seen_loaded.end(), s) != rb->seen_loaded.end();
      if (!waited) rb->seen_loaded.push_back(s);
    }
    if (!waited) DFH_HIP(hipStreamWaitEvent(s, rb->ev_loaded, 0));
    b->gsegs.push_back({rb, at, seg_rows[g]});
    at += seg_rows[g];
  }
  b->g_rows = v_rows;
  b->g_off = v_off;
  b->g_lab = v_lab;
  b->gather_any_value = any_value;
  b->gather_pending = true;
  // what the count pass needs on top: the first row of every tile of LOC_TILE positions (the last row that starts at or before
  // the tile), behind the row numbers in the same page-locked block; a tile may span LOC_GATHER_ROWS rows, a minibatch
  // LOC_GATHER_SEGS buffers
  {
    GatherSrc& gs = b->gsrc;
    const size_t ntiles = (nnz + LOC_TILE - 1) / LOC_TILE;
    uint32_t* h_tile = reinterpret_cast<uint32_t*>(b->h_stage + o_tile);
    bool ok = nnz > 0 && b->gsegs.size() <= (size_t)LOC_GATHER_SEGS;
    size_t r = 0;
    for (size_t t = 0; t < ntiles && ok; ++t) {
      const uint32_t p = (uint32_t)(t * LOC_TILE);
      while (r + 1 < nrows && h_off[r + 1] <= p) ++r;   // the last row with off[r] <= p
      h_tile[t] = (uint32_t)r;
      if (t > 0 && r - h_tile[t - 1] + 1 > (size_t)LOC_GATHER_ROWS) ok = false;
    }
    if (ok) {
      h_tile[ntiles] = (uint32_t)nrows;
      if (nrows - h_tile[ntiles - 1] > (size_t)LOC_GATHER_ROWS) ok = false;   // (the last tile's rows, trailing empty ones included)
    }
    b->gather_fusable = ok;
    gs.nseg = (int)b->gsegs.size();
    for (int g = 0; g <= LOC_GATHER_SEGS; ++g) gs.seg_row0[g] = (uint32_t)nrows;
    for (int g = 0; g < LOC_GATHER_SEGS; ++g) {
      const bool have = g < gs.nseg && ok;
      gs.seg_row0[g] = have ? (uint32_t)b->gsegs[g].at : (uint32_t)nrows;
      gs.src_off[g] = have ? b->gsegs[g].rb->d_off : nullptr;
      gs.src_idx[g] = have ? b->gsegs[g].rb->d_idx : nullptr;
      gs.src_val[g] = (have && b->gsegs[g].rb->has_value) ? b->gsegs[g].rb->d_val : nullptr;
    }
    gs.h_rows = v_rows;
    gs.h_off = v_off;
    gs.h_lab = v_lab;
    gs.h_tile_row = reinterpret_cast<const uint32_t*>(b->d_stage_view + o_tile);
    gs.dst_raw = b->d_raw;
    gs.dst_val = any_value ? b->d_value : nullptr;
    gs.dst_off = b->d_offset;
    gs.dst_lab = b->d_label;
  }
  b->nrows = nrows;
  b->nnz = nnz;
  b->has_value = any_value;
  b->has_cnt = false;
  b->localized = false;
  b->looked_up = nullptr;
  lap(3);  // gather queued
  // Localizer::Compact + the key-index probe, same phase: ONE ev_ready at the end
  b->defer_ready = true;
  rc = localize_impl(b, max_index, nullptr);
  lap(4);  // Localizer queued
  if (!rc && nnz > kSmallBatchPairs && !c->single_queue) {   // (a small minibatch: the step's own pass probes, see dfh_batch_lookup)
    hipLaunchKernelGGL(k_lookup, dim3(grid_for_threads(b->nnz, c)), dim3(256), 0, s, t->v, b->d_feaids, b->d_U, 0u, b->d_urow,
                       (const float*)nullptr, b->d_col_ptr, 0, (uint32_t*)nullptr, 0, (uint2*)nullptr, AucFin{nullptr, 0u, nullptr});
    if (hipGetLastError() != hipSuccess) rc = DFH_ERR_HIP;
    b->looked_up = t;
  }
  b->defer_ready = false;
  if (rc) return rc;
  rc = prep_end(b);
  lap(5);  // lookup queued, ev_ready recorded
  if (prof) ++b->n_prof;
  return rc;
}

int dfh_localize(dfh_batch* b, uint64_t max_index) {
  DFH_ARG(b && b->nrows > 0, "dfh_localize: no batch loaded");
  DFH_ARG(max_index != 0, "max_index must be nonzero");
  return localize_impl(b, max_index, nullptr);
}

int dfh_localize_lookup(dfh_table* t, dfh_batch* b, uint64_t max_index) {
  DFH_ARG(t && b && t->ctx == b->ctx, "dfh_localize_lookup: bad argument");
  DFH_ARG(b->nrows > 0, "dfh_localize_lookup: no batch loaded");
  DFH_ARG(max_index != 0, "max_index must be nonzero");
  return localize_impl(b, max_index, t);
}

int dfh_batch_set_option(dfh_batch* b, const char* name, int value) {
  DFH_ARG(b && name, "NULL argument");
  if (std::string(name) == "force_radix_sort") {
    b->force_radix = value != 0;
    return DFH_OK;
  }
  if (std::string(name) == "compute_auc") {
    b->compute_auc = value != 0;
    return DFH_OK;
  }
  if (std::string(name) == "force_sort_fallback") {
    b->force_sort_fallback = value != 0;
    return DFH_OK;
  }
  if (std::string(name) == "reset_splitters") {
    if (value) b->spl_P = 0;  // the next dfh_localize bootstraps its splitters from a sample again
    return DFH_OK;
  }
  set_error(std::string("unknown batch option ") + name);
  return DFH_ERR_ARG;
}

int dfh_batch_lookup(dfh_table* t, dfh_batch* b) {
  DFH_ARG(t && b && t->ctx == b->ctx, "dfh_batch_lookup: bad argument");
  if (!b->localized) {
    set_error("dfh_batch_lookup: batch is not localized");
    return DFH_ERR_STATE;
  }
  dfh_ctx* c = b->ctx;
  if (b->nnz == 0) return DFH_OK;
  // A small minibatch is bound by the number of launches, not by what they move (C2, 7 500 pairs: the worker loop's step
  // takes exactly as long as the host needs to queue it): the step's own lookup pass probes in the launch it makes anyway.
  // 50.9 -> 44.8 us per step on the rcv1 shape (profiles/r05b_c2_launch_bound.txt).
  if (b->nnz <= kSmallBatchPairs) return DFH_OK;
  if (c->single_queue) return DFH_OK;  // no preparation stream to probe on: the step's own pass probes and pushes in one
  DFH_HIP(hipSetDevice(c->device));
  int rc = table_reserve(t, b->nnz);  // U <= nnz keys may be new
  if (rc) return rc;
  rc = prep_begin(b);
  if (rc) return rc;
  {
    hipStream_t ps = prep_of(b);
    TimeScope ts(c, DFH_K_LOOKUP, ps);
    hipLaunchKernelGGL(k_lookup, dim3(grid_for_threads(b->nnz, c)), dim3(256), 0, ps, t->v, b->d_feaids, b->d_U, 0u,
                       b->d_urow, (const float*)nullptr, b->d_col_ptr, 0, (uint32_t*)nullptr, 0, (uint2*)nullptr, AucFin{nullptr, 0u, nullptr});
  }
  DFH_HIP(hipGetLastError());
  b->looked_up = t;
  return prep_end(b);
}

int dfh_batch_load_localized_host(dfh_batch* b, size_t nrows, const size_t* offset, const uint32_t* index, const float* value,
                                  const float* label, const uint64_t* feaids, const float* feacnt, size_t U) {
  DFH_ARG(b && offset && label && feaids, "dfh_batch_load_localized_host: NULL argument");
  DFH_ARG(nrows >= 1 && nrows <= b->max_rows, "nrows out of range");
  std::vector<uint32_t> off32;
  int rc = to_u32_offsets(offset, nrows, &off32);
  if (rc) return rc;
  const size_t base = offset[0], nnz = off32[nrows];
  DFH_ARG(nnz <= b->max_nnz && U <= b->max_nnz, "batch exceeds max_nnz");
  DFH_ARG(nnz == 0 || index, "index is NULL");
  for (size_t u = 0; u < U; ++u) {
    DFH_ARG(feaids[u] != kEmptyKey, "key ~0 is reserved");
    DFH_ARG(u == 0 || feaids[u] > feaids[u - 1], "feaids must be strictly ascending (Localizer output)");
  }
  // key-ordered occurrence view: stable counting sort of the nnz by compact index
  std::vector<uint32_t> col_ptr(U + 1, 0), s_row(std::max<size_t>(nnz, 1));
  std::vector<float> s_val(value ? std::max<size_t>(nnz, 1) : 0);
  for (size_t j = 0; j < nnz; ++j) {
    DFH_ARG(index[base + j] < U, "compact index out of range");
    ++col_ptr[index[base + j] + 1];
  }
  for (size_t u = 0; u < U; ++u) col_ptr[u + 1] += col_ptr[u];
  {
    std::vector<uint32_t> fill(col_ptr.begin(), col_ptr.end() - 1);
    for (size_t i = 0; i < nrows; ++i)
      for (uint32_t j = off32[i]; j < off32[i + 1]; ++j) {
        uint32_t q = fill[index[base + j]]++;
        s_row[q] = (uint32_t)i;
        if (value) s_val[q] = value[base + j];
      }
  }
  DFH_HIP(hipSetDevice(b->ctx->device));
  phase_begin(b);
  rc = prep_begin(b);
  if (rc) return rc;
  hipStream_t s = prep_of(b);
  b->d_raw = b->o_raw; b->d_offset = b->o_offset; b->d_value = b->o_value; b->d_label = b->o_label;
  b->looked_up = nullptr;
  uint32_t U32 = (uint32_t)U;
  DFH_HIP(hipMemcpyAsync(b->d_offset, off32.data(), (nrows + 1) * 4, hipMemcpyHostToDevice, s));
  if (nnz) {
    DFH_HIP(hipMemcpyAsync(b->d_index, index + base, nnz * 4, hipMemcpyHostToDevice, s));
    DFH_HIP(hipMemcpyAsync(b->d_s_row, s_row.data(), nnz * 4, hipMemcpyHostToDevice, s));
    if (value) {
      DFH_HIP(hipMemcpyAsync(b->d_value, value + base, nnz * 4, hipMemcpyHostToDevice, s));
      DFH_HIP(hipMemcpyAsync(b->d_s_val, s_val.data(), nnz * 4, hipMemcpyHostToDevice, s));
    }
  }
  DFH_HIP(hipMemcpyAsync(b->d_label, label, nrows * 4, hipMemcpyHostToDevice, s));
  if (U) DFH_HIP(hipMemcpyAsync(b->d_feaids, feaids, U * 8, hipMemcpyHostToDevice, s));
  if (U && feacnt) DFH_HIP(hipMemcpyAsync(b->d_feacnt, feacnt, U * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(b->d_col_ptr, col_ptr.data(), (U + 1) * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(b->d_U, &U32, 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_seg_lists_reset, dim3(1), dim3(64), 0, s, b->d_mid, b->d_hot, b->d_few, b->d_U + 2);
  if (U) {
    hipLaunchKernelGGL(k_seg_lists, dim3((unsigned)std::min<size_t>((U + 1023) / 1024, 256)), dim3(1024), 0, s, b->d_col_ptr,
                       b->d_U, b->d_mid, b->d_hot, b->d_few, b->d_mid_ent, b->d_hot_ent, b->d_few_ent);
  }
  b->seg_nb = 1;
  DFH_HIP(hipGetLastError());
  DFH_HIP(hipStreamSynchronize(s));
  b->nrows = nrows;
  b->nnz = nnz;
  b->has_value = value != nullptr;
  b->has_cnt = feacnt != nullptr;
  b->localized = true;
  return prep_end(b);
}

int dfh_batch_shape(dfh_batch* b, size_t* nrows, size_t* nnz, size_t* U) {
  DFH_ARG(b, "NULL batch");
  if (nrows) *nrows = b->nrows;
  if (nnz) *nnz = b->nnz;
  if (U) {
    DFH_ARG(b->localized, "batch is not localized");
    uint32_t u = 0;
    int rc = flush_pending(b);
    if (rc) return rc;
    rc = sync_all(b->ctx);
    if (rc) return rc;
    DFH_HIP(hipMemcpyAsync(&u, b->d_U, 4, hipMemcpyDeviceToHost, b->ctx->stream));
    DFH_HIP(hipStreamSynchronize(b->ctx->stream));
    *U = u;
  }
  return DFH_OK;
}

int dfh_batch_get_localized(dfh_batch* b, size_t* U, uint64_t* feaids, float* feacnt, uint32_t* index) {
  DFH_ARG(b && b->localized, "batch is not localized");
  size_t u = 0;
  int rc = dfh_batch_shape(b, nullptr, nullptr, &u);
  if (rc) return rc;
  if (U) *U = u;
  hipStream_t s = b->ctx->stream;
  if (feaids && u) DFH_HIP(hipMemcpyAsync(feaids, b->d_feaids, u * 8, hipMemcpyDeviceToHost, s));
  if (feacnt && u) {
    if (!b->has_cnt) {
      hipLaunchKernelGGL(k_loc_counts, dim3(grid_for_threads(u, b->ctx)), dim3(256), 0, s, b->d_col_ptr, b->d_U, b->d_feacnt);
      DFH_HIP(hipGetLastError());
    }
    DFH_HIP(hipMemcpyAsync(feacnt, b->d_feacnt, u * 4, hipMemcpyDeviceToHost, s));
  }
  if (index && b->nnz) DFH_HIP(hipMemcpyAsync(index, b->d_index, b->nnz * 4, hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  return DFH_OK;
}

int dfh_batch_device_keys(dfh_batch* b, const uint64_t** d_feaids, const float** d_feacnt, size_t* U) {
  DFH_ARG(b && b->localized, "batch is not localized");
  if (int rcf = flush_pending(b)) return rcf;  // the caller is about to read the arrays
  if (d_feaids) *d_feaids = b->d_feaids;
  if (d_feacnt) {
    if (!b->has_cnt) {
      // ordered after the batch's preparation, on the stream the consumers of the counts run on
      int rc = main_begin(b);
      if (rc) return rc;
      hipLaunchKernelGGL(k_loc_counts, dim3(grid_for_threads(b->nnz, b->ctx)), dim3(256), 0, b->ctx->stream, b->d_col_ptr,
                         b->d_U, b->d_feacnt);
      DFH_HIP(hipGetLastError());
      b->has_cnt = true;
    }
    *d_feacnt = b->d_feacnt;
  }
  if (U) return dfh_batch_shape(b, nullptr, nullptr, U);  // synchronises; pass NULL to stay asynchronous
  return DFH_OK;
}

int dfh_batch_key_ranges_device(dfh_batch* b, int nparts, const uint64_t* d_splits, int64_t* d_bounds) {
  DFH_ARG(b && b->localized && d_bounds && nparts >= 1 && nparts <= 1024, "dfh_batch_key_ranges_device: bad argument");
  dfh_ctx* c = b->ctx;
  int rc = main_begin(b);
  if (rc) return rc;
  const uint64_t span = nparts == 1 ? ~0ULL : (~0ULL / (uint64_t)nparts) + 1;
  hipLaunchKernelGGL(k_key_ranges64, dim3((nparts + 256) / 256), dim3(256), 0, c->stream, b->d_feaids, b->d_U, nparts, span,
                     d_splits, d_bounds, (uint32_t)(b->nnz == 0));
  DFH_HIP(hipGetLastError());
  return DFH_OK;
}

int dfh_batch_key_ranges(dfh_batch* b, int nparts, uint32_t* bounds) {
  DFH_ARG(b && b->localized && bounds && nparts >= 1 && nparts <= 1024, "dfh_batch_key_ranges: bad argument");
  dfh_ctx* c = b->ctx;
  int rc = ensure_scratch(c, (size_t)(nparts + 1) * 4 + 256);
  if (rc) return rc;
  uint32_t* d_bounds = static_cast<uint32_t*>(c->scratch);
  // span = ceil(2^64 / nparts): shard d owns keys in [d*span, (d+1)*span)
  const uint64_t span = nparts == 1 ? ~0ULL : (~0ULL / (uint64_t)nparts) + 1;
  hipStream_t s = prep_of(b);
  if (b->nnz == 0) {
    for (int d = 0; d <= nparts; ++d) bounds[d] = 0;
    return DFH_OK;
  }
  if (int rcf = flush_pending(b)) return rcf;
  hipLaunchKernelGGL(k_key_ranges, dim3((nparts + 256) / 256), dim3(256), 0, s, b->d_feaids, b->d_U, nparts, span, d_bounds);
  DFH_HIP(hipGetLastError());
  DFH_HIP(hipMemcpyAsync(bounds, d_bounds, (size_t)(nparts + 1) * 4, hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  return DFH_OK;
}

// ------------------------------------------------------------ the fused step
int dfh_sgd_step(dfh_table* t, dfh_batch* b, int is_train, int push_cnt) {
  DFH_ARG(t && b, "dfh_sgd_step: NULL argument");
  DFH_ARG(t->ctx == b->ctx, "table and batch must share a context");
  if (!b->localized) {
    set_error("dfh_sgd_step: batch is not localized (call dfh_localize first)");
    return DFH_ERR_STATE;
  }
  if (is_train) {
    if (int rca = require_aux(t, "dfh_sgd_step(is_train)")) return rca;
  }
  dfh_ctx* c = t->ctx;
  DFH_HIP(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const int k = t->v.k, kp = t->v.kp;
  DFH_ARG(kp <= 256, "V_dim > 256 is not supported by the fused step");
  int rc = ensure_xv(b, kp);
  if (rc) return rc;
  b->nrows_seen += (float)b->nrows;
  RiderSet none;  // (stages of later minibatches due in a launch this step does not make: they run alone)
  if (b->nnz == 0) {
    // rows without features: pred = 0 for every example; nothing to pull or push
    RowSrc src = table_src(t, b->d_urow);
    rc = main_begin(b);
    if (rc) return rc;
    collect_riders(c, 0, false, 0, &none);
    rc = launch_forward(b, src, k, kp, nullptr, nullptr, true);
    if (rc) return rc;
    collect_riders(c, 2, false, 0, &none);
    return main_end(b);
  }
  const bool refrand = t->v.p.init_mode == DFH_INIT_REFRAND && k > 0;
  const uint32_t Nb = (uint32_t)b->nnz;  // upper bound of U for grids
  if (b->looked_up != t) {  // the step's own lookup probes the index: up to U <= nnz new keys
    rc = table_reserve(t, Nb);
    if (rc) return rc;
  }
  rc = main_begin(b);
  if (rc) return rc;
  // Pull: key -> row (+ epoch-0 Push(kFeaCount), sgd_learner.cc:214-217).  When dfh_batch_lookup
  // already resolved the rows on a preparation stream, what remains here is one pass over the
  // known rows: the count push (it must stay ordered with the previous step's update) and
  // {row, w} per key for the forward (w is current: the previous step's update precedes it on this
  // stream), so that the forward touches nothing of a row but its V lines.
  const bool pre = b->looked_up == t;
  uint2* uw = b->d_uw;
  // a training step's update kernel rewrites every key's header: it takes the count push of the keys that have their V
  // along (k_lookup then only reads them)
  const bool defer_cnt = push_cnt && is_train && c->upd_kernel != 0;
  {
    // DFH_GAP_TRACE=1 (measurement): the step's lookup dispatch carries its own start / stop events, like the forward and the
    // update, and dfh_ctx_get_timing prints the time between the end of one timed launch and the start of the next
    static const bool gap_trace = getenv("DFH_GAP_TRACE") != nullptr;
    hipEvent_t la = nullptr, lb = nullptr;
    if (gap_trace && ((c->timing >> DFH_K_LOOKUP) & 1u)) {
      la = TimeScope::get(c);
      lb = TimeScope::get(c);
    }
    TimeScope ts(c, (la && lb) ? DFH_K_COUNT : DFH_K_LOOKUP);
    // (the lookup's first block also adds up the AUC slots this batch object's previous step left behind)
    const int gl = grid_for_threads(Nb, c);
    // the pass that sees every unique key's segment lists the parts of the very hot ones for the update's split role
    const SplitOut so = (is_train && c->upd_kernel && c->upd_split) ? SplitOut{b->d_split_ent, b->d_U + 2, 0u} : SplitOut{nullptr, nullptr, 0u};
    RiderSet rs;  // single-queue step: the stages of later minibatches' Localizer that belong into this launch
    collect_riders(c, 0, true, ((uint32_t)gl + 7u) / 8u, &rs);
    if (rs.n)
      hipLaunchKernelGGL(k_lookup_riders, dim3((((unsigned)gl + 7u) / 8u + rs.ngroups) * 8u), dim3(256), rider_smem(rs), s, t->v, b->d_feaids,
                         b->d_U, 0u, b->d_urow, b->has_cnt ? b->d_feacnt : (const float*)nullptr, b->d_col_ptr,
                         push_cnt ? (defer_cnt ? 2 : 1) : 0, refrand ? b->d_need : (uint32_t*)nullptr, pre ? 1 : 0, uw, auc_pending(b),
                         so, (uint32_t)gl, rs);
    else if (la && lb) {
      hipExtLaunchKernelGGL(k_lookup_step, dim3(gl), dim3(256), 0, s, la, lb, 0, t->v, b->d_feaids, b->d_U, 0u, b->d_urow,
                            b->has_cnt ? b->d_feacnt : (const float*)nullptr, b->d_col_ptr, push_cnt ? (defer_cnt ? 2 : 1) : 0,
                            refrand ? b->d_need : (uint32_t*)nullptr, pre ? 1 : 0, uw, auc_pending(b), so);
      c->spans.push_back({DFH_K_LOOKUP, la, lb, 1});
    } else
    hipLaunchKernelGGL(k_lookup_step, dim3(gl), dim3(256), 0, s, t->v, b->d_feaids, b->d_U, 0u, b->d_urow,
                       b->has_cnt ? b->d_feacnt : (const float*)nullptr, b->d_col_ptr, push_cnt ? (defer_cnt ? 2 : 1) : 0,
                       refrand ? b->d_need : (uint32_t*)nullptr, pre ? 1 : 0, uw, auc_pending(b), so);
    b->auc_pending_n = 0;
  }
  DFH_HIP(hipGetLastError());
  if (push_cnt && refrand) {
    rc = refrand_flush(t, b->d_feaids, b->d_U, Nb, b->d_urow, b->d_need, b->d_rank, b->d_total);
    if (rc) return rc;
  }
  RowSrc src = table_src(t, b->d_urow);
  rc = launch_forward(b, src, k, kp, uw, nullptr, true);
  if (rc) return rc;
  // BinClassMetric::AUC of every minibatch (sgd_learner.cc:153-155): in a training step it rides in the update launch
  // (k_update_fused: the pair counting is VALU work beside a memory-bound kernel), otherwise it is a launch of its own
  bool auc_rides = b->compute_auc && is_train && c->auc_in_update != 0;
  if (b->compute_auc && !auc_rides) {
    TimeScope ts(c, DFH_K_AUC);
    rc = launch_auc(b);
    if (rc) return rc;
  }
  if (is_train) {
    if (refrand) DFH_HIP(hipMemsetAsync(b->d_need, 0, (size_t)Nb * 4, s));
    const bool wanted = auc_rides;
    rc = launch_backward<true>(b, src, t->v, nullptr, 0, k, kp, b->d_need, kAllKeys, uw, defer_cnt, &auc_rides, true);
    if (rc) return rc;
    if (wanted && !auc_rides) {  // k_backward_all ran (upd_kernel = 0) or the minibatch is beyond the pair-counting size
      TimeScope ts(c, DFH_K_AUC);
      rc = launch_auc(b);
      if (rc) return rc;
    }
    if (refrand) {
      rc = refrand_flush(t, b->d_feaids, b->d_U, Nb, b->d_urow, b->d_need, b->d_rank, b->d_total);
      if (rc) return rc;
    }
  } else {
    BatchView bv = batch_view(b);
    collect_riders(c, 2, false, 0, &none);
    hipLaunchKernelGGL((k_penalty<1>), dim3(std::min(grid_for_waves(Nb, c), PROG_SLOTS)), dim3(256), 0, s, bv, src, t->v, k, kp, kAllKeys);
    DFH_HIP(hipGetLastError());
  }
  return main_end(b);
}

int dfh_batch_forward(dfh_batch* b, int V_dim, const float* d_rows) {
  DFH_ARG(b && b->localized && d_rows, "dfh_batch_forward: bad argument");
  const int kp = (V_dim + 3) / 4 * 4;
  int rc = ensure_xv(b, kp);
  if (rc) return rc;
  b->nrows_seen += (float)b->nrows;
  rc = main_begin(b);
  if (rc) return rc;
  rc = launch_forward(b, packed_src(d_rows, V_dim), V_dim, kp);
  if (rc) return rc;
  return main_end(b);  // a prediction-only step ends here; dfh_batch_backward moves the mark
}

int dfh_batch_backward(dfh_batch* b, int V_dim, const float* d_rows, float* d_grads) {
  DFH_ARG(b && b->localized && d_rows && d_grads, "dfh_batch_backward: bad argument");
  const int kp = (V_dim + 3) / 4 * 4;
  if (b->nnz == 0) return DFH_OK;
  TableView dummy{};
  int rc = launch_backward<false>(b, packed_src(d_rows, V_dim), dummy, d_grads, dfh_row_stride(V_dim), V_dim, kp, nullptr);
  if (rc) return rc;
  return main_end(b);
}

int dfh_batch_progress(dfh_batch* b, dfh_progress* out, int reset) {
  DFH_ARG(b && out, "NULL argument");
  std::vector<double> p(2 * PROG_SLOTS + 1);
  hipStream_t s = b->ctx->stream;
  {
    int rc = auc_flush_pending(b);  // the last step's AUC slots, if no later step has added them up
    if (rc) return rc;
    rc = sync_all(b->ctx);
    if (rc) return rc;
  }
  DFH_HIP(hipMemcpyAsync(p.data(), b->d_prog, p.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  double loss = 0, pen = 0;
  for (int i = 0; i < PROG_SLOTS; ++i) {
    loss += p[PROG_LOSS * PROG_SLOTS + i];
    pen += p[PROG_PENALTY * PROG_SLOTS + i];
  }
  out->loss = (float)loss;
  out->penalty = (float)pen;
  out->auc = (float)p[PROG_AUC * PROG_SLOTS];
  out->nnz_w = 0;
  out->nrows = b->nrows_seen;
  if (reset) {
    DFH_HIP(hipMemsetAsync(b->d_prog, 0, (2 * PROG_SLOTS + 64) * sizeof(double), s));
    b->nrows_seen = 0;
  }
  return DFH_OK;
}

int dfh_batch_get_pred(dfh_batch* b, float* pred) {
  DFH_ARG(b && pred, "NULL argument");
  hipStream_t s = b->ctx->stream;
  DFH_HIP(hipMemcpyAsync(pred, b->d_pred, b->nrows * 4, hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  return DFH_OK;
}

int dfh_auc_times_n(dfh_ctx* c, const float* label, const float* pred, size_t n, float* auc_n) {
  DFH_ARG(c && auc_n, "NULL argument");
  *auc_n = 1.0f;
  if (n == 0) return DFH_OK;
  DFH_ARG(label && pred && n < 0xFFFFFFF0ULL, "dfh_auc_times_n: bad argument");
  DFH_HIP(hipSetDevice(c->device));
  const bool pairs = n <= AUC_PAIRS_MAX_N;
  size_t tb = 0;
  if (!pairs)
    rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 32,
                              c->stream);
  int rc = ensure_scratch(c, 2 * padded<float>(n) + (pairs ? 0 : 4 * padded<uint32_t>(n)) + tb + padded<uint32_t>(AUC_PART_WORDS) + 2048);
  if (rc) return rc;
  Carver cv(c->scratch);
  float* d_l = cv.take<float>(n);
  float* d_p = cv.take<float>(n);
  double* d_o = cv.take<double>(1);
  uint32_t* d_part = cv.take<uint32_t>(AUC_PART_WORDS);
  hipStream_t s = c->stream;
  DFH_HIP(hipMemcpyAsync(d_l, label, n * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemcpyAsync(d_p, pred, n * 4, hipMemcpyHostToDevice, s));
  DFH_HIP(hipMemsetAsync(d_o, 0, sizeof(double), s));
  if (pairs) {
    hipLaunchKernelGGL(k_auc_pairs, dim3(auc_units((uint32_t)n)), dim3(256), 0, s, d_p, d_l, (uint32_t)n, d_part);
    hipLaunchKernelGGL(k_auc_finalize, dim3(1), dim3(256), 0, s, AucFin{d_part, (uint32_t)n, d_o});
  } else {
    uint32_t* k0 = cv.take<uint32_t>(n);
    uint32_t* k1 = cv.take<uint32_t>(n);
    uint32_t* l0 = cv.take<uint32_t>(n);
    uint32_t* l1 = cv.take<uint32_t>(n);
    char* temp = cv.take<char>(tb + 16);
    hipLaunchKernelGGL(k_auc_keys, dim3(grid_for_threads(n, c)), dim3(256), 0, s, d_p, d_l, (uint32_t)n, k0, l0);
    DFH_HIP(rocprim::radix_sort_pairs(temp, tb, k0, k1, l0, l1, n, 0, 32, s));
    hipLaunchKernelGGL(k_auc_area, dim3(1), dim3(1024), 0, s, l1, (uint32_t)n, d_o);
  }
  DFH_HIP(hipGetLastError());
  double o = 0;
  DFH_HIP(hipMemcpyAsync(&o, d_o, sizeof(double), hipMemcpyDeviceToHost, s));
  DFH_HIP(hipStreamSynchronize(s));
  *auc_n = (float)o;
  return DFH_OK;
}

}  // extern "C"

#include "dfh_shard.hip"
