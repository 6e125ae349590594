/*
 * difacto_hip.h — the C ABI of the MI355X-native FM/SGD worker path.
 *
 * This is the drop-in boundary: plain C, opaque handles, caller-owned buffers,
 * int return codes (0 = ok; message via dfh_last_error()).  Each entry point
 * names the reference interface (dmlc/difacto, file:line) it replaces, so the
 * C++ adaptors in difacto_amd/host/ (HipFMLoss : Loss, DeviceStore : Store) and
 * any other FFI (ctypes in difacto_amd/capi.py) bind 1:1.
 *
 * Conventions
 *   - "keys" are the byte-reversed feature ids the reference's Localizer emits
 *     (ReverseBytes(id % max_index), src/data/localizer.cc:24); ~0ULL is reserved.
 *   - host-pointer calls ("literal" API) are synchronous: on return the outputs
 *     are valid and the model state is updated.
 *   - device-pointer calls are asynchronous on the context's HIP stream; use
 *     dfh_ctx_sync() or stream-ordered work of your own.
 *   - there is NO CPU fallback: every compute entry point runs HIP kernels on
 *     the context's device and fails with DFH_ERR_HIP if that is impossible.
 */
#ifndef DIFACTO_HIP_H_
#define DIFACTO_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  DFH_OK = 0,
  DFH_ERR_ARG = 1,      /* bad argument / reference CHECK would fail */
  DFH_ERR_HIP = 2,      /* HIP runtime error */
  DFH_ERR_CAPACITY = 3, /* model table is full */
  DFH_ERR_STATE = 4     /* call sequence error (e.g. step before localize) */
};

/* Store value types — include/difacto/store.h:31-33 */
enum { DFH_FEA_COUNT = 1, DFH_WEIGHT = 2, DFH_GRADIENT = 3 };

/* how a lazily allocated embedding row is initialised */
enum {
  DFH_INIT_REFRAND = 0, /* bit-compatible with SGDUpdater::InitV's glibc rand_r chain
                           (src/sgd/sgd_updater.cc:140-147); serial in key order */
  DFH_INIT_HASH = 1     /* counter-based hash of (key, j, seed): order/sharding independent */
};

/* SGDUpdaterParam — src/sgd/sgd_param.h:66-107: same names, meanings, defaults */
typedef struct {
  float l1, l2, V_l2;
  float lr, lr_beta, V_lr, V_lr_beta;
  float V_init_scale;
  int V_dim;
  int V_threshold;
  unsigned seed;
  int init_mode; /* DFH_INIT_* */
} dfh_updater_param;

/* sgd::Progress — src/sgd/sgd_utils.h:40-75 */
typedef struct {
  float loss, penalty, auc, nnz_w, nrows;
} dfh_progress;

typedef struct dfh_ctx dfh_ctx;     /* a device + a HIP stream */
typedef struct dfh_table dfh_table; /* one shard of the model: replaces Store+Updater state */
typedef struct dfh_batch dfh_batch; /* a device-resident minibatch + its workspace */

const char* dfh_last_error(void);
void dfh_updater_param_default(dfh_updater_param* p, int V_dim);

/* ---------------------------------------------------------------- context */
/* stream: an existing hipStream_t (e.g. torch's current stream) or NULL to create one */
int dfh_ctx_create(int device, void* stream, dfh_ctx** out);
int dfh_ctx_destroy(dfh_ctx* ctx);
int dfh_ctx_sync(dfh_ctx* ctx);
void* dfh_ctx_stream(dfh_ctx* ctx);
int dfh_ctx_device(dfh_ctx* ctx);

/* Pipelining: with enable = n in 1..4 batch preparation (dfh_batch_load_*, dfh_localize,
 * dfh_batch_lookup) runs on n preparation streams of the lowest priority, taken round-robin, so
 * batches t+1 .. t+n are prepared while batch t trains — the overlap the reference gets from its
 * reader thread (src/sgd/sgd_learner.cc:196-224); steps are still applied strictly in batch order.
 * Ordering per batch object is kept with events inside the library; use n + 1 (better n + 2)
 * dfh_batch objects in rotation.  0 = everything on the context's stream. */
int dfh_ctx_set_pipeline(dfh_ctx* ctx, int enable);
/* launch tuning, validated (unknown names / out-of-range values: DFH_ERR_ARG):
 *   "fwd_depth"         4 | 5 | 8 | 10  independent V-row loads per forward lane (default 5)
 *   "fwd_blocks"        0 .. 16384      cap on the forward grid, 0 = one wave per example (default)
 *   "bwd_small_blocks"  1 .. 65536      cap on the short-segment blocks of the backward launch (default 2048)
 *   "prep_priority"     -1 | 0 | 1      stream priority of the preparation streams: lowest (default),
 *                                       normal, highest; before dfh_ctx_set_pipeline creates them
 *   "upd_kernel", "upd_*_blocks", "upd_interleave", "event_flags": see csrc/dfh_api.hip (measurement switches)
 *   "grow_initial_rows" 16 .. 2^28      first allocation of a growing table (dfh_table_create, capacity_rows = 0; default 2^20)
 *   "auc_in_update"     0 | 1           BinClassMetric::AUC of a training step (batch option "compute_auc") as the first
 *                                       blocks of the update launch (default 1) or as a launch of its own
 *   "single_queue"      0 | 1           1: no preparation stream — dfh_localize only notes the Localizer's four stages, which then
 *                                       ride as extra blocks of the launches later dfh_sgd_step calls make (the two minibatches the
 *                                       reference keeps in flight, src/sgd/sgd_learner.cc:196-224, on ONE hardware queue; prepare
 *                                       two ahead).  Same results bit for bit; pays for small minibatches only (default 0)
 *   "rider_slot_count / _scatter / _sort / _emit"  0 lookup | 1 forward | 2 update (+ 4: as a launch of its own before it):
 *                                       which launch of a step carries the stage;  "rider_period_lookup / _forward / _update"
 *                                       1 .. 4096 and "rider_start_*" 0 .. 100: where the rider blocks sit in the carrier's grid
 *   "upd_split"         0 | 1           keys with more than 4 096 occurrences in a minibatch go through the update kernel in
 *                                       parts of 1 024, a block per part (default 1; 0 = one block walks the whole segment)
 *   "owner_per_key"     0 | 1           dfh_shard_step's owner side per distinct key (dfh_shard_count_pull_multi +
 *                                       dfh_shard_push_grad_listed) instead of per received entry; same results, not faster
 *                                       (default 0; measurement switch) */
int dfh_ctx_set_option(dfh_ctx* ctx, const char* name, int value);

/* optional per-kernel timing with HIP events recorded on the context's stream
 * (what bench.py's roofline block reads); ids index total_ms[]/calls[] */
enum {
  DFH_K_LOCALIZE = 0, DFH_K_LOOKUP, DFH_K_FORWARD, DFH_K_BACKWARD, DFH_K_PULL, DFH_K_PUSH, DFH_K_MISC, DFH_K_AUC, DFH_K_COUNT
};
int dfh_ctx_set_timing(dfh_ctx* ctx, int enable);
/* time only the kernels whose bit (1u << DFH_K_*) is set.  Single-launch kernels (forward,
 * backward) carry their event pair on the dispatch itself (hipExtLaunchKernelGGL: the kernel's own
 * begin/end, as a profiler reports it); multi-launch groups (Localizer, lookup) are bracketed by
 * recorded events, each of which drains the stream (~10 us on MI355X) — so production loops
 * leave timing off and a benchmark samples: one kernel, every n-th step */
int dfh_ctx_set_timing_mask(dfh_ctx* ctx, uint32_t mask);
int dfh_ctx_get_timing(dfh_ctx* ctx, int reset, double* total_ms, uint64_t* calls); /* synchronises */
const char* dfh_kernel_name(int id);

/* ------------------------------------------------------- id transforms (a1) */
/* ReverseBytes / EncodeFeaGrpID — include/difacto/base.h:39-51,60-63 (host helpers) */
uint64_t dfh_reverse_bytes(uint64_t x);
uint64_t dfh_encode_fea_grp_id(uint64_t x, int gid, int nbits);

/* ------------------------------------------------------------ model table */
/* Replaces SGDUpdater's unordered_map<feaid_t,SGDEntry> (src/sgd/sgd_updater.h:78)
 * with a row store in HBM: capacity_rows fixed-stride rows + an open-addressing
 * key index.  Unseen keys are inserted as zero rows on first touch, exactly as
 * SGDUpdater::Get/Update do through model_[id] (sgd_updater.cc:44,66,87).
 * capacity_rows = 0: the table GROWS like the reference's map (it starts at 2^20 rows and is re-allocated at twice
 * the size whenever fewer than 32 launches' worth of new keys would still fit; row ids are stable, only the key index
 * is rebuilt; needs old + new arrays side by side, so a model beyond a third of the HBM wants an explicit capacity).
 * capacity_rows > 0: fixed; inserting beyond it is reported as DFH_ERR_CAPACITY, never silent.  At most 2^28 - 1 rows (the four top bits of a row word carry flags). */
int dfh_table_create(dfh_ctx* ctx, const dfh_updater_param* p, uint64_t capacity_rows, dfh_table** out);
/* rows the arrays hold now, and how often the table has grown (0 for a fixed capacity) */
int dfh_table_capacity(dfh_table* t, uint64_t* capacity_rows, uint64_t* grows);
int dfh_table_destroy(dfh_table* t);
int dfh_table_size(dfh_table* t, uint64_t* nkeys); /* synchronises */
int dfh_table_param(dfh_table* t, dfh_updater_param* out);
uint64_t dfh_table_bytes(dfh_table* t);
/* SGDUpdater::has_aux_ (src/sgd/sgd_updater.h:80): cleared by dfh_table_load of a file saved
 * without optimiser state (or by a host that parsed such a file itself); while cleared, every
 * gradient push and training step fails with DFH_ERR_STATE — the reference's
 * CHECK(has_aux_) << "no aux data" (src/sgd/sgd_updater.cc:75) */
int dfh_table_set_has_aux(dfh_table* t, int has_aux);
int dfh_table_has_aux(dfh_table* t);

/* Store::Pull(fea_ids, kWeight, vals, lens) -> SGDUpdater::Get
 * (include/difacto/store.h:69-73, src/store/store_local.h:36-44, src/sgd/sgd_updater.cc:32-56).
 * Host pointers.  vals capacity n*(1+V_dim), lens capacity n; ragged layout and
 * lens in {1, 1+V_dim}; *nlens == 0 iff V_dim == 0 — as the reference. */
int dfh_pull(dfh_table* t, const uint64_t* keys, size_t n, float* vals, size_t* nvals, int* lens, size_t* nlens);

/* Store::Push(fea_ids, kFeaCount|kGradient, vals, lens) -> SGDUpdater::Update
 * (include/difacto/store.h:53-57, src/store/store_local.h:24-34, src/sgd/sgd_updater.cc:58-148):
 * FTRL on w, AdaGrad on V, lazy InitV.  Host pointers.  Keys must be unique (checked: DFH_ERR_ARG). */
int dfh_push(dfh_table* t, const uint64_t* keys, size_t n, int val_type, const float* vals, size_t nvals,
             const int* lens, size_t nlens);

/* model I/O helpers (Updater::Save/Load are TODO in the reference, src/sgd/sgd_updater.h:44-50):
 * dump every entry: keys[n], scal[n*4] = {fea_cnt,w,sqrt_g,z}, has_V[n], V[n*2*V_dim] */
int dfh_table_export(dfh_table* t, uint64_t cap, uint64_t* keys, float* scal, int* has_V, float* V, uint64_t* n);
int dfh_table_import(dfh_table* t, uint64_t n, const uint64_t* keys, const float* scal, const int* has_V, const float* V);
/* Updater::Save / Load to a file (include/difacto/updater.h:40-47; TODO stubs in the reference,
 * src/sgd/sgd_updater.h:44-50, so the format is ours — the one the C++ host's --model_out / --model_in
 * read and write).  save_aux: also keep fea_cnt, the FTRL state and the AdaGrad accumulators, i.e.
 * everything needed to continue training; without it, entries with w == 0 and no V are dropped.
 * dfh_table_load imports only keys in [key_lo, key_hi) (key_hi == 0: no upper bound): a shard
 * loads its own range out of any number of part files, whatever sharding wrote them.
 * Host-side I/O; both synchronise. */
int dfh_table_save(dfh_table* t, const char* path, int save_aux, uint64_t* n_saved);
int dfh_table_load(dfh_table* t, const char* path, uint64_t key_lo, uint64_t key_hi, int* has_aux, uint64_t* n_loaded);

/* warm start (resume / benchmark preload): insert n unique keys (DEVICE pointer)
 * with w = w0, fea_cnt = cnt0 and an allocated V (hash init).  Asynchronous. */
int dfh_table_warm_start(dfh_table* t, const uint64_t* d_keys, size_t n, float w0, float cnt0);

/* ------------------------------------------------ literal Loss API (a6-a8) */
/* FMLoss::Predict (src/loss/fm_loss.h:67-119).  Host pointers; pred is
 * accumulated into (caller zeroes it, src/sgd/sgd_learner.cc:142).  w_pos/V_pos
 * NULL => dense weights, no V (the V_dim==0 / LogitLoss case). */
int dfh_fm_predict(dfh_ctx* ctx, int V_dim, size_t nrows, const size_t* offset, const uint32_t* index,
                   const float* value, const float* weights, size_t nweights, const int* w_pos,
                   const int* V_pos, size_t npos, float* pred);

/* FMLoss::CalcGrad (src/loss/fm_loss.h:148-199).  Host pointers; grad (shape of
 * weights) is accumulated into.  Stateless: recomputes X*V instead of relying
 * on the preceding Predict's members. */
int dfh_fm_calcgrad(dfh_ctx* ctx, int V_dim, size_t nrows, const size_t* offset, const uint32_t* index,
                    const float* value, const float* label, const float* weights, size_t nweights,
                    const int* w_pos, const int* V_pos, size_t npos, const float* pred, float* grad);

/* Loss::Evaluate (include/difacto/loss.h:57-66): sum_i log(1+exp(-y_i pred_i)) */
int dfh_loss_evaluate(dfh_ctx* ctx, const float* label, const float* pred, size_t n, float* objv);

/* BinClassMetric::AUC (src/loss/bin_class_metric.h:35-56): returns AUC * n */
int dfh_auc_times_n(dfh_ctx* ctx, const float* label, const float* pred, size_t n, float* auc_n);

/* ------------------------------------------------- device-resident batches */
int dfh_batch_create(dfh_ctx* ctx, size_t max_rows, size_t max_nnz, dfh_batch** out);
/* n batch objects of one size out of ONE device allocation (freed with the last of them): what a worker loop that rotates a
 * dozen objects creates at the start of a job — one allocation instead of n (a job's start-up) */
int dfh_batch_create_many(dfh_ctx* ctx, int n, size_t max_rows, size_t max_nnz, dfh_batch** out);
int dfh_batch_destroy(dfh_batch* b);

/* raw minibatch = what Reader::Value() hands the worker (dmlc::RowBlock<feaid_t>,
 * src/reader/reader.h:49-51): CSR with u64 feature ids.  value may be NULL (binary).  The arrays
 * are staged in page-locked memory of the batch object and copied asynchronously: on return the
 * caller may reuse them, and nothing has waited for the device. */
int dfh_batch_load_host(dfh_batch* b, size_t nrows, const size_t* offset, const uint64_t* index,
                        const float* value, const float* label);
/* same, from DEVICE pointers (u32 offsets); copies into the batch's own buffers */
int dfh_batch_load_device(dfh_batch* b, size_t nrows, size_t nnz, const uint32_t* d_offset,
                          const uint64_t* d_index, const float* d_value, const float* d_label);

/* same, WITHOUT a copy: the batch reads the caller's device arrays in place; they must stay
 * valid and unchanged until the work queued on this batch has finished */
int dfh_batch_attach_device(dfh_batch* b, size_t nrows, size_t nnz, const uint32_t* d_offset,
                            const uint64_t* d_index, const float* d_value, const float* d_label);

/* Localizer::Compact on device (src/data/localizer.h:41-51, localizer.cc:11-103):
 * keys = ReverseBytes(id % max_index), sorted unique keys + counts + compact
 * index per nnz — bit-exact with the reference — plus the key-ordered view the
 * backward pass's segmented sum walks. */
int dfh_localize(dfh_batch* b, uint64_t max_index);
/* dfh_localize is a pure function of the minibatch, but a batch object remembers the exact quantiles
 * of the last minibatch it localized and partitions the next one with them (consecutive minibatches
 * of one stream share their key distribution); they affect speed only, never the result.
 * tuning / test switches: "force_radix_sort" = 1 makes dfh_localize take its
 * large-batch path (library LSD radix sort) whatever the batch size;
 * "force_sort_fallback" = 1 sorts every bucket through the oversize-bucket path;
 * "reset_splitters" = 1 forgets the stored quantiles (the next call samples again);
 * "compute_auc" = 1 makes dfh_sgd_step accumulate BinClassMetric::AUC (x nrows) of every
 * batch into dfh_progress.auc, as src/sgd/sgd_learner.cc:153-155 does */
int dfh_batch_set_option(dfh_batch* b, const char* name, int value);

/* resolve the batch's unique keys to table rows ahead of dfh_sgd_step (inserting
 * unseen keys as zero rows, src/sgd/sgd_updater.cc:44): the random probes of the key index run
 * with the preparation work, and the step's own pass over the keys (count push, current w) finds
 * the rows known.  Optional: dfh_sgd_step probes itself when this was not called. */
int dfh_batch_lookup(dfh_table* t, dfh_batch* b);
/* dfh_localize and dfh_batch_lookup in one call: the Localizer's last pass (the thread that writes a unique key has it
 * in hand) probes the key index itself — one launch and one cross-stream event fewer per minibatch
 * (src/data/localizer.cc:11-103 followed by src/sgd/sgd_updater.cc:32-56's model_[id] lookups).  Same results as the two
 * calls; on MI355X at the C3 size the two calls are 0.4 % faster per step (the longer last pass runs beside the forward
 * kernel), so the worker loops use those. */
int dfh_localize_lookup(dfh_table* t, dfh_batch* b, uint64_t max_index);

/* Device feed.  BatchReader's shuffle buffer (src/reader/batch_reader.cc:38-52: batch_size x shuffle rows, permuted, cut into
 * minibatches) held in HBM: the host uploads every buffer once (dfh_rowbuf_load_host, callable from a reader thread: it uses a
 * stream of its own and returns when the caller's arrays are free) and a minibatch is gathered out of at most a few buffers
 * by row number on the device (dfh_batch_gather_rows, in place of dfh_batch_load_host): `offset` [nrows + 1] and `label`
 * [nrows] are the minibatch's own (the host knows the row lengths), segment g takes rows[g][0 .. seg_rows[g]) of bufs[g], in
 * order.  A buffer without values counts as all ones.  The same minibatch as the host-side gather, byte for byte. */
typedef struct dfh_rowbuf dfh_rowbuf;
int dfh_rowbuf_create(dfh_ctx* c, size_t max_rows, size_t max_nnz, dfh_rowbuf** out);
int dfh_rowbuf_destroy(dfh_rowbuf* rb);
int dfh_rowbuf_load_host(dfh_rowbuf* rb, size_t nrows, const size_t* offset, const uint64_t* index, const float* value);
/* the same for a buffer whose ids / values were never assembled on the host: `offset` [nrows + 1] are the buffer's own,
 * slice g brings nnz[g] ids (and values, or NULL = all ones) that follow those of slice g - 1 — one copy per slice, straight
 * out of the parser's chunks (round 4: assembling 31 MB per buffer on one host thread was what the worker loop waited for) */
int dfh_rowbuf_load_host_slices(dfh_rowbuf* rb, size_t nrows, const size_t* offset, int nslices, const uint64_t* const* index,
                                const float* const* value, const size_t* nnz);
int dfh_batch_gather_rows(dfh_batch* b, size_t nrows, const size_t* offset, const float* label, int nseg, dfh_rowbuf* const* bufs,
                          const uint32_t* const* rows, const size_t* seg_rows);
/* dfh_batch_gather_rows + dfh_localize + dfh_batch_lookup as ONE preparation phase — what a worker loop queues per
 * minibatch (src/sgd/sgd_learner.cc:196-224: read, localize, pull): the minibatch's description is read by the gather
 * kernel where the host wrote it (page-locked, mapped: no copy is queued), the phase records one cross-stream event.
 * Same results as the three calls; about half their host time per minibatch. */
int dfh_batch_prepare_rows(dfh_table* t, dfh_batch* b, size_t nrows, const size_t* offset, const float* label, int nseg,
                           dfh_rowbuf* const* bufs, const uint32_t* const* rows, const size_t* seg_rows, uint64_t max_index);

/* already-localized batch from the host (what SGDLearner hands its batch thread,
 * src/sgd/sgd_learner.cc:203-212): feaids sorted unique, compact u32 index */
int dfh_batch_load_localized_host(dfh_batch* b, size_t nrows, const size_t* offset, const uint32_t* index,
                                  const float* value, const float* label, const uint64_t* feaids,
                                  const float* feacnt, size_t U);
/* read back the localizer's outputs (any may be NULL); synchronises */
int dfh_batch_get_localized(dfh_batch* b, size_t* U, uint64_t* feaids, float* feacnt, uint32_t* index);
int dfh_batch_shape(dfh_batch* b, size_t* nrows, size_t* nnz, size_t* U); /* synchronises for U */

/* ---------------------------------------------------------- the fused step */
/* The batch executor of SGDLearner::IterateData (src/sgd/sgd_learner.cc:131-178)
 * for one localized batch, entirely on device:
 *   [push_cnt: Push(kFeaCount)] -> Pull -> Predict -> Evaluate(+penalty,+AUC)
 *   -> [is_train: CalcGrad -> Push(kGradient) -> FTRL/AdaGrad in place].
 * Asynchronous; metrics accumulate in the batch and are read with dfh_batch_progress. */
int dfh_sgd_step(dfh_table* t, dfh_batch* b, int is_train, int push_cnt);
int dfh_batch_progress(dfh_batch* b, dfh_progress* out, int reset); /* synchronises */
int dfh_batch_get_pred(dfh_batch* b, float* pred);                  /* synchronises */

/* -------------------------------------------- sharded (multi-GPU) building blocks */
/* Rows cross the wire in a fixed-stride layout of dfh_row_stride(V_dim) floats:
 *   [w, has_V (0/1 as float), 0, 0 | V[0..V_dim) zero-padded to a multiple of 4]
 * gradients use the same layout: [gw, has_V, 0, 0 | gV...].  All DEVICE pointers. */
size_t dfh_row_stride(int V_dim);
/* owner side of Pull: look up / insert n unique keys, write rows */
int dfh_shard_pull(dfh_table* t, const uint64_t* d_keys, size_t n, float* d_rows);
/* owner side of Push(kFeaCount) and Push(kGradient) for n unique keys */
int dfh_shard_push_count(dfh_table* t, const uint64_t* d_keys, size_t n, const float* d_cnt);
int dfh_shard_push_grad(dfh_table* t, const uint64_t* d_keys, size_t n, const float* d_grads);
/* worker side: Predict+Evaluate on pulled rows [U x stride], then (is_train)
 * CalcGrad into d_grads [U x stride]; pointers to the batch's device feaids/cnt */
int dfh_batch_forward(dfh_batch* b, int V_dim, const float* d_rows);
int dfh_batch_backward(dfh_batch* b, int V_dim, const float* d_rows, float* d_grads);
int dfh_batch_device_keys(dfh_batch* b, const uint64_t** d_feaids, const float** d_feacnt, size_t* U);
/* key-range partition of the batch's (ascending) unique keys over nparts shards:
 * shard d owns keys in [d*span, (d+1)*span), span = ceil(2^64/nparts) — the range
 * partitioning ReverseBytes exists for (include/difacto/base.h:29-38).  HOST output
 * bounds[nparts+1]: shard d gets feaids[bounds[d] .. bounds[d+1]).  Synchronises. */
int dfh_batch_key_ranges(dfh_batch* b, int nparts, uint32_t* bounds);
/* the same bounds as int64 into DEVICE memory d_bounds[nparts+1], enqueued on the main
 * stream after the batch's preparation; does not synchronise (d_bounds[nparts] = U).
 * d_splits == NULL: the uniform partition above.  Otherwise d_splits[nparts-1] (device,
 * ascending): shard d >= 1 owns keys in [d_splits[d-1], d_splits[d]) — still contiguous key
 * ranges, with boundaries the caller balanced for its id space (feature-group ids sit in the
 * top bits of a reversed key, base.h:60-63, so uniform ranges can be badly skewed). */
int dfh_batch_key_ranges_device(dfh_batch* b, int nparts, const uint64_t* d_splits, int64_t* d_bounds);

/* Owner side in resolved form, for keys arriving from several source ranks in one step
 * (StoreLocal::Push/Pull per source, src/store/store_local.h:24-44; SGDUpdater::Get/Update,
 * src/sgd/sgd_updater.cc:32-102).  dfh_shard_resolve probes/inserts ALL received keys once
 * (the same key may appear under several sources) and writes their row ids; Pull is then one
 * gather over all of them; the two Push kinds take one source's slice at a time (keys unique
 * inside a call) and are applied in call order.  V_init must be "hash". */
int dfh_shard_resolve(dfh_table* t, const uint64_t* d_keys, size_t n, uint32_t* d_rowid);
int dfh_shard_pull_resolved(dfh_table* t, const uint32_t* d_rowid, size_t n, float* d_rows);
int dfh_shard_push_count_resolved(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, size_t n, const float* d_cnt);
int dfh_shard_push_grad_resolved(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, size_t n, const float* d_grads);
/* The same for ALL source ranks of a step in one launch per operation.  d_keys is the
 * concatenation of nsrc ascending key lists, source s holding entries [seg[s], seg[s+1]) (seg:
 * HOST array of nsrc+1 offsets, nsrc <= 32; n = seg[nsrc] < 2^27 entries).  d_rowid holds
 * dfh_shard_multi_words(n, nsrc) words: [0, n) the entries' row words (the row id in the low 29
 * bits — what dfh_shard_pull_resolved reads), behind them, per entry, the indices of the other
 * entries that carry the same key (written by resolve_multi for ONE entry per key, the first to
 * be resolved; read by the Push calls).  In the two Push calls that entry applies every
 * source's value in ascending source order (the result of nsrc per-source calls) and stores the
 * row once.  push_grad_multi ends the step for these rows; a step without it (validation) ends
 * with dfh_shard_release.  mask_slot (0 or 1) names which of two per-row step words the step
 * uses: an owner may hold two steps at once — one resolved and pulled, the other still awaiting
 * its gradients (the reference keeps two minibatches in flight, sgd_learner.cc:219-223) — and
 * they must use different slots. */
size_t dfh_shard_multi_words(size_t n, int nsrc);
int dfh_shard_resolve_multi(dfh_table* t, const uint64_t* d_keys, const size_t* seg, int nsrc, int mask_slot,
                            uint32_t* d_rowid);
int dfh_shard_push_count_multi(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc,
                               int mask_slot, const float* d_cnt);
int dfh_shard_push_grad_multi(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc,
                              int mask_slot, const float* d_grads);
/* The owner side per DISTINCT key (what dfh_shard_step uses): dfh_shard_count_pull_multi is Push(kFeaCount) of all sources
 * (d_cnt; NULL = no counts this step) and Pull in one launch — a key's row is read once and written to the output row of
 * every entry that carries the key (d_rows: n rows of dfh_row_stride floats, entry order) — and leaves, behind the extras in
 * d_rowid, the step's keys as lists; dfh_shard_push_grad_listed is Push(kGradient) of all sources over those lists (same
 * seg / mask_slot; d_grads in entry order; ends the step like push_grad_multi).  Results equal the per-entry calls above. */
int dfh_shard_count_pull_multi(dfh_table* t, uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc, int mask_slot,
                               const float* d_cnt, float* d_rows);
int dfh_shard_push_grad_listed(dfh_table* t, const uint32_t* d_rowid, const uint64_t* d_keys, const size_t* seg, int nsrc,
                               int mask_slot, const float* d_grads);
int dfh_shard_release(dfh_table* t, const uint32_t* d_rowid, size_t n, int mask_slot);
/* fetch and report the table's sticky device-side error word (capacity exceeded, gradient
 * with V for a row without V); synchronises */
int dfh_table_check(dfh_table* t);

/* ------------------------------------------------------ sharded store (multi-GPU) */
/* The ps-lite Push / Pull of the reference (include/difacto/store.h:53-93, src/store) for the ranks
 * of one node, one process per GPU: rank r owns a contiguous range of the reversed keys (uniform:
 * [r*span, (r+1)*span), span = ceil(2^64/world); or the explicit split keys given to
 * dfh_shard_create) and trains its own minibatches.  Nothing here is a collective on host data:
 * all traffic is device buffers over the communicator. */
typedef struct dfh_comm dfh_comm;
typedef struct dfh_shard dfh_shard;
#define DFH_COMM_ID_BYTES 128
/* RCCL transport (ncclSend / ncclRecv over xGMI).  Rank 0 obtains an id and hands it to the other
 * ranks out of band (a file, an env var, MPI ...); every rank then creates its communicator. */
int dfh_comm_unique_id(void* id128);
int dfh_comm_create_rccl(dfh_ctx* ctx, int rank, int world, const void* id128, dfh_comm** out);
/* Host-callback transport: the library stages every exchange through host memory and calls
 * fn(user, send, send_bytes[world], recv, recv_bytes[world]) — an all-to-all-v of bytes, contiguous in peer
 * order on both sides; 0 = ok.  For process groups RCCL cannot serve (several ranks sharing one GPU in
 * a test, a gloo / MPI / socket fabric of the host). */
typedef int (*dfh_alltoallv_fn)(void* user, const void* send, const size_t* send_bytes, void* recv, const size_t* recv_bytes);
int dfh_comm_create_callback(dfh_ctx* ctx, int rank, int world, dfh_alltoallv_fn fn, void* user, dfh_comm** out);
/* Loop-back transport — MEASUREMENT ONLY: rank `rank` of a `world`-rank job alone on its GPU.  Every exchange of
 * dfh_shard_step moves messages of the exact sizes a real job would move, as device-to-device copies instead of wires:
 * what this rank "sends" is read out of its send buffer (and dropped), what it "receives" is copied out of a buffer the
 * caller has FED for that kind of exchange (dfh_comm_loopback_feed) — laid out like the receive side of the exchange
 * (contiguous in peer order, or at the receive offsets) — so that every owner-side and worker-side kernel of the step
 * runs at its real size on valid data, on a chip no other rank shares (Store::Push / Pull of the reference with the
 * seven other workers played back: include/difacto/store.h:53-93, src/sgd/sgd_learner.cc:78-89).  The rank's own slot
 * of an exchange is a real self-copy.  An exchange whose kind was not fed echoes the rank's own send data.
 * Optional wire model: an exchange holds its stream for latency_us + (largest per-peer message, either direction) /
 * link_gbps (one xGMI link per peer and direction), whichever of that and the copies is longer; link_gbps = 0: copies only. */
enum { DFH_XCHG_COUNTS = 0, DFH_XCHG_KEYS, DFH_XCHG_CNT, DFH_XCHG_ROWS, DFH_XCHG_GRADS, DFH_XCHG_OTHER, DFH_XCHG_KINDS };
int dfh_comm_create_loopback(dfh_ctx* ctx, int rank, int world, dfh_comm** out);
/* the source of the NEXT exchange of `kind` (a FIFO per kind; sticky != 0: of every later exchange of that kind
 * until the next feed).  d_src must stay valid until that exchange has run. */
int dfh_comm_loopback_feed(dfh_comm* c, int kind, const void* d_src, int sticky);
int dfh_comm_loopback_wire(dfh_comm* c, double link_gbps, double latency_us);
/* modelled wire time (microseconds) of all exchanges since the last reset */
int dfh_comm_loopback_wire_time(dfh_comm* c, int reset, double* us);
int dfh_comm_destroy(dfh_comm* c);
int dfh_comm_rank(dfh_comm* c);
int dfh_comm_world(dfh_comm* c);
/* sum of n <= 64 host doubles over the ranks (same result on every rank): progress merging, votes */
int dfh_comm_allreduce_sum(dfh_comm* c, double* vals, int n);
/* every rank's `bytes` host bytes to every rank: recv holds world * bytes, in rank order.  COLLECTIVE. */
int dfh_comm_allgather(dfh_comm* c, const void* send, size_t bytes, void* recv);
/* payload bytes this rank has sent to / received from OTHER ranks and the number of message groups, since the last
 * reset (what the N > 1 bench line prices against the xGMI links: `roofline_exchange`) */
int dfh_comm_stats(dfh_comm* c, int reset, uint64_t* bytes_sent, uint64_t* bytes_recv, uint64_t* groups);
/* which transport is bound: "rccl <version> from <file> (already loaded by the host process | loaded by libdifacto_hip)"
 * or "host callback transport".  A host that already carries an RCCL (PyTorch bundles one) shares it. */
int dfh_comm_info(dfh_comm* c, char* buf, size_t n);
/* start-up self-check, collective: a first exchange of the ranks' ids, polled; DFH_ERR_STATE after timeout_s seconds
 * if a peer never joins (bad rendezvous) or if the ids do not add up — fails loudly instead of hanging the first step.
 * The reference's Store has no such call: its ps-lite van blocks in Postoffice::Barrier (include/difacto/store.h:53-93
 * is the interface this transport serves). */
int dfh_comm_selfcheck(dfh_comm* c, double timeout_s);
/* what the wires give: `reps` grouped exchanges (after two untimed ones) in which this rank sends bytes_per_peer to and
 * receives bytes_per_peer from EVERY other rank — the shape of dfh_shard_step's key / row / gradient exchanges (the
 * reference's ps-lite Push / Pull traffic, include/difacto/store.h:53-93), every xGMI link of a full mesh carrying one
 * message each way at once — timed on the context's stream.  *us_per_exchange = the average; bytes_per_peer over it is
 * what one link gave per direction.  COLLECTIVE; 0 with one rank.  bench.py --gpus N runs it before the timed region. */
int dfh_comm_wire_probe(dfh_comm* c, size_t bytes_per_peer, int reps, double* us_per_exchange);
/* Split keys that balance the shards on the DATA instead of on the key space: every rank hands in a sample of the
 * reversed keys it will see (n may differ per rank, 0 allowed); the samples are gathered and splits[world-1] receives
 * the (identical on every rank) quantiles of their union — the argument for dfh_shard_create.  With feature-group ids
 * in the low bits of an id (EncodeFeaGrpID, include/difacto/base.h:60-63) ReverseBytes moves them to the top of the
 * key and a uniform cut of the key space is badly skewed (39 criteo slots over 8 shards: one shard 2.3x the
 * average).  COLLECTIVE.  With no key in any sample the uniform split keys are returned. */
int dfh_shard_balanced_splits(dfh_comm* c, const uint64_t* sample_keys, size_t n, uint64_t* splits);

/* this rank's shard of the model + the exchange buffers.  splits: world-1 ascending first keys of
 * shards 1.., identical on every rank, or NULL for the uniform ranges.  The table must use
 * DFH_INIT_HASH (order- and sharding-independent V init). */
int dfh_shard_create(dfh_table* t, dfh_comm* c, const uint64_t* splits, dfh_shard** out);
int dfh_shard_destroy(dfh_shard* s);
/* [key_lo, key_hi) owned by this rank (key_hi = 0: no upper bound) — the range to hand dfh_table_load */
int dfh_shard_owned_range(dfh_shard* s, const uint64_t* splits, uint64_t* key_lo, uint64_t* key_hi);
/* One step of SGDLearner::IterateData (src/sgd/sgd_learner.cc:131-178) against the sharded model:
 * keys -> owners, rows back, Predict / Evaluate / CalcGrad on the pulled rows, gradients -> owners,
 * applied in ascending source-rank order (zero staleness).  COLLECTIVE: every rank calls it the same
 * number of times with the same is_train / push_cnt; a rank whose data is exhausted passes
 * b = NULL and keeps serving its shard.  *any_active = 0 when no rank had a minibatch (the step was a
 * no-op: the epoch is over).  Asynchronous apart from one small host wait; progress accumulates in
 * the batch as in dfh_sgd_step. */
int dfh_shard_step(dfh_shard* s, dfh_batch* b, int is_train, int push_cnt, int* any_active);
/* Optional, COLLECTIVE like the step: called on every rank right before dfh_shard_step, it names the
 * minibatch of the FOLLOWING step (its dfh_localize already queued; NULL if this rank has none left).
 * The step then sends that minibatch's per-owner key counts on their way inside itself — behind its own
 * backward pass, while its gradients travel — and the following dfh_shard_step (which must be given
 * exactly that batch) starts with the counts already on the host instead of waiting for them mid-way:
 * what the reference's batch tracker achieves by keeping two minibatches in flight
 * (sgd_learner.cc:219-223), here without staleness.  No-op with one rank. */
int dfh_shard_prefetch_counts(dfh_shard* s, dfh_batch* b_next);
/* How dfh_shard_step moves the rows.  0 (default) = sync: one minibatch at a time, every exchange on the context's
 * stream, zero staleness.  1 = overlap: TWO minibatches in flight, what the reference's batch tracker does
 * (src/sgd/sgd_learner.cc:219-223: the next minibatch is issued while one is pending, so its Pull may be served
 * before the previous Push has landed).  Inside dfh_shard_step(b) the minibatch named by the preceding
 * dfh_shard_prefetch_counts has its counts, keys and rows exchanged on a second stream while b computes:
 *     collectives' stream:  counts(t+1)  K(t+1)              G(t)            RW(t+1)
 *     main stream:          L(t) F(t) [gradient rows | own keys updated in place]  R(t+1)        P(t)
 * so the gradients of t travel while the owners pull for t+1, and the rows of t+1 travel while the gradients of t
 * are applied.  Staleness: the keys a rank owns itself are read with zero staleness (L(t+1) runs after P(t)); rows
 * pulled from other owners miss the gradients the OTHER ranks computed in step t (staleness 1, exactly one
 * minibatch), and contain the owner's own step-t update.  Every table operation runs on the main stream in program
 * order: nothing races on a row, every pulled row is one consistent version of its key.  The announced minibatch
 * is pulled (and its epoch-0 counts pushed) with the push_cnt of the call it rides in: the flag is a property of the
 * job (src/sgd/sgd_learner.cc:201-202), keep it constant from one announcement to the step.  Call between steps only
 * (no minibatch under way); COLLECTIVE in effect: every rank must use the same mode. */
int dfh_shard_set_exchange(dfh_shard* s, int mode);
/* Exchange buffers for minibatches of up to batch_keys unique keys and up to recv_keys keys received from the other
 * ranks per step, allocated NOW (after dfh_shard_set_exchange: the overlapped exchange keeps two sets).  Optional: without
 * it — and for a minibatch beyond these sizes — dfh_shard_step grows the buffers when it first meets the size, which
 * drains the streams and re-allocates in the middle of a step (the reference's store has no counterpart: ps-lite
 * allocates per message).  With key ranges balanced on the data, recv_keys ~ batch_keys; 2 x is generous. */
int dfh_shard_reserve(dfh_shard* s, size_t batch_keys, size_t recv_keys);
/* Per-stage device time of the steps since the last reset (HIP events around every stage; they cost the streams a
 * few microseconds each: for a diagnostic pass, not for the timed run).  ms[DFH_SHARD_STAGES]: counts, L (own keys'
 * lookup), K, R, RW, F (forward + backward / own update), G, P; *steps = steps covered. */
enum { DFH_SHARD_STAGE_COUNTS = 0, DFH_SHARD_STAGE_L, DFH_SHARD_STAGE_K, DFH_SHARD_STAGE_R, DFH_SHARD_STAGE_RW, DFH_SHARD_STAGE_F,
       DFH_SHARD_STAGE_G, DFH_SHARD_STAGE_P, DFH_SHARD_STAGES };
int dfh_shard_set_timing(dfh_shard* s, int enable);
int dfh_shard_get_timing(dfh_shard* s, int reset, double* ms, uint64_t* steps);

/* The literal Store::Pull / Store::Push (include/difacto/store.h:53-73) against the SHARDED model, with host arrays in the
 * layout of dfh_pull / dfh_push: what the reference's worker loop calls once per minibatch when it runs call by call
 * (device_path = literal) instead of through dfh_shard_step.  keys: ascending, unique (a Localizer's output).
 * COLLECTIVE: every rank makes the same sequence of calls with the same val_type; a rank with nothing to ask passes
 * n = 0 and still serves its shard.  One exchange each way per call, everything on the context's stream, synchronous.
 * Pushes from several ranks to one key are applied in ascending rank order (FTRL / AdaGrad are not linear: the order is
 * part of the result, and fixed). */
int dfh_shard_pull_host(dfh_shard* s, const uint64_t* keys, size_t n, float* vals, size_t* nvals, int* lens, size_t* nlens);
int dfh_shard_push_host(dfh_shard* s, const uint64_t* keys, size_t n, int val_type, const float* vals, size_t nvals, const int* lens,
                        size_t nlens);

/* raw device memory for hosts without a HIP runtime of their own */
int dfh_malloc(dfh_ctx* ctx, size_t bytes, void** dptr);
int dfh_free(dfh_ctx* ctx, void* dptr);
int dfh_memcpy_h2d(dfh_ctx* ctx, void* dst, const void* src, size_t bytes);
int dfh_memcpy_d2h(dfh_ctx* ctx, void* dst, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* DIFACTO_HIP_H_ */
