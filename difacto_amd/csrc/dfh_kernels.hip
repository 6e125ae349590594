// HIP kernels of the FM/SGD worker path for gfx950 (MI355X, CDNA4).
//
// Kernel inventory (DESIGN.md has the roofline of each):
//   k_lookup              keys -> table rows (insert-on-miss) [+ Push(kFeaCount)], leaves {row, w}
//                         per key for the forward
//   k_forward<L,D>        fused gather + FMLoss::Predict + logistic slope + logloss
//   k_seg_lists           compacts the keys with long segments into the mid / hot lists
//   k_backward_all<L,F,LEAN,EXACT>   segmented sum over duplicate keys = FMLoss::CalcGrad in ONE
//                         launch (hot / mid / short-segment roles by block range);
//                         F=1: fused in-place FTRL/AdaGrad (SGDUpdater::Update)
//   k_resolve / k_pull_resolved / k_push_grad_resolved (+ k_pull_rows / k_push_grad)
//                         owner side of the sharded store
//   k_refrand_*           rand_r-compatible lazy InitV (parity mode)
//   k_predict_generic / k_calcgrad_generic / k_logloss   literal Loss API
//   k_auc_pairs           BinClassMetric::AUC by pair counting (minibatch-sized n); k_auc_keys / k_auc_area
//                         around a library radix sort beyond
//   k_rdx_*               Localizer::Compact around a library radix sort (very large batches);
//                         the sample-sort Localizer (k_loc_*) lives in dfh_localize.hip
//
// All kernels assume 64-lane wavefronts and are launched with 256-thread
// blocks (4 waves) unless noted.
#include "dfh_internal.h"

namespace dfh {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over lanes whose ids differ only in bits >= log2(L) (same "sub" lane of every group)
template <int L>
__device__ __forceinline__ float cross_group_sum(float v) {
#pragma unroll
  for (int o = L; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the L lanes of one group
template <int L>
__device__ __forceinline__ float in_group_sum(float v) {
#pragma unroll
  for (int o = L >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// streaming variants (nt): model rows are touched once per step; keeping them out of the L2's
// way leaves the batch-sized arrays (XV, slopes, occurrence lists) resident there
typedef float nt_float4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  nt_float4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_float4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  nt_float4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<nt_float4*>(p));
}

// acc + a * b the way SpMV does it (src/common/spmv.h:125, :155): an operand `a` that is exactly zero is SKIPPED, not
// multiplied.  Neutral for finite b; for b = Inf / NaN (a non-finite feature value) it is the difference between the
// reference's result and NaN: a zero weight (pulled as 0: no entry yet, or clipped by l1) and a slope that underflowed to 0
// do not see the value at all.  V rows: a key WITHOUT V is skipped by SpMM (V_pos = -1, spmm.h:108) and is an all-zero row
// here, loaded speculatively; the forward multiplies it by x = 0 instead of the value, decided by the key's has-V flag
// (kHasV in its row word, the header flag of packed rows) — an ALLOCATED row is used whatever it holds, as in the reference
// (round 6: the test of the loaded slice for all-zero that stood in for the flag through round 5 is gone, and with it the
// one documented deviation).
__device__ __forceinline__ float fma_skip0(float a, float b, float acc) { return a != 0.f ? __builtin_fmaf(a, b, acc) : acc; }

// The sharding-independent V init: must match oracle/difacto_oracle.c:orc_hash_init_value
__device__ __forceinline__ float hash_init_value(uint64_t key, int j, unsigned seed, float scale) {
  uint64_t a = splitmix64(key ^ (0xD1B54A32D192ED03ULL * ((uint64_t)seed + 1ULL)));
  uint64_t h = splitmix64(a + (uint64_t)j);
  uint32_t r = (uint32_t)(h >> 40);
  float u = (float)r * (1.0f / 16777216.0f);
  return (u - 0.5f) * scale;
}

// ---- key index: open addressing, linear probing.  Launches that probe the table may run
// concurrently (preparation streams resolve the keys of later minibatches while the main stream
// pulls, pushes or imports) and may carry the same new key, inside one launch (owner side of the
// sharded store: several source ranks) or across launches.  The thread that wins a slot's 64-bit
// CAS takes the next row and publishes its id with a device-scope atomic store; every other
// carrier of the key finds the key in the slot and waits for the id (the table is created with
// all row words = ~0).  A row id is the whole message: rows are zero until an update writes them.
__device__ __forceinline__ uint32_t find_or_insert(const TableView& t, uint64_t key) {
  uint64_t h = splitmix64(key) & t.hmask;
  // Two lanes of ONE wavefront may carry the same new key (k_resolve_multi: several source ranks), and lanes of a
  // wavefront make progress together: a lane waiting for a row id must not keep the winner of its slot from storing it.
  // Every iteration is therefore straight-line: (1) claim, (2) the winners publish, (3) everyone looks — in program
  // order, no early exit between them, so that (2) of a winner always precedes (3) of the lanes that share its
  // wavefront.  The wait is bounded: a slot whose row id never appears sets error bit 3 instead of hanging the device.
  for (uint32_t spins = 0;;) {
    uint64_t k = __hip_atomic_load(&t.ht[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool won = false;
    if (k == kEmptyKey) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&t.ht[h].key),
                                               (unsigned long long)kEmptyKey, (unsigned long long)key);
      won = old == kEmptyKey;
      k = won ? key : (uint64_t)old;
    }
    uint32_t r = kNoRow;
    if (won) {
      r = atomicAdd(t.nrows, 1u);
      if (r >= t.capacity) {
        atomicOr(t.err, 1u);
        r = t.capacity - 1;  // keep memory safe; the host reports DFH_ERR_CAPACITY
      }
      __hip_atomic_store(&t.ht[h].row, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (k == key) {
      if (!won) r = __hip_atomic_load(&t.ht[h].row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (r != kNoRow) return r;
      if (++spins > (1u << 22)) {  // seconds: the winner of this slot is gone
        atomicOr(t.err, 8u);
        return 0u;
      }
      __builtin_amdgcn_s_sleep(1);
      continue;  // the winner of this slot (another wavefront) is between its CAS and its store
    }
    h = (h + 1) & t.hmask;
  }
}

// growth of a table (dfh_api.hip table_grow): every {key, row} of the old index into the new, larger one.  Runs alone on
// the table (every stream drained); distinct keys only race for slots.
__global__ void k_rehash(const HEntry* __restrict__ old_ht, uint64_t old_slots, HEntry* __restrict__ ht, uint64_t hmask) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < old_slots; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = old_ht[i].key;
    if (key == kEmptyKey) continue;
    uint64_t h = splitmix64(key) & hmask;
    for (;;) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&ht[h].key), (unsigned long long)kEmptyKey,
                                               (unsigned long long)key);
      if (old == kEmptyKey) {
        ht[h].row = old_ht[i].row;
        break;
      }
      h = (h + 1) & hmask;
    }
  }
}

// find_or_insert for a whole 256-thread block of DISTINCT keys (k_lookup: one thread per unique key of a minibatch), rows
// handed out per BLOCK: the threads that win a slot are counted (ballot, one LDS atomic per wavefront) and ONE atomicAdd on the
// table's row counter serves them all.  Every row a launch allocates used to cost an atomic on that one address per
// wavefront with a winner, and same-address atomics serialise at ~12 ns each on this part: a first-epoch minibatch (every key
// new: ~147 000 inserts) queued ~1 600 of them, ~20 us of one memory channel's time beside whatever else ran (profiles/r05j_*:
// "probe + 20 us"), now a quarter of that.  Same protocol as find_or_insert towards concurrent launches: a slot's winner publishes
// the row id with a device-scope store, other carriers of the key wait for it — but only AFTER this block's own winners have
// published (nobody waits inside the counted phase: a block that waited there could hold back the winner it waits for).
// All threads of the block call it together (`active`: this thread has a key); sh: 8 shared words.
__device__ __forceinline__ uint32_t find_or_insert_block(const TableView& t, uint64_t key, bool active, uint32_t* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // sh may still be read from the previous call
  if (threadIdx.x == 0) sh[0] = 0u;
  __syncthreads();
  uint64_t h = splitmix64(key) & t.hmask;
  uint32_t r = kNoRow;
  bool won = false, pending = false;
  if (active) {
    for (;;) {
      uint64_t k = __hip_atomic_load(&t.ht[h].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (k == kEmptyKey) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&t.ht[h].key), (unsigned long long)kEmptyKey,
                                                 (unsigned long long)key);
        if (old == kEmptyKey) {
          won = true;
          break;
        }
        k = (uint64_t)old;
      }
      if (k == key) {
        r = __hip_atomic_load(&t.ht[h].row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = r == kNoRow;  // its winner (another launch) is between its CAS and its store: looked at again below
        break;
      }
      h = (h + 1) & t.hmask;
    }
  }
  // rows for this block's winners: rank inside the wavefront from the ballot, the wavefront's offset from one LDS atomic
  const unsigned long long m = __ballot(won);
  uint32_t woff = 0;
  if (lane == 0 && m) woff = atomicAdd(&sh[0], (uint32_t)__popcll(m));
  woff = __shfl(woff, 0, 64);
  __syncthreads();
  if (threadIdx.x == 0) sh[1] = sh[0] ? atomicAdd(t.nrows, sh[0]) : 0u;
  __syncthreads();
  if (won) {
    r = sh[1] + woff + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (r >= t.capacity) {
      atomicOr(t.err, 1u);
      r = t.capacity - 1;  // keep memory safe; the host reports DFH_ERR_CAPACITY
    }
    __hip_atomic_store(&t.ht[h].row, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (pending) {
    for (uint32_t spins = 0;; ++spins) {
      r = __hip_atomic_load(&t.ht[h].row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (r != kNoRow) break;
      if (spins > (1u << 22)) {  // seconds: the winner of this slot is gone
        atomicOr(t.err, 8u);
        r = 0u;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  (void)w;
  return r;
}

// read-only probe (export / tests)
__device__ __forceinline__ uint32_t find_only(const TableView& t, uint64_t key) {
  uint64_t h = splitmix64(key) & t.hmask;
  for (;;) {
    uint64_t k = t.ht[h].key;
    if (k == key) return t.ht[h].row;
    if (k == kEmptyKey) return kNoRow;
    h = (h + 1) & t.hmask;
  }
}

// ---------------------------------------------------------------------------
// The per-key update arithmetic, kept operation-for-operation with
// src/sgd/sgd_updater.cc (no FMA contraction: the reference is built for
// generic x86-64 where a*b+c rounds twice).
// ---------------------------------------------------------------------------
#pragma clang fp contract(off)

// SGDUpdater::UpdateW — FTRL-proximal, src/sgd/sgd_updater.cc:104-120.
// returns the new w; hdr fields are updated in registers by the caller.
__device__ __forceinline__ float ftrl_update_w(float gw, float w, float& sqrt_g, float& z,
                                               const dfh_updater_param& P) {
  float sg = sqrt_g;
  gw += w * P.l2;                       // :108
  float nsg = sqrtf(sg * sg + gw * gw); // :109
  sqrt_g = nsg;
  z -= gw - (nsg - sg) / P.lr * w;      // :111
  float l1 = P.l1;
  if (z <= l1 && z >= -l1) return 0.0f; // :115-116
  float eta = (P.lr_beta + nsg) / P.lr; // :118
  return (z > 0 ? z - l1 : z + l1) / eta; // :119
}

// SGDUpdater::UpdateV — AdaGrad, one coordinate, src/sgd/sgd_updater.cc:129-138, operation for
// operation (IEEE-rounded square root and division, no contraction): given the same gradient the
// new V and accumulator are the reference's, bit for bit.  -DDFH_FAST_ADAGRAD swaps in the hardware
// v_sqrt_f32 / v_rcp_f32 approximations (1 ulp each) for A/B timing of the update kernel only.
__device__ __forceinline__ void adagrad_update_v(float gv, float& v, float& acc, const dfh_updater_param& P) {
  float g = gv + P.V_l2 * v;           // :132
  float cg = acc;                      // :133
#ifdef DFH_FAST_ADAGRAD
  float ncg = __builtin_amdgcn_sqrtf(cg * cg + g * g);
  float eta = P.V_lr * __builtin_amdgcn_rcpf(ncg + P.V_lr_beta);
#else
  float ncg = sqrtf(cg * cg + g * g);  // :134
  float eta = P.V_lr / (ncg + P.V_lr_beta);  // :135
#endif
  acc = ncg;
  v -= eta * g;                        // :136
}

// the reference's REFRAND value: (rand_r(&seed) / (real_t)RAND_MAX - 0.5) * scale  (:144)
__device__ __forceinline__ uint32_t lcg_step(uint32_t x) { return x * 1103515245u + 12345u; }
__device__ __forceinline__ float refrand_value(uint32_t& state, float scale) {
  // glibc rand_r: 3 LCG steps, 11+10+10 bits
  uint32_t next = lcg_step(state);
  int result = (int)((next / 65536u) % 2048u);
  next = lcg_step(next);
  result = (result << 10) ^ (int)((next / 65536u) % 1024u);
  next = lcg_step(next);
  result = (result << 10) ^ (int)((next / 65536u) % 1024u);
  state = next;
  float q = (float)result / 2147483648.0f;  // (real_t)RAND_MAX == 2^31 in float
  return (float)(((double)q - 0.5) * (double)scale);
}

#pragma clang fp contract(fast)

// LCG jump-ahead: state after n steps, by squaring the affine map x -> a x + c
__device__ __forceinline__ uint32_t lcg_jump(uint32_t x, uint64_t n) {
  uint32_t a = 1103515245u, c = 12345u;  // current power of the map
  uint32_t ra = 1u, rc = 0u;             // accumulated map (identity)
  while (n) {
    if (n & 1) {
      ra = ra * a;
      rc = rc * a + c;
    }
    c = c * a + c;
    a = a * a;
    n >>= 1;
  }
  return ra * x + rc;
}

// write V (hash init) + zero accumulators for row r; one thread does the row.  Rare (once per key)
// (once per key).
__device__ __forceinline__ void init_v_hash_row(const TableView& t, uint32_t r, uint64_t key) {
  float* va = t.va + (size_t)r * (2 * t.kp);
  for (int j = 0; j < t.kp; ++j) {
    va[j] = j < t.k ? hash_init_value(key, j, t.p.seed, t.p.V_init_scale) : 0.0f;
    va[t.kp + j] = 0.0f;
  }
}

// the same for one lane's 16 B slice (coordinates d .. d + 3, d < kp) of a row that a whole lane group initialises
__device__ __forceinline__ void init_v_hash_slice(const TableView& t, uint32_t r, uint64_t key, int d) {
  float* va = t.va + (size_t)r * (2 * t.kp);
  float4 nv;
  nv.x = d + 0 < t.k ? hash_init_value(key, d + 0, t.p.seed, t.p.V_init_scale) : 0.0f;
  nv.y = d + 1 < t.k ? hash_init_value(key, d + 1, t.p.seed, t.p.V_init_scale) : 0.0f;
  nv.z = d + 2 < t.k ? hash_init_value(key, d + 2, t.p.seed, t.p.V_init_scale) : 0.0f;
  nv.w = d + 3 < t.k ? hash_init_value(key, d + 3, t.p.seed, t.p.V_init_scale) : 0.0f;
  st4(va + d, nv);
  st4(va + t.kp + d, make_float4(0.f, 0.f, 0.f, 0.f));
}

#ifndef DFH_BWD_SMALL
#define DFH_BWD_SMALL 8
#endif
#ifndef DFH_BWD_MID
#define DFH_BWD_MID 64
#endif
#ifndef DFH_BWD_DEPTH
#define DFH_BWD_DEPTH 2
#endif
#ifndef DFH_FWD_WAVES
#define DFH_FWD_WAVES 5
#endif
#ifndef DFH_BWD_SMALL_DEPTH
#define DFH_BWD_SMALL_DEPTH 2
#endif
#ifndef DFH_BWD_INTERLEAVE
#define DFH_BWD_INTERLEAVE 0
#endif
constexpr uint32_t BWD_SMALL = DFH_BWD_SMALL;
constexpr uint32_t BWD_MID = DFH_BWD_MID;
constexpr int BWD_DEPTH = DFH_BWD_DEPTH;
constexpr int BWD_SMALL_DEPTH = DFH_BWD_SMALL_DEPTH;
#ifndef DFH_BWD_THREADS
#define DFH_BWD_THREADS 512
#endif
constexpr int BWD_THREADS = DFH_BWD_THREADS;    // threads per block of k_backward_all
#ifdef DFH_BWD_TRACE
__device__ unsigned long long g_bwd_trace[3 * 8192];
#endif

// BinClassMetric::AUC without a sort, for minibatch-sized n: the reference's area is the number of
// (positive j, negative i) pairs in which j comes before i in the sorted order (bin_class_metric.h:44-50),
// and "before" can be decided pair by pair: pred_j < pred_i, ties by index (the order a stable sort gives;
// std::sort leaves the order of ties unspecified).  The count is an integer: exact, whatever the order.
// Round 4: only positive columns are compared (a tile of AUC_TILE examples is compacted to its positives in LDS,
// any order) and (image of pred, index) is ONE 64-bit key, so a pair costs a broadcast LDS read, one v_cmp_lt_u64
// and one add-with-carry: ~2e7 pairs of 2 VALU operations for a 10 000-row minibatch with 25 % positives, where
// round 3 spent 1e8 pairs of 7.  auc_pairs_block is one unit of work: 256 rows (this block's threads) against the
// column tiles ct, ct + nct, ...; units are independent, write their count to a slot of their own — no atomics,
// no fence, no ticket (round 3's three same-address atomics + a device-scope fence per block were most of the
// kernel's 27 us) — and may run as blocks of k_auc_pairs or as a role of k_update_fused (the update launch has idle
// VALUs).  auc_finalize_block (one block, any later launch on the stream) adds the slots up and turns
// {area, positives} into AUC * n (:51-53).
constexpr int AUC_TILE = 1024;
constexpr uint32_t AUC_PAIRS_MAX_N = 32768;   // beyond: the radix-sort path (n^2 would pass the cost of sorting)
constexpr uint32_t AUC_COL_SPLIT = 16;        // column tiles are dealt to this many units per row tile (10 000 rows: one tile per unit)
constexpr uint32_t AUC_MAX_ROWTILES = AUC_PAIRS_MAX_N / 256;
constexpr uint32_t AUC_PART_WORDS = AUC_MAX_ROWTILES * (AUC_COL_SPLIT + 1);  // [units] counts | [row tiles] positives

__device__ __forceinline__ uint32_t auc_key(float pred) {
  const uint32_t bits = __float_as_uint(pred + 0.0f);  // -0 and +0 compare equal in the reference: one image
  return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
__host__ __device__ inline uint32_t auc_nct(uint32_t n) {  // units per row tile
  const uint32_t ntile = (n + AUC_TILE - 1) / AUC_TILE;
  return ntile < AUC_COL_SPLIT ? ntile : AUC_COL_SPLIT;
}
__host__ __device__ inline uint32_t auc_units(uint32_t n) { return ((n + 255) / 256) * auc_nct(n); }

// a finished set of units waiting for its finalisation: part (device), n examples; n == 0: nothing pending
struct AucFin {
  const uint32_t* part;
  uint32_t n;
  double* out_slot;
};

// unit `unit` of auc_units(n): 256-thread blocks only.  part[unit] = pairs counted; part[auc_units(n) + rt] = positives of
// row tile rt (written by the tile's first unit)
constexpr size_t AUC_SMEM = (size_t)AUC_TILE * 8 + 16 * 4;  // working memory of auc_pairs_block
__device__ __forceinline__ void auc_pairs_block(const float* __restrict__ pred, const float* __restrict__ label, uint32_t n, uint32_t unit,
                                                uint32_t* __restrict__ part, char* smem /* AUC_SMEM bytes, 16 B aligned */) {
  unsigned long long* colp = reinterpret_cast<unsigned long long*>(smem);  // positives of the current column tile: image of pred_j << 32 | j
  uint32_t* red = reinterpret_cast<uint32_t*>(colp + AUC_TILE);            // [8]
  uint32_t& npos_tile = red[8];
  const uint32_t ntile = (n + AUC_TILE - 1) / AUC_TILE;
  const uint32_t nct = auc_nct(n);
  const uint32_t rt = unit / nct, ct = unit % nct;
  const uint32_t i = rt * 256u + threadIdx.x;
  unsigned long long ki = 0;
  bool neg = false, pos = false;
  if (i < n) {
    ki = ((unsigned long long)auc_key(pred[i]) << 32) | i;
    pos = label[i] > 0;
    neg = !pos;
  }
  uint32_t cnt = 0;
  const int lane = threadIdx.x & 63;
  for (uint32_t tile = ct; tile < ntile; tile += nct) {
    __syncthreads();  // the previous tile has been consumed
    if (threadIdx.x == 0) npos_tile = 0;
    const uint32_t j0 = tile * AUC_TILE;
    // all of the tile's loads first (one round trip), then the compaction: a ballot per wave and ONE LDS atomic per wave
    // and quarter
    float pj[AUC_TILE / 256], lj[AUC_TILE / 256];
#pragma unroll
    for (int r = 0; r < AUC_TILE / 256; ++r) {
      const uint32_t j = min(j0 + r * 256u + threadIdx.x, n - 1);
      pj[r] = pred[j];
      lj[r] = label[j];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < AUC_TILE / 256; ++r) {
      const uint32_t j = j0 + r * 256u + threadIdx.x;
      const bool pj_pos = j < n && lj[r] > 0;
      const unsigned long long m = __ballot(pj_pos);
      uint32_t base = 0;
      if (lane == 0 && m) base = atomicAdd(&npos_tile, (uint32_t)__popcll(m));
      base = __shfl(base, 0, 64);
      if (pj_pos) colp[base + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)auc_key(pj[r]) << 32) | j;
    }
    __syncthreads();
    const uint32_t np = npos_tile;
#pragma unroll 8
    for (uint32_t t = 0; t < np; ++t) cnt += colp[t] < ki ? 1u : 0u;  // (pred_j, j) before (pred_i, i)
  }
  if (!neg) cnt = 0;
  uint32_t npos = pos ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    npos += __shfl_xor(npos, o, 64);
  }
  __syncthreads();
  if (lane == 0) {
    red[threadIdx.x >> 6] = cnt;
    red[4 + (threadIdx.x >> 6)] = npos;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[unit] = red[0] + red[1] + red[2] + red[3];
    if (ct == 0) part[auc_units(n) + rt] = red[4] + red[5] + red[6] + red[7];
  }
}

// one 256-thread block: the units' slots -> AUC * n, added to *out_slot
__device__ __forceinline__ void auc_finalize_block(const AucFin f) {
  __shared__ unsigned long long fsum[2][4];
  const uint32_t nu = auc_units(f.n), nrt = (f.n + 255) / 256;
  unsigned long long area = 0, tp = 0;
  for (uint32_t q = threadIdx.x; q < nu; q += 256u) area += f.part[q];
  for (uint32_t q = threadIdx.x; q < nrt; q += 256u) tp += f.part[nu + q];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    area += __shfl_xor(area, o, 64);
    tp += __shfl_xor(tp, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    fsum[0][threadIdx.x >> 6] = area;
    fsum[1][threadIdx.x >> 6] = tp;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a_all = (double)(fsum[0][0] + fsum[0][1] + fsum[0][2] + fsum[0][3]);
    const double tpd = (double)(fsum[1][0] + fsum[1][1] + fsum[1][2] + fsum[1][3]);
    const double nn = (double)f.n;
    double auc_n;
    if (tpd == 0.0 || tpd == nn) {
      auc_n = 1.0;  // :51 (the reference returns 1, not n)
    } else {
      const double a = a_all / (tpd * (nn - tpd));
      auc_n = (a < 0.5 ? 1.0 - a : a) * nn;
    }
    *f.out_slot += auc_n;
  }
}

__global__ void __launch_bounds__(256) k_auc_pairs(const float* __restrict__ pred, const float* __restrict__ label, uint32_t n,
                                                   uint32_t* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) char smem[AUC_SMEM];
  auc_pairs_block(pred, label, n, blockIdx.x, part, smem);
}
__global__ void __launch_bounds__(256) k_auc_finalize(AucFin f) { auc_finalize_block(f); }

// ---------------------------------------------------------------------------
// k_lookup: one thread per unique key.  urow[u] = row of feaids[u] (inserted as
// a zero row if unseen: sgd_updater.cc:44).  With cnt != NULL it also applies
// Push(kFeaCount): fea_cnt += cnt, maybe InitV (sgd_updater.cc:62-73).
// cnt_from_ptr: counts are the segment lengths of the localized batch.
// ---------------------------------------------------------------------------
// The parts of the keys with more than HOT_SPLIT_MIN occurrences in the minibatch (SegLists::split_ent), listed by the pass that
// visits every unique key before the update and has its segment at hand (col_ptr): one atomic on the list's counter per such key —
// a handful per minibatch at most.  (Round 6's first form listed them in k_loc_emit: the inlined loop took that kernel from 28 to
// 36 registers, and an emit wave of 36 no longer fits in the 32 registers five 96-register waves of k_update_fused leave free on
// a SIMD: the first epoch's steps, where emit runs beside the update, lost 15 % — profiles/r06r_*.)
struct SplitOut {
  SegEnt* ent;        // NULL: no list (the update's hot role takes every key whole)
  uint32_t* n;        // entries so far (zeroed by the Localizer's count pass)
  uint32_t u_base;    // rank of the first key this launch sees (the sharded store looks up its own keys only)
};
__device__ __forceinline__ void lookup_split(const SplitOut& so, uint32_t u, uint32_t beg, uint32_t end) {
  if (!(DFH_HOT_SPLIT_BUILD & 1) || !so.ent || end - beg <= HOT_SPLIT_MIN) return;
  const uint32_t nparts = (end - beg + HOT_SPLIT - 1u) / HOT_SPLIT;
  const uint32_t base = atomicAdd(so.n, nparts);
  for (uint32_t p = 0; p < nparts; ++p)
    so.ent[base + p] = make_uint4(u + so.u_base, beg + p * HOT_SPLIT, min(beg + (p + 1u) * HOT_SPLIT, end), (p << 16) | nparts);
}

template <bool SPLIT = true>
__device__ __forceinline__ void lookup_body(const TableView& t, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ d_n,
                         uint32_t n_static, uint32_t* __restrict__ urow, const float* __restrict__ cnt,
                         const uint32_t* __restrict__ col_ptr, int push_cnt, uint32_t* __restrict__ need_init,
                         int rows_known, uint2* __restrict__ uw, AucFin fin, const uint32_t bid, const uint32_t nblk,
                         const SplitOut so = SplitOut{nullptr, nullptr, 0u}) {
  // a step's lookup also closes the AUC its batch object's previous step left pending (dfh_sgd_step; fin.n == 0: none)
  if (fin.n && bid == 0) auc_finalize_block(fin);
  uint32_t n = d_n ? *d_n : n_static;
  __shared__ uint32_t ins_sh[8];
  // (block-uniform trip count: the insert below is a block-wide call)
  for (uint32_t u0 = bid * blockDim.x; u0 < n; u0 += nblk * blockDim.x) {
    const uint32_t u = u0 + threadIdx.x;
    const bool active = u < n;
    uint64_t key = active ? keys[u] : 0ull;
    // rows_known: urow was filled by an earlier (prep-stream) lookup of the same keys
    // (rows_known: a row word of k_resolve_multi may carry its worker mark in bit 31 — dfh_shard_push_count_resolved)
    uint32_t r;
    if (rows_known) r = active ? (urow[u] & kRowMask) : 0u;
    else if (blockDim.x == 256) r = find_or_insert_block(t, key, active, ins_sh);
    else r = active ? find_or_insert(t, key) : 0u;
    if (!active) continue;
    if (urow && !rows_known) urow[u] = r;
    if (SPLIT && col_ptr) lookup_split(so, u, col_ptr[u], col_ptr[u + 1]);
    float w = 0.f;
    bool hasv = false;
    if (push_cnt) {
      float c = cnt ? cnt[u] : (float)(col_ptr[u + 1] - col_ptr[u]);
      RowHdr& h = t.hdr[r];
      w = h.w;
      // push_cnt == 2: a training step follows whose update kernel rewrites this header anyway.  A key that HAS its V can
      // take its count there (k_update_fused adds it): nothing reads fea_cnt of such a key again — it only gates InitV
      // (sgd_updater.cc:62-73, :122-126) — so the value is the same and this pass stays read-only for it.
      if (push_cnt == 2 && uw && h.has_V != 0) {
        if (need_init) need_init[u] = 0u;
        uw[u] = make_uint2(r | kCountLater | kHasV | ((col_ptr != nullptr && col_ptr[u + 1] - col_ptr[u] == 1u) ? kSingleRow : 0u),
                           __float_as_uint(w));
        continue;
      }
      float fc = h.fea_cnt + c;
      h.fea_cnt = fc;
      hasv = h.has_V != 0;
      bool init = t.k > 0 && !hasv && w != 0 && fc > (float)t.p.V_threshold;
      if (t.p.init_mode == DFH_INIT_HASH) {
        if (init) {
          init_v_hash_row(t, r, key);
          h.has_V = 1;
        }
      } else if (need_init) {
        need_init[u] = init ? 1u : 0u;   // (REFRAND: the row is written by k_refrand_init right after this launch, before the forward)
      }
      hasv = hasv || init;   // InitV at count-push time (sgd_updater.cc:69-72): this step's Pull sees the new V
    } else if (uw) {
      const uint2 wh = *reinterpret_cast<const uint2*>(&t.hdr[r]);  // {w, has_V}: one 8 B load
      w = __uint_as_float(wh.x);
      hasv = wh.y != 0u;
    }
    // {row, w} per unique key, batch-local and L2-resident: the forward then touches nothing of a
    // row but its V lines (the weight is read here once per KEY instead of once per nonzero)
    // bit 30 of the row word: the key occurs once in this minibatch (k_update_fused updates such keys example by
    // example, without the key-ordered view)
    if (uw) {
      const bool single = col_ptr != nullptr && col_ptr[u + 1] - col_ptr[u] == 1u;
      uw[u] = make_uint2(r | (single ? kSingleRow : 0u) | (hasv ? kHasV : 0u), __float_as_uint(w));
    }
  }
}

// (two kernels: the probe that runs on a preparation stream BESIDE the update must stay small in registers — its waves take what
// five 96-register update waves leave free on a SIMD — so only the step's own pass carries the split-list code)
__global__ void k_lookup(TableView t, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ d_n,
                         uint32_t n_static, uint32_t* __restrict__ urow, const float* __restrict__ cnt,
                         const uint32_t* __restrict__ col_ptr, int push_cnt, uint32_t* __restrict__ need_init,
                         int rows_known, uint2* __restrict__ uw, AucFin fin) {
  lookup_body<false>(t, keys, d_n, n_static, urow, cnt, col_ptr, push_cnt, need_init, rows_known, uw, fin, blockIdx.x, gridDim.x);
}
__global__ void k_lookup_step(TableView t, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ d_n,
                              uint32_t n_static, uint32_t* __restrict__ urow, const float* __restrict__ cnt,
                              const uint32_t* __restrict__ col_ptr, int push_cnt, uint32_t* __restrict__ need_init,
                              int rows_known, uint2* __restrict__ uw, AucFin fin, SplitOut so) {
  lookup_body<true>(t, keys, d_n, n_static, urow, cnt, col_ptr, push_cnt, need_init, rows_known, uw, fin, blockIdx.x, gridDim.x, so);
}

// sharded store: {u | kRemoteRow, w} for the keys OTHER ranks own, from the rows they sent (row u of
// the pulled-rows buffer belongs to key u; the slots of this rank's own keys [lo, hi) are unused)
struct UwRemote {
  const float* rows;
  size_t stride;
  const uint32_t* d_U;
  uint32_t lo, hi;
  uint2* uw;                 // the whole minibatch's array (the lookup's `uw` starts at the rank's own keys)
  const uint32_t* col_ptr;   // likewise
  SplitOut so;               // the others' keys can be very hot too (u_base 0: ranks are the minibatch's)
};
__device__ __forceinline__ void uw_remote_body(const UwRemote& m) {
  const uint32_t U = *m.d_U;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
    if (u - m.lo < m.hi - m.lo) continue;
    // (bit 30: the key occurs once in the minibatch — the singles role of the mixed update launch takes it)
    const uint32_t single = (m.col_ptr && m.col_ptr[u + 1] - m.col_ptr[u] == 1u) ? kSingleRow : 0u;
    if (m.col_ptr) lookup_split(m.so, u, m.col_ptr[u], m.col_ptr[u + 1]);
    const float2 wh = *reinterpret_cast<const float2*>(m.rows + (size_t)u * m.stride);   // [w, has_V (0 / 1 as float), ...]
    m.uw[u] = make_uint2(u | kRemoteRow | single | (wh.y != 0.f ? kHasV : 0u), __float_as_uint(wh.x));
  }
}
__global__ void k_uw_remote(const float* __restrict__ rows, size_t stride, const uint32_t* __restrict__ d_U, uint32_t lo,
                            uint32_t hi, uint2* __restrict__ uw, const uint32_t* __restrict__ col_ptr, SplitOut so) {
  uw_remote_body(UwRemote{rows, stride, d_U, lo, hi, uw, col_ptr, so});
}
// the own keys' lookup and the others' row words in ONE launch (overlapped exchange: the rows of the other owners arrived
// during the previous step; one launch boundary less on the main stream of the sharded step)
__global__ void k_lookup_uw_remote(TableView t, const uint64_t* __restrict__ keys, uint32_t n_static, uint32_t* __restrict__ urow,
                                   const float* __restrict__ cnt, const uint32_t* __restrict__ col_ptr, int push_cnt,
                                   uint2* __restrict__ uw, AucFin fin, UwRemote m, SplitOut so) {
  lookup_body(t, keys, nullptr, n_static, urow, cnt, col_ptr, push_cnt, nullptr, 0, uw, fin, blockIdx.x, gridDim.x, so);
  uw_remote_body(m);
}

// ---------------------------------------------------------------------------
// REFRAND lazy init (parity mode): rows flagged in need[] are initialised in
// ascending key order from the mutated rand_r chain, exactly as the serial
// loop of SGDUpdater::Update would (sgd_updater.cc:62-73, :86-95, :140-147).
// k_refrand_scan: single block, exclusive scan of need[0..n) -> rank[], total.
// ---------------------------------------------------------------------------
__global__ void k_refrand_scan(const uint32_t* __restrict__ need, const uint32_t* __restrict__ d_n, uint32_t n_static,
                               uint32_t* __restrict__ rank, uint32_t* __restrict__ total) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  uint32_t n = d_n ? *d_n : n_static;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += blockDim.x) {
    uint32_t i = base + threadIdx.x;
    uint32_t v = i < n ? need[i] : 0;
    // inclusive scan inside the wave
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t y = __shfl_up(s, o, 64);
      if (lane_id() >= o) s += y;
    }
    int w = threadIdx.x >> 6;
    if (lane_id() == 63) wsum[w] = s;
    __syncthreads();
    uint32_t woff = 0;
    for (int j = 0; j < w; ++j) woff += wsum[j];
    uint32_t carry = carry_s;
    if (i < n) rank[i] = carry + woff + s - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry_s = carry + woff + s;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry_s;
}

__global__ void k_refrand_init(TableView t, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ d_n,
                               uint32_t n_static, const uint32_t* __restrict__ urow,
                               const uint32_t* __restrict__ need, const uint32_t* __restrict__ rank) {
  uint32_t n = d_n ? *d_n : n_static;
  uint32_t state0 = *t.rng_state;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
    if (!need[u]) continue;
    uint32_t r = urow ? urow[u] : find_only(t, keys[u]);
    uint32_t st = lcg_jump(state0, (uint64_t)3 * (uint64_t)t.k * (uint64_t)rank[u]);
    float* va = t.va + (size_t)r * (2 * t.kp);
    for (int j = 0; j < t.kp; ++j) {
      va[j] = j < t.k ? refrand_value(st, t.p.V_init_scale) : 0.0f;
      va[t.kp + j] = 0.0f;
    }
    t.hdr[r].has_V = 1;
  }
}

__global__ void k_refrand_advance(TableView t, const uint32_t* __restrict__ total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *t.rng_state = lcg_jump(*t.rng_state, (uint64_t)3 * (uint64_t)t.k * (uint64_t)(*total));
  }
}

// ---------------------------------------------------------------------------
// k_forward<L>: FMLoss::Predict (src/loss/fm_loss.h:67-119) fused with the
// gather (Store::Pull, sgd_updater.cc:32-56), the logistic slope
// (fm_loss.h:157-161) and Loss::Evaluate (include/difacto/loss.h:57-66).
//
// One wavefront per example.  A V row is read by L = kp/4 lanes as one float4
// each (16 B/lane: k=64 -> 16 lanes x 16 B = two full 128 B lines), so a wave
// has G = 64/L rows in flight per load instruction.  The wave first stages up
// to 64 nnz of the example (one coalesced index load, one gather of
// {row, w, has_V}), then the groups walk the staged nnz with cross-lane
// broadcasts, issuing FWD_DEPTH independent row loads before consuming any
// (the kernel is latency-bound on the random gather: depth, not width, is
// what fills the memory pipe).
//
//   pred_i = sum_j x_ij w_j + 1/2 sum_d [ (sum_j x_ij V_jd)^2 - sum_j x_ij^2 V_jd^2 ]
// ---------------------------------------------------------------------------
// progress partials: prog[kind * PROG_SLOTS + blockIdx.x], summed on the host
constexpr int PROG_SLOTS = 16384;
constexpr int PROG_LOSS = 0, PROG_PENALTY = 1, PROG_AUC = 2;  // [PROG_AUC * PROG_SLOTS] is a single accumulator

// MIXED (sharded store): a key's V row lives either in this rank's table (the keys this rank owns) or in
// the buffer of rows pulled from the other owners; bit 31 of the row word k_lookup / k_uw_remote leave
// in uw[] says which, and the address is chosen per row.
struct MixSrc {
  const float* vbase2;   // V of the pulled rows (packed layout)
  size_t vstride2;
};

// block `bid` of `nblk` 256-thread blocks (the whole launch of k_forward; a block range of a launch that also carries
// riders: dfh_riders.hip)
template <int L, int FWD_DEPTH, bool MIXED>
__device__ __forceinline__ void forward_body(const BatchView& b, const RowSrc& src, const int k, const int kp, const MixSrc& mix,
                                             const uint32_t bid, const uint32_t nblk, double* blk /* [4] shared */) {
  constexpr int G = 64 / L;
  const int lane = lane_id();
  const int grp = lane / L;
  const int sub = lane % L;
  const bool sub_ok = sub * 4 < kp;  // L may exceed kp/4 when kp/4 is not a power of two
  const int wave = (bid * 256u + threadIdx.x) >> 6;
  const int nwaves = (nblk * 256u) >> 6;
  double loss_acc = 0.0;

  for (uint32_t i = wave; i < b.nrows; i += nwaves) {
    const uint32_t beg = b.offset[i], end = b.offset[i + 1];
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xxvv = make_float4(0.f, 0.f, 0.f, 0.f);
    float wsum = 0.f;
    for (uint32_t base = beg; base < end; base += 64) {
      const uint32_t j = base + lane;
      const bool valid = j < end;
      uint32_t r = 0;
      float x = 0.f;
      uint32_t hv = 0;
      if (valid) {
        x = b.value ? b.value[j] : 1.0f;
        if (b.uw) {
          // {row, w} of the key from the batch-local table k_lookup left in L2: no header access;
          // a row without V holds zeros, so its flag is not needed either
          const uint2 e = b.uw[b.index[j]];
          r = e.x & (kRowMask | kRemoteRow);
          hv = e.x & kHasV;   // SpMM::Times skips a key without V (spmm.h:108); an allocated row is used whatever it holds
          wsum = fma_skip0(__uint_as_float(e.y), x, wsum);  // spmv.h:125
        } else {
          const uint32_t u = b.index[j];
          r = src.urow ? src.urow[u] : u;
          const float* wp = src.wbase + (size_t)r * src.wstride;
          // {w, has_V} are adjacent: one 8 B load
          float2 wf = *reinterpret_cast<const float2*>(wp);
          hv = __float_as_uint(wf.y);
          wsum = fma_skip0(wf.x, x, wsum);  // spmv.h:125
        }
      }
      const int cnt = min(64u, end - base);
      if (k > 0) {
        for (int t0 = 0; t0 < cnt; t0 += FWD_DEPTH * G) {
          float4 v[FWD_DEPTH];
          float xs[FWD_DEPTH];
#pragma unroll
          for (int q = 0; q < FWD_DEPTH; ++q) {
            const int t = t0 + q * G + grp;
            const uint32_t rr = __shfl(r, t & 63, 64);
            const float xx = __shfl(x, t & 63, 64);
            const uint32_t hh = __shfl(hv, t & 63, 64);
            // the row load does not wait for has_V: a row without V holds zeros
            const bool ok = t < cnt && sub_ok;
            xs[q] = (ok && hh != 0) ? xx : 0.f;
            if (MIXED) {
              const bool rem = (rr & kRemoteRow) != 0u;
              const float* vb = rem ? mix.vbase2 : src.vbase;
              const size_t vs = rem ? mix.vstride2 : src.vstride;
              v[q] = ok ? ld4(vb + (size_t)(rr & ~kRemoteRow) * vs + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
              v[q] = ok ? ld4(src.vbase + (size_t)rr * src.vstride + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
#pragma unroll
          for (int q = 0; q < FWD_DEPTH; ++q) {
            // a key WITHOUT V is skipped by SpMM (V_pos = -1, spmm.h:108): its x is 0 here (xs[q], from the key's has-V flag),
            // so that an Inf / NaN feature value on such a key meets the speculatively loaded row of zeros as 0 * 0, not as NaN
            const float xx = xs[q];
            xv.x += v[q].x * xx; xv.y += v[q].y * xx; xv.z += v[q].z * xx; xv.w += v[q].w * xx;
            const float x2 = xx * xx;
            xxvv.x += (v[q].x * v[q].x) * x2; xxvv.y += (v[q].y * v[q].y) * x2;
            xxvv.z += (v[q].z * v[q].z) * x2; xxvv.w += (v[q].w * v[q].w) * x2;
          }
        }
      }
    }
    wsum = wave_sum(wsum);
    float pred = wsum;
    if (k > 0) {
      // combine the G groups' partial rows: every group now holds the full XV / XXVV slice
      xv.x = cross_group_sum<L>(xv.x); xv.y = cross_group_sum<L>(xv.y);
      xv.z = cross_group_sum<L>(xv.z); xv.w = cross_group_sum<L>(xv.w);
      xxvv.x = cross_group_sum<L>(xxvv.x); xxvv.y = cross_group_sum<L>(xxvv.y);
      xxvv.z = cross_group_sum<L>(xxvv.z); xxvv.w = cross_group_sum<L>(xxvv.w);
      // s = sum_d (XV_d^2 - XXVV_d), per-dimension difference first as fm_loss.h:113
      float s = (xv.x * xv.x - xxvv.x) + (xv.y * xv.y - xxvv.y) + (xv.z * xv.z - xxvv.z) + (xv.w * xv.w - xxvv.w);
      s = in_group_sum<L>(s);
      pred += 0.5f * s;
      pred = pred > 20.f ? 20.f : (pred < -20.f ? -20.f : pred);  // fm_loss.h:118 (only when V_dim > 0, :77)
      if (grp == 0 && sub_ok && b.xv) st4(b.xv + (size_t)i * kp + sub * 4, xv);
    }
    if (lane == 0) {
      const float y = b.label[i] > 0 ? 1.0f : -1.0f;
      b.pred[i] = pred;
      b.slope[i] = -y / (1.0f + expf(y * pred));            // fm_loss.h:160
      // log(1 + exp(-y pred)) (loss.h:63) in its stable fp32 form; the batch sum is kept in double
      const float m = -y * pred;
      loss_acc += (double)(fmaxf(m, 0.f) + log1pf(expf(-fabsf(m))));
    }
  }
  // the batch's logloss: one private slot per block (same-address atomics
  // serialise at ~12 ns each on this chip: 2k blocks would cost 25 us)
  if (lane == 0) blk[threadIdx.x >> 6] = loss_acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&b.prog[PROG_LOSS * PROG_SLOTS + (bid % PROG_SLOTS)], blk[0] + blk[1] + blk[2] + blk[3]);
}

template <int L, int FWD_DEPTH, bool MIXED>
__global__ void __launch_bounds__(256, DFH_FWD_WAVES) k_forward(BatchView b, RowSrc src, int k, int kp, MixSrc mix) {
  __shared__ double blk[4];
  forward_body<L, FWD_DEPTH, MIXED>(b, src, k, kp, mix, blockIdx.x, gridDim.x, blk);
}

// ---------------------------------------------------------------------------
// Backward / update: FMLoss::CalcGrad (src/loss/fm_loss.h:148-199) as a
// segmented sum over the key-ordered occurrence list the Localizer's sort
// leaves behind (runs of equal key, src/data/localizer.cc:28):
//   gw_u   = sum_occ x p_i
//   gV_u,d = sum_occ x (p_i XV_i,d) - V_u,d sum_occ x^2 p_i
// FUSED: apply SGDUpdater::Update for the key in place (FTRL on w, AdaGrad on
// V, lazy InitV) — no gradient ever reaches HBM.
// !FUSED: write [gw, has_V, 0, 0 | gV] rows of `gstride` floats (exchange layout).
//
// Segment lengths are Zipf-distributed (1 .. ~B/8), so ONE launch (k_backward_all)
// runs three roles, chosen by block range (long chains first so they start early):
//   hot    one key of the hot list (cnt > BWD_MID) per block and iteration: the
//          block's waves split the segment, partials combined in LDS in wave order
//   mid    one key of the mid list (BWD_SMALL < cnt <= BWD_MID) per wave
//   short  one L-lane group per key (G keys per wave), occurrences summed serially
//          in row order — the reference's own order
// The two lists are compacted once per minibatch by k_seg_lists.  Every key is
// handled by exactly one role; nothing is communicated between blocks.  Each role
// issues all of a key's independent loads (header, V row, AdaGrad row, occurrence
// list) before consuming any; V rows are read speculatively (a row without V holds
// zeros, in the table and in packed rows alike).
// ---------------------------------------------------------------------------

// a launch of k_backward_all / k_penalty may be restricted to the keys whose rank u lies inside
// (inv = 0) or outside (inv = 1) [lo, hi): the sharded store runs the keys this rank owns through the
// fused in-place update and the others through the gradient-row form.  Default: every key.
struct KeyRange {
  uint32_t lo, hi, inv;  // inv bit 0: the keys OUTSIDE [lo, hi); bit 1: the gradient-row form also adds up the penalty of the
                         // rows it reads (EvaluatePenalty over the weights pulled from other owners, sgd_learner.cc:249-273)
};
__device__ __forceinline__ bool key_in(const KeyRange& kr, uint32_t u) { return ((u - kr.lo < kr.hi - kr.lo) ? 1u : 0u) != (kr.inv & 1u); }
__device__ __forceinline__ bool key_pen(const KeyRange& kr) { return (kr.inv & 2u) != 0; }

struct KeySums {
  float gw;    // sum p x
  float xxp;   // sum p x^2
  float4 gv;   // sum (XV p) x, this lane's 4 dims
};

struct KeyRow {   // what a role prefetches for one key
  uint32_t r;
  float w_old;
  bool has_v;
  float sqrt_g, z, fea_cnt;
  float4 v, acc;  // this lane's slices
};

template <bool FUSED>
__device__ __forceinline__ KeyRow load_key_row(const RowSrc& src, const TableView& t, uint32_t u, int sub, bool sub_ok,
                                               int k, int kp, bool resolved = true) {
  KeyRow kr;
  // resolved = false: the key is outside the launch's range and may have no row id (speculative load: row 0)
  kr.r = src.urow ? (resolved ? src.urow[u] : 0u) : u;
  const float* wp = src.wbase + (size_t)kr.r * src.wstride;
  kr.v = make_float4(0.f, 0.f, 0.f, 0.f);
  kr.acc = kr.v;
  kr.sqrt_g = kr.z = kr.fea_cnt = 0.f;
  // independent loads, issued back to back
  float4 h0 = ld4(wp);  // table: {w, has_V, sqrt_g, z}; packed: {w, has_V, 0, 0}
  if (k > 0 && sub_ok) {
    kr.v = FUSED ? ld4_nt(src.vbase + (size_t)kr.r * src.vstride + sub * 4) : ld4(src.vbase + (size_t)kr.r * src.vstride + sub * 4);
    if (FUSED) kr.acc = ld4_nt(t.va + (size_t)kr.r * (2 * kp) + kp + sub * 4);
  }
  if (FUSED) kr.fea_cnt = t.hdr[kr.r].fea_cnt;
  kr.w_old = h0.x;
  kr.has_v = k > 0 && __float_as_uint(h0.y) != 0;
  kr.sqrt_g = h0.z;
  kr.z = h0.w;
  if (!kr.has_v) kr.v = make_float4(0.f, 0.f, 0.f, 0.f);
  return kr;
}

// Everything after the sums, executed by the L lanes of ONE group (the caller
// masks the others).
template <int L, bool FUSED>
__device__ __forceinline__ void finish_key(const BatchView& b, const TableView& t, uint32_t u, const KeyRow& kr, int sub,
                                           bool sub_ok, KeySums s, float* __restrict__ grads, size_t gstride, int k,
                                           int kp, uint32_t* __restrict__ need_init, double& pen_acc, bool pen_rows = false) {
  float4 gv = s.gv;
  const float4 v = kr.v;
  if (kr.has_v) {
    // grad_V = X'(diag(p) XV) - diag(XXp) V   (fm_loss.h:181-198)
    gv.x -= v.x * s.xxp; gv.y -= v.y * s.xxp; gv.z -= v.z * s.xxp; gv.w -= v.w * s.xxp;
  }
  if (!FUSED) {
    float* g = grads + (size_t)u * gstride;
    if (sub == 0) st4(g, make_float4(s.gw, kr.has_v ? 1.0f : 0.0f, 0.f, 0.f));
    if (sub_ok && k > 0) st4(g + 4 + sub * 4, kr.has_v ? gv : make_float4(0.f, 0.f, 0.f, 0.f));
    if (pen_rows) {  // the rows came from another owner: their penalty is nobody else's to count
      if (kr.has_v && sub_ok) pen_acc += (double)(0.5f * t.p.V_l2 * (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
      if (sub == 0) pen_acc += (double)(t.p.l1 * fabsf(kr.w_old) + 0.5f * t.p.l2 * kr.w_old * kr.w_old);
    }
    return;
  }
  // penalty of the PULLED weights (SGDLearner::EvaluatePenalty, sgd_learner.cc:249-273)
  // per-lane partials in fp32 (a handful of terms each), widened when the block flushes
  if (kr.has_v && sub_ok) pen_acc += (double)(0.5f * t.p.V_l2 * (v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
  RowHdr* hp = t.hdr + kr.r;
  if (sub == 0) {
    const float w_old = kr.w_old;
    pen_acc += (double)(t.p.l1 * fabsf(w_old) + 0.5f * t.p.l2 * w_old * w_old);
    // SGDUpdater::Update(kGradient) for this key (sgd_updater.cc:86-95)
    float sqrt_g = kr.sqrt_g, z = kr.z;
    const float w_new = ftrl_update_w(s.gw, w_old, sqrt_g, z, t.p);
    // one 16 B store of {w, has_V, sqrt_g, z}
    uint32_t hv = kr.has_v ? 1u : 0u;
    // lazy InitV when w leaves zero (sgd_updater.cc:122-126)
    if (w_old == 0 && w_new != 0 && k > 0 && !kr.has_v && kr.fea_cnt > (float)t.p.V_threshold) {
      if (t.p.init_mode == DFH_INIT_HASH) {
        init_v_hash_row(t, kr.r, b.feaids[u]);
        hv = 1u;
      } else {
        need_init[u] = 1;
      }
    }
    st4(reinterpret_cast<float*>(hp), make_float4(w_new, __uint_as_float(hv), sqrt_g, z));
  }
  if (kr.has_v && sub_ok) {
    float* va = t.va + (size_t)kr.r * (2 * kp);
    float4 acc = kr.acc;
    float4 nv = v;
    adagrad_update_v(gv.x, nv.x, acc.x, t.p);
    adagrad_update_v(gv.y, nv.y, acc.y, t.p);
    adagrad_update_v(gv.z, nv.z, acc.z, t.p);
    adagrad_update_v(gv.w, nv.w, acc.w, t.p);
    // padded coordinates (>= k) stay exactly zero
    const int d0 = sub * 4;
    if (d0 + 0 >= k) { nv.x = 0.f; acc.x = 0.f; }
    if (d0 + 1 >= k) { nv.y = 0.f; acc.y = 0.f; }
    if (d0 + 2 >= k) { nv.z = 0.f; acc.z = 0.f; }
    if (d0 + 3 >= k) { nv.w = 0.f; acc.w = 0.f; }
    st4_nt(va + sub * 4, nv);
    st4_nt(va + kp + sub * 4, acc);
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of the per-lane penalty partials into one of PROG_SLOTS slots
// (distinct addresses: no same-address atomic serialisation)
__device__ __forceinline__ void flush_penalty(const BatchView& b, double pen_acc) {
  __shared__ double pen_blk[16];
  pen_acc = wave_sum_d(pen_acc);
  if (lane_id() == 0) pen_blk[threadIdx.x >> 6] = pen_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += pen_blk[i];
    if (t != 0.0) atomicAdd(&b.prog[PROG_PENALTY * PROG_SLOTS + (blockIdx.x % PROG_SLOTS)], t);
  }
}

// partial sums of occurrences [beg,end) taken by this wave: tiles of 64
// starting at tile w0, stride wstep tiles; result: gw/xxp wave-reduced, gv
// reduced across groups (every group holds the sums of its lane slice)
template <int L>
__device__ __forceinline__ KeySums wave_segment_sums(const BatchView& b, uint32_t beg, uint32_t end, uint32_t w0,
                                                     uint32_t wstep, bool want_v, bool sub_ok, int grp, int sub, int kp) {
  constexpr int G = 64 / L;
  const int lane = lane_id();
  KeySums s;
  s.gw = 0.f; s.xxp = 0.f; s.gv = make_float4(0.f, 0.f, 0.f, 0.f);
  for (uint32_t base = beg + w0 * 64; base < end; base += wstep * 64) {
    const uint32_t j = base + lane;
    const bool valid = j < end;
    uint32_t row = 0;
    float x = 0.f, p = 0.f;
    if (valid) {
      row = b.s_row[j];
      x = b.s_val ? b.s_val[j] : 1.0f;
      p = b.slope[row];
      s.gw = fma_skip0(p, x, s.gw);          // spmv.h:155-163
      s.xxp = fma_skip0(p, x * x, s.xxp);    // fm_loss.h:171-178 with XX = value^2
    }
    if (want_v) {
      const int cnt = min(64u, end - base);
      for (int t0 = 0; t0 < cnt; t0 += BWD_DEPTH * G) {
        float4 a[BWD_DEPTH];
        float xs[BWD_DEPTH], ps[BWD_DEPTH];
#pragma unroll
        for (int q = 0; q < BWD_DEPTH; ++q) {
          const int tt = t0 + q * G + grp;
          const uint32_t rowi = __shfl(row, tt & 63, 64);
          const float xx = __shfl(x, tt & 63, 64);
          const float pp = __shfl(p, tt & 63, 64);
          const bool ok = tt < cnt && sub_ok;
          xs[q] = ok ? xx : 0.f;
          ps[q] = pp;
          a[q] = ok ? ld4(b.xv + (size_t)rowi * kp + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < BWD_DEPTH; ++q) {
          const float pp = ps[q], xx = xs[q];
          s.gv.x += (a[q].x * pp) * xx; s.gv.y += (a[q].y * pp) * xx;
          s.gv.z += (a[q].z * pp) * xx; s.gv.w += (a[q].w * pp) * xx;
        }
      }
    }
  }
  s.gw = wave_sum(s.gw);
  s.xxp = wave_sum(s.xxp);
  if (want_v) {
    s.gv.x = cross_group_sum<L>(s.gv.x); s.gv.y = cross_group_sum<L>(s.gv.y);
    s.gv.z = cross_group_sum<L>(s.gv.z); s.gv.w = cross_group_sum<L>(s.gv.w);
  }
  return s;
}

// ---- long segments (cnt > BWD_SMALL), taken from the two lists k_seg_lists compacts once per
// minibatch: every wave of the mid blocks takes mid keys (whole wave per key), every hot block
// takes hot keys one at a time (its NW waves split the segment, partials combined through LDS in
// wave order).  The lists are in arbitrary order; a key's result does not depend on it.
template <int L, bool FUSED, int NW>
__device__ __forceinline__ void mid_role(const BatchView& b, const RowSrc& src, const TableView& t, float* __restrict__ grads,
                                         size_t gstride, int k, int kp, uint32_t* __restrict__ need_init, uint32_t wave,
                                         uint32_t nwaves, uint32_t nb, const KeyRange& rg, double& pen_acc) {
  const int lane = lane_id();
  const int grp = lane / L;
  const int sub = lane % L;
  const bool sub_ok = sub * 4 < kp;
  // waves are dealt to the list buckets: G waves per bucket when there are more waves than buckets
  if (nb == 0) return;
  const uint32_t G = max(1u, nwaves / nb), ngroups = nwaves / G;
  const uint32_t grp_w = wave / G, sub_w = wave % G;
  if (grp_w >= ngroups) return;
  for (uint32_t lb = grp_w; lb < nb; lb += ngroups) {
    const uint2 co = b.seg.mid[lb];
    const uint32_t nm = co.x;
    if (nm == 0) continue;
    const SegEnt* __restrict__ ent = b.seg.mid_ent + co.y;
    for (uint32_t q = sub_w; q < nm; q += G) {
      const SegEnt e = ent[q];
      const uint32_t u = e.x;
      if (!key_in(rg, u)) continue;  // uniform per wave
      const uint32_t beg = e.y, end = e.z;
      KeySums s = wave_segment_sums<L>(b, beg, end, 0, 1, k > 0, sub_ok, grp, sub, kp);
      // the key's row is fetched after the sums: one more round trip for a segment of 9+ occurrences,
      // 14 registers fewer alive through the loop (the kernel runs at 64 registers, 8 waves per SIMD)
      if (grp == 0) {
        const KeyRow kr = load_key_row<FUSED>(src, t, u, sub, sub_ok, k, kp);
        finish_key<L, FUSED>(b, t, u, kr, sub, sub_ok, s, grads, gstride, k, kp, need_init, pen_acc, key_pen(rg));
      }
    }
  }
}

template <int L, bool FUSED, int NW>
__device__ __forceinline__ void hot_role(const BatchView& b, const RowSrc& src, const TableView& t, float* __restrict__ grads,
                                         size_t gstride, int k, int kp, uint32_t* __restrict__ need_init, uint32_t blk,
                                         uint32_t nblk, uint32_t nb, const KeyRange& rg, double& pen_acc) {
  __shared__ float part[NW][2 + 256];  // per wave: gw, xxp, gv[kp <= 256]
  const int lane = lane_id();
  const int grp = lane / L;
  const int sub = lane % L;
  const bool sub_ok = sub * 4 < kp;
  const int w = threadIdx.x >> 6;
  // blocks are dealt to the list buckets like the waves of the mid role (uniform per block: the
  // barriers below are reached by all of its threads)
  if (nb == 0) return;
  const uint32_t G = max(1u, nblk / nb), ngroups = nblk / G;
  const uint32_t grp_b = blk / G, sub_b = blk % G;
  if (grp_b >= ngroups) return;
  for (uint32_t lb = grp_b; lb < nb; lb += ngroups) {
  const uint2 co = b.seg.hot[lb];
  const uint32_t nh = co.x;
  if (nh == 0) continue;
  const SegEnt* __restrict__ ent = b.seg.hot_ent + co.y;
  for (uint32_t q = sub_b; q < nh; q += G) {
    const SegEnt e = ent[q];
    const uint32_t u = e.x;
    if (!key_in(rg, u)) continue;  // uniform per block
    const uint32_t beg = e.y, end = e.z;
    KeySums s = wave_segment_sums<L>(b, beg, end, (uint32_t)w, NW, k > 0, sub_ok, grp, sub, kp);
    __syncthreads();  // previous key's partials consumed
    if (grp == 0) {
      if (sub == 0) { part[w][0] = s.gw; part[w][1] = s.xxp; }
      if (sub_ok) {
        part[w][2 + sub * 4 + 0] = s.gv.x; part[w][2 + sub * 4 + 1] = s.gv.y;
        part[w][2 + sub * 4 + 2] = s.gv.z; part[w][2 + sub * 4 + 3] = s.gv.w;
      }
    }
    __syncthreads();
    if (w == 0 && grp == 0) {
      KeySums tot;
      tot.gw = 0.f; tot.xxp = 0.f; tot.gv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        tot.gw += part[i][0];
        tot.xxp += part[i][1];
        if (sub_ok) {
          tot.gv.x += part[i][2 + sub * 4 + 0]; tot.gv.y += part[i][2 + sub * 4 + 1];
          tot.gv.z += part[i][2 + sub * 4 + 2]; tot.gv.w += part[i][2 + sub * 4 + 3];
        }
      }
      const KeyRow kr = load_key_row<FUSED>(src, t, u, sub, sub_ok, k, kp);  // after the sums, as in the mid role
      finish_key<L, FUSED>(b, t, u, kr, sub, sub_ok, tot, grads, gstride, k, kp, need_init, pen_acc, key_pen(rg));
    }
  }
  }
}

// ---- small segments (cnt <= BWD_SMALL): one L-lane group per key, G keys per wave,
// occurrences summed serially in row order (the reference's own order).  Its own
// launch so that it gets its own register budget (8 waves/SIMD): the kernel is a
// chain of dependent random accesses and only occupancy hides them.  Runs
// concurrently with k_backward_big (other stream): the two touch disjoint keys.
template <int L, bool FUSED>
__device__ __forceinline__ void small_role(const BatchView& b, const RowSrc& src, const TableView& t, float* __restrict__ grads,
                                           size_t gstride, int k, int kp, uint32_t* __restrict__ need_init, uint32_t wave,
                                           uint32_t nwaves, const KeyRange& rg, double& pen_acc) {
  constexpr int G = 64 / L;
  const int lane = lane_id();
  const int grp = lane / L;
  const int sub = lane % L;
  const bool sub_ok = sub * 4 < kp;
  const uint32_t U = *b.d_U;
  {
    for (uint32_t u0 = wave * G; u0 < U; u0 += nwaves * G) {
      // no branch before the loads: the segment bounds, the row id and then the row itself are
      // requested for every lane group (keys left to the mid / hot roles waste one speculative read)
      const uint32_t u = min(u0 + grp, U - 1);
      const uint32_t beg = b.col_ptr[u], end_all = b.col_ptr[u + 1];
      const bool in_rg = key_in(rg, u);
      const KeyRow kr = load_key_row<FUSED>(src, t, u, sub, sub_ok, k, kp, in_rg);
      const bool mine = (u0 + grp) < U && end_all - beg <= BWD_SMALL && in_rg;
      const uint32_t end = mine ? end_all : beg;
      KeySums s;
      s.gw = 0.f; s.xxp = 0.f; s.gv = make_float4(0.f, 0.f, 0.f, 0.f);
      for (uint32_t j0 = beg; j0 < end; j0 += BWD_SMALL_DEPTH) {
        float4 a[BWD_SMALL_DEPTH];
        float xs[BWD_SMALL_DEPTH], ps[BWD_SMALL_DEPTH];
#pragma unroll
        for (int q = 0; q < BWD_SMALL_DEPTH; ++q) {
          const uint32_t j = j0 + q;
          const bool ok = j < end;
          const uint32_t row = ok ? b.s_row[j] : 0;
          xs[q] = ok ? (b.s_val ? b.s_val[j] : 1.0f) : 0.f;
          ps[q] = ok ? b.slope[row] : 0.f;
          a[q] = (ok && k > 0 && sub_ok) ? ld4(b.xv + (size_t)row * kp + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < BWD_SMALL_DEPTH; ++q) {  // ascending rows: the reference's order (spmm.h:137-156)
          const float pp = ps[q], xx = xs[q];
          s.gw = fma_skip0(pp, xx, s.gw);
          s.xxp = fma_skip0(pp, xx * xx, s.xxp);
          s.gv.x += (a[q].x * pp) * xx; s.gv.y += (a[q].y * pp) * xx;
          s.gv.z += (a[q].z * pp) * xx; s.gv.w += (a[q].w * pp) * xx;
        }
      }
      if (mine) finish_key<L, FUSED>(b, t, u, kr, sub, sub_ok, s, grads, gstride, k, kp, need_init, pen_acc, key_pen(rg));
    }
  }
}

// ---------------------------------------------------------------------------
// k_update_small<L>: the table-resident, fused form of the short-segment role
// (cnt <= BWD_SMALL, ~95 % of the keys of a Zipf batch), written lean on the
// pattern tools/rmw_bench.hip measures at ~40 us for a whole batch: a compact
// argument block (few SGPRs), compile-time row strides when kp == 4L, one level
// of index loads (segment bounds + row id), then every independent load of the
// key at once (header, V, accumulators, first occurrences).
// ---------------------------------------------------------------------------
struct SmallArgs {
  const uint32_t* d_U;
  const uint32_t* col_ptr;
  const uint32_t* urow;
  const uint32_t* s_row;
  const float* s_val;
  const float* slope;
  const float* xv;
  const uint64_t* feaids;
  RowHdr* hdr;
  float* va;
  uint32_t* need_init;
  double* prog;
  int k, kp;
  dfh_updater_param p;
};

#ifndef DFH_BWD_SMALL_KEYS
#define DFH_BWD_SMALL_KEYS 1
#endif
// KPG keys per lane group and iteration, all of their loads in flight together: the role is a chain
// of dependent round trips (segment bounds + row id -> model row + occurrence list -> slopes + XV
// rows), and with the register file capping the waves per SIMD, more independent chains per wave
// are the only way to more requests in flight.
template <int L, bool EXACT>
__device__ __forceinline__ void small_role_lean(const SmallArgs& a, uint32_t wave, uint32_t nwaves, const KeyRange& rg, double& pen_acc) {
  constexpr int G = 64 / L;
  constexpr int KPG = DFH_BWD_SMALL_KEYS;
  const int lane = lane_id();
  const int grp = lane / L;
  const int sub = lane % L;
  const int kp = EXACT ? 4 * L : a.kp;
  const int k = a.k;
  const bool sub_ok = EXACT ? true : (sub * 4 < kp);
  const uint32_t U = *a.d_U;
  float pen = 0.f;
  for (uint32_t u0 = wave * (G * KPG); u0 < U; u0 += nwaves * (G * KPG)) {
    uint32_t u[KPG], beg[KPG], end[KPG], r[KPG];
    bool mine[KPG];
#pragma unroll
    for (int h = 0; h < KPG; ++h) {  // round trip 1: segment bounds + row id
      const uint32_t uu = u0 + h * G + grp;
      u[h] = min(uu, U - 1);
      beg[h] = a.col_ptr[u[h]];
      end[h] = a.col_ptr[u[h] + 1];
      // keys outside the launch's range have no row id in urow (the sharded store resolves only the keys
      // this rank owns): their speculative row loads below go to row 0
      r[h] = key_in(rg, u[h]) ? a.urow[u[h]] : 0u;
    }
    float4 h0[KPG], v[KPG], acc[KPG], gv[KPG];
    float fea_cnt[KPG], gw[KPG], xxp[KPG];
    uint32_t rows[KPG][BWD_SMALL_DEPTH];
    float xs[KPG][BWD_SMALL_DEPTH];
#pragma unroll
    for (int h = 0; h < KPG; ++h) {  // round trip 2: the model row and the first occurrences, back to back
      mine[h] = (u0 + h * G + grp) < U && end[h] - beg[h] <= BWD_SMALL && key_in(rg, u[h]);
      if (!mine[h]) end[h] = beg[h];
      const RowHdr* hp = a.hdr + r[h];
      const float* va = a.va + (size_t)r[h] * (2 * kp);
      h0[h] = ld4(reinterpret_cast<const float*>(hp));  // {w, has_V, sqrt_g, z}
      fea_cnt[h] = hp->fea_cnt;
      v[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      acc[h] = v[h];
      if (k > 0 && sub_ok) {
        v[h] = ld4_nt(va + sub * 4);
        acc[h] = ld4_nt(va + kp + sub * 4);
      }
#pragma unroll
      for (int q = 0; q < BWD_SMALL_DEPTH; ++q) {
        const bool ok = beg[h] + q < end[h];
        rows[h][q] = ok ? a.s_row[beg[h] + q] : 0u;
        xs[h][q] = ok ? (a.s_val ? a.s_val[beg[h] + q] : 1.0f) : 0.f;
      }
      gw[h] = 0.f;
      xxp[h] = 0.f;
      gv[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    uint32_t longest = 0;
#pragma unroll
    for (int h = 0; h < KPG; ++h) longest = max(longest, end[h] - beg[h]);
    for (uint32_t j0 = 0; j0 < longest; j0 += BWD_SMALL_DEPTH) {
      float ps[KPG][BWD_SMALL_DEPTH];
      float4 av[KPG][BWD_SMALL_DEPTH];
#pragma unroll
      for (int h = 0; h < KPG; ++h) {  // round trip 3 (+): slopes and XV rows of these occurrences
#pragma unroll
        for (int q = 0; q < BWD_SMALL_DEPTH; ++q) {
          const bool ok = beg[h] + j0 + q < end[h];
          ps[h][q] = ok ? a.slope[rows[h][q]] : 0.f;
          av[h][q] = (ok && k > 0 && sub_ok) ? ld4(a.xv + (size_t)rows[h][q] * kp + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      const bool more = j0 + BWD_SMALL_DEPTH < longest;
#pragma unroll
      for (int h = 0; h < KPG; ++h) {
#pragma unroll
        for (int q = 0; q < BWD_SMALL_DEPTH; ++q) {  // ascending rows: the reference's order (spmm.h:137-156)
          const float pp = ps[h][q], xx = xs[h][q];
          gw[h] = fma_skip0(pp, xx, gw[h]); xxp[h] = fma_skip0(pp, xx * xx, xxp[h]);
          gv[h].x += (av[h][q].x * pp) * xx; gv[h].y += (av[h][q].y * pp) * xx;
          gv[h].z += (av[h][q].z * pp) * xx; gv[h].w += (av[h][q].w * pp) * xx;
        }
        if (more) {  // the next occurrences of the longer segments
#pragma unroll
          for (int q = 0; q < BWD_SMALL_DEPTH; ++q) {
            const uint32_t j = beg[h] + j0 + BWD_SMALL_DEPTH + q;
            const bool ok = j < end[h];
            rows[h][q] = ok ? a.s_row[j] : 0u;
            xs[h][q] = ok ? (a.s_val ? a.s_val[j] : 1.0f) : 0.f;
          }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < KPG; ++h) {
      if (!mine[h]) continue;
      RowHdr* hp = a.hdr + r[h];
      float* va = a.va + (size_t)r[h] * (2 * kp);
      const float w_old = h0[h].x;
      const bool has_v = k > 0 && __float_as_uint(h0[h].y) != 0;
      float4 g4 = gv[h];
      const float4 vv = v[h];
      if (has_v) {
        // grad_V = X'(diag(p) XV) - diag(XXp) V   (fm_loss.h:181-198); rows without V hold zeros
        g4.x -= vv.x * xxp[h]; g4.y -= vv.y * xxp[h]; g4.z -= vv.z * xxp[h]; g4.w -= vv.w * xxp[h];
        if (sub_ok) pen += 0.5f * a.p.V_l2 * (vv.x * vv.x + vv.y * vv.y + vv.z * vv.z + vv.w * vv.w);
      }
      if (sub == 0) {
        pen += a.p.l1 * fabsf(w_old) + 0.5f * a.p.l2 * w_old * w_old;
        float sqrt_g = h0[h].z, z = h0[h].w;
        const float w_new = ftrl_update_w(gw[h], w_old, sqrt_g, z, a.p);
        uint32_t hv = has_v ? 1u : 0u;
        if (w_old == 0 && w_new != 0 && k > 0 && !has_v && fea_cnt[h] > (float)a.p.V_threshold) {  // sgd_updater.cc:122-126
          if (a.p.init_mode == DFH_INIT_HASH) {
            const uint64_t key = a.feaids[u[h]];
            for (int j = 0; j < kp; ++j) {
              va[j] = j < k ? hash_init_value(key, j, a.p.seed, a.p.V_init_scale) : 0.0f;
              va[kp + j] = 0.0f;
            }
            hv = 1u;
          } else {
            a.need_init[u[h]] = 1;
          }
        }
        st4(reinterpret_cast<float*>(hp), make_float4(w_new, __uint_as_float(hv), sqrt_g, z));
      }
      if (has_v && sub_ok) {
        float4 nv = vv, na = acc[h];
        adagrad_update_v(g4.x, nv.x, na.x, a.p);
        adagrad_update_v(g4.y, nv.y, na.y, a.p);
        adagrad_update_v(g4.z, nv.z, na.z, a.p);
        adagrad_update_v(g4.w, nv.w, na.w, a.p);
        if (!EXACT || k != kp) {
          const int d0 = sub * 4;
          if (d0 + 0 >= k) { nv.x = 0.f; na.x = 0.f; }
          if (d0 + 1 >= k) { nv.y = 0.f; na.y = 0.f; }
          if (d0 + 2 >= k) { nv.z = 0.f; na.z = 0.f; }
          if (d0 + 3 >= k) { nv.w = 0.f; na.w = 0.f; }
        }
        st4_nt(va + sub * 4, nv);
        st4_nt(va + kp + sub * 4, na);
      }
    }
  }
  pen_acc += (double)pen;  // penalty of the pulled weights (sgd_learner.cc:249-273); flushed once per block
}

// ---------------------------------------------------------------------------
// k_backward_all: ONE launch for the whole backward/update of a minibatch.  Blocks
// [0, nb_big) run the long-segment roles (they start first: their chains are the
// longest), the others the short-segment role.  LEAN selects the table-specialised
// short-segment code (fused update on the resident table).
// ---------------------------------------------------------------------------
#ifndef DFH_BWD_WAVES
#define DFH_BWD_WAVES 8
#endif
template <int L, bool FUSED, bool LEAN, bool EXACT>
__global__ void __launch_bounds__(BWD_THREADS, DFH_BWD_WAVES) k_backward_all(BatchView b, RowSrc src, TableView t, float* __restrict__ grads,
                                                                size_t gstride, int k, int kp, uint32_t* __restrict__ need_init,
                                                                uint32_t nb_hot, uint32_t nb_mid, uint32_t nlist, KeyRange rg) {
  constexpr int NW = BWD_THREADS / 64;
  double pen_acc = 0.0;
#ifdef DFH_BWD_TRACE
  const unsigned long long trace_t0 = wall_clock64();
#endif
  // block -> role.  The long-segment blocks come first in dispatch order (their chains are the
  // longest), optionally interleaved 1 : (R - 1) with short-segment blocks so that the first
  // resident set is not long-segment blocks only
  const uint32_t nb_big = nb_hot + nb_mid;
  uint32_t big_id = blockIdx.x, small_id = blockIdx.x - min(blockIdx.x, nb_big);
  bool is_big = blockIdx.x < nb_big;
  if (DFH_BWD_INTERLEAVE > 1) {
    constexpr uint32_t R = DFH_BWD_INTERLEAVE > 1 ? DFH_BWD_INTERLEAVE : 2;
    const uint32_t before = min(nb_big, (blockIdx.x + R - 1) / R);  // long-segment blocks with a smaller id
    is_big = blockIdx.x % R == 0 && blockIdx.x / R < nb_big;
    big_id = blockIdx.x / R;
    small_id = blockIdx.x - before;
  }
  if (is_big && big_id < nb_hot) {
    hot_role<L, FUSED, NW>(b, src, t, grads, gstride, k, kp, need_init, big_id, nb_hot, nlist, rg, pen_acc);
  } else if (is_big) {
    mid_role<L, FUSED, NW>(b, src, t, grads, gstride, k, kp, need_init, (big_id - nb_hot) * NW + (threadIdx.x >> 6),
                           nb_mid * NW, nlist, rg, pen_acc);
  } else {
    const uint32_t wave = small_id * NW + (threadIdx.x >> 6);
    const uint32_t nwaves = (gridDim.x - nb_big) * NW;
    if (LEAN) {
      SmallArgs sa;
      sa.d_U = b.d_U; sa.col_ptr = b.col_ptr; sa.urow = src.urow; sa.s_row = b.s_row; sa.s_val = b.s_val;
      sa.slope = b.slope; sa.xv = b.xv; sa.feaids = b.feaids; sa.hdr = t.hdr; sa.va = t.va; sa.need_init = need_init;
      sa.prog = b.prog; sa.k = k; sa.kp = kp; sa.p = t.p;
      small_role_lean<L, EXACT>(sa, wave, nwaves, rg, pen_acc);
    } else {
      small_role<L, FUSED>(b, src, t, grads, gstride, k, kp, need_init, wave, nwaves, rg, pen_acc);
    }
  }
  if (FUSED || key_pen(rg)) flush_penalty(b, pen_acc);
#ifdef DFH_BWD_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 8192) {  // measurement build only (tools/): per-block role and wall-clock span
    g_bwd_trace[blockIdx.x * 3 + 0] = is_big ? (big_id < nb_hot ? 0ull : 1ull) : 2ull;
    g_bwd_trace[blockIdx.x * 3 + 1] = trace_t0;
    g_bwd_trace[blockIdx.x * 3 + 2] = wall_clock64();
  }
#endif
}

// k_seg_lists: one pass over the unique keys of a localized minibatch, compacting the keys with
// 2 .. BWD_SMALL ("few"), BWD_SMALL+1 .. BWD_MID ("mid") and more ("hot") occurrences into ONE list
// bucket each (one atomic per block and list; the three counters zeroed before).  Used where the minibatch did not come out of k_rdx_emit (library-sort Localizer for
// very large batches, batches localized on the host).
__global__ void __launch_bounds__(1024) k_seg_lists(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ d_U,
                                                    uint2* __restrict__ mid0, uint2* __restrict__ hot0, uint2* __restrict__ few0,
                                                    SegEnt* __restrict__ mid_ent, SegEnt* __restrict__ hot_ent,
                                                    SegEnt* __restrict__ few_ent) {
  __shared__ uint32_t cnt[3], base[3];
  const uint32_t U = *d_U;
  for (uint32_t u0 = blockIdx.x * blockDim.x; u0 < U; u0 += gridDim.x * blockDim.x) {
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t u = u0 + threadIdx.x;
    uint32_t len = 0, slot = 0, cbeg = 0, cend = 0;
    if (u < U) {
      cbeg = col_ptr[u];
      cend = col_ptr[u + 1];
      len = cend - cbeg;
    }
    const int which = len > BWD_MID ? 1 : (len > BWD_SMALL ? 0 : (len > 1 ? 2 : -1));
    if (which >= 0) slot = atomicAdd(&cnt[which], 1u);
    __syncthreads();
    if (threadIdx.x == 0 && cnt[0]) base[0] = atomicAdd(&mid0->x, cnt[0]);
    if (threadIdx.x == 1 && cnt[1]) base[1] = atomicAdd(&hot0->x, cnt[1]);
    if (threadIdx.x == 2 && cnt[2]) base[2] = atomicAdd(&few0->x, cnt[2]);
    __syncthreads();
    const SegEnt e = make_uint4(u, cbeg, cend, 0u);
    if (which == 0) mid_ent[base[0] + slot] = e;
    if (which == 1) hot_ent[base[1] + slot] = e;
    if (which == 2) few_ent[base[2] + slot] = e;
    __syncthreads();
  }
}

// one list bucket at offset 0, empty: what k_seg_lists adds to
__global__ void k_seg_lists_reset(uint2* mid0, uint2* hot0, uint2* few0, uint32_t* split_n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *mid0 = make_uint2(0u, 0u);
    *hot0 = make_uint2(0u, 0u);
    *few0 = make_uint2(0u, 0u);
    *split_n = 0u;   // (these paths list whole segments only: no part lists)
  }
}

// penalty only (validation batches: no backward pass)
template <int L>
__global__ void __launch_bounds__(256) k_penalty(BatchView b, RowSrc src, TableView t, int k, int kp, KeyRange rg) {
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t U = *b.d_U;
  double pen_acc = 0.0;
  for (uint32_t u = wave; u < U; u += nwaves) {
    if (!key_in(rg, u)) continue;
    const uint32_t r = src.urow ? src.urow[u] : u;
    const float2 wf = *reinterpret_cast<const float2*>(src.wbase + (size_t)r * src.wstride);
    float pen = 0.f;
    if (k > 0 && __float_as_uint(wf.y) != 0) {
      for (int d = lane; d < k; d += 64) {
        float vv = src.vbase[(size_t)r * src.vstride + d];
        pen += 0.5f * t.p.V_l2 * vv * vv;
      }
    }
    pen = wave_sum(pen);
    if (lane == 0) pen_acc += (double)pen + (double)t.p.l1 * fabs((double)wf.x) + 0.5 * (double)t.p.l2 * (double)wf.x * (double)wf.x;
  }
  flush_penalty(b, lane == 0 ? pen_acc : 0.0);
}

// ---------------------------------------------------------------------------
// Owner side of the sharded store (device pointers, fixed-stride rows).
// k_pull_rows: Store::Pull -> SGDUpdater::Get for n unique keys: one wave per
// key copies [w, has_V, 0, 0 | V] into the exchange buffer.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pull_rows(TableView t, const uint64_t* __restrict__ keys, uint32_t n,
                                                   float* __restrict__ rows, size_t stride) {
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t u = wave; u < n; u += nwaves) {
    uint32_t r = 0;
    if (lane == 0) r = find_or_insert(t, keys[u]);
    r = __shfl(r, 0, 64);
    const RowHdr h = t.hdr[r];
    float* out = rows + (size_t)u * stride;
    if (lane == 0) st4(out, make_float4(h.w, h.has_V ? 1.0f : 0.0f, 0.f, 0.f));
    const float* va = t.va + (size_t)r * (2 * t.kp);
    for (int d = lane * 4; d < t.kp; d += 256) {
      st4(out + 4 + d, h.has_V ? ld4(va + d) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  }
}

// Push(kGradient) for n unique keys from fixed-stride gradient rows; one wave per key.
// The V part is applied iff the gradient row says V was present at pull time
// (lens[i] > 1 in sgd_updater.cc:90).
__global__ void __launch_bounds__(256) k_push_grad(TableView t, const uint64_t* __restrict__ keys, uint32_t n,
                                                   const float* __restrict__ grads, size_t stride,
                                                   uint32_t* __restrict__ need_init, uint32_t* __restrict__ urow_out) {
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t u = wave; u < n; u += nwaves) {
    uint32_t r = 0;
    if (lane == 0) r = find_or_insert(t, keys[u]);
    r = __shfl(r, 0, 64);
    if (urow_out && lane == 0) urow_out[u] = r;
    const float* g = grads + (size_t)u * stride;
    const float4 g0 = ld4(g);
    const bool had_v = g0.y != 0.0f;
    RowHdr* hp = t.hdr + r;
    if (need_init && lane == 0) need_init[u] = 0;
    if (lane == 0) {
      const float w_old = hp->w;
      float sqrt_g = hp->sqrt_g, z = hp->z;
      const float w_new = ftrl_update_w(g0.x, w_old, sqrt_g, z, t.p);
      hp->w = w_new;
      hp->sqrt_g = sqrt_g;
      hp->z = z;
      if (w_old == 0 && w_new != 0 && t.k > 0 && hp->has_V == 0 && hp->fea_cnt > (float)t.p.V_threshold) {
        if (t.p.init_mode == DFH_INIT_HASH) {
          init_v_hash_row(t, r, keys[u]);
          hp->has_V = 1;
        } else if (need_init) {
          need_init[u] = 1;
        }
      }
    }
    if (had_v && hp->has_V == 0) {  // CHECK(e.V != nullptr), sgd_updater.cc:92
      if (lane == 0) atomicOr(t.err, 4u);
    } else if (had_v) {
      float* va = t.va + (size_t)r * (2 * t.kp);
      for (int d = lane; d < t.k; d += 64) {
        float vv = va[d], acc = va[t.kp + d];
        adagrad_update_v(g[4 + d], vv, acc, t.p);
        va[d] = vv;
        va[t.kp + d] = acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Owner side, resolved form: the keys an owner receives from G source ranks are
// resolved to table rows ONCE per step (k_resolve, duplicates across sources
// allowed), then Pull is one gather over all of them and the two Push kinds run
// per source rank on known rows (no probing, sequential in source order).
// ---------------------------------------------------------------------------
__global__ void k_resolve(TableView t, const uint64_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ rowid) {
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x)
    rowid[u] = find_or_insert(t, keys[u]);
}

// Pull on resolved rows: L lanes per key copy [w, has_V, 0, 0 | V] (float4 per lane)
template <int L>
__global__ void __launch_bounds__(256) k_pull_resolved(TableView t, const uint32_t* __restrict__ rowid, uint32_t n,
                                                       float* __restrict__ rows, size_t stride) {
  constexpr int GPW = 64 / L;
  const int lane = lane_id();
  const int sub = lane % L;
  const uint32_t group = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * GPW + lane / L;
  const uint32_t ngroups = ((gridDim.x * blockDim.x) >> 6) * GPW;
  for (uint32_t u = group; u < n; u += ngroups) {
    const uint32_t r = rowid[u] & kRowMask;  // (k_resolve_multi marks a key's worker entry in bit 31)
    const float4 h0 = ld4(reinterpret_cast<const float*>(t.hdr + r));  // {w, has_V, sqrt_g, z}
    const bool hv = __float_as_uint(h0.y) != 0u;
    float* out = rows + (size_t)u * stride;
    if (sub == 0) st4(out, make_float4(h0.x, hv ? 1.0f : 0.0f, 0.f, 0.f));
    const float* va = t.va + (size_t)r * (2 * t.kp);
    for (int d = sub * 4; d < t.kp; d += 4 * L) st4(out + 4 + d, hv ? ld4(va + d) : make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

// Push(kGradient) on resolved rows of ONE source rank (unique inside the launch): L lanes per key
template <int L>
__global__ void __launch_bounds__(256) k_push_grad_resolved(TableView t, const uint32_t* __restrict__ rowid,
                                                            const uint64_t* __restrict__ keys, uint32_t n,
                                                            const float* __restrict__ grads, size_t stride) {
  constexpr int GPW = 64 / L;
  const int lane = lane_id();
  const int sub = lane % L;
  const uint32_t group = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * GPW + lane / L;
  const uint32_t ngroups = ((gridDim.x * blockDim.x) >> 6) * GPW;
  for (uint32_t u = group; u < n; u += ngroups) {
    const uint32_t r = rowid[u] & kRowMask;
    const float* g = grads + (size_t)u * stride;
    const float4 g0 = ld4(g);
    const bool had_v = g0.y != 0.0f;
    RowHdr* hp = t.hdr + r;
    const uint32_t has_v = hp->has_V;
    if (had_v && !has_v) {  // CHECK(e.V != nullptr), sgd_updater.cc:92
      if (sub == 0) atomicOr(t.err, 4u);
    } else if (had_v) {
      float* va = t.va + (size_t)r * (2 * t.kp);
      for (int d = sub * 4; d < t.kp; d += 4 * L) {
        const float4 gv = ld4(g + 4 + d);
        float4 v = ld4(va + d), acc = ld4(va + t.kp + d);
        adagrad_update_v(gv.x, v.x, acc.x, t.p);
        adagrad_update_v(gv.y, v.y, acc.y, t.p);
        adagrad_update_v(gv.z, v.z, acc.z, t.p);
        adagrad_update_v(gv.w, v.w, acc.w, t.p);
        if (d + 0 >= t.k) { v.x = 0.f; acc.x = 0.f; }
        if (d + 1 >= t.k) { v.y = 0.f; acc.y = 0.f; }
        if (d + 2 >= t.k) { v.z = 0.f; acc.z = 0.f; }
        if (d + 3 >= t.k) { v.w = 0.f; acc.w = 0.f; }
        st4(va + d, v);
        st4(va + t.kp + d, acc);
      }
    }
    if (sub == 0) {
      const float w_old = hp->w;
      float sqrt_g = hp->sqrt_g, z = hp->z;
      const float w_new = ftrl_update_w(g0.x, w_old, sqrt_g, z, t.p);
      hp->w = w_new;
      hp->sqrt_g = sqrt_g;
      hp->z = z;
      // lazy InitV when w leaves zero (sgd_updater.cc:122-126); HASH init only on this path
      if (w_old == 0 && w_new != 0 && t.k > 0 && has_v == 0 && hp->fea_cnt > (float)t.p.V_threshold) {
        init_v_hash_row(t, r, keys[u]);
        hp->has_V = 1;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Owner side, one launch for ALL source ranks of a step.  The received keys are the
// concatenation of nsrc ascending lists (SegOff).  k_resolve_multi resolves every entry to
// its row and elects ONE entry per key, the first to arrive, as the key's WORKER: it claims the
// row's step word (RowHdr::pad[slot], 0 outside a step) with a compare-and-swap, leaving
// (entry + 1) << 5 there; every later entry of the key bumps the word's low 5 bits (one
// atomic add: the old value names the worker and the entry's place k) and writes its own index
// into the worker's extras, extra[worker * XS + k].  In the Push kernels the worker — bit 31 of
// its row word — has the number of the key's other entries with the header it reads anyway and
// their indices in one 16 / 32 B read; it orders them by source (sources are concatenated in
// ascending order: by entry index) and applies their values one after the other in source
// order — exactly what per-source launches would do — storing the row once.  Every other entry
// skips on its row word alone.  (Round 4 searched each later source's list by bisection, 14
// dependent loads per source; a linked list through the entries, tried first in round 5, still
// cost one dependent — and, written by another launch, cold — load per entry: 57 us for the
// gradient push at C4 size against 25 us for the same rows without the multi-source keys.)
// The last Push of a step (or k_release_rows) clears the step word.
// rowid holds multi_words(n, nsrc) words: [0, n) the row words, from multi_extra_base(n) on
// XS = multi_xs(nsrc) extras per entry.
// ---------------------------------------------------------------------------
struct SegOff {
  uint32_t off[33];  // entries of source s are [off[s], off[s+1])
  int nsrc;
  int slot;          // which of the two per-row step words (RowHdr::pad[0..1]) this step uses: two steps
                     // may be in flight on an owner (one resolved and pulled, the other awaiting its gradients)
};
constexpr uint32_t ROW_ID_MASK = kRowMask;  // a table holds fewer than 2^28 rows
constexpr uint32_t ROW_WORKER = 0x80000000u;
constexpr int MULTI_FAST = 8;                  // entries per key ordered in registers (one node: 7 peers)
__host__ __device__ __forceinline__ uint32_t multi_xs(int nsrc) { return (uint32_t)((nsrc - 1 + 3) / 4 * 4 > 0 ? (nsrc - 1 + 3) / 4 * 4 : 4); }
__host__ __device__ __forceinline__ size_t multi_extra_base(size_t n) { return (n + 3) & ~(size_t)3; }
// Behind the row words and the extras, the key lists k_count_pull_multi leaves for k_push_grad_chunks: the entries are cut
// into chunks of MULTI_CH; a chunk's keys of ONE source (its workers without extras) stand in slist[c * MULTI_CH ...) as
// {entry, row}, its keys of several sources in mrec[(c * MULTI_CH + j) * (XS + 4) ...) as {entry | m << 27, row, -, -,
// the other entries in ascending (= source) order}, chdr[c] = {keys of one source, keys of several}.
#ifndef DFH_MULTI_CH
#define DFH_MULTI_CH 32   // (16 / 32 / 64 / 128: push 48.7 / 45.1 / 53.2 / 66.6 us, count + pull 36 / 36 / 40 / 45 us at N = 8 size)
#endif
constexpr uint32_t MULTI_CH = DFH_MULTI_CH;
__host__ __device__ __forceinline__ size_t multi_slist_base(size_t n, int nsrc) { return multi_extra_base(n) + n * multi_xs(nsrc); }
__host__ __device__ __forceinline__ size_t multi_ch_entries(size_t n) { return (n + MULTI_CH - 1) / MULTI_CH * MULTI_CH; }  // whole chunks: the readers request a chunk's first entries with its counts
__host__ __device__ __forceinline__ size_t multi_mrec_base(size_t n, int nsrc) { return multi_slist_base(n, nsrc) + 2 * multi_ch_entries(n); }
__host__ __device__ __forceinline__ size_t multi_chdr_base(size_t n, int nsrc) {
  return multi_mrec_base(n, nsrc) + multi_ch_entries(n) * (multi_xs(nsrc) + 4);
}
__host__ __device__ __forceinline__ size_t multi_words(size_t n, int nsrc) {
  return multi_chdr_base(n, nsrc) + 2 * ((n + MULTI_CH - 1) / MULTI_CH + 1);
}

__global__ void k_resolve_multi(TableView t, const uint64_t* __restrict__ keys, SegOff g, uint32_t* __restrict__ rowid) {
  const uint32_t n = g.off[g.nsrc];
  const uint32_t XS = multi_xs(g.nsrc);
  uint32_t* __restrict__ extra = rowid + multi_extra_base(n);
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const uint32_t r = find_or_insert(t, keys[e]);
    uint32_t* word = &t.hdr[r].pad[g.slot];
    uint32_t old = atomicCAS(word, 0u, (e + 1) << 5);
    const bool worker = old == 0u;
    if (!worker) {  // the key has its worker: take a place among its extras
      // A source's list is unique, so a key has at most nsrc - 1 <= XS extras.  More entries on one row are an error of the
      // caller (a list that repeats a key) or a table that overflowed (find_or_insert parks every key beyond the capacity on
      // the last row): flagged, and the count SATURATES — the Push kernels are queued before the host reads the error word,
      // and a count beyond the extras (or a carry out of the low five bits into the worker's index) would have them read
      // indices that were never written.
      const uint32_t lim = (uint32_t)(g.nsrc - 1);
      for (;;) {
        if ((old & 31u) >= lim) {
          atomicOr(t.err, 2u);
          break;
        }
        const uint32_t seen = atomicCAS(word, old, old + 1u);
        if (seen == old) {
          extra[(size_t)((old >> 5) - 1u) * XS + (old & 31u)] = e;
          break;
        }
        old = seen;
      }
    }
    rowid[e] = r | (worker ? ROW_WORKER : 0u);
  }
}

__global__ void k_release_rows(TableView t, const uint32_t* __restrict__ rowid, uint32_t n, int slot) {
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const uint32_t rw = rowid[e];
    if (rw & ROW_WORKER) t.hdr[rw & ROW_ID_MASK].pad[slot] = 0;
  }
}

// Push(kFeaCount) of all sources: one thread per entry, the key's worker adds every source's count
// (small integers: the sum is exact in any order) and then takes the InitV decision once —
// w does not change during count pushes and fea_cnt only grows, so the outcome equals the
// sequential one (sgd_updater.cc:62-73)
__global__ void k_push_count_multi(TableView t, const uint32_t* __restrict__ rowid, const uint64_t* __restrict__ keys,
                                   SegOff g, const float* __restrict__ cnt) {
  const uint32_t n = g.off[g.nsrc];
  const uint32_t XS = multi_xs(g.nsrc);
  const uint32_t* __restrict__ extra = rowid + multi_extra_base(n);
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const uint32_t rw = rowid[e];
    if (!(rw & ROW_WORKER)) continue;  // another entry of this key works for it
    const uint32_t r = rw & ROW_ID_MASK;
    RowHdr& h = t.hdr[r];
    float fc = h.fea_cnt + cnt[e];
    const uint32_t m = min(h.pad[g.slot] & 31u, XS);
    for (uint32_t i = 0; i < m; ++i) fc += cnt[extra[(size_t)e * XS + i]];
    h.fea_cnt = fc;
    if (t.k > 0 && h.has_V == 0 && h.w != 0 && fc > (float)t.p.V_threshold) {
      init_v_hash_row(t, r, keys[e]);
      h.has_V = 1;
    }
  }
}

// Push(kGradient) of all sources: L lanes per entry; the key's worker applies the sources' gradient rows
// one after the other (FTRL on w with lazy InitV, AdaGrad on V iff the rows were pulled with V) on
// registers and stores the row once; clears the row's step word.  The header, the worker's own gradient
// row, the V / accumulator slices and the extras are requested together, before anything is known about
// the key: a key of one source (80 %) is row word -> {header, row, gradient} -> store, a key of several
// sources one round trip more (the other sources' gradient rows, four at a time).
// measurement switches (tools/owner_bench.py at N = 8 size, profiles/r05a_*): streaming hints on the row loads / stores change
// nothing (58.1 against 57.4 us), nor does the number of gradient rows requested per round trip (1 / 2 / 8: 51.1 / 46.6 / 50.5
// against 47.4 us for 4) or the waves per SIMD those leave room for
#ifndef DFH_PGM_NT
#define DFH_PGM_NT 0
#endif
#ifndef DFH_PGM_BATCH
#define DFH_PGM_BATCH 4
#endif
constexpr int PGM_BATCH = DFH_PGM_BATCH;
template <int L>
__global__ void __launch_bounds__(256) k_push_grad_multi(TableView t, const uint32_t* __restrict__ rowid,
                                                         const uint64_t* __restrict__ keys, SegOff g,
                                                         const float* __restrict__ grads, size_t stride) {
  constexpr int GPW = 64 / L;
  const int lane = lane_id();
  const int sub = lane % L;
  const uint32_t group = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * GPW + lane / L;
  const uint32_t ngroups = ((gridDim.x * blockDim.x) >> 6) * GPW;
  const uint32_t n = g.off[g.nsrc];
  const uint32_t XS = multi_xs(g.nsrc);
  const uint32_t* __restrict__ extra = rowid + multi_extra_base(n);
  const int d = sub * 4;
  const bool d_ok = d < t.kp;  // this lane's V slice (4 * L >= kp by dispatch: one float4 per lane covers the row)
  for (uint32_t e = group; e < n; e += ngroups) {
    const uint32_t rw = rowid[e];
    if (!(rw & ROW_WORKER)) continue;
    const uint32_t r = rw & ROW_ID_MASK;
    RowHdr* hp = t.hdr + r;
    float* va = t.va + (size_t)r * (2 * t.kp);
    // lanes beyond the row (and V_dim = 0) read valid addresses they do not use: the loads stay unconditional
    const int goff = d_ok ? 4 + d : 0;
    const float* vp = d_ok ? va + d : reinterpret_cast<const float*>(hp);
    const float* ap = d_ok ? va + t.kp + d : reinterpret_cast<const float*>(hp);
    // one round trip: header (both halves), own gradient row, V and accumulator slices, the extras
    const float4 h0 = ld4(reinterpret_cast<const float*>(hp));  // {w, has_V, sqrt_g, z}
    const float fea_cnt = hp->fea_cnt;
    const uint32_t word = hp->pad[g.slot];
    const float* g_own = grads + (size_t)e * stride;
    const float4 go0 = ld4(g_own);
    const float4 go_v = ld4(g_own + goff);
    float4 v = DFH_PGM_NT ? ld4_nt(vp) : ld4(vp), acc = DFH_PGM_NT ? ld4_nt(ap) : ld4(ap);
    const uint32_t* xp = extra + (size_t)e * XS;
    const uint4 x0 = *reinterpret_cast<const uint4*>(xp);
    const uint4 x1 = *reinterpret_cast<const uint4*>(xp + (XS >= 8 ? 4 : 0));
    float w = h0.x, sqrt_g = h0.z, z = h0.w;
    uint32_t has_v = __float_as_uint(h0.y);
    const bool had_v = go0.y != 0.0f;  // every source pulled the same model version: one answer
    if (had_v && !has_v) {             // CHECK(e.V != nullptr), sgd_updater.cc:92
      if (sub == 0) atomicOr(t.err, 4u);
      if (sub == 0) hp->pad[g.slot] = 0;
      continue;
    }
    if (!(had_v && d_ok)) v = acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto apply = [&](float gw, const float4& gv) {
      const float w_old = w;
      w = ftrl_update_w(gw, w_old, sqrt_g, z, t.p);
      // lazy InitV when w leaves zero (sgd_updater.cc:122-126); the pulled rows had no V, so no
      // gradient of this step touches the fresh values
      if (w_old == 0 && w != 0 && t.k > 0 && has_v == 0 && fea_cnt > (float)t.p.V_threshold) {
        // every lane of the group reaches this with the same w: each writes its own 16 B of V and of the accumulators
        if (d_ok) init_v_hash_slice(t, r, keys[e], d);
        has_v = 1;
      }
      if (had_v && d_ok) {
        adagrad_update_v(gv.x, v.x, acc.x, t.p);
        adagrad_update_v(gv.y, v.y, acc.y, t.p);
        adagrad_update_v(gv.z, v.z, acc.z, t.p);
        adagrad_update_v(gv.w, v.w, acc.w, t.p);
      }
    };
    const uint32_t m = min(word & 31u, XS);  // entries of this key besides the worker's own
    if (m == 0) {                   // the usual case: one source carries the key
      apply(go0.x, go_v);
    } else if (m < MULTI_FAST) {
      // the worker's and the other sources' entries in ascending order: an odd-even transposition network on 8 registers
      uint32_t ent[MULTI_FAST] = {e, x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z};
#pragma unroll
      for (int i = 1; i < MULTI_FAST; ++i)
        if ((uint32_t)i > m) ent[i] = 0xFFFFFFFFu;
#pragma unroll
      for (int pass = 0; pass < MULTI_FAST; ++pass) {
#pragma unroll
        for (int i = pass & 1; i + 1 < MULTI_FAST; i += 2) {
          const uint32_t lo = min(ent[i], ent[i + 1]), hi = max(ent[i], ent[i + 1]);
          ent[i] = lo;
          ent[i + 1] = hi;
        }
      }
      // PGM_BATCH gradient rows per round trip (unconditional loads on clamped entries), applied in order
#pragma unroll
      for (int b0 = 0; b0 < MULTI_FAST; b0 += PGM_BATCH) {
        if ((uint32_t)b0 > m) break;  // uniform per lane group
        float gws[PGM_BATCH];
        float4 gvs[PGM_BATCH];
#pragma unroll
        for (int i = 0; i < PGM_BATCH; ++i) {
          const float* gp = grads + (size_t)((uint32_t)(b0 + i) <= m ? ent[b0 + i] : e) * stride;
          gws[i] = ld4(gp).x;
          gvs[i] = ld4(gp + goff);
        }
#pragma unroll
        for (int i = 0; i < PGM_BATCH; ++i)
          if ((uint32_t)(b0 + i) <= m) apply(gws[i], gvs[i]);
      }
    } else {
      // more than 8 sources carry the key (a job beyond one node): the next entry in ascending order is searched among
      // the worker's own and its extras for every application
      uint32_t cur = 0xFFFFFFFFu;
      bool first = true;
      for (uint32_t done = 0; done <= m; ++done) {
        uint32_t best = (first || e > cur) ? e : 0xFFFFFFFFu;
        for (uint32_t i = 0; i < m; ++i) {
          const uint32_t x = xp[i];
          if ((first || x > cur) && x < best) best = x;
        }
        const float* gp = grads + (size_t)best * stride;
        apply(ld4(gp).x, ld4(gp + goff));
        cur = best;
        first = false;
      }
    }
    if (sub == 0) {
      st4(reinterpret_cast<float*>(hp), make_float4(w, __uint_as_float(has_v), sqrt_g, z));
      hp->pad[g.slot] = 0;
    }
    if (had_v && d_ok) {
      if (d + 0 >= t.k) { v.x = 0.f; acc.x = 0.f; }
      if (d + 1 >= t.k) { v.y = 0.f; acc.y = 0.f; }
      if (d + 2 >= t.k) { v.z = 0.f; acc.z = 0.f; }
      if (d + 3 >= t.k) { v.w = 0.f; acc.w = 0.f; }
      if (DFH_PGM_NT) {
        st4_nt(va + d, v);
        st4_nt(va + t.kp + d, acc);
      } else {
        st4(va + d, v);
        st4(va + t.kp + d, acc);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Owner side per DISTINCT key (round 6).  k_count_pull_multi is Push(kFeaCount) of all sources and Pull in ONE launch:
// a key's worker adds the sources' counts, takes the InitV decision once (sgd_updater.cc:62-73), reads the row ONCE and
// writes [w, has_V, 0, 0 | V] to the output row of every entry of the key (its own and its extras') — the duplicate keys of
// k_pull_resolved re-read their row (81.7 MB counted for 58.8 MB at N = 8 size), and the count push was a launch of its own.
// On the way it leaves the step's keys as lists per chunk of MULTI_CH entries (layout above), so that k_push_grad_chunks
// runs over keys, not entries: in k_push_grad_multi 36 % of the lane groups idle on entries that are not workers and a
// wave takes the several-sources path (a third round trip) when ONE of its four keys has extras — 59 % of the waves for
// 20 % of the keys.  Here a wave's keys are all of one kind, the keys of several sources have their other entries in
// source order with the first read, and their gradient rows are requested with the row: two round trips for both kinds.
// ---------------------------------------------------------------------------
template <int L, bool HAS_CNT>
__global__ void __launch_bounds__(256) k_count_pull_multi(TableView t, uint32_t* __restrict__ rowid, const uint64_t* __restrict__ keys,
                                                          SegOff g, const float* __restrict__ cnt, float* __restrict__ rows,
                                                          size_t stride) {
  constexpr uint32_t G = 256 / L;
  __shared__ uint32_t sh_n[2];
  const uint32_t sub = threadIdx.x % L, grp = threadIdx.x / L;
  const uint32_t n = g.off[g.nsrc];
  const uint32_t XS = multi_xs(g.nsrc), RS = XS + 4;
  const uint32_t* __restrict__ extra = rowid + multi_extra_base(n);
  uint2* __restrict__ slist = reinterpret_cast<uint2*>(rowid + multi_slist_base(n, g.nsrc));
  uint32_t* __restrict__ mrec = rowid + multi_mrec_base(n, g.nsrc);
  uint2* __restrict__ chdr = reinterpret_cast<uint2*>(rowid + multi_chdr_base(n, g.nsrc));
  const uint32_t nchunk = (n + MULTI_CH - 1) / MULTI_CH;
  const int d = (int)sub * 4;
  const bool d_ok = d < t.kp;
  for (uint32_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    if (threadIdx.x == 0) sh_n[0] = sh_n[1] = 0;
    __syncthreads();
    for (uint32_t i = grp; i < MULTI_CH; i += G) {
      const uint32_t e = c * MULTI_CH + i;
      if (e >= n) continue;
      const uint32_t rw = rowid[e];
      if (!(rw & ROW_WORKER)) continue;  // another entry of this key works for it
      const uint32_t r = rw & ROW_ID_MASK;
      RowHdr* hp = t.hdr + r;
      const float* va = t.va + (size_t)r * (2 * t.kp);
      // one round trip: the header (both halves), this lane's V slice, the extras
      const float4 h0 = ld4(reinterpret_cast<const float*>(hp));  // {w, has_V, sqrt_g, z}
      const float fea_cnt = hp->fea_cnt;
      const uint32_t word = hp->pad[g.slot];
      float4 v = ld4(d_ok ? va + d : reinterpret_cast<const float*>(hp));
      const uint32_t* xp = extra + (size_t)e * XS;
      const uint4 x0 = *reinterpret_cast<const uint4*>(xp);
      const uint4 x1 = *reinterpret_cast<const uint4*>(xp + (XS >= 8 ? 4 : 0));
      const uint32_t m = min(word & 31u, XS);  // entries of this key besides the worker's own
      bool hv = __float_as_uint(h0.y) != 0u;
      // the other entries in ascending order (sources are concatenated in ascending order): a network on 8 registers; more
      // than 8 (a job beyond one node) are ordered by selection out of memory, below
      uint32_t ent[MULTI_FAST] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      const bool fast = m <= (uint32_t)MULTI_FAST;
      if (m > 0 && fast) {
#pragma unroll
        for (int j = 0; j < MULTI_FAST; ++j)
          if ((uint32_t)j >= m || (j >= 4 && XS < 8)) ent[j] = 0xFFFFFFFFu;
#pragma unroll
        for (int pass = 0; pass < MULTI_FAST; ++pass) {
#pragma unroll
          for (int j = pass & 1; j + 1 < MULTI_FAST; j += 2) {
            const uint32_t lo = min(ent[j], ent[j + 1]), hi = max(ent[j], ent[j + 1]);
            ent[j] = lo;
            ent[j + 1] = hi;
          }
        }
      }
      if (HAS_CNT) {
        float fc = fea_cnt + cnt[e];  // small integers: the sum is exact in any order
        if (fast) {
#pragma unroll
          for (int j = 0; j < MULTI_FAST; ++j)
            if ((uint32_t)j < m) fc += cnt[ent[j]];
        } else {
          for (uint32_t j = 0; j < m; ++j) fc += cnt[xp[j]];
        }
        if (t.k > 0 && !hv && h0.x != 0 && fc > (float)t.p.V_threshold) {
          // every lane of the group writes its own 16 B of V and of the accumulators, and answers with what it wrote
          if (d_ok) {
            init_v_hash_slice(t, r, keys[e], d);
            v.x = d + 0 < t.k ? hash_init_value(keys[e], d + 0, t.p.seed, t.p.V_init_scale) : 0.0f;
            v.y = d + 1 < t.k ? hash_init_value(keys[e], d + 1, t.p.seed, t.p.V_init_scale) : 0.0f;
            v.z = d + 2 < t.k ? hash_init_value(keys[e], d + 2, t.p.seed, t.p.V_init_scale) : 0.0f;
            v.w = d + 3 < t.k ? hash_init_value(keys[e], d + 3, t.p.seed, t.p.V_init_scale) : 0.0f;
          }
          hv = true;
          if (sub == 0) hp->has_V = 1;
        }
        if (sub == 0) hp->fea_cnt = fc;
      }
      if (!hv) v = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 head = make_float4(h0.x, hv ? 1.0f : 0.0f, 0.f, 0.f);
      auto put = [&](uint32_t idx) {
        float* out = rows + (size_t)idx * stride;
        if (sub == 0) st4(out, head);
        if (d_ok) st4(out + 4 + d, v);
      };
      put(e);
      if (fast) {
#pragma unroll
        for (int j = 0; j < MULTI_FAST; ++j)
          if ((uint32_t)j < m) put(ent[j]);
      } else {
        for (uint32_t j = 0; j < m; ++j) put(xp[j]);
      }
      if (sub == 0) {
        if (m == 0) {
          slist[(size_t)c * MULTI_CH + atomicAdd(&sh_n[0], 1u)] = make_uint2(e, r);
        } else {
          uint32_t* rec = mrec + ((size_t)c * MULTI_CH + atomicAdd(&sh_n[1], 1u)) * RS;
          *reinterpret_cast<uint4*>(rec) = make_uint4(e | (m << 27), r, 0u, 0u);
          if (fast) {
            *reinterpret_cast<uint4*>(rec + 4) = make_uint4(ent[0], ent[1], ent[2], ent[3]);
            if (XS >= 8) *reinterpret_cast<uint4*>(rec + 8) = make_uint4(ent[4], ent[5], ent[6], ent[7]);
          } else {
            uint32_t cur = 0;
            for (uint32_t done = 0; done < m; ++done) {  // selection: the next entry in ascending order
              uint32_t best = 0xFFFFFFFFu;
              for (uint32_t j = 0; j < m; ++j) {
                const uint32_t x = xp[j];
                if ((done == 0 || x > cur) && x < best) best = x;
              }
              rec[4 + done] = best;
              cur = best;
            }
          }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) chdr[c] = make_uint2(sh_n[0], sh_n[1]);
    __syncthreads();
  }
}

// Push(kGradient) of all sources over the key lists of k_count_pull_multi: L lanes per KEY, a block per chunk; FTRL on w
// with lazy InitV, AdaGrad on V iff the rows were pulled with V, the sources' gradient rows applied in source order on
// registers, the row stored once, its step word cleared (what k_push_grad_multi does per entry).
template <int L>
__global__ void __launch_bounds__(256) k_push_grad_chunks(TableView t, const uint32_t* __restrict__ rowid,
                                                          const uint64_t* __restrict__ keys, SegOff g,
                                                          const float* __restrict__ grads, size_t stride) {
  constexpr uint32_t G = 256 / L;
  const uint32_t sub = threadIdx.x % L, grp = threadIdx.x / L;
  const uint32_t n = g.off[g.nsrc];
  const uint32_t XS = multi_xs(g.nsrc), RS = XS + 4;
  const uint2* __restrict__ slist = reinterpret_cast<const uint2*>(rowid + multi_slist_base(n, g.nsrc));
  const uint32_t* __restrict__ mrec = rowid + multi_mrec_base(n, g.nsrc);
  const uint2* __restrict__ chdr = reinterpret_cast<const uint2*>(rowid + multi_chdr_base(n, g.nsrc));
  const uint32_t nchunk = (n + MULTI_CH - 1) / MULTI_CH;
  const int d = (int)sub * 4;
  const bool d_ok = d < t.kp;  // this lane's V slice (4 * L >= kp by dispatch: one float4 per lane covers the row)
  const int goff = d_ok ? 4 + d : 0;
  for (uint32_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    const uint2 ch = chdr[c];
    const size_t base = (size_t)c * MULTI_CH;
    uint2 en = slist[base + min(grp, MULTI_CH - 1)];  // requested with the chunk's counts
    uint4 r0 = *reinterpret_cast<const uint4*>(mrec + (base + min(grp, MULTI_CH - 1)) * RS);
    // ---- keys of one source: row word -> {header, row, gradient} -> store
    for (uint32_t i = grp; i < ch.x; i += G) {
      if (i != grp) en = slist[base + i];
      const uint32_t e = en.x, r = en.y;
      RowHdr* hp = t.hdr + r;
      float* va = t.va + (size_t)r * (2 * t.kp);
      // lanes beyond the row (and V_dim = 0) read valid addresses they do not use: the loads stay unconditional
      const float* vp = d_ok ? va + d : reinterpret_cast<const float*>(hp);
      const float* ap = d_ok ? va + t.kp + d : reinterpret_cast<const float*>(hp);
      const float4 h0 = ld4(reinterpret_cast<const float*>(hp));  // {w, has_V, sqrt_g, z}
      const float fea_cnt = hp->fea_cnt;
      const float* g_own = grads + (size_t)e * stride;
      const float4 go0 = ld4(g_own);
      const float4 gv = ld4(g_own + goff);
      float4 v = ld4(vp), acc = ld4(ap);
      float w = h0.x, sqrt_g = h0.z, z = h0.w;
      uint32_t has_v = __float_as_uint(h0.y);
      const bool had_v = go0.y != 0.0f;
      if (had_v && !has_v) {  // CHECK(e.V != nullptr), sgd_updater.cc:92
        if (sub == 0) atomicOr(t.err, 4u);
        if (sub == 0) hp->pad[g.slot] = 0;
        continue;
      }
      const float w_old = w;
      w = ftrl_update_w(go0.x, w_old, sqrt_g, z, t.p);
      // lazy InitV when w leaves zero (sgd_updater.cc:122-126); the pulled rows had no V, so no gradient touches the fresh values
      if (w_old == 0 && w != 0 && t.k > 0 && has_v == 0 && fea_cnt > (float)t.p.V_threshold) {
        if (d_ok) init_v_hash_slice(t, r, keys[e], d);
        has_v = 1;
      }
      if (sub == 0) {
        st4(reinterpret_cast<float*>(hp), make_float4(w, __uint_as_float(has_v), sqrt_g, z));
        hp->pad[g.slot] = 0;
      }
      if (had_v && d_ok) {
        adagrad_update_v(gv.x, v.x, acc.x, t.p);
        adagrad_update_v(gv.y, v.y, acc.y, t.p);
        adagrad_update_v(gv.z, v.z, acc.z, t.p);
        adagrad_update_v(gv.w, v.w, acc.w, t.p);
        if (d + 0 >= t.k) { v.x = 0.f; acc.x = 0.f; }
        if (d + 1 >= t.k) { v.y = 0.f; acc.y = 0.f; }
        if (d + 2 >= t.k) { v.z = 0.f; acc.z = 0.f; }
        if (d + 3 >= t.k) { v.w = 0.f; acc.w = 0.f; }
        st4(va + d, v);
        st4(va + t.kp + d, acc);
      }
    }
    // ---- keys of several sources: record -> {header, row, the first four gradient rows} -> [the others] -> store
    for (uint32_t i = grp; i < ch.y; i += G) {
      const uint32_t* rec = mrec + (base + i) * RS;
      if (i != grp) r0 = *reinterpret_cast<const uint4*>(rec);
      const uint4 x0 = *reinterpret_cast<const uint4*>(rec + 4);
      const uint32_t e = r0.x & 0x07FFFFFFu, m = r0.x >> 27, r = r0.y;
      RowHdr* hp = t.hdr + r;
      float* va = t.va + (size_t)r * (2 * t.kp);
      const float* vp = d_ok ? va + d : reinterpret_cast<const float*>(hp);
      const float* ap = d_ok ? va + t.kp + d : reinterpret_cast<const float*>(hp);
      const float4 h0 = ld4(reinterpret_cast<const float*>(hp));
      const float fea_cnt = hp->fea_cnt;
      const float* g_own = grads + (size_t)e * stride;
      const float4 go0 = ld4(g_own);
      const float4 go_v = ld4(g_own + goff);
      float4 v = ld4(vp), acc = ld4(ap);
      float w = h0.x, sqrt_g = h0.z, z = h0.w;
      uint32_t has_v = __float_as_uint(h0.y);
      const bool had_v = go0.y != 0.0f;  // every source pulled the same model version: one answer
      if (had_v && !has_v) {
        if (sub == 0) atomicOr(t.err, 4u);
        if (sub == 0) hp->pad[g.slot] = 0;
        continue;
      }
      if (!(had_v && d_ok)) v = acc = make_float4(0.f, 0.f, 0.f, 0.f);
      auto apply = [&](float gw, const float4& gv) {
        const float w_old = w;
        w = ftrl_update_w(gw, w_old, sqrt_g, z, t.p);
        if (w_old == 0 && w != 0 && t.k > 0 && has_v == 0 && fea_cnt > (float)t.p.V_threshold) {
          if (d_ok) init_v_hash_slice(t, r, keys[e], d);
          has_v = 1;
        }
        if (had_v && d_ok) {
          adagrad_update_v(gv.x, v.x, acc.x, t.p);
          adagrad_update_v(gv.y, v.y, acc.y, t.p);
          adagrad_update_v(gv.z, v.z, acc.z, t.p);
          adagrad_update_v(gv.w, v.w, acc.w, t.p);
        }
      };
      // the first four of the other entries' gradient rows travel with the row (unconditional loads on clamped entries)
      const uint32_t xs[4] = {x0.x, x0.y, x0.z, x0.w};
      float gws[4];
      float4 gvs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* gp = grads + (size_t)((uint32_t)j < m ? xs[j] : e) * stride;
        gws[j] = ld4(gp).x;
        gvs[j] = ld4(gp + goff);
      }
      bool own_done = false;  // the worker's own entry takes its place in the ascending order
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if ((uint32_t)j < m) {
          if (!own_done && e < xs[j]) {
            apply(go0.x, go_v);
            own_done = true;
          }
          apply(gws[j], gvs[j]);
        }
      }
      for (uint32_t j = 4; j < m; ++j) {  // rare: a key five or more sources carry
        const uint32_t x = rec[4 + j];
        const float* gp = grads + (size_t)x * stride;
        const float gw = ld4(gp).x;
        const float4 gv = ld4(gp + goff);
        if (!own_done && e < x) {
          apply(go0.x, go_v);
          own_done = true;
        }
        apply(gw, gv);
      }
      if (!own_done) apply(go0.x, go_v);
      if (sub == 0) {
        st4(reinterpret_cast<float*>(hp), make_float4(w, __uint_as_float(has_v), sqrt_g, z));
        hp->pad[g.slot] = 0;
      }
      if (had_v && d_ok) {
        if (d + 0 >= t.k) { v.x = 0.f; acc.x = 0.f; }
        if (d + 1 >= t.k) { v.y = 0.f; acc.y = 0.f; }
        if (d + 2 >= t.k) { v.z = 0.f; acc.z = 0.f; }
        if (d + 3 >= t.k) { v.w = 0.f; acc.w = 0.f; }
        st4(va + d, v);
        st4(va + t.kp + d, acc);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Literal Loss API kernels: arbitrary V_dim, arbitrary w_pos/V_pos into a
// ragged weights array (host SArray semantics).  Sums run serially in the
// reference's order per output element, so these agree with the CPU path to
// rounding of FMA contraction only.
// ---------------------------------------------------------------------------
// one wave per example; lanes stride over the embedding dimension
__global__ void __launch_bounds__(256) k_predict_generic(uint32_t nrows, const uint32_t* __restrict__ offset,
                                                         const uint32_t* __restrict__ index, const float* __restrict__ value,
                                                         const float* __restrict__ weights, const int* __restrict__ w_pos,
                                                         const int* __restrict__ V_pos, int k, float* __restrict__ pred,
                                                         float* __restrict__ xv_out, const float* __restrict__ label,
                                                         const float* __restrict__ pred_in, float* __restrict__ slope_out) {
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t i = wave; i < nrows; i += nwaves) {
    const uint32_t beg = offset[i], end = offset[i + 1];
    float wsum = 0.f;
    if (pred && lane == 0) {
      for (uint32_t j = beg; j < end; ++j) {  // SpMV::Times, spmv.h:119-132
        const uint32_t u = index[j];
        const int pj = w_pos ? w_pos[u] : (int)u;
        if (pj < 0) continue;
        const float xj = weights[pj];
        if (xj == 0.f) continue;  // spmv.h:125
        wsum += value ? xj * value[j] : xj;
      }
    }
    float s = 0.f;
    if (k > 0) {
      for (int d = lane; d < k; d += 64) {
        float a = 0.f, bb = 0.f;
        for (uint32_t j = beg; j < end; ++j) {  // SpMM::Times x2, spmm.h:105-118
          const int pj = V_pos[index[j]];
          if (pj < 0) continue;
          const float vv = weights[pj + d];
          if (value) {
            const float x = value[j];
            a += vv * x;
            bb += (vv * vv) * (x * x);
          } else {
            a += vv;
            bb += vv * vv;
          }
        }
        if (xv_out) xv_out[(size_t)i * k + d] = a;
        s += a * a - bb;
      }
      s = wave_sum(s);
    }
    if (lane == 0) {
      if (pred) {
        float pr = pred[i] + wsum;  // pred is accumulated into
        if (k > 0) {
          pr += 0.5f * s;
          pr = pr > 20.f ? 20.f : (pr < -20.f ? -20.f : pr);
        }
        pred[i] = pr;
      }
      if (slope_out) {
        const float y = label[i] > 0 ? 1.0f : -1.0f;
        slope_out[i] = -y / (1.0f + expf(y * pred_in[i]));
      }
    }
  }
}

// one wave per column (unique key): grad[w_pos[u]] += X'p ; grad[V_pos[u]+d] += ...
__global__ void __launch_bounds__(256) k_calcgrad_generic(uint32_t ncols, const uint32_t* __restrict__ col_ptr,
                                                          const uint32_t* __restrict__ s_row, const float* __restrict__ s_val,
                                                          const float* __restrict__ weights, const int* __restrict__ w_pos,
                                                          const int* __restrict__ V_pos, int k, const float* __restrict__ slope,
                                                          const float* __restrict__ xv, float* __restrict__ grad) {
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t u = wave; u < ncols; u += nwaves) {
    const uint32_t beg = col_ptr[u], end = col_ptr[u + 1];
    const int pw = w_pos ? w_pos[u] : (int)u;
    const int pv = (k > 0 && V_pos) ? V_pos[u] : -1;
    if (lane == 0) {
      float gw = 0.f;
      for (uint32_t j = beg; j < end; ++j) {  // SpMV::TransTimes, spmv.h:152-168, ascending rows
        const float p = slope[s_row[j]];
        if (p == 0.f) continue;  // spmv.h:155
        gw += s_val ? p * s_val[j] : p;
      }
      if (pw >= 0) grad[pw] += gw;
    }
    if (pv >= 0) {
      float xxp = 0.f;
      // every lane recomputes XXp serially (identical value, reference order)
      for (uint32_t j = beg; j < end; ++j) {
        const float p = slope[s_row[j]];
        if (p == 0.f) continue;  // spmv.h:155
        const float x = s_val ? s_val[j] : 1.0f;
        xxp += p * (x * x);
      }
      for (int d = lane; d < k; d += 64) {
        float g = grad[pv + d] - weights[pv + d] * xxp;  // fm_loss.h:181-188
        for (uint32_t j = beg; j < end; ++j) {           // spmm.h:137-156, ascending rows
          const uint32_t row = s_row[j];
          const float a = xv[(size_t)row * k + d] * slope[row];
          g += s_val ? a * s_val[j] : a;
        }
        grad[pv + d] = g;
      }
    }
  }
}

// Loss::Evaluate over host-provided pred (literal API)
__global__ void k_logloss(const float* __restrict__ label, const float* __restrict__ pred, uint32_t n, double* out) {
  double acc = 0.0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float y = label[i] > 0 ? 1.0f : -1.0f;
    acc += log(1.0 + exp((double)(-y * pred[i])));
  }
  __shared__ double sh[256];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, sh[0]);
}

// ---------------------------------------------------------------------------
// Device Localizer::Compact (src/data/localizer.cc:11-103) around a stable
// radix sort of (key, position) pairs.
// ---------------------------------------------------------------------------
// keys[i] = ReverseBytes(id % max_index), pos[i] = i   (localizer.cc:22-26)
__global__ void k_rdx_keys(const uint64_t* __restrict__ raw, uint32_t nnz, uint64_t max_index,
                           uint64_t* __restrict__ keys, uint32_t* __restrict__ pos) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x) {
    keys[i] = reverse_bytes(raw[i] % max_index);
    pos[i] = i;
  }
}

// head flags of runs of equal keys (localizer.cc:35-48)
__global__ void k_rdx_heads(const uint64_t* __restrict__ skeys, uint32_t nnz, uint32_t* __restrict__ head) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x) {
    head[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1u : 0u;
  }
}

// uid = inclusive_scan(head) - 1.  Emits the dictionary, the segment starts,
// the compact index per nnz (RemapIndex, localizer.cc:63-77) and the
// key-ordered occurrence view (row, value) the backward pass walks.
__global__ void k_rdx_emit(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ spos,
                           const uint32_t* __restrict__ head, const uint32_t* __restrict__ uid_incl, uint32_t nnz,
                           uint32_t nrows, const uint32_t* __restrict__ offset, const float* __restrict__ value,
                           uint64_t* __restrict__ feaids, uint32_t* __restrict__ col_ptr, uint32_t* __restrict__ index,
                           uint32_t* __restrict__ s_row, float* __restrict__ s_val, uint32_t* __restrict__ d_U) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += gridDim.x * blockDim.x) {
    const uint32_t uid = uid_incl[i] - 1;
    const uint32_t pos = spos[i];
    if (head[i]) {
      feaids[uid] = skeys[i];
      col_ptr[uid] = i;
    }
    index[pos] = uid;
    // row of nnz position pos: last r with offset[r] <= pos
    uint32_t lo = 0, hi = nrows;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (offset[mid] <= pos) lo = mid; else hi = mid;
    }
    s_row[i] = lo;
    if (value) s_val[i] = value[pos];
    if (i == nnz - 1) {
      *d_U = uid + 1;
      col_ptr[uid + 1] = nnz;
    }
  }
}

__global__ void k_set_u32(uint32_t* p, uint32_t v) { *p = v; }

// bounds[d] = first unique key of the batch owned by shard d (keys ascending; owner = key / span)
__global__ void k_key_ranges(const uint64_t* __restrict__ feaids, const uint32_t* __restrict__ d_U, int nparts,
                             uint64_t span, uint32_t* __restrict__ bounds) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d > nparts) return;
  const uint32_t U = *d_U;
  if (d == nparts) {
    bounds[d] = U;
    return;
  }
  const uint64_t first = (uint64_t)d * span;  // d < nparts so this does not overflow
  uint32_t lo = 0, hi = U;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (feaids[mid] < first) lo = mid + 1; else hi = mid;
  }
  bounds[d] = lo;
}

// warm start (model preload for benchmarks / resume): one wave per unique key:
// row gets w = w0, fea_cnt = cnt0 and an allocated, hash-initialised V.
__global__ void __launch_bounds__(256) k_warm_start(TableView t, const uint64_t* __restrict__ keys, uint64_t n, float w0,
                                                    float cnt0) {
  const int lane = lane_id();
  const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t u = wave; u < n; u += nwaves) {
    const uint64_t key = keys[u];
    uint32_t r = 0;
    if (lane == 0) r = find_or_insert(t, key);
    r = __shfl(r, 0, 64);
    if (lane == 0) {
      RowHdr h = {};
      h.w = w0;
      h.has_V = t.k > 0 ? 1u : 0u;
      h.sqrt_g = 0.f;
      h.z = 0.f;
      h.fea_cnt = cnt0;
      h.pad[0] = h.pad[1] = h.pad[2] = 0;
      t.hdr[r] = h;
    }
    float* va = t.va + (size_t)r * (2 * t.kp);
    for (int d = lane; d < t.kp; d += 64) {
      va[d] = d < t.k ? hash_init_value(key, d, t.p.seed, t.p.V_init_scale) : 0.0f;
      va[t.kp + d] = 0.0f;
    }
  }
}

// feacnt[u] = segment length (float), for reading the localizer's output back
// the same bounds as int64 in caller-owned device memory (asynchronous sharded step)
// splits != NULL: shard d (d >= 1) starts at key splits[d-1] (ascending) instead of d*span
__global__ void k_key_ranges64(const uint64_t* __restrict__ feaids, const uint32_t* __restrict__ d_U, int nparts,
                               uint64_t span, const uint64_t* __restrict__ splits, int64_t* __restrict__ bounds,
                               uint32_t empty) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d > nparts) return;
  const uint32_t U = empty ? 0u : *d_U;
  if (d == nparts) {
    bounds[d] = U;
    return;
  }
  const uint64_t first = d == 0 ? 0ULL : (splits ? splits[d - 1] : (uint64_t)d * span);
  uint32_t lo = 0, hi = U;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (feaids[mid] < first) lo = mid + 1; else hi = mid;
  }
  bounds[d] = lo;
}

__global__ void k_loc_counts(const uint32_t* __restrict__ col_ptr, const uint32_t* __restrict__ d_U, float* __restrict__ cnt) {
  const uint32_t U = *d_U;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) {
    cnt[u] = (float)(col_ptr[u + 1] - col_ptr[u]);
  }
}

// ---------------------------------------------------------------------------
// BinClassMetric::AUC (src/loss/bin_class_metric.h:35-56): sort by prediction,
// area = sum over negatives of the positives ranked below them.
// k_auc_keys: order-preserving u32 image of the float prediction + label bit.
// k_auc_area: one block walks the sorted labels (n is a minibatch: ~1e4).
// ---------------------------------------------------------------------------
__global__ void k_auc_keys(const float* __restrict__ pred, const float* __restrict__ label, uint32_t n,
                           uint32_t* __restrict__ keys, uint32_t* __restrict__ pos) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t b = __float_as_uint(pred[i] + 0.0f);  // -0 and +0 compare equal in the reference: one image
    keys[i] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    pos[i] = label[i] > 0 ? 1u : 0u;
  }
}

__global__ void __launch_bounds__(1024) k_auc_area(const uint32_t* __restrict__ sorted_pos, uint32_t n, double* __restrict__ out_slot) {
  __shared__ uint32_t wsum[16];
  __shared__ double dsum[16];
  uint32_t carry = 0;  // positives seen so far (identical in every thread)
  double area = 0.0;
  for (uint32_t base = 0; base < n; base += blockDim.x) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t p = i < n ? sorted_pos[i] : 0u;
    uint32_t s = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(s, o, 64);
      if ((int)(threadIdx.x & 63) >= o) s += y;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < (int)(threadIdx.x >> 6)) woff += wsum[w];
      tot += wsum[w];
    }
    if (i < n && p == 0) area += (double)(carry + woff + s);  // cum_tp at a negative
    carry += tot;
  }
  area = wave_sum_d(area);
  if ((threadIdx.x & 63) == 0) dsum[threadIdx.x >> 6] = area;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int w = 0; w < 16; ++w) a += dsum[w];
    const double tp = (double)carry, nn = (double)n;
    double auc_n;
    if (carry == 0 || carry == n) {
      auc_n = 1.0;  // :51 (the reference returns 1, not n)
    } else {
      a /= tp * (nn - tp);
      auc_n = (a < 0.5 ? 1.0 - a : a) * nn;
    }
    *out_slot += auc_n;
  }
}

// table export: one thread per hash slot
__global__ void k_export(TableView t, uint64_t nslots, uint64_t cap, uint64_t* keys, float* scal, int* has_V, float* V,
                         unsigned long long* counter) {
  for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < nslots; s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t key = t.ht[s].key;
    if (key == kEmptyKey) continue;
    const unsigned long long o = atomicAdd(counter, 1ULL);
    if (o >= cap) continue;
    const uint32_t r = t.ht[s].row;
    const RowHdr h = t.hdr[r];
    keys[o] = key;
    scal[o * 4 + 0] = h.fea_cnt;
    scal[o * 4 + 1] = h.w;
    scal[o * 4 + 2] = h.sqrt_g;
    scal[o * 4 + 3] = h.z;
    has_V[o] = h.has_V ? 1 : 0;
    if (t.k > 0) {
      const float* va = t.va + (size_t)r * (2 * t.kp);
      for (int d = 0; d < t.k; ++d) {
        V[o * 2 * t.k + d] = h.has_V ? va[d] : 0.f;
        V[o * 2 * t.k + t.k + d] = h.has_V ? va[t.kp + d] : 0.f;
      }
    }
  }
}

__global__ void k_import(TableView t, uint64_t n, const uint64_t* keys, const float* scal, const int* has_V, const float* V) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t r = find_or_insert(t, keys[i]);
    RowHdr h = {};
    h.fea_cnt = scal[i * 4 + 0];
    h.w = scal[i * 4 + 1];
    h.sqrt_g = scal[i * 4 + 2];
    h.z = scal[i * 4 + 3];
    h.has_V = has_V[i] ? 1u : 0u;
    h.pad[0] = h.pad[1] = h.pad[2] = 0;
    t.hdr[r] = h;
    if (t.k > 0) {
      float* va = t.va + (size_t)r * (2 * t.kp);
      for (int d = 0; d < t.kp; ++d) {
        va[d] = (has_V[i] && d < t.k) ? V[i * 2 * t.k + d] : 0.f;
        va[t.kp + d] = (has_V[i] && d < t.k) ? V[i * 2 * t.k + t.k + d] : 0.f;
      }
    }
  }
}

}  // namespace dfh
