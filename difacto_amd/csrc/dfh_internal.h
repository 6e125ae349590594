// Internal declarations shared by the HIP translation units of libdifacto_hip.
// gfx950 (MI355X / CDNA4) only: 64-lane wavefronts are assumed throughout.
#ifndef DFH_INTERNAL_H_
#define DFH_INTERNAL_H_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "difacto_hip.h"

namespace dfh {

constexpr int kWave = 64;
constexpr uint64_t kEmptyKey = ~0ULL;
constexpr uint32_t kNoRow = ~0u;

// ---------------------------------------------------------------------------
// HBM layout of one model shard (DESIGN.md "Data layout")
//
//   ht   : HEntry[H]      open-addressing index, H = pow2 >= 2*capacity, 16 B/entry
//   hdr  : RowHdr[C]      32 B/row: the scalars of SGDEntry (sgd_updater.h:19-29)
//   va   : float[C][2*kp] per row [V[0..kp) | AdaGrad accumulators[0..kp)],
//                         kp = V_dim rounded up to 4 floats -> 16 B aligned,
//                         128 B-line aligned rows when kp*8 % 128 == 0 (k=64: 512 B)
// ---------------------------------------------------------------------------
struct __attribute__((aligned(16))) HEntry {
  uint64_t key;
  uint32_t row;
  uint32_t pad;
};

struct __attribute__((aligned(32))) RowHdr {
  float w;         // FTRL weight
  uint32_t has_V;  // V allocated (e.V != nullptr)
  float sqrt_g;    // FTRL sqrt of the squared-gradient sum
  float z;         // FTRL z (reference sign convention, sgd_updater.cc:111)
  float fea_cnt;   // feature occurrence count
  uint32_t pad[3];
};
static_assert(sizeof(HEntry) == 16, "HEntry");
static_assert(sizeof(RowHdr) == 32, "RowHdr");

// device-side view of a table, passed to kernels by value
struct TableView {
  HEntry* ht;
  uint64_t hmask;
  RowHdr* hdr;
  float* va;
  uint32_t* nrows;     // device counter of allocated rows
  uint32_t* err;       // device error word (bit0: capacity, bit1: a repeated key inside one source's list (k_resolve_multi), bit2: gradient with V for a row without V, bit3: key-index wait gave up)
  uint32_t* rng_state; // REFRAND: the mutated rand_r seed (sgd_updater.cc:144)
  uint32_t capacity;
  int k;   // V_dim
  int kp;  // padded V_dim (multiple of 4)
  dfh_updater_param p;
};

// the batch's device scalar block d_U[64]: [0] = U, [1] = REFRAND total

// Long-segment lists for the backward pass: the keys whose segments (runs of equal key in the
// key-ordered view) are longer than BWD_SMALL ("mid") / BWD_MID ("hot"), in *nb list buckets:
// bucket q holds cnt entries (values u = rank of the key) at ent[off ..].  The sample-sort
// Localizer fills one list bucket per sort bucket (k_loc_emit: no inter-block compaction);
// k_seg_lists fills a single one.  Order inside and across buckets is arbitrary.  The number of
// list buckets is known to the host when it queues the Localizer and travels as a kernel argument.
// an entry carries the key's rank AND its segment {u, beg, end, 0}: the role that walks a list has the occurrence
// range with the entry's own 16 B load instead of a dependent col_ptr[u], col_ptr[u + 1] round trip (round 4)
typedef uint4 SegEnt;
struct SegLists {
  const uint2* mid;         // [nb] {cnt, off}
  const SegEnt* mid_ent;
  const uint2* hot;         // [nb] {cnt, off}
  const SegEnt* hot_ent;
  const uint2* few;         // [nb] {cnt, off}: keys with 2 .. BWD_SMALL occurrences (k_update_fused only)
  const SegEnt* few_ent;
  // round 6: a key with more than HOT_SPLIT_MIN occurrences (a bias-like feature, a missing-value token: thousands per minibatch) is
  // ALSO listed part by part — {u, beg_p, end_p, p << 16 | nparts}, the parts of a key consecutive — so that k_update_fused can
  // give every part a block of its own (upd_split_role); it stays in the hot list for the consumers that take whole segments
  const SegEnt* split_ent;
  const uint32_t* split_n;  // entries in split_ent (device scalar: reset by the Localizer's count pass, filled by k_lookup_step)
};
#ifndef DFH_HOT_SPLIT
#define DFH_HOT_SPLIT 1024
#endif
#ifndef DFH_HOT_SPLIT_MIN
#define DFH_HOT_SPLIT_MIN 4096
#endif
constexpr uint32_t HOT_SPLIT = DFH_HOT_SPLIT;       // occurrences per part
// ... of the segments LONGER than this.  A key of up to ~4 000 occurrences is 16 tiles per wave of its hot-role block: a chain
// that ends inside the launch; a key in every one of 10 000 rows is 40 tiles per wave and sets the launch's length (120 against
// 72 us).  The crossover, measured on minibatches whose first n slots carry their most popular id in a share f of the rows
// (bench.py --bias-slots n --head-share f; profiles/r06hs_hot_split_thresholds.txt, one box per table): thirteen keys of
// 2 500-3 800 occurrences 81.9 M examples/sec split (thresholds 1 536 / 2 048 / 3 072 alike; parts of 512: 80.2) against 84.9-85.5
// whole; thirteen of 4 000-5 300: 83.5 split against 81.1 whole; 26 of ~5 000: 85.7 against 81.9.  (C3's own largest keys,
// ~1 300 occurrences, split in two cost the default step 2 % in the role's first form, profiles/r06p_*.)
constexpr uint32_t HOT_SPLIT_MIN = DFH_HOT_SPLIT_MIN;
#ifndef DFH_HOT_SPLIT_BUILD
#define DFH_HOT_SPLIT_BUILD 3   // measurement builds: bit 0 the parts are listed (k_lookup_step), bit 1 k_update_fused has the split role
#endif

// flag bits of the row word k_lookup / k_uw_remote leave per unique key in uw[]: a table holds fewer
// than 2^28 rows (dfh_table_create), so the four top bits are free
constexpr uint32_t kRemoteRow = 0x80000000u;  // the row is in the pulled-rows buffer, not in the table (sharded store)
constexpr uint32_t kSingleRow = 0x40000000u;  // the key occurs exactly once in this minibatch
constexpr uint32_t kCountLater = 0x20000000u; // k_lookup left this key's Push(kFeaCount) to the step's update kernel
// the key HAS its V (SGDEntry::V != nullptr as of this step's Pull, sgd_updater.cc:47-52): the forward multiplies a key's V row by
// its feature value iff this bit is set — SpMM::Times skips a key without V (V_pos = -1, spmm.h:108) and does NOT skip an
// allocated row that happens to hold zeros (round 6; until then the forward tested the loaded 16 B slice for all-zero instead,
// which made an allocated all-zero slice times an Inf / NaN value neutral where the reference gives NaN)
constexpr uint32_t kHasV = 0x10000000u;
constexpr uint32_t kRowMask = 0x0FFFFFFFu;

// "row source" seen by the forward / backward kernels: either the table
// itself (rows addressed through urow[u]) or a packed [U x stride] buffer of
// pulled rows (multi-GPU exchange layout, dfh_row_stride()).
struct RowSrc {
  const float* wbase;   // &row0.w ; has_V flag lives at wbase[1] (as u32 or float!=0)
  size_t wstride;       // floats between consecutive rows' w
  const float* vbase;   // &row0.V[0]
  size_t vstride;       // floats between consecutive rows' V
  const uint32_t* urow; // optional indirection u -> row (NULL: identity)
  int flag_is_float;    // packed rows carry has_V as float
};

// device pointers + sizes of a localized minibatch (all arrays live in dfh_batch)
struct BatchView {
  uint32_t nrows;
  uint32_t nnz;
  const uint32_t* d_U;      // number of unique keys (device scalar)
  const uint32_t* offset;   // [nrows+1]
  const uint32_t* index;    // [nnz] compact key rank per nnz (row order)
  const float* value;       // [nnz] or NULL
  const float* label;       // [nrows]
  const uint64_t* feaids;   // [U] ascending reversed keys
  const uint32_t* col_ptr;  // [U+1] segment starts in the key-ordered view
  const uint32_t* s_row;    // [nnz] row of each occurrence, key order (ties: row order)
  const float* s_val;       // [nnz] value of each occurrence, key order, or NULL
  uint32_t* urow;           // [U] table row of each unique key (filled by lookup)
  const uint2* uw;          // [U] {table row, w} per unique key as of this step's k_lookup, else NULL
  float* pred;              // [nrows]
  float* slope;             // [nrows] p_i = -y/(1+exp(y pred))
  float* xv;                // [nrows x kp]
  double* prog;             // [2][PROG_SLOTS] per-block partials: logloss, penalty
  SegLists seg;             // keys with BWD_SMALL < occurrences <= BWD_MID (mid) / more (hot)
};

// ------------------------------------------------------------- error plumbing
void set_error(const std::string& msg);
#define DFH_HIP(call)                                                              \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess) {                                                       \
      ::dfh::set_error(std::string(#call) + ": " + hipGetErrorString(e__));        \
      return DFH_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)
#define DFH_ARG(cond, msg)                 \
  do {                                     \
    if (!(cond)) {                         \
      ::dfh::set_error(msg);               \
      return DFH_ERR_ARG;                  \
    }                                      \
  } while (0)

// ------------------------------------------------------ host+device id helpers
__host__ __device__ inline uint64_t reverse_bytes(uint64_t x) {
  // include/difacto/base.h:39-51 — swaps 32/16/8/4-bit groups: nibble reversal
  x = x << 32 | x >> 32;
  x = (x & 0x0000FFFF0000FFFFULL) << 16 | (x & 0xFFFF0000FFFF0000ULL) >> 16;
  x = (x & 0x00FF00FF00FF00FFULL) << 8 | (x & 0xFF00FF00FF00FF00ULL) >> 8;
  x = (x & 0x0F0F0F0F0F0F0F0FULL) << 4 | (x & 0xF0F0F0F0F0F0F0F0ULL) >> 4;
  return x;
}

__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

}  // namespace dfh
#endif  // DFH_INTERNAL_H_
