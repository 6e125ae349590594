// Device Localizer::Compact (src/data/localizer.cc:11-103) as a hand-written
// sample sort for gfx950.
//
// The reference sorts (ReverseBytes(id % max_index), position) pairs by key
// (localizer.cc:22-29), walks the sorted run to emit the unique keys with their
// counts (:35-48) and maps every nnz to the rank of its key (:63-77).  A
// minibatch is small for a GPU (N ~ 4e5 pairs of 12 B: the whole thing lives
// in L2 / the memory-side cache), so the job is launch- and latency-bound; the
// design minimises dependent passes.  Steady state = FOUR launches:
//
//   k_loc_count    per tile of 2048 pairs: bucket of every pair (binary search over <= 1023
//                  splitters in LDS) + LDS histogram, whose atomic's return value is the pair's
//                  rank inside (tile, bucket); one global atomic per (tile, bucket) reserves the
//                  run's place inside its bucket — no scan pass: the order of the runs inside a
//                  bucket is irrelevant, the bucket is sorted next.  Side job: row of every nnz.
//   k_loc_scatter  bucket starts (block scan of the 1024 totals) + pairs -> bucket-major order.  From here on a pair is
//                  (key, TAG), tag = pos << tb | row mod 2^tb: tags order like positions, and emit gets the row of a
//                  pair back from the tag and the CSR offsets instead of gathering it from a 1.5 MB array (LocView::tb).
//                  A bucket's area is filled XCD group by XCD group (LocView::btotal), so that the short runs of the
//                  tiles running on one XCD merge into whole lines in that XCD's L2.
//   k_loc_sort     one block per bucket: merge sort by (key, pos) (64-wide runs ranked in registers with v_readlane
//                  broadcasts, then log2(n/64) rounds of merge-by-binary-search in LDS); bucket summary
//   k_loc_emit     one block per bucket: unique ids before the bucket from the summaries, then the
//                  Localizer's outputs (dictionary, segment starts, compact index per nnz), the
//                  key-ordered (row, value) view for the backward pass, the long-segment lists of
//                  k_backward_all, and THE SPLITTERS OF THE NEXT CALL: the exact P-quantiles of this
//                  minibatch's sorted (key, pos) order.
//
// Splitters are a property of the data distribution, not of the minibatch: consecutive
// minibatches of one stream are partitioned almost perfectly by the previous one's quantiles
// (buckets of 381 +- ~30 pairs at C3 size), which removes the sampling and sample-ranking passes
// and — more important — the 2-3x oversize buckets that used to set the sort kernel's time.  They
// affect balance only, never the result: any splitters give the same, unique sorted order, and
// a bucket that outgrows the LDS budget is sorted through global memory.  The first call on a
// batch object (or a change of size class) bootstraps them from a jittered sample:
//
//   k_ss_sample    jittered stratified sample of the composite (key, pos)
//   k_ss_rank      all-pairs rank of the samples, tiled over the whole chip
//   k_loc_splitters  every LOC_OVERSAMPLE-th sample becomes a splitter
//
// Splitting on the COMPOSITE (key, pos) makes every element distinct, so buckets stay balanced
// however skewed the key distribution is (a feature present in every row just spreads over
// several buckets); equal keys that straddle a bucket boundary are stitched in k_loc_emit.
// Measured dead ends, kept out: fusing the four passes pairwise (count + scatter around a grid-wide
// barrier; sort + emit with ticketed buckets, the sorted pairs kept in LDS and two tagged hand-offs
// between blocks) — bit-exact, but every hand-off between blocks inside a launch is a device-scope
// round trip across the 8 XCDs and costs more than the launch boundary it replaces (21 + 51 us against
// 19 + 34 us; 66 + 233 us with release / acquire fences, which write back and invalidate the L2);
// a bitonic network (60+ barrier-separated stages at one or two
// waves per SIMD: ~1 us of dependent issue latency each), full O(n^2) ranking of a bucket
// (VALU-bound: 7+ us even when perfectly balanced), an 8-pass LSD radix sort (18 launches).
// The result is the fully sorted pair list (ties by position), i.e. bit-identical to the
// reference's outputs.  Everything is deterministic.
#ifndef DFH_LOCALIZE_HIP_
#define DFH_LOCALIZE_HIP_
#include "dfh_internal.h"

namespace dfh {

constexpr int LOC_OVERSAMPLE = 8;      // bootstrap: samples per bucket
constexpr int LOC_MAX_BUCKETS = 1024;   // the small size class: count / scatter keep 1024 splitters + a histogram in 16 KB of LDS
constexpr int LOC_BIG_BUCKETS = 4096;   // round 4: the large size class (minibatches of 0.7 .. 2.9 M pairs): the same four kernels
                                        // with 4096-bucket LDS tables (64 KB in k_loc_count), in place of the library radix sort
constexpr int LOC_AVG_BUCKET = 384;    // target pairs per bucket
constexpr int LOC_MIN_AVG = 48;        // stored splitters are reused while N / P stays in [MIN, MAX]
constexpr int LOC_MAX_AVG = 700;
constexpr int LOC_LDS_CAP = 1024;      // pairs a bucket may hold to be sorted in LDS
#ifndef DFH_LOC_TILE
#define DFH_LOC_TILE 2048
#endif
#ifndef DFH_LOC_TILE_THREADS
#define DFH_LOC_TILE_THREADS 1024
#endif
constexpr int LOC_TILE = DFH_LOC_TILE;         // pairs per block in count / scatter
constexpr int LOC_TILE_THREADS = DFH_LOC_TILE_THREADS;
constexpr int LOC_PER_THREAD = LOC_TILE / LOC_TILE_THREADS;
constexpr int LOC_SORT_THREADS = 256;
constexpr int LOC_EMIT_THREADS = 256;

struct LocView {
  const uint64_t* raw;    // [N] raw feature ids
  uint32_t n;             // N
  uint64_t max_index;
  int P;                  // buckets
  int bstride;            // row stride of btotal: LOC_MAX_BUCKETS or LOC_BIG_BUCKETS, the size class of this call
  int ntiles;
  int force_global;       // tests: sort every bucket through the global-memory path
  // The 32-bit tie-break carried through the sort is a TAG: pos << tb | (row of pos) mod 2^tb, tb = clz(N - 1) capped
  // at 16.  Tags order like positions (pos is unique and sits in the high bits), and k_loc_emit gets the row back from the
  // tag and `offset` (the rows congruent to the low bits: 2 candidates at C3 size) instead of gathering rowid[pos] from
  // a 1.5 MB array that every XCD's L2 fetches for itself (12.5 MB of fabric reads per minibatch).
  int tb;
  uint32_t nrows;
  const uint32_t* offset; // [nrows + 1] the minibatch's CSR offsets
  const uint32_t* rowid;  // [N] row of every position: written by k_loc_count, read (coalesced) by k_loc_scatter
  // bootstrap
  uint64_t* smp_key;      // [P * LOC_OVERSAMPLE] jittered samples of the composite (key, pos)
  uint32_t* smp_pos;
  uint32_t* smp_rank;     // sorted position of every sample
  // persistent per batch object
  uint64_t* spl_key;      // [P - 1] splitters: bucket b holds composites in [spl[b-1], spl[b])
  uint32_t* spl_pos;
  // per call
  uint32_t* packed;       // [N] bucket << 16 | rank-in-(tile, bucket)
  uint32_t* run_off;      // [ntiles * P] offset of every (tile, bucket) run inside its bucket
  // [LOC_XCDS][LOC_MAX_BUCKETS] pairs per (XCD group of tiles, bucket); zero between calls (k_loc_sort resets it).
  // A bucket's area is filled XCD group by XCD group: the ~2-pair runs of the tiles that run on one XCD land next to each
  // other and merge into whole lines in that XCD's L2 before they are written back (runs of tiles on different XCDs
  // interleaved in arrival order were written back as one partial line each: 14 MB for 4.7 MB of pairs).
  uint32_t* btotal;
  uint32_t* bstart;       // [P + 1]
  uint64_t* bkeys;        // [N] bucket-major keys (unsorted inside a bucket)
  uint32_t* bpos;         // [N] tags
  uint64_t* skeys;        // [N] sorted keys
  uint32_t* spos;         // [N] tags in sorted order
  uint64_t* first_key;    // [P] bucket summaries
  uint64_t* last_key;
  uint32_t* nheads;       // [P] runs of equal keys inside the bucket
  uint32_t* lh;           // [P] local index of the bucket's last run head (0: one run only)
  uint32_t* split_n;      // entries of the split list (SegLists::split_ent): zeroed by the count pass, filled by the step's k_lookup
};

// long-segment lists for k_backward_all, one slot range per list bucket (dfh_internal.h: SegLists)
struct SegListsOut {
  uint2* mid;  // [P] {cnt, off}
  SegEnt* mid_ent;
  uint2* hot;
  SegEnt* hot_ent;
  uint2* few;   // keys with 2 .. BWD_SMALL occurrences (k_update_fused)
  SegEnt* few_ent;
};

// ReverseBytes(id % max_index), localizer.cc:24; the 64-bit modulo is skipped for the
// default max_index = 2^64-1 (x % (2^64-1) is x, except the all-ones id which maps to 0)
__device__ __forceinline__ uint64_t make_key(uint64_t id, uint64_t max_index) {
  const uint64_t m = (max_index == ~0ULL) ? (id == ~0ULL ? 0ULL : id) : id % max_index;
  return reverse_bytes(m);
}

// tiles are dealt to the XCDs round robin (tile number mod 8 — the workgroup id of a launch of its own, and of a rider too:
// riders sit in the launch in groups of 8 blocks); a wrong guess costs write-combining, not correctness
constexpr int LOC_XCDS = 8;

__device__ __forceinline__ bool comp_less(uint64_t ka, uint32_t pa, uint64_t kb, uint32_t pb) {
  return ka < kb || (ka == kb && pa < pb);
}

// block-wide exclusive scan helper: returns the exclusive prefix of `val` over
// the block's threads; *total (optional) receives the block sum
template <int NWAVES>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t val, uint32_t* wsum /* [NWAVES] shared */, uint32_t* total) {
  uint32_t s = val;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(s, o, 64);
    if ((int)(threadIdx.x & 63) >= o) s += y;
  }
  __syncthreads();  // wsum may still be read from a previous call
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) {
    const uint32_t x = wsum[w];
    if (w < (int)(threadIdx.x >> 6)) woff += x;
    tot += x;
  }
  if (total) *total = tot;
  return woff + s - val;
}

// the same for a running maximum: returns max over the threads BEFORE this one (0 if none)
template <int NWAVES>
__device__ __forceinline__ uint32_t block_exclusive_max(uint32_t val, uint32_t* wmax /* [NWAVES] shared */, uint32_t* total) {
  uint32_t s = val;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(s, o, 64);
    if ((int)(threadIdx.x & 63) >= o) s = max(s, y);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 63) wmax[threadIdx.x >> 6] = s;
  __syncthreads();
  uint32_t before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) {
    const uint32_t x = wmax[w];
    if (w < (int)(threadIdx.x >> 6)) before = max(before, x);
    tot = max(tot, x);
  }
  if (total) *total = tot;
  // exclusive inside the wave: the inclusive value of the previous lane
  uint32_t prev = __shfl_up(s, 1, 64);
  if ((threadIdx.x & 63) == 0) prev = 0;
  return max(before, prev);
}

// ---------------------------------------------------------------------------------------
// bootstrap of the splitters (first call on a batch object / new size class)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ss_sample_pos(uint32_t t, uint32_t stride, uint32_t n) {
  const uint32_t j = (uint32_t)(splitmix64(t) >> 33) % stride;
  return min(t * stride + j, n - 1);
}

// the S = LOC_OVERSAMPLE * P jittered samples of the composite (key, pos).  Jittered stratified
// sampling: sample t is a hashed position inside its own stratum [t*stride, (t+1)*stride)
// (distinct by construction).  A plain stride aliases with the row structure — 39 features per
// row, stride 63: only 13 of the 39 slots were ever sampled and whole slots collapsed into one bucket.
__global__ void __launch_bounds__(256) k_ss_sample(LocView v) {
  const uint32_t S = (uint32_t)v.P * LOC_OVERSAMPLE;
  const uint32_t stride = max(1u, v.n / S);
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < S; t += gridDim.x * blockDim.x) {
    const uint32_t i = ss_sample_pos(t, stride, v.n);
    v.smp_key[t] = make_key(v.raw[i], v.max_index);
    v.smp_pos[t] = i << v.tb;  // a tag with row bits 0: splitters only have to be SOME composite values
    v.smp_rank[t] = 0;
  }
}

// ranking of the samples, tiled over the whole chip: block (i, j) compares 256 "row" samples (one
// per thread, in registers) with 256 "column" samples (LDS broadcast reads) and adds the partial
// ranks with one atomic per sample.  The composites are distinct (ties by sample index), so the
// final rank of a sample is its sorted position.
__global__ void __launch_bounds__(256) k_ss_rank(LocView v) {
  __shared__ uint64_t ck[256];
  __shared__ uint32_t cp[256];
  const uint32_t S = (uint32_t)v.P * LOC_OVERSAMPLE;
  const uint32_t ntile = (S + 255) / 256;
  const uint32_t ti = blockIdx.x / ntile, tj = blockIdx.x % ntile;
  const uint32_t col = tj * 256 + threadIdx.x;
  ck[threadIdx.x] = col < S ? v.smp_key[col] : ~0ULL;
  cp[threadIdx.x] = col < S ? v.smp_pos[col] : ~0u;
  const uint32_t row = ti * 256 + threadIdx.x;
  const uint64_t mk = row < S ? v.smp_key[row] : 0;
  const uint32_t mp = row < S ? v.smp_pos[row] : 0;
  __syncthreads();
  const uint32_t lim = min(256u, S - tj * 256);
  uint32_t rank = 0;
#pragma unroll 8
  for (uint32_t j = 0; j < lim; ++j) {
    // small batches sample the same position more than once: equal composites are ordered by sample index
    const bool less = comp_less(ck[j], cp[j], mk, mp) || (ck[j] == mk && cp[j] == mp && tj * 256 + j < row);
    rank += less ? 1u : 0u;
  }
  if (row < S && rank) atomicAdd(&v.smp_rank[row], rank);
}

// sample with rank r = m * LOC_OVERSAMPLE (1 <= m <= P-1) is splitter m-1
__global__ void __launch_bounds__(256) k_loc_splitters(LocView v) {
  const uint32_t S = (uint32_t)v.P * LOC_OVERSAMPLE;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < S; t += gridDim.x * blockDim.x) {
    const uint32_t r = v.smp_rank[t];
    if (r > 0 && r % LOC_OVERSAMPLE == 0) {
      v.spl_key[r / LOC_OVERSAMPLE - 1] = v.smp_key[t];
      v.spl_pos[r / LOC_OVERSAMPLE - 1] = v.smp_pos[t];
    }
  }
}

// ---------------------------------------------------------------------------------------
// count: bucket + rank-in-(tile, bucket) of every pair; the tile's runs reserve their places
// ---------------------------------------------------------------------------------------
// Device feed (dfh_batch_prepare_rows): the minibatch is DESCRIBED — row numbers into device-resident row buffers (the shuffle
// buffers of BatchReader, src/reader/batch_reader.cc:29-78), its own offsets and labels, all in page-locked host memory the
// device reads in place — and k_loc_count gathers it while it counts: a tile takes the rows that cover its 2048 positions
// (the host, which wrote the offsets, also wrote the first row of every tile), stages their {offset, source position} in
// LDS, reads every pair's raw id straight out of the row buffer and leaves it in the minibatch's own array for
// k_loc_scatter.  Until round 6 a launch of its own (k_gather_rows_staged, 16 us at C3 size) opened the preparation chain.
constexpr int LOC_GATHER_ROWS = 1024;   // rows a tile may span (the host checks; beyond: the gather runs as its own launch)
constexpr int LOC_GATHER_SEGS = 4;      // row buffers one minibatch may draw from
struct GatherSrc {
  int nseg;                                // 0: no gather (v.raw holds the minibatch already)
  uint32_t seg_row0[LOC_GATHER_SEGS + 1];  // rows [seg_row0[g], seg_row0[g + 1]) of the minibatch come from buffer g
  const uint32_t* src_off[LOC_GATHER_SEGS];
  const uint64_t* src_idx[LOC_GATHER_SEGS];
  const float* src_val[LOC_GATHER_SEGS];   // NULL: a buffer without values holds ones
  const uint32_t* h_rows;     // [nrows] row numbers inside their buffers   (page-locked host memory, mapped)
  const uint32_t* h_off;      // [nrows + 1] the minibatch's own offsets    (likewise)
  const float* h_lab;         // [nrows]                                    (likewise)
  const uint32_t* h_tile_row; // [ntiles + 1] first row whose range reaches into the tile; [ntiles] = nrows   (likewise)
  uint64_t* dst_raw;          // the minibatch's arrays in HBM
  float* dst_val;             // or NULL (no buffer of the minibatch carries values)
  uint32_t* dst_off;
  float* dst_lab;
};
constexpr size_t loc_gather_smem() { return (size_t)(LOC_GATHER_ROWS + 1) * 4 + 4 + (size_t)LOC_GATHER_ROWS * 8; }
// a[sg] of a small array inside a by-value kernel argument: compile-time indices and selects (a run-time index would make the
// compiler copy the whole argument struct into scratch)
template <typename T>
__device__ __forceinline__ T loc_sel(const T (&a)[LOC_GATHER_SEGS], uint32_t sg) {
  T r = a[0];
#pragma unroll
  for (int x = 1; x < LOC_GATHER_SEGS; ++x) r = sg == (uint32_t)x ? a[x] : r;
  return r;
}

// Every stage below is a BLOCK FUNCTION: block `bid` of `nblk` of its stage, THREADS threads, working memory carved out of
// `smem` — so that a stage can run as a launch of its own (the k_loc_* wrappers) or as a RIDER: a block range of a launch
// that exists anyway (k_lookup / k_forward / k_update_fused of an earlier minibatch's step, dfh_riders.h), which is how the
// single-queue step gets the Localizer of minibatch t+1 done without a second hardware queue.
template <int MAXB>
constexpr size_t loc_count_smem() { return (size_t)MAXB * 16; }
template <int MAXB, int THREADS>
constexpr size_t loc_scatter_smem() { return (size_t)MAXB * 4 + (THREADS / 64) * 4; }
constexpr size_t loc_sort_smem() { return (size_t)LOC_LDS_CAP * 24 + 2 * (LOC_SORT_THREADS / 64) * 4; }
constexpr size_t loc_emit_smem() { return (2 * (LOC_EMIT_THREADS / 64) + 4) * 4; }

template <int MAXB, int THREADS, bool GATHER = false>
__device__ __forceinline__ void loc_count_block(const LocView& v, const uint32_t bid, const uint32_t nblk, char* smem,
                                                const GatherSrc* gp = nullptr) {
  constexpr int PER = LOC_TILE / THREADS;
  static_assert(PER * THREADS == LOC_TILE, "a tile is a whole number of elements per thread");
  uint64_t* sk = reinterpret_cast<uint64_t*>(smem);
  uint32_t* sp = reinterpret_cast<uint32_t*>(smem + (size_t)MAXB * 8);
  uint32_t* hist = sp + MAXB;
  const int P = v.P;
  for (int b = threadIdx.x; b < P; b += THREADS) {
    hist[b] = 0;
    sk[b] = b < P - 1 ? v.spl_key[b] : ~0ULL;  // spl[P-1] = +inf
    sp[b] = b < P - 1 ? v.spl_pos[b] : ~0u;
  }
  const uint32_t base = bid * LOC_TILE;
  if (bid == 0 && threadIdx.x == 0 && v.split_n) *v.split_n = 0u;   // (the step's k_lookup appends to the list)
  uint64_t key[PER];
  if (GATHER) {
    // the rows that cover this tile: their offsets and where they start in their row buffer, in LDS
    const GatherSrc& g = *gp;
    uint32_t* roff = reinterpret_cast<uint32_t*>(smem + loc_count_smem<MAXB>());   // [nr + 1] the rows' offsets in the minibatch
    uint64_t* rsrc = reinterpret_cast<uint64_t*>(roff + LOC_GATHER_ROWS + 2);       // [nr] source position << 2 | buffer
    const uint32_t row0 = g.h_tile_row[bid], row1 = g.h_tile_row[bid + 1];          // rows [row0, row1] reach into the tile
    const uint32_t nr = min(row1, v.nrows - 1u) - row0 + 1u;                        // (the host made sure: <= LOC_GATHER_ROWS)
    for (uint32_t q = threadIdx.x; q <= nr; q += THREADS) roff[q] = g.h_off[row0 + q];
    for (uint32_t q = threadIdx.x; q < nr; q += THREADS) {
      const uint32_t r = row0 + q;
      uint32_t sg = 0;
#pragma unroll
      for (int x = 1; x < LOC_GATHER_SEGS; ++x) sg += (x < g.nseg && r >= g.seg_row0[x]) ? 1u : 0u;
      rsrc[q] = ((uint64_t)loc_sel(g.src_off, sg)[g.h_rows[r]] << 2) | sg;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const uint32_t i = base + e * THREADS + threadIdx.x;
      key[e] = ~0ULL;
      if (i < v.n) {
        // the last of the staged rows that starts at or before i (rows without nonzeros share their successor's offset)
        uint32_t lo = 0, hi = nr - 1u;
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (roff[mid] <= i) lo = mid; else hi = mid - 1u;
        }
        const uint64_t sw = rsrc[lo];
        const uint32_t sg = (uint32_t)(sw & 3u);
        const size_t at = (size_t)(sw >> 2) + (i - roff[lo]);
        const uint64_t id = loc_sel(g.src_idx, sg)[at];
        g.dst_raw[i] = id;   // k_loc_scatter reads the minibatch's own array
        if (g.dst_val) {
          const float* sv = loc_sel(g.src_val, sg);
          g.dst_val[i] = sv ? sv[at] : 1.0f;
        }
        key[e] = make_key(id, v.max_index);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < PER; ++e) {  // independent loads first
      const uint32_t i = base + e * THREADS + threadIdx.x;
      key[e] = i < v.n ? make_key(v.raw[i], v.max_index) : ~0ULL;
    }
  }
  // side job, independent of the sort until k_loc_emit: rowid[pos] = row of nnz position pos,
  // this block's share of the rows
  {
    uint32_t* __restrict__ rowid = const_cast<uint32_t*>(v.rowid);
    // (gathering: the minibatch's offsets and labels arrive in HBM by this pass too — another block's share, so they are read
    // where the host wrote them)
    const uint32_t* __restrict__ offset = GATHER ? gp->h_off : v.offset;
    const uint32_t rpb = (v.nrows + nblk - 1) / nblk;
    const uint32_t r0 = bid * rpb, r1 = min(v.nrows, r0 + rpb);
    for (uint32_t r = r0 + threadIdx.x; r < r1; r += THREADS) {
      const uint32_t b0 = offset[r], e = offset[r + 1];
      if (GATHER) {
        gp->dst_off[r] = b0;
        gp->dst_lab[r] = gp->h_lab[r];
        if (r + 1 == v.nrows) gp->dst_off[r + 1] = e;
      }
      for (uint32_t j = b0; j < e; ++j) rowid[j] = r;
    }
  }
  __syncthreads();
  // PER interleaved binary searches: first b with (key, i) < spl[b]
  int lo[PER], hi[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) { lo[e] = 0; hi[e] = P - 1; }
  for (int step = P; step > 1; step = (step + 1) >> 1) {  // ceil(log2 P) rounds
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      if (lo[e] < hi[e]) {
        const uint32_t i = base + e * THREADS + threadIdx.x;
        const int mid = (lo[e] + hi[e]) >> 1;
        // the pair's own row bits are not known here (another block writes rowid[i]); they only matter for the pair
        // that IS a splitter, and that one may fall on either side of it
        if (comp_less(key[e], i << v.tb, sk[mid], sp[mid])) hi[e] = mid; else lo[e] = mid + 1;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const uint32_t i = base + e * THREADS + threadIdx.x;
    if (i < v.n) {
      const uint32_t r = atomicAdd(&hist[lo[e]], 1u);
      v.packed[i] = ((uint32_t)lo[e] << 16) | r;
    }
  }
  __syncthreads();
  // one global atomic per non-empty (tile, bucket) run: its offset inside the bucket.  The order in
  // which the tiles arrive differs from run to run; the bucket is sorted afterwards, so the
  // result does not.
  for (int b = threadIdx.x; b < P; b += THREADS) {
    const uint32_t h = hist[b];
    v.run_off[bid * P + b] = h ? atomicAdd(&v.btotal[(bid & (LOC_XCDS - 1)) * MAXB + b], h) : 0u;
  }
}

template <int MAXB>
__global__ void __launch_bounds__(LOC_TILE_THREADS) k_loc_count(LocView v) {
  __shared__ __attribute__((aligned(16))) char smem[loc_count_smem<MAXB>()];
  loc_count_block<MAXB, LOC_TILE_THREADS>(v, blockIdx.x, gridDim.x, smem);
}
template <int MAXB>
__global__ void __launch_bounds__(LOC_TILE_THREADS) k_loc_count_gather(LocView v, GatherSrc g) {
  __shared__ __attribute__((aligned(16))) char smem[loc_count_smem<MAXB>() + loc_gather_smem()];
  loc_count_block<MAXB, LOC_TILE_THREADS, true>(v, blockIdx.x, gridDim.x, smem, &g);
}


// ---- scatter into bucket-major order; every block derives the bucket starts from the totals
// (block 0 publishes them)
template <int MAXB, int THREADS>
__device__ __forceinline__ void loc_scatter_block(const LocView& v, const uint32_t bid, char* smem) {
  constexpr int PER = LOC_TILE / THREADS;
  constexpr int LOC_BPT = MAXB / THREADS;  // buckets per thread in the scan of the bucket totals
  static_assert(LOC_BPT * THREADS == MAXB, "the scan of the bucket totals gives every thread the same number of buckets");
  uint32_t* off = reinterpret_cast<uint32_t*>(smem);
  uint32_t* wsum = off + MAXB;
  const int P = v.P;
  const uint32_t xcd = bid & (LOC_XCDS - 1);
  const uint32_t base = bid * LOC_TILE;
  uint64_t raw[PER];
  uint32_t pk[PER], rw[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const uint32_t i = min(base + e * THREADS + threadIdx.x, v.n - 1);  // clamped: the loads stay unconditional
    raw[e] = v.raw[i];
    pk[e] = v.packed[i];
    rw[e] = v.rowid[i];
  }
  // exclusive scan of the bucket totals: LOC_BPT consecutive buckets per thread
  const int b0 = threadIdx.x * LOC_BPT;
  uint32_t tt[LOC_BPT], ro[LOC_BPT];
  uint32_t sum = 0;
#pragma unroll
  for (int q = 0; q < LOC_BPT; ++q) {
    // loads on a clamped bucket, masked afterwards: nine independent loads in flight instead of nine guarded ones
    const int bb = min(b0 + q, P - 1);
    const uint32_t in = b0 + q < P ? 1u : 0u;
    tt[q] = 0;
    ro[q] = v.run_off[bid * P + bb];
#pragma unroll
    for (int x = 0; x < LOC_XCDS; ++x) {
      const uint32_t h = v.btotal[x * MAXB + bb];
      tt[q] += h;
      ro[q] += (x < (int)xcd ? 1u : 0u) * h;  // the groups before this tile's
    }
    tt[q] *= in;
    ro[q] *= in;
    sum += tt[q];
  }
  uint32_t total;
  uint32_t ex = block_exclusive_scan<THREADS / 64>(sum, wsum, &total);
#pragma unroll
  for (int q = 0; q < LOC_BPT; ++q) {
    if (b0 + q < P) {
      off[b0 + q] = ex + ro[q];
      if (bid == 0) v.bstart[b0 + q] = ex;
    }
    ex += tt[q];
  }
  if (bid == 0 && threadIdx.x == 0) v.bstart[P] = total;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const uint32_t i = base + e * THREADS + threadIdx.x;
    if (i < v.n) {
      const uint32_t dst = off[pk[e] >> 16] + (pk[e] & 0xFFFFu);
      v.bkeys[dst] = make_key(raw[e], v.max_index);
      v.bpos[dst] = (i << v.tb) | (rw[e] & ((1u << v.tb) - 1u));
    }
  }
}

template <int MAXB>
__global__ void __launch_bounds__(LOC_TILE_THREADS) k_loc_scatter(LocView v) {
  __shared__ __attribute__((aligned(16))) char smem[loc_scatter_smem<MAXB, LOC_TILE_THREADS>()];
  loc_scatter_block<MAXB, LOC_TILE_THREADS>(v, blockIdx.x, smem);
}

// ---- sort of one bucket [beg, beg + n) of the bucket-major arrays by (key, pos).  Buckets of up to
// LOC_LDS_CAP pairs are sorted in LDS and STAY there (*sk / *sp point into ak/ap or bk/bp); larger
// ones (stale or unlucky splitters; kept for correctness and bounded at n log n) go through the same
// rounds on the global arrays and end up in v.skeys / v.spos.  All threads of the block call it.
__device__ __forceinline__ bool loc_sort_bucket(const LocView& v, uint32_t beg, uint32_t n, uint64_t* ak, uint32_t* ap,
                                                uint64_t* bk, uint32_t* bp, const uint64_t** sk_out, const uint32_t** sp_out) {
  const uint64_t* gk = v.bkeys + beg;
  const uint32_t* gp = v.bpos + beg;
  const bool in_lds = n <= LOC_LDS_CAP && !v.force_global;
  uint64_t* sk = bk;  // where the sorted bucket ends up (LDS path)
  uint32_t* sp = bp;
  if (in_lds) {
    // runs of 64: a wave ranks its 64 pairs against each other in registers — lane j's pair is broadcast with v_readlane, 64
    // unrolled compares, no LDS round trip per compare (round 5: the LDS form waited out one ds_read latency per compare,
    // 3.5 us per run) — and writes the sorted run to LDS
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t c0 = (threadIdx.x >> 6) * 64; c0 < n; c0 += LOC_SORT_THREADS) {  // wave-uniform: every lane is active below
      const uint32_t idx = c0 + lane;
      const bool valid = idx < n;
      const uint64_t mk = valid ? gk[idx] : ~0ULL;  // the padding pair is less than nothing
      const uint32_t mp = valid ? gp[idx] : ~0u;
      const int klo = (int)(uint32_t)mk, khi = (int)(uint32_t)(mk >> 32);
      uint32_t rank = 0;
#pragma unroll 8  // (fully unrolled, the compiler reads all 192 words first and spills them)
      for (int j = 0; j < 64; ++j) {
        const uint64_t ok = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(khi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane(klo, j);
        const uint32_t op = (uint32_t)__builtin_amdgcn_readlane((int)mp, j);
        rank += comp_less(ok, op, mk, mp) ? 1u : 0u;
      }
      if (valid) {
        bk[c0 + rank] = mk;
        bp[c0 + rank] = mp;
      }
    }
    __syncthreads();
    // merge rounds: element's slot = its offset in its run + (# smaller elements in the sibling run).  A thread carries its
    // (up to) LOC_LDS_CAP / LOC_SORT_THREADS elements through the search together: one LDS latency per step, not one per
    // element and step.  The count is built from descending powers of two (the sibling run has at most L elements).
    constexpr int E = LOC_LDS_CAP / LOC_SORT_THREADS;
    static_assert(E * LOC_SORT_THREADS == LOC_LDS_CAP && LOC_SORT_THREADS % 64 == 0, "a thread carries E elements of a full bucket");
    uint64_t* dk = ak;
    uint32_t* dp = ap;
    uint32_t sh = 6;
    for (uint32_t L = 64; L < n; L <<= 1, ++sh) {
      uint64_t mk[E];
      uint32_t mp[E], cnt[E], sb[E], len[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t idx = threadIdx.x + e * LOC_SORT_THREADS;
        const bool valid = idx < n;
        len[e] = 0;
        cnt[e] = 0;
        if (e * LOC_SORT_THREADS >= n) continue;  // block-uniform: a bucket of n <= 256 e costs e elements' work
        mk[e] = valid ? sk[idx] : 0ULL;
        mp[e] = valid ? sp[idx] : 0u;
        const uint32_t sib = ((idx >> sh) ^ 1u) << sh;
        sb[e] = min(sib, n);
        len[e] = valid ? min(sib + L, n) - sb[e] : 0u;
        cnt[e] = 0;
      }
      for (uint32_t step = L; step; step >>= 1) {
        uint64_t ok[E];
        uint32_t op[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (e * LOC_SORT_THREADS >= n) continue;
          const uint32_t t = cnt[e] + step;
          const uint32_t at = t <= len[e] ? sb[e] + t - 1u : 0u;  // slot 0 when there is nothing to test: always readable
          ok[e] = sk[at];
          op[e] = sp[at];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (e * LOC_SORT_THREADS >= n) continue;
          const uint32_t t = cnt[e] + step;
          if (t <= len[e] && comp_less(ok[e], op[e], mk[e], mp[e])) cnt[e] = t;
        }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t idx = threadIdx.x + e * LOC_SORT_THREADS;
        if (idx < n) {
          const uint32_t r = idx >> sh;
          const uint32_t dst = ((r & ~1u) << sh) + (idx - (r << sh)) + cnt[e];
          dk[dst] = mk[e];
          dp[dst] = mp[e];
        }
      }
      __syncthreads();
      uint64_t* tk = sk; sk = dk; dk = tk;
      uint32_t* tp = sp; sp = dp; dp = tp;
    }
    *sk_out = sk;
    *sp_out = sp;
  } else {
    // oversize bucket (stale or unlucky splitters; kept for correctness and bounded at n log n):
    // the same run-ranking + merge rounds on the global arrays, ping-ponging between the
    // bucket-major and the sorted buffers
    uint64_t* xk = v.bkeys + beg;
    uint32_t* xp = v.bpos + beg;
    uint64_t* yk = v.skeys + beg;
    uint32_t* yp = v.spos + beg;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t c0 = (threadIdx.x >> 6) * 64; c0 < n; c0 += LOC_SORT_THREADS) {
      const uint32_t idx = c0 + lane;
      const bool valid = idx < n;
      const uint64_t mk = valid ? xk[idx] : ~0ULL;
      const uint32_t mp = valid ? xp[idx] : ~0u;
      const uint32_t lim = min(64u, n - c0);
      uint32_t rank = 0;
      for (uint32_t j = 0; j < lim; ++j) rank += comp_less(xk[c0 + j], xp[c0 + j], mk, mp) ? 1u : 0u;
      if (valid) {
        yk[c0 + rank] = mk;
        yp[c0 + rank] = mp;
      }
    }
    __syncthreads();  // block-scope visibility of global stores
    uint64_t* srck = yk; uint32_t* srcp = yp;
    uint64_t* dstk = xk; uint32_t* dstp = xp;
    for (uint32_t L = 64; L < n; L <<= 1) {
      for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const uint64_t mk = srck[idx];
        const uint32_t mp = srcp[idx];
        const uint32_t r = idx / L;
        const uint32_t pair_base = (r & ~1u) * L;
        const uint32_t sib = (r ^ 1u) * L;
        uint32_t lo = min(sib, n), hi = min(sib + L, n);
        const uint32_t sib_beg = lo;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (comp_less(srck[mid], srcp[mid], mk, mp)) lo = mid + 1; else hi = mid;
        }
        const uint32_t dst = pair_base + (idx - r * L) + (lo - sib_beg);
        dstk[dst] = mk;
        dstp[dst] = mp;
      }
      __syncthreads();
      uint64_t* tk = srck; srck = dstk; dstk = tk;
      uint32_t* tp = srcp; srcp = dstp; dstp = tp;
    }
    if (srck != yk) {  // result must live in skeys/spos
      for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
        yk[t] = srck[t];
        yp[t] = srcp[t];
      }
    }
    __syncthreads();
  }
  if (!in_lds) {
    *sk_out = v.skeys + beg;
    *sp_out = v.spos + beg;
  }
  return in_lds;
}

// bucket summary: {runs of equal keys, local index of the last run head, first key, last key}; valid in
// thread 0 only.  red: [2][LOC_SORT_THREADS / 64] shared words.
struct BucketSummary {
  uint32_t nheads, lh;
  uint64_t first_key, last_key;
};
__device__ __forceinline__ BucketSummary loc_bucket_summary(const uint64_t* sk, uint32_t n, uint32_t (*red)[LOC_SORT_THREADS / 64]) {
  uint32_t cnt = 0, last = 0;
  for (uint32_t t = 1 + threadIdx.x; t < n; t += blockDim.x) {
    if (sk[t] != sk[t - 1]) {
      ++cnt;
      last = t;  // ascending t per thread
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    last = max(last, (uint32_t)__shfl_xor(last, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = cnt;
    red[1][threadIdx.x >> 6] = last;
  }
  __syncthreads();
  BucketSummary r;
  r.nheads = 1;  // position 0 opens the first run
  r.lh = 0;
  r.first_key = 0;
  r.last_key = 0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < LOC_SORT_THREADS / 64; ++w) {
      r.nheads += red[0][w];
      r.lh = max(r.lh, red[1][w]);
    }
    r.first_key = sk[0];
    r.last_key = sk[n - 1];
  }
  return r;
}

// ---- sort one bucket; summary of its runs of equal keys (four-launch form).  Block `bid` of `nblk`: the grid may be
// smaller than the number of buckets (a capped grid leaves more of the chip to the training step this work runs beside)
__device__ __forceinline__ void loc_sort_block(const LocView& v, const uint32_t bid, const uint32_t nblk, char* smem) {
  uint64_t* ak = reinterpret_cast<uint64_t*>(smem);
  uint64_t* bk = ak + LOC_LDS_CAP;
  uint32_t* ap = reinterpret_cast<uint32_t*>(bk + LOC_LDS_CAP);
  uint32_t* bp = ap + LOC_LDS_CAP;
  uint32_t (*red)[LOC_SORT_THREADS / 64] = reinterpret_cast<uint32_t (*)[LOC_SORT_THREADS / 64]>(bp + LOC_LDS_CAP);
  for (uint32_t b = bid; b < (uint32_t)v.P; b += nblk) {
    __syncthreads();  // LDS of the previous bucket is done with
    const uint32_t beg = v.bstart[b], end = v.bstart[b + 1];
    const uint32_t n = end - beg;
    if (threadIdx.x < LOC_XCDS) v.btotal[threadIdx.x * v.bstride + b] = 0;  // consumed by k_loc_scatter: ready for the next call
    if (n == 0) {
      if (threadIdx.x == 0) {
        v.nheads[b] = 0;
        v.lh[b] = 0;
        v.first_key[b] = 0;
        v.last_key[b] = 0;
      }
      continue;
    }
    const uint64_t* sk;
    const uint32_t* sp;
    if (loc_sort_bucket(v, beg, n, ak, ap, bk, bp, &sk, &sp)) {
      for (uint32_t t = threadIdx.x; t < n; t += LOC_SORT_THREADS) {
        v.skeys[beg + t] = sk[t];
        v.spos[beg + t] = sp[t];
      }
    }
    const BucketSummary sm = loc_bucket_summary(sk, n, red);
    if (threadIdx.x == 0) {
      v.nheads[b] = sm.nheads;
      v.lh[b] = sm.lh;
      v.first_key[b] = sm.first_key;
      v.last_key[b] = sm.last_key;
    }
  }
}
__global__ void __launch_bounds__(LOC_SORT_THREADS) k_loc_sort(LocView v) {
  __shared__ __attribute__((aligned(16))) char smem[loc_sort_smem()];
  loc_sort_block(v, blockIdx.x, gridDim.x, smem);
}

// ---- emit: one block per bucket stitches itself to its predecessors (unique keys before the
// bucket; does its first key continue the previous bucket's last key; where did the run that is
// open at the bucket's start begin) and writes the Localizer's outputs.
//
// Segment lengths — needed to sort the keys into the roles of k_backward_all / k_update_fused
// (one occurrence: no list; 2 .. BWD_SMALL: "few"; .. BWD_MID: "mid"; more: "hot") — are known where a segment ENDS: the head of the next run (or the end of the
// minibatch) closes it, and the closing thread knows the start from the running maximum of head
// positions.  Every bucket lists the long segments it closes in its own slot range (at most
// n_b / 2 + 2 few, n_b / (BWD_SMALL + 1) + 2 mid and n_b / (BWD_MID + 1) + 2 hot ones: no atomics between
// blocks, no compaction pass).
// the writing half of emit for one bucket whose sorted pairs are sk / sp [0, n) (LDS or global):
// `cont`: the bucket's first key continues the previous non-empty bucket's last key; `ubase`: unique
// keys before the bucket; `carry1`: position + 1 of the last run head before the bucket (0: none).
// n_mid / n_hot: shared counters, zeroed by the caller before its last barrier.
template <int NW, bool PROBE>
__device__ __forceinline__ void loc_emit_bucket(const LocView& v, uint32_t b, uint32_t beg, uint32_t n, const uint64_t* sk,
                                                const uint32_t* sp, uint32_t cont, uint32_t ubase, uint32_t carry1,
                                                const float* __restrict__ value,
                                                uint64_t* __restrict__ feaids, uint32_t* __restrict__ col_ptr,
                                                uint32_t* __restrict__ index, uint32_t* __restrict__ s_row,
                                                float* __restrict__ s_val, uint32_t* __restrict__ d_U, const SegListsOut& sl,
                                                uint32_t* wsum, uint32_t* wmax, uint32_t* n_mid, uint32_t* n_hot,
                                                uint32_t* n_few, const TableView& tab, uint32_t* __restrict__ urow) {
  const uint32_t P = (uint32_t)v.P;
  const uint32_t moff = beg / (BWD_SMALL + 1) + 2 * b, hoff = beg / (BWD_MID + 1) + 2 * b, foff = beg / 2 + 2 * b;
  uint32_t run_heads = 0, run_max1 = carry1;
  for (uint32_t base = 0; base < n; base += blockDim.x) {
    const uint32_t t = base + threadIdx.x;
    const uint32_t i = beg + t;
    const bool valid = t < n;
    uint64_t key = 0;
    uint32_t tag = 0;
    bool head = false;
    if (valid) {
      key = sk[t];
      tag = sp[t];
      head = t == 0 ? !cont : key != sk[t - 1];
    }
    const uint32_t pos = tag >> v.tb;
    uint32_t nh, mx;
    const uint32_t ex = block_exclusive_scan<NW>(head ? 1u : 0u, wsum, &nh);
    const uint32_t pm = block_exclusive_max<NW>(head ? i + 1 : 0u, wmax, &mx);
    if (valid) {
      const uint32_t incl = run_heads + ex + (head ? 1u : 0u);
      const uint32_t uid = ubase + incl - 1;          // cont && no head yet: the previous bucket's last id
      const uint32_t prev1 = max(run_max1, pm);       // start + 1 of the run open just before this element
      if (head) {
        feaids[uid] = key;
        col_ptr[uid] = i;
        // the key-index probe of the step (dfh_localize_lookup): the thread that writes a unique key has it in hand
        if (PROBE) urow[uid] = find_or_insert(tab, key);
        if (prev1) {  // closes segment uid - 1 = [prev1 - 1, i)
          const uint32_t len = i - (prev1 - 1);
          const SegEnt e = make_uint4(uid - 1, prev1 - 1, i, 0u);
          if (len > BWD_MID) sl.hot_ent[hoff + atomicAdd(n_hot, 1u)] = e;
          else if (len > BWD_SMALL) sl.mid_ent[moff + atomicAdd(n_mid, 1u)] = e;
          else if (len > 1) sl.few_ent[foff + atomicAdd(n_few, 1u)] = e;
        }
      }
      index[pos] = uid;  // RemapIndex, localizer.cc:63-77
      // the row of pos: the largest row congruent to the tag's low bits whose range starts at or before pos
      uint32_t row = tag & ((1u << v.tb) - 1u);
      {
        // candidates row + k 2^tb, k = 0 .. kmax: their offsets ascend, the row is the last one that starts at or before pos
        // (2 candidates at C3 size: the loop ends at once; the large size class has up to ~70: a binary search)
        const uint32_t step = 1u << v.tb;
        uint32_t lo = 0, hi = v.nrows > row ? (v.nrows - 1u - row) >> v.tb : 0u;  // k in [lo, hi]
        while (lo < hi) {
          const uint32_t mid = (lo + hi + 1u) >> 1;
          if (v.offset[row + mid * step] <= pos) lo = mid; else hi = mid - 1u;
        }
        row += lo * step;
      }
      s_row[i] = row;
      if (value) s_val[i] = value[pos];
      if (i == v.n - 1) {  // the end of the minibatch closes the last segment
        *d_U = uid + 1;
        col_ptr[uid + 1] = v.n;
        const uint32_t lbeg = head ? i : prev1 - 1, len = v.n - lbeg;
        const SegEnt e = make_uint4(uid, lbeg, v.n, 0u);
        if (len > BWD_MID) sl.hot_ent[hoff + atomicAdd(n_hot, 1u)] = e;
        else if (len > BWD_SMALL) sl.mid_ent[moff + atomicAdd(n_mid, 1u)] = e;
        else if (len > 1) sl.few_ent[foff + atomicAdd(n_few, 1u)] = e;
      }
      // the splitters of the next call: the exact P-quantiles of this sorted order
      if (P > 1) {
        const uint64_t m = ((uint64_t)i * P + v.n - 1) / v.n;  // smallest m with m * n / P >= i
        if (m >= 1 && m <= P - 1 && (m * v.n) / P == i) {
          v.spl_key[m - 1] = key;
          v.spl_pos[m - 1] = tag;
        }
      }
    }
    run_heads += nh;
    run_max1 = max(run_max1, mx);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sl.mid[b] = make_uint2(*n_mid, moff);
    sl.hot[b] = make_uint2(*n_hot, hoff);
    sl.few[b] = make_uint2(*n_few, foff);
  }
}

// what emit writes, apart from the next call's splitters (LocView)
struct EmitOut {
  const float* value;    // [N] feature values in row order, or NULL
  uint64_t* feaids;
  uint32_t* col_ptr;
  uint32_t* index;
  uint32_t* s_row;
  float* s_val;
  uint32_t* d_U;
  SegListsOut sl;
};

template <bool PROBE>
__device__ __forceinline__ void loc_emit_block(const LocView& v, const EmitOut& o, const TableView& tab, uint32_t* __restrict__ urow,
                                               const uint32_t bid, const uint32_t nblk, char* smem) {
  constexpr int NW = LOC_EMIT_THREADS / 64;
  uint32_t* wsum = reinterpret_cast<uint32_t*>(smem);
  uint32_t* wmax = wsum + NW;
  uint32_t& sh_cont = wmax[NW];
  uint32_t& n_mid = wmax[NW + 1];
  uint32_t& n_hot = wmax[NW + 2];
  uint32_t& n_few = wmax[NW + 3];
  const SegListsOut& sl = o.sl;
  const uint32_t P = (uint32_t)v.P;
  for (uint32_t b = bid; b < P; b += nblk) {
    __syncthreads();  // the shared counters of the previous bucket have been published
    const uint32_t beg = v.bstart[b], end = v.bstart[b + 1];
    const uint32_t n = end - beg;
    if (threadIdx.x == 0) {
      n_mid = 0;
      n_hot = 0;
      n_few = 0;
      if (n == 0) {
        sl.mid[b] = make_uint2(0u, 0u);
        sl.hot[b] = make_uint2(0u, 0u);
        sl.few[b] = make_uint2(0u, 0u);
      }
    }
    if (n == 0) continue;
    // over the buckets q < b: unique keys (nheads[q] - cont[q], cont[q]: first_key[q] equals the last key of the
    // previous NON-EMPTY bucket) and the position of the last run head that opens a new key
    uint32_t part = 0, carry1 = 0;  // carry1: position + 1 (0: none)
    for (uint32_t q0 = threadIdx.x; q0 <= b; q0 += LOC_EMIT_THREADS) {
      const uint32_t bq = v.bstart[q0], nq = v.bstart[q0 + 1] - bq;
      if (nq == 0) continue;
      int p = (int)q0 - 1;
      while (p >= 0 && v.bstart[p + 1] == v.bstart[p]) --p;
      const uint32_t c = (p >= 0 && v.first_key[q0] == v.last_key[p]) ? 1u : 0u;
      if (q0 == b) {
        sh_cont = c;
      } else {
        part += v.nheads[q0] - c;
        const uint32_t l = v.lh[q0];
        if (l > 0) carry1 = max(carry1, bq + l + 1);
        else if (!c) carry1 = max(carry1, bq + 1);
      }
    }
    uint32_t ubase, carry_all;
    block_exclusive_scan<NW>(part, wsum, &ubase);
    block_exclusive_max<NW>(carry1, wmax, &carry_all);
    __syncthreads();
    loc_emit_bucket<NW, PROBE>(v, b, beg, n, v.skeys + beg, v.spos + beg, sh_cont, ubase, carry_all, o.value, o.feaids, o.col_ptr,
                               o.index, o.s_row, o.s_val, o.d_U, sl, wsum, wmax, &n_mid, &n_hot, &n_few, tab, urow);
  }
}

template <bool PROBE>
__global__ void __launch_bounds__(LOC_EMIT_THREADS) k_loc_emit(LocView v, EmitOut o, TableView tab, uint32_t* __restrict__ urow) {
  __shared__ __attribute__((aligned(16))) char smem[loc_emit_smem()];
  loc_emit_block<PROBE>(v, o, tab, urow, blockIdx.x, gridDim.x, smem);
}

}  // namespace dfh
#endif  // DFH_LOCALIZE_HIP_
