// Device Localizer::Compact (src/data/localizer.cc:11-103) as a hand-written
// sample sort for gfx950.
//
// The reference sorts (ReverseBytes(id % max), position) pairs by key
// (localizer.cc:22-29), walks the sorted run to emit the unique keys with their
// counts (:35-48) and maps every nnz to the rank of its key (:63-77).  A
// minibatch is small for a GPU (N ~ 4e5 pairs of 12 B: the whole thing lives
// in L2), so an 8-pass LSD radix sort is launch- and latency-bound (9 x 2
// launches, ~110 us measured with rocPRIM).  Instead, one partition pass and
// one local pass:
//
//   k_ss_sample     jittered samples of the composite (key, pos); also the row of every nnz
//   k_ss_rank       rank of every sample among all samples, tiled over the whole
//                   chip: every SS_OVERSAMPLE-th becomes a splitter
//   k_ss_count      per tile: bucket of every pair (binary search over the
//                   splitters in LDS) + LDS histogram; the LDS atomic's return
//                   value is the pair's rank inside (tile, bucket)
//   k_ss_scan       per bucket: prefix over tiles, bucket totals
//   k_ss_scatter    pairs -> bucket-major order
//   k_ss_sort       one block per bucket: merge sort in LDS by (key, pos)
//                   (64-wide runs by ranking, then log2(n/64) rounds of
//                   merge-by-binary-search), run heads + local unique ranks
//   k_ss_emit       stitch the buckets (unique keys before each one); dictionary, segment starts, compact index per nnz and the
//                   key-ordered (row, value) view for the backward pass
//
// Splitting on the COMPOSITE (key, pos) makes every element distinct, so
// buckets stay balanced however skewed the key distribution is (a feature
// present in every row just spreads over several buckets); equal keys that
// straddle a bucket boundary are stitched in k_ss_emit.  Measured
// dead ends, kept out: a bitonic network (60+ barrier-separated stages at one
// or two waves per SIMD: ~1 us of dependent issue latency each, 50-65 us per
// bucket pass) and full O(n^2) ranking of a bucket (the largest bucket sets the
// kernel time: 80-175 us).  Merge-by-binary-search needs log2(n/64) barriers.
// The result is the fully sorted pair list (ties by position), i.e.
// bit-identical to the reference's outputs.  Everything is deterministic.
#ifndef DFH_LOCALIZE_HIP_
#define DFH_LOCALIZE_HIP_
#include "dfh_internal.h"

namespace dfh {

constexpr int SS_OVERSAMPLE = 8;      // samples per bucket
constexpr int SS_MAX_BUCKETS = 1024;   // => the sample sort serves batches up to 524 k pairs
constexpr int SS_AVG_BUCKET = 512;    // target pairs per bucket
constexpr int SS_LDS_CAP = 2048;      // pairs a bucket may hold to be sorted in LDS (4x the mean)
constexpr int SS_TILE = 4096;         // pairs per block in count / scatter
constexpr int SS_TILE_THREADS = 1024;
constexpr int SS_PER_THREAD = SS_TILE / SS_TILE_THREADS;
constexpr int SS_SORT_THREADS = 512;
constexpr int SS_SCAN_BUCKETS = 64;   // buckets per block in k_ss_scan

struct SSView {
  const uint64_t* raw;    // [N] raw feature ids
  uint32_t n;             // N
  uint64_t max_index;
  int P;                  // buckets
  int ntiles;
  int force_global;       // tests: sort every bucket through the global-memory path
  uint64_t* smp_key;      // [P * SS_OVERSAMPLE] jittered samples of the composite (key, pos)
  uint32_t* smp_pos;
  uint32_t* smp_rank;     // [P * SS_OVERSAMPLE] sorted position of every sample
  uint64_t* spl_key;      // [P] splitters (bucket b holds composites in [spl[b-1], spl[b]) )
  uint32_t* spl_pos;
  uint32_t* packed;       // [N] bucket << 16 | rank-in-(tile,bucket)
  uint32_t* hist;         // [ntiles * P]
  uint32_t* run_off;      // [ntiles * P] offset of every (tile, bucket) run inside its bucket
  uint32_t* btotal;       // [P] pairs per bucket
  uint32_t* bstart;       // [P + 1]
  uint64_t* bkeys;        // [N] bucket-major keys (unsorted inside a bucket)
  uint32_t* bpos;         // [N]
  uint64_t* skeys;        // [N] sorted keys
  uint32_t* spos;         // [N] sorted positions
  uint32_t* luid;         // [N] 1-based unique rank inside the bucket
  uint32_t* head;         // [N] run head inside the bucket
  uint64_t* first_key;    // [P] bucket meta
  uint64_t* last_key;
  uint32_t* nheads;
  uint32_t* ubase;        // [P] unique keys before the bucket
  uint32_t* cont;         // [P] bucket's first key continues the previous bucket's last key
};

// ReverseBytes(id % max_index), localizer.cc:24; the 64-bit modulo is skipped for the
// default max_index = 2^64-1 (x % (2^64-1) is x, except the all-ones id which maps to 0)
__device__ __forceinline__ uint64_t make_key(uint64_t id, uint64_t max_index) {
  const uint64_t m = (max_index == ~0ULL) ? (id == ~0ULL ? 0ULL : id) : id % max_index;
  return reverse_bytes(m);
}

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

__device__ __forceinline__ uint32_t ss_sample_pos(uint32_t t, uint32_t stride, uint32_t n) {
  const uint32_t j = (uint32_t)(splitmix64(t) >> 33) % stride;
  return min(t * stride + j, n - 1);
}

// ---- sampling: the S = SS_OVERSAMPLE * P jittered samples of the composite (key, pos);
// side job, independent of the sort until k_ss_emit: rowid[pos] = row of nnz position pos
__global__ void __launch_bounds__(256) k_ss_sample(SSView v, uint32_t nrows, const uint32_t* __restrict__ offset,
                                                   uint32_t* __restrict__ rowid) {
  const uint32_t S = (uint32_t)v.P * SS_OVERSAMPLE;
  // jittered stratified sampling: sample t is a hashed position inside its own stratum
  // [t*stride, (t+1)*stride) (distinct by construction).  A plain stride aliases with the
  // row structure — 39 features per row, stride 63: only 13 of the 39 slots were ever
  // sampled and whole slots collapsed into one bucket.
  const uint32_t stride = max(1u, v.n / S);
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
  for (uint32_t t = tid; t < S; t += nthreads) {
    const uint32_t i = ss_sample_pos(t, stride, v.n);
    v.smp_key[t] = make_key(v.raw[i], v.max_index);
    v.smp_pos[t] = i;
    v.smp_rank[t] = 0;
  }
  for (uint32_t r = tid; r < nrows; r += nthreads) {
    const uint32_t e = offset[r + 1];
    for (uint32_t j = offset[r]; j < e; ++j) rowid[j] = r;
  }
}

// ---- ranking of the samples, tiled over the whole chip: block (i, j) compares 256 "row"
// samples (one per thread, in registers) with 256 "column" samples (LDS broadcast reads) and
// adds the partial ranks with one atomic per sample.  The composites are distinct, so the
// final rank of a sample is its sorted position; every SS_OVERSAMPLE-th becomes a splitter
// (picked up by k_ss_count).  (One block sorting the samples costs 50-60 us: a single CU is
// issue-bound on it.)
__global__ void __launch_bounds__(256) k_ss_rank(SSView v) {
  __shared__ uint64_t ck[256];
  __shared__ uint32_t cp[256];
  const uint32_t S = (uint32_t)v.P * SS_OVERSAMPLE;
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
  for (uint32_t j = 0; j < lim; ++j) rank += comp_less(ck[j], cp[j], mk, mp) ? 1u : 0u;
  if (row < S && rank) atomicAdd(&v.smp_rank[row], rank);
}

// ---- count: bucket + rank-in-(tile,bucket) of every pair; per-tile histogram
__global__ void __launch_bounds__(SS_TILE_THREADS) k_ss_count(SSView v) {
  __shared__ uint64_t sk[SS_MAX_BUCKETS];
  __shared__ uint32_t sp[SS_MAX_BUCKETS];
  __shared__ uint32_t hist[SS_MAX_BUCKETS];
  const int P = v.P;
  for (int b = threadIdx.x; b < P; b += blockDim.x) hist[b] = 0;
  // splitters: sample with rank r = m * SS_OVERSAMPLE (m >= 1) is splitter m-1; spl[P-1] = +inf
  {
    const uint32_t S = (uint32_t)P * SS_OVERSAMPLE;
    for (uint32_t t0 = threadIdx.x; t0 < S; t0 += blockDim.x * 8) {  // 8 independent rank loads in flight
      uint32_t r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t t = t0 + u * blockDim.x;
        r[u] = t < S ? v.smp_rank[t] : 1u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t t = t0 + u * blockDim.x;
        if (r[u] > 0 && r[u] % SS_OVERSAMPLE == 0) {
          sk[r[u] / SS_OVERSAMPLE - 1] = v.smp_key[t];
          sp[r[u] / SS_OVERSAMPLE - 1] = v.smp_pos[t];
        }
      }
    }
    if (threadIdx.x == 0) {
      sk[P - 1] = ~0ULL;
      sp[P - 1] = ~0u;
    }
  }
  const uint32_t base = blockIdx.x * SS_TILE;
  uint64_t key[SS_PER_THREAD];
#pragma unroll
  for (int e = 0; e < SS_PER_THREAD; ++e) {  // independent loads first
    const uint32_t i = base + e * SS_TILE_THREADS + threadIdx.x;
    key[e] = i < v.n ? make_key(v.raw[i], v.max_index) : ~0ULL;
  }
  __syncthreads();
  // SS_PER_THREAD interleaved binary searches: first b with (key, i) < spl[b]  (spl[P-1] = +inf)
  int lo[SS_PER_THREAD], hi[SS_PER_THREAD];
#pragma unroll
  for (int e = 0; e < SS_PER_THREAD; ++e) { lo[e] = 0; hi[e] = P - 1; }
  for (int step = P; step > 1; step = (step + 1) >> 1) {  // ceil(log2 P) rounds
#pragma unroll
    for (int e = 0; e < SS_PER_THREAD; ++e) {
      if (lo[e] < hi[e]) {
        const uint32_t i = base + e * SS_TILE_THREADS + threadIdx.x;
        const int mid = (lo[e] + hi[e]) >> 1;
        if (comp_less(key[e], i, sk[mid], sp[mid])) hi[e] = mid; else lo[e] = mid + 1;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < SS_PER_THREAD; ++e) {
    const uint32_t i = base + e * SS_TILE_THREADS + threadIdx.x;
    if (i < v.n) {
      const uint32_t r = atomicAdd(&hist[lo[e]], 1u);
      v.packed[i] = ((uint32_t)lo[e] << 16) | r;
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < P; b += blockDim.x) v.hist[blockIdx.x * P + b] = hist[b];
}

// ---- scan: run_off[tile][b] = pairs of bucket b in earlier tiles; btotal[b].
// One block per SS_SCAN_BUCKETS buckets; 64 lanes = 64 consecutive buckets
// (coalesced), the block's 4 waves split the tiles and combine through LDS.
__global__ void __launch_bounds__(256) k_ss_scan(SSView v) {
  __shared__ uint32_t wtot[4][SS_SCAN_BUCKETS];
  const int P = v.P;
  const int b = blockIdx.x * SS_SCAN_BUCKETS + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int per = (v.ntiles + 3) / 4;
  const int t_beg = w * per, t_end = min(v.ntiles, t_beg + per);
  const uint32_t* __restrict__ hist = v.hist;
  uint32_t* __restrict__ run_off = v.run_off;
  uint32_t sum = 0;
  if (b < P) {
    for (int t0 = t_beg; t0 < t_end; t0 += 8) {
      uint32_t h[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) h[q] = (t0 + q < t_end) ? hist[(t0 + q) * P + b] : 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += h[q];
    }
  }
  wtot[w][threadIdx.x & 63] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (int q = 0; q < w; ++q) run += wtot[q][threadIdx.x & 63];
  if (b < P) {
    for (int t0 = t_beg; t0 < t_end; t0 += 8) {
      uint32_t h[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) h[q] = (t0 + q < t_end) ? hist[(t0 + q) * P + b] : 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (t0 + q < t_end) run_off[(t0 + q) * P + b] = run;
        run += h[q];
      }
    }
    if (w == 3) v.btotal[b] = run;
  }
}

// ---- scatter into bucket-major order; every block derives the bucket starts
// from the totals (block 0 publishes them)
__global__ void __launch_bounds__(SS_TILE_THREADS) k_ss_scatter(SSView v) {
  __shared__ uint32_t off[SS_MAX_BUCKETS];
  __shared__ uint32_t wsum[SS_TILE_THREADS / 64];
  const int P = v.P;
  // exclusive scan of btotal[0..P): BPT consecutive buckets per thread
  {
    constexpr int BPT = SS_MAX_BUCKETS / SS_TILE_THREADS;
    const int b0 = threadIdx.x * BPT;
    uint32_t tt[BPT];
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      tt[q] = b0 + q < P ? v.btotal[b0 + q] : 0u;
      sum += tt[q];
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan<SS_TILE_THREADS / 64>(sum, wsum, &total);
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
      if (b0 + q < P) {
        off[b0 + q] = ex;
        if (blockIdx.x == 0) v.bstart[b0 + q] = ex;
      }
      ex += tt[q];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) v.bstart[P] = total;
  }
  const uint32_t base = blockIdx.x * SS_TILE;
  uint64_t raw[SS_PER_THREAD];
  uint32_t pk[SS_PER_THREAD];
#pragma unroll
  for (int e = 0; e < SS_PER_THREAD; ++e) {
    const uint32_t i = base + e * SS_TILE_THREADS + threadIdx.x;
    raw[e] = i < v.n ? v.raw[i] : 0;
    pk[e] = i < v.n ? v.packed[i] : 0;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < P; b += blockDim.x) off[b] += v.run_off[blockIdx.x * P + b];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < SS_PER_THREAD; ++e) {
    const uint32_t i = base + e * SS_TILE_THREADS + threadIdx.x;
    if (i < v.n) {
      const uint32_t dst = off[pk[e] >> 16] + (pk[e] & 0xFFFFu);
      v.bkeys[dst] = make_key(raw[e], v.max_index);
      v.bpos[dst] = i;
    }
  }
}

// ---- sort one bucket; heads and 1-based local unique ranks
__global__ void __launch_bounds__(SS_SORT_THREADS) k_ss_sort(SSView v) {
  __shared__ uint64_t ak[SS_LDS_CAP];
  __shared__ uint32_t ap[SS_LDS_CAP];
  __shared__ uint64_t bk[SS_LDS_CAP];
  __shared__ uint32_t bp[SS_LDS_CAP];
  __shared__ uint32_t wsum[SS_SORT_THREADS / 64];
  const uint32_t b = blockIdx.x;
  const uint32_t beg = v.bstart[b], end = v.bstart[b + 1];
  const uint32_t n = end - beg;
  if (n == 0) {
    if (threadIdx.x == 0) {
      v.nheads[b] = 0;
      v.first_key[b] = 0;
      v.last_key[b] = 0;
    }
    return;
  }
  const uint64_t* gk = v.bkeys + beg;
  const uint32_t* gp = v.bpos + beg;
  const bool in_lds = n <= SS_LDS_CAP && !v.force_global;
  uint64_t* sk = bk;  // where the sorted bucket ends up (LDS path)
  uint32_t* sp = bp;
  if (in_lds) {
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
      ak[t] = gk[t];
      ap[t] = gp[t];
    }
    __syncthreads();
    // runs of 64: rank every element inside its 64-chunk (LDS broadcast reads, no barriers)
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t c0 = (threadIdx.x >> 6) * 64; c0 < n; c0 += SS_SORT_THREADS) {
      const uint32_t idx = c0 + lane;
      const bool valid = idx < n;
      const uint64_t mk = valid ? ak[idx] : ~0ULL;
      const uint32_t mp = valid ? ap[idx] : ~0u;
      const uint32_t lim = min(64u, n - c0);
      uint32_t rank = 0;
      for (uint32_t j = 0; j < lim; ++j) rank += comp_less(ak[c0 + j], ap[c0 + j], mk, mp) ? 1u : 0u;
      if (valid) {
        bk[c0 + rank] = mk;
        bp[c0 + rank] = mp;
      }
    }
    __syncthreads();
    // merge rounds: element's slot = its offset in its run + (# smaller elements in the sibling run)
    uint64_t* dk = ak;
    uint32_t* dp = ap;
    for (uint32_t L = 64; L < n; L <<= 1) {
      for (uint32_t idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const uint64_t mk = sk[idx];
        const uint32_t mp = sp[idx];
        const uint32_t r = idx / L;
        const uint32_t pair_base = (r & ~1u) * L;
        const uint32_t sib = (r ^ 1u) * L;
        uint32_t lo = min(sib, n), hi = min(sib + L, n);
        const uint32_t sib_beg = lo;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (comp_less(sk[mid], sp[mid], mk, mp)) lo = mid + 1; else hi = mid;
        }
        const uint32_t dst = pair_base + (idx - r * L) + (lo - sib_beg);
        dk[dst] = mk;
        dp[dst] = mp;
      }
      __syncthreads();
      uint64_t* tk = sk; sk = dk; dk = tk;
      uint32_t* tp = sp; sp = dp; dp = tp;
    }
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
      v.skeys[beg + t] = sk[t];
      v.spos[beg + t] = sp[t];
    }
  } else {
    // oversize bucket (an outlier of the sampling; kept for correctness and bounded
    // at n log n): the same run-ranking + merge rounds on the global arrays,
    // ping-ponging between the bucket-major and the sorted buffers
    uint64_t* xk = v.bkeys + beg;
    uint32_t* xp = v.bpos + beg;
    uint64_t* yk = v.skeys + beg;
    uint32_t* yp = v.spos + beg;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t c0 = (threadIdx.x >> 6) * 64; c0 < n; c0 += SS_SORT_THREADS) {
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
  // heads + inclusive scan of heads over the bucket
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += blockDim.x) {
    const uint32_t t = base + threadIdx.x;
    uint32_t h = 0;
    if (t < n) {
      const uint64_t k = in_lds ? sk[t] : v.skeys[beg + t];
      h = (t == 0 || k != (in_lds ? sk[t - 1] : v.skeys[beg + t - 1])) ? 1u : 0u;
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<SS_SORT_THREADS / 64>(h, wsum, &total);
    if (t < n) {
      v.head[beg + t] = h;
      v.luid[beg + t] = carry + ex + h;
    }
    carry += total;
  }
  if (threadIdx.x == 0) {
    v.nheads[b] = carry;
    v.first_key[b] = in_lds ? sk[0] : v.skeys[beg];
    v.last_key[b] = in_lds ? sk[n - 1] : v.skeys[beg + n - 1];
  }
}

// ---- emit: one block per bucket stitches itself to its predecessors (unique
// keys before the bucket; does its first key continue the previous bucket's
// last key?) and writes the Localizer's outputs
__global__ void __launch_bounds__(256) k_ss_emit(SSView v, const uint32_t* __restrict__ rowid,
                                                 const float* __restrict__ value, uint64_t* __restrict__ feaids,
                                                 uint32_t* __restrict__ col_ptr, uint32_t* __restrict__ index,
                                                 uint32_t* __restrict__ s_row, float* __restrict__ s_val,
                                                 uint32_t* __restrict__ d_U) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t sh_cont;
  const uint32_t b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0) { d_U[SEG_N_WORD] = 0; d_U[SEG_N_WORD + 1] = 0; }  // for k_seg_lists
  const uint32_t beg = v.bstart[b], end = v.bstart[b + 1];
  if (beg == end) return;
  // uniq(b') = nheads[b'] - cont[b'] summed over b' < b; cont[b'] compares first_key[b'] with the
  // last key of the previous NON-EMPTY bucket
  uint32_t part = 0;
  for (uint32_t q0 = threadIdx.x; q0 <= b; q0 += blockDim.x) {
    const uint32_t nq = v.bstart[q0 + 1] - v.bstart[q0];
    if (nq == 0) continue;
    int p = (int)q0 - 1;
    while (p >= 0 && v.bstart[p + 1] == v.bstart[p]) --p;
    const uint32_t c = (p >= 0 && v.first_key[q0] == v.last_key[p]) ? 1u : 0u;
    if (q0 == b) sh_cont = c; else part += v.nheads[q0] - c;
  }
  uint32_t total;
  block_exclusive_scan<4>(part, wsum, &total);
  __syncthreads();
  const uint32_t c = sh_cont;
  const uint32_t ub = total;
  for (uint32_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
    const uint32_t uid = ub + v.luid[i] - 1 - c;
    const bool head = v.head[i] && !(c && i == beg);
    const uint32_t pos = v.spos[i];
    if (head) {
      feaids[uid] = v.skeys[i];
      col_ptr[uid] = i;
    }
    index[pos] = uid;  // RemapIndex, localizer.cc:63-77
    s_row[i] = rowid[pos];
    if (value) s_val[i] = value[pos];
    if (i == v.n - 1) {
      *d_U = uid + 1;
      col_ptr[uid + 1] = v.n;
    }
  }
}

}  // namespace dfh
#endif  // DFH_LOCALIZE_HIP_
