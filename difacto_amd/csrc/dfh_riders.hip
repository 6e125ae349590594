// The single-queue step: the Localizer of the NEXT minibatches as rider block ranges of the step's own three launches.
//
// The reference keeps two minibatches in flight with threads (reader + batch tracker, src/sgd/sgd_learner.cc:196-224).
// Through round 5 this library did it with a second HIP stream: Localizer + probe of minibatch t+1 on a low-priority
// preparation stream beside step t, ordered by two events per minibatch.  Measured there (profiles/r05g_*, r05z_*): a
// cross-stream hand-over costs the waiting queue 8.8 us, a kernel boundary inside one stream 0.8 us; k_loc_count lived
// 30.9 us beside the update against 8.1 us alone; the step without any Localizer took 92 us, with it 120.
//
// Here nothing runs on a second queue.  A stage of the sample sort (dfh_localize.hip: count, scatter, sort, emit — each a
// block function) of a LATER minibatch rides as extra blocks of a launch the current step makes anyway (k_lookup,
// k_forward, k_update_fused), exactly as the AUC's pair counting has ridden in the update launch since round 4.  The
// stages of one minibatch need a grid-wide ordering between them; the kernel boundaries of the stream give it for free:
//
//   launch of step t      default riders (dfh_ctx_set_option rider_slot_*)
//   L  k_lookup(t)        scatter(t+1)
//   F  k_forward(t)       sort(t+1)
//   U  k_update_fused(t)  emit(t+1), count(t+2)
//
// so that minibatch t+1 is localized when step t ends and its own lookup pass (probe + count push in one) opens step t+1.
// Rider blocks sit in the launch in groups of 8 (workgroups go round-robin over the 8 XCDs: a count / scatter tile keeps
// its XCD, main block b keeps the XCD it has without riders), one group every `period` groups while they last.
// Results are bit-identical to the stages run as launches of their own: the same block functions on the same arguments.
#ifndef DFH_RIDERS_HIP_
#define DFH_RIDERS_HIP_
#include <cstddef>

#include "dfh_internal.h"

namespace dfh {

enum RiderKind : uint32_t { RID_COUNT = 0, RID_SCATTER = 1, RID_SORT = 2, RID_EMIT = 3, RID_STAGES = 4 };
constexpr int RID_THREADS = 256;   // every carrier launch has 256-thread blocks
constexpr int MAX_RIDERS = 4;

struct Rider {
  LocView v;
  EmitOut o;        // RID_EMIT only
  uint32_t kind;
  uint32_t nblk;    // blocks of this stage (count / scatter: tiles; sort / emit: <= buckets)
};

struct RiderSet {
  uint32_t n;                      // riders in this launch (0: none)
  uint32_t ngroups;                // rider blocks / 8 (every rider's range is padded to a multiple of 8)
  uint32_t period;                 // one rider group every `period` groups of 8 blocks (1: all riders first)
  uint32_t start;                  // main groups dispatched before the first rider group (0: riders from the first block on)
  uint32_t first[MAX_RIDERS + 1];  // rider j owns rider blocks [first[j], first[j + 1])
  Rider r[MAX_RIDERS];
};

// LDS a launch needs for its riders (dynamic shared memory of the carrier launch)
inline size_t rider_smem(const RiderSet& rs) {
  size_t need = 0;
  for (uint32_t j = 0; j < rs.n; ++j) {
    size_t x = 0;
    switch (rs.r[j].kind) {
      case RID_COUNT: x = loc_count_smem<LOC_MAX_BUCKETS>(); break;
      case RID_SCATTER: x = loc_scatter_smem<LOC_MAX_BUCKETS, RID_THREADS>(); break;
      case RID_SORT: x = loc_sort_smem(); break;
      default: x = loc_emit_smem(); break;
    }
    need = x > need ? x : need;
  }
  return need;
}

// launch index -> rider block (true, idx among the rider blocks) or main block (false, idx among the main blocks)
__device__ __forceinline__ bool rider_map(const RiderSet& rs, const uint32_t i, uint32_t& idx) {
  uint32_t g = i >> 3;
  const uint32_t l = i & 7u;
  if (g < rs.start) {  // the carrier's first blocks: its long chains are under way before the first rider takes a slot
    idx = i;
    return false;
  }
  g -= rs.start;
  const uint32_t q = g / rs.period, rem = g - q * rs.period;
  if (rem == 0 && q < rs.ngroups) {
    idx = q * 8u + l;
    return true;
  }
  const uint32_t before = min(q + 1u, rs.ngroups);  // rider groups at or before group g
  idx = (rs.start + g - before) * 8u + l;
  return false;
}

// rider block `idx` of the launch: one block of one stage of one later minibatch.  `rs` points INTO THE KERNEL ARGUMENT
// SEGMENT (rider_args): a run-time index into the by-value argument struct would make the compiler copy all of it into
// scratch, and a compile-time index per rider four inlined copies of every stage in every carrier kernel; read through the
// segment's own address the rider's arguments are scalar loads at a uniform offset.
__device__ __forceinline__ void run_rider(const RiderSet* __restrict__ rs, const uint32_t idx, char* smem) {
  uint32_t j = 0;
#pragma unroll
  for (int q = 1; q < MAX_RIDERS; ++q) j += idx >= rs->first[q] ? 1u : 0u;   // (first[q] = first[n] for q > n: no rider beyond n - 1)
  j = min(j, (uint32_t)MAX_RIDERS - 1u);
  const Rider& r = rs->r[j];
  const uint32_t bid = idx - rs->first[j];
  const uint32_t nblk = r.nblk;
  if (j >= rs->n || bid >= nblk) return;  // padding up to the next group of 8
  const uint32_t kind = r.kind;
  if (kind == RID_COUNT) loc_count_block<LOC_MAX_BUCKETS, RID_THREADS>(r.v, bid, nblk, smem);
  else if (kind == RID_SCATTER) loc_scatter_block<LOC_MAX_BUCKETS, RID_THREADS>(r.v, bid, smem);
  else if (kind == RID_SORT) loc_sort_block(r.v, bid, nblk, smem);
  else loc_emit_block<false>(r.v, r.o, TableView{}, nullptr, bid, nblk, smem);
}
// the RiderSet argument of the running kernel where the command processor put it: KA mirrors the kernel's parameter list
// (arguments sit in the segment in order, each at its natural alignment)
template <typename KA>
__device__ __forceinline__ const RiderSet* rider_args() {
  typedef __attribute__((address_space(4))) const char* ka_ptr;   // the constant address space: scalar loads
  return (const RiderSet*)((ka_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KA, rs));
}

extern __shared__ __attribute__((aligned(16))) char dfh_dyn_smem[];

// ---- L: the step's pass over its unique keys (k_lookup) + riders
__global__ void __launch_bounds__(RID_THREADS, 6) k_lookup_riders(TableView t, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ d_n,
                                                              uint32_t n_static, uint32_t* __restrict__ urow, const float* __restrict__ cnt,
                                                              const uint32_t* __restrict__ col_ptr, int push_cnt,
                                                              uint32_t* __restrict__ need_init, int rows_known, uint2* __restrict__ uw,
                                                              AucFin fin, SplitOut so, uint32_t nblk_main, RiderSet rs) {
  struct KA {
    TableView t; const uint64_t* keys; const uint32_t* d_n; uint32_t n_static; uint32_t* urow; const float* cnt; const uint32_t* col_ptr;
    int push_cnt; uint32_t* need_init; int rows_known; uint2* uw; AucFin fin; SplitOut so; uint32_t nblk_main; RiderSet rs;
  };
  uint32_t idx;
  if (rider_map(rs, blockIdx.x, idx)) {
    run_rider(rider_args<KA>(), idx, dfh_dyn_smem);
    return;
  }
  if (idx >= nblk_main) return;
  lookup_body(t, keys, d_n, n_static, urow, cnt, col_ptr, push_cnt, need_init, rows_known, uw, fin, idx, nblk_main, so);
}

// ---- F: k_forward + riders (the table's own rows: the fused single-GPU step)
template <int L, int FWD_DEPTH>
__global__ void __launch_bounds__(RID_THREADS, DFH_FWD_WAVES) k_forward_riders(BatchView b, RowSrc src, int k, int kp, uint32_t nblk_main,
                                                                              RiderSet rs) {
  struct KA { BatchView b; RowSrc src; int k, kp; uint32_t nblk_main; RiderSet rs; };
  uint32_t idx;
  if (rider_map(rs, blockIdx.x, idx)) {
    run_rider(rider_args<KA>(), idx, dfh_dyn_smem);
    return;
  }
  if (idx >= nblk_main) return;
  forward_body<L, FWD_DEPTH, false>(b, src, k, kp, MixSrc{nullptr, 0}, idx, nblk_main, reinterpret_cast<double*>(dfh_dyn_smem));
}

// ---- U: k_update_fused + riders
template <int L, bool EXACT, bool HAS_VAL>
__global__ void __launch_bounds__(UPD_THREADS, DFH_UPD_WAVES) k_update_fused_riders(UpdArgs a, uint32_t nblk_main, RiderSet rs) {
  static_assert(UPD_THREADS == RID_THREADS, "riders are 256-thread blocks");
  struct KA { UpdArgs a; uint32_t nblk_main; RiderSet rs; };
  uint32_t idx;
  if (rider_map(rs, blockIdx.x, idx)) {
    run_rider(rider_args<KA>(), idx, dfh_dyn_smem);
    return;
  }
  if (idx >= nblk_main) return;
  update_body<L, EXACT, HAS_VAL, false>(a, idx, nblk_main, dfh_dyn_smem);
}

}  // namespace dfh
#endif  // DFH_RIDERS_HIP_
