// k_update_fused: FMLoss::CalcGrad (src/loss/fm_loss.h:148-199) + Store::Push(kGradient) ->
// SGDUpdater::Update (src/sgd/sgd_updater.cc:74-138) on the resident table, ONE launch, for gfx950.
//
// The segment lengths of a minibatch's keys are Zipf-distributed: at C3 size 80 % of the unique keys
// occur exactly once, 17 % two to eight times, and the 4 000 keys beyond hold half of all occurrences
// (up to ~1 300 each).  Four roles, chosen by block range, longest dependent chains first:
//
//   hot      > BWD_MID occurrences      one key per block: its waves split the segment in tiles of 64
//                                       occurrences, partial sums meet in LDS in wave order
//   mid      BWD_SMALL+1 .. BWD_MID     one key per wave, one tile
//   few      2 .. BWD_SMALL             one L-lane group per key, occurrences summed serially in row
//                                       order (the reference's own order, spmm.h:137-156)
//   singles  1                          EXAMPLE by example, no key-ordered view at all: the wave that
//                                       owns example i holds p_i and XV_i once (one coalesced 4*kp B
//                                       read instead of a random one per key), picks the example's
//                                       single-occurrence keys out of {row, flags} words k_lookup left
//                                       in uw[] and updates up to DFH_UPD_ROUNDS * (64/L) rows per
//                                       round trip.  gw = p x, gV = (XV p) x - V x^2 p: one term, no sum.
//
// hot / mid / few walk per-bucket key lists the Localizer's emit pass writes (k_loc_emit; k_seg_lists
// for the other Localizer paths).  A tile keeps DFH_UPD_DEPTH XV-row loads per lane in flight (the
// long-segment roles are chains of L2 round trips: depth, not occupancy, shortens them).  Every key is
// updated by exactly one role, by one lane group: deterministic, no float atomics, nothing is
// communicated between blocks.  V rows are read speculatively (a row without V holds zeros).
// The update arithmetic is ftrl_update_w / adagrad_update_v of dfh_kernels.hip (operation for
// operation the reference's); the kernel is pinned to 8 waves per SIMD (64 registers) like
// k_backward_all, for the same reason (co-residency with the preparation stream).
#ifndef DFH_UPDATE_HIP_
#define DFH_UPDATE_HIP_
#include "dfh_internal.h"

namespace dfh {

#ifndef DFH_UPD_DEPTH
#define DFH_UPD_DEPTH 8
#endif
#ifndef DFH_UPD_ROUNDS
#define DFH_UPD_ROUNDS 2
#endif
#ifndef DFH_UPD_FEW_DEPTH
#define DFH_UPD_FEW_DEPTH 4
#endif
#ifndef DFH_UPD_ROLES
#define DFH_UPD_ROLES 15  // measurement builds only: bit mask of the roles compiled in (hot, mid, few, singles)
#endif
#ifndef DFH_UPD_WAVES
#define DFH_UPD_WAVES 5
#endif
#ifndef DFH_UPD_THREADS
#define DFH_UPD_THREADS 256
#endif
constexpr int UPD_THREADS = DFH_UPD_THREADS;
constexpr int UPD_NW = UPD_THREADS / 64;

// one compact argument block (the roles use disjoint parts of it)
struct UpdArgs {
  // the minibatch, row order
  const uint32_t* offset;   // [nrows + 1]
  const uint32_t* index;    // [nnz] rank of the key of every nonzero
  const float* value;       // [nnz] or NULL (binary features)
  // per unique key
  const uint2* uw;          // [U] {table row | kSingleRow, w} as of this step's k_lookup
  const uint32_t* col_ptr;  // [U + 1] segment starts in the key-ordered view
  const uint64_t* feaids;   // [U]
  const float* feacnt;      // add_cnt: explicit occurrence counts per key, or NULL (the segment lengths)
  // key-ordered view
  const uint32_t* s_row;    // [nnz]
  const float* s_val;       // [nnz] or NULL
  // per example, left by k_forward
  const float* slope;       // [nrows] p_i
  const float* xv;          // [nrows x kp]
  // the model
  RowHdr* hdr;
  float* va;
  uint32_t* need_init;      // REFRAND: keys whose V is to be initialised after the launch
  double* prog;
  SegLists seg;
  uint32_t nrows;
  uint32_t nlist;           // list buckets
  uint32_t nb_hot, nb_mid, nb_few;  // blocks per role; the remaining blocks of the grid take the singles
  uint32_t ileave;          // R > 1: every R-th block (in dispatch order) is a list-role block until those run out;
                            // 0 / 1: all list-role blocks first
  int k, kp;
  KeyRange rg;              // sharded store: only the keys this rank owns
  dfh_updater_param p;
  // BinClassMetric::AUC of the minibatch's predictions as the first nb_auc blocks of the launch (auc_pairs_block:
  // VALU work beside a memory-bound kernel), or nb_auc = 0
  const float* auc_pred;
  const float* auc_label;
  uint32_t* auc_part;       // the units' slots (auc_pairs_block); finalised by a later launch (auc_finalize_block)
  uint32_t nb_auc;
  // MIXED (sharded store, round 5): the keys OTHER ranks own (kRemoteRow in the row word; their "row" is the key's rank u)
  // read [w, has_V, 0, 0 | V] from the rows those owners sent and leave [gw, has_V, 0, 0 | gV] for them — CalcGrad in the
  // exchange layout, what k_backward_all<FUSED = false> did in a launch of its own — while this rank's own keys are
  // updated in place, all in ONE launch
  const float* rrows;       // pulled rows, rstride floats each, indexed by the key's rank
  float* grows;             // gradient rows out, same layout
  uint32_t rstride;
  // round 6: the keys with more than HOT_SPLIT_MIN occurrences, one block per PART of HOT_SPLIT (upd_split_role): partial sums per part in
  // split_part[entry][UPD_SPLIT_STRIDE], a ticket per key in split_ticket[first entry] (zero between launches)
  float* split_part;
  uint32_t* split_ticket;
  uint32_t nb_split;        // 1: the hot role's blocks take the split list's parts first; 0: keys of any length go through the hot role whole
};
constexpr int UPD_SPLIT_STRIDE = 4 + 256;   // gw, xxp, -, - | gv[kp <= 256]

// Model rows (V, accumulators) are loaded and stored with streaming (nt) hints.  Measured dead end, kept as
// -DDFH_UPD_NT=1: the same hints on everything else that is read once per launch (occurrence lists, row words, segment
// bounds, list entries, the 16 B header loads / stores) so that only XV and the slopes would stay in an XCD's 4 MB L2:
// 60.8 against 58.9 us stand-alone, 82.5 against 84.4 M examples/sec on one box, HBM-side traffic unchanged (213 MB).
#ifndef DFH_UPD_NT
#define DFH_UPD_NT 0
#endif
__device__ __forceinline__ uint32_t ldu_s(const uint32_t* p) { return DFH_UPD_NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ float ldf_s(const float* p) { return DFH_UPD_NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ uint32_t ld_rowword(const uint2* p) {  // .x of {row | flags, w}
  return DFH_UPD_NT ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p)) : p->x;
}

// Keys whose V row is to be initialised (lazy InitV, hash mode) by this block: noted by the lane that takes the decision
// (upd_apply), written by the whole block once its role is through (upd_init_rows) — 16 B of V and of the accumulators per
// lane instead of one lane's 2 kp scalar stores, and outside the roles' loops, whose register budget it does not touch
// (written inline at the end of upd_apply the same stores cost the warm step 1.4 %).  One array per block in LDS.
constexpr uint32_t UPD_INIT_CAP = 511;
constexpr int UPD_INIT_BATCH = 4;   // rows a lane group initialises per round trip (their keys are fetched together)
__device__ __forceinline__ uint32_t* upd_init_list() {
  // [0]: entries noted (may exceed the capacity: the surplus was written inline); then {key rank u, table row r} per entry
  __shared__ uint32_t list[2 + 2 * UPD_INIT_CAP];
  return list;
}
// Round 6: the epilogue used to fetch every listed row's word and key one row after the other — a dependent global round trip
// per row and lane group, eight in a row for the ~125 rows a singles block of a COLD step lists (every key of the first epoch's
// minibatches is new), i.e. most of the 25 us a cold update launch took beyond a warm one (profiles/r05j_*).  The noting lane
// now leaves the row id beside the key's rank, and a lane group fetches the keys of UPD_INIT_BATCH rows together.
__device__ __forceinline__ void upd_init_rows(const UpdArgs& a, int L) {
  __syncthreads();
  const uint32_t* il = upd_init_list();
  const uint32_t n = min(il[0], UPD_INIT_CAP);
  const int k = a.k, kp = a.kp;
  const uint32_t G = UPD_THREADS / L;
  for (uint32_t e0 = threadIdx.x / L; e0 < n; e0 += G * UPD_INIT_BATCH) {
    uint64_t key[UPD_INIT_BATCH];
    uint32_t row[UPD_INIT_BATCH];
#pragma unroll
    for (int q = 0; q < UPD_INIT_BATCH; ++q) {  // entries past the list: the last one again (an unconditional load), never written
      const uint32_t e = min(e0 + q * G, n - 1u);
      row[q] = il[3 + 2 * e];
      key[q] = a.feaids[il[2 + 2 * e]];
    }
#pragma unroll
    for (int q = 0; q < UPD_INIT_BATCH; ++q) {
      if (e0 + q * G >= n) break;
      float* va = a.va + (size_t)row[q] * (size_t)(2 * kp);
      for (int d0 = (threadIdx.x % L) * 4; d0 < kp; d0 += L * 4) {
        float4 nv;
        nv.x = d0 + 0 < k ? hash_init_value(key[q], d0 + 0, a.p.seed, a.p.V_init_scale) : 0.f;
        nv.y = d0 + 1 < k ? hash_init_value(key[q], d0 + 1, a.p.seed, a.p.V_init_scale) : 0.f;
        nv.z = d0 + 2 < k ? hash_init_value(key[q], d0 + 2, a.p.seed, a.p.V_init_scale) : 0.f;
        nv.w = d0 + 3 < k ? hash_init_value(key[q], d0 + 3, a.p.seed, a.p.V_init_scale) : 0.f;
        st4_nt(va + d0, nv);
        st4_nt(va + kp + d0, make_float4(0.f, 0.f, 0.f, 0.f));
      }
    }
  }
}

// SGDUpdater::Update(kGradient) for one key whose sums are complete: executed by the L lanes of ONE
// group (the caller masks the others).  h0 = {w, has_V, sqrt_g, z}; vv / ac: this lane's V and
// accumulator slices; g4: sum of (XV p) x over the occurrences.
template <bool EXACT, bool MIXED = false>
__device__ __forceinline__ void upd_apply(const UpdArgs& a, uint32_t r, uint32_t u, const float4 h0, const float4 vv, const float4 ac,
                                          float gw, float xxp, float4 g4, int sub, bool sub_ok, int k, int kp, float& pen, float fc,
                                          float cnt, bool cnt_later, bool remote = false) {
  const float w_old = h0.x;
  const bool has_v = k > 0 && __float_as_uint(h0.y) != 0u;
  RowHdr* hp = a.hdr + r;
  float* va = a.va + (size_t)r * (size_t)(2 * kp);
  if (has_v) {
    // grad_V = X'(diag(p) XV) - diag(XXp) V   (fm_loss.h:181-198)
    g4.x -= vv.x * xxp; g4.y -= vv.y * xxp; g4.z -= vv.z * xxp; g4.w -= vv.w * xxp;
    // penalty of the PULLED weights (SGDLearner::EvaluatePenalty, sgd_learner.cc:249-273)
    if (sub_ok) pen += 0.5f * a.p.V_l2 * (vv.x * vv.x + vv.y * vv.y + vv.z * vv.z + vv.w * vv.w);
  }
  if (MIXED && remote) {  // another rank's key (uniform per lane group): its gradient row, for the owner to apply
    float* g = a.grows + (size_t)u * a.rstride;
    if (sub == 0) {
      pen += a.p.l1 * fabsf(w_old) + 0.5f * a.p.l2 * w_old * w_old;
      st4_nt(g, make_float4(gw, has_v ? 1.0f : 0.0f, 0.f, 0.f));
    }
    if (sub_ok && k > 0) st4_nt(g + 4 + sub * 4, has_v ? g4 : make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  if (sub == 0) {
    pen += a.p.l1 * fabsf(w_old) + 0.5f * a.p.l2 * w_old * w_old;
    float sqrt_g = h0.z, z = h0.w;
    const float w_new = ftrl_update_w(gw, w_old, sqrt_g, z, a.p);
    uint32_t hv = has_v ? 1u : 0u;
    // Push(kFeaCount) of a key that has its V, left to this kernel by k_lookup (sgd_updater.cc:62-73: fea_cnt += cnt; the
    // InitV test there cannot fire for it)
    if (cnt_later) hp->fea_cnt = fc + (a.feacnt ? a.feacnt[u] : cnt);
    // lazy InitV when w leaves zero (sgd_updater.cc:122-126); a key without V had its count pushed by k_lookup
    if (w_old == 0 && w_new != 0 && k > 0 && !has_v && fc > (float)a.p.V_threshold) {
      if (a.p.init_mode == DFH_INIT_HASH) {
        // the row is written at the end of the block by all of its lanes (upd_init_rows); here only its name is noted.
        // Until round 4 this lane wrote the row alone, 2 kp scalar stores behind 2 kp hashes: with every key of a
        // minibatch new (an empty table's first steps) the launch took 147 us against 61 (profiles/r04t_cold_start.txt)
        uint32_t* il = upd_init_list();
        const uint32_t slot = atomicAdd(&il[0], 1u);
        if (slot < UPD_INIT_CAP) {
          il[2 + 2 * slot] = u;
          il[3 + 2 * slot] = r;
        } else {  // list full: as before
          const uint64_t key = a.feaids[u];
          for (int j = 0; j < kp; ++j) {
            va[j] = j < k ? hash_init_value(key, j, a.p.seed, a.p.V_init_scale) : 0.0f;
            va[kp + j] = 0.0f;
          }
        }
        hv = 1u;
      } else {
        a.need_init[u] = 1;
      }
    }
    if (DFH_UPD_NT) st4_nt(reinterpret_cast<float*>(hp), make_float4(w_new, __uint_as_float(hv), sqrt_g, z));  // one 16 B store
    else st4(reinterpret_cast<float*>(hp), make_float4(w_new, __uint_as_float(hv), sqrt_g, z));
  }
  if (has_v && sub_ok) {
    float4 nv = vv, na = ac;
    adagrad_update_v(g4.x, nv.x, na.x, a.p);
    adagrad_update_v(g4.y, nv.y, na.y, a.p);
    adagrad_update_v(g4.z, nv.z, na.z, a.p);
    adagrad_update_v(g4.w, nv.w, na.w, a.p);
    if (!EXACT || k != kp) {  // padded coordinates (>= k) stay exactly zero
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

// the model row of a key, for the group that will apply its update: {w, has_V, sqrt_g, z}, V and
// accumulator slices, three independent loads.  NO load of this file sits behind a per-lane condition:
// a conditional load compiles to a branch with the wait for its result inside, one load in flight at a
// time.  Lanes with nothing to fetch read a valid address instead (row 0, their group's last row, the
// row's first slice) and never use what arrives.
__device__ __forceinline__ void upd_load_row(const UpdArgs& a, uint32_t r, int sub, bool sub_ok, int kp, float4& h0, float4& vv,
                                             float4& ac, float& fc) {
  const float* va = a.va + (size_t)r * (size_t)(2 * kp) + (sub_ok ? sub * 4 : 0);
  fc = a.hdr[r].fea_cnt;  // same 128 B line as h0
  h0 = DFH_UPD_NT ? ld4_nt(reinterpret_cast<const float*>(a.hdr + r)) : ld4(reinterpret_cast<const float*>(a.hdr + r));
  vv = ld4_nt(va);
  ac = ld4_nt(va + kp);
}

// MIXED: the same for a key whose row word may name another rank's key (kRemoteRow: "row" = the key's rank u, in the
// rows its owner sent).  The ADDRESSES are selected per lane group, the loads stay unconditional; a pulled row has no
// accumulators (its V slice is read twice) and no count.
__device__ __forceinline__ void upd_load_row_mixed(const UpdArgs& a, uint32_t rw, int sub, bool sub_ok, int kp, float4& h0, float4& vv,
                                                   float4& ac, float& fc) {
  const bool remote = (rw & kRemoteRow) != 0u;
  const uint32_t r = rw & kRowMask;
  const float* prow = a.rrows + (size_t)r * a.rstride;
  const float* va = a.va + (size_t)r * (size_t)(2 * kp) + (sub_ok ? sub * 4 : 0);
  const float* pv = remote ? prow + 4 + (sub_ok ? sub * 4 : 0) : va;
  const float* pa = remote ? pv : va + kp;
  const float* ph = remote ? prow : reinterpret_cast<const float*>(a.hdr + r);
  const float* pc = remote ? prow + 2 : &a.hdr[r].fea_cnt;   // a pulled row holds 0 there
  fc = *pc;
  h0 = ld4(ph);
  vv = ld4_nt(pv);
  ac = ld4_nt(pa);
}
template <bool MIXED>
__device__ __forceinline__ void upd_load(const UpdArgs& a, uint32_t rw, int sub, bool sub_ok, int kp, float4& h0, float4& vv, float4& ac,
                                         float& fc) {
  if (MIXED) upd_load_row_mixed(a, rw, sub, sub_ok, kp, h0, vv, ac, fc);
  else upd_load_row(a, rw & kRowMask, sub, sub_ok, kp, h0, vv, ac, fc);
}

// 16 B from a batch-sized array: uniform base + 32-bit byte offset (one address register per load in flight)
__device__ __forceinline__ float4 ld4_off(const float* base, uint32_t byte_off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
// 0/1 factors instead of selects on loaded values: a select whose operand is a load is turned back into a
// branch around that load by the code generator
__device__ __forceinline__ float upd_mask(bool c) { return c ? 1.0f : 0.0f; }

// sums over the occurrences [beg, end) this wave takes: tiles of 64 starting at tile w0, stride wstep
// tiles.  Per tile: one coalesced read of the occurrence list, one gather of the slopes, then the XV
// rows, DB loads per lane in flight (64/L rows per load instruction).  Result: gw / xxp wave-reduced,
// gv reduced across the groups (every group holds the sums of its lane slice).
template <int L, int DB, bool HAS_VAL>
__device__ __forceinline__ KeySums upd_tile_sums(const UpdArgs& a, uint32_t beg, uint32_t end, uint32_t w0, uint32_t wstep, int grp,
                                                 int sub, bool sub_ok, int k, int kp) {
  constexpr int G = 64 / L;
  const int lane = lane_id();
  const uint32_t lane_off = (sub_ok ? sub * 16 : 0), row_bytes = (uint32_t)kp * 4u;
  KeySums s;
  s.gw = 0.f; s.xxp = 0.f; s.gv = make_float4(0.f, 0.f, 0.f, 0.f);
  for (uint32_t base = beg + w0 * 64u; base < end; base += wstep * 64u) {
    const uint32_t j = min(base + lane, end - 1);  // lanes past the end re-read the last occurrence; their x is 0
    const uint32_t row = ldu_s(a.s_row + j);
    const float x = (HAS_VAL ? ldf_s(a.s_val + j) : 1.0f) * upd_mask(base + lane < end);
    const float p = a.slope[row];
    s.gw = fma_skip0(p, x, s.gw);          // spmv.h:155-163: a slope that is exactly 0 is skipped
    s.xxp = fma_skip0(p, x * x, s.xxp);    // fm_loss.h:171-178 with XX = value^2
    const int cnt = (int)min(64u, end - base);
    for (int t0 = 0; t0 < cnt; t0 += DB * G) {
      float4 av[DB];
#pragma unroll
      for (int q = 0; q < DB; ++q) {  // DB row loads in flight per lane; occurrences past cnt: some valid row, factor 0
        const int tt = t0 + q * G + grp;
        const uint32_t rowi = __shfl(row, tt & 63, 64);
        av[q] = ld4_off(a.xv, rowi * row_bytes + lane_off);
      }
#pragma unroll
      for (int q = 0; q < DB; ++q) {
        // the factors are fetched when the row has arrived: registers hold rows, not broadcasts
        const int tt = t0 + q * G + grp;
        const float pp = __shfl(p, tt & 63, 64);
        const float xl = __shfl(x, tt & 63, 64);
        const float xx = tt < cnt ? xl : 0.f;
        s.gv.x += (av[q].x * pp) * xx; s.gv.y += (av[q].y * pp) * xx;
        s.gv.z += (av[q].z * pp) * xx; s.gv.w += (av[q].w * pp) * xx;
      }
    }
  }
  s.gw = wave_sum(s.gw);
  s.xxp = wave_sum(s.xxp);
  s.gv.x = cross_group_sum<L>(s.gv.x); s.gv.y = cross_group_sum<L>(s.gv.y);
  s.gv.z = cross_group_sum<L>(s.gv.z); s.gv.w = cross_group_sum<L>(s.gv.w);
  return s;
}

// list buckets are dealt to teams of `units` (waves or blocks): T units per bucket when there are more
// units than buckets.  -> false: this unit has no team
struct UpdTeam {
  uint32_t first, stride;  // buckets first, first + stride, ...
  uint32_t sub, size;      // this unit's place in its team, units per team
};
__device__ __forceinline__ bool upd_team(uint32_t unit, uint32_t units, uint32_t nb, UpdTeam& t) {
  if (nb == 0 || units == 0) return false;
  t.size = max(1u, units / nb);
  t.stride = units / t.size;
  t.first = unit / t.size;
  t.sub = unit % t.size;
  return t.first < t.stride;
}

// ---- hot: one key per block and iteration
constexpr size_t UPD_HOT_SMEM = (size_t)UPD_NW * (2 + 256) * 4;
template <int L, bool EXACT, int DB, bool HAS_VAL, bool MIXED>
__device__ __forceinline__ void upd_hot_role(const UpdArgs& a, uint32_t blk, uint32_t nblk, float& pen, char* smem /* UPD_HOT_SMEM bytes */) {
  float (*part)[2 + 256] = reinterpret_cast<float (*)[2 + 256]>(smem);  // per wave: gw, xxp, gv[kp <= 256]
  const int lane = lane_id();
  const int grp = lane / L, sub = lane % L;
  const int kp = EXACT ? 4 * L : a.kp, k = a.k;
  const bool sub_ok = EXACT ? true : (sub * 4 < kp);
  const int w = threadIdx.x >> 6;
  UpdTeam tm;
  if (!upd_team(blk, nblk, a.nlist, tm)) return;  // uniform per block: the barriers below are reached by all of its threads
  for (uint32_t lb = tm.first; lb < a.nlist; lb += tm.stride) {
    const uint2 co = a.seg.hot[lb];
    if (co.x == 0) continue;
    const SegEnt* __restrict__ ent = a.seg.hot_ent + co.y;
    for (uint32_t q = tm.sub; q < co.x; q += tm.size) {
      const SegEnt e = ent[q];  // {u, beg, end}: the segment comes with the entry
      const uint32_t u = e.x;
      if (!key_in(a.rg, u)) continue;  // uniform per block
      const uint32_t beg = e.y, end = e.z;
      if ((DFH_HOT_SPLIT_BUILD & 2) && a.nb_split && end - beg > HOT_SPLIT_MIN) continue;  // its parts are in the split list: upd_split_role
      const uint32_t rw = ld_rowword(a.uw + u);
      const uint32_t r = rw & kRowMask;
      const KeySums s = upd_tile_sums<L, DB, HAS_VAL>(a, beg, end, (uint32_t)w, UPD_NW, grp, sub, sub_ok, k, kp);
      __syncthreads();  // the previous key's partials have been consumed
      if (grp == 0) {
        if (sub == 0) { part[w][0] = s.gw; part[w][1] = s.xxp; }
        if (sub_ok) {
          part[w][2 + sub * 4 + 0] = s.gv.x; part[w][2 + sub * 4 + 1] = s.gv.y;
          part[w][2 + sub * 4 + 2] = s.gv.z; part[w][2 + sub * 4 + 3] = s.gv.w;
        }
      }
      __syncthreads();
      if (w == 0 && grp == 0) {
        float4 h0, vv, ac;
        float fc;
        upd_load<MIXED>(a, rw, sub, sub_ok, kp, h0, vv, ac, fc);
        float gw = 0.f, xxp = 0.f;
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < UPD_NW; ++i) {  // wave order: deterministic
          gw += part[i][0];
          xxp += part[i][1];
          if (sub_ok) {
            g4.x += part[i][2 + sub * 4 + 0]; g4.y += part[i][2 + sub * 4 + 1];
            g4.z += part[i][2 + sub * 4 + 2]; g4.w += part[i][2 + sub * 4 + 3];
          }
        }
        upd_apply<EXACT, MIXED>(a, r, u, h0, vv, ac, gw, xxp, g4, sub, sub_ok, k, kp, pen, fc, (float)(end - beg), (rw & kCountLater) != 0u,
                                (rw & kRemoteRow) != 0u);
      }
    }
  }
}

// ---- split: one PART (<= HOT_SPLIT occurrences) of a very hot key per block and iteration.  A key in every row of the minibatch
// (a bias-like feature, the token of a missing value) is a segment of B occurrences: in the hot role ONE block walks it, 40
// tiles per wave at C3 size, ~3 dependent round trips each — a launch as long as that chain whatever else it holds.  Here
// every part has a block; a part's sums go to split_part[entry] (stores, then a device-scope RELEASE: the L2's dirty lines are
// written back — no invalidate), a ticket per key counts the parts that have arrived, and the block that brings the LAST part
// adds the partials up IN PART ORDER (reads that bypass the L2: agent-scope atomic loads) — deterministic, whichever block
// comes last — fetches the row and applies the update.  Nobody waits for anybody.
template <int L, bool EXACT, int DB, bool HAS_VAL, bool MIXED>
__device__ __forceinline__ void upd_split_role(const UpdArgs& a, uint32_t blk, uint32_t nblk, float& pen, char* smem) {
  float (*part)[2 + 256] = reinterpret_cast<float (*)[2 + 256]>(smem);
  const int lane = lane_id();
  const int grp = lane / L, sub = lane % L;
  const int kp = EXACT ? 4 * L : a.kp, k = a.k;
  const bool sub_ok = EXACT ? true : (sub * 4 < kp);
  const int w = threadIdx.x >> 6;
  const uint32_t n = *a.seg.split_n;
  for (uint32_t j = blk; j < n; j += nblk) {
    const SegEnt e = a.seg.split_ent[j];
    const uint32_t u = e.x, beg = e.y, end = e.z, p = e.w >> 16, nparts = e.w & 0xFFFFu;
    if (!key_in(a.rg, u)) continue;  // uniform per block (and per key: none of its parts takes a ticket)
    const KeySums s = upd_tile_sums<L, DB, HAS_VAL>(a, beg, end, (uint32_t)w, UPD_NW, grp, sub, sub_ok, k, kp);
    __syncthreads();  // the previous part's partials have been consumed
    if (grp == 0) {
      if (sub == 0) { part[w][0] = s.gw; part[w][1] = s.xxp; }
      if (sub_ok) {
        part[w][2 + sub * 4 + 0] = s.gv.x; part[w][2 + sub * 4 + 1] = s.gv.y;
        part[w][2 + sub * 4 + 2] = s.gv.z; part[w][2 + sub * 4 + 3] = s.gv.w;
      }
    }
    __syncthreads();
    if (w == 0 && grp == 0) {   // (wave-uniform up to the lane group: the other groups of wave 0 idle through this)
      float gw = 0.f, xxp = 0.f;
      float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < UPD_NW; ++i) {  // wave order: deterministic
        gw += part[i][0];
        xxp += part[i][1];
        if (sub_ok) {
          g4.x += part[i][2 + sub * 4 + 0]; g4.y += part[i][2 + sub * 4 + 1];
          g4.z += part[i][2 + sub * 4 + 2]; g4.w += part[i][2 + sub * 4 + 3];
        }
      }
      const uint32_t first = j - p;   // a key's parts are consecutive entries
      float* mine = a.split_part + (size_t)j * UPD_SPLIT_STRIDE;
      if (sub == 0) { mine[0] = gw; mine[1] = xxp; }
      if (sub_ok) st4(mine + 4 + sub * 4, g4);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this part's sums are in memory before its ticket is
      uint32_t t = 0;
      if (sub == 0) t = atomicAdd(a.split_ticket + first, 1u);
      t = __shfl(t, 0, 64);   // (lane 0 is sub 0 of group 0)
      if (t == nparts - 1u) {   // the last part of its key to arrive: add the parts up in part order, apply
        gw = 0.f; xxp = 0.f;
        g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t q = 0; q < nparts; ++q) {
          const float* pq = a.split_part + (size_t)(first + q) * UPD_SPLIT_STRIDE;
          gw += __hip_atomic_load(pq + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          xxp += __hip_atomic_load(pq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float* gq = pq + 4 + (sub_ok ? sub * 4 : 0);
          g4.x += __hip_atomic_load(gq + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g4.y += __hip_atomic_load(gq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g4.z += __hip_atomic_load(gq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          g4.w += __hip_atomic_load(gq + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t kbeg = a.seg.split_ent[first].y, kend = a.seg.split_ent[first + nparts - 1u].z;
        const uint32_t rw = ld_rowword(a.uw + u);
        float4 h0, vv, ac;
        float fc;
        upd_load<MIXED>(a, rw, sub, sub_ok, kp, h0, vv, ac, fc);
        upd_apply<EXACT, MIXED>(a, rw & kRowMask, u, h0, vv, ac, gw, xxp, g4, sub, sub_ok, k, kp, pen, fc, (float)(kend - kbeg),
                                (rw & kCountLater) != 0u, (rw & kRemoteRow) != 0u);
        if (sub == 0) a.split_ticket[first] = 0u;   // ready for the next launch
      }
    }
  }
}

// ---- mid: one key per wave and iteration
template <int L, bool EXACT, int DB, bool HAS_VAL, bool MIXED>
__device__ __forceinline__ void upd_mid_role(const UpdArgs& a, uint32_t wave, uint32_t nwaves, float& pen) {
  const int lane = lane_id();
  const int grp = lane / L, sub = lane % L;
  const int kp = EXACT ? 4 * L : a.kp, k = a.k;
  const bool sub_ok = EXACT ? true : (sub * 4 < kp);
  UpdTeam tm;
  if (!upd_team(wave, nwaves, a.nlist, tm)) return;
  for (uint32_t lb = tm.first; lb < a.nlist; lb += tm.stride) {
    const uint2 co = a.seg.mid[lb];
    if (co.x == 0) continue;
    const SegEnt* __restrict__ ent = a.seg.mid_ent + co.y;
    for (uint32_t q = tm.sub; q < co.x; q += tm.size) {
      const SegEnt e = ent[q];
      const uint32_t u = e.x;
      if (!key_in(a.rg, u)) continue;  // uniform per wave
      const uint32_t beg = e.y, end = e.z;
      const uint32_t rw = ld_rowword(a.uw + u);
      const uint32_t r = rw & kRowMask;
      const KeySums s = upd_tile_sums<L, DB, HAS_VAL>(a, beg, end, 0u, 1u, grp, sub, sub_ok, k, kp);
      // the key's row is fetched after the sums: one more round trip for a segment of 9+ occurrences,
      // 12 registers fewer alive through the tile
      if (grp == 0) {
        float4 h0, vv, ac;
        float fc;
        upd_load<MIXED>(a, rw, sub, sub_ok, kp, h0, vv, ac, fc);
        upd_apply<EXACT, MIXED>(a, r, u, h0, vv, ac, s.gw, s.xxp, s.gv, sub, sub_ok, k, kp, pen, fc, (float)(end - beg),
                                (rw & kCountLater) != 0u, (rw & kRemoteRow) != 0u);
      }
    }
  }
}

// ---- few: one L-lane group per key, occurrences in row order
template <int L, bool EXACT, bool HAS_VAL, bool MIXED>
__device__ __forceinline__ void upd_few_role(const UpdArgs& a, uint32_t wave, uint32_t nwaves, float& pen) {
  constexpr int G = 64 / L;
  constexpr int FD = DFH_UPD_FEW_DEPTH;
  const int lane = lane_id();
  const int grp = lane / L, sub = lane % L;
  const int kp = EXACT ? 4 * L : a.kp, k = a.k;
  const bool sub_ok = EXACT ? true : (sub * 4 < kp);
  const uint32_t lane_off = (sub_ok ? sub * 16 : 0), row_bytes = (uint32_t)kp * 4u;
  UpdTeam tm;
  if (!upd_team(wave, nwaves, a.nlist, tm)) return;
  for (uint32_t lb = tm.first; lb < a.nlist; lb += tm.stride) {
    const uint2 co = a.seg.few[lb];
    const uint32_t n = co.x;
    if (n == 0) continue;
    const SegEnt* __restrict__ ent = a.seg.few_ent + co.y;
    for (uint32_t q0 = tm.sub * G; q0 < n; q0 += tm.size * G) {
      // round trip 1: the entry {key rank, segment}
      const uint32_t q = q0 + grp;
      const SegEnt e = ent[min(q, n - 1)];
      const uint32_t u = e.x;
      const bool act = q < n && key_in(a.rg, u);
      const uint32_t beg = e.y;
      const uint32_t len_all = e.z - beg;
      const uint32_t len = act ? len_all : 0u;
      const uint32_t last = beg + max(len_all, 1u) - 1u;
      // round trip 2: the row word — and, with it, the first occurrences (the segment is known already)
      const uint32_t rw = ld_rowword(a.uw + u);
      // a group without a key of its own (list exhausted, key of another rank) reads row 0 and drops it
      const uint32_t r = (act && (MIXED || (rw & kRemoteRow) == 0u)) ? (rw & kRowMask) : 0u;
      // round trip 3: the model row (beside the slopes and XV rows of the first occurrences)
      float4 h0, vv, ac;
      float fc;
      upd_load<MIXED>(a, act ? rw : 0u, sub, sub_ok, kp, h0, vv, ac, fc);
      float gw = 0.f, xxp = 0.f;
      float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (uint32_t j0 = 0; __ballot(j0 < len) != 0ull; j0 += FD) {
        uint32_t rows[FD];
        float xs[FD];
#pragma unroll
        for (int d = 0; d < FD; ++d) {  // occurrences past the segment: its last one again, factor 0
          const uint32_t jj = min(beg + j0 + d, last);
          rows[d] = ldu_s(a.s_row + jj);
          xs[d] = (HAS_VAL ? ldf_s(a.s_val + jj) : 1.0f) * upd_mask(j0 + d < len);
        }
        float ps[FD];
        float4 av[FD];
#pragma unroll
        for (int d = 0; d < FD; ++d) {  // round trip 3 (+): slopes and XV rows of these occurrences
          ps[d] = a.slope[rows[d]];
          av[d] = ld4_off(a.xv, rows[d] * row_bytes + lane_off);
        }
#pragma unroll
        for (int d = 0; d < FD; ++d) {  // ascending rows: the reference's order (spmm.h:137-156)
          const float pp = ps[d], xx = xs[d];
          gw = fma_skip0(pp, xx, gw);
          xxp = fma_skip0(pp, xx * xx, xxp);
          g4.x += (av[d].x * pp) * xx; g4.y += (av[d].y * pp) * xx;
          g4.z += (av[d].z * pp) * xx; g4.w += (av[d].w * pp) * xx;
        }
      }
      if (act) upd_apply<EXACT, MIXED>(a, r, u, h0, vv, ac, gw, xxp, g4, sub, sub_ok, k, kp, pen, fc, (float)len_all, (rw & kCountLater) != 0u,
                                       (rw & kRemoteRow) != 0u);
    }
  }
}

// ---- singles: example by example
// one tile of <= 64 nonzeros of example i: lane l holds nonzero base + l (u = rank of its key, x its value, rw its row
// word, valid = it exists); p = the example's slope, xvi = this lane's slice of XV_i.  Shared by the singles role of
// k_update_fused (and, in the round-4 experiment noted at the end of this file, by the forward's epilogue).
template <int L, bool EXACT, int RB, bool MIXED = false>
__device__ __forceinline__ void upd_singles_tile(const UpdArgs& a, bool valid, uint32_t u, float x, uint32_t rw, float p, const float4 xvi,
                                                 int grp, int sub, bool sub_ok, int k, int kp, float& pen) {
  constexpr int G = 64 / L;
  const int lane = lane_id();
  const bool single = valid && (MIXED ? (rw & kSingleRow) != 0u : (rw & (kSingleRow | kRemoteRow)) == kSingleRow) && key_in(a.rg, u);
  const unsigned long long mask = __ballot(single);
  const int n1 = __popcll(mask);
  if (n1 == 0) return;
  // compact the single-occurrence nonzeros to the low lanes (a permutation of the wave: the
  // others are packed behind them)
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  const int dest = (single ? rank : n1 + (lane - rank)) * 4;
  const uint32_t c_r = (uint32_t)__builtin_amdgcn_ds_permute(dest, (int)(rw & (kRowMask | kCountLater | (MIXED ? kRemoteRow : 0u))));
  const uint32_t c_u = (uint32_t)__builtin_amdgcn_ds_permute(dest, (int)u);
  const float c_x = __int_as_float(__builtin_amdgcn_ds_permute(dest, __float_as_int(x)));
  for (int t0 = 0; t0 < n1; t0 += RB * G) {
    float4 h0[RB], vv[RB], ac[RB];
    uint32_t rr[RB], uu[RB];
    float xs[RB], fcs[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {  // RB * G model rows per round trip; groups past the last key fetch it again
      const int t = min(t0 + q * G + grp, n1 - 1);
      rr[q] = __shfl(c_r, t, 64);
      uu[q] = __shfl(c_u, t, 64);
      xs[q] = __shfl(c_x, t, 64);
      upd_load<MIXED>(a, rr[q], sub, sub_ok, kp, h0[q], vv[q], ac[q], fcs[q]);
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      if (t0 + q * G + grp < n1) {
        const float xx = xs[q];
        // one occurrence: the segmented sums of the other roles with a single term (0 + term)
        const float gw = p != 0.f ? p * xx : 0.f, xxp = p != 0.f ? p * (xx * xx) : 0.f;  // spmv.h:155
        const float4 g4 = make_float4((xvi.x * p) * xx, (xvi.y * p) * xx, (xvi.z * p) * xx, (xvi.w * p) * xx);
        upd_apply<EXACT, MIXED>(a, rr[q] & kRowMask, uu[q], h0[q], vv[q], ac[q], gw, xxp, g4, sub, sub_ok, k, kp, pen, fcs[q], 1.0f,
                                (rr[q] & kCountLater) != 0u, (rr[q] & kRemoteRow) != 0u);
      }
    }
  }
}

template <int L, bool EXACT, int RB, bool HAS_VAL, bool MIXED>
__device__ __forceinline__ void upd_singles_role(const UpdArgs& a, uint32_t wave, uint32_t nwaves, float& pen) {
  const int lane = lane_id();
  const int grp = lane / L, sub = lane % L;
  const int kp = EXACT ? 4 * L : a.kp, k = a.k;
  const bool sub_ok = EXACT ? true : (sub * 4 < kp);
  for (uint32_t i = wave; i < a.nrows; i += nwaves) {
    const uint32_t beg = a.offset[i], end = a.offset[i + 1];
    if (beg == end) continue;
    // the example's slope and its XV slice: once per example, coalesced
    const float p = a.slope[i];
    const float4 xvi = ld4_off(a.xv, i * ((uint32_t)kp * 4u) + (sub_ok ? sub * 16 : 0));
    for (uint32_t base = beg; base < end; base += 64u) {
      const bool valid = base + lane < end;
      const uint32_t j = min(base + lane, end - 1);
      const uint32_t u = ldu_s(a.index + j);
      const float x = HAS_VAL ? ldf_s(a.value + j) : 1.0f;
      const uint32_t rw = ld_rowword(a.uw + u);
      upd_singles_tile<L, EXACT, RB, MIXED>(a, valid, u, x, rw, p, xvi, grp, sub, sub_ok, k, kp, pen);
    }
  }
}

// penalty of the pulled weights (sgd_learner.cc:249-273): per-lane fp32 partials (a handful of terms each),
// widened here; one private slot per block (same-address atomics serialise)
__device__ __forceinline__ void upd_flush_penalty(double* prog, float pen, const uint32_t blk) {
  __shared__ double pen_blk[UPD_NW];
  const uint32_t w = threadIdx.x >> 6;
  const double pw = wave_sum_d((double)pen);
  if (lane_id() == 0) pen_blk[w] = pw;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < UPD_NW; ++i) t += pen_blk[i];
    if (t != 0.0) atomicAdd(&prog[PROG_PENALTY * PROG_SLOTS + (blk % PROG_SLOTS)], t);
  }
}

// block `bid` of the update's `nblk` blocks (the whole launch of k_update_fused; a block range of a launch that also
// carries riders: dfh_riders.hip).  smem: max(AUC_SMEM, UPD_HOT_SMEM) bytes — a block is an AUC unit or a role, never both.
constexpr size_t UPD_SMEM = AUC_SMEM > UPD_HOT_SMEM ? AUC_SMEM : UPD_HOT_SMEM;
template <int L, bool EXACT, bool HAS_VAL, bool MIXED>
__device__ __forceinline__ void update_body(const UpdArgs& a, const uint32_t blk_in, const uint32_t nblk, char* smem) {
  if (blk_in < a.nb_auc) {  // uniform per block; nb_auc is auc_units(nrows) rounded up to 8 (see the XCD note below)
    if (blk_in < auc_units(a.nrows)) auc_pairs_block(a.auc_pred, a.auc_label, a.nrows, blk_in, a.auc_part, smem);
    return;
  }
  if (threadIdx.x == 0) upd_init_list()[0] = 0u;
  __syncthreads();
  float pen = 0.f;
  const uint32_t w = threadIdx.x >> 6;

  // block -> role.  The list roles (hot, mid, few) are chains of dependent round trips on few bytes, the singles
  // stream most of the launch's HBM traffic: dispatched in that order the former would hold every block slot of
  // the chip for the length of their chains before the first model row moves.  Interleaved 1 : (R - 1) both kinds
  // are resident from the start.
  const uint32_t nb_list = a.nb_hot + a.nb_mid + a.nb_few, nb_single = nblk - a.nb_auc - nb_list;
  uint32_t bid = blk_in - a.nb_auc;
  bool list_role = bid < nb_list;
  if (a.ileave > 1u) {
    // groups of [8 list blocks, 8 (R - 1) singles blocks] while both kinds last, then the rest: list blocks, singles blocks.
    // In units of EIGHT because workgroups go round-robin over the 8 XCDs and the singles role wants its block b on the XCD
    // that ran block b of the forward (the XV rows and slopes of examples 4b .. 4b + 3 are in THAT L2): singles block s
    // must sit at a launch index congruent to s modulo 8 (round 3's interleave dealt single blocks and broke that).  Measured
    // in round 4 with the alignment kept: 84.0 (R = 2) / 83.4 (R = 3) against 85.9 M examples/sec — list roles first stays.
    const uint32_t R = a.ileave, G = 8u * R, P = min(nb_list / 8u, nb_single / (8u * (R - 1u)));
    if (bid < P * G) {
      const uint32_t g = bid / G, j = bid % G;
      list_role = j < 8u;
      bid = list_role ? g * 8u + j : g * 8u * (R - 1u) + (j - 8u);
    } else {
      const uint32_t rem = bid - P * G, left = nb_list - P * 8u;
      list_role = rem < left;
      bid = list_role ? P * 8u + rem : P * 8u * (R - 1u) + (rem - left);
    }
  } else if (!list_role) {
    bid -= nb_list;
  }
  if (!list_role) {
    if (DFH_UPD_ROLES & 8)
    upd_singles_role<L, EXACT, DFH_UPD_ROUNDS, HAS_VAL, MIXED>(a, bid * UPD_NW + w, nb_single * UPD_NW, pen);
  } else if (bid < a.nb_hot) {
    if (DFH_UPD_ROLES & 1) {
      // the parts of the very hot keys first — the longest chains of the launch, taken by the hot role's own blocks (a list that is
      // empty on most data: one load per block) — then the hot keys
      if ((DFH_HOT_SPLIT_BUILD & 2) && a.nb_split) upd_split_role<L, EXACT, DFH_UPD_DEPTH, HAS_VAL, MIXED>(a, bid, a.nb_hot, pen, smem);
      upd_hot_role<L, EXACT, DFH_UPD_DEPTH, HAS_VAL, MIXED>(a, bid, a.nb_hot, pen, smem);
    }
  } else if ((bid -= a.nb_hot) < a.nb_mid) {
    if (DFH_UPD_ROLES & 2)
    upd_mid_role<L, EXACT, DFH_UPD_DEPTH, HAS_VAL, MIXED>(a, bid * UPD_NW + w, a.nb_mid * UPD_NW, pen);
  } else {
    bid -= a.nb_mid;
    if (DFH_UPD_ROLES & 4)
    upd_few_role<L, EXACT, HAS_VAL, MIXED>(a, bid * UPD_NW + w, a.nb_few * UPD_NW, pen);
  }
  upd_init_rows(a, L);
  upd_flush_penalty(a.prog, pen, blk_in);
}

template <int L, bool EXACT, bool HAS_VAL, bool MIXED = false>
__global__ void __launch_bounds__(UPD_THREADS, DFH_UPD_WAVES) k_update_fused(UpdArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[UPD_SMEM];
  update_body<L, EXACT, HAS_VAL, MIXED>(a, blockIdx.x, gridDim.x, smem);
}

// Measured dead end (round 4; profiles/r04a_fwd_singles_ab.txt): the singles role as the EPILOGUE OF THE
// FORWARD (k_forward_singles: the wave that owns example i updates the example's single-occurrence keys right after its
// gather, holding p_i and XV_i in registers; upd_singles_tile on the same values, the list roles left to k_update_fused).
// Bit-identical, built, removed: 86.9 -> 79.6 M examples/sec (83.5 M at 4 waves per SIMD).  (1) The V lines are NOT
// L2 hits: the 640 waves resident on an XCD gather 10 KB each = 6.4 MB through a 4 MB L2, so the fused kernel read
// 153.9 MB against 72.8 (forward) + 45 (accumulator and header lines) = 118 expected; forward + update reads only went
// 205.1 -> 197.8 MB.  (2) The list roles alone are a 33 us chain of dependent round trips (59 MB at 1.8 TB/s) that used
// to hide under the singles' streaming: 57.5 + 32.8 us serial against 19.9 + 59.3 us.
}  // namespace dfh
#endif  // DFH_UPDATE_HIP_
