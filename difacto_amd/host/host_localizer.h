/**
 * host_localizer.h — Localizer::Compact on the host, for the literal
 * (host-array) path and for tests.  Same contract as the reference's
 * src/data/localizer.{h,cc}: keys k = ReverseBytes(id % max_index); the unique
 * keys come out ascending with their occurrence counts; every nnz is remapped
 * to the rank of its key; the CSR shape is preserved.
 */
#ifndef DIFACTO_HOST_HOST_LOCALIZER_H_
#define DIFACTO_HOST_HOST_LOCALIZER_H_
#include <algorithm>
#include <limits>
#include <vector>
#include "data/row_block.h"
#include "difacto/base.h"

namespace difacto {

class Localizer {
 public:
  explicit Localizer(feaid_t max_index = std::numeric_limits<feaid_t>::max(), int nthreads = DEFAULT_NTHREADS)
      : max_index_(max_index) { (void)nthreads; }

  void Compact(const dmlc::RowBlock<feaid_t>& blk, dmlc::data::RowBlockContainer<unsigned>* compacted,
               std::vector<feaid_t>* uniq_idx = nullptr, std::vector<real_t>* idx_frq = nullptr) {
    CHECK_NOTNULL(compacted);
    compacted->Clear();
    if (uniq_idx) uniq_idx->clear();
    if (idx_frq) idx_frq->clear();
    if (blk.size == 0) return;
    const size_t base = blk.offset[0], nnz = blk.offset[blk.size] - base;
    CHECK_LT(nnz, static_cast<size_t>(std::numeric_limits<unsigned>::max()));
    std::vector<std::pair<feaid_t, unsigned>> kv(nnz);
    for (size_t i = 0; i < nnz; ++i) kv[i] = {ReverseBytes(blk.index[base + i] % max_index_), static_cast<unsigned>(i)};
    std::sort(kv.begin(), kv.end());
    compacted->index.resize(nnz);
    unsigned rank = 0;
    std::vector<feaid_t> local_u;
    std::vector<feaid_t>* u = uniq_idx ? uniq_idx : &local_u;
    for (size_t i = 0; i < nnz; ++i) {
      if (i == 0 || kv[i].first != kv[i - 1].first) {
        u->push_back(kv[i].first);
        if (idx_frq) idx_frq->push_back(0);
        rank = static_cast<unsigned>(u->size() - 1);
      }
      if (idx_frq) idx_frq->back() += 1;
      compacted->index[kv[i].second] = rank;
    }
    compacted->offset.resize(blk.size + 1);
    for (size_t i = 0; i <= blk.size; ++i) compacted->offset[i] = blk.offset[i] - base;
    if (blk.value) compacted->value.assign(blk.value + base, blk.value + base + nnz);
    if (blk.label) compacted->label.assign(blk.label, blk.label + blk.size);
    compacted->max_index = u->empty() ? 0 : static_cast<unsigned>(u->size() - 1);
  }

 private:
  feaid_t max_index_;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_HOST_LOCALIZER_H_
