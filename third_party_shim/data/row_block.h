// Minimal stand-in for dmlc-core's src/data/row_block.h (absent submodule):
// an owning CSR builder.  Written from scratch; semantics taken from how the
// reference uses it (BatchReader/Localizer): a fresh container has
// offset == {0}; Push(Row)/Push(RowBlock) append; GetBlock() returns a view
// whose optional arrays are nullptr when empty.
#ifndef SHIM_DATA_ROW_BLOCK_H_
#define SHIM_DATA_ROW_BLOCK_H_
#include <algorithm>
#include <memory>
#include <new>
#include <utility>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "dmlc/data.h"

namespace dmlc {
namespace data {

// resize(n) without a value leaves the new elements uninitialised: the readers size an array and then fill every element
// of it (a gather of 3 MB per minibatch was being zero-filled first)
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
  template <typename U> struct rebind { typedef DefaultInitAllocator<U> other; };
  DefaultInitAllocator() = default;
  template <typename U> DefaultInitAllocator(const DefaultInitAllocator<U>&) {}
  template <typename U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};

template <typename IndexType>
struct RowBlockContainer {
  std::vector<size_t> offset;
  std::vector<real_t> label;
  std::vector<real_t> weight;
#ifdef DMLC_SHIM_STD_VECTORS
  // oracle/_ref: the reference's own headers take these arrays as plain std::vector<T>* (compressed_row_block.h:56-75)
  std::vector<IndexType> index;
  std::vector<real_t> value;
#else
  std::vector<IndexType, DefaultInitAllocator<IndexType>> index;
  std::vector<real_t, DefaultInitAllocator<real_t>> value;
#endif
  /*! \brief the largest index stored — exact after Push(Row) and after an assignment (Localizer::RemapIndex sets it,
   *  src/data/localizer.cc:102); Push(RowBlock) and bulk appenders only mark it stale (the scan was a second pass over
   *  every byte the shuffle buffer's assembly copies): read it through GetMaxIndex(), which rescans when needed */
  IndexType max_index;
  bool max_index_stale;

  RowBlockContainer() { Clear(); }

  /*! \brief max_index, recomputed first if bulk appends have left it stale */
  inline IndexType GetMaxIndex() {
    if (max_index_stale) {
      IndexType m = 0;
      for (const IndexType& id : index) m = std::max(m, id);
      max_index = m;
      max_index_stale = false;
    }
    return max_index;
  }

  inline void Clear() {
    offset.assign(1, 0);
    label.clear();
    weight.clear();
    index.clear();
    value.clear();
    max_index = 0;
    max_index_stale = false;
  }
  inline size_t Size() const { return offset.size() - 1; }
  inline size_t MemCostBytes() const {
    return offset.size() * sizeof(size_t) + (label.size() + weight.size() + value.size()) * sizeof(real_t) +
           index.size() * sizeof(IndexType);
  }

  /*! \brief append one example */
  template <typename I>
  inline void Push(Row<I> row) {
    label.push_back(row.label);
    if (!weight.empty() || row.weight != 1.0f) {
      // keep weights dense once any non-unit weight shows up
      weight.resize(label.size() - 1, 1.0f);
      weight.push_back(row.weight);
    }
    for (size_t i = 0; i < row.length; ++i) {
      IndexType id = static_cast<IndexType>(row.index[i]);
      index.push_back(id);
      max_index = std::max(max_index, id);
    }
    if (row.value != nullptr) {
      value.insert(value.end(), row.value, row.value + row.length);
    }
    offset.push_back(index.size());
  }

  /*! \brief append a whole block (block.index/value point at its first nnz) */
  template <typename I>
  inline void Push(RowBlock<I> blk) {
    if (blk.size == 0) return;
    size_t nnz = blk.offset[blk.size] - blk.offset[0];
    if (blk.label != nullptr) label.insert(label.end(), blk.label, blk.label + blk.size);
    if (blk.weight != nullptr) weight.insert(weight.end(), blk.weight, blk.weight + blk.size);
    // one pass (converts when I != IndexType), no zero fill.  max_index goes stale instead of being rescanned here:
    // nothing on the worker path reads it (the Localizer is given its modulus explicitly, src/sgd/sgd_learner.cc:203)
    max_index_stale = max_index_stale || nnz > 0;
    if (std::is_same<I, IndexType>::value && nnz >= (size_t(1) << 18) && BulkCopyThreads() > 1) {
      // a shuffle buffer is assembled out of slices of several MB each: the copy is split over a few threads (one core
      // copies ~15 GB/s; the assembly of the 31 MB buffers was what the worker loop waited for)
      const size_t at = index.size();
      index.resize(at + nnz);
      if (blk.value != nullptr) value.resize(at + nnz);
      const int nt = BulkCopyThreads();
#pragma omp parallel for num_threads(nt) schedule(static)
      for (int t = 0; t < nt; ++t) {
        const size_t lo = nnz * t / nt, hi = nnz * (t + 1) / nt;
        memcpy(static_cast<void*>(&index[at + lo]), static_cast<const void*>(blk.index + lo), (hi - lo) * sizeof(IndexType));
        if (blk.value != nullptr) memcpy(&value[at + lo], blk.value + lo, (hi - lo) * sizeof(real_t));
      }
    } else {
      index.insert(index.end(), blk.index, blk.index + nnz);
      if (blk.value != nullptr) value.insert(value.end(), blk.value, blk.value + nnz);
    }
    size_t shift = offset.back();
    for (size_t i = 0; i < blk.size; ++i) {
      offset.push_back(shift + blk.offset[i + 1] - blk.offset[0]);
    }
  }

  /*! \brief threads of a bulk append (DIFACTO_ASSEMBLY_THREADS, default 4, 1 .. 16) */
  static int BulkCopyThreads() {
    static const int n = [] {
      const char* e = getenv("DIFACTO_ASSEMBLY_THREADS");
      return std::max(1, std::min(e ? atoi(e) : 4, 16));
    }();
    return n;
  }

  /*! \brief view of the stored rows */
  inline RowBlock<IndexType> GetBlock() const {
    if (!label.empty()) CHECK_EQ(label.size() + 1, offset.size());
    CHECK_EQ(offset.back(), index.size());
    CHECK(value.empty() || value.size() == offset.back());
    RowBlock<IndexType> b;
    b.size = offset.size() - 1;
    b.offset = offset.data();
    b.label = label.empty() ? nullptr : label.data();
    b.weight = weight.empty() ? nullptr : weight.data();
    b.index = index.empty() ? nullptr : index.data();
    b.value = value.empty() ? nullptr : value.data();
    return b;
  }
};

}  // namespace data
}  // namespace dmlc
#endif  // SHIM_DATA_ROW_BLOCK_H_
