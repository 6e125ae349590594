// Minimal stand-in for dmlc-core's data.h: the CSR minibatch view types the
// FM/SGD path passes around (dmlc-core is an absent submodule of the
// reference).  Written from scratch from the field usage in the reference:
//   RowBlock<I>{size, offset, label, weight, index, value}, operator[] -> Row<I>
#ifndef SHIM_DMLC_DATA_H_
#define SHIM_DMLC_DATA_H_
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include "./logging.h"

namespace dmlc {

typedef float real_t;
typedef unsigned index_t;

/*! \brief one sparse example: a view into a RowBlock */
template <typename IndexType>
struct Row {
  real_t label;
  real_t weight;
  size_t length;
  const IndexType* index;
  const real_t* value;  // may be nullptr => all ones
  inline IndexType get_index(size_t i) const { return index[i]; }
  inline real_t get_value(size_t i) const { return value == nullptr ? 1.0f : value[i]; }
  template <typename V>
  inline V SDot(const V* w, size_t size) const {
    V s = 0;
    for (size_t i = 0; i < length; ++i) {
      CHECK(index[i] < size);
      s += (value == nullptr ? w[index[i]] : w[index[i]] * value[i]);
    }
    return s;
  }
};

/*! \brief a batch of sparse examples in CSR form; does not own memory */
template <typename IndexType>
struct RowBlock {
  size_t size = 0;
  const size_t* offset = nullptr;
  const real_t* label = nullptr;
  const real_t* weight = nullptr;
  const IndexType* index = nullptr;
  const real_t* value = nullptr;

  inline Row<IndexType> operator[](size_t i) const {
    CHECK(i < size);
    Row<IndexType> r;
    r.label = label ? label[i] : 0;
    r.weight = weight ? weight[i] : 1.0f;
    r.length = offset[i + 1] - offset[i];
    r.index = index + offset[i];
    r.value = value ? value + offset[i] : nullptr;
    return r;
  }
  inline size_t MemCostBytes() const {
    size_t nnz = offset[size] - offset[0];
    size_t c = (size + 1) * sizeof(size_t) + nnz * sizeof(IndexType);
    if (label) c += size * sizeof(real_t);
    if (weight) c += size * sizeof(real_t);
    if (value) c += nnz * sizeof(real_t);
    return c;
  }
  inline RowBlock Slice(size_t begin, size_t end) const {
    CHECK(begin <= end && end <= size);
    RowBlock r;
    r.size = end - begin;
    r.offset = offset + begin;
    r.label = label ? label + begin : nullptr;
    r.weight = weight ? weight + begin : nullptr;
    r.index = index;
    r.value = value;
    return r;
  }
};

}  // namespace dmlc
#endif  // SHIM_DMLC_DATA_H_
