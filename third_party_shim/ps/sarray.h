// Minimal stand-in for ps-lite's sarray.h (ps-lite is an absent, un-vendored
// submodule of the reference).  Written from scratch to the behaviour the
// reference's code relies on:
//   * shared ownership: copy/assign share the buffer (pointer semantics)
//   * SArray(n, v=0) fills; resize(n, v=0) keeps the old prefix, fills the tail
//   * CopyFrom(ptr,n) / CopyFrom(SArray) deep-copy
//   * SArray<A>(SArray<B>) reinterprets the same bytes (size scaled)
//   * SArray(shared_ptr<vector<V>>) shares the vector's storage
//   * reset(ptr, n, deleter) adopts foreign memory
//   * segment(b, e) returns a sharing sub-array
#ifndef SHIM_PS_SARRAY_H_
#define SHIM_PS_SARRAY_H_
#include <cstring>
#include <initializer_list>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include "dmlc/logging.h"

namespace ps {

template <typename V>
class SArray {
 public:
  SArray() {}
  ~SArray() {}

  /*! \brief n elements, each set to val */
  explicit SArray(size_t size, V val = 0) { resize(size, val); }

  /*! \brief zero-copy reinterpretation of another array's bytes */
  template <typename W>
  explicit SArray(const SArray<W>& arr) { *this = arr; }

  template <typename W>
  void operator=(const SArray<W>& arr) {
    size_ = arr.size() * sizeof(W) / sizeof(V);
    CHECK_EQ(size_ * sizeof(V), arr.size() * sizeof(W)) << "cannot be divided";
    capacity_ = arr.capacity() * sizeof(W) / sizeof(V);
    ptr_ = std::shared_ptr<V>(arr.ptr(), reinterpret_cast<V*>(arr.data()));
  }

  /*! \brief adopt (or merely view, if !deletable) a C array */
  SArray(V* data, size_t size, bool deletable = false) {
    if (deletable) {
      reset(data, size, [](V* p) { delete[] p; });
    } else {
      reset(data, size, [](V*) {});
    }
  }

  /*! \brief deep copy of a std::vector */
  explicit SArray(const std::vector<V>& vec) { CopyFrom(vec.data(), vec.size()); }

  /*! \brief share a std::vector's storage */
  explicit SArray(const std::shared_ptr<std::vector<V>>& vec) {
    ptr_ = std::shared_ptr<V>(vec, vec->data());
    size_ = vec->size();
    capacity_ = size_;
  }

  SArray(const std::initializer_list<V>& list) {
    if (list.size()) CopyFrom(list.begin(), list.size());
  }
  void operator=(const std::initializer_list<V>& list) {
    clear();
    if (list.size()) CopyFrom(list.begin(), list.size());
  }

  template <typename Deleter>
  void reset(V* data, size_t size, Deleter del) {
    size_ = size;
    capacity_ = size;
    ptr_.reset(data, del);
  }

  void CopyFrom(const V* data, size_t size) {
    V* buf = new V[size + 1];  // +1 keeps data() non-null for size 0
    if (size) memcpy(buf, data, size * sizeof(V));
    reset(buf, size, [](V* p) { delete[] p; });
  }
  void CopyFrom(const SArray<V>& other) {
    if (this == &other) return;
    CopyFrom(other.data(), other.size());
  }
  template <typename ForwardIt>
  void CopyFrom(const ForwardIt& first, const ForwardIt& last) {
    size_t n = static_cast<size_t>(std::distance(first, last));
    V* buf = new V[n + 1];
    reset(buf, n, [](V* p) { delete[] p; });
    V* d = buf;
    for (auto it = first; it != last; ++it) *d++ = *it;
  }

  void resize(size_t size, V val = 0) {
    size_t cur = size_;
    if (capacity_ >= size) {
      size_ = size;
    } else {
      V* buf = new V[size + 5];
      if (size_) memcpy(buf, data(), size_ * sizeof(V));
      reset(buf, size, [](V* p) { delete[] p; });
      capacity_ = size + 5;
    }
    if (size > cur) {
      V* p = data() + cur;
      if (val == 0) {
        memset(p, 0, (size - cur) * sizeof(V));
      } else {
        for (size_t i = 0; i < size - cur; ++i) p[i] = val;
      }
    }
  }
  void reserve(size_t size) {
    if (capacity_ >= size) return;
    size_t old = size_;
    resize(size);
    size_ = old;
  }
  void clear() { reset(static_cast<V*>(nullptr), 0, [](V*) {}); }

  inline bool empty() const { return size() == 0; }
  inline size_t size() const { return size_; }
  inline size_t capacity() const { return capacity_; }

  inline V* begin() { return data(); }
  inline const V* begin() const { return data(); }
  inline V* end() { return data() + size(); }
  inline const V* end() const { return data() + size(); }
  inline V* data() const { return ptr_.get(); }
  inline std::shared_ptr<V>& ptr() { return ptr_; }
  inline const std::shared_ptr<V>& ptr() const { return ptr_; }

  inline V back() const { CHECK(!empty()); return data()[size_ - 1]; }
  inline V front() const { CHECK(!empty()); return data()[0]; }
  inline V& operator[](size_t i) { return data()[i]; }
  inline const V& operator[](size_t i) const { return data()[i]; }

  inline void push_back(const V& val) {
    if (size_ == capacity_) reserve(size_ * 2 + 5);
    data()[size_++] = val;
  }
  void pop_back() { if (size_) --size_; }
  void append(const SArray<V>& arr) {
    if (arr.empty()) return;
    size_t old = size_;
    resize(size_ + arr.size());
    memcpy(data() + old, arr.data(), arr.size() * sizeof(V));
  }

  /*! \brief [begin, end) sharing this array's storage */
  SArray<V> segment(size_t begin, size_t end) const {
    CHECK_GE(end, begin);
    CHECK_LE(end, size());
    SArray<V> ret;
    ret.ptr_ = std::shared_ptr<V>(ptr_, data() + begin);
    ret.size_ = end - begin;
    ret.capacity_ = end - begin;
    return ret;
  }

 private:
  template <typename W> friend class SArray;
  size_t size_ = 0;
  size_t capacity_ = 0;
  std::shared_ptr<V> ptr_;
};

template <typename V>
std::ostream& operator<<(std::ostream& os, const SArray<V>& a) {
  os << "[" << a.size() << "]:";
  size_t n = a.size() < 10 ? a.size() : 10;
  for (size_t i = 0; i < n; ++i) os << " " << a[i];
  if (n < a.size()) os << " ...";
  return os;
}

}  // namespace ps
#endif  // SHIM_PS_SARRAY_H_
