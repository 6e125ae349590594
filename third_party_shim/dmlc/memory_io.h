// Minimal stand-in for dmlc-core's memory_io.h: a Stream over a std::string.
#ifndef SHIM_DMLC_MEMORY_IO_H_
#define SHIM_DMLC_MEMORY_IO_H_
#include <algorithm>
#include <string>
#include "./io.h"
namespace dmlc {
class MemoryStringStream : public SeekStream {
 public:
  explicit MemoryStringStream(std::string* buf) : buf_(buf), pos_(0) {}
  size_t Read(void* ptr, size_t size) override {
    size_t n = std::min(size, buf_->size() - pos_);
    if (n) memcpy(ptr, buf_->data() + pos_, n);
    pos_ += n;
    return n;
  }
  void Write(const void* ptr, size_t size) override {
    if (pos_ + size > buf_->size()) buf_->resize(pos_ + size);
    if (size) memcpy(&(*buf_)[pos_], ptr, size);
    pos_ += size;
  }
  void Seek(size_t pos) override { pos_ = pos; }
  size_t Tell() override { return pos_; }
 private:
  std::string* buf_;
  size_t pos_;
};
}  // namespace dmlc
#endif  // SHIM_DMLC_MEMORY_IO_H_
