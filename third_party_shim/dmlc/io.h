// Minimal stand-in for dmlc-core's io.h: the abstract byte Stream used by
// Updater::Load/Save plus a local-file implementation (no hdfs/s3).
#ifndef SHIM_DMLC_IO_H_
#define SHIM_DMLC_IO_H_
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "./logging.h"

namespace dmlc {

class Stream {
 public:
  virtual size_t Read(void* ptr, size_t size) = 0;
  virtual void Write(const void* ptr, size_t size) = 0;
  virtual ~Stream() {}
  /*! \brief open a local file; flag is "r", "w" or "a"; nullptr if allow_null and it fails */
  static inline Stream* Create(const char* uri, const char* const flag, bool allow_null = false);

  template <typename T>
  inline void WritePOD(const T& v) { Write(&v, sizeof(T)); }
  template <typename T>
  inline bool ReadPOD(T* v) { return Read(v, sizeof(T)) == sizeof(T); }
};

/*! \brief Stream with random access */
class SeekStream : public Stream {
 public:
  virtual void Seek(size_t pos) = 0;
  virtual size_t Tell() = 0;
};

namespace shim {
class FileStream : public SeekStream {
 public:
  explicit FileStream(FILE* fp) : fp_(fp) {}
  ~FileStream() override { if (fp_) fclose(fp_); }
  size_t Read(void* ptr, size_t size) override { return fread(ptr, 1, size, fp_); }
  void Write(const void* ptr, size_t size) override {
    CHECK_EQ(fwrite(ptr, 1, size, fp_), size) << "short write";
  }
  void Seek(size_t pos) override { fseek(fp_, static_cast<long>(pos), SEEK_SET); }
  size_t Tell() override { return static_cast<size_t>(ftell(fp_)); }
 private:
  FILE* fp_;
};
}  // namespace shim

inline Stream* Stream::Create(const char* uri, const char* const flag, bool allow_null) {
  std::string path(uri);
  if (path.compare(0, 7, "file://") == 0) path = path.substr(7);
  std::string mode(flag);
  if (mode.find('b') == std::string::npos) mode += "b";
  FILE* fp = fopen(path.c_str(), mode.c_str());
  if (fp == nullptr) {
    CHECK(allow_null) << "cannot open " << uri << " with mode " << flag;
    return nullptr;
  }
  return new shim::FileStream(fp);
}

}  // namespace dmlc
#endif  // SHIM_DMLC_IO_H_
