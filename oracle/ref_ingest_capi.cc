/**
 * oracle/_ref C API, data-format half  —  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The REFERENCE'S OWN on-disk format code, compiled where it lies under /root/reference:
 *   difacto::CompressedRowBlock   src/data/compressed_row_block.h:20-142  (DIFACTO_USE_LZ4=1, the image's liblz4)
 *   difacto::CriteoParser         src/reader/criteo_parser.h:40-101       (DIFACTO_USE_CITY=1; CityHash64 is served
 *                                 by oracle/city_checker.cc — the library is absent, see ref_shim/city.h)
 *   difacto::AdfeaParser          src/reader/adfea_parser.h:33-88         (dmlc-core's strtonum.h helpers: ref_shim/data/strtonum.h)
 * against the interface stand-ins in ref_shim/ (dmlc::InputSplit, dmlc::data::ParserImpl).
 * Used by tests/ and tools/make_golden_ingest.py to pin the product's .rec decoder and criteo parser
 * (difacto_amd/host/batch_reader.h) and the Python writers in oracle/ingest.py.  The RecordIO framing around
 * the compressed blocks is dmlc-core's (absent): that layer stays checked against oracle/ingest.py only.
 */
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "difacto/base.h"
#include "data/compressed_row_block.h"
#include "reader/criteo_parser.h"
#include "reader/adfea_parser.h"

using namespace difacto;

namespace {
// the whole text as ONE chunk, handed out once
class MemSplit : public dmlc::InputSplit {
 public:
  MemSplit(const char* p, size_t n) : buf_(p, p + n), done_(false) {}
  void BeforeFirst() override { done_ = false; }
  bool NextChunk(Blob* out) override {
    if (done_ || buf_.empty()) return false;
    out->dptr = buf_.data();
    out->size = buf_.size();
    done_ = true;
    return true;
  }

 private:
  std::vector<char> buf_;
  bool done_;
};
}  // namespace

extern "C" {

/** CompressedRowBlock::Compress<feaid_t> (:23-50).  Returns the record's size, or -1 when `cap` is too small. */
long ref_crb_compress(size_t nrows, const size_t* offset, const float* label, const uint64_t* index, const float* value,
                      const float* weight, char* out, size_t cap) {
  dmlc::RowBlock<feaid_t> blk;
  blk.size = nrows;
  blk.offset = offset;
  blk.label = label;
  blk.index = index;
  blk.value = value;
  blk.weight = weight;
  std::string str;
  CompressedRowBlock().Compress(blk, &str);
  if (str.size() > cap) return -1;
  memcpy(out, str.data(), str.size());
  return static_cast<long>(str.size());
}

/** CompressedRowBlock::Decompress<feaid_t> (:59-74).  Returns rows; counts of the optional arrays in n[4] =
 *  {nnz, values, weights, labels}; -1 when a capacity is too small. */
long ref_crb_decompress(const char* data, size_t size, size_t row_cap, size_t nnz_cap, size_t* offset, float* label,
                        uint64_t* index, float* value, float* weight, long* n) {
  dmlc::data::RowBlockContainer<feaid_t> blk;
  CompressedRowBlock().Decompress(data, size, &blk);
  const size_t nrows = blk.offset.size() - 1;
  if (nrows > row_cap || blk.index.size() > nnz_cap) return -1;
  memcpy(offset, blk.offset.data(), (nrows + 1) * sizeof(size_t));
  memcpy(label, blk.label.data(), blk.label.size() * sizeof(float));
  memcpy(index, blk.index.data(), blk.index.size() * sizeof(uint64_t));
  memcpy(value, blk.value.data(), blk.value.size() * sizeof(float));
  memcpy(weight, blk.weight.data(), blk.weight.size() * sizeof(float));
  n[0] = static_cast<long>(blk.index.size());
  n[1] = static_cast<long>(blk.value.size());
  n[2] = static_cast<long>(blk.weight.size());
  n[3] = static_cast<long>(blk.label.size());
  return static_cast<long>(nrows);
}

/** CriteoParser::ParseNext (:40-94) over one chunk of text.  Returns rows, -1 when a capacity is too small. */
long ref_criteo_parse(const char* text, size_t len, int is_train, size_t row_cap, size_t nnz_cap, size_t* offset, float* label,
                      uint64_t* index, long* nnz) {
  CriteoParser parser(new MemSplit(text, len), is_train != 0);  // the parser owns and deletes its source (:28-30)
  std::vector<dmlc::data::RowBlockContainer<feaid_t> > data;
  offset[0] = 0;
  *nnz = 0;
  if (!parser.ParseNext(&data)) return 0;
  const dmlc::data::RowBlockContainer<feaid_t>& blk = data[0];
  const size_t nrows = blk.offset.size() - 1;
  if (nrows > row_cap || blk.index.size() > nnz_cap) return -1;
  memcpy(offset, blk.offset.data(), (nrows + 1) * sizeof(size_t));
  memcpy(label, blk.label.data(), blk.label.size() * sizeof(float));
  memcpy(index, blk.index.data(), blk.index.size() * sizeof(uint64_t));
  *nnz = static_cast<long>(blk.index.size());
  return static_cast<long>(nrows);
}

/** AdfeaParser::ParseNext (src/reader/adfea_parser.h:33-88) over one chunk of text (NUL-terminated here: the parser reads
 *  the character behind the chunk, :58 / :62).  Returns rows, -1 when a capacity is too small. */
long ref_adfea_parse(const char* text, size_t len, size_t row_cap, size_t nnz_cap, size_t* offset, float* label, uint64_t* index,
                     long* nnz) {
  std::string z(text, len);
  z.push_back('\0');
  class ZSplit : public dmlc::InputSplit {
   public:
    ZSplit(std::string* s, size_t n) : s_(s), n_(n), done_(false) {}
    void BeforeFirst() override { done_ = false; }
    bool NextChunk(Blob* out) override {
      if (done_ || n_ == 0) return false;
      out->dptr = &(*s_)[0];
      out->size = n_;
      done_ = true;
      return true;
    }
   private:
    std::string* s_;
    size_t n_;
    bool done_;
  };
  AdfeaParser parser(new ZSplit(&z, len));
  std::vector<dmlc::data::RowBlockContainer<feaid_t> > data;
  offset[0] = 0;
  *nnz = 0;
  if (!parser.ParseNext(&data)) return 0;
  const dmlc::data::RowBlockContainer<feaid_t>& blk = data[0];
  const size_t nrows = blk.offset.size() - 1;
  if (nrows > row_cap || blk.index.size() > nnz_cap || blk.label.size() > row_cap) return -1;
  memcpy(offset, blk.offset.data(), (nrows + 1) * sizeof(size_t));
  memcpy(label, blk.label.data(), blk.label.size() * sizeof(float));
  memcpy(index, blk.index.data(), blk.index.size() * sizeof(uint64_t));
  *nnz = static_cast<long>(blk.index.size());
  return static_cast<long>(nrows);
}

/** the checker-side hash the parser above was linked with */
uint64_t ref_city_checker_hash64(const char* s, size_t len) { return CityHash64(s, len); }

}  // extern "C"
