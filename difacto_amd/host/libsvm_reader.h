/**
 * libsvm_reader.h — minimal text reader for the host side of the path.
 * Stands in for the reference's Reader/BatchReader over dmlc-core's InputSplit
 * and LibSVMParser (src/reader/reader.h, batch_reader.{h,cc}): "label idx:val ..."
 * lines, file split in num_parts byte ranges cut at line ends, fixed-size
 * minibatches, optional shuffle buffer and negative down-sampling, and the
 * "all values are one -> drop the value array" rule (batch_reader.cc:71-73).
 * Feature indices are taken as they are (no 1-based shift), as the reference does.
 */
#ifndef DIFACTO_HOST_LIBSVM_READER_H_
#define DIFACTO_HOST_LIBSVM_READER_H_
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "data/row_block.h"
#include "difacto/base.h"

namespace difacto {

class LibsvmBatchReader {
 public:
  LibsvmBatchReader(const std::string& uri, unsigned part_index, unsigned num_parts, unsigned batch_size,
                    unsigned shuffle_buf_size = 0, float neg_sampling = 1.0f)
      : batch_size_(batch_size), shuf_buf_(shuffle_buf_size), neg_sampling_(neg_sampling), pos_(0), rng_(0), seed_(0) {
    CHECK_GT(batch_size, 0u);
    if (shuf_buf_) CHECK_GE(shuf_buf_, batch_size_);
    ReadPart(uri, part_index, num_parts);
  }

  /*! \brief next minibatch; false when the part is exhausted */
  bool Next() {
    batch_.Clear();
    while (batch_.Size() < batch_size_ && pos_ < all_.Size()) {
      if (shuf_buf_ && order_pos_ == order_.size()) RefillOrder();
      size_t i = shuf_buf_ ? order_[order_pos_++] : pos_;
      ++pos_;
      auto row = all_.GetBlock()[i];
      if (neg_sampling_ < 1.0f && row.label <= 0) {
        float p = static_cast<float>(rand_r(&seed_)) / static_cast<float>(RAND_MAX);
        if (p > 1 - neg_sampling_) continue;
      }
      batch_.Push(row);
    }
    bool binary = true;
    for (auto f : batch_.value)
      if (f != 1) { binary = false; break; }
    if (binary) batch_.value.clear();
    out_ = batch_.GetBlock();
    return out_.size > 0;
  }
  const dmlc::RowBlock<feaid_t>& Value() const { return out_; }

 private:
  void RefillOrder() {
    size_t n = std::min<size_t>(shuf_buf_, all_.Size() - pos_);
    order_.resize(n);
    for (size_t i = 0; i < n; ++i) order_[i] = pos_ + i;
    std::shuffle(order_.begin(), order_.end(), rng_);
    order_pos_ = 0;
  }

  void ReadPart(const std::string& uri, unsigned part, unsigned nparts) {
    FILE* fp = fopen(uri.c_str(), "rb");
    CHECK(fp != nullptr) << "cannot open " << uri;
    fseek(fp, 0, SEEK_END);
    long size = ftell(fp);
    long beg = size / nparts * part, end = (part + 1 == nparts) ? size : size / nparts * (part + 1);
    // a part starts at the first line start at or after `beg` and ends with the line crossing `end`
    if (beg > 0) {
      fseek(fp, beg - 1, SEEK_SET);
      int c;
      while ((c = fgetc(fp)) != EOF && c != '\n') {}
    } else {
      fseek(fp, 0, SEEK_SET);
    }
    std::string line;
    std::vector<feaid_t> idx;
    std::vector<dmlc::real_t> val;
    while (ftell(fp) < end) {
      line.clear();
      int c;
      while ((c = fgetc(fp)) != EOF && c != '\n') line.push_back(static_cast<char>(c));
      if (line.empty() && c == EOF) break;
      ParseLine(line, &idx, &val);
      if (c == EOF) break;
    }
    fclose(fp);
  }

  void ParseLine(const std::string& line, std::vector<feaid_t>* idx, std::vector<dmlc::real_t>* val) {
    const char* p = line.c_str();
    char* e;
    while (*p == ' ' || *p == '\t') ++p;
    if (*p == 0 || *p == '#') return;
    float label = strtof(p, &e);
    CHECK(e != p) << "bad libsvm line: " << line;
    p = e;
    idx->clear();
    val->clear();
    while (true) {
      while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
      if (*p == 0) break;
      feaid_t id = strtoull(p, &e, 10);
      CHECK(e != p) << "bad libsvm token in: " << line;
      p = e;
      float v = 1.0f;
      if (*p == ':') {
        ++p;
        v = strtof(p, &e);
        p = e;
      }
      idx->push_back(id);
      val->push_back(v);
    }
    dmlc::Row<feaid_t> row;
    row.label = label;
    row.weight = 1.0f;
    row.length = idx->size();
    row.index = idx->data();
    row.value = val->data();
    all_.Push(row);
  }

  unsigned batch_size_, shuf_buf_;
  float neg_sampling_;
  size_t pos_;
  std::vector<size_t> order_;
  size_t order_pos_ = 0;
  std::mt19937 rng_;
  unsigned int seed_;
  dmlc::data::RowBlockContainer<feaid_t> all_, batch_;
  dmlc::RowBlock<feaid_t> out_;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_LIBSVM_READER_H_
