/**
 * batch_reader.h — the data side of the worker loop: Reader + BatchReader of the reference
 * (src/reader/reader.h:27-58, src/reader/batch_reader.{h,cc}) over this build's own file splitting
 * and parsers (dmlc-core's InputSplit / LibSVMParser / RecordIO are an absent submodule):
 *
 *   format      source                                   parser here
 *   libsvm      "label idx:val ..." text                 LibsvmChunkParser   (ids taken as they are)
 *   criteo      criteo CTR text, label + 13 + 26         CriteoChunkParser   src/reader/criteo_parser.h:40-94
 *   criteo_test the same without the label column        CriteoChunkParser(is_train = false)
 *   rec         RecordIO of LZ4 CompressedRowBlocks      CrbRecordParser     src/reader/crb_parser.h:30-40,
 *                                                                            src/data/compressed_row_block.h:56-75
 *
 * A file is cut into num_parts byte ranges; text parts start at the first line start at or after
 * their first byte and end with the line crossing their last; RecordIO parts at record heads (the
 * magic word at a 4-byte-aligned position followed by a head or whole-record flag).  Rows come in
 * chunks (a chunk of text / one record).  A chunk is FETCHED from the file in order (cheap) and PARSED
 * (tokenising, CityHash64, LZ4: all of the time) by a pool of background threads, several chunks in
 * flight, delivered to the consumer in file order — dmlc's ThreadedParser (reader.h:44) overlaps one
 * parser thread with the consumer; one thread feeds 0.7-2.8 M rows/s, the device takes 77 M.
 *
 * BatchReader::Next() fills minibatches of batch_size rows across chunk borders, with the optional
 * shuffle buffer (a nested reader of shuffle_buf_size rows whose order is permuted) and negative
 * down-sampling of batch_reader.cc:32-77, and drops an all-ones value array (:71-73).
 */
#ifndef DIFACTO_HOST_BATCH_READER_H_
#define DIFACTO_HOST_BATCH_READER_H_
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "./cityhash.h"
#include "./lz4_block.h"
#include "data/row_block.h"
#include "difacto/base.h"

namespace difacto {

typedef dmlc::data::RowBlockContainer<feaid_t> RowChunk;

/*! \brief a parser yields the rows of its part of the file chunk by chunk, in two stages so that the
 * expensive one can run on several threads: Fetch (sequential, file order) and Parse (a pure function
 * of the fetched bytes) */
/*! \brief the bytes of one chunk: a view into the memory-mapped file (text), or owned (a RecordIO record
 * put together from its parts; the last lines of a text file, copied so that they end with a NUL) */
struct RawChunk {
  const char* data = nullptr;
  size_t size = 0;
  std::string own;
  void Own() { data = own.data(); size = own.size(); }
};

class ChunkParser {
 public:
  virtual ~ChunkParser() {}
  /*! \brief the raw bytes of the next chunk; false at the end of the part.  Called under the reader's lock. */
  virtual bool Fetch(RawChunk* raw) = 0;
  /*! \brief rows of a fetched chunk into *out (cleared first; may stay empty).  Thread-safe. */
  virtual void Parse(const RawChunk& raw, RowChunk* out) const = 0;
  /*! \brief single-threaded convenience: the next non-empty chunk of rows */
  bool ParseNext(RowChunk* out) {
    out->Clear();
    while (out->Size() == 0) {
      if (!Fetch(&raw_)) return false;
      Parse(raw_, out);
    }
    return true;
  }

 private:
  RawChunk raw_;
};

// ---------------------------------------------------------------------------------------------
// text: byte range [beg, end) of the file, whole lines, ~chunk_bytes at a time
// ---------------------------------------------------------------------------------------------
class TextChunks {
 public:
  TextChunks(const std::string& uri, unsigned part, unsigned nparts, size_t chunk_bytes) : chunk_bytes_(chunk_bytes) {
    fd_ = open(uri.c_str(), O_RDONLY);
    CHECK(fd_ >= 0) << "cannot open " << uri;
    struct stat st;
    CHECK(fstat(fd_, &st) == 0) << "cannot stat " << uri;
    size_ = static_cast<size_t>(st.st_size);
    if (size_ == 0) return;
    // the file is mapped, not read: fetching a chunk is finding its last line end, and the parser
    // threads pull the bytes out of the page cache themselves
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    CHECK(m != MAP_FAILED) << "cannot map " << uri;
    base_ = static_cast<const char*>(m);
    madvise(m, size_, MADV_SEQUENTIAL);
    size_t beg = size_ / nparts * part;
    end_ = (part + 1 == nparts) ? size_ : size_ / nparts * (part + 1);
    if (beg > 0) {  // the part starts at the first line start at or after beg
      const char* nl = static_cast<const char*>(memchr(base_ + beg - 1, '\n', size_ - (beg - 1)));
      beg = nl ? static_cast<size_t>(nl - base_) + 1 : size_;
    }
    pos_ = beg;
  }
  ~TextChunks() {
    if (base_) munmap(const_cast<char*>(base_), size_);
    if (fd_ >= 0) close(fd_);
  }
  /*! \brief next run of whole lines (no trailing partial line); false when the part is exhausted */
  bool Next(RawChunk* out) {
    out->own.clear();
    out->data = nullptr;
    out->size = 0;
    if (pos_ >= end_) return false;
    // (Measured dead end, round 5: sixteenths of a chunk for a job's first 64 chunks, so that the whole parser pool works on
    // the first shuffle buffer — start-up 77 -> 64 ms, but the pool of chunk containers then recycles 64 small containers
    // for ever and the steady state fell 70 -> 54-62 M rows/s on criteo text; profiles/r05d_e2e_chunk_ramp_ab.txt.)
    size_t stop = std::min(pos_ + chunk_bytes_, end_);
    if (base_[stop - 1] != '\n') {  // finish the line that crosses the chunk (or the part) border
      const char* nl = static_cast<const char*>(memchr(base_ + stop, '\n', size_ - stop));
      stop = nl ? static_cast<size_t>(nl - base_) + 1 : size_;
    }
    if (stop == size_) {
      // the last bytes of the file: copied, so that the number parsers find a NUL behind them even when
      // the file does not end with a newline (or ends exactly at a page border)
      out->own.assign(base_ + pos_, stop - pos_);
      out->Own();
    } else {
      out->data = base_ + pos_;
      out->size = stop - pos_;
    }
    pos_ = stop;
    return true;
  }

 private:
  int fd_ = -1;
  const char* base_ = nullptr;
  size_t size_ = 0, pos_ = 0, end_ = 0;
  size_t chunk_bytes_;
};

/*! \brief "label idx[:val] idx[:val] ..." per line; '#' starts a comment line */
class LibsvmChunkParser : public ChunkParser {
 public:
  LibsvmChunkParser(const std::string& uri, unsigned part, unsigned nparts, size_t chunk_bytes)
      : src_(uri, part, nparts, chunk_bytes) {}
  bool Fetch(RawChunk* raw) override { return src_.Next(raw); }
  void Parse(const RawChunk& raw, RowChunk* out) const override {
    out->Clear();
    const char* p = raw.data;
    const char* const end = p + raw.size;
    while (p < end) {
      const char* eol = static_cast<const char*>(memchr(p, '\n', end - p));
      if (!eol) eol = end;
      ParseLine(p, eol, out);
      p = eol + 1;
    }
  }

 private:
  static void ParseLine(const char* p, const char* eol, RowChunk* out) {
    while (p < eol && (*p == ' ' || *p == '\t')) ++p;
    if (p >= eol || *p == '#' || *p == '\r') return;
    char* e;
    // strtof / strtoull skip leading white space, a newline included: a number must start where we stand and end inside
    // the line, or the next line's label would be taken for a missing value
    const float label = strtof(p, &e);  // the file's last lines are NUL-terminated (TextChunks)
    CHECK(e != p && e <= eol) << "bad libsvm line: " << std::string(p, eol - p);
    p = e;
    out->label.push_back(label);
    for (;;) {
      while (p < eol && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
      if (p >= eol) break;
      // 1 .. 19 plain digits (what every id is): read in place — strtoull gives the same number for them and is called
      // for everything else (a sign, 20+ digits with its overflow rule, garbage with its error)
      feaid_t id = 0;
      const char* q = p;
      while (q < eol && static_cast<unsigned>(*q - '0') < 10u) id = id * 10 + static_cast<feaid_t>(*q++ - '0');
      if (q == p || q - p > 19) {
        id = strtoull(p, &e, 10);
        CHECK(e != p && e <= eol) << "bad libsvm token in: " << std::string(p, eol - p);
        q = e;
      }
      p = q;
      float v = 1.0f;
      if (p < eol && *p == ':') {
        ++p;
        CHECK(p < eol && !isspace(static_cast<unsigned char>(*p))) << "libsvm feature " << id << " has no value after ':'";
        if (*p == '1' && (p + 1 == eol || p[1] == ' ' || p[1] == '\t' || p[1] == '\r')) {   // ":1", the binary files' value
          ++p;
        } else {
          v = strtof(p, &e);
          CHECK(e != p && e <= eol) << "bad libsvm value of feature " << id;
          p = e;
        }
      }
      out->index.push_back(id);
      out->value.push_back(v);
    }
    out->offset.push_back(out->index.size());
  }
  TextChunks src_;
};

/**
 * \brief adfea CTR text (src/reader/adfea_parser.h:33-88): blank-separated tokens.  `idx:gid` is a feature, its id
 * EncodeFeaGrpID(idx, gid, 12) (:62-64; "the top bits store the feature group id"); the plain numbers come in threes —
 * a line id, a count, the label — of which the third opens a row and is 1 iff its first character is '1' (:67-76).
 * Line ends are blanks like any other: the reference parses a chunk as one token stream, and so does this (a chunk
 * starts at a line start, i.e. at a line id).  Numbers are read the way dmlc-core's strtoull reads them: digits
 * accumulated without an overflow check.
 */
class AdfeaChunkParser : public ChunkParser {
 public:
  AdfeaChunkParser(const std::string& uri, unsigned part, unsigned nparts, size_t chunk_bytes) : src_(uri, part, nparts, chunk_bytes) {}
  bool Fetch(RawChunk* raw) override { return src_.Next(raw); }
  void Parse(const RawChunk& raw, RowChunk* out) const override {
    out->Clear();
    Parse(raw.data, raw.data + raw.size, out);
  }
  static void Parse(const char* p, const char* end, RowChunk* blk) {
    auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f'; };
    auto digit = [](char c) { return c >= '0' && c <= '9'; };
    int i = 0;
    while (p != end && blank(*p)) ++p;
    while (p != end) {
      const char* head = p;
      feaid_t num = 0;
      while (p != end && digit(*p)) num = num * 10 + static_cast<feaid_t>(*p++ - '0');
      CHECK(head != p) << "adfea: a token that does not start with a digit: " << std::string(head, std::min<size_t>(end - head, 24));
      if (p != end && *p == ':') {
        ++p;
        feaid_t gid = 0;
        while (p != end && digit(*p)) gid = gid * 10 + static_cast<feaid_t>(*p++ - '0');
        CHECK_LT(gid, static_cast<feaid_t>(1) << 12) << "adfea: feature group id beyond 12 bits";   // EncodeFeaGrpID, base.h:60-63
        blk->index.push_back((num << 12) | gid);
      } else if (i == 2) {  // skip the line id and the first count
        i = 0;
        if (!blk->label.empty()) blk->offset.push_back(blk->index.size());
        blk->label.push_back(*head == '1' ? 1.0f : 0.0f);
      } else {
        ++i;
      }
      while (p != end && blank(*p)) ++p;
    }
    if (!blk->label.empty()) blk->offset.push_back(blk->index.size());
  }

 private:
  TextChunks src_;
};

/**
 * \brief criteo CTR text (src/reader/criteo_parser.h:40-94): tab-separated
 *   <label> <13 integer features> <26 categorical features of 8 hex characters>
 * an empty field is a missing feature; feature i of a row becomes the id
 *   EncodeFeaGrpID(CityHash64(token), i, 12) = (hash << 12) | i     (:72, :84; base.h:60-63)
 * and the row has no values (binary).
 */
class CriteoChunkParser : public ChunkParser {
 public:
  CriteoChunkParser(const std::string& uri, unsigned part, unsigned nparts, size_t chunk_bytes, bool is_train)
      : src_(uri, part, nparts, chunk_bytes), is_train_(is_train) {}
  bool Fetch(RawChunk* raw) override { return src_.Next(raw); }
  void Parse(const RawChunk& raw, RowChunk* out) const override {
    out->Clear();
    Parse(raw.data, raw.data + raw.size, is_train_, out);
  }
  /*! \brief the reference's parse loop over one chunk of text.  Two implementations with the same result (ParseFast takes
   *  the rows it recognises as regular and hands every other row to the same ParseRow): DIFACTO_SLOW_PARSE=1, or a CPU
   *  without AVX2 / BMI / POPCNT, keeps the plain loop */
  static void Parse(const char* p, const char* end, bool is_train, RowChunk* blk) {
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi") && __builtin_cpu_supports("popcnt") &&
                             getenv("DIFACTO_SLOW_PARSE") == nullptr;
    if (fast) ParseFast(p, end, is_train, blk); else ParseSlow(p, end, is_train, blk);
  }
  static void ParseSlow(const char* p, const char* end, bool is_train, RowChunk* blk) {
    while (p != end) {
      while (p != end && (*p == '\r' || *p == '\n')) ++p;
      if (p == end) break;
      p = ParseRow(p, end, is_train, blk);
    }
  }
  /*! \brief one row the way the reference reads it (criteo_parser.h:55-93), p at its first character (not a line end);
   *  returns where the next row's search starts */
  static const char* ParseRow(const char* p, const char* end, bool is_train, RowChunk* blk) {
    const char* pp;
    if (is_train) {  // :59-66
      pp = Find(p, end, '\t');
      CHECK(p != pp) << "no label.., try criteo_test";
      blk->label.push_back(static_cast<float>(atof(p)));
      p = pp == end ? end : pp + 1;
    } else {
      blk->label.push_back(0);
    }
    for (feaid_t i = 0; i < 13 && p != end; ++i) {  // :69-76 integer features
      pp = Find(p, end, '\t');
      if (pp > p) blk->index.push_back(EncodeFeaGrpID(CityHash64(p, pp - p), static_cast<int>(i), 12));
      p = pp == end ? end : pp + 1;
    }
    for (int i = 0; i < 26; ++i) {  // :79-90 categorical features
      if (p == end) break;
      if (*p == '\n' || *p == '\r') break;  // short row
      if (isspace(static_cast<unsigned char>(*p))) { ++p; continue; }  // missing feature
      CHECK_GE(end - p, 8) << "truncated categorical feature";
      pp = p + 8;
      CHECK(pp == end || isspace(static_cast<unsigned char>(*pp))) << "categorical feature " << i << " is not 8 characters";
      blk->index.push_back(EncodeFeaGrpID(CityHash64(p, 8), i + 13, 12));
      if (pp == end) { p = end; break; }
      p = pp + 1;
      if (*pp == '\n' || *pp == '\r') break;
    }
    blk->offset.push_back(blk->index.size());
    return p;
  }

  /*! \brief Parse for the rows every real file is made of: `ntab` tabs, then '\n', every categorical field empty or 8
   *  characters that do not start with a blank, no '\r'.  The plain loop spends its time in mispredicted branches (where
   *  a field ends, whether it is missing, which length class CityHash takes: ~30 per row); here the '\t' / '\n' positions of
   *  64 bytes at a time come out of three vector compares, a missing field costs a discarded store, and the two hash
   *  shapes that occur (1-7 digits, 8 characters) are straight-line code.  Any row that is not of that shape — and every
   *  row from the first '\r' of the chunk on — goes through ParseRow, so the result is ParseSlow's byte for byte
   *  (tests/test_ingest.py fuzzes the two against each other).  396 -> ~130 ns per row on the build container. */
  __attribute__((target("avx2,bmi,popcnt"))) static void ParseFast(const char* p, const char* end, bool is_train, RowChunk* blk) {
    const char* const base = p;
    if (static_cast<size_t>(end - p) >= (1ULL << 31)) return ParseSlow(p, end, is_train, blk);   // 32-bit positions below
    const int ntab = is_train ? 39 : 38;
    const size_t kWin = 1 << 14;   // bytes of text scanned ahead at a time
    std::vector<uint32_t> posv(2 * kWin + 128), nlv(2 * kWin + 128);
    uint32_t* pos = posv.data();   // offsets (from base) of the '\t' and '\n' in [p, scanned), in order
    uint32_t* nlq = nlv.data();    // indices into pos[] of the '\n' among them
    size_t head = 0, npos = 0, nl_head = 0, nnl = 0;
    const char* scanned = p;
    bool cr = false;
    size_t n = blk->index.size(), cap = n;   // ids are written through a pointer; the vector is sized in steps
    auto flush = [&] { blk->index.resize(n); };
    while (p != end) {
      while (p != end && (*p == '\r' || *p == '\n')) ++p;
      if (p == end) break;
      if (!cr) {
        const uint32_t at = static_cast<uint32_t>(p - base);
        while (head < npos && pos[head] < at) ++head;                 // delimiters the last row (or the skip) went past
        while (nl_head < nnl && nlq[nl_head] < head) ++nl_head;
        while (nl_head == nnl && scanned != end && !cr) {             // no complete line in sight: scan on
          if (head) {   // compact
            for (size_t i = head; i < npos; ++i) pos[i - head] = pos[i];
            npos -= head;
            head = 0;
            nnl = nl_head = 0;   // (none left: that is why we are here)
          }
          if (npos > kWin) break;   // a line of > 16 K fields: not ours
          // ParseRow (a short row) and the skip over blank lines read the text themselves and may have left p BEYOND what
          // was scanned: delimiters before p must never enter pos[] — they would be counted into the next row's `ntab`
          // (a short row ending at a window boundary made the next row's d[] point before its first field)
          if (scanned < p) scanned = p;
          const char* stop = static_cast<size_t>(end - scanned) > kWin ? scanned + kWin : end;
          cr = ScanDelims(base, scanned, stop, pos, &npos, nlq, &nnl);
          scanned = stop;
        }
      }
      const char* nl = (!cr && nl_head < nnl) ? base + pos[nlq[nl_head]] : nullptr;
      if (nl == nullptr || nlq[nl_head] != head + ntab || end - nl < 8) {   // not a regular row (or too close to the end
        flush();                                                           // of the chunk for the 8-byte loads below)
        p = ParseRow(p, end, is_train, blk);
        n = cap = blk->index.size();
        continue;
      }
      if (cap - n < 40) {
        cap = std::max<size_t>(2 * cap, n + (1 << 16));
        blk->index.resize(cap);
      }
      feaid_t* idx = blk->index.data();
      const size_t n0 = n;
      const uint32_t* d = pos + head;   // d[j]: the delimiter that ends field j
      const char* f = p;                // start of the current field
      float label = 0;
      int j = 0;
      bool bad = false;
      if (is_train) {
        const size_t len = base + d[0] - f;
        if (len == 1 && static_cast<unsigned>(*f - '0') < 10u) label = static_cast<float>(*f - '0');
        else if (len == 0) bad = true;   // ParseRow reports it
        else label = static_cast<float>(atof(f));
        f = base + d[0] + 1;
        j = 1;
      }
      for (int i = 0; i < 13; ++i, ++j) {   // integer features: hashed whatever they hold, skipped when empty
        const size_t len = base + d[j] - f;
        const uint64_t h = len < 8 ? HashUpTo7(f, len) : CityHash64(f, len);
        idx[n] = (h << 12) | static_cast<feaid_t>(i);   // EncodeFeaGrpID(h, i, 12): i < 39 < 2^12, its checks are moot
        n += len != 0;
        f = base + d[j] + 1;
      }
      for (int i = 0; i < 26; ++i, ++j) {   // categorical features: 8 characters or nothing
        const size_t len = base + d[j] - f;
        const unsigned char c = static_cast<unsigned char>(*f);
        bad |= (len != 0) & ((len != 8) | (c == ' ') | (c == '\v') | (c == '\f'));
        idx[n] = (Hash8(f) << 12) | static_cast<feaid_t>(i + 13);
        n += len != 0;
        f = base + d[j] + 1;
      }
      if (bad) {
        n = n0;
        flush();
        p = ParseRow(p, end, is_train, blk);
        n = cap = blk->index.size();
        continue;
      }
      blk->label.push_back(label);
      blk->offset.push_back(n);
      p = nl + 1;
      head += ntab + 1;
      ++nl_head;
    }
    flush();
  }

 private:
  /*! \brief CityHash64 of a string of 0 .. 7 bytes (city.cc HashLen0to16, its len < 8 cases) without a branch on the
   *  length class: both shapes are computed — every load stays inside [s, s + 8), which the caller guarantees readable */
  __attribute__((always_inline)) static inline uint64_t HashUpTo7(const char* s, size_t len) {
    using namespace city;
    const uint64_t l = len ? len : 1;   // (len 0: the value is discarded; keeps the loads inside the field's 8 bytes)
    // 1 .. 3 bytes
    const uint32_t a = static_cast<uint8_t>(s[0]), b = static_cast<uint8_t>(s[l >> 1]), c = static_cast<uint8_t>(s[(l - 1) & 7]);
    const uint32_t y = a + (b << 8), z = static_cast<uint32_t>(l) + (c << 2);
    const uint64_t small = ShiftMix(y * k2 ^ z * k0) * k2;
    // 4 .. 7 bytes
    const uint64_t mul = k2 + l * 2;
    const uint64_t lo = Fetch32(s), hi = Fetch32(s + ((l - 4) & 3));
    const uint64_t mid = HashLen16(l + (lo << 3), hi, mul);
    return l >= 4 ? mid : small;
  }
  /*! \brief CityHash64 of exactly 8 bytes (HashLen0to16, len >= 8 with both words the same) */
  __attribute__((always_inline)) static inline uint64_t Hash8(const char* s) {
    using namespace city;
    const uint64_t mul = k2 + 16;
    const uint64_t b = Fetch64(s), a = b + k2;
    const uint64_t c = Rotate(b, 37) * mul + a;
    const uint64_t d = (Rotate(a, 25) + b) * mul;
    return HashLen16(c, d, mul);
  }
  /*! \brief appends the offsets of every '\t' and '\n' of [s, e) to pos[] and, for the '\n', their index in pos[] to nlq[];
   *  true if the range holds a '\r' */
  __attribute__((target("avx2,bmi,popcnt"))) static bool ScanDelims(const char* base, const char* s, const char* e, uint32_t* pos,
                                                                    size_t* npos, uint32_t* nlq, size_t* nnl) {
    const __m256i T = _mm256_set1_epi8('\t'), N = _mm256_set1_epi8('\n'), R = _mm256_set1_epi8('\r');
    size_t n = *npos, q = *nnl;
    uint32_t any_cr = 0;
    for (; e - s >= 64; s += 64) {
      const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s));
      const __m256i v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + 32));
      const uint64_t mt = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v0, T))) |
                          (static_cast<uint64_t>(static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v1, T)))) << 32);
      const uint64_t mn = static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v0, N))) |
                          (static_cast<uint64_t>(static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(v1, N)))) << 32);
      any_cr |= static_cast<uint32_t>(_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(v0, R), _mm256_cmpeq_epi8(v1, R))));
      uint64_t m = mt | mn;
      const uint32_t off = static_cast<uint32_t>(s - base);
      const int cnt = __builtin_popcountll(m);
      for (uint64_t w = mn; w; w &= w - 1) nlq[q++] = static_cast<uint32_t>(n + __builtin_popcountll(m & ((w & -w) - 1)));
      // positions in batches of eight unconditional stores (the surplus ones land beyond n and are overwritten)
      uint32_t* out = pos + n;
      for (int k = 0; k < 8; ++k) { out[k] = off + static_cast<uint32_t>(__builtin_ctzll(m | (1ULL << 63))); m &= m - 1; }
      if (cnt > 8) {
        for (int k = 8; k < 16; ++k) { out[k] = off + static_cast<uint32_t>(__builtin_ctzll(m | (1ULL << 63))); m &= m - 1; }
        for (int k = 16; k < cnt; ++k) { out[k] = off + static_cast<uint32_t>(__builtin_ctzll(m)); m &= m - 1; }
      }
      n += cnt;
    }
    for (; s != e; ++s) {
      const char ch = *s;
      if (ch == '\t' || ch == '\n') {
        if (ch == '\n') nlq[q++] = static_cast<uint32_t>(n);
        pos[n++] = static_cast<uint32_t>(s - base);
      }
      any_cr |= ch == '\r';
    }
    *npos = n;
    *nnl = q;
    return any_cr != 0;
  }
  static const char* Find(const char* p, const char* end, int c) {
    while (p != end && *p != c) ++p;
    return p;
  }
  TextChunks src_;
  bool is_train_;
};

// ---------------------------------------------------------------------------------------------
// RecordIO (dmlc-core include/dmlc/recordio.h, restated from its published format): records are
//   [magic 0xced7230a][lrec = cflag << 29 | length][payload padded to 4 bytes]
// cflag 0: whole record; 1 / 2 / 3: first / middle / last part of a record whose payload contained
// the magic word at an aligned position (the writer cuts there and drops the word; the reader puts
// it back between the parts).
// ---------------------------------------------------------------------------------------------
class RecordIOPart {
 public:
  static const uint32_t kMagic = 0xced7230a;
  // The file is mapped, like the text formats: a whole record (cflag 0 — every record whose payload does not contain the
  // magic word at an aligned position) is handed out as a view, and the parser threads pull its bytes out of the page
  // cache themselves.  Read with fread under the reader's lock (until round 4) the 4.4 GB of the end-to-end .rec file
  // went through one thread twice (string::resize zero-fill + the copy): ~0.45 s of the 0.47 s the epoch took.
  RecordIOPart(const std::string& uri, unsigned part, unsigned nparts) {
    fd_ = open(uri.c_str(), O_RDONLY);
    CHECK(fd_ >= 0) << "cannot open " << uri;
    struct stat st;
    CHECK(fstat(fd_, &st) == 0) << "cannot stat " << uri;
    size_ = static_cast<size_t>(st.st_size);
    if (size_ == 0) return;
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    CHECK(m != MAP_FAILED) << "cannot map " << uri;
    base_ = static_cast<const char*>(m);
    madvise(m, size_, MADV_SEQUENTIAL);
    size_t beg = size_ / nparts * part;
    end_ = (part + 1 == nparts) ? size_ : size_ / nparts * (part + 1);
    beg = (beg + 3) & ~static_cast<size_t>(3);
    // the part starts at the first record head at or after beg
    size_t pos = beg;
    while (pos + 8 <= size_) {
      const uint32_t flag = (Word(pos + 4) >> 29U) & 7U;
      if (Word(pos) == kMagic && (flag == 0 || flag == 1)) break;
      pos += 4;
    }
    pos_ = pos;
  }
  ~RecordIOPart() {
    if (base_) munmap(const_cast<char*>(base_), size_);
    if (fd_ >= 0) close(fd_);
  }
  /*! \brief next whole record whose head lies inside the part: a view into the mapping, or (a record stored in parts)
   *  put together in out->own */
  bool NextRecord(RawChunk* out) {
    out->own.clear();
    out->data = nullptr;
    out->size = 0;
    if (pos_ >= end_ || pos_ + 8 > size_) return false;
    bool parts = false;
    for (;;) {
      CHECK(pos_ + 8 <= size_) << "truncated RecordIO file";
      CHECK_EQ(Word(pos_), kMagic) << "invalid RecordIO file";
      const uint32_t h = Word(pos_ + 4);
      const uint32_t flag = (h >> 29U) & 7U, len = h & ((1U << 29U) - 1U);
      const size_t padded = (static_cast<size_t>(len) + 3U) & ~static_cast<size_t>(3);
      CHECK(pos_ + 8 + len <= size_) << "truncated RecordIO file";   // (the last record's padding may be missing)
      const char* payload = base_ + pos_ + 8;
      pos_ += 8 + padded;
      if (flag == 0 && !parts) {
        out->data = payload;
        out->size = len;
        return true;
      }
      parts = true;
      out->own.append(payload, len);
      if (flag == 0 || flag == 3) break;
      const uint32_t m = kMagic;
      out->own.append(reinterpret_cast<const char*>(&m), 4);
    }
    out->Own();
    return true;
  }

 private:
  uint32_t Word(size_t at) const {
    uint32_t w;
    memcpy(&w, base_ + at, 4);
    return w;
  }
  int fd_ = -1;
  const char* base_ = nullptr;
  size_t size_ = 0, pos_ = 0, end_ = 0;
};

/*! \brief CompressedRowBlock::Decompress (src/data/compressed_row_block.h:56-75, :120-133) */
inline void DecompressRowBlock(const char* data, size_t size, RowChunk* blk) {
  static const int kCrbMagic = 1196140743;
  size_t cur = 0;
  auto read_int = [&]() {
    CHECK_LE(cur + sizeof(int), size) << "truncated compressed row block";
    int v;
    memcpy(&v, data + cur, sizeof(int));
    cur += sizeof(int);
    return v;
  };
  auto inflate = [&](void* dst, size_t bytes) {
    const int cp = read_int();
    if (cp <= 0) return false;  // array absent
    CHECK_LE(cur + cp, size) << "truncated compressed row block";
    CHECK_EQ(Lz4DecompressBlock(data + cur, cp, static_cast<char*>(dst), bytes), static_cast<long>(bytes))
        << "corrupt LZ4 block in a compressed row block";
    cur += cp;
    return true;
  };
  blk->Clear();
  CHECK_EQ(read_int(), kCrbMagic) << "wrong data format";
  CHECK_EQ(read_int(), static_cast<int>(sizeof(feaid_t))) << "wrong indextype";
  const int nrows = read_int();
  CHECK_GE(nrows, 0);
  // sizes read from the file are not trusted: an LZ4 block expands at most 255 x, so the bytes that are left
  // bound every array that may follow
  const size_t max_out = (size - cur) * 255 + 64;
  CHECK_LE(static_cast<size_t>(nrows) * sizeof(real_t), max_out) << "corrupt compressed row block: " << nrows << " rows in " << size << " bytes";
  blk->label.resize(nrows);
  if (!inflate(blk->label.data(), nrows * sizeof(real_t))) blk->label.clear();
  blk->offset.resize(nrows + 1);
  CHECK(inflate(blk->offset.data(), (nrows + 1) * sizeof(size_t))) << "compressed row block without offsets";
  for (int i = 0; i < nrows; ++i) CHECK_LE(blk->offset[i], blk->offset[i + 1]) << "corrupt compressed row block: offsets decrease at row " << i;
  const size_t nnz = blk->offset[nrows] - blk->offset[0];
  CHECK_LE(nnz * sizeof(real_t), max_out) << "corrupt compressed row block: " << nnz << " nonzeros in " << size << " bytes";
  if (blk->offset[0] != 0) for (auto& o : blk->offset) o -= blk->offset[0];
  blk->index.resize(nnz);
  if (!inflate(blk->index.data(), nnz * sizeof(feaid_t))) blk->index.clear();
  blk->value.resize(nnz);
  if (!inflate(blk->value.data(), nnz * sizeof(real_t))) blk->value.clear();
  blk->weight.resize(nrows);
  if (!inflate(blk->weight.data(), nrows * sizeof(real_t))) blk->weight.clear();
  if (blk->label.empty()) blk->label.assign(nrows, 0);
}

/*! \brief one RecordIO record = one compressed row block (src/reader/crb_parser.h:30-40) */
class CrbRecordParser : public ChunkParser {
 public:
  CrbRecordParser(const std::string& uri, unsigned part, unsigned nparts) : src_(uri, part, nparts) {}
  bool Fetch(RawChunk* raw) override {
    return src_.NextRecord(raw);
  }
  void Parse(const RawChunk& raw, RowChunk* out) const override {
    CHECK_NE(raw.size, 0u);
    DecompressRowBlock(raw.data, raw.size, out);
  }

 private:
  RecordIOPart src_;
};

/*! \brief Reader (src/reader/reader.h:27-58): format -> parser; chunks are parsed by a pool of threads,
 * up to 2 x threads of them in flight, and handed to the consumer in file order */
class Reader {
 public:
  Reader(const std::string& uri, const std::string& format, unsigned part, unsigned nparts, size_t chunk_bytes = 1 << 24,
         int nthreads = 0) {
    if (const char* e = getenv("DIFACTO_CHUNK_BYTES")) chunk_bytes = std::max<size_t>(64, strtoull(e, nullptr, 10));  // text formats
    if (format == "libsvm") {
      parser_.reset(new LibsvmChunkParser(uri, part, nparts, chunk_bytes));
    } else if (format == "criteo") {
      parser_.reset(new CriteoChunkParser(uri, part, nparts, chunk_bytes, true));
    } else if (format == "criteo_test") {
      parser_.reset(new CriteoChunkParser(uri, part, nparts, chunk_bytes, false));
    } else if (format == "adfea") {
      parser_.reset(new AdfeaChunkParser(uri, part, nparts, chunk_bytes));
    } else if (format == "rec") {
      parser_.reset(new CrbRecordParser(uri, part, nparts));
    } else {
      LOG(FATAL) << "unknown format " << format << " (this build reads libsvm, criteo, criteo_test, adfea and rec)";
    }
    if (nthreads <= 0) {
      // DIFACTO_PARSER_THREADS, else an eighth of the CPUs this process may use (several readers may be alive: one per
      // rank of a node, a training and a validation reader), at least 1, at most 16.  "May use" = the hardware threads
      // or the container's CPU-time quota, whichever is smaller: the GPU boxes of this build give a container 16 cores'
      // worth (cgroup cpu.max "1600000 100000") of a 256-thread host, and threads beyond the quota only get the whole
      // process throttled (round 4: 32 / 48 parser threads there summed to 9 / 16 s of parsing against 6.4 s with 16,
      // end to end -25 %; 8 .. 16 threads are within 3 % of each other since the parsers got faster)
      const char* e = getenv("DIFACTO_PARSER_THREADS");
      nthreads = e ? atoi(e) : std::max(static_cast<int>(std::thread::hardware_concurrency() / 8), 1);
      if (!e) nthreads = std::min(nthreads, std::max(1, QuotaCpus() - 4));   // the loop, the uploads and the cutters run too
      nthreads = std::max(1, std::min(nthreads, 16));
    }
    slots_.resize(2 * static_cast<size_t>(nthreads));
    for (int t = 0; t < nthreads; ++t) workers_.emplace_back([this] { Work(); });
  }
  /*! \brief CPUs' worth of time the container's cgroup allows (cgroup v2 cpu.max "quota period"), or the hardware threads */
  static int QuotaCpus() {
    int cpus = static_cast<int>(std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32];
      long period = 0;
      if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        const long quota = atol(q);
        if (quota > 0) cpus = std::min<long>(cpus, std::max<long>(1, (quota + period - 1) / period));
      }
      fclose(f);
    }
    return std::max(cpus, 1);
  }
  ~Reader() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_)
      if (w.joinable()) w.join();
    if (getenv("DIFACTO_PROFILE"))
      LOG(INFO) << "reader: " << take_ << " chunks, " << workers_.size() << " parser threads: parsing " << t_parse_
                << " s (sum over threads), consumer waited " << t_wait_ << " s";
  }
  /*! \brief the next non-empty chunk of rows in file order; false when the part is exhausted */
  bool Next() {
    for (;;) {
      std::unique_lock<std::mutex> lk(mu_);
      Slot& s = slots_[take_ % slots_.size()];
      // chunk number take_ is parsed, or no chunk will ever get that number
      const auto w0 = std::chrono::steady_clock::now();
      cv_.wait(lk, [&] { return (s.state == kParsed && s.seq == take_) || (fetch_done_ && take_ >= fetched_); });
      t_wait_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
      if (!(s.state == kParsed && s.seq == take_)) return false;
      std::swap(cur_, s.rows);
      s.state = kFree;
      ++take_;
      lk.unlock();
      cv_.notify_all();
      if (cur_.Size() == 0) continue;  // a chunk of comment / empty lines
      blk_ = cur_.GetBlock();
      return true;
    }
  }
  const dmlc::RowBlock<feaid_t>& Value() const { return blk_; }
  /*! \brief Next(), but the chunk is handed over for keeps (slices of it may outlive the next call) */
  bool NextShared(std::shared_ptr<RowChunk>* out) {
    if (!Next()) return false;
    // the container comes out of a pool and goes back to it when the last slice lets go: the parser threads keep
    // writing into the same few dozen allocations (warm pages, which also is what the runtime's pageable uploads like:
    // out of freshly allocated chunks they ran ten times slower)
    std::shared_ptr<ChunkPool> pool = pool_;
    RowChunk* raw = nullptr;
    {
      std::lock_guard<std::mutex> lk(pool->mu);
      if (!pool->spare.empty()) {
        raw = pool->spare.back().release();
        pool->spare.pop_back();
      }
    }
    if (!raw) raw = new RowChunk();
    raw->Clear();
    std::shared_ptr<RowChunk> c(raw, [pool](RowChunk* p) {
      std::lock_guard<std::mutex> lk(pool->mu);
      pool->spare.emplace_back(p);
    });
    std::swap(*c, cur_);   // cur_ (and through it a parser's slot) gets the recycled storage
    blk_ = c->GetBlock();
    *out = std::move(c);
    return true;
  }

 private:
  struct ChunkPool {
    std::mutex mu;
    std::vector<std::unique_ptr<RowChunk>> spare;
  };
  std::shared_ptr<ChunkPool> pool_ = std::make_shared<ChunkPool>();
  enum State { kFree, kBusy, kParsed };
  struct Slot {
    State state = kFree;
    size_t seq = 0;
    RawChunk raw;
    RowChunk rows;
  };
  void Work() {
    for (;;) {
      Slot* s = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        // chunk number fetched_ goes into slot fetched_ % n: wait until the consumer has emptied it
        cv_.wait(lk, [&] { return stop_ || fetch_done_ || slots_[fetched_ % slots_.size()].state == kFree; });
        if (stop_ || fetch_done_) return;
        s = &slots_[fetched_ % slots_.size()];
        // the fetch itself stays under the lock: chunks leave the file in order
        if (!parser_->Fetch(&s->raw)) {
          fetch_done_ = true;
          lk.unlock();
          cv_.notify_all();
          return;
        }
        s->state = kBusy;
        s->seq = fetched_++;
      }
      const auto p0 = std::chrono::steady_clock::now();
      parser_->Parse(s->raw, &s->rows);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - p0).count();
      {
        std::lock_guard<std::mutex> lk(mu_);
        s->state = kParsed;
        t_parse_ += dt;
      }
      cv_.notify_all();
    }
  }
  std::unique_ptr<ChunkParser> parser_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Slot> slots_;
  size_t fetched_ = 0, take_ = 0;   // chunks handed to workers / to the consumer
  bool fetch_done_ = false, stop_ = false;
  double t_parse_ = 0, t_wait_ = 0;   // DIFACTO_PROFILE
  RowChunk cur_;
  dmlc::RowBlock<feaid_t> blk_;
};

/*! \brief BatchReader (src/reader/batch_reader.{h,cc}) */
/*! \brief rows of shuffle buffer number `buf` (1, 2, ..), in minibatch order: how a minibatch is DESCRIBED instead of copied
 * when the buffers live in device memory and the rows are gathered there (BatchReader::Describe) */
struct RowSeg {
  uint64_t buf;
  std::vector<unsigned> rows;
};

/*! \brief rows [row0, row0 + nrows) of a parsed chunk that stays alive as long as somebody names it: how a shuffle buffer is
 * handed to the device feed WITHOUT being assembled on the host (round 4: the 31 MB copy per buffer was what the worker
 * loop waited for) — the buffer's offsets and labels are assembled, its ids / values are uploaded slice by slice */
struct BufSlice {
  std::shared_ptr<RowChunk> chunk;
  size_t row0, nrows;
  const feaid_t* index() const { return chunk->index.data() + chunk->offset[row0]; }
  const real_t* value() const { return chunk->value.empty() ? nullptr : chunk->value.data() + chunk->offset[row0]; }
  size_t nnz() const { return chunk->offset[row0 + nrows] - chunk->offset[row0]; }
};

/*! \brief view of a row container; a described minibatch has offsets and labels but no index / value arrays */
inline dmlc::RowBlock<feaid_t> ViewOf(const RowChunk& c) {
  if (!c.index.empty() || c.offset.back() == 0) return c.GetBlock();
  dmlc::RowBlock<feaid_t> b;
  b.size = c.offset.size() - 1;
  b.offset = c.offset.data();
  b.label = c.label.empty() ? nullptr : c.label.data();
  b.weight = nullptr;
  b.index = nullptr;
  b.value = nullptr;
  return b;
}

/*! \brief anything that hands out row blocks one after another (BatchReader, and the thread that runs one ahead) */
class BatchSource {
 public:
  virtual ~BatchSource() {}
  /*! \brief the description of Value() when the source describes its minibatches (else empty) */
  virtual const std::vector<RowSeg>& Aux() const {
    static const std::vector<RowSeg> none;
    return none;
  }
  virtual void MoveAux(std::vector<RowSeg>* dst) { dst->clear(); }
  /*! \brief the slices Value() is made of when the source keeps its blocks as slices of parsed chunks (else empty) */
  virtual const std::vector<BufSlice>& Slices() const {
    static const std::vector<BufSlice> none;
    return none;
  }
  virtual void MoveSlices(std::vector<BufSlice>* dst) { dst->clear(); }
  /*! \brief next block; false when exhausted.  Value() stays valid until the next call */
  virtual bool Next() = 0;
  virtual const dmlc::RowBlock<feaid_t>& Value() const = 0;
  /*! \brief hands the rows of Value() over to *dst: a swap when the source owns them, a copy when Value() is a view */
  virtual void MoveOut(RowChunk* dst) = 0;
};

/**
 * Runs a BatchSource `depth` blocks ahead on its own thread: the reference reads the next minibatch while the tracker
 * thread executes the current one (sgd_learner.cc:196-224); here both the shuffle buffer (its assembly from parsed
 * chunks) and the minibatches cut from it (permutation + row gather) are produced ahead of their consumer.  The order
 * of the blocks is the inner source's: one producer, one consumer, a ring of depth + 1 row containers handed over by swap.
 */
class PrefetchSource : public BatchSource {
 public:
  PrefetchSource(BatchSource* inner, int depth) : inner_(inner), slots_(static_cast<size_t>(std::max(depth, 1)) + 1) {
    worker_ = std::thread([this] { Work(); });
  }
  ~PrefetchSource() override {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
  }
  bool Next() override {
    std::unique_lock<std::mutex> lk(mu_);
    if (held_) {  // the block handed out by the previous call goes back to the producer
      slots_[(tail_ - 1) % slots_.size()].full = false;
      held_ = false;
      cv_.notify_all();
    }
    cv_.wait(lk, [&] { return slots_[tail_ % slots_.size()].full || (done_ && tail_ == head_); });
    Slot& s = slots_[tail_ % slots_.size()];
    if (!s.full) return false;
    blk_ = ViewOf(s.rows);
    aux_ = &s.aux;
    slices_ = &s.slices;
    ++tail_;
    held_ = true;
    return blk_.size > 0;
  }
  const dmlc::RowBlock<feaid_t>& Value() const override { return blk_; }
  const std::vector<RowSeg>& Aux() const override { return aux_ ? *aux_ : BatchSource::Aux(); }
  const std::vector<BufSlice>& Slices() const override { return slices_ ? *slices_ : BatchSource::Slices(); }
  void MoveAux(std::vector<RowSeg>* dst) override {
    std::lock_guard<std::mutex> lk(mu_);
    CHECK(held_);
    std::swap(slots_[(tail_ - 1) % slots_.size()].aux, *dst);
  }
  void MoveOut(RowChunk* dst) override {
    std::lock_guard<std::mutex> lk(mu_);
    CHECK(held_);
    std::swap(slots_[(tail_ - 1) % slots_.size()].rows, *dst);
    blk_ = dmlc::RowBlock<feaid_t>();
  }

 private:
  struct Slot {
    bool full = false;
    RowChunk rows;
    std::vector<RowSeg> aux;
    std::vector<BufSlice> slices;
  };
  void Work() {
    for (;;) {
      Slot* s;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !slots_[head_ % slots_.size()].full; });
        if (stop_) return;
        s = &slots_[head_ % slots_.size()];
      }
      const bool ok = inner_->Next();  // outside the lock: this is the work being overlapped
      if (ok) {
        inner_->MoveAux(&s->aux);   // before MoveOut: a source may reset its description there
        inner_->MoveSlices(&s->slices);
        inner_->MoveOut(&s->rows);
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (ok) {
          s->full = true;
          ++head_;
        } else {
          done_ = true;
        }
      }
      cv_.notify_all();
      if (!ok) return;
    }
  }
  std::unique_ptr<BatchSource> inner_;
  std::vector<Slot> slots_;
  std::thread worker_;
  std::mutex mu_;
  std::condition_variable cv_;
  size_t head_ = 0, tail_ = 0;  // blocks produced / handed out
  bool held_ = false, done_ = false, stop_ = false;
  dmlc::RowBlock<feaid_t> blk_;
  const std::vector<RowSeg>* aux_ = nullptr;
  const std::vector<BufSlice>* slices_ = nullptr;
};

/**
 * The shuffle buffer's permutation.  The reference calls std::random_shuffle (batch_reader.cc:47), i.e. libstdc++'s
 * loop `swap(v[i], v[rand() % (i + 1)])` over the process-wide glibc rand() in its default state (seed 1) — its runs
 * are reproducible because nothing else in that process draws from rand().  In this process other libraries do (the HIP
 * runtime and RCCL, from their own threads), so the same call gave a different permutation on every run.  RefRand
 * restates glibc's default generator (random_r TYPE_3: additive feedback r[i] = r[i-3] + r[i-31] over 31 words filled
 * by the Lehmer step 16807 x mod 2^31-1, the first 310 outputs discarded, result = word >> 1) as a private,
 * process-wide stream: the permutations are the reference's and the same on every run.  host_tests.cc checks the
 * stream against rand() after srand(1) and the shuffle against std::random_shuffle.
 */
class RefRand {
 public:
  explicit RefRand(unsigned seed = 1) { Seed(seed); }
  void Seed(unsigned seed) {
    int32_t r[34];
    r[0] = static_cast<int32_t>(seed ? seed : 1);
    for (int i = 1; i < 31; ++i) {
      const int64_t x = (16807LL * r[i - 1]) % 2147483647LL;
      r[i] = static_cast<int32_t>(x < 0 ? x + 2147483647LL : x);
    }
    for (int i = 0; i < 31; ++i) st_[i] = static_cast<uint32_t>(r[i]);
    f_ = 3;
    b_ = 0;
    for (int i = 0; i < 310; ++i) Next();
  }
  /*! \brief the next value of rand(): 0 .. RAND_MAX (2^31 - 1) */
  int Next() {
    st_[f_] += st_[b_];
    const int out = static_cast<int>(st_[f_] >> 1);
    f_ = f_ == 30 ? 0 : f_ + 1;
    b_ = b_ == 30 ? 0 : b_ + 1;
    return out;
  }
  /*! \brief libstdc++'s std::random_shuffle(first, last) driven by this stream */
  template <typename T>
  void Shuffle(std::vector<T>* v) {
    for (size_t i = 1; i < v->size(); ++i) {
      const size_t j = static_cast<size_t>(Next()) % (i + 1);
      if (i != j) std::swap((*v)[i], (*v)[j]);
    }
  }
  /*! \brief the process-wide stream of the batch readers (the reference's rand() state outlives its readers too) */
  static RefRand* Global() {
    static RefRand g;
    return &g;
  }
  static std::mutex* GlobalLock() {
    static std::mutex m;
    return &m;
  }

 private:
  uint32_t st_[31];
  int f_ = 3, b_ = 0;
};

class BatchReader : public BatchSource {
 public:
  /*! \brief slice_buffers (readers with a shuffle buffer whose minibatches will be DESCRIBED, see DescribeSlices): the
   *  buffers are not assembled on the host — offsets and labels only, the ids / values stay in the parsed chunks */
  typedef std::function<void(const dmlc::RowBlock<feaid_t>& buffer, const std::vector<BufSlice>& slices, uint64_t serial)> SliceFn;
  /*! \brief slice_buffers (readers with a shuffle buffer whose minibatches will be DESCRIBED): the buffers are not
   *  assembled on the host — offsets and labels only, the ids / values stay in the parsed chunks.  `on_built`, if given, is
   *  called for every buffer (serial 1, 2, ..) on the thread that builds the buffers, as soon as the buffer exists — one
   *  buffer AHEAD of the one being cut into minibatches: where the device feed starts its upload, so that the copy
   *  is over when the first minibatch of the buffer wants its rows */
  BatchReader(const std::string& uri, const std::string& format, unsigned part_index, unsigned num_parts, unsigned batch_size,
              unsigned shuffle_buf_size = 0, float neg_sampling = 1.0f, bool slice_buffers = false, SliceFn on_built = nullptr)
      : batch_size_(batch_size), shuf_buf_(shuffle_buf_size), neg_sampling_(neg_sampling) {
    CHECK_GT(batch_size, 0u);
    if (shuf_buf_) {
      CHECK_GE(shuf_buf_, batch_size_);
      // the next shuffle buffer is assembled (on its own thread) while this one is being cut into minibatches
      BatchReader* inner = new BatchReader(uri, format, part_index, num_parts, shuf_buf_);
      inner->keep_slices_ = slice_buffers;
      inner->on_built_ = on_built;
      sliced_ = slice_buffers;
      uploaded_early_ = on_built != nullptr;
      // (sliced buffers are offsets + labels only: two ahead, so that their uploads have a lead)
      buf_reader_.reset(new PrefetchSource(inner, slice_buffers ? 2 : 1));
    } else {
      CHECK(!slice_buffers) << "slice_buffers needs a shuffle buffer";
      reader_.reset(new Reader(uri, format, part_index, num_parts, 1 << 24));
    }
  }

  /**
   * Describe instead of copy (readers with a shuffle buffer only): every buffer is handed to `on_buffer` once, when it becomes
   * current (serial 1, 2, ..), and a minibatch is its offsets, its labels and the list of (buffer, rows) it is made of
   * (Aux()) — same permutation, same sampling draws, same minibatch boundaries as the copying reader; the row data is
   * gathered wherever the buffers were put (the device: sgd_learner.cc's device feed).
   */
  typedef std::function<void(const dmlc::RowBlock<feaid_t>& buffer, uint64_t serial)> BufferFn;
  /*! \brief `before_release` (optional) is called before the storage of the buffer last handed to on_buffer is given up
   *  (the next buffer is about to be fetched, or the reader ends): an on_buffer that keeps reading the arrays after it
   *  returns — an upload running beside the description of the buffer's minibatches — finishes there */
  void Describe(BufferFn on_buffer, std::function<void()> before_release = nullptr) {
    CHECK(shuf_buf_) << "only a reader with a shuffle buffer can describe its minibatches";
    CHECK(!sliced_) << "a reader built with slice_buffers hands its buffers out through DescribeSlices";
    describe_ = true;
    on_buffer_ = on_buffer;
    before_release_ = before_release;
  }
  /*! \brief Describe for a reader built with slice_buffers: `buffer` carries size / offset / label (index and value are
   *  NULL), `slices` the parsed chunks its rows live in, in order.  on_buffer may be empty when the reader was built with
   *  on_built (the buffers were handed out as they were built) */
  void DescribeSlices(SliceFn on_buffer, std::function<void()> before_release = nullptr) {
    CHECK(shuf_buf_ && sliced_) << "DescribeSlices needs a reader built with a shuffle buffer and slice_buffers";
    CHECK(on_buffer || uploaded_early_) << "nobody takes the buffers";
    describe_ = true;
    on_slices_ = on_buffer;
    before_release_ = before_release;
  }
  const std::vector<BufSlice>& Slices() const override { return slices_; }
  void MoveSlices(std::vector<BufSlice>* dst) override {
    std::swap(slices_, *dst);
    slices_.clear();
  }
  const std::vector<RowSeg>& Aux() const override { return segs_; }
  void MoveAux(std::vector<RowSeg>* dst) override {
    std::swap(segs_, *dst);
    segs_.clear();
  }

  /*! \brief next minibatch; false when the part is exhausted (batch_reader.cc:32-77) */
  ~BatchReader() override {
    if (before_release_) before_release_();
    if (getenv("DIFACTO_PROFILE") && (t_fill_ + t_shuf_ + t_sel_ + t_app_) > 0)
      LOG(INFO) << "batch reader (" << batch_size_ << " rows, shuffle buffer " << shuf_buf_ << "): next chunk / buffer " << t_fill_
                << " s, permutation " << t_shuf_ << " s, row selection " << t_sel_ << " s, row gather " << t_app_ << " s";
  }
  bool Next() override {
    batch_.Clear();
    segs_.clear();
    slices_.clear();
    view_ = false;
    // a whole minibatch inside the current chunk, rows taken as they come: hand out a view of the chunk's
    // arrays instead of copying 39 ids per row (dmlc's RowBlock convention: offset holds absolute positions
    // into index / value).  Valid until the next call, like the copy.
    if (shuf_buf_ == 0 && neg_sampling_ == 1.0f && !keep_slices_ && end_ - start_ >= batch_size_) {
      out_blk_.size = batch_size_;
      out_blk_.offset = in_blk_.offset + start_;
      out_blk_.label = in_blk_.label + start_;
      out_blk_.weight = nullptr;
      out_blk_.index = in_blk_.index;
      out_blk_.value = in_blk_.value;
      if (out_blk_.value) {  // an all-ones value array is dropped (batch_reader.cc:71-73)
        bool binary = true;
        for (size_t j = out_blk_.offset[0]; j < out_blk_.offset[batch_size_]; ++j)
          if (out_blk_.value[j] != 1) { binary = false; break; }
        if (binary) out_blk_.value = nullptr;
      }
      start_ += batch_size_;
      view_ = true;
      return true;
    }
    while (batch_.offset.size() < batch_size_ + 1) {
      if (start_ == end_) {
        const double f0 = Now();
        if (shuf_buf_ == 0) {
          if (keep_slices_) {
            if (!reader_->NextShared(&cur_chunk_)) break;
          } else if (!reader_->Next()) {
            break;
          }
          in_blk_ = reader_->Value();
          t_fill_ += Now() - f0;
        } else {
          if (before_release_) before_release_();
          if (!buf_reader_->Next()) break;
          in_blk_ = buf_reader_->Value();
          ++buf_serial_;
          if (describe_ && sliced_) {
            if (on_slices_) on_slices_(in_blk_, buf_reader_->Slices(), buf_serial_);
          } else if (describe_) {
            on_buffer_(in_blk_, buf_serial_);
          }
          const double f1 = Now();
          t_fill_ += f1 - f0;
          if (rdp_.size() != in_blk_.size) {
            rdp_.resize(in_blk_.size);
            for (size_t i = 0; i < in_blk_.size; ++i) rdp_[i] = static_cast<unsigned>(i);
          }
          {  // the reference's std::random_shuffle on its own rand() stream (RefRand above)
            std::lock_guard<std::mutex> lk(*RefRand::GlobalLock());
            RefRand::Global()->Shuffle(&rdp_);
          }
          t_shuf_ += Now() - f1;
        }
        start_ = 0;
        end_ = in_blk_.size;
      }
      const size_t len = std::min(end_ - start_, batch_size_ + 1 - batch_.offset.size());
      if (shuf_buf_ == 0 && neg_sampling_ == 1.0f) {
        Push(start_, len);
      } else {
        // the rows picked by the permutation / the sampling first (same draws, same order as the
        // reference's row-by-row loop, batch_reader.cc:55-63), then ONE append of all of them
        const double s0 = Now();
        sel_.clear();
        for (size_t i = start_; i < start_ + len; ++i) {
          const size_t j = shuf_buf_ ? rdp_[i] : i;  // (the reference reads an unset rdp_ here when only sampling)
          const float p = static_cast<float>(rand_r(&seed_)) / static_cast<float>(RAND_MAX);
          if (neg_sampling_ < 1.0f && in_blk_.label[j] <= 0 && p > 1 - neg_sampling_) continue;
          sel_.push_back(j);
        }
        const double s1 = Now();
        if (describe_) DescribeRows(); else AppendRows();
        t_sel_ += s1 - s0;
        t_app_ += Now() - s1;
      }
      start_ += len;
    }
    bool binary = true;
    for (auto f : batch_.value)
      if (f != 1) { binary = false; break; }
    if (binary) batch_.value.clear();
    out_blk_ = ViewOf(batch_);
    if (on_built_ && out_blk_.size > 0) on_built_(out_blk_, slices_, ++built_serial_);
    return out_blk_.size > 0;
  }
  const dmlc::RowBlock<feaid_t>& Value() const override { return out_blk_; }
  void MoveOut(RowChunk* dst) override {
    if (view_) {  // Value() is a view of the current chunk: copy it out
      dst->Clear();
      dmlc::RowBlock<feaid_t> slice = out_blk_;
      slice.index = out_blk_.index + out_blk_.offset[0];
      slice.value = out_blk_.value ? out_blk_.value + out_blk_.offset[0] : nullptr;
      dst->Push(slice);
    } else {
      std::swap(batch_, *dst);
    }
    out_blk_ = dmlc::RowBlock<feaid_t>();
  }

 private:
  void Push(size_t pos, size_t len) {  // batch_reader.cc:80-96
    if (!len) return;
    CHECK_LE(pos + len, in_blk_.size);
    if (keep_slices_) {  // offsets and labels only; the ids / values stay where the parser put them
      const size_t r0 = batch_.label.size();
      batch_.label.insert(batch_.label.end(), in_blk_.label + pos, in_blk_.label + pos + len);
      batch_.offset.resize(r0 + 1 + len);
      const size_t shift = batch_.offset[r0], o0 = in_blk_.offset[pos];
      for (size_t i = 0; i < len; ++i) batch_.offset[r0 + 1 + i] = shift + (in_blk_.offset[pos + 1 + i] - o0);
      slices_.push_back(BufSlice{cur_chunk_, pos, len});
      return;
    }
    dmlc::RowBlock<feaid_t> slice;
    slice.weight = nullptr;
    slice.size = len;
    slice.offset = in_blk_.offset + pos;
    slice.label = in_blk_.label + pos;
    slice.index = in_blk_.index + in_blk_.offset[pos];
    slice.value = in_blk_.value ? in_blk_.value + in_blk_.offset[pos] : nullptr;
    PushSlice(slice);
  }
  // describe mode: the offsets and labels of rows sel_[] of the current buffer onto batch_, the rows themselves onto segs_
  void DescribeRows() {
    if (sel_.empty()) return;
    const size_t r0 = batch_.label.size(), nsel = sel_.size();
    batch_.label.resize(r0 + nsel);
    batch_.offset.resize(batch_.offset.size() + nsel);
    size_t at = batch_.offset[r0];
    RowSeg seg;
    seg.buf = buf_serial_;
    seg.rows.resize(nsel);
    for (size_t q = 0; q < nsel; ++q) {
      at += in_blk_.offset[sel_[q] + 1] - in_blk_.offset[sel_[q]];
      batch_.offset[r0 + 1 + q] = at;
      batch_.label[r0 + q] = in_blk_.label[sel_[q]];
      seg.rows[q] = static_cast<unsigned>(sel_[q]);
    }
    segs_.push_back(std::move(seg));
  }
  // rows sel_[] of in_blk_, in that order, onto batch_: what Push(j, 1) per row gives, without the per-row
  // bookkeeping (10 000 single-row slices per minibatch were the slowest thing on the host)
  void AppendRows() {
    if (sel_.empty()) return;
    size_t add = 0;
    for (size_t j : sel_) add += in_blk_.offset[j + 1] - in_blk_.offset[j];
    const size_t nnz_before = batch_.index.size();
    if (in_blk_.value && batch_.value.size() < nnz_before) batch_.value.resize(nnz_before, 1.0f);
    batch_.index.resize(nnz_before + add);
    if (in_blk_.value) batch_.value.resize(nnz_before + add);
    else if (!batch_.value.empty()) batch_.value.resize(nnz_before + add, 1.0f);
    const size_t r0 = batch_.label.size(), nsel = sel_.size();
    batch_.label.resize(r0 + nsel);
    batch_.offset.resize(batch_.offset.size() + nsel);
    size_t at = nnz_before;
    for (size_t q = 0; q < nsel; ++q) {  // where every row goes
      at += in_blk_.offset[sel_[q] + 1] - in_blk_.offset[sel_[q]];
      batch_.offset[r0 + 1 + q] = at;
      batch_.label[r0 + q] = in_blk_.label[sel_[q]];
    }
    // the copy itself is a gather of ~300 B rows from a buffer far larger than the caches: the rows a few
    // steps ahead are prefetched, so that their misses overlap instead of following one another
    // Every row's destination is known, so the rows can be split over threads (DIFACTO_GATHER_THREADS, default 1).  Measured
    // on the 256-thread host of the GPU box, 640 minibatches of 10 000 rows: 0.25-0.31 s with 1 thread, 0.24-0.29 s with
    // 4, 0.17-0.24 s with 8 — the rows come out of a buffer another thread assembled (another CCD's cache), not out of
    // this core's, and more cores do not change where the lines are.
    constexpr size_t kAhead = 32;  // tools/gather_host_bench.cc on the GPU box's CPU: 12 rows ahead 0.300 ms, 32 rows 0.274 ms per minibatch
    const int nth = nsel >= 2048 ? GatherThreads() : 1;
#pragma omp parallel for num_threads(nth) schedule(static) if (nth > 1)
    for (size_t q = 0; q < nsel; ++q) {
      if (q + kAhead < nsel) {
        const size_t pb = in_blk_.offset[sel_[q + kAhead]], pn = in_blk_.offset[sel_[q + kAhead] + 1] - pb;
        const char* pp = reinterpret_cast<const char*>(in_blk_.index + pb);
        for (size_t x = 0; x < pn * sizeof(feaid_t); x += 64) __builtin_prefetch(pp + x, 0, 3);
        if (in_blk_.value) __builtin_prefetch(in_blk_.value + pb, 0, 3);
      }
      const size_t b = in_blk_.offset[sel_[q]], n = in_blk_.offset[sel_[q] + 1] - b;
      const size_t dst = batch_.offset[r0 + q];
      // offsets are absolute positions into index / value
      memcpy(&batch_.index[dst], in_blk_.index + b, n * sizeof(feaid_t));
      if (in_blk_.value) memcpy(&batch_.value[dst], in_blk_.value + b, n * sizeof(real_t));
    }
    // RowBlockContainer::max_index goes stale (GetMaxIndex() rescans on demand): nothing on this path reads it, and both a
    // max inside the copy loop and a pass of its own cost the gather measurably (tools/gather_host_bench.cc)
    batch_.max_index_stale = batch_.max_index_stale || add > 0;
  }
  static int GatherThreads() {
    static const int n = [] {
      const char* e = getenv("DIFACTO_GATHER_THREADS");
      return std::max(1, std::min(e ? atoi(e) : 1, 16));
    }();
    return n;
  }
  // a file may mix blocks with and without a value array (binary blocks drop it,
  // compressed_row_block.h:36-44): inside one minibatch a missing array means ones
  void PushSlice(const dmlc::RowBlock<feaid_t>& slice) {
    const size_t nnz_before = batch_.index.size();
    const size_t nnz = slice.offset[slice.size] - slice.offset[0];
    if (slice.value && batch_.value.size() < nnz_before) batch_.value.resize(nnz_before, 1.0f);
    batch_.Push(slice);
    if (!slice.value && !batch_.value.empty()) batch_.value.resize(nnz_before + nnz, 1.0f);
  }
  unsigned batch_size_, shuf_buf_;
  float neg_sampling_;
  std::unique_ptr<Reader> reader_;
  std::unique_ptr<BatchSource> buf_reader_;
  bool view_ = false;  // Value() points into the current chunk, not into batch_
  bool describe_ = false;
  BufferFn on_buffer_;
  SliceFn on_slices_;
  std::function<void()> before_release_;
  SliceFn on_built_;           // (buffer-building reader) called for every block it builds
  uint64_t built_serial_ = 0;
  bool uploaded_early_ = false;
  bool keep_slices_ = false;   // this reader (the one that builds shuffle buffers) keeps its blocks as slices of chunks
  bool sliced_ = false;        // this reader's shuffle buffers arrive as slices
  std::shared_ptr<RowChunk> cur_chunk_;
  std::vector<BufSlice> slices_;
  uint64_t buf_serial_ = 0;
  std::vector<RowSeg> segs_;
  size_t start_ = 0, end_ = 0;
  dmlc::RowBlock<feaid_t> in_blk_, out_blk_;
  RowChunk batch_;
  std::vector<unsigned> rdp_;
  std::vector<size_t> sel_;
  static double Now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  double t_fill_ = 0, t_shuf_ = 0, t_sel_ = 0, t_app_ = 0;   // DIFACTO_PROFILE
  unsigned int seed_ = 0;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_BATCH_READER_H_
