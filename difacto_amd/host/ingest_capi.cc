/**
 * ingest_capi.cc — extern "C" view of the host-side ingest code (cityhash.h, lz4_block.h, batch_reader.h)
 * for the CPU tests in tests/test_ingest.py (ctypes).  Host only: no device is touched.
 */
#include <cstring>
#include <map>
#include "./batch_reader.h"

using namespace difacto;

extern "C" {

uint64_t ingest_cityhash64(const char* s, size_t len) { return CityHash64(s, len); }

long ingest_lz4_decompress(const char* src, size_t src_size, char* dst, size_t dst_cap) {
  return Lz4DecompressBlock(src, src_size, dst, dst_cap);
}

/*! \brief restarts the process-wide shuffle stream (RefRand) at the reference's default seed */
void ingest_reset_shuffle_stream() {
  std::lock_guard<std::mutex> lk(*RefRand::GlobalLock());
  RefRand::Global()->Seed(1);
}

/**
 * reads one part of a file through BatchReader and concatenates its minibatches.  Outputs are
 * caller-allocated with the given capacities; returns the number of rows, or -1 when a capacity is
 * too small.  *has_value = 0 when every minibatch dropped its (all ones) values.
 */
long ingest_read(const char* uri, const char* format, unsigned part, unsigned nparts, unsigned batch_size, unsigned shuffle,
                 float neg_sampling, size_t row_cap, size_t nnz_cap, size_t* offset, float* label, uint64_t* index, float* value,
                 int* has_value, long* nbatches) {
  // DIFACTO_INGEST_PREFETCH = n: through PrefetchSource, n minibatches ahead on a reader thread, as the worker loops read
  const char* pf = getenv("DIFACTO_INGEST_PREFETCH");
  // DIFACTO_INGEST_DESCRIBE = 1 (with a shuffle buffer): the reader describes its minibatches (BatchReader::Describe) and the
  // rows are gathered here from copies of the buffers it announced — what the device feed does in HBM
  const int describe_mode = getenv("DIFACTO_INGEST_DESCRIBE") && shuffle > 0 ? atoi(getenv("DIFACTO_INGEST_DESCRIBE")) : 0;
  const bool describe = describe_mode > 0;
  const bool sliced = describe_mode == 2;   // 2: the buffers arrive as slices of the parsed chunks (never assembled on the host)
  std::mutex bmu;
  std::map<uint64_t, RowChunk> buffers;
  std::unique_ptr<BatchReader> br(new BatchReader(uri, format, part, nparts, batch_size, shuffle, neg_sampling, sliced));
  if (describe && sliced)
    br->DescribeSlices([&](const dmlc::RowBlock<feaid_t>& blk, const std::vector<BufSlice>& slices, uint64_t serial) {
      std::lock_guard<std::mutex> lk(bmu);
      RowChunk& c = buffers[serial];
      c.Clear();
      // what the device feed does with one copy per slice: the buffer's own offsets / labels + the slices' ids / values in order
      c.offset.assign(blk.offset, blk.offset + blk.size + 1);
      for (auto& o : c.offset) o -= blk.offset[0];
      c.label.assign(blk.label, blk.label + blk.size);
      bool any_value = false;
      for (const BufSlice& sl : slices) any_value = any_value || sl.value() != nullptr;
      size_t rows = 0;
      for (const BufSlice& sl : slices) {
        c.index.insert(c.index.end(), sl.index(), sl.index() + sl.nnz());
        if (any_value) {
          if (sl.value()) c.value.insert(c.value.end(), sl.value(), sl.value() + sl.nnz());
          else c.value.resize(c.index.size(), 1.0f);
        }
        rows += sl.nrows;
      }
      if (rows != blk.size || c.index.size() != c.offset.back()) c.Clear();   // caught below as a mismatch
    });
  else if (describe)
    br->Describe([&](const dmlc::RowBlock<feaid_t>& blk, uint64_t serial) {
      std::lock_guard<std::mutex> lk(bmu);
      RowChunk& c = buffers[serial];
      c.Clear();
      dmlc::RowBlock<feaid_t> slice = blk;
      slice.index = blk.index + blk.offset[0];
      slice.value = blk.value ? blk.value + blk.offset[0] : nullptr;
      c.Push(slice);   // kept until the consumer has gathered the last minibatch that names it (released below)
    });
  std::unique_ptr<BatchSource> src(br.release());
  if (pf && atoi(pf) > 0) src.reset(new PrefetchSource(src.release(), atoi(pf)));
  BatchSource& reader = *src;
  RowChunk rebuilt;
  size_t rows = 0, nnz = 0;
  long nb = 0;
  offset[0] = 0;
  *has_value = 0;
  while (reader.Next()) {
    if (describe) {  // gather the described rows
      const auto& d = reader.Value();
      rebuilt.Clear();
      size_t q = 0;
      std::lock_guard<std::mutex> lk(bmu);
      for (const RowSeg& seg : reader.Aux()) {
        auto it = buffers.find(seg.buf);
        if (it == buffers.end()) return -2;
        const RowChunk& c = it->second;
        for (unsigned r : seg.rows) {
          const size_t lo = c.offset[r], n = c.offset[r + 1] - lo;
          if (d.offset[q + 1] - d.offset[q] != n || d.label[q] != c.label[r]) return -3;   // the description's own offsets / labels
          if (!c.value.empty() && rebuilt.value.size() < rebuilt.index.size()) rebuilt.value.resize(rebuilt.index.size(), 1.0f);
          rebuilt.index.insert(rebuilt.index.end(), c.index.begin() + lo, c.index.begin() + lo + n);
          if (!c.value.empty()) rebuilt.value.insert(rebuilt.value.end(), c.value.begin() + lo, c.value.begin() + lo + n);
          else if (!rebuilt.value.empty()) rebuilt.value.resize(rebuilt.index.size(), 1.0f);
          rebuilt.label.push_back(c.label[r]);
          rebuilt.offset.push_back(rebuilt.index.size());
          ++q;
        }
      }
      if (q != d.size) return -4;
      // minibatches take their rows from the buffers in order (with down-sampling one minibatch may span many of them):
      // everything before the last buffer this one names is exhausted
      uint64_t last = 0;
      for (const RowSeg& seg : reader.Aux()) last = std::max<uint64_t>(last, seg.buf);
      while (!buffers.empty() && buffers.begin()->first < last) buffers.erase(buffers.begin());
      bool binary = true;   // the copying reader drops an all-ones value array per minibatch (batch_reader.cc:71-73)
      for (auto f : rebuilt.value)
        if (f != 1) { binary = false; break; }
      if (binary) rebuilt.value.clear();
    }
    const auto b = describe ? rebuilt.GetBlock() : reader.Value();
    const size_t bn = b.offset[b.size] - b.offset[0];
    if (rows + b.size > row_cap || nnz + bn > nnz_cap) return -1;
    for (size_t i = 0; i < b.size; ++i) {
      label[rows + i] = b.label[i];
      offset[rows + i + 1] = nnz + (b.offset[i + 1] - b.offset[0]);
    }
    memcpy(index + nnz, b.index + b.offset[0], bn * sizeof(uint64_t));
    if (b.value) {
      memcpy(value + nnz, b.value + b.offset[0], bn * sizeof(float));
      *has_value = 1;
    } else {
      for (size_t i = 0; i < bn; ++i) value[nnz + i] = 1.0f;
    }
    rows += b.size;
    nnz += bn;
    ++nb;
  }
  *nbatches = nb;
  return static_cast<long>(rows);
}

/**
 * CriteoChunkParser on a buffer: mode 0 = the plain loop (ParseSlow), 1 = ParseFast (regular rows through the vector
 * scan, everything else through the same ParseRow), 2 = whichever Parse() picks on this CPU.  Returns the number of
 * rows, -1 when a capacity is too small, -2 when mode 1 is asked for on a CPU without AVX2 / BMI / POPCNT.
 */
long ingest_parse_criteo(const char* text, size_t len, int is_train, int mode, size_t row_cap, size_t nnz_cap, size_t* offset,
                         float* label, uint64_t* index) {
  RowChunk c;
  c.Clear();
  if (mode == 1) {
    if (!(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi") && __builtin_cpu_supports("popcnt"))) return -2;
    CriteoChunkParser::ParseFast(text, text + len, is_train != 0, &c);
  } else if (mode == 0) {
    CriteoChunkParser::ParseSlow(text, text + len, is_train != 0, &c);
  } else {
    CriteoChunkParser::Parse(text, text + len, is_train != 0, &c);
  }
  if (c.label.size() > row_cap || c.index.size() > nnz_cap) return -1;
  if (c.offset.size() != c.label.size() + 1) return -3;
  memcpy(offset, c.offset.data(), c.offset.size() * sizeof(size_t));
  memcpy(label, c.label.data(), c.label.size() * sizeof(float));
  memcpy(index, c.index.data(), c.index.size() * sizeof(uint64_t));
  return static_cast<long>(c.label.size());
}

}  // extern "C"
