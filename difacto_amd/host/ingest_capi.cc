/**
 * ingest_capi.cc — extern "C" view of the host-side ingest code (cityhash.h, lz4_block.h, batch_reader.h)
 * for the CPU tests in tests/test_ingest.py (ctypes).  Host only: no device is touched.
 */
#include <cstring>
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
  std::unique_ptr<BatchSource> src(new BatchReader(uri, format, part, nparts, batch_size, shuffle, neg_sampling));
  if (pf && atoi(pf) > 0) src.reset(new PrefetchSource(src.release(), atoi(pf)));
  BatchSource& reader = *src;
  size_t rows = 0, nnz = 0;
  long nb = 0;
  offset[0] = 0;
  *has_value = 0;
  while (reader.Next()) {
    const auto& b = reader.Value();
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

}  // extern "C"
