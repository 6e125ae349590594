// ThreadSanitizer driver for the reader threads (batch_reader.h: parser pool -> shuffle buffer assembled ahead -> minibatches
// cut ahead): 20 readers over the rcv1 fixture with and without shuffle buffer / down-sampling, prefetch depth 1..3, some
// consumers stopping early.  Built and run by tests/test_ingest.py::test_reader_threads_under_tsan:
//   g++ -fsanitize=thread -O1 -g -std=c++14 -fopenmp -Iinclude -Ithird_party_shim -Idifacto_amd/host tools/tsan_reader.cc -lpthread
#include <atomic>
#include <cstdio>
#include "batch_reader.h"
using namespace difacto;
int main(int argc, char** argv) {
  size_t rows = 0, sum = 0;
  for (int rep = 0; rep < 24; ++rep) {
    std::atomic<size_t> announced{0};
    // rep % 8 == 7: the device feed's mode (round 4): buffers as slices of the parsed chunks, announced by the thread that
    // BUILDS them (on_built), read here on yet another thread's behalf: chunk pool, shared ownership, the description
    const bool sliced = rep % 8 == 7;
    BatchReader::SliceFn on_built;
    if (sliced)
      on_built = [&announced](const dmlc::RowBlock<feaid_t>& blk, const std::vector<BufSlice>& slices, uint64_t serial) {
        size_t s = blk.size + serial;
        for (const BufSlice& sl : slices)
          for (size_t i = 0; i < sl.nnz(); ++i) s += sl.index()[i];
        announced += s;
      };
    BatchReader* br = new BatchReader(argv[1], "libsvm", 0, 1, 7, (rep % 2 || sliced) ? 35 : 0, rep % 3 ? 1.0f : 0.7f, sliced, on_built);
    if (sliced) br->DescribeSlices(nullptr);
    else if (rep % 4 == 3)   // describe mode with assembled buffers: announced on the reader's thread
      br->Describe([&announced](const dmlc::RowBlock<feaid_t>& blk, uint64_t serial) { announced += blk.size + serial; });
    PrefetchSource r(br, 1 + rep % 3);
    int n = 0;
    while (r.Next()) {
      const auto& b = r.Value();
      rows += b.size;
      for (const RowSeg& g : r.Aux()) sum += g.rows.size() + g.buf;
      if (b.index)
        for (size_t i = b.offset[0]; i < b.offset[b.size]; ++i) sum += b.index[i];
      if (++n == 5 && rep % 5 == 4) break;   // a consumer that stops early
    }
    sum += announced.load() % 1000;
  }
  printf("rows %zu sum %zu\n", rows, sum);
  return 0;
}
