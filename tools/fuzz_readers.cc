// AddressSanitizer / UBSan harness for the file formats (batch_reader.h: mapped RecordIO + LZ4 block decoder with its
// fixed-size copies, CriteoChunkParser::ParseFast, the adfea and libsvm parsers): a good file is corrupted in four ways
// (byte flips, truncation, a burst of 0xFF / 0, a splice) and parsed; every outcome must be rows or a dmlc::Error (built
// with -DDMLC_LOG_FATAL_THROW=1), never a memory error.  Built and run by tests/test_ingest.py::
// test_readers_on_corrupted_files_under_asan:   fuzz_readers <format> <good file> <iterations> <scratch file>
#include <cstdio>
#include <fstream>
#include <random>
#include "batch_reader.h"
using namespace difacto;
static std::string slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv) {
  const std::string fmt = argv[1];
  const std::string good = slurp(argv[2]);
  const int iters = atoi(argv[3]);
  std::mt19937_64 g(5);
  long ok = 0, err = 0, rows = 0;
  for (int it = 0; it < iters; ++it) {
    std::string s = good;
    const int kind = it % 4;
    if (kind == 0) { for (int k = 0; k < 1 + (int)(g() % 8); ++k) s[g() % s.size()] = (char)g(); }          // byte flips
    else if (kind == 1) { s.resize(g() % s.size()); }                                                       // truncation
    else if (kind == 2) { size_t a = g() % s.size(); for (size_t i = a; i < std::min(s.size(), a + 64); ++i) s[i] = (char)(g() % 3 ? 0xFF : 0); }   // a burst
    else { size_t a = g() % s.size(), b = g() % s.size(); s = s.substr(0, a) + s.substr(b); }                // splice
    const char* path = argv[4];
    { std::ofstream f(path, std::ios::binary); f.write(s.data(), s.size()); }
    try {
      std::unique_ptr<ChunkParser> p;
      if (fmt == "rec") p.reset(new CrbRecordParser(path, 0, 1));
      else if (fmt == "criteo") p.reset(new CriteoChunkParser(path, 0, 1, 1 << 14, true));
      else if (fmt == "adfea") p.reset(new AdfeaChunkParser(path, 0, 1, 1 << 14));
      else p.reset(new LibsvmChunkParser(path, 0, 1, 1 << 14));
      RowChunk c;
      while (p->ParseNext(&c)) {
        rows += c.Size();
        uint64_t acc = 0;   // touch everything the parse produced
        for (auto v : c.index) acc += v;
        for (auto v : c.offset) acc += v;
        if (acc == 0x1234567) printf("!");
      }
      ++ok;
    } catch (const dmlc::Error&) {
      ++err;
    }
  }
  printf("%s: %d corrupted inputs: %ld parsed, %ld rejected, %ld rows\n", fmt.c_str(), iters, ok, err, rows);
}
