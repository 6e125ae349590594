// Host-only microbenchmark of the reader's row gather (batch_reader.h: AppendRows): 10 000 rows of 39 u64 ids picked by a
// random permutation out of a 100 000-row block another thread wrote, variants of the copy loop.
// g++ -O2 -std=c++14 -pthread tools/gather_host_bench.cc -o build/gather_host_bench && build/gather_host_bench
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <thread>
#include <vector>
typedef uint64_t feaid_t;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Block { std::vector<size_t> offset; std::vector<feaid_t> index; };

template <int AHEAD, int LOC, bool MAXLOOP>
static feaid_t gather(const Block& in, const std::vector<unsigned>& sel, size_t s0, size_t n, std::vector<size_t>& off, std::vector<feaid_t>& idx) {
  off.resize(n + 1);
  size_t at = 0;
  off[0] = 0;
  for (size_t q = 0; q < n; ++q) { at += in.offset[sel[s0 + q] + 1] - in.offset[sel[s0 + q]]; off[q + 1] = at; }
  idx.resize(at);
  feaid_t mx = 0;
  for (size_t q = 0; q < n; ++q) {
    if (AHEAD && q + AHEAD < n) {
      const size_t pb = in.offset[sel[s0 + q + AHEAD]], pn = in.offset[sel[s0 + q + AHEAD] + 1] - pb;
      const char* pp = reinterpret_cast<const char*>(in.index.data() + pb);
      for (size_t x = 0; x < pn * sizeof(feaid_t); x += 64) __builtin_prefetch(pp + x, 0, LOC);
    }
    const size_t b = in.offset[sel[s0 + q]], m = in.offset[sel[s0 + q] + 1] - b;
    const feaid_t* src = in.index.data() + b;
    feaid_t* dst = idx.data() + off[q];
    if (MAXLOOP) {
      for (size_t x = 0; x < m; ++x) { dst[x] = src[x]; mx = std::max(mx, src[x]); }
    } else {
      memcpy(dst, src, m * sizeof(feaid_t));
    }
  }
  if (!MAXLOOP) for (size_t x = 0; x < at; ++x) mx = std::max(mx, idx[x]);
  return mx;
}

int main() {
  const size_t R = 100000, S = 39, B = 10000;
  Block blk;
  std::thread writer([&] {  // another thread (another core's cache) assembles the block, as the reader's buffer thread does
    std::mt19937_64 g(1);
    blk.offset.resize(R + 1);
    blk.index.resize(R * S);
    for (size_t i = 0; i <= R; ++i) blk.offset[i] = i * S;
    for (auto& v : blk.index) v = g();
  });
  writer.join();
  std::vector<unsigned> sel(R);
  std::iota(sel.begin(), sel.end(), 0u);
  std::shuffle(sel.begin(), sel.end(), std::mt19937(7));
  std::vector<size_t> off;
  std::vector<feaid_t> idx;
  feaid_t sink = 0;
#define RUN(name, ...)                                                                   \
  do {                                                                                   \
    double best = 1e9;                                                                   \
    for (int rep = 0; rep < 5; ++rep) {                                                  \
      const double t0 = now();                                                           \
      for (size_t s0 = 0; s0 + B <= R; s0 += B) sink ^= gather<__VA_ARGS__>(blk, sel, s0, B, off, idx); \
      best = std::min(best, (now() - t0) / (R / B));                                     \
    }                                                                                    \
    printf("%-34s %.3f ms per minibatch\n", name, best * 1e3);                           \
  } while (0)
  RUN("ahead 12, nta, copy+max loop (now)", 12, 0, true);
  RUN("ahead 12, L3 hint, copy+max loop", 12, 3, true);
  RUN("ahead 32, L3 hint, copy+max loop", 32, 3, true);
  RUN("ahead 64, L3 hint, copy+max loop", 64, 3, true);
  RUN("no prefetch, copy+max loop", 0, 0, true);
  RUN("ahead 12, nta, memcpy + max pass", 12, 0, false);
  RUN("ahead 32, L3 hint, memcpy + max pass", 32, 3, false);
  RUN("no prefetch, memcpy + max pass", 0, 0, false);
  printf("(%llu)\n", (unsigned long long)sink);
  return 0;
}
