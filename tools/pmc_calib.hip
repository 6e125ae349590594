// Calibration of the memory-side counters on gfx950 for SMALL random accesses: how many bytes does the L2 request from
// the fabric for a random 16 B / 32 B / 64 B / 256 B read, and what does a 16 B / 32 B / 128 B random write cost?
// Every kernel makes NREQ requests to distinct random places of a 16 GiB table (nothing is reused, nothing fits a cache),
// so the true useful bytes are known; run under
//   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_DRAM_32B_sum ... --kernel-trace -- tools/pmc_calib.bin
// and compare (tools/pmc_table.py).  Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o tools/pmc_calib.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// LANES lanes of 16 B per request: 1 -> 16 B, 2 -> 32 B, 4 -> 64 B, 8 -> 128 B, 16 -> 256 B; the request starts on a
// (LANES * 16)-byte boundary inside a random 512 B row
template <int LANES>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ table, const uint32_t* __restrict__ rows, uint32_t nreq,
                                              float* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = min(t / LANES, nreq - 1), sub = t % LANES;
  const float4 v = table[(size_t)rows[i] * 32 + sub];
  if (v.x == 12345.678f) out[0] = v.x + v.y;
}
template <int LANES>
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ table, const uint32_t* __restrict__ rows, uint32_t nreq) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t / LANES >= nreq) return;
  table[(size_t)rows[t / LANES] * 32 + t % LANES] = make_float4(1.f, 2.f, 3.f, (float)t);
}
template <int LANES>
__global__ void __launch_bounds__(256) k_rmw(float4* __restrict__ table, const uint32_t* __restrict__ rows, uint32_t nreq) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t / LANES >= nreq) return;
  float4* p = table + (size_t)rows[t / LANES] * 32 + t % LANES;
  float4 v = *p;
  v.x += 1.f;
  *p = v;
}

int main(int argc, char** argv) {
  const uint32_t nreq = argc > 1 ? (uint32_t)atol(argv[1]) : 1000000u;
  const size_t nrows = (size_t)16 << 21;  // 2^25 rows of 512 B = 16 GiB
  float4* table;
  uint32_t* d_rows;
  float* d_out;
  CK(hipMalloc(&table, nrows * 512));
  CK(hipMemset(table, 0, nrows * 512));
  CK(hipMalloc(&d_rows, (size_t)nreq * 4));
  CK(hipMalloc(&d_out, 64));
  std::vector<uint32_t> h(nreq);
  uint64_t s = 88172645463325252ULL;
  auto fill = [&]() {  // distinct-enough random rows: 1e6 draws from 3.4e7, ~1.5 % repeats
    for (auto& r : h) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      r = (uint32_t)(s % nrows);
    }
    CK(hipMemcpy(d_rows, h.data(), (size_t)nreq * 4, hipMemcpyHostToDevice));
  };
#define RUN(kern, lanes, ...)                                                                      \
  do {                                                                                             \
    fill();                                                                                        \
    const unsigned blocks = (unsigned)(((size_t)nreq * lanes + 255) / 256);                        \
    hipLaunchKernelGGL((kern<lanes>), dim3(blocks), dim3(256), 0, 0, table, d_rows, nreq, ##__VA_ARGS__); \
    CK(hipDeviceSynchronize());                                                                    \
    printf("%-8s %3d B x %u requests = %.1f MB useful\n", #kern, lanes * 16, nreq, nreq * lanes * 16 / 1e6); \
  } while (0)
  RUN(k_read, 1, d_out);
  RUN(k_read, 2, d_out);
  RUN(k_read, 4, d_out);
  RUN(k_read, 8, d_out);
  RUN(k_read, 16, d_out);
  RUN(k_write, 1);
  RUN(k_write, 2);
  RUN(k_write, 4);
  RUN(k_write, 8);
  RUN(k_rmw, 1);
  RUN(k_rmw, 2);
  RUN(k_rmw, 8);
  return 0;
}
