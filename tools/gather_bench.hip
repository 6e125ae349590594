// Microbenchmark: what does MI355X deliver on the access pattern of the FM path?
//   gather : random 256 B rows (16 lanes x float4) out of a table of S bytes
//   rmw    : random 512 B rows read + written back (the in-place update)
// Prints GB/s per (table size, depth = independent loads in flight per lane).
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ table, const uint32_t* __restrict__ rows,
                                                size_t nreq, size_t row_floats, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t base = wave * 4 * DEPTH; base < nreq; base += nwaves * 4 * DEPTH) {
    float4 v[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      size_t i = base + q * 4 + grp;
      uint32_t r = i < nreq ? rows[i] : 0;
      v[q] = *reinterpret_cast<const float4*>(table + (size_t)r * row_floats + sub * 4);
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
  }
  if (acc.x == 12345.678f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

__global__ void __launch_bounds__(256) k_rmw(float* __restrict__ table, const uint32_t* __restrict__ rows, size_t nreq,
                                             size_t row_floats) {
  // one 16-lane group per row: read 2 x 256 B (V | acc), write both back
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t base = wave * 4; base < nreq; base += nwaves * 4) {
    size_t i = base + grp;
    if (i >= nreq) continue;
    float* p = table + (size_t)rows[i] * row_floats + sub * 4;
    float4 a = *reinterpret_cast<float4*>(p);
    float4 b = *reinterpret_cast<float4*>(p + 64);
    a.x += 1.f; b.y += a.x;
    *reinterpret_cast<float4*>(p) = a;
    *reinterpret_cast<float4*>(p + 64) = b;
  }
}

int main(int argc, char** argv) {
  const size_t nreq = argc > 1 ? (size_t)atol(argv[1]) : (4u << 20);  // row requests per launch
  std::vector<uint32_t> h(nreq);
  uint32_t* d_rows; float* d_out;
  CK(hipMalloc(&d_rows, nreq * 4)); CK(hipMalloc(&d_out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t row_floats = 128;  // 512 B rows as in the table (V | acc), gather reads the first 256 B
  for (double gb : {1.0, 17.0}) {
    size_t nrows = (size_t)(gb * 1e9 / (row_floats * 4));
    float* table; CK(hipMalloc(&table, nrows * row_floats * 4)); CK(hipMemset(table, 0, nrows * row_floats * 4));
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < nreq; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % nrows); }
    CK(hipMemcpy(d_rows, h.data(), nreq * 4, hipMemcpyHostToDevice));
    auto run = [&](auto kern, int blocks, double bytes_per_req, const char* name) {
      for (int it = 0; it < 2; ++it) kern(blocks);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) kern(blocks);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
      printf("table %6.2f GB  %-12s blocks %6d : %8.1f us  %7.1f GB/s\n", gb, name, blocks, ms * 1e3, nreq * bytes_per_req / (ms * 1e6));
    };
    for (int blocks : {2048, 8192, 32768}) {
      run([&](int b) { hipLaunchKernelGGL(k_gather<1>, dim3(b), dim3(256), 0, 0, table, d_rows, nreq, row_floats, d_out); }, blocks, 256, "gather d=1");
      run([&](int b) { hipLaunchKernelGGL(k_gather<4>, dim3(b), dim3(256), 0, 0, table, d_rows, nreq, row_floats, d_out); }, blocks, 256, "gather d=4");
      run([&](int b) { hipLaunchKernelGGL(k_gather<8>, dim3(b), dim3(256), 0, 0, table, d_rows, nreq, row_floats, d_out); }, blocks, 256, "gather d=8");
    }
    run([&](int b) { hipLaunchKernelGGL(k_rmw, dim3(b), dim3(256), 0, 0, table, d_rows, nreq, row_floats); }, 8192, 1024, "rmw 512B");
    run([&](int b) { hipLaunchKernelGGL(k_rmw, dim3(b), dim3(256), 0, 0, table, d_rows, nreq, row_floats); }, 65536, 1024, "rmw 512B");
    CK(hipFree(table));
  }
  return 0;
}
