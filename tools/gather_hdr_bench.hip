// Microbenchmark: does the separate 32 B header array cost the forward gather its bandwidth?
//   a) 256 B V rows only (512 B stride, as the table has them)
//   b) a) + 8 B {w, has_V} from a separate 32 B-stride header array (the current layout)
//   c) header and V row contiguous: 272 B rows, 16 B header first (candidate layout)
// 390 k random requests into 33 M rows, depth-5 loads in flight, one 16-lane group per row.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_hdr_bench.hip -o tools/gather_hdr_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int DEPTH>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ table, const float* __restrict__ hdr,
                                                const uint32_t* __restrict__ rows, size_t nreq, size_t row_floats,
                                                float* __restrict__ out) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  float wsum = 0.f;
  for (size_t base = wave * 4 * DEPTH; base < nreq; base += nwaves * 4 * DEPTH) {
    float4 v[DEPTH];
    float2 h[DEPTH];
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) {
      size_t i = base + q * 4 + grp;
      uint32_t r = rows[i < nreq ? i : nreq - 1];
      if (MODE == 0) {
        v[q] = *reinterpret_cast<const float4*>(table + (size_t)r * row_floats + sub * 4);
        h[q] = make_float2(0.f, 0.f);
      } else if (MODE == 1) {
        v[q] = *reinterpret_cast<const float4*>(table + (size_t)r * row_floats + sub * 4);
        h[q] = *reinterpret_cast<const float2*>(hdr + (size_t)r * 8);
      } else {
        const float* p = table + (size_t)r * row_floats;   // [hdr 4 floats | V 64 floats]
        h[q] = *reinterpret_cast<const float2*>(p);
        v[q] = *reinterpret_cast<const float4*>(p + 4 + sub * 4);
      }
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; wsum += h[q].x + h[q].y; }
  }
  if (acc.x == 12345.678f || wsum == 3.25f) out[0] = acc.x + acc.y + acc.z + acc.w + wsum;
}

int main(int argc, char** argv) {
  const size_t nreq = argc > 1 ? (size_t)atol(argv[1]) : 390000;
  const size_t nrows = 33000000;
  std::vector<uint32_t> h(nreq);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < nreq; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % nrows); }
  uint32_t* d_rows; float* d_out;
  CK(hipMalloc(&d_rows, nreq * 4)); CK(hipMalloc(&d_out, 64));
  CK(hipMemcpy(d_rows, h.data(), nreq * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *t512, *hdr, *t272;
  CK(hipMalloc(&t512, nrows * 512)); CK(hipMemset(t512, 0, nrows * 512));
  CK(hipMalloc(&hdr, nrows * 32)); CK(hipMemset(hdr, 0, nrows * 32));
  CK(hipMalloc(&t272, nrows * 272)); CK(hipMemset(t272, 0, nrows * 272));
  auto run = [&](auto kern, const char* name, double bytes) {
    for (int it = 0; it < 3; ++it) kern();
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) kern();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms / 20 < best ? ms / 20 : best;
    }
    printf("%-34s : %7.1f us  %7.1f GB/s algorithmic (%.0f B/row)\n", name, best * 1e3, nreq * bytes / (best * 1e6), bytes);
  };
  for (int blocks : {2500, 4096, 8192}) {
    printf("blocks %d\n", blocks);
    run([&] { hipLaunchKernelGGL((k_gather<0, 5>), dim3(blocks), dim3(256), 0, 0, t512, hdr, d_rows, nreq, (size_t)128, d_out); }, "a) V rows only (512 B stride)", 256);
    run([&] { hipLaunchKernelGGL((k_gather<1, 5>), dim3(blocks), dim3(256), 0, 0, t512, hdr, d_rows, nreq, (size_t)128, d_out); }, "b) V rows + separate 32 B headers", 260);
    run([&] { hipLaunchKernelGGL((k_gather<2, 5>), dim3(blocks), dim3(256), 0, 0, t272, hdr, d_rows, nreq, (size_t)68, d_out); }, "c) header + V contiguous (272 B)", 260);
  }
  return 0;
}
