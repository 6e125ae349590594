// What does the memory system give for the update kernel's access pattern when the rows are NOT already in
// the 256 MB memory-side cache?  tools/rmw_bench.hip and gather_bench.hip repeat ONE index set (80-100 MB:
// it stays in the Infinity Cache between launches), a training step touches new rows every minibatch.
// Here NSETS different random index sets are cycled (a set returns after NSETS x ~80 MB of other traffic).
//   gather    : 390 k random 256 B rows read (forward's pattern), stride 512 B table
//   rmw       : 147 k random rows: read V + acc (512 B), write both
//   rmw+hdr   : + 16 B read / 16 B write in a separate 32 B-stride header array (today's layout)
//   rmw640    : header inside the row: stride 640 B = [V 256 | acc 256 | hdr 32 | pad 96] (5 lines, one DRAM page)
//   rmw+hdr x2: two rows per lane group in flight
// usage: fresh_bench.bin [nsets=32] [same=0]   (same=1: one set, the old methodology)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256, 5) k_gather(const float* __restrict__ table, size_t stride, const uint32_t* __restrict__ rows,
                                                  size_t nreq, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t base = wave * 20; base < nreq; base += nwaves * 20) {
    float4 v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      size_t i = base + q * 4 + grp;
      if (i >= nreq) i = nreq - 1;
      v[q] = *reinterpret_cast<const float4*>(table + (size_t)rows[i] * stride + sub * 4);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
  }
  if (acc.x == 12345.678f) out[0] = acc.x + acc.y + acc.z + acc.w;
}

// HDR: 0 none, 1 separate array (16 B read / 16 B written of a 32 B header), 2 inside the row at float offset 128,
// 3 / 4 / 5: separate array whose header is 32 / 64 / 128 B, read and written WHOLE by 2 / 4 / 8 lanes (is the
// cost of the header its partial-line write?).  KPG rows per lane group in flight.
template <int HDR, int KPG>
__global__ void __launch_bounds__(256, 8) k_rmw(float* __restrict__ table, size_t stride, float* __restrict__ hdr,
                                               const uint32_t* __restrict__ rows, size_t nreq) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t base = wave * 4 * KPG; base < nreq; base += nwaves * 4 * KPG) {
    float4 a[KPG], b[KPG], h[KPG];
    float* p[KPG];
    float* hp[KPG];
#pragma unroll
    for (int q = 0; q < KPG; ++q) {
      size_t i = base + q * 4 + grp;
      if (i >= nreq) i = nreq - 1;
      const uint32_t r = rows[i];
      p[q] = table + (size_t)r * stride + sub * 4;
      constexpr int HL = HDR == 3 ? 2 : (HDR == 4 ? 4 : (HDR == 5 ? 8 : 1));  // lanes that carry the header
      hp[q] = HDR == 2 ? table + (size_t)r * stride + 128 : hdr + (size_t)r * (HL * 4 < 8 ? 8 : HL * 4) + (sub < HL ? sub * 4 : 0);
      a[q] = *reinterpret_cast<float4*>(p[q]);
      b[q] = *reinterpret_cast<float4*>(p[q] + 64);
      h[q] = HDR ? *reinterpret_cast<float4*>(hp[q]) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < KPG; ++q) {
      if (base + q * 4 + grp >= nreq) continue;
      a[q].x += 1.f + h[q].x; b[q].y += a[q].x; h[q].z += 1.f;
      *reinterpret_cast<float4*>(p[q]) = a[q];
      *reinterpret_cast<float4*>(p[q] + 64) = b[q];
      constexpr int HL2 = HDR == 3 ? 2 : (HDR == 4 ? 4 : (HDR == 5 ? 8 : 1));
      if (HDR && sub < HL2) *reinterpret_cast<float4*>(hp[q]) = h[q];
    }
  }
}

int main(int argc, char** argv) {
  const int nsets = argc > 1 ? atoi(argv[1]) : 32;
  const int same = argc > 2 ? atoi(argv[2]) : 0;
  const size_t nrows = 33000000, NU = 147000, NG = 390000;
  float *table, *hdr, *out; uint32_t *d_u, *d_g;
  CK(hipMalloc(&table, nrows * 640)); CK(hipMemset(table, 0, nrows * 640));
  CK(hipMalloc(&hdr, nrows * 128)); CK(hipMemset(hdr, 0, nrows * 128));
  CK(hipMalloc(&out, 256));
  std::vector<uint32_t> hu(NU * nsets), hg(NG * nsets);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (auto& x : hu) x = (uint32_t)(rnd() % nrows);
  for (auto& x : hg) x = (uint32_t)(rnd() % nrows);
  CK(hipMalloc(&d_u, hu.size() * 4)); CK(hipMalloc(&d_g, hg.size() * 4));
  CK(hipMemcpy(d_u, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name, double bytes) {
    for (int blocks : {2048, 4096, 8192}) {
      for (int it = 0; it < nsets; ++it) kern(blocks, same ? 0 : it);
      CK(hipDeviceSynchronize());
      const int reps = 2 * nsets;
      CK(hipEventRecord(e0));
      for (int it = 0; it < reps; ++it) kern(blocks, same ? 0 : it % nsets);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("%-14s blocks %5d : %7.1f us  %6.2f TB/s\n", name, blocks, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    }
  };
  printf("nsets %d%s\n", nsets, same ? " (ONE set repeated)" : "");
  run([&](int b, int it) { hipLaunchKernelGGL(k_gather, dim3(b), dim3(256), 0, 0, table, (size_t)128, d_g + (size_t)it * NG, NG, out); },
      "gather256", NG * 256.0);
  run([&](int b, int it) { hipLaunchKernelGGL(k_gather, dim3(b), dim3(256), 0, 0, table, (size_t)160, d_g + (size_t)it * NG, NG, out); },
      "gather256/640", NG * 256.0);
#define RMW(H, K, ST) [&](int b, int it) { hipLaunchKernelGGL((k_rmw<H, K>), dim3(b), dim3(256), 0, 0, table, (size_t)ST, hdr, d_u + (size_t)it * NU, NU); }
  run(RMW(0, 1, 128), "rmw", NU * 1024.0);
  run(RMW(1, 1, 128), "rmw+hdr", NU * 1056.0);
  run(RMW(3, 1, 128), "rmw+hdr32", NU * 1088.0);
  run(RMW(4, 1, 128), "rmw+hdr64", NU * 1152.0);
  run(RMW(5, 1, 128), "rmw+hdr128", NU * 1280.0);
  run(RMW(1, 2, 128), "rmw+hdr x2", NU * 1056.0);
  run(RMW(2, 1, 160), "rmw640", NU * 1056.0);
  run(RMW(2, 2, 160), "rmw640 x2", NU * 1056.0);
  return 0;
}
