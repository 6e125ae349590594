// Which ingredient of the update kernel costs what?  147 k random 512 B rows (17 GB table):
//   base   : row index -> read V,acc (16 lanes x float4 x 2) -> write both
//   +hdr   : also read 16 B + 4 B and write 16 B of a separate 32 B-stride header array (1 GB)
//   +xv    : also gather K random 256 B rows of a 2.56 MB array (L2 resident) per key
//   +math  : IEEE sqrt/div per element as the AdaGrad/FTRL update does
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int HDR, int XV, int MATH>
__global__ void __launch_bounds__(256, 8) k_upd(float* __restrict__ table, float* __restrict__ hdr, const float* __restrict__ xv,
                                               const uint32_t* __restrict__ rows, const uint32_t* __restrict__ xrows, size_t nreq) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, sub = lane & 15;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t base = wave * 4; base < nreq; base += nwaves * 4) {
    size_t i = base + grp;
    if (i >= nreq) continue;
    const uint32_t r = rows[i];
    float* p = table + (size_t)r * 128 + sub * 4;
    float4 a = *reinterpret_cast<float4*>(p);
    float4 b = *reinterpret_cast<float4*>(p + 64);
    float4 h = make_float4(0, 0, 0, 0);
    float cnt = 0;
    if (HDR) {
      h = *reinterpret_cast<float4*>(hdr + (size_t)r * 8);
      cnt = hdr[(size_t)r * 8 + 4];
    }
    float4 g = make_float4(0, 0, 0, 0);
    if (XV) {
#pragma unroll
      for (int q = 0; q < XV; ++q) {
        const uint32_t xr = xrows[i * XV + q];
        const float4 x = *reinterpret_cast<const float4*>(xv + (size_t)xr * 64 + sub * 4);
        g.x += x.x; g.y += x.y; g.z += x.z; g.w += x.w;
      }
    }
    if (MATH) {
      float* av = &a.x; float* bv = &b.x; float* gv = &g.x;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float gg = gv[d] + 0.01f * av[d];
        float n = sqrtf(bv[d] * bv[d] + gg * gg);
        bv[d] = n;
        av[d] -= 0.01f / (n + 1.0f) * gg;
      }
      if (sub == 0) {
        float sg = h.z, w = h.x, gw = g.x + cnt * 1e-9f;
        float nsg = sqrtf(sg * sg + gw * gw);
        h.w -= gw - (nsg - sg) / 0.01f * w;
        h.z = nsg;
        h.x = (h.w > 1.f ? h.w - 1.f : h.w + 1.f) / ((1.0f + nsg) / 0.01f);
      }
    } else {
      a.x += 1.f + g.x; b.y += a.x + cnt;
    }
    *reinterpret_cast<float4*>(p) = a;
    *reinterpret_cast<float4*>(p + 64) = b;
    if (HDR && sub == 0) *reinterpret_cast<float4*>(hdr + (size_t)r * 8) = h;
  }
}

int main(int argc, char** argv) {
  const size_t nreq = argc > 1 ? (size_t)atol(argv[1]) : 147000;
  const size_t nrows = 33000000;
  float *table, *hdr, *xv; uint32_t *d_rows, *d_xrows;
  CK(hipMalloc(&table, nrows * 512)); CK(hipMemset(table, 0, nrows * 512));
  CK(hipMalloc(&hdr, nrows * 32)); CK(hipMemset(hdr, 0, nrows * 32));
  CK(hipMalloc(&xv, 10000 * 256)); CK(hipMemset(xv, 0, 10000 * 256));
  std::vector<uint32_t> h(nreq), hx(nreq * 4);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < nreq; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % nrows); }
  for (size_t i = 0; i < nreq * 4; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hx[i] = (uint32_t)(s % 10000); }
  CK(hipMalloc(&d_rows, nreq * 4)); CK(hipMalloc(&d_xrows, nreq * 16));
  CK(hipMemcpy(d_rows, h.data(), nreq * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_xrows, hx.data(), nreq * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name) {
    for (int blocks : {2048, 4096, 9216}) {
      for (int it = 0; it < 3; ++it) kern(blocks);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) kern(blocks);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
      printf("%-22s blocks %5d : %7.1f us\n", name, blocks, ms * 1e3);
    }
  };
#define K(H, X, M) [&](int b) { hipLaunchKernelGGL((k_upd<H, X, M>), dim3(b), dim3(256), 0, 0, table, hdr, xv, d_rows, d_xrows, nreq); }
  run(K(0, 0, 0), "base");
  run(K(1, 0, 0), "+hdr");
  run(K(1, 1, 0), "+hdr+xv1");
  run(K(1, 4, 0), "+hdr+xv4");
  run(K(1, 0, 1), "+hdr+math");
  run(K(1, 4, 1), "+hdr+xv4+math");
  return 0;
}
