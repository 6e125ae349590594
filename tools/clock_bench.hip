// calibrate: shader clock under a short single-kernel burst, LDS dependent-read latency,
// global dependent-load latency (L2 hit), kernel launch floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fma_chain(float* out, int iters) {
  float a = threadIdx.x * 1e-9f, b = 1.000001f;
  for (int i = 0; i < iters; ++i) { a = a * b + 1e-9f; a = a * b + 1e-9f; a = a * b + 1e-9f; a = a * b + 1e-9f; }
  if (a == 123.f) out[0] = a;
}
__global__ void k_lds_chain(unsigned* out, int iters) {
  __shared__ unsigned idx[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  unsigned p = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) p = idx[p];
  long long t1 = clock64();
  if (threadIdx.x == 0) { out[0] = p; out[1] = (unsigned)((t1 - t0) / iters); }
}
__global__ void k_gl_chain(const unsigned* __restrict__ tab, unsigned* out, int iters, unsigned mask) {
  unsigned p = threadIdx.x + blockIdx.x * 64;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) p = tab[p & mask];
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = p; out[1] = (unsigned)((t1 - t0) / iters); }
}
__global__ void k_empty(unsigned* out) { if (out == nullptr) out[0] = 1; }

int main() {
  float* d; unsigned* u; CK(hipMalloc(&d, 256)); CK(hipMalloc(&u, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    int iters = 250000;  // 1M dependent fma
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma_chain, dim3(1), dim3(64), 0, 0, d, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fma chain 1 wave: %.3f ms for 1M dependent fma -> %.2f ns each\n", ms, ms * 1e6 / 1e6);
  }
  for (int waves : {1, 4, 16}) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(64 * waves), 0, 0, u, 100000); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h[2]; CK(hipMemcpy(h, u, 8, hipMemcpyDeviceToHost));
    printf("lds dependent read, %2d waves: %.2f ns each (clock64 ticks/iter %u)\n", waves, ms * 1e6 / 1e5, h[1]);
  }
  for (unsigned mb : {1u, 64u, 1024u}) {
    size_t n = (size_t)mb * 1024 * 1024 / 4;
    unsigned* tab; CK(hipMalloc(&tab, n * 4));
    unsigned* ht = (unsigned*)malloc(n * 4);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; ht[i] = (unsigned)(s % n); }
    CK(hipMemcpy(tab, ht, n * 4, hipMemcpyHostToDevice));
    for (int blocks : {1, 2048}) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_gl_chain, dim3(blocks), dim3(64), 0, 0, tab, u, 2000, (unsigned)(n - 1)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned h[2]; CK(hipMemcpy(h, u, 8, hipMemcpyDeviceToHost));
      printf("global dependent load, table %4u MB, %4d waves: %.1f ns each (ticks %u)\n", mb, blocks, ms * 1e6 / 2000, h[1]);
    }
    CK(hipFree(tab)); free(ht);
  }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1)); printf("empty kernel back-to-back: %.2f us each\n", ms);
  return 0;
}
