// Would "index slot + header words in ONE 128 B line"  pay in the PIPELINED step?  Today a key's two passes
// read two DIFFERENT lines: the probe on the preparation stream reads the key's index line, the step's light pass ~100 us
// later reads its header line.  Merged, both passes would read the SAME line.  The bytes requested are equal; the question is
// whether the second read of a line that was fetched ~100 us earlier is cheaper than a first read of another line.
//   A  pass 1 reads line L1[k], pass 2 reads line L2[k]   (two arrays: today)
//   B  pass 1 reads line L1[k], pass 2 reads line L1[k]   (merged)
// 147 000 random keys of a 33 M-slot table per pass, 32 different key sets cycled (so that nothing is left from the previous
// iteration), ~100 us of unrelated streaming traffic (256 MB read) between the passes, as in the step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_pass(const float4* __restrict__ lines, const uint32_t* __restrict__ slot, uint32_t n, float* out) {
  float acc = 0.f;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 v = lines[(size_t)slot[i] * 8];   // 16 B of a 128 B line
    acc += v.x + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t nslots = 33000000, NK = 147000, NS = 32;
  float4 *l1, *l2, *filler; float* out; uint32_t* d_slot;
  CK(hipMalloc(&l1, nslots * 128)); CK(hipMemset(l1, 0, nslots * 128));
  CK(hipMalloc(&l2, nslots * 128)); CK(hipMemset(l2, 0, nslots * 128));
  const size_t fill_bytes = 512ull << 20;
  CK(hipMalloc(&filler, fill_bytes)); CK(hipMemset(filler, 0, fill_bytes));
  CK(hipMalloc(&out, 256));
  std::vector<uint32_t> h(NK * NS);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (auto& x : h) x = (uint32_t)(rnd() % nslots);
  CK(hipMalloc(&d_slot, h.size() * 4)); CK(hipMemcpy(d_slot, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int merged = 0; merged < 2; ++merged) {
    for (int with_filler = 0; with_filler < 2; ++with_filler) {
      double t2 = 0; const int reps = 64;
      for (int it = 0; it < reps + 8; ++it) {
        const uint32_t* ks = d_slot + (size_t)(it % NS) * NK;
        hipLaunchKernelGGL(k_pass, dim3(576), dim3(256), 0, 0, l1, ks, (uint32_t)NK, out);
        if (with_filler) hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, filler, fill_bytes / 2 / 16, out);  // ~256 MB: what a step moves between the two passes
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_pass, dim3(576), dim3(256), 0, 0, merged ? l1 : l2, ks, (uint32_t)NK, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 8) t2 += ms;
      }
      printf("second pass reads %-32s %-34s : %6.2f us\n", merged ? "the SAME line (merged layout)" : "ANOTHER line (index | header)",
             with_filler ? "after 256 MB of other traffic" : "right after the first pass", t2 / reps * 1e3);
    }
  }
  return 0;
}
