// Slot-major `index` for fixed-width rows: what the Localizer's emit pass would save on its 4-byte scatter
// and what the readers of `index` (the forward's prologue, the update's singles role) would pay.  C3 shape: B = 10 000 rows x
// 39 slots, pairs in key order = grouped by slot (the slot id is in the top bits of a reversed key), 381 pairs per emit block.
//   W row    index[row * 39 + slot] = rank      today: every write of a block lands in a different line of the 1.5 MB array
//   W slot   index[slot * B + row] = rank       slot-major: a block's writes stay inside one 40 KB region
//   W slot8  the same, blocks dealt so that ONE XCD owns a slot's region (block b takes a bucket of a slot = b mod 8 ...)
//   R row    wave per example: 39 consecutive words (2 lines), then the {row, w} words of those keys
//   R slot   wave per example: 39 words at stride B (39 lines, shared with the 31 neighbouring examples)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr uint32_t B = 10000, S = 39, N = B * S, PER = 381;

template <int MODE>
__global__ void __launch_bounds__(256) k_write(const uint32_t* __restrict__ row_of, const uint32_t* __restrict__ bucket_of_block,
                                               uint32_t* __restrict__ index, uint32_t nb) {
  const uint32_t bk = MODE == 2 ? bucket_of_block[blockIdx.x] : blockIdx.x;
  if (bk >= nb) return;
  for (uint32_t t = threadIdx.x; t < PER; t += blockDim.x) {
    const uint32_t p = bk * PER + t;
    if (p >= N) break;
    const uint32_t g = p / B, row = row_of[p];
    index[MODE == 0 ? row * S + g : g * B + row] = p >> 2;   // ("rank": any value)
  }
}
template <int MODE>
__global__ void __launch_bounds__(256) k_read(const uint32_t* __restrict__ index, const uint2* __restrict__ uw, uint32_t U, float* out) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= B) return;
  float acc = 0.f;
  if (lane < S) {
    const uint32_t u = index[MODE == 0 ? wave * S + lane : lane * B + wave] % U;
    const uint2 e = uw[u];
    acc = __uint_as_float(e.y) + (float)e.x;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  std::vector<uint32_t> row_of(N), bob;
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (uint32_t g = 0; g < S; ++g) {   // within a slot the pairs are in key order: their rows are a random permutation
    std::vector<uint32_t> perm(B);
    for (uint32_t i = 0; i < B; ++i) perm[i] = i;
    for (uint32_t i = B - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (i + 1)]);
    for (uint32_t i = 0; i < B; ++i) row_of[g * B + i] = perm[i];
  }
  const uint32_t nb = (N + PER - 1) / PER;
  // slot8: launch slot b (XCD b mod 8) takes the next bucket whose slot is congruent to b mod 8
  std::vector<std::vector<uint32_t>> by(8);
  for (uint32_t bk = 0; bk < nb; ++bk) by[((bk * PER) / B) % 8].push_back(bk);
  size_t mx = 0; for (auto& v : by) mx = std::max(mx, v.size());
  bob.assign(mx * 8, 0xFFFFFFFFu);
  for (int x = 0; x < 8; ++x) for (size_t i = 0; i < by[x].size(); ++i) bob[i * 8 + x] = by[x][i];
  uint32_t *d_row, *d_bob, *d_index; uint2* d_uw; float* out;
  const uint32_t U = 147000;
  CK(hipMalloc(&d_row, N * 4)); CK(hipMemcpy(d_row, row_of.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_bob, bob.size() * 4)); CK(hipMemcpy(d_bob, bob.data(), bob.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_index, N * 4 * 8)); CK(hipMemset(d_index, 0, N * 4 * 8));
  CK(hipMalloc(&d_uw, U * 8)); CK(hipMemset(d_uw, 0, U * 8));
  CK(hipMalloc(&out, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 8; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 64; ++i) launch(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s %7.2f us\n", name, ms / 64 * 1e3);
  };
  // 8 copies of the index array cycled: a launch never finds its lines from the previous one in an L2
  timeit("W row    index[row * 39 + slot]", [&](int i) { hipLaunchKernelGGL(k_write<0>, dim3(nb), dim3(256), 0, 0, d_row, d_bob, d_index + (size_t)(i % 8) * N, nb); });
  timeit("W slot   index[slot * B + row]", [&](int i) { hipLaunchKernelGGL(k_write<1>, dim3(nb), dim3(256), 0, 0, d_row, d_bob, d_index + (size_t)(i % 8) * N, nb); });
  timeit("W slot8  slot-major, one XCD per slot", [&](int i) { hipLaunchKernelGGL(k_write<2>, dim3((unsigned)bob.size()), dim3(256), 0, 0, d_row, d_bob, d_index + (size_t)(i % 8) * N, nb); });
  timeit("R row    39 consecutive words + {row, w} gather", [&](int i) { hipLaunchKernelGGL(k_read<0>, dim3(B / 4), dim3(256), 0, 0, d_index + (size_t)(i % 8) * N, d_uw, U, out); });
  timeit("R slot   39 words at stride B + {row, w} gather", [&](int i) { hipLaunchKernelGGL(k_read<1>, dim3(B / 4), dim3(256), 0, 0, d_index + (size_t)(i % 8) * N, d_uw, U, out); });
  return 0;
}
