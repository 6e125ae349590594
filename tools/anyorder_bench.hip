// Can two kernels of ONE stream run side by side?  hipExtLaunchKernelGGL(..., flags = hipExtAnyOrderLaunch) asks the runtime to
// queue a dispatch WITHOUT the AQL barrier bit: the command processor starts it as soon as it reaches the packet, without
// waiting for the packets before it to complete; the next ordinary dispatch (barrier bit set) waits for everything before it.
// If that works on gfx950, the single-queue step can run the Localizer's stages beside the step's own launches with no second
// hardware queue and no events (hip_ext.h says "not supported on GFX9xx" for the module form of the call).
//   pair   = [A ordinary][B any-order]: T us each -> T (concurrent) or 2 T (serial)?
//   order  = [A ordinary][B any-order][C ordinary]: C must see what A and B wrote.
// hipcc --offload-arch=gfx950 -O3 -o tools/anyorder_bench.bin tools/anyorder_bench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_spin(unsigned long long ticks, unsigned* flag, unsigned val) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (flag && threadIdx.x == 0 && blockIdx.x == 0) *flag = val;
}
__global__ void k_check(const unsigned* fa, const unsigned* fb, unsigned val, unsigned* bad) {
  if (threadIdx.x == 0 && (*fa != val || *fb != val)) atomicAdd(bad, 1u);
}

int main() {
  int khz = 100000;
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* d; CK(hipMalloc(&d, 1024)); CK(hipMemset(d, 0, 1024));
  unsigned *fa = d, *fb = d + 64, *bad = d + 128;
  const int R = 200;
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (double T : {5.0, 20.0}) {
    const unsigned long long ticks = (unsigned long long)(T * 1e-3 * khz);
    for (int blocks : {1, 256, 2048}) {
      for (int mode = 0; mode < 3; ++mode) {   // 0: both ordinary; 1: B any-order; 2: A any-order too
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipDeviceSynchronize()); CK(hipEventRecord(t0, s));
          for (int i = 0; i < R; ++i) {
            hipExtLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, ticks, fa, (unsigned)i);
            hipExtLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, s, nullptr, nullptr, mode >= 1 ? hipExtAnyOrderLaunch : 0, ticks, fb, (unsigned)i);
            hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, s, fa, fb, (unsigned)i, bad);
          }
          CK(hipEventRecord(t1, s)); CK(hipEventSynchronize(t1));
          CK(hipEventElapsedTime(&ms, t0, t1));
        }
        unsigned nbad = 0; CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemset(bad, 0, 4));
        printf("T %4.0f us, %4d blocks, %-28s %7.2f us per [A][B][check] round, %u ordering violations\n", T, blocks,
               mode == 0 ? "A, B ordinary" : mode == 1 ? "B any-order" : "A and B any-order", ms / R * 1e3, nbad);
      }
    }
  }
  return 0;
}
