// What does it cost a stream to hand over to another stream of the same device?  (DESIGN 4 "lessons": an event record or a
// cross-stream wait costs the stream ~10-13 us — the overlapped sharded step pays eight of them, the pipelined single-GPU step
// four.)  Two streams play ping-pong: A runs a kernel of ~T us, signals; B waits, runs a kernel of ~T us, signals back; A waits.
// Time per round trip minus 2 T = the cost of two hand-overs.  Mechanisms:
//   events (default flags) / events (hipEventDisableTiming | hipEventDisableSystemFence: what the library uses)
//   hipStreamWriteValue32 + hipStreamWaitValue32 on signal memory (hipExtMallocWithFlags, hipMallocSignalMemory)
//   one stream, no hand-over (the floor: a kernel boundary)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_spin(unsigned long long ticks, unsigned* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x7fffffffffffffffull) *out = 1;
}

int main() {
  int khz = 100000;
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  unsigned* out; CK(hipMalloc(&out, 256));
  const int R = 200;
  const double T = 5.0;  // us per kernel
  const unsigned long long ticks = (unsigned long long)(T * 1e-3 * khz);
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto report = [&](const char* name, float ms, int kernels_per_round) {
    printf("%-72s %7.2f us per round trip, %6.2f us beyond its %d kernels\n", name, ms / R * 1e3, ms / R * 1e3 - kernels_per_round * T, kernels_per_round);
  };
  // floor: one stream
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipDeviceSynchronize()); CK(hipEventRecord(t0, a));
    for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, out); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, out); }
    CK(hipEventRecord(t1, a)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    if (rep) report("one stream, two kernels per round (no hand-over)", ms, 2);
  }
  for (int mode = 0; mode < 2; ++mode) {
    const unsigned fl = mode ? (hipEventDisableTiming | hipEventDisableSystemFence) : hipEventDefault;
    hipEvent_t ea[2], eb[2];
    for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&ea[i], fl)); CK(hipEventCreateWithFlags(&eb[i], fl)); }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize()); CK(hipEventRecord(t0, a));
      for (int i = 0; i < R; ++i) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, out);
        CK(hipEventRecord(ea[i & 1], a));
        CK(hipStreamWaitEvent(b, ea[i & 1], 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks, out);
        CK(hipEventRecord(eb[i & 1], b));
        CK(hipStreamWaitEvent(a, eb[i & 1], 0));
      }
      CK(hipEventRecord(t1, a)); CK(hipEventSynchronize(t1));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep) report(mode ? "events, DisableTiming | DisableSystemFence (the library's)" : "events, default flags", ms, 2);
    }
  }
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (can) {
    unsigned *sig = nullptr, *sig2 = nullptr;   // (signal memory: exactly 8 bytes per allocation)
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 8, hipMallocSignalMemory) != hipSuccess ||
        hipExtMallocWithFlags(reinterpret_cast<void**>(&sig2), 8, hipMallocSignalMemory) != hipSuccess) {
      printf("hipExtMallocWithFlags(hipMallocSignalMemory) failed: %s\n", hipGetErrorString(hipGetLastError()));
      return 0;
    }
    CK(hipStreamWriteValue32(a, sig, 0, 0)); CK(hipStreamWriteValue32(a, sig2, 0, 0)); CK(hipStreamSynchronize(a));
    unsigned seq = 0;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize()); CK(hipEventRecord(t0, a));
      for (int i = 0; i < R; ++i) {
        ++seq;
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks, out);
        CK(hipStreamWriteValue32(a, sig, seq, 0));
        CK(hipStreamWaitValue32(b, sig, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks, out);
        CK(hipStreamWriteValue32(b, sig2, seq, 0));
        CK(hipStreamWaitValue32(a, sig2, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
      }
      CK(hipEventRecord(t1, a)); CK(hipEventSynchronize(t1));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep) report("hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory", ms, 2);
    }
  }
  return 0;
}
