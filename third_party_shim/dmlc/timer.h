// Stand-in for dmlc-core's timer.h
#ifndef SHIM_DMLC_TIMER_H_
#define SHIM_DMLC_TIMER_H_
#include <chrono>
namespace dmlc {
inline double GetTime() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace dmlc
#endif  // SHIM_DMLC_TIMER_H_
