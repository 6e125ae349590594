/**
 * device_context.h — process-wide handle on the GPU (dfh_ctx) shared by the C++
 * adaptors, and the error bridge from the C ABI's return codes to the
 * reference's error convention (CHECK / LOG(FATAL): abort with a message).
 */
#ifndef DIFACTO_HOST_DEVICE_CONTEXT_H_
#define DIFACTO_HOST_DEVICE_CONTEXT_H_
#include <cstdlib>
#include "difacto/base.h"
#include "difacto_hip.h"
#include "dmlc/logging.h"

// The device Localizer (dfh_localize) always sorts by ReverseBytes(id) and the sharded store's key ranges rely on it
// (include/difacto/base.h:29-51).  A host built with the reference's NO_REVERSE_ID=1 (Makefile:15-17: REVERSE_FEATURE_ID=0)
// would hand the device ids the host headers no longer reverse: refused here, at compile time, instead of training on a
// differently ordered model (SURVEY 8a trap 11).
#if !REVERSE_FEATURE_ID
#error "the device path needs REVERSE_FEATURE_ID=1: build the host without NO_REVERSE_ID (the device Localizer reverses ids itself)"
#endif

namespace difacto {

#define DFH_CALL(expr)                                                            \
  do {                                                                            \
    int rc__ = (expr);                                                            \
    CHECK_EQ(rc__, DFH_OK) << #expr << " failed: " << dfh_last_error();           \
  } while (0)

class DeviceContext {
 public:
  /*! \brief the context on device $DIFACTO_DEVICE (default 0); created on first use */
  static dfh_ctx* Get() {
    static DeviceContext inst;
    return inst.ctx_;
  }

 private:
  DeviceContext() {
    const char* d = getenv("DIFACTO_DEVICE");
    int dev = d ? atoi(d) : 0;
    DFH_CALL(dfh_ctx_create(dev, nullptr, &ctx_));
  }
  ~DeviceContext() { dfh_ctx_destroy(ctx_); }
  dfh_ctx* ctx_ = nullptr;
};

}  // namespace difacto
#endif  // DIFACTO_HOST_DEVICE_CONTEXT_H_
