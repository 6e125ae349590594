// Force-included (-include) when compiling the reference's own sources for
// oracle/_ref: standard headers the reference picks up transitively through
// the real dmlc-core / ps-lite headers, which our shims do not drag in.
#ifndef ORACLE_REF_PRELUDE_H_
#define ORACLE_REF_PRELUDE_H_
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <unordered_map>
#endif
