/**
 * difacto/base.h — scalar types, keyword arguments and the feature-id
 * transforms of the FM/SGD path.
 *
 * Interface-compatible re-statement of the reference's include/difacto/base.h
 * (same names, same semantics) so that code written against dmlc/difacto
 * compiles against this tree unchanged.  Written for the MI355X build; no
 * reference source text is reproduced.
 */
#ifndef DIFACTO_BASE_H_
#define DIFACTO_BASE_H_
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <utility>
#include <vector>
#include "dmlc/logging.h"

namespace difacto {

/*! \brief weights, gradients and predictions are fp32 (reference: base.h:16) */
typedef float real_t;
/*! \brief feature ids are 64-bit (reference: base.h:20) */
typedef uint64_t feaid_t;
/*! \brief ordered key=value list; Init() methods consume what they know and return the rest */
typedef std::vector<std::pair<std::string, std::string>> KWArgs;

/*! \brief worker-side thread count used where the reference uses OpenMP (base.h:28) */
#define DEFAULT_NTHREADS 2

#ifndef REVERSE_FEATURE_ID
#define REVERSE_FEATURE_ID 1
#endif

/**
 * \brief spread feature ids over the 64-bit key space by reversing their
 *        nibbles (32/16/8/4-bit group swaps), an involution.  Range-partitioning
 *        the reversed ids is what keeps model shards balanced (reference: base.h:39-51).
 */
inline feaid_t ReverseBytes(feaid_t x) {
#if REVERSE_FEATURE_ID
  x = (x << 32) | (x >> 32);
  x = ((x & 0x0000FFFF0000FFFFULL) << 16) | ((x & 0xFFFF0000FFFF0000ULL) >> 16);
  x = ((x & 0x00FF00FF00FF00FFULL) << 8) | ((x & 0xFF00FF00FF00FF00ULL) >> 8);
  x = ((x & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((x & 0xF0F0F0F0F0F0F0F0ULL) >> 4);
#endif
  return x;
}

/*! \brief tag a feature id with its group (slot) id in the low nbits (reference: base.h:60-63) */
inline feaid_t EncodeFeaGrpID(feaid_t x, int gid, int nbits) {
  CHECK_GE(gid, 0);
  CHECK_LT(gid, 1 << nbits);
  return (x << nbits) | static_cast<feaid_t>(gid);
}

/*! \brief recover the group id (reference: base.h:71-73) */
inline feaid_t DecodeFeaGrpID(feaid_t x, int nbits) { return x % (1 << nbits); }

/*! \brief process role, from DMLC_ROLE (reference: base.h:76-84) */
inline char* GetRole() { return getenv("DMLC_ROLE"); }
inline bool IsDistributed() { return GetRole() != nullptr; }
inline bool IsScheduler() { return !IsDistributed() || !strcmp(GetRole(), "scheduler"); }
inline bool IsWorker() { return !IsDistributed() || !strcmp(GetRole(), "worker"); }
inline bool IsServer() { return !IsDistributed() || !strcmp(GetRole(), "server"); }

#define LL LOG(ERROR)

/*! \brief "[n]: a b c ... y z" */
template <typename V>
inline std::string DebugStr(const V* data, int n, int m = 5) {
  std::stringstream ss;
  ss << "[" << n << "]: ";
  for (int i = 0; i < n; ++i) {
    if (n > 2 * m && i == m) {
      ss << "... ";
      i = n - m;
    }
    ss << data[i] << " ";
  }
  return ss.str();
}
template <typename Vec>
inline std::string DebugStr(const Vec& vec) { return DebugStr(vec.data(), static_cast<int>(vec.size())); }

/*! \brief squared 2-norm, accumulated in double */
template <typename Vec>
inline real_t Norm2(const Vec& vec) {
  double n = 0;
  for (real_t v : vec) n += static_cast<double>(v) * v;
  return static_cast<real_t>(n);
}

}  // namespace difacto
#endif  // DIFACTO_BASE_H_
