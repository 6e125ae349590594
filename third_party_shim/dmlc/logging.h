// Minimal stand-in for dmlc-core's logging.h (dmlc-core is an un-vendored,
// absent submodule of the reference).  Written from scratch: only the macros
// the FM/SGD hot path uses — CHECK*, CHECK_NOTNULL, LOG(sev) — with the
// reference build's semantics (-DDMLC_LOG_FATAL_THROW=0: FATAL aborts).
#ifndef SHIM_DMLC_LOGGING_H_
#define SHIM_DMLC_LOGGING_H_
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <sstream>
#include <string>
#include <stdexcept>

namespace dmlc {

/*! \brief thrown instead of abort() when DMLC_LOG_FATAL_THROW != 0 */
struct Error : public std::runtime_error {
  explicit Error(const std::string& s) : std::runtime_error(s) {}
};

#ifndef DMLC_LOG_FATAL_THROW
#define DMLC_LOG_FATAL_THROW 0
#endif

namespace shim {
enum Severity { kINFO = 0, kWARNING = 1, kERROR = 2, kFATAL = 3 };

class LogLine {
 public:
  LogLine(const char* file, int line, Severity sev) : sev_(sev) {
    static const char* names[] = {"INFO", "WARNING", "ERROR", "FATAL"};
    time_t t = time(nullptr);
    struct tm tmv;
    localtime_r(&t, &tmv);
    char buf[16];
    snprintf(buf, sizeof(buf), "%02d:%02d:%02d", tmv.tm_hour, tmv.tm_min, tmv.tm_sec);
    os_ << "[" << buf << "] " << names[sev] << " " << file << ":" << line << ": ";
  }
  std::ostream& stream() { return os_; }
#if DMLC_LOG_FATAL_THROW
  ~LogLine() noexcept(false) {
#else
  ~LogLine() {
#endif
    os_ << "\n";
    std::cerr << os_.str();
    std::cerr.flush();
    if (sev_ == kFATAL) {
#if DMLC_LOG_FATAL_THROW
      throw Error(os_.str());
#else
      abort();
#endif
    }
  }
 private:
  std::ostringstream os_;
  Severity sev_;
};

// swallows a stream expression so `cond ? (void)0 : Voidify() & LOG...` types
struct Voidify {
  void operator&(std::ostream&) {}
};

template <typename T>
inline T CheckNotNull(const char* file, int line, const char* expr, T&& p) {
  if (p == nullptr) {
    LogLine(file, line, kFATAL).stream() << "Check notnull: " << expr;
  }
  return std::forward<T>(p);
}
}  // namespace shim
}  // namespace dmlc

#define LOG_INFO    ::dmlc::shim::LogLine(__FILE__, __LINE__, ::dmlc::shim::kINFO)
#define LOG_WARNING ::dmlc::shim::LogLine(__FILE__, __LINE__, ::dmlc::shim::kWARNING)
#define LOG_ERROR   ::dmlc::shim::LogLine(__FILE__, __LINE__, ::dmlc::shim::kERROR)
#define LOG_FATAL   ::dmlc::shim::LogLine(__FILE__, __LINE__, ::dmlc::shim::kFATAL)
#define LOG(sev) LOG_##sev.stream()

#define CHECK(cond)                                                    \
  (cond) ? (void)0 : ::dmlc::shim::Voidify() &                         \
      LOG(FATAL) << "Check failed: " #cond << ' '

#define SHIM_CHECK_OP(op, a, b)                                        \
  ((a) op (b)) ? (void)0 : ::dmlc::shim::Voidify() &                   \
      LOG(FATAL) << "Check failed: " #a " " #op " " #b << " (" << (a)  \
                 << " vs " << (b) << ") "

#define CHECK_EQ(a, b) SHIM_CHECK_OP(==, a, b)
#define CHECK_NE(a, b) SHIM_CHECK_OP(!=, a, b)
#define CHECK_LT(a, b) SHIM_CHECK_OP(<, a, b)
#define CHECK_LE(a, b) SHIM_CHECK_OP(<=, a, b)
#define CHECK_GT(a, b) SHIM_CHECK_OP(>, a, b)
#define CHECK_GE(a, b) SHIM_CHECK_OP(>=, a, b)
#define CHECK_NOTNULL(p) \
  ::dmlc::shim::CheckNotNull(__FILE__, __LINE__, #p, (p))

#endif  // SHIM_DMLC_LOGGING_H_
