// oracle/ref_shim/data/strtonum.h — TEST INFRASTRUCTURE.  Stand-in for dmlc-core's src/data/strtonum.h (absent
// submodule).  criteo_parser.h includes it but calls only libc's atof (src/reader/criteo_parser.h:63);
// adfea_parser.h (src/reader/adfea_parser.h:38-40, :58-75) uses dmlc::data::isspace / isdigit / strtoull.  These are
// restated from the header's published behaviour, not copied: character classes without locale, and an unsigned
// parse that skips leading blanks, takes an optional sign, accumulates digits WITHOUT an overflow check (it wraps;
// libc's saturates) and reports where it stopped.
#ifndef ORACLE_REF_SHIM_DATA_STRTONUM_H_
#define ORACLE_REF_SHIM_DATA_STRTONUM_H_
#include <cctype>
#include <cstdint>
#include <cstdlib>
namespace dmlc {
namespace data {
inline bool isspace(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f'; }
inline bool isdigit(char c) { return c >= '0' && c <= '9'; }
inline uint64_t strtoull(const char* nptr, char** endptr, int base) {
  const char* p = nptr;
  while (isspace(*p)) ++p;
  bool positive = true;
  if (*p == '-') { positive = false; ++p; } else if (*p == '+') { ++p; }
  uint64_t value = 0;
  for (; isdigit(*p); ++p) value = value * static_cast<uint64_t>(base) + static_cast<uint64_t>(*p - '0');
  if (endptr) *endptr = const_cast<char*>(p);
  return positive ? value : static_cast<uint64_t>(-static_cast<int64_t>(value));
}
}  // namespace data
}  // namespace dmlc
#endif  // ORACLE_REF_SHIM_DATA_STRTONUM_H_
