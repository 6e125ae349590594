// oracle/ref_shim/data/strtonum.h — TEST INFRASTRUCTURE.  criteo_parser.h includes dmlc-core's strtonum.h but
// calls only libc's atof (src/reader/criteo_parser.h:63).
#ifndef ORACLE_REF_SHIM_DATA_STRTONUM_H_
#define ORACLE_REF_SHIM_DATA_STRTONUM_H_
#include <cctype>
#include <cstdlib>
#endif  // ORACLE_REF_SHIM_DATA_STRTONUM_H_
