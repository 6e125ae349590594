// oracle/ref_shim/city.h — TEST INFRASTRUCTURE.  What `#include <city.h>` resolves to when the reference's
// src/reader/criteo_parser.h is compiled into oracle/_ref with DIFACTO_USE_CITY=1: Google's cityhash is not
// installed on this image, so the declaration is served by oracle/city_checker.cc, a
// from-scratch transcription of the published CityHash64 v1.1 algorithm kept on the checker side.  It is
// independent text from the product's difacto_amd/host/cityhash.h; the tests compare the two (and the Python
// transcription in oracle/ingest.py).  CityHash64 itself stays UNPINNED beyond CityHash64("") = k2: no
// reference build of the library and no published vectors exist here.
#ifndef ORACLE_REF_SHIM_CITY_H_
#define ORACLE_REF_SHIM_CITY_H_
#include <cstddef>
#include <cstdint>
uint64_t CityHash64(const char* buf, size_t len);
#endif  // ORACLE_REF_SHIM_CITY_H_
