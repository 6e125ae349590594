// oracle/ref_shim/city.h — TEST INFRASTRUCTURE.  What `#include <city.h>` resolves to when the reference's
// src/reader/criteo_parser.h is compiled into oracle/_ref with DIFACTO_USE_CITY=1: Google's cityhash is not
// installed on this image, so the declaration is served by oracle/city_checker.cc, a
// from-scratch transcription of the published CityHash64 v1.1 algorithm kept on the checker side.  It is
// independent text from the product's difacto_amd/host/cityhash.h; the tests compare the two (and the Python
// transcription in oracle/ingest.py).  The library itself is absent, but Google's code of the same algorithm is in
// the image (Abseil's hash_internal::CityHash64 inside pyarrow's libarrow_compute.so): the product's and the Python
// transcription are pinned to it (tests/test_ingest.py::test_cityhash64_against_abseil), this one through them.
#ifndef ORACLE_REF_SHIM_CITY_H_
#define ORACLE_REF_SHIM_CITY_H_
#include <cstddef>
#include <cstdint>
uint64_t CityHash64(const char* buf, size_t len);
#endif  // ORACLE_REF_SHIM_CITY_H_
