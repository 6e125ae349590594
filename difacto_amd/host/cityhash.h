/**
 * cityhash.h — CityHash64, written from the published algorithm (Pike & Alakuijala, CityHash v1.1,
 * city.cc: HashLen0to16 / HashLen17to32 / HashLen33to64 / the 64-byte main loop).
 *
 * The reference hashes every criteo token with it (src/reader/criteo_parser.h:96-101, behind
 * USE_CITY=1; the library is the third-party dependency cityhash 1.1.1 that dmlc-core's build
 * downloads — absent here, like dmlc-core itself, and there is no network).  Parity status: pinned
 * (round 4) to Google's own code of the algorithm found in the image — Abseil's
 * hash_internal::CityHash64 (CityHash v1.1) inside pyarrow's libarrow_compute.so — on every length
 * class (tests/test_ingest.py::test_cityhash64_against_abseil), besides the published constant
 * CityHash64("") = k2 = 0x9ae16a3b2f90404f and an independent Python transcription (oracle/ingest.py).
 * Criteo tokens are 1-16 bytes long and take the HashLen0to16 branch only.
 */
#ifndef DIFACTO_HOST_CITYHASH_H_
#define DIFACTO_HOST_CITYHASH_H_
#include <cstdint>
#include <cstring>
#include <utility>

namespace difacto {
namespace city {

static const uint64_t k0 = 0xc3a5c85c97cb3127ULL;
static const uint64_t k1 = 0xb492b66fbe98f273ULL;
static const uint64_t k2 = 0x9ae16a3b2f90404fULL;

inline uint64_t Fetch64(const char* p) { uint64_t r; memcpy(&r, p, 8); return r; }   // little-endian hosts
inline uint32_t Fetch32(const char* p) { uint32_t r; memcpy(&r, p, 4); return r; }
inline uint64_t Rotate(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
inline uint64_t ShiftMix(uint64_t v) { return v ^ (v >> 47); }
inline uint64_t Bswap64(uint64_t v) { return __builtin_bswap64(v); }

inline uint64_t HashLen16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul;
  b ^= (b >> 47);
  b *= mul;
  return b;
}
inline uint64_t HashLen16(uint64_t u, uint64_t v) { return HashLen16(u, v, 0x9ddfea08eb382d69ULL); }

inline uint64_t HashLen0to16(const char* s, size_t len) {
  if (len >= 8) {
    const uint64_t mul = k2 + len * 2;
    const uint64_t a = Fetch64(s) + k2;
    const uint64_t b = Fetch64(s + len - 8);
    const uint64_t c = Rotate(b, 37) * mul + a;
    const uint64_t d = (Rotate(a, 25) + b) * mul;
    return HashLen16(c, d, mul);
  }
  if (len >= 4) {
    const uint64_t mul = k2 + len * 2;
    const uint64_t a = Fetch32(s);
    return HashLen16(len + (a << 3), Fetch32(s + len - 4), mul);
  }
  if (len > 0) {
    const uint8_t a = static_cast<uint8_t>(s[0]);
    const uint8_t b = static_cast<uint8_t>(s[len >> 1]);
    const uint8_t c = static_cast<uint8_t>(s[len - 1]);
    const uint32_t y = static_cast<uint32_t>(a) + (static_cast<uint32_t>(b) << 8);
    const uint32_t z = static_cast<uint32_t>(len) + (static_cast<uint32_t>(c) << 2);
    return ShiftMix(y * k2 ^ z * k0) * k2;
  }
  return k2;
}

inline uint64_t HashLen17to32(const char* s, size_t len) {
  const uint64_t mul = k2 + len * 2;
  const uint64_t a = Fetch64(s) * k1;
  const uint64_t b = Fetch64(s + 8);
  const uint64_t c = Fetch64(s + len - 8) * mul;
  const uint64_t d = Fetch64(s + len - 16) * k2;
  return HashLen16(Rotate(a + b, 43) + Rotate(c, 30) + d, a + Rotate(b + k2, 18) + c, mul);
}

inline std::pair<uint64_t, uint64_t> WeakHashLen32WithSeeds(uint64_t w, uint64_t x, uint64_t y, uint64_t z, uint64_t a, uint64_t b) {
  a += w;
  b = Rotate(b + a + z, 21);
  const uint64_t c = a;
  a += x;
  a += y;
  b += Rotate(a, 44);
  return std::make_pair(a + z, b + c);
}
inline std::pair<uint64_t, uint64_t> WeakHashLen32WithSeeds(const char* s, uint64_t a, uint64_t b) {
  return WeakHashLen32WithSeeds(Fetch64(s), Fetch64(s + 8), Fetch64(s + 16), Fetch64(s + 24), a, b);
}

inline uint64_t HashLen33to64(const char* s, size_t len) {
  const uint64_t mul = k2 + len * 2;
  uint64_t a = Fetch64(s) * k2;
  uint64_t b = Fetch64(s + 8);
  const uint64_t c = Fetch64(s + len - 24);
  const uint64_t d = Fetch64(s + len - 32);
  const uint64_t e = Fetch64(s + 16) * k2;
  const uint64_t f = Fetch64(s + 24) * 9;
  const uint64_t g = Fetch64(s + len - 8);
  const uint64_t h = Fetch64(s + len - 16) * mul;
  const uint64_t u = Rotate(a + g, 43) + (Rotate(b, 30) + c) * 9;
  const uint64_t v = ((a + g) ^ d) + f + 1;
  const uint64_t w = Bswap64((u + v) * mul) + h;
  const uint64_t x = Rotate(e + f, 42) + c;
  const uint64_t y = (Bswap64((v + w) * mul) + g) * mul;
  const uint64_t z = e + f + c;
  a = Bswap64((x + z) * mul + y) + b;
  b = ShiftMix((z + a) * mul + d + h) * mul;
  return b + x;
}

}  // namespace city

inline uint64_t CityHash64(const char* s, size_t len) {
  using namespace city;
  if (len <= 32) return len <= 16 ? HashLen0to16(s, len) : HashLen17to32(s, len);
  if (len <= 64) return HashLen33to64(s, len);
  // strings over 64 bytes: hash the end first, then loop over 64-byte chunks keeping 56 bytes of state
  uint64_t x = Fetch64(s + len - 40);
  uint64_t y = Fetch64(s + len - 16) + Fetch64(s + len - 56);
  uint64_t z = HashLen16(Fetch64(s + len - 48) + len, Fetch64(s + len - 24));
  std::pair<uint64_t, uint64_t> v = WeakHashLen32WithSeeds(s + len - 64, len, z);
  std::pair<uint64_t, uint64_t> w = WeakHashLen32WithSeeds(s + len - 32, y + k1, x);
  x = x * k1 + Fetch64(s);
  len = (len - 1) & ~static_cast<size_t>(63);
  do {
    x = Rotate(x + y + v.first + Fetch64(s + 8), 37) * k1;
    y = Rotate(y + v.second + Fetch64(s + 48), 42) * k1;
    x ^= w.second;
    y += v.first + Fetch64(s + 40);
    z = Rotate(z + w.first, 33) * k1;
    v = WeakHashLen32WithSeeds(s, v.second * k1, x + w.first);
    w = WeakHashLen32WithSeeds(s + 32, z + w.second, y + Fetch64(s + 16));
    std::swap(z, x);
    s += 64;
    len -= 64;
  } while (len != 0);
  return HashLen16(HashLen16(v.first, w.first) + ShiftMix(y) * k1 + z, HashLen16(v.second, w.second) + x);
}

}  // namespace difacto
#endif  // DIFACTO_HOST_CITYHASH_H_
