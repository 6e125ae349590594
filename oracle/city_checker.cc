// oracle/city_checker.cc — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// CityHash64 (Google cityhash v1.1, city.cc) restated from the published algorithm for the checker side:
// it serves `CityHash64` to the reference's criteo parser compiled into oracle/_ref (ref_shim/city.h).
// Parity status: the library is absent from this image and the reference holds no vectors for it; the tests
// cross-check this text, the Python transcription in oracle/ingest.py and the product's host/cityhash.h against each
// other, and (round 4) the latter two against Abseil's hash_internal::CityHash64 — Google's code of CityHash v1.1,
// found inside pyarrow's libarrow_compute.so — on every length class (tests/test_ingest.py).
#include <cstdint>
#include <cstring>
#include <utility>

#include "ref_shim/city.h"

namespace {
typedef uint64_t u64;
typedef std::pair<u64, u64> u128;
const u64 k0 = 0xc3a5c85c97cb3127ULL, k1 = 0xb492b66fbe98f273ULL, k2 = 0x9ae16a3b2f90404fULL;

inline u64 F64(const char* p) { u64 r; memcpy(&r, p, 8); return r; }   // little-endian host
inline u64 F32(const char* p) { uint32_t r; memcpy(&r, p, 4); return r; }
inline u64 Rot(u64 v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
inline u64 Mix(u64 v) { return v ^ (v >> 47); }
inline u64 Swap(u64 v) { return __builtin_bswap64(v); }
inline u64 H16(u64 u, u64 v, u64 mul) {
  u64 a = (u ^ v) * mul;
  a ^= (a >> 47);
  u64 b = (v ^ a) * mul;
  b ^= (b >> 47);
  return b * mul;
}
inline u64 H16(u64 u, u64 v) { return H16(u, v, 0x9ddfea08eb382d69ULL); }

u64 Len0to16(const char* s, size_t len) {
  if (len >= 8) {
    u64 mul = k2 + len * 2, a = F64(s) + k2, b = F64(s + len - 8);
    u64 c = Rot(b, 37) * mul + a, d = (Rot(a, 25) + b) * mul;
    return H16(c, d, mul);
  }
  if (len >= 4) {
    u64 mul = k2 + len * 2, a = F32(s);
    return H16(len + (a << 3), F32(s + len - 4), mul);
  }
  if (len > 0) {
    uint8_t a = s[0], b = s[len >> 1], c = s[len - 1];
    uint32_t y = (uint32_t)a + ((uint32_t)b << 8), z = (uint32_t)len + ((uint32_t)c << 2);
    return Mix(y * k2 ^ z * k0) * k2;
  }
  return k2;
}
u64 Len17to32(const char* s, size_t len) {
  u64 mul = k2 + len * 2, a = F64(s) * k1, b = F64(s + 8), c = F64(s + len - 8) * mul, d = F64(s + len - 16) * k2;
  return H16(Rot(a + b, 43) + Rot(c, 30) + d, a + Rot(b + k2, 18) + c, mul);
}
u128 Weak(u64 w, u64 x, u64 y, u64 z, u64 a, u64 b) {
  a += w;
  b = Rot(b + a + z, 21);
  u64 c = a;
  a += x;
  a += y;
  b += Rot(a, 44);
  return u128(a + z, b + c);
}
u128 Weak(const char* s, u64 a, u64 b) { return Weak(F64(s), F64(s + 8), F64(s + 16), F64(s + 24), a, b); }
u64 Len33to64(const char* s, size_t len) {
  u64 mul = k2 + len * 2, a = F64(s) * k2, b = F64(s + 8), c = F64(s + len - 24), d = F64(s + len - 32);
  u64 e = F64(s + 16) * k2, f = F64(s + 24) * 9, g = F64(s + len - 8), h = F64(s + len - 16) * mul;
  u64 u = Rot(a + g, 43) + (Rot(b, 30) + c) * 9, v = ((a + g) ^ d) + f + 1, w = Swap((u + v) * mul) + h;
  u64 x = Rot(e + f, 42) + c, y = (Swap((v + w) * mul) + g) * mul, z = e + f + c;
  a = Swap((x + z) * mul + y) + b;
  b = Mix((z + a) * mul + d + h) * mul;
  return b + x;
}
}  // namespace

uint64_t CityHash64(const char* s, size_t len) {
  if (len <= 32) return len <= 16 ? Len0to16(s, len) : Len17to32(s, len);
  if (len <= 64) return Len33to64(s, len);
  u64 x = F64(s + len - 40), y = F64(s + len - 16) + F64(s + len - 56);
  u64 z = H16(F64(s + len - 48) + len, F64(s + len - 24));
  u128 v = Weak(s + len - 64, len, z), w = Weak(s + len - 32, y + k1, x);
  x = x * k1 + F64(s);
  len = (len - 1) & ~static_cast<size_t>(63);
  do {
    x = Rot(x + y + v.first + F64(s + 8), 37) * k1;
    y = Rot(y + v.second + F64(s + 48), 42) * k1;
    x ^= w.second;
    y += v.first + F64(s + 40);
    z = Rot(z + w.first, 33) * k1;
    v = Weak(s, v.second * k1, x + w.first);
    w = Weak(s + 32, z + w.second, y + F64(s + 16));
    std::swap(z, x);
    s += 64;
    len -= 64;
  } while (len != 0);
  return H16(H16(v.first, w.first) + Mix(y) * k1 + z, H16(v.second, w.second) + x);
}
