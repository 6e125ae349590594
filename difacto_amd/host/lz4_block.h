/**
 * lz4_block.h — decoder for the LZ4 BLOCK format (the payload of LZ4_compress_default /
 * LZ4_decompress_safe, which the reference's CompressedRowBlock uses per array:
 * src/data/compressed_row_block.h:104-133), written from the published block-format description:
 * a block is a sequence of
 *     token (hi nibble: literal length, lo nibble: match length - 4; 15 = more length bytes follow,
 *     each adding 0..255, the run ending at the first byte < 255)
 *     literals | 16-bit little-endian match offset (1..65535, into the already decoded output)
 * and ends with a literals-only sequence.  Matches may overlap their own output (offset < length).
 * Bounds are checked on every step: a malformed block returns -1, never reads or writes outside.
 * Checked against the real liblz4 of this image in tests/test_ingest.py (fixtures written by
 * LZ4_compress_default through ctypes).
 */
#ifndef DIFACTO_HOST_LZ4_BLOCK_H_
#define DIFACTO_HOST_LZ4_BLOCK_H_
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace difacto {

/*! \brief returns the number of bytes written to dst (<= dst_cap), or -1 on a malformed block */
inline long Lz4DecompressBlock(const char* src, size_t src_size, char* dst, size_t dst_cap) {
  const uint8_t* ip = reinterpret_cast<const uint8_t*>(src);
  const uint8_t* const iend = ip + src_size;
  uint8_t* op = reinterpret_cast<uint8_t*>(dst);
  uint8_t* const oend = op + dst_cap;
  if (src_size == 0) return dst_cap == 0 ? 0 : -1;
  for (;;) {
    if (ip >= iend) return -1;
    const unsigned token = *ip++;
    size_t lit = token >> 4;
    // short sequence with room to spare on both sides (the bulk of a block of ids: a few literal bytes, a short match):
    // 16 literal bytes and 18 match bytes are copied whatever the lengths say, the pointers advance by the lengths.
    // Row blocks of random 64-bit ids are thousands of such sequences; with a libc memcpy per piece the decoder ran at
    // 1.5 GB/s (tools/gpu_r04v.sh: 200 ns per row of the end-to-end .rec file)
    if (lit != 15 && (token & 15) != 15 && static_cast<size_t>(iend - ip) >= 16 + 2 && static_cast<size_t>(oend - op) >= 16 + 18 + 15) {
      memcpy(op, ip, 16);
      ip += lit;
      op += lit;
      const size_t offset = static_cast<size_t>(ip[0]) | (static_cast<size_t>(ip[1]) << 8);
      ip += 2;
      if (offset == 0 || offset > static_cast<size_t>(op - reinterpret_cast<uint8_t*>(dst))) return -1;
      const size_t mlen = (token & 15) + 4;   // 4 .. 18
      const uint8_t* match = op - offset;
      if (offset >= 16) {
        memcpy(op, match, 16);
        memcpy(op + 16, match + 16, 2);
      } else if (offset >= 8) {
        memcpy(op, match, 8);
        memcpy(op + 8, match + 8, 8);
        memcpy(op + 16, match + 16, 2);
      } else {
        for (size_t i = 0; i < mlen; ++i) op[i] = match[i];  // overlapping copy: byte by byte
      }
      op += mlen;
      continue;
    }
    if (lit == 15) {
      uint8_t b;
      do {
        if (ip >= iend) return -1;
        b = *ip++;
        lit += b;
      } while (b == 255);
    }
    if (lit > static_cast<size_t>(iend - ip) || lit > static_cast<size_t>(oend - op)) return -1;
    memcpy(op, ip, lit);
    ip += lit;
    op += lit;
    if (ip == iend) break;  // the last sequence carries literals only
    if (iend - ip < 2) return -1;
    const size_t offset = static_cast<size_t>(ip[0]) | (static_cast<size_t>(ip[1]) << 8);
    ip += 2;
    if (offset == 0 || offset > static_cast<size_t>(op - reinterpret_cast<uint8_t*>(dst))) return -1;
    size_t mlen = token & 15;
    if (mlen == 15) {
      uint8_t b;
      do {
        if (ip >= iend) return -1;
        b = *ip++;
        mlen += b;
      } while (b == 255);
    }
    mlen += 4;
    if (mlen > static_cast<size_t>(oend - op)) return -1;
    const uint8_t* match = op - offset;
    if (offset >= mlen) {
      memcpy(op, match, mlen);
      op += mlen;
    } else {
      for (size_t i = 0; i < mlen; ++i) *op++ = *match++;  // overlapping copy: byte by byte
    }
  }
  return static_cast<long>(op - reinterpret_cast<uint8_t*>(dst));
}

}  // namespace difacto
#endif  // DIFACTO_HOST_LZ4_BLOCK_H_
