"""TEST INFRASTRUCTURE ONLY — CPU restatements for the data formats either side of the path
(SURVEY.md 8f rank 3), used by tests/test_ingest.py to check difacto_amd/host/batch_reader.h:

  cityhash64(bytes)       CityHash64 v1.1, transcribed from the published algorithm (the third-party
                          dependency cityhash 1.1.1 of the reference — pulled by dmlc-core's build — is
                          absent here).  Pinned since round 4 to Google's own code of the algorithm: Abseil's
                          hash_internal::CityHash64 (CityHash v1.1), exported by pyarrow's libarrow_compute.so in
                          this image — tests/test_ingest.py::test_cityhash64_against_abseil, every length class.
  parse_criteo(text)      CriteoParser::ParseNext (/root/reference/src/reader/criteo_parser.h:40-94)
  parse_adfea(text)       AdfeaParser::ParseNext (/root/reference/src/reader/adfea_parser.h:33-88)
  lz4_compress(bytes)     LZ4_compress_default of the REAL liblz4 present in this image (ctypes): fixtures
                          for the from-scratch decoder, so that one IS pinned to the library the reference links
  write_crb_record(...)   CompressedRowBlock::Compress (/root/reference/src/data/compressed_row_block.h:26-54)
  write_recordio(...)     dmlc-core RecordIOWriter::WriteRecord (published format: magic, cflag << 29 | length,
                          4-byte padding, records cut at aligned occurrences of the magic word)
"""
import ctypes as C
import ctypes.util
import struct

import numpy as np

M64 = (1 << 64) - 1
K0, K1, K2 = 0xc3a5c85c97cb3127, 0xb492b66fbe98f273, 0x9ae16a3b2f90404f


def _f64(s, i):
    return struct.unpack_from("<Q", s, i)[0]


def _f32(s, i):
    return struct.unpack_from("<I", s, i)[0]


def _rot(v, s):
    return v if s == 0 else ((v >> s) | (v << (64 - s))) & M64


def _smix(v):
    return v ^ (v >> 47)


def _bswap(v):
    return int.from_bytes(v.to_bytes(8, "little"), "big")


def _hl16(u, v, mul=0x9ddfea08eb382d69):
    a = ((u ^ v) * mul) & M64
    a ^= a >> 47
    b = ((v ^ a) * mul) & M64
    b ^= b >> 47
    return (b * mul) & M64


def _weak(w, x, y, z, a, b):
    a = (a + w) & M64
    b = _rot((b + a + z) & M64, 21)
    c = a
    a = (a + x) & M64
    a = (a + y) & M64
    b = (b + _rot(a, 44)) & M64
    return (a + z) & M64, (b + c) & M64


def _weak_s(s, i, a, b):
    return _weak(_f64(s, i), _f64(s, i + 8), _f64(s, i + 16), _f64(s, i + 24), a, b)


def cityhash64(s):
    s = bytes(s)
    n = len(s)
    if n <= 16:
        if n >= 8:
            mul = (K2 + n * 2) & M64
            a = (_f64(s, 0) + K2) & M64
            b = _f64(s, n - 8)
            c = (_rot(b, 37) * mul + a) & M64
            d = ((_rot(a, 25) + b) * mul) & M64
            return _hl16(c, d, mul)
        if n >= 4:
            mul = (K2 + n * 2) & M64
            return _hl16((n + (_f32(s, 0) << 3)) & M64, _f32(s, n - 4), mul)
        if n > 0:
            a, b, c = s[0], s[n >> 1], s[n - 1]
            y = (a + (b << 8)) & 0xFFFFFFFF
            z = (n + (c << 2)) & 0xFFFFFFFF
            return (_smix(((y * K2) & M64) ^ ((z * K0) & M64)) * K2) & M64
        return K2
    if n <= 32:
        mul = (K2 + n * 2) & M64
        a = (_f64(s, 0) * K1) & M64
        b = _f64(s, 8)
        c = (_f64(s, n - 8) * mul) & M64
        d = (_f64(s, n - 16) * K2) & M64
        return _hl16((_rot((a + b) & M64, 43) + _rot(c, 30) + d) & M64, (a + _rot((b + K2) & M64, 18) + c) & M64, mul)
    if n <= 64:
        mul = (K2 + n * 2) & M64
        a = (_f64(s, 0) * K2) & M64
        b = _f64(s, 8)
        c = _f64(s, n - 24)
        d = _f64(s, n - 32)
        e = (_f64(s, 16) * K2) & M64
        f = (_f64(s, 24) * 9) & M64
        g = _f64(s, n - 8)
        h = (_f64(s, n - 16) * mul) & M64
        u = (_rot((a + g) & M64, 43) + ((_rot(b, 30) + c) * 9)) & M64
        v = ((((a + g) & M64) ^ d) + f + 1) & M64
        w = (_bswap(((u + v) * mul) & M64) + h) & M64
        x = (_rot((e + f) & M64, 42) + c) & M64
        y = ((_bswap(((v + w) * mul) & M64) + g) * mul) & M64
        z = (e + f + c) & M64
        a = (_bswap((((x + z) * mul) + y) & M64) + b) & M64
        b = (_smix((((z + a) * mul) + d + h) & M64) * mul) & M64
        return (b + x) & M64
    x = _f64(s, n - 40)
    y = (_f64(s, n - 16) + _f64(s, n - 56)) & M64
    z = _hl16((_f64(s, n - 48) + n) & M64, _f64(s, n - 24))
    v = _weak_s(s, n - 64, n, z)
    w = _weak_s(s, n - 32, (y + K1) & M64, x)
    x = (x * K1 + _f64(s, 0)) & M64
    left = (n - 1) & ~63
    i = 0
    while True:
        x = (_rot((x + y + v[0] + _f64(s, i + 8)) & M64, 37) * K1) & M64
        y = (_rot((y + v[1] + _f64(s, i + 48)) & M64, 42) * K1) & M64
        x ^= w[1]
        y = (y + v[0] + _f64(s, i + 40)) & M64
        z = (_rot((z + w[0]) & M64, 33) * K1) & M64
        v = _weak_s(s, i, (v[1] * K1) & M64, (x + w[0]) & M64)
        w = _weak_s(s, i + 32, (z + w[1]) & M64, (y + _f64(s, i + 16)) & M64)
        z, x = x, z
        i += 64
        left -= 64
        if left == 0:
            break
    return _hl16((_hl16(v[0], w[0]) + ((_smix(y) * K1) & M64) + z) & M64, (_hl16(v[1], w[1]) + x) & M64)


def encode_fea_grp_id(x, gid, nbits):
    return ((x << nbits) | gid) & M64   # include/difacto/base.h:60-63


def parse_criteo(text, is_train=True):
    """-> (offset, label, index) — rows of <label>\\t<13 ints>\\t<26 categorical>, fields may be empty"""
    off, lab, idx = [0], [], []
    for line in text.replace(b"\r", b"").split(b"\n"):
        if not line:
            continue
        f = line.split(b"\t")
        if is_train:
            lab.append(float(f[0].decode()))
            f = f[1:]
        else:
            lab.append(0.0)
        for i, tok in enumerate(f[:39]):
            if tok:
                idx.append(encode_fea_grp_id(cityhash64(tok), i, 12))
        off.append(len(idx))
    return np.array(off, np.uint64), np.array(lab, np.float32), np.array(idx, np.uint64)


def parse_adfea(text):
    """-> (offset, label, index): AdfeaParser::ParseNext over one chunk (adfea_parser.h:50-84).  Blank-separated tokens;
    `idx:gid` is a feature, id = EncodeFeaGrpID(idx, gid, 12); the plain numbers come in threes — line id, a count, the
    label (1 iff its first character is '1') — and the third one opens a row"""
    off, lab, idx = [0], [], []
    i = 0
    for tok in text.split():
        assert tok[:1].isdigit(), tok
        head, sep, rest = tok.partition(b":")
        if sep:
            assert rest == b"" or rest.isdigit(), tok   # (anything else ends the reference's run in a CHECK, :59)
            gid = int(rest or b"0")
            assert 0 <= gid < 4096
            idx.append(((int(head) << 12) | gid) & (2 ** 64 - 1))
        elif i == 2:
            i = 0
            if lab:
                off.append(len(idx))
            lab.append(1.0 if tok[:1] == b"1" else 0.0)
        else:
            i += 1
    if lab:
        off.append(len(idx))
    return np.array(off, np.uint64), np.array(lab, np.float32), np.array(idx, np.uint64)


# ---- the real liblz4 of this image
_lz4 = None


def liblz4():
    global _lz4
    if _lz4 is None:
        for name in ("liblz4.so.1", ctypes.util.find_library("lz4"), "/usr/lib/x86_64-linux-gnu/liblz4.so.1",
                     "/opt/conda/lib/liblz4.so.1"):
            if not name:
                continue
            try:
                _lz4 = C.CDLL(name)
                break
            except OSError:
                continue
        if _lz4 is None:
            raise OSError("liblz4 is not installed")
        _lz4.LZ4_compressBound.argtypes = [C.c_int]
        _lz4.LZ4_compress_default.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        _lz4.LZ4_decompress_safe.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    return _lz4


def lz4_compress(data):
    L = liblz4()
    data = bytes(data)
    cap = L.LZ4_compressBound(len(data))
    dst = C.create_string_buffer(cap)
    n = L.LZ4_compress_default(data, dst, len(data), cap)
    assert n > 0
    return dst.raw[:n]


CRB_MAGIC = 1196140743
REC_MAGIC = 0xced7230a


def write_crb_record(offset, label, index, value=None, weight=None):
    """CompressedRowBlock::Compress for IndexType = feaid_t (u64), size_t offsets"""
    nrows = len(offset) - 1
    out = [struct.pack("<iii", CRB_MAGIC, 8, nrows)]

    def put(arr):
        if arr is None:
            out.append(struct.pack("<i", 0))
            return
        cp = lz4_compress(np.ascontiguousarray(arr).tobytes())
        out.append(struct.pack("<i", len(cp)))
        out.append(cp)

    if value is not None and np.all(np.asarray(value) == 1):
        value = None   # :36-44: all-ones values are dropped
    put(np.asarray(label, np.float32))
    put(np.asarray(offset, np.uint64))
    put(np.asarray(index, np.uint64))
    put(None if value is None else np.asarray(value, np.float32))
    put(None if weight is None else np.asarray(weight, np.float32))
    return b"".join(out)


def write_recordio(records):
    """RecordIOWriter::WriteRecord for every record -> the bytes of a .rec file"""
    out = []
    magic = struct.pack("<I", REC_MAGIC)
    for rec in records:
        rec = bytes(rec)
        n = len(rec)
        lower = (n >> 2) << 2
        dptr = 0
        for i in range(0, lower, 4):
            if rec[i:i + 4] == magic:   # cut here: the word itself is dropped, the reader restores it
                out.append(magic + struct.pack("<I", ((1 if dptr == 0 else 2) << 29) | (i - dptr)))
                out.append(rec[dptr:i])
                dptr = i + 4
        out.append(magic + struct.pack("<I", ((3 if dptr != 0 else 0) << 29) | (n - dptr)))
        out.append(rec[dptr:])
        out.append(b"\0" * ((4 - (n - dptr) % 4) % 4))
    return b"".join(out)
