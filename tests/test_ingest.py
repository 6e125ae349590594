"""CPU tests of the data formats either side of the path, host code only:
difacto_amd/host/{cityhash,lz4_block,batch_reader}.h through build/libdifacto_ingest.so (ctypes) against
oracle/ingest.py — criteo text (CityHash64 + slot tag, src/reader/criteo_parser.h:40-94), RecordIO files of
LZ4 compressed row blocks (src/data/compressed_row_block.h, src/reader/crb_parser.h) and libsvm, each
also read in several parts (the data split of SGDLearner::RunEpoch)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ing():
    from difacto_amd import build
    build.build_host()
    L = C.CDLL(os.path.join(ROOT, "build", "libdifacto_ingest.so"))
    L.ingest_cityhash64.restype = C.c_uint64
    L.ingest_cityhash64.argtypes = [C.c_char_p, C.c_size_t]
    L.ingest_lz4_decompress.restype = C.c_long
    L.ingest_lz4_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.ingest_reset_shuffle_stream.restype = None
    L.ingest_read.restype = C.c_long
    L.ingest_read.argtypes = [C.c_char_p, C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_size_t, C.c_size_t,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_long)]
    return L


def read_all(L, path, fmt, part=0, nparts=1, batch=64, shuffle=0, neg=1.0, cap_rows=200000, cap_nnz=4000000):
    off = np.zeros(cap_rows + 1, np.uint64)
    lab = np.zeros(cap_rows, np.float32)
    idx = np.zeros(cap_nnz, np.uint64)
    val = np.zeros(cap_nnz, np.float32)
    hv, nb = C.c_int(0), C.c_long(0)
    n = L.ingest_read(str(path).encode(), fmt.encode(), part, nparts, batch, shuffle, neg, cap_rows, cap_nnz,
                      off.ctypes.data, lab.ctypes.data, idx.ctypes.data, val.ctypes.data, C.byref(hv), C.byref(nb))
    assert n >= 0
    nnz = int(off[n])
    return dict(offset=off[:n + 1], label=lab[:n], index=idx[:nnz], value=val[:nnz], has_value=bool(hv.value), nbatches=nb.value)


def test_cityhash64_known_answer_and_transcription(ing):
    from oracle import ingest as oi
    assert ing.ingest_cityhash64(b"", 0) == 0x9ae16a3b2f90404f == oi.cityhash64(b"")   # the published constant k2
    rng = np.random.default_rng(3)
    for n in list(range(0, 70)) + [127, 128, 129, 191, 192, 193, 255, 256, 1000]:
        for _ in range(3):
            s = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
            assert ing.ingest_cityhash64(s, n) == oi.cityhash64(s), n
    # what the criteo parser feeds it: decimal integers and 8 hex characters
    for tok in (b"0", b"1", b"-1", b"4", b"1382", b"123456789", b"68fd1e64", b"80e26c9b", b"fb936136", b"ffffffff"):
        assert ing.ingest_cityhash64(tok, len(tok)) == oi.cityhash64(tok)


def _abseil_cityhash64():
    """Google's own CityHash64 as compiled into this image: Abseil's hash_internal::CityHash64 (absl/hash/internal/city.cc,
    the CityHash v1.1 code) is exported by pyarrow's libarrow_compute.so.  None when it cannot be found."""
    import re
    import subprocess
    try:
        import pyarrow
        import pyarrow.compute  # noqa: F401  (loads libarrow.so, which libarrow_compute.so needs)
        d = os.path.dirname(pyarrow.__file__)
        for name in sorted(os.listdir(d)):
            if not name.startswith("libarrow_compute.so"):
                continue
            path = os.path.join(d, name)
            syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, timeout=60).stdout
            for line in syms.splitlines():
                if re.search(r"hash_internal10CityHash64EPKcm$", line):
                    f = getattr(C.CDLL(path), line.split()[-1])
                    f.restype = C.c_uint64
                    f.argtypes = [C.c_char_p, C.c_size_t]
                    return f
    except Exception:  # noqa: BLE001
        return None
    return None


def test_cityhash64_against_abseil(ing):
    """pins CityHash64 beyond the one published constant: the product's implementation (difacto_amd/host/cityhash.h) and
    the checker's transcription (oracle/ingest.py) against Google's own code of the same algorithm — Abseil's
    hash_internal::CityHash64, CityHash v1.1 (what cityhash 1.1.1, the reference's dependency, computes) — on every
    length class: 0-16, 17-32, 33-64 and the 64-byte loop, plus the tokens a criteo row is made of"""
    from oracle import ingest as oi
    ref = _abseil_cityhash64()
    if ref is None:
        pytest.skip("no Abseil CityHash64 symbol in this image (pyarrow's libarrow_compute.so)")
    assert ref(b"", 0) == 0x9ae16a3b2f90404f
    rng = np.random.default_rng(17)
    n = 0
    for ln in list(range(0, 260)) + [511, 512, 513, 1000, 4097]:
        for _ in range(4):
            s = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
            want = ref(s, ln)
            assert ing.ingest_cityhash64(s, ln) == want, ln
            if ln <= 80 or ln % 64 < 2:
                assert oi.cityhash64(s) == want, ln
            n += 1
    for tok in [b"%d" % v for v in list(range(0, 1200, 7)) + [-1, -2, 10 ** 6, 2 ** 31, 10 ** 12]] + \
               [b"%08x" % v for v in rng.integers(0, 1 << 32, size=300)]:
        assert ing.ingest_cityhash64(tok, len(tok)) == ref(tok, len(tok)) == oi.cityhash64(tok)
    assert n > 1000


def test_lz4_block_decoder_against_liblz4(ing):
    from oracle import ingest as oi
    rng = np.random.default_rng(5)
    cases = [b"", b"a", b"abcd" * 3, b"\0" * 100000, bytes(range(256)) * 40, rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
             rng.integers(0, 4, 300000, dtype=np.uint8).tobytes(), np.arange(50000, dtype=np.uint64).tobytes(),
             (b"x" * 17 + b"yz") * 5000, np.repeat(rng.normal(size=3000).astype(np.float32), 7).tobytes()]
    for raw in cases:
        if not raw:
            continue
        cp = oi.lz4_compress(raw)
        dst = C.create_string_buffer(max(len(raw), 1))
        assert ing.ingest_lz4_decompress(cp, len(cp), dst, len(raw)) == len(raw)
        assert dst.raw[:len(raw)] == raw
        # a destination that is too small, and truncated / corrupted input: an error, never an overrun
        if len(raw) > 8:
            assert ing.ingest_lz4_decompress(cp, len(cp), dst, len(raw) - 1) == -1
            assert ing.ingest_lz4_decompress(cp, len(cp) - 1, dst, len(raw)) in (-1, len(raw) - 1, len(raw))
        bad = bytearray(cp)
        if len(bad) > 4:
            bad[len(bad) // 2] ^= 0xFF
            r = ing.ingest_lz4_decompress(bytes(bad), len(bad), dst, len(raw))
            assert -1 <= r <= len(raw)


def _criteo_text(rng, nrows):
    lines = []
    for _ in range(nrows):
        f = [str(int(rng.random() < 0.25))]
        for _ in range(13):
            f.append("" if rng.random() < 0.2 else str(int(rng.integers(-2, 5000))))
        for _ in range(26):
            f.append("" if rng.random() < 0.1 else "%08x" % int(rng.integers(0, 1 << 32)))
        lines.append("\t".join(f))
    return ("\n".join(lines) + "\n").encode()


def test_criteo_text_reader(ing, tmp_path):
    from oracle import ingest as oi
    rng = np.random.default_rng(7)
    text = _criteo_text(rng, 700)
    path = tmp_path / "criteo.txt"
    path.write_bytes(text)
    off, lab, idx = oi.parse_criteo(text)
    got = read_all(ing, path, "criteo", batch=100)
    assert got["nbatches"] == 7 and not got["has_value"]
    assert np.array_equal(got["offset"], off) and np.array_equal(got["label"], lab) and np.array_equal(got["index"], idx)
    assert np.all((idx & np.uint64(0xFFF)) < 39)   # the slot tag in the low 12 bits (EncodeFeaGrpID, base.h:60-63)
    # three parts: every row exactly once
    parts = [read_all(ing, path, "criteo", part=p, nparts=3, batch=64) for p in range(3)]
    assert sum(len(p["label"]) for p in parts) == 700
    assert np.array_equal(np.concatenate([p["index"] for p in parts]), idx)
    # criteo_test: no label column
    text2 = b"\n".join(l.split(b"\t", 1)[1] for l in text.split(b"\n") if l) + b"\n"
    p2 = tmp_path / "criteo_test.txt"
    p2.write_bytes(text2)
    got2 = read_all(ing, p2, "criteo_test", batch=100)
    assert np.array_equal(got2["index"], idx) and not got2["label"].any()


def test_rec_reader_recordio_of_compressed_row_blocks(ing, tmp_path):
    from oracle import ingest as oi
    rng = np.random.default_rng(9)
    blocks, recs = [], []
    for b in range(9):
        nrows = int(rng.integers(1, 400))
        lens = rng.integers(0, 40, size=nrows)
        off = np.zeros(nrows + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        nnz = int(off[-1])
        idx = rng.integers(0, 2 ** 63, size=nnz, dtype=np.uint64)
        val = None if b % 2 == 0 else rng.normal(size=nnz).astype(np.float32)
        lab = np.where(rng.random(nrows) < 0.3, 1.0, 0.0).astype(np.float32)
        if b == 3 and nnz >= 4:   # a payload word equal to the RecordIO magic: the writer must cut the record there
            idx[:] = np.uint64(oi.REC_MAGIC) | (np.uint64(oi.REC_MAGIC) << np.uint64(32))
        blocks.append((off, lab, idx, val))
        recs.append(oi.write_crb_record(off, lab, idx, val))
    # a record that certainly contains the magic word at an aligned position (uncompressible by construction is not
    # needed: RecordIO frames the COMPRESSED bytes): append the word to one record's tail region via a raw record test
    data = oi.write_recordio(recs)
    path = tmp_path / "data.rec"
    path.write_bytes(data)
    got = read_all(ing, path, "rec", batch=50)
    want_idx = np.concatenate([b[2] for b in blocks])
    want_lab = np.concatenate([b[1] for b in blocks])
    assert np.array_equal(got["label"], want_lab) and np.array_equal(got["index"], want_idx)
    want_val = np.concatenate([np.ones(len(b[2]), np.float32) if b[3] is None else b[3] for b in blocks])
    assert got["has_value"] and np.array_equal(got["value"], want_val)
    parts = [read_all(ing, path, "rec", part=p, nparts=4, batch=33) for p in range(4)]
    assert sum(len(p["label"]) for p in parts) == len(want_lab)
    assert np.array_equal(np.concatenate([p["index"] for p in parts]), want_idx)


def test_recordio_records_cut_at_the_magic_word(ing, tmp_path):
    """a compressed row block whose BYTES contain 0xced7230a at an aligned offset: written as a multi-part
    record (cflag 1 .. 3), read back whole"""
    from oracle import ingest as oi
    magic = struct.pack("<I", oi.REC_MAGIC)
    nrows = 6
    off = np.arange(nrows + 1, dtype=np.uint64) * np.uint64(2)
    idx = np.arange(2 * nrows, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    lab = np.ones(nrows, np.float32)
    rec = oi.write_crb_record(off, lab, idx)
    # the 12-byte header of a compressed row block is {magic, 8, nrows}; find an aligned spot inside an LZ4 literal run
    framed = oi.write_recordio([rec, rec])
    assert framed.count(magic) == 2
    # now a record with the word inside: labels whose float bits are the magic, left uncompressed by LZ4 (too short to match)
    lab2 = np.frombuffer(magic * nrows, np.float32).copy()
    rec2 = oi.write_crb_record(off, lab2, idx)
    framed2 = oi.write_recordio([rec, rec2, rec])
    if rec2.find(magic) % 4 == 0:
        assert framed2.count(magic) > 3   # the writer had to cut
    path = tmp_path / "cut.rec"
    path.write_bytes(framed2)
    got = read_all(ing, path, "rec", batch=4)
    assert len(got["label"]) == 3 * nrows
    assert got["label"][nrows:2 * nrows].tobytes() == lab2.tobytes()
    assert np.array_equal(got["index"], np.concatenate([idx, idx, idx]))


def test_libsvm_reader_parts_and_sampling(ing, tmp_path):
    rng = np.random.default_rng(11)
    lines, rows = [], []
    for i in range(1000):
        n = int(rng.integers(0, 12))
        ids = rng.integers(1, 10 ** 12, size=n)
        vals = np.round(rng.normal(size=n), 3)
        lab = 1 if rng.random() < 0.3 else -1
        rows.append((lab, ids, vals))
        lines.append(" ".join([str(lab)] + ["%d:%g" % (a, b) for a, b in zip(ids, vals)]))
    path = tmp_path / "d.libsvm"
    path.write_text("\n".join(lines) + "\n")
    got = read_all(ing, path, "libsvm", batch=128)
    assert len(got["label"]) == 1000 and got["nbatches"] == 8
    assert np.array_equal(got["index"], np.concatenate([r[1] for r in rows]).astype(np.uint64))
    assert np.allclose(got["value"], np.concatenate([r[2] for r in rows]).astype(np.float32))
    parts = [read_all(ing, path, "libsvm", part=p, nparts=5, batch=77) for p in range(5)]
    assert np.array_equal(np.concatenate([p["index"] for p in parts]), got["index"])
    # negative down-sampling keeps every positive and about half of the negatives
    s = read_all(ing, path, "libsvm", batch=100, neg=0.5)
    npos = int((got["label"] > 0).sum())
    assert int((s["label"] > 0).sum()) == npos and 0.3 * (1000 - npos) < (s["label"] <= 0).sum() < 0.7 * (1000 - npos)
    # the shuffle buffer permutes rows inside the buffer: the same multiset of rows
    sh = read_all(ing, path, "libsvm", batch=50, shuffle=200)
    assert len(sh["label"]) == 1000 and sorted(sh["index"].tolist()) == sorted(got["index"].tolist())
    assert not np.array_equal(sh["index"], got["index"])


@pytest.mark.parametrize("depth", [1, 2, 5])
def test_reader_threads_hand_out_the_same_minibatches(ing, tmp_path, monkeypatch, depth):
    """the worker loops read through PrefetchSource (minibatches cut `depth` ahead on a reader thread; the shuffle buffer
    itself is assembled one ahead on another): the same minibatches in the same order as the plain BatchReader, for every
    way a minibatch comes about — a view of a parsed chunk, rows appended across chunks, the shuffle buffer's permutation
    (the process-wide RefRand stream restarted for each read), negative down-sampling — and for batch sizes that do and
    do not divide the chunks"""
    from oracle import ingest as oi
    rng = np.random.default_rng(17 + depth)
    text = _criteo_text(rng, 2500)
    txt = tmp_path / "t.criteo"
    txt.write_bytes(text)
    off, lab, idx = oi.parse_criteo(text)
    recs = [oi.write_crb_record(off[a:a + 251] - off[a], lab[a:a + 250], idx[int(off[a]):int(off[a + 250])]) for a in range(0, 2500, 250)]
    rec = tmp_path / "t.rec"
    rec.write_bytes(oi.write_recordio(recs))
    monkeypatch.setenv("DIFACTO_CHUNK_BYTES", "20000")   # ~60 rows of criteo text per chunk
    cases = [dict(batch=64), dict(batch=50, shuffle=500), dict(batch=97, shuffle=970, neg=0.6), dict(batch=250), dict(batch=1000, shuffle=1000),
             dict(batch=7, neg=0.5)]
    for fmt, path in (("criteo", txt), ("rec", rec)):
        for kw in cases:
            monkeypatch.delenv("DIFACTO_INGEST_PREFETCH", raising=False)
            ing.ingest_reset_shuffle_stream()
            want = read_all(ing, path, fmt, **kw)
            monkeypatch.setenv("DIFACTO_INGEST_PREFETCH", str(depth))
            ing.ingest_reset_shuffle_stream()
            got = read_all(ing, path, fmt, **kw)
            assert got["nbatches"] == want["nbatches"] and got["has_value"] == want["has_value"], (fmt, kw)
            for k in ("offset", "label", "index", "value"):
                assert np.array_equal(got[k], want[k]), (fmt, kw, k)
        # and the shuffle really is a different order, the same rows
        ing.ingest_reset_shuffle_stream()
        plain = read_all(ing, path, fmt, batch=50)
        ing.ingest_reset_shuffle_stream()
        shuf = read_all(ing, path, fmt, batch=50, shuffle=500)
        assert not np.array_equal(plain["index"], shuf["index"]) and sorted(plain["index"].tolist()) == sorted(shuf["index"].tolist())


@pytest.mark.parametrize("mode", ["1", "2"])   # 1: assembled buffers; 2: buffers as slices of the parsed chunks (round 4)
@pytest.mark.parametrize("depth", [0, 2])
def test_described_minibatches_equal_the_copied_ones(ing, tmp_path, monkeypatch, depth, mode):
    """BatchReader::Describe (what the device feed reads): the shuffle buffers are announced once each and a minibatch is
    its offsets, labels and (buffer, rows) list; gathering those rows from copies of the buffers must give the copying
    reader's minibatches exactly — same permutation, same sampling draws, same boundaries, minibatches that straddle two
    buffers, a last short buffer — directly and through the reader thread"""
    from oracle import ingest as oi
    rng = np.random.default_rng(29 + depth)
    text = _criteo_text(rng, 2300)
    txt = tmp_path / "t.criteo"
    txt.write_bytes(text)
    lines = []
    for i in range(1700):   # libsvm with real values, some all-ones rows
        n = int(rng.integers(0, 9))
        ids = rng.integers(1, 10 ** 9, size=n)
        vals = np.ones(n) if i % 3 == 0 else np.round(rng.normal(size=n), 2)
        lines.append(" ".join(["1" if rng.random() < 0.4 else "-1"] + ["%d:%g" % (a, b) for a, b in zip(ids, vals)]))
    svm = tmp_path / "t.libsvm"
    svm.write_text("\n".join(lines) + "\n")
    monkeypatch.setenv("DIFACTO_CHUNK_BYTES", "15000")
    cases = [dict(batch=50, shuffle=500), dict(batch=97, shuffle=485, neg=0.6), dict(batch=64, shuffle=64), dict(batch=300, shuffle=700),
             dict(batch=1000, shuffle=1000, neg=0.3)]
    for fmt, path in (("criteo", txt), ("libsvm", svm)):
        for kw in cases:
            monkeypatch.delenv("DIFACTO_INGEST_DESCRIBE", raising=False)
            monkeypatch.delenv("DIFACTO_INGEST_PREFETCH", raising=False)
            ing.ingest_reset_shuffle_stream()
            want = read_all(ing, path, fmt, **kw)
            monkeypatch.setenv("DIFACTO_INGEST_DESCRIBE", mode)
            if depth:
                monkeypatch.setenv("DIFACTO_INGEST_PREFETCH", str(depth))
            ing.ingest_reset_shuffle_stream()
            got = read_all(ing, path, fmt, **kw)
            assert got["nbatches"] == want["nbatches"], (fmt, kw)
            for k in ("offset", "label", "index", "value"):
                assert np.array_equal(got[k], want[k]), (fmt, kw, k)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("depth", [0, 2])
def test_described_minibatches_spanning_many_buffers(ing, tmp_path, monkeypatch, depth, mode):
    """with strong down-sampling (the reference DROPS a negative with probability neg_sampling,
    batch_reader.cc:57-63, so values near 1 on data with few positives) ONE minibatch draws its rows
    from dozens of shuffle buffers, and the reader runs further ahead in buffers than any fixed ring holds.  The
    description's consumer keeps every announced buffer until the last minibatch that names it has been gathered
    (ingest_capi.cc here, DeviceFeed in the worker loop): still exactly the copying reader's minibatches"""
    rng = np.random.default_rng(41 + depth)
    lines = []
    for i in range(6000):   # 3 % positives
        n = int(rng.integers(1, 7))
        ids = rng.integers(1, 10 ** 9, size=n)
        lines.append(" ".join(["1" if rng.random() < 0.03 else "-1"] + ["%d:%g" % (a, b) for a, b in zip(ids, np.round(rng.normal(size=n), 2))]))
    svm = tmp_path / "sparse_pos.libsvm"
    svm.write_text("\n".join(lines) + "\n")
    monkeypatch.setenv("DIFACTO_CHUNK_BYTES", "9000")
    for kw in (dict(batch=100, shuffle=100, neg=0.98), dict(batch=150, shuffle=300, neg=0.95), dict(batch=64, shuffle=64, neg=0.99)):
        monkeypatch.delenv("DIFACTO_INGEST_DESCRIBE", raising=False)
        monkeypatch.delenv("DIFACTO_INGEST_PREFETCH", raising=False)
        ing.ingest_reset_shuffle_stream()
        want = read_all(ing, svm, "libsvm", **kw)
        kept = len(want["label"])
        assert kept < 0.12 * 6000 and want["nbatches"] >= 2   # most negatives dropped: a minibatch spans >= 8 buffers on average
        monkeypatch.setenv("DIFACTO_INGEST_DESCRIBE", mode)
        if depth:
            monkeypatch.setenv("DIFACTO_INGEST_PREFETCH", str(depth))
        ing.ingest_reset_shuffle_stream()
        got = read_all(ing, svm, "libsvm", **kw)
        assert got["nbatches"] == want["nbatches"], kw
        for k in ("offset", "label", "index", "value"):
            assert np.array_equal(got[k], want[k]), (kw, k)


def test_reader_threads_under_tsan(tmp_path):
    """the reader's threads (parser pool, shuffle-buffer thread, minibatch thread) under ThreadSanitizer: no report"""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "tsan_reader")
    cc = ["g++", "-fsanitize=thread", "-O1", "-g", "-std=c++14", "-fopenmp", "-Wno-unknown-pragmas", "-Wno-sign-compare",
          "-DDMLC_LOG_FATAL_THROW=0", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "third_party_shim"),
          "-I" + os.path.join(ROOT, "difacto_amd", "host"), "-o", exe, os.path.join(ROOT, "tools", "tsan_reader.cc"), "-lpthread"]
    r = subprocess.run(cc, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "tsan" in (r.stderr or "").lower():
        pytest.skip("ThreadSanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, DIFACTO_PARSER_THREADS="3", DIFACTO_CHUNK_BYTES="4096", TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "rcv1_100.libsvm")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.startswith("rows ")


def test_readers_on_corrupted_files_under_asan(tmp_path):
    """the four file formats under AddressSanitizer + UBSan (tools/fuzz_readers.cc): good criteo / .rec / adfea / libsvm
    files with byte flips, truncations, bursts and splices — rows or an error, never a read or write outside a buffer (the
    .rec records are views of the mapped file and the LZ4 decoder copies short sequences with fixed-size moves; ParseFast
    loads 8 bytes per categorical field)"""
    import shutil
    import subprocess
    from oracle import ingest as oi
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "fuzz_readers")
    cc = ["g++", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-O1", "-g", "-std=c++14", "-fopenmp",
          "-Wno-unknown-pragmas", "-Wno-sign-compare", "-DDMLC_LOG_FATAL_THROW=1", "-I" + os.path.join(ROOT, "include"),
          "-I" + os.path.join(ROOT, "third_party_shim"), "-I" + os.path.join(ROOT, "difacto_amd", "host"), "-o", exe,
          os.path.join(ROOT, "tools", "fuzz_readers.cc"), "-lpthread"]
    r = subprocess.run(cc, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower():
        pytest.skip("AddressSanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    rng = np.random.default_rng(3)
    text = _criteo_text(rng, 300)
    off, lab, idx = oi.parse_criteo(text)
    files = {"criteo": text}
    files["rec"] = oi.write_recordio([oi.write_crb_record(off[a:a + 101] - off[a], lab[a:a + 100], idx[int(off[a]):int(off[a + 100])])
                                      for a in range(0, 300, 100)])
    rows = [idx[int(off[r]):int(off[r + 1])] for r in range(300)]
    files["adfea"] = "".join(" ".join(["%d" % r, "%d" % len(ids), "%d" % lab[r]] + ["%d:%d" % (int(v) >> 12, int(v) & 0xFFF) for v in ids]) + "\n"
                             for r, ids in enumerate(rows)).encode()
    files["libsvm"] = "".join(" ".join(["%d" % lab[r]] + ["%d:%g" % (int(v) >> 12, 0.5 + (int(v) & 7)) for v in ids]) + "\n"
                              for r, ids in enumerate(rows)).encode()
    for fmt, blob in files.items():
        good = tmp_path / ("good." + fmt)
        good.write_bytes(blob)
        r = subprocess.run([exe, fmt, str(good), "120", str(tmp_path / "scratch.bin")], capture_output=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        out = r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")
        assert r.returncode == 0 and "AddressSanitizer" not in out and "runtime error" not in out, (fmt, out[-2500:])
        assert "120 corrupted inputs" in out


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_parser_pool_keeps_file_order(ing, tmp_path, monkeypatch, threads):
    """the chunks of a part are parsed by a pool of threads (batch_reader.h: Reader) and must reach the
    consumer in file order whatever the number of threads and however small the chunks: libsvm and criteo
    text cut into ~100 chunks (with comment-only and empty chunks in between), a .rec file of 40 records"""
    from oracle import ingest as oi
    rng = np.random.default_rng(threads)
    monkeypatch.setenv("DIFACTO_PARSER_THREADS", str(threads))
    monkeypatch.setenv("DIFACTO_CHUNK_BYTES", "2048")
    lines, ids_all = [], []
    for i in range(3000):
        if i % 500 == 250:
            lines.extend(["# " + "x" * 100] * 60)   # more than one chunk of comments: an empty chunk of rows
        n = int(rng.integers(1, 12))
        ids = rng.integers(1, 10 ** 12, size=n)
        ids_all.append(ids)
        lines.append(" ".join([str(i % 2)] + ["%d:%g" % (a, 0.5) for a in ids]))
    path = tmp_path / "d.libsvm"
    path.write_text("\n".join(lines) + "\n")
    got = read_all(ing, path, "libsvm", batch=97)
    assert len(got["label"]) == 3000
    assert np.array_equal(got["index"], np.concatenate(ids_all).astype(np.uint64))
    assert np.array_equal(got["label"], (np.arange(3000) % 2).astype(np.float32))
    parts = [read_all(ing, path, "libsvm", part=p, nparts=3, batch=50) for p in range(3)]
    assert np.array_equal(np.concatenate([p["index"] for p in parts]), got["index"])
    # criteo text
    rows = 1500
    ints = rng.integers(0, 1000, size=(rows, 13))
    cats = rng.integers(0, 2 ** 32, size=(rows, 26), dtype=np.uint64)
    text = "".join("%d\t%s\t%s\n" % (i % 2, "\t".join(map(str, ints[i])), "\t".join("%08x" % c for c in cats[i]))
                   for i in range(rows)).encode()
    cpath = tmp_path / "d.criteo"
    cpath.write_bytes(text)
    off, lab, idx = oi.parse_criteo(text)
    gc = read_all(ing, cpath, "criteo", batch=128)
    assert np.array_equal(gc["index"], idx) and np.array_equal(gc["offset"], off) and np.array_equal(gc["label"], lab)
    # .rec: 40 records of 50 rows
    recs, want = [], []
    for r in range(40):
        n = 50
        o = np.arange(n + 1, dtype=np.uint64) * 7
        ix = rng.integers(1, 2 ** 60, size=n * 7, dtype=np.uint64)
        recs.append(oi.write_crb_record(o, np.full(n, r, np.float32), ix))
        want.append(ix)
    rpath = tmp_path / "d.rec"
    rpath.write_bytes(oi.write_recordio(recs))
    gr = read_all(ing, rpath, "rec", batch=64)
    assert np.array_equal(gr["index"], np.concatenate(want))
    assert np.array_equal(gr["label"], np.repeat(np.arange(40), 50).astype(np.float32))


# ---------------------------------------------------------------------------------------------------------
# pinned to the REFERENCE'S OWN format code: tests/golden/ref_ingest.npz was written by
# tools/make_golden_ingest.py through oracle/_ref = the reference's CompressedRowBlock::Compress
# (src/data/compressed_row_block.h:23-50) and CriteoParser::ParseNext (src/reader/criteo_parser.h:40-94).
# RecordIO framing (dmlc-core, absent) stays checked against oracle/ingest.py only; CityHash64 is pinned to Abseil's
# copy of Google's code (test_cityhash64_against_abseil), the three transcriptions here agree with each other.
# ---------------------------------------------------------------------------------------------------------
GOLDEN_INGEST = os.path.join(ROOT, "tests", "golden", "ref_ingest.npz")


def _golden_blocks(g):
    i = 0
    while "crb_rec_%d" % i in g:
        blk = {k: (g["crb_%d_%s" % (i, k)] if "crb_%d_%s" % (i, k) in g else None) for k in ("offset", "label", "index", "value", "weight")}
        yield g["crb_rec_%d" % i].tobytes(), blk
        i += 1


def test_rec_reader_on_blocks_compressed_by_the_reference(ing, tmp_path):
    """the product's .rec reader (RecordIO + from-scratch LZ4 + CompressedRowBlock layout) decodes records the
    REFERENCE's CompressedRowBlock::Compress wrote (golden fixture), incl. dropped all-ones values, weights,
    and a block without nonzeros"""
    from oracle import ingest as oi
    g = np.load(GOLDEN_INGEST)
    recs, blocks = zip(*_golden_blocks(g))
    assert len(recs) >= 8
    path = tmp_path / "ref.rec"
    path.write_bytes(oi.write_recordio(recs))
    got = read_all(ing, path, "rec", batch=37)
    want_lab = np.concatenate([b["label"] for b in blocks])
    want_idx = np.concatenate([b["index"] for b in blocks])
    lens = np.concatenate([np.diff(b["offset"].astype(np.int64)) for b in blocks])
    assert np.array_equal(got["label"], want_lab)
    assert np.array_equal(np.diff(got["offset"].astype(np.int64)), lens)
    assert np.array_equal(got["index"], want_idx)
    want_val = np.concatenate([np.ones(len(b["index"]), np.float32) if (b["value"] is None or np.all(b["value"] == 1)) else b["value"]
                               for b in blocks])
    assert got["has_value"] and np.array_equal(got["value"], want_val)
    # the Python writer the other .rec tests use produces the reference's bytes
    for rec, b in zip(recs, blocks):
        assert oi.write_crb_record(b["offset"], b["label"], b["index"], b["value"], b["weight"]) == rec


def test_criteo_parser_on_the_reference_parsers_output(ing, tmp_path):
    """CriteoChunkParser against what the reference's CriteoParser::ParseNext made of the same text (golden fixture):
    missing fields, rows short of categorical features, CRLF, blank lines, a chunk ending without a newline, odd
    labels and integer tokens, the criteo_test format.  (CityHash64 itself: test_cityhash64_against_abseil — the ids here pin the slot tag,
    the field splitting and the row cutting.)"""
    g = np.load(GOLDEN_INGEST)
    i = 0
    while "criteo_text_%d" % i in g:
        text = g["criteo_text_%d" % i].tobytes()
        train = bool(g["criteo_%d_train" % i])
        path = tmp_path / ("c%d.txt" % i)
        path.write_bytes(text)
        got = read_all(ing, path, "criteo" if train else "criteo_test", batch=50)
        assert np.array_equal(got["offset"], g["criteo_%d_offset" % i]), i
        assert np.array_equal(got["label"], g["criteo_%d_label" % i]), i
        assert np.array_equal(got["index"], g["criteo_%d_index" % i]), i
        i += 1
    assert i >= 9


def test_adfea_reader_on_the_reference_parsers_output(ing, tmp_path):
    """data_format = adfea (src/reader/reader.h:40-41): AdfeaChunkParser against what the reference's own
    AdfeaParser::ParseNext made of the same text (golden fixture, written through oracle/_ref) — CRLF, blank lines, tabs and
    form feeds between tokens, rows without features, labels that merely start with '1', 2^64 - 1 and an id that wraps,
    no final newline — and against the Python transcription; then a larger file read in parts"""
    from oracle import ingest as oi
    g = np.load(GOLDEN_INGEST)
    i = 0
    while "adfea_text_%d" % i in g:
        text = g["adfea_text_%d" % i].tobytes()
        path = tmp_path / ("a%d.txt" % i)
        path.write_bytes(text)
        got = read_all(ing, path, "adfea", batch=7)
        want = [g["adfea_%d_%s" % (i, k)] for k in ("offset", "label", "index")]
        assert np.array_equal(got["offset"], want[0]) and np.array_equal(got["label"], want[1]), i
        assert np.array_equal(got["index"], want[2]) and not got["has_value"], i
        py = oi.parse_adfea(text)
        assert all(np.array_equal(a, b) for a, b in zip(py, want)), i
        i += 1
    assert i >= 7
    rng = np.random.default_rng(23)
    lines = []
    for r in range(3000):
        n = int(rng.integers(0, 40))
        lines.append(" ".join(["%d" % r, "%d" % n, "%d" % rng.integers(0, 2)] +
                              ["%d:%d" % (rng.integers(0, 2 ** 50), rng.integers(0, 4096)) for _ in range(n)]))
    text = ("\n".join(lines) + "\n").encode()
    path = tmp_path / "big.adfea"
    path.write_bytes(text)
    off, lab, idx = oi.parse_adfea(text)
    os.environ["DIFACTO_CHUNK_BYTES"] = "20000"   # many chunks: a chunk starts at a line start, i.e. at a line id
    try:
        whole = read_all(ing, path, "adfea", batch=100)
        parts = [read_all(ing, path, "adfea", part=p, nparts=4, batch=64) for p in range(4)]
    finally:
        del os.environ["DIFACTO_CHUNK_BYTES"]
    assert np.array_equal(whole["offset"], off) and np.array_equal(whole["label"], lab) and np.array_equal(whole["index"], idx)
    assert sum(len(p["label"]) for p in parts) == 3000
    assert np.array_equal(np.concatenate([p["index"] for p in parts]), idx)
    assert np.array_equal(np.concatenate([p["label"] for p in parts]), lab)


def test_reference_format_code_live_when_built(ing, tmp_path):
    """in the build container (oracle/_ref with the data-format half): random row blocks through the reference's
    Compress -> product reader, oracle/ingest.py's writer -> the reference's Decompress, random criteo text through
    both parsers, and the checker-side CityHash64 against the product's"""
    from oracle import bindings as ob, ingest as oi
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built here")
    R = ob.Ref()
    if not R.has_ingest:
        pytest.skip("oracle/_ref built without the data-format half (no lz4.h)")
    rng = np.random.default_rng(11)
    recs, blocks = [], []
    for b in range(6):
        nrows = int(rng.integers(1, 200))
        off = np.zeros(nrows + 1, np.uint64)
        off[1:] = np.cumsum(rng.integers(0, 30, size=nrows))
        nnz = int(off[-1])
        idx = rng.integers(0, 2 ** 64 - 1, size=nnz, dtype=np.uint64)
        val = None if b % 2 else rng.normal(size=nnz).astype(np.float32)
        lab = rng.integers(0, 2, size=nrows).astype(np.float32)
        rec = R.crb_compress(off, lab, idx, val)
        assert rec == oi.write_crb_record(off, lab, idx, val)
        d = R.crb_decompress(oi.write_crb_record(off, lab, idx, val))
        assert np.array_equal(d["offset"], off) and np.array_equal(d["label"], lab) and np.array_equal(d["index"], idx)
        assert (d["value"] is None) == (val is None) and (val is None or np.array_equal(d["value"], val))
        recs.append(rec)
        blocks.append((off, lab, idx, val))
    path = tmp_path / "live.rec"
    path.write_bytes(oi.write_recordio(recs))
    got = read_all(ing, path, "rec", batch=64)
    assert np.array_equal(got["index"], np.concatenate([b[2] for b in blocks]))
    assert np.array_equal(got["label"], np.concatenate([b[1] for b in blocks]))
    text = _criteo_text(rng, 300)
    p2 = tmp_path / "live.txt"
    p2.write_bytes(text)
    off, lab, idx = R.criteo_parse(text)
    got = read_all(ing, p2, "criteo", batch=128)
    assert np.array_equal(got["offset"], off) and np.array_equal(got["label"], lab) and np.array_equal(got["index"], idx)
    o2, l2, i2 = oi.parse_criteo(text)
    assert np.array_equal(o2, off) and np.array_equal(i2, idx)
    for n in list(range(0, 80)) + [127, 128, 129, 255, 256, 1000]:
        s = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert R.city_checker_hash64(s) == ing.ingest_cityhash64(s, n) == oi.cityhash64(s), n
    # random adfea text through the reference's AdfeaParser and the product's reader
    lines = []
    for r in range(500):
        n = int(rng.integers(0, 30))
        lines.append("\t".join(["%d" % r, "%d" % n, "%d" % rng.integers(0, 3)] +
                               ["%d:%d" % (rng.integers(0, 2 ** 62), rng.integers(0, 4096)) for _ in range(n)]))
    text = ("\n".join(lines) + "\n").encode()
    p3 = tmp_path / "live.adfea"
    p3.write_bytes(text)
    off, lab, idx = R.adfea_parse(text)
    got = read_all(ing, p3, "adfea", batch=64)
    assert np.array_equal(got["offset"], off) and np.array_equal(got["label"], lab) and np.array_equal(got["index"], idx)


def _parse_criteo_mode(L, text, train, mode):
    L.ingest_parse_criteo.restype = C.c_long
    L.ingest_parse_criteo.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    cap_rows, cap_nnz = text.count(b"\n") + 2, 40 * (text.count(b"\n") + 2)
    off = np.zeros(cap_rows + 1, np.uint64)
    lab = np.zeros(cap_rows, np.float32)
    idx = np.zeros(cap_nnz, np.uint64)
    # the text sits at the very end of its own allocation (no slack behind it): ParseFast's 8-byte loads must stay inside
    buf = C.create_string_buffer(text, len(text)) if text else C.create_string_buffer(1)
    n = L.ingest_parse_criteo(buf, len(text), int(train), mode, cap_rows, cap_nnz, off.ctypes.data, lab.ctypes.data, idx.ctypes.data)
    if n == -2:
        pytest.skip("no AVX2 / BMI / POPCNT on this CPU: Parse() keeps the plain loop")
    assert n >= 0, n
    return off[:n + 1].copy(), lab[:n].copy(), idx[:int(off[n])].copy()


def test_criteo_fast_parser_equals_the_plain_loop(ing):
    """CriteoChunkParser::ParseFast (vector scan for '\\t' / '\\n', straight-line CityHash for 1-7 and 8 bytes) takes the rows
    it recognises as regular and leaves every other row to the reference-shaped ParseRow: same rows, labels and ids as
    ParseSlow on the golden texts of the reference's parser (CRLF, short rows, blank lines, no final newline) and on
    fuzzed text — rows that span scan windows, long integer tokens, junk in integer fields, rows short of fields"""
    from oracle import ingest as oi
    g = np.load(GOLDEN_INGEST)
    i = 0
    while "criteo_text_%d" % i in g:
        text = g["criteo_text_%d" % i].tobytes()
        train = bool(g["criteo_%d_train" % i])
        for mode in (0, 1, 2):
            off, lab, idx = _parse_criteo_mode(ing, text, train, mode)
            assert np.array_equal(off, g["criteo_%d_offset" % i]) and np.array_equal(lab, g["criteo_%d_label" % i]), (i, mode)
            assert np.array_equal(idx, g["criteo_%d_index" % i]), (i, mode)
        i += 1
    rng = np.random.default_rng(41)
    labels = [b"1", b"0", b"0.5", b"-1", b"12", b"1e0"]
    nrows_total = 0
    for it in range(160):
        train = it % 5 != 0
        flavour = it % 8      # 5: CRLF, 6: rows short of categorical fields + junk in integer fields, 7: no final newline
        rows = int(rng.integers(1, 700 if it % 16 == 0 else 60))
        lines = []
        for r in range(rows):
            if rng.integers(17) == 0:
                lines.append(b"")   # blank line
            f = [labels[int(rng.integers(len(labels)))]] if train else []
            weird = flavour == 6 and rng.integers(3) == 0
            for _ in range(13):
                tok = b""
                if rng.integers(4):
                    tok = str(int(rng.integers(10 ** 12)))[:int(rng.integers(1, 13))].encode()
                    if rng.integers(9) == 0:
                        tok = b"-" + tok
                    if weird and rng.integers(7) == 0:
                        tok += b" x"
                f.append(tok)
            ncat = int(rng.integers(1, 26)) if weird and rng.integers(2) else 26   # (>= 1: the 13th integer field ends with a tab)
            f += [b"%08x" % int(rng.integers(1 << 32)) if rng.integers(6) else b"" for _ in range(ncat)]
            lines.append(b"\t".join(f))
        eol = b"\r\n" if flavour == 5 else b"\n"
        text = eol.join(lines) + (b"" if flavour == 7 else eol)
        a = _parse_criteo_mode(ing, text, train, 0)
        b = _parse_criteo_mode(ing, text, train, 1)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (it, flavour)
        nrows_total += len(a[1])
        if flavour < 5 and it % 4 == 0:   # regular text: the Python transcription agrees too
            off, lab, idx = oi.parse_criteo(text, is_train=train)
            assert np.array_equal(a[0], off) and np.array_equal(a[1], lab) and np.array_equal(a[2], idx)
    assert nrows_total > 3000
    # the fast path's own corner cases: (i) CRLF appearing only late in a chunk (regular rows before it went through the vector
    # scan, everything from the first '\r' on goes through ParseRow); (ii) an integer field longer than a scan window (16 KB:
    # the row's line end is not in sight after one scan); (iii) rows whose line end sits in the last 8 bytes of the buffer
    def row(train=True, eol=b"\n", long_int=0):
        f = [b"1"] if train else []
        f += [b"7" * long_int if (i == 3 and long_int) else (b"%d" % (i * 37)) for i in range(13)]
        f += [b"%08x" % (0x9e3779b9 * (i + 1) & 0xffffffff) for i in range(26)]
        return b"\t".join(f) + eol
    texts = [row() * 900 + row(eol=b"\r\n") * 40 + row() * 10,
             row() * 30 + row(long_int=20000) + row() * 30 + row(long_int=40000) + row() * 5,
             row() * 3 + row()[:-9] + b"\n",                      # last row short of one categorical field, '\n' at the very end
             row() * 3 + row(eol=b"")]                              # no final newline: the last row ends with the buffer
    for t_i, text in enumerate(texts):
        a = _parse_criteo_mode(ing, text, True, 0)
        b = _parse_criteo_mode(ing, text, True, 1)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), t_i
        assert len(a[1]) == text.count(b"\n") + (0 if text.endswith(b"\n") else 1)
    # (iv) ADVICE r4: a short row or a run of blank lines whose end falls ON a scan-window boundary (multiples of 16 384
    # bytes from the chunk's start, and from wherever a scan resumed), followed by a row with an empty 13th integer field:
    # the stale delimiters before p once made the next row look regular with its fields in the wrong place (criteo_test:
    # a wild CityHash64 length)
    kwin = 1 << 14

    empty13 = lambda train: b"\t".join(([b"1"] if train else []) + [b"%d" % (i * 5) for i in range(12)] + [b""] +
                                         [b"%08x" % (0x1234567 * (i + 3) & 0xffffffff) for i in range(26)]) + b"\n"
    nwin = 0

    def short_cats(train, gap):
        """a row with all integer fields and only 5 categorical ones, `gap` bytes long (one integer token stretched)"""
        f = ([b"1"] if train else []) + [b"%d" % (i * 3) for i in range(13)] + [b"%08x" % (77 * (i + 1)) for i in range(5)]
        fixed = len(b"\t".join(f)) + 1
        f[4] = f[4] + b"7" * (gap - fixed)
        return b"\t".join(f) + b"\n"
    for train in (True, False):
        for mult in (1, 2, 3):
            for shape in range(3):
                if shape == 0:    # a short row `x...x\n` whose '\n' is the last byte of the window (it swallows the next row's
                    filler = lambda gap: b"x" * (gap - 1) + b"\n"                      # first field, as in the reference)
                elif shape == 1:  # blank lines up to the boundary
                    filler = lambda gap: b"\n" * gap
                else:             # a row short of categorical fields ending on the boundary
                    filler = lambda gap: short_cats(train, gap)
                text = row(train) * 2
                while len(text) + len(row(train)) <= mult * kwin - 200:
                    text += row(train)
                text += filler(mult * kwin - len(text))
                assert len(text) == mult * kwin
                text += empty13(train) + row(train) * 3 + empty13(train) + row(train)
                a = _parse_criteo_mode(ing, text, train, 0)
                b = _parse_criteo_mode(ing, text, train, 1)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (train, mult, shape)
                nwin += 1
    assert nwin == 18
