#!/usr/bin/env python
"""Generates tests/golden/ref_ingest.npz with the REFERENCE'S OWN format code (oracle/_ref: CompressedRowBlock::Compress,
src/data/compressed_row_block.h:23-50; CriteoParser::ParseNext, src/reader/criteo_parser.h:40-94) — run in the build
container, where /root/reference exists; the fixture travels, the reference does not.

  crb_rec_<i>      bytes CompressedRowBlock::Compress<feaid_t> wrote for block i
  crb_<i>_{offset,label,index,value,weight}   the block that went in (value / weight absent when NULL)
  criteo_text_<i>  a chunk of criteo text; criteo_<i>_{offset,label,index}: what CriteoParser::ParseNext made of it
                   (criteo_<i>_train = 0: the criteo_test format, no label column)
  adfea_text_<i>   a chunk of adfea text; adfea_<i>_{offset,label,index}: what AdfeaParser::ParseNext
                   (src/reader/adfea_parser.h:33-88) made of it
CityHash64 inside the parser is oracle/city_checker.cc (the library is absent): the ids pin slot tagging, field
splitting and row cutting, not the hash (that one is pinned to Abseil's CityHash64 in tests/test_ingest.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bindings as ob  # noqa: E402


def criteo_row(rng, p_int=0.2, p_cat=0.1, ncat=26, label=True):
    f = [str(int(rng.random() < 0.25))] if label else []
    for _ in range(13):
        f.append("" if rng.random() < p_int else str(int(rng.integers(-3, 100000))))
    for _ in range(ncat):
        f.append("" if rng.random() < p_cat else "%08x" % int(rng.integers(0, 1 << 32)))
    return "\t".join(f)


def criteo_cases(rng):
    """texts the reference accepts (it CHECK-fails on rows whose last token is cut by the end of the chunk, and reads
    past a row's newline when a short row ends in an EMPTY field: those are not formats, they are aborts)"""
    cases = []
    cases.append(("\n".join(criteo_row(rng) for _ in range(16)) + "\n", 1))                     # plain
    cases.append(("\r\n".join(criteo_row(rng) for _ in range(8)) + "\r\n", 1))                  # CRLF
    cases.append(("\n\n\r\n" + "\n\n".join(criteo_row(rng) for _ in range(10)) + "\n\n\n", 1))   # blank lines
    cases.append(("\n".join(criteo_row(rng, p_int=0.9, p_cat=0.9) for _ in range(12)) + "\n", 1))  # mostly missing
    cases.append(("\n".join(criteo_row(rng, p_int=1.0, p_cat=1.0) for _ in range(5)) + "\n", 1))   # nothing but tabs
    # short rows: fewer categorical fields, the last one present (the row ends right after a full 8-character token)
    rows = []
    for n in (1, 5, 13, 25, 26):
        r = criteo_row(rng, p_cat=0.0, ncat=n)
        rows.append(r)
    cases.append(("\n".join(rows) + "\n", 1))
    # the chunk ends without a newline, after the tab that follows the 25th categorical field (the 26th is missing)
    body = "\n".join(criteo_row(rng) for _ in range(7))
    last = criteo_row(rng, p_cat=0.0, ncat=25) + "\t"
    cases.append((body + "\n" + last, 1))
    # labels that are not 0/1, integer fields that are not small integers
    rows = []
    for lab in ("0", "1", "-1", "0.5", "3", "1e0"):
        f = [lab] + [t for t in ("123456789012", "-17", "0", "00", "4.5", "x", "", "7", "8", "9", "10", "11", "12")]
        f += ["%08x" % int(rng.integers(0, 1 << 32)) for _ in range(26)]
        rows.append("\t".join(f))
    cases.append(("\n".join(rows) + "\n", 1))
    cases.append(("\n".join(criteo_row(rng, label=False) for _ in range(8)) + "\n", 0))           # criteo_test
    return cases


def adfea_row(rng, lineid, nfeat=None, sep=" ", label=None):
    n = int(rng.integers(0, 30)) if nfeat is None else nfeat
    feats = ["%d:%d" % (int(rng.integers(0, 2 ** 52)), int(rng.integers(0, 4096))) for _ in range(n)]
    lab = str(int(rng.integers(0, 2))) if label is None else label
    return sep.join([str(lineid), str(n), lab] + feats)


def adfea_cases(rng):
    """chunks of adfea text (src/reader/adfea_parser.h): `lineid count label idx:gid ...`, blanks of any kind between tokens"""
    cases = []
    cases.append("\n".join(adfea_row(rng, 100 + i) for i in range(16)) + "\n")                       # plain
    cases.append("\r\n".join(adfea_row(rng, i) for i in range(8)) + "\r\n")                            # CRLF
    cases.append("  \n\n" + "\n \t\n".join(adfea_row(rng, i, sep="\t ") for i in range(9)) + " \f\n\n")  # blank lines, tabs, form feed
    cases.append("\n".join(adfea_row(rng, i, nfeat=0) for i in range(5)) + "\n")                      # rows without features
    cases.append("\n".join(adfea_row(rng, i, label=l) for i, l in enumerate(["1", "0", "10", "01", "11", "7"])) + "\n")  # label = first char == '1'
    cases.append("7 2 1 18446744073709551615:4095 99999999999999999999999:3\n8 1 0 0:0\n")               # 2^64 - 1, and an idx that wraps
    cases.append("\n".join(adfea_row(rng, i) for i in range(6)))                                      # no final newline
    return cases


def crb_blocks(rng):
    blocks = []
    for b in range(8):
        nrows = int(rng.integers(1, 60))
        lens = rng.integers(0, 24, size=nrows)
        off = np.zeros(nrows + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        nnz = int(off[-1])
        idx = rng.integers(0, 2 ** 63, size=nnz, dtype=np.uint64) * np.uint64(2) + np.uint64(b & 1)
        lab = np.where(rng.random(nrows) < 0.3, 1.0, 0.0).astype(np.float32)
        val = None
        if b % 3 == 1:
            val = rng.normal(size=nnz).astype(np.float32)
        elif b % 3 == 2:
            val = np.ones(nnz, np.float32)   # all ones: Compress drops them (:36-44)
        wgt = rng.random(nrows).astype(np.float32) if b in (4, 5) else None
        blocks.append(dict(offset=off, label=lab, index=idx, value=val, weight=wgt))
    # a block without nonzeros.  (Offsets that do not start at zero are not a format: Compress would write them as they are,
    # and the consumer of the decompressed container, RowBlockContainer::GetBlock, checks offset.back() == index.size().)
    blocks.append(dict(offset=np.zeros(4, np.uint64), label=np.ones(3, np.float32), index=np.zeros(0, np.uint64), value=None,
                       weight=None))
    return blocks


def main():
    R = ob.Ref()
    assert R.has_ingest, "oracle/_ref was built without the data-format half (no lz4.h found)"
    rng = np.random.default_rng(20260926)
    out = {}
    for i, blk in enumerate(crb_blocks(rng)):
        rec = R.crb_compress(blk["offset"], blk["label"], blk["index"], blk["value"], blk["weight"])
        out["crb_rec_%d" % i] = np.frombuffer(rec, np.uint8)
        for k, v in blk.items():
            if v is not None:
                out["crb_%d_%s" % (i, k)] = v
    for i, (text, train) in enumerate(criteo_cases(rng)):
        tb = text.encode()
        off, lab, idx = R.criteo_parse(tb, is_train=bool(train))
        out["criteo_text_%d" % i] = np.frombuffer(tb, np.uint8)
        out["criteo_%d_train" % i] = np.array(train)
        out["criteo_%d_offset" % i], out["criteo_%d_label" % i], out["criteo_%d_index" % i] = off, lab, idx
    rng2 = np.random.default_rng(20260927)   # (its own stream: the arrays above stay what they were before adfea was added)
    for i, text in enumerate(adfea_cases(rng2)):
        tb = text.encode()
        off, lab, idx = R.adfea_parse(tb)
        out["adfea_text_%d" % i] = np.frombuffer(tb, np.uint8)
        out["adfea_%d_offset" % i], out["adfea_%d_label" % i], out["adfea_%d_index" % i] = off, lab, idx
    path = os.path.join(ROOT, "tests", "golden", "ref_ingest.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", sum(1 for k in out if k.startswith("crb_rec_")), "row blocks,",
          sum(1 for k in out if k.startswith("criteo_text_")), "criteo texts,", sum(1 for k in out if k.startswith("adfea_text_")),
          "adfea texts")


if __name__ == "__main__":
    main()
