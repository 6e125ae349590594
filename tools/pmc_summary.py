#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd sqlite) per kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM section; re-calibrated here with
tools/gather_bench.hip on known byte counts, see profiles/r01_pmc_calibration.txt):
  FETCH_SIZE counts 64 B per 128 B request for 16 B/lane reads -> x2
  WRITE_SIZE is 1:1
Units of the raw counters are KiB.
usage: pmc_summary.py <fetch.db> <write.db> <out.json> [<out.txt>]
"""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                      "group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def short(name):
    n = name.replace("void ", "").replace("dfh::", "")
    return n.split("(")[0]


def main(fetch_db, write_db, out_json, out_txt=None):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    out = {}
    lines = ["%-40s %8s %16s %16s %16s" % ("kernel", "launches", "FETCH_SIZE KiB", "read bytes (x2)", "write bytes")]
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] * 2 + w.get(k, (0, 0))[1])):
        fk, wk = f.get(k, (0, 0.0)), w.get(k, (0, 0.0))
        rd, wr = fk[1] * 1024 * 2, wk[1] * 1024
        out[short(k)] = dict(launches=int(fk[0] or wk[0]), fetch_size_kib_raw=fk[1], write_size_kib_raw=wk[1],
                             read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
        lines.append("%-40s %8d %16.1f %16.0f %16.0f" % (short(k)[:40], int(fk[0] or wk[0]), fk[1], rd, wr))
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    text = "\n".join(lines) + "\n"
    if out_txt:
        open(out_txt, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
