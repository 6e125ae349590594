#!/usr/bin/env python
"""Print the kernel timeline of one steady-state step from a rocprofv3 (rocpd sqlite)
kernel trace: every dispatch between two consecutive launches of an anchor kernel, with
start/end relative to the first and the stream it ran on.  Usage:
    rocpd_timeline.py <db> [anchor-substring] [occurrence-from-the-end] [out]"""
import sqlite3
import sys


def main(db_path, anchor="k_forward", back=5, out=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end, stream_id, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < back + 1:
        raise SystemExit("anchor %r seen %d times only" % (anchor, len(idx)))
    a, b = idx[-back - 1], idx[-back]
    t0 = rows[a][1]
    lines = ["# one step of %s (anchor %s, %d-th from the end); times in us relative to the anchor's start" % (db_path, anchor, back),
             "%10s %10s %9s %7s %6s  %s" % ("start", "end", "dur", "stream", "queue", "kernel")]
    for r in rows[a:b + 1]:
        name = r[0] if len(r[0]) <= 100 else r[0][:97] + "..."
        lines.append("%10.1f %10.1f %9.1f %7s %6s  %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4], name))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "k_forward", int(a[3]) if len(a) > 3 else 5, a[4] if len(a) > 4 else None)
