#!/usr/bin/env python
"""Anchor-to-anchor intervals of a rocprofv3 (rocpd sqlite) kernel trace: how regular the steps are.  Prints percentiles of
the time between consecutive launches of the anchor kernel, the share of the wall time spent in intervals longer than
1.5x the median, and the position of the long ones.  Usage: rocpd_intervals.py <db> [anchor-substring] [out]"""
import sqlite3
import sys

import numpy as np


def main(db_path, anchor="k_forward", out=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    st = np.array([r[1] for r in rows if anchor in r[0]], np.float64) / 1e3
    d = np.diff(st)
    med = float(np.median(d))
    longs = np.nonzero(d > 1.5 * med)[0]
    lines = ["# %s: %d launches of %s; interval between consecutive launches, us" % (db_path, len(st), anchor),
             "min %.1f  p10 %.1f  median %.1f  p90 %.1f  p99 %.1f  max %.1f  mean %.1f" % (
                 d.min(), np.percentile(d, 10), med, np.percentile(d, 90), np.percentile(d, 99), d.max(), d.mean()),
             "intervals > 1.5 x median: %d of %d, holding %.1f %% of the span (%.3f of %.3f s)" % (
                 len(longs), len(d), 100 * d[longs].sum() / d.sum(), d[longs].sum() / 1e6, d.sum() / 1e6),
             "positions of the first 40 long intervals (index: us): " + " ".join("%d:%.0f" % (i, d[i]) for i in longs[:40])]
    # what ran inside the longest interval
    if len(longs):
        i = int(longs[np.argmax(d[longs])])
        a, b = st[i] * 1e3, st[i + 1] * 1e3
        lines.append("# the longest interval (%.0f us), every dispatch inside it:" % d[i])
        for r in rows:
            if a <= r[1] < b:
                lines.append("%10.1f %10.1f %9.1f %5s  %s" % ((r[1] - a) / 1e3, (r[2] - a) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:90]))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "k_forward", a[3] if len(a) > 3 else None)
