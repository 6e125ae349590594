#!/usr/bin/env python
"""Interference analysis of a pipelined run from a rocprofv3 (rocpd sqlite) kernel trace: for every
launch of a main-stream kernel, how long each preparation-stream kernel ran beside it, and a
least-squares fit  duration = base + sum_k slowdown_k * overlap_k  (slowdown_k = extra time of the
main kernel per unit of time kernel k runs concurrently).   Usage: rocpd_overlap.py <db> [out]"""
import sqlite3
import sys

import numpy as np

MAIN = ["k_lookup", "k_forward", "k_backward_all"]


def short(n):
    n = n.replace("void ", "").replace("dfh::", "")
    return n.split("<")[0].split("(")[0]


def main(db_path, out=None):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = [(short(r[0]), r[1] / 1e3, r[2] / 1e3) for r in rows]
    rows = [r for r in rows if not r[0].startswith("k_warm_start") and not r[0].startswith("__amd")]
    # steady state: drop the first third
    t_lo = rows[len(rows) // 3][1]
    rows = [r for r in rows if r[1] >= t_lo]
    others = sorted({r[0] for r in rows if r[0] not in MAIN})
    starts = {k: np.array([r[1] for r in rows if r[0] == k]) for k in others + MAIN}
    ends = {k: np.array([r[2] for r in rows if r[0] == k]) for k in others + MAIN}
    lines = ["# %s: per-launch overlap of main-stream kernels with the other kernels (us)" % db_path]
    for m in MAIN:
        inst = [(r[1], r[2]) for r in rows if r[0] == m]
        if not inst:
            continue
        dur = np.array([e - s for s, e in inst])
        cols = [k for k in others + [x for x in MAIN if x != m]]
        X = np.zeros((len(inst), len(cols)))
        for j, k in enumerate(cols):
            for i, (s, e) in enumerate(inst):
                lo = np.maximum(starts[k], s)
                hi = np.minimum(ends[k], e)
                X[i, j] = np.clip(hi - lo, 0, None).sum()
        A = np.concatenate([np.ones((len(inst), 1)), X], 1)
        coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
        alone = dur[X.sum(1) < 0.5]
        lines.append("%s: %d launches, mean %.1f us (min %.1f max %.1f); %d launches with no overlap: mean %.1f; fitted base %.1f"
                     % (m, len(inst), dur.mean(), dur.min(), dur.max(), len(alone), alone.mean() if len(alone) else float("nan"), coef[0]))
        for j, k in enumerate(cols):
            if X[:, j].mean() > 0.05:
                lines.append("    beside %-16s mean overlap %6.1f us   slowdown %+.2f us per us   => %+.1f us per launch"
                             % (k, X[:, j].mean(), coef[1 + j], coef[1 + j] * X[:, j].mean()))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
