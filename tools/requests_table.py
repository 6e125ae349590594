#!/usr/bin/env python
"""Memory-side requests per kernel and per step from two rocprofv3 --pmc passes (TCC_EA0_RDREQ_sum, TCC_EA0_WRREQ_sum; rocpd
sqlite) of the same bench.py command.  A "step" = one k_forward launch: every kernel's total over the run is divided by the
number of forward launches (the model fill — k_warm_start, memsets — is left out).
usage: requests_table.py <rdreq.db> <wrreq.db> <out.json> <out.txt> [label]"""
import json
import sqlite3
import sys


def short(name):
    return name.replace("void ", "").replace("dfh::", "").split("(")[0]


def totals(db_path, counter):
    db = sqlite3.connect(db_path)
    return {short(k): (n, s) for k, n, s in db.execute(
        "select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,))}


def main(rd_db, wr_db, out_json, out_txt, label=""):
    rd, wr = totals(rd_db, "TCC_EA0_RDREQ_sum"), totals(wr_db, "TCC_EA0_WRREQ_sum")
    skip = ("k_warm_start", "__amd_rocclr_fillBuffer", "k_ss_", "k_loc_splitters")
    steps = max(v[0] for k, v in rd.items() if k.startswith("k_forward"))
    rows, tot_r, tot_w = [], 0.0, 0.0
    for k in sorted(set(rd) | set(wr)):
        if k.startswith(skip):
            continue
        r, w = rd.get(k, (0, 0.0)), wr.get(k, (0, 0.0))
        n = r[0] or w[0]
        per_r, per_w = r[1] / steps, w[1] / steps
        tot_r += per_r
        tot_w += per_w
        rows.append((k, n, n / steps, r[1] / max(r[0], 1), w[1] / max(w[0], 1), per_r, per_w))
    out = dict(label=label, steps=steps, read_requests_per_step=tot_r, write_requests_per_step=tot_w, requests_per_step=tot_r + tot_w,
               kernels={k: dict(launches_per_step=lps, read_requests_per_launch=a, write_requests_per_launch=b) for k, n, lps, a, b, _, _ in rows})
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    lines = ["# %s: memory-side requests (TCC_EA0_RDREQ_sum: reads of 32 / 64 / 128 B; TCC_EA0_WRREQ_sum: writes of 32 / 64 B), %d steps" % (label, steps),
             "%-40s %9s %14s %14s %14s %14s" % ("kernel", "per step", "reads/launch", "writes/launch", "reads/step", "writes/step")]
    for k, n, lps, a, b, pr, pw in sorted(rows, key=lambda x: -(x[5] + x[6])):
        lines.append("%-40s %9.2f %14.0f %14.0f %14.0f %14.0f" % (k[:40], lps, a, b, pr, pw))
    lines.append("%-40s %9s %14s %14s %14.0f %14.0f   = %.3f M requests per step" % ("TOTAL", "", "", "", tot_r, tot_w, (tot_r + tot_w) / 1e6))
    text = "\n".join(lines) + "\n"
    open(out_txt, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
