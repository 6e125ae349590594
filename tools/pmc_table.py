#!/usr/bin/env python
"""Per-kernel averages of whatever counters a set of rocprofv3 --pmc passes (rocpd sqlite) hold.
usage: pmc_table.py <out.txt> <pass1.db> [<pass2.db> ...]"""
import sqlite3
import sys


def short(name):
    return name.replace("void ", "").replace("dfh::", "").split("(")[0]


def main(out, *dbs):
    table, counters = {}, []
    for path in dbs:
        db = sqlite3.connect(path)
        for kernel, counter, n, avg in db.execute(
                "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            table.setdefault(short(kernel), {})[counter] = avg
            if counter not in counters:
                counters.append(counter)
    keep = [k for k in table if k.startswith(("k_", "dfh"))]
    lines = ["# per-launch averages; one --pmc pass per counter group (see tools/collect_profiles.sh)",
             "%-34s " % "kernel" + " ".join("%22s" % c[:22] for c in counters)]
    for k in sorted(keep):
        lines.append("%-34s " % k[:34] + " ".join("%22.1f" % table[k].get(c, float("nan")) for c in counters))
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:])
