#!/usr/bin/env python3
"""Kernel summary (the columns of rocprofv3 --stats) from a rocprofv3 rocpd database.
usage: rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(100 * r[2] / tot, 4), r[4], r[5]])
for r in rows[:6]:
    print("%-70s %8d calls  avg %10.1f ns" % (r[0][:70], r[1], r[3]))


def timed_region(db_path, kernel_substr, n_last):
    """Average duration and period of the LAST n dispatches of a kernel (the bench's timed iterations)."""
    d = sqlite3.connect(db_path)
    rows = list(d.execute("select start, end from kernels where name like ? order by start", ("%" + kernel_substr + "%",)))[-n_last:]
    dur = [e - s for s, e in rows]
    return sum(dur) / len(dur), (rows[-1][1] - rows[0][0]) / len(rows), len(rows)


if len(sys.argv) > 4:  # rocpd_stats.py results.db out.csv kernel_substring n_last
    avg, period, n = timed_region(sys.argv[1], sys.argv[3], int(sys.argv[4]))
    print("last %d dispatches of %s: average duration %.1f ns, average period %.1f ns" % (n, sys.argv[3], avg, period))
