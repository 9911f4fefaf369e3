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
