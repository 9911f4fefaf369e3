#!/usr/bin/env python3
"""Per-dispatch statistics of PMC counters for the kernels whose name contains a pattern, from a rocprofv3 rocpd
database; optionally restricted to the LAST n dispatches of that kernel (the bench's timed iterations).
usage: rocpd_pmc.py results.db kernel_substring [n_last] [COUNTER ...]      (no counter named: all that were collected)
Prints one line per counter and, last, one JSON object {counter: {mean_all, mean_work, median_work, sum, n, n_work}}.
A dispatch "does work" when its counter value exceeds 1/50 of the median of the upper half (no-op launches that find
the sweep finished read and execute next to nothing)."""
import json
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
n_last = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 0
want = [a for a in sys.argv[3:] if not a.isdigit()]
names = [r[0] for r in db.execute("select distinct counter_name from pmc_events")]
out = {}
for ctr in (want or names):
    rows = list(db.execute("select dispatch_id, sum(counter_value) from pmc_events where counter_name = ? and name like ? group by dispatch_id order by dispatch_id",
                           (ctr, "%" + pat + "%")))
    vals = np.array([r[1] for r in rows], dtype=np.float64)
    if n_last:
        vals = vals[-n_last:]
    if not vals.size:
        continue
    thr = np.median(vals[vals >= np.median(vals)]) / 50.0
    work = vals[vals > thr]
    out[ctr] = {"n": int(vals.size), "n_work": int(work.size), "mean_all": float(vals.mean()), "mean_work": float(work.mean()) if work.size else 0.0,
                "median_work": float(np.median(work)) if work.size else 0.0, "sum": float(vals.sum())}
    print("%-22s dispatches %d (doing work %d) mean_all %.3f mean_work %.3f median_work %.3f max %.3f" % (
        ctr, vals.size, work.size, vals.mean(), out[ctr]["mean_work"], out[ctr]["median_work"], vals.max()))
print(json.dumps(out))
