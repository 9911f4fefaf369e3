#!/usr/bin/env python3
"""Per-dispatch mean of one PMC counter for the kernels whose name contains a pattern.
usage: rocpd_pmc.py results.db COUNTER kernel_substring [min_value_for_working_launches]"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
ctr, pat = sys.argv[2], sys.argv[3]
vals = np.array([r[0] for r in db.execute("select counter_value from pmc_events where counter_name = ? and name like ?", (ctr, "%" + pat + "%"))])
work = vals[vals > (float(sys.argv[4]) if len(sys.argv) > 4 else 0.0)]
print("%s dispatches %d (doing work %d) mean_all %.3f mean_work %.3f median_work %.3f max %.3f (per %s dispatch)" % (
    ctr, vals.size, work.size, vals.mean(), work.mean(), np.median(work), vals.max(), pat))
