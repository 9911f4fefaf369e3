#!/usr/bin/env python3
"""One line per bench JSON: the figures the option scans compare.  usage: bsum.py file.json ..."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    r, c = d["roofline"], d["config"]
    a = r.get("anatomy_us") or {}
    print("%-28s %6.3f M/s  period %5.2f us  launches %7.1f  acc/launch %6.1f  carried/acc %.2f  nnz %8.1f | loop %5.1f drain %4.1f handoff %4.1f draw %5.1f gap %4.1f" % (
        f.split("/")[-1], d["value"] / 1e6, r["kernel_ms_avg"] * 1e3, c.get("working_launches_per_iter", 0), r.get("accepted_per_launch", 0),
        r.get("columns_carried_per_accepted", 0), c["nnz_updates_per_iter"], a.get("entry_and_streaming_loop_us", 0),
        a.get("wave_reduction_and_drain_us", 0), a.get("hand_off_tickets_us", 0), a.get("draw_phase_us", 0), a.get("launch_gap_and_skew_us", 0)))
