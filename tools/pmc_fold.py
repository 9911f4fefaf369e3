#!/usr/bin/env python3
"""Fold the PMC pass summaries of tools/profile_round.sh into the two JSON files bench.py quotes (with their source):
  <R>_<tag>_pmc_traffic.json   memory-side bytes per working launch of k_sweep_batch over the TIMED iterations
                            (FETCH_SIZE doubled -- the gfx950 correction of MI355X_MICROARCH.md, HBM section -- plus
                            WRITE_SIZE; both counters are in KiB)
  <R>_<tag>_pmc_valu.json    SQ / GRBM counters of the same launches: VALU issue share, wave occupancy, wait shares
usage: pmc_fold.py <dir with the pass outputs> <round tag> [workload tag, default c4]"""
import json
import os
import sys

O, R = sys.argv[1], sys.argv[2]
TAG = sys.argv[3] if len(sys.argv) > 3 else "c4"


def last_json(path):
    with open(path) as f:
        return json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])


def bench_of(name):
    return last_json(os.path.join(O, "pmc_%s.json" % name))


def work(name, ctr):
    return last_json(os.path.join(O, "%s_%s_pmc_%s.txt" % (R, TAG, name)))[ctr]


b = bench_of("FETCH_SIZE")
cfg = b["config"]
cmd = "rocprofv3 --pmc <counters> -- python3 bench.py --steps %d --warmup %d --no-cpu-baseline --no-anatomy" % (b["steps"], b["warmup"])
f, w = work("FETCH_SIZE", "FETCH_SIZE"), work("WRITE_SIZE", "WRITE_SIZE")
traffic = (2.0 * f["mean_work"] + w["mean_work"]) * 1024.0
json.dump({
    "command": cmd + " (FETCH_SIZE and WRITE_SIZE in separate passes, counters only); dispatches of the timed iterations only "
               "(the last launches_per_iter x steps dispatches of the sweep kernel), summarised by tools/rocpd_pmc.py + tools/pmc_fold.py",
    "workload": cfg["workload"], "N": cfg["N"], "M": cfg["M"], "missing_rate": cfg.get("missing_rate", 0.0), "batch": cfg["batch"], "steps": b["steps"], "warmup": b["warmup"],
    "kernel": b["roofline"]["kernel"],
    "dispatches_timed": f["n"], "dispatches_doing_work": f["n_work"],
    "FETCH_SIZE_KiB_mean_per_working_launch": f["mean_work"], "FETCH_SIZE_KiB_median_per_working_launch": f["median_work"],
    "WRITE_SIZE_KiB_mean_per_working_launch": w["mean_work"],
    "correction": "gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact",
    "traffic_bytes_per_launch": traffic,
    "traffic_bytes_per_iteration": (2.0 * f["sum"] + w["sum"]) * 1024.0 / b["steps"],
    "accepted_per_launch": cfg.get("accepted_per_launch"), "columns_streamed_per_accepted": cfg.get("columns_streamed_per_accepted"),
}, open(os.path.join(O, "%s_%s_pmc_traffic.json" % (R, TAG)), "w"), indent=1)

s1 = {k: work("SQ1", k) for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE")}
s2 = {k: work("SQ2", k) for k in ("SQ_INSTS_SALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                  "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE")}
b1 = bench_of("SQ1")
# SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles summed over all waves (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units");
# GRBM_GUI_ACTIVE counts shader-engine clock cycles the GPU was busy with the dispatch.  A SIMD issues at most one VALU
# instruction per quad-cycle, so the chip-wide VALU issue capacity of a launch is GUI_ACTIVE / 4 quad-cycles x 1024 SIMDs
# (256 CUs x 4): valu_issue_frac = SQ_ACTIVE_INST_VALU / that.
# rocprofv3 sums a counter over its instances: GRBM_GUI_ACTIVE comes back as the sum over the 8 XCDs (its per-launch value / 8
# x the launch's duration gives the engine clock, 2.1-2.4 GHz), SQ counters as the sum over all SIMDs' waves
XCDS = 8
gui = s1["GRBM_GUI_ACTIVE"]["mean_work"] / XCDS
cap = gui / 4.0 * 1024.0
json.dump({
    "command": cmd + " (two SQ passes)", "workload": b1["config"]["workload"], "N": b1["config"]["N"], "M": b1["config"]["M"], "missing_rate": b1["config"].get("missing_rate", 0.0),
    "kernel": b1["roofline"]["kernel"],
    "steps": b1["steps"], "warmup": b1["warmup"], "dispatches_timed": s1["SQ_INSTS_VALU"]["n"],
    "per_working_launch_mean": {k: v["mean_work"] for k, v in {**s1, **{k: v for k, v in s2.items() if k != "GRBM_GUI_ACTIVE"}}.items()},
    "GRBM_GUI_ACTIVE_second_pass": s2["GRBM_GUI_ACTIVE"]["mean_work"], "gui_active_cycles_per_xcd": gui,
    "units": "SQ_ACTIVE_INST_*, SQ_WAVE_CYCLES, SQ_WAIT_*: quad-cycles summed over waves; SQ_BUSY_CYCLES: cycles summed over shader engines / XCDs as rocprofv3 "
             "reports it; GRBM_GUI_ACTIVE: cycles; SQ_INSTS_*: wave-instructions",
    "valu_issue_frac": s1["SQ_ACTIVE_INST_VALU"]["mean_work"] / cap if cap else None,
    "valu_issue_frac_definition": "SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 4 x 1024 SIMDs): share of the chip's VALU issue slots used over the whole launch",
    "wave_occupancy_of_1024_simds": s1["SQ_WAVE_CYCLES"]["mean_work"] / cap if cap else None,
    "wave_occupancy_definition": "SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 4 x 1024): average resident waves per SIMD over the launch (idle CUs during the draw phase pull it down)",
    "wait_any_share_of_wave_cycles": s2["SQ_WAIT_ANY"]["mean_work"] / s1["SQ_WAVE_CYCLES"]["mean_work"] if s1["SQ_WAVE_CYCLES"]["mean_work"] else None,
    "active_inst_any_share_of_wave_cycles": s2["SQ_ACTIVE_INST_ANY"]["mean_work"] / s1["SQ_WAVE_CYCLES"]["mean_work"] if s1["SQ_WAVE_CYCLES"]["mean_work"] else None,
    "valu_insts_per_wave": s1["SQ_INSTS_VALU"]["mean_work"] / s1["SQ_WAVES"]["mean_work"] if s1["SQ_WAVES"]["mean_work"] else None,
}, open(os.path.join(O, "%s_%s_pmc_valu.json" % (R, TAG)), "w"), indent=1)
print(open(os.path.join(O, "%s_%s_pmc_traffic.json" % (R, TAG))).read())
print(open(os.path.join(O, "%s_%s_pmc_valu.json" % (R, TAG))).read())
