#!/bin/bash
# Option scan on the GPU box: bench.py on one config under several library options; one line per variant.
# usage: tools/scan_opts.sh <config> <steps> "<opt list 1>" "<opt list 2>" ...   (an opt list: "name=value name=value", "" = defaults)
set -uo pipefail
cd "$(dirname "$0")/.."
cfg=$1; steps=$2; shift 2
for v in "$@"; do
  args=""
  for kv in $v; do args="$args --opt $kv"; done
  out=$(timeout -k 10 300 python3 bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-anatomy $args 2>/dev/null | tail -1)
  python3 - "$cfg" "$v" "$out" <<'PY'
import json, sys
cfg, v, out = sys.argv[1:4]
try:
    d = json.loads(out)
    r = d["roofline"]
    print("%-4s %-40s %8.3f M markers/s  %7.2f ms/step  rounds %8.1f  us/round %6.2f  events %8.1f  drift %.1e" % (
        cfg, v or "(defaults)", d["value"] / 1e6, d["ms_per_step"], r.get("rounds_per_iter", 0), r.get("us_per_round", 0), r.get("events_per_iter", 0), r.get("eps_sum_drift_max", 0)))
except Exception as e:
    print(cfg, v, "FAILED", repr(e), out[-300:])
PY
done
