#!/bin/bash
# Run GPU steps one after the other on the box; a step that fails with an ordinary error does not stop the next one,
# a step that was KILLED (its own timeout: 124 / 137) or that ABORTED (134 / 139: a GPU fault ends the process that way) does -- nothing further is started on a GPU that may be hung.
# usage: gpu_steps.sh "<seconds>|<log name>|<command line>" ...
mkdir -p gpurun_out
for step in "$@"; do
  T=${step%%|*}; rest=${step#*|}; LOG=${rest%%|*}; CMD=${rest#*|}
  echo "== step: $CMD (limit ${T}s) -> gpurun_out/$LOG"
  timeout -k 10 $T bash -c "$CMD" > gpurun_out/$LOG 2>&1
  rc=$?
  echo "== rc $rc"; tail -n 4 gpurun_out/$LOG
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then echo "== killed at its limit or aborted (GPU fault?): stopping here"; exit $rc; fi
done
