#!/bin/bash
# One parameterised sweep instead of a script per experiment (run on the GPU box through gpurun):
#   tools/gpu/sweep.sh OUT VAR "v1 v2 ..." [REPS] -- <command ...>
# runs <command> once per value with VAR=value exported, REPS times each (default 1), keeps every log under
# gpurun_out/OUT/ and prints one line per run: the value of the bench JSON line if the command printed one, else its
# last line.  VAR may be "-" (no variable: a plain repeat).  Every run is bounded by `timeout -k 10 ${SWEEP_TIMEOUT:-300}`
# and a run that times out or fails ends the sweep (no GPU step is started behind a killed one).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$1; VAR=$2; VALS=$3; shift 3
REPS=1
if [ "${1:-}" != "--" ]; then REPS=$1; shift; fi
shift
mkdir -p "$OUT"
for v in $VALS; do
  for rep in $(seq 1 "$REPS"); do
    log="$OUT/${VAR}_${v}_$rep.log"
    if [ "$VAR" = "-" ]; then timeout -k 10 "${SWEEP_TIMEOUT:-300}" "$@" > "$log" 2>&1
    else env "$VAR=$v" timeout -k 10 "${SWEEP_TIMEOUT:-300}" "$@" > "$log" 2>&1; fi
    rc=$?
    python3 - "$log" "$VAR" "$v" "$rc" <<'PY'
import json, sys
log, var, v, rc = sys.argv[1:5]
lines = [l for l in open(log, errors="replace").read().splitlines() if l.strip()]
last = lines[-1] if lines else ""
try:
    d = json.loads(last)
    extra = " split %s" % d["pool_split"] if d.get("pool_split") else ""
    print("%s=%s rc=%s %.3f M %s/s form %s%s" % (var, v, rc, d["value"] / 1e6, d.get("unit", "?").split("/")[0], d.get("step_form"), extra))
except Exception:
    print("%s=%s rc=%s %s" % (var, v, rc, last[:160]))
PY
    if [ "$rc" -ne 0 ]; then echo "sweep stopped: rc $rc"; exit "$rc"; fi
  done
done
