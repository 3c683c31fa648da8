#!/bin/bash
# A/B of the deferred exact evaluation (PQV_DEFER = 0 / 1: never / by rule) on bench workloads: q/s, ms, exact evaluations per query
R=${GRAFT_REPO_ROOT:-/root/repo}
for wl in "$@"; do
  X=""; [ $wl = refbench ] && X="--k 100"
  for d in ${DEFER_VALUES:-0 1}; do
    PQV_DEFER=$d python $R/bench.py --workload $wl $X --no-cpu --no-secondary --no-configs --single 0 --recall 0 --steps 200 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=l['counters']
print('$wl defer=$d', round(l['value']), round(l['ms_per_step'],4), 'serial', round(l.get('ms_per_step_serial',0),4), 'exact/q', round(c['screen_survivors']/max(1,c['queries']),1), 'kernel_ms', round(l['roofline']['kernel_ms'],4), 'min_bytes_frac', round(l['roofline']['min_bytes_frac'],3))"
  done
done
