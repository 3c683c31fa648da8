#!/bin/bash
# C5 forms A/B: tests of the brute screen, then one bench run per PQV_BRUTE_RING mode on c5s and c5
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests -x -q -m gpu -k "brute" 2>&1 | tail -3
for wl in c5s c5; do for m in 0 1 2; do for rep in 1 2; do
  PQV_BRUTE_RING=$m python bench.py --workload $wl --steps 5 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$wl ring=$m', 'qps %.0f ms %.3f frac %.3f parity %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['ok']))"
done; done; done
