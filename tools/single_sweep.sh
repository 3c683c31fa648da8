#!/bin/bash
# tools/single_sweep.sh "ENV=VAL ..."...: C3 single-query p50 / p99 per environment setting (one searcher per run)
for envs in "$@"; do
  env $envs python bench.py --no-cpu --no-secondary --recall 0 --parity-queries 0 --steps 5 --single 300 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['single_query']
print('$envs', 'p50 %.1f p99 %.1f mean %.1f' % (s['p50_us'], s['p99_us'], s['mean_us']))"
done
