#!/bin/bash
# tools/single_sweep.sh "ENV=VAL ..."...: C3 single-query p50 / p99 per environment setting (one searcher per run)
for envs in "$@"; do
  env $envs python bench.py --no-cpu --no-secondary --no-configs --recall 0 --parity-queries 0 --steps 5 --single 300 > /dev/null 2>&1
  python -c "
import json
s = json.load(open('bench_full.json'))['single_query']
print('$envs', 'device p50 %.1f p99 %.1f | pqv_topk p50 %.1f p99 %.1f' % (s['p50_us'], s['p99_us'], s['host_api_p50_us'], s['host_api_p99_us']))"
done
