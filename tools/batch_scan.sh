#!/bin/bash
# Throughput against queries per step on one workload: tools/batch_scan.sh <workload> <out-file> [nq ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
wl=${1:-c3}; out=${2:-/dev/stdout}; shift 2
[ $# -eq 0 ] && set -- 128 256 512 768 1024 1536 2048 4096
echo "# $wl: queries/step, QPS, ms/step (2 stream lanes), re-rank kernels ms (serial), min_bytes GB, screen survivors per query" > $out
for n in "$@"; do
  python bench.py --workload $wl --no-cpu --steps 30 --single 0 --recall 0 --nq $n 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']; c = d['counters']
print(f\"{$n:6d} {d['value']:12.0f} {d['ms_per_step']:8.3f} {r['kernel_ms']:8.3f} {r['min_bytes'] / 1e9:8.2f} {c['screen_survivors'] / max(1, c['queries']):8.1f}\")
" >> $out
done
cat $out
