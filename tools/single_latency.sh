#!/bin/bash
# single-query latency under option sets: tools/single_latency.sh "ENV=.." ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for e in "$@"; do
  line=$(env $e python bench.py --no-cpu --steps 5 --recall 0 --single 300 2>/dev/null | tail -1)
  python - "$e" "$line" <<'PY'
import json, sys
e, line = sys.argv[1], sys.argv[2]
d = json.loads(line); s = d["single_query"]
print(f"{e:50s} p50 {s['p50_us']:.1f} p99 {s['p99_us']:.1f}  {s['dispatch'][:150]}")
PY
done
