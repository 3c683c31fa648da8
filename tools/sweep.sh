#!/bin/bash
# usage: tools/sweep.sh <out-file> <bench args...> -- "ENV=.. ENV=.." "ENV=.." ...   (one bench run per env set)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
out=$1; shift
args=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do args+=("$1"); shift; done; shift
: > $out
for e in "$@"; do
  line=$(env $e python bench.py "${args[@]}" --no-cpu 2>/dev/null | tail -1)
  python - "$e" "$line" >> $out <<'PY'
import json, sys
e, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line); r = d["roofline"]
    print(f"{e:60s} qps {d['value']:12.1f} ms/step {d['ms_per_step']:.4f} kernel_ms {r.get('kernel_ms', 0):.4f} surv {d['counters']['screen_survivors']}")
except Exception as ex:
    print(f"{e:60s} FAILED {ex} {line[:200]}")
PY
done
cat $out
