#!/bin/bash
# All bench records of a round: tools/bench_round.sh <tag>  ->  gpurun_out/<tag>/bench_*.json
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-round}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
run() { name=$1; shift; "$@" 2>$O/bench_$name.err | tail -1 > $O/bench_$name.json; python - <<PY
import json
try:
    r = json.load(open("$O/bench_$name.json")); print("$name", round(r["value"], 1), r["unit"], round(r["ms_per_step"], 4), "ms/step; kernel_ms", r["roofline"].get("kernel_ms"), "frac", r["roofline"].get("frac"))
except Exception as e:
    print("$name FAILED", e)
PY
}
run c2           python bench.py --workload c2 --steps 40
run c2_serial    python bench.py --workload c2 --steps 40 --streams 1
run c2_f32screen env PQV_SCREEN_F16=0 python bench.py --workload c2 --steps 40
run c2_single    python bench.py --workload c2 --steps 20 --single 200
run c3           python bench.py --workload c3 --steps 20
run c4           python bench.py --workload c4 --steps 10
run rb           python bench.py --workload refbench --k 100 --steps 6
run c2s2         python bench.py --workload c2s2 --steps 40
run c2s4         python bench.py --workload c2s4 --steps 40
run c2s8         python bench.py --workload c2s8 --steps 40
run c3h          python bench.py --workload c3h --steps 20
run c2d          python bench.py --workload c2d --steps 40
run c5           python bench.py --workload c5 --steps 5
