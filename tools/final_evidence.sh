#!/bin/bash
# End-of-round evidence that is not part of tools/bench_round.sh / profile_round.sh:
#   one-query rocprofv3 trace + FETCH_SIZE on C3, and `bench.py --gpus 2` started the way the driver starts N = 1
#   (no launcher; two ranks sharing the one device over gloo, and the 1-rank RCCL exchange through the C ABI).
#   tools/final_evidence.sh <tag>  ->  gpurun_out/prof_<tag>/*, gpurun_out/<tag>_multi/*
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r03}; M=$R/gpurun_out/${tag}_multi; mkdir -p $M
cd $R
timeout 600 python bench.py --gpus 2 --backend gloo --workload c2 --no-cpu --steps 20 > $M/bench_c2_gpus2_gloo.json 2> $M/bench_c2_gpus2_gloo.err; echo "gpus2 gloo rc=$?"
timeout 600 python bench.py --force-dist --workload c2 --no-cpu --steps 20 > $M/bench_c2_rccl_1rank.json 2> $M/bench_c2_rccl_1rank.err; echo "force-dist rc=$?"
for f in $M/*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "n_gpus", d["n_gpus"], round(d["value"], 1), d["unit"], "exchange", json.dumps(d.get("exchange"))[:400])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 900 bash tools/profile_single.sh $tag c3 | tail -40
