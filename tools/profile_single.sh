#!/bin/bash
# rocprofv3 evidence for ONE-query calls on a workload: kernel trace + FETCH_SIZE of the screened default and of the
# per-query streaming kernel (PQV_RERANK_MODE=1, the north-star kernel as literally specified).
#   tools/profile_single.sh <tag> <workload>  ->  gpurun_out/prof_<tag>/<tag>_<workload>_single_{trace,pmc_FETCH_SIZE}.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r02}; wl=${2:-c3}
O=$R/gpurun_out/prof_$tag; mkdir -p $O
B="python $R/bench.py --workload $wl --no-cpu --no-secondary --no-configs --recall 0 --steps 2 --warmup 1 --single 100"
: > $O/${tag}_${wl}_single_trace.txt; : > $O/${tag}_${wl}_single_pmc_FETCH_SIZE.txt
for mode in rule stream; do
  echo "## PQV_RERANK_MODE=$mode (rule: screened dispatch, stream: stream_kernel per query)" | tee -a $O/${tag}_${wl}_single_trace.txt >> $O/${tag}_${wl}_single_pmc_FETCH_SIZE.txt
  PQV_RERANK_MODE=$mode rocprofv3 --kernel-trace --stats -d $O/skt -- $B > $O/single_$mode.json 2>/dev/null
  python $R/tools/rocpd_summary.py $(find $O/skt -name "*.db" | head -1) --match pqv | awk '$0 ~ /^#/ || $0 ~ / (10[0-9]|1[1-9][0-9]|2[0-9][0-9]) +[0-9.]+ +[0-9.]+ +/' >> $O/${tag}_${wl}_single_trace.txt
  python -c "import json,sys; d=json.loads(open('$O/single_$mode.json').read().strip().splitlines()[-1]); print('# single_query', json.dumps(d['single_query']))" >> $O/${tag}_${wl}_single_trace.txt
  rm -rf $O/skt
  PQV_RERANK_MODE=$mode rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/spmc -- $B > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $O/spmc -name "*.db" | head -1) --match pqv | grep FETCH_SIZE | awk '$0 ~ / (10[0-9]|1[1-9][0-9]|2[0-9][0-9]) +[0-9.]+ +[0-9.]+ *$/' >> $O/${tag}_${wl}_single_pmc_FETCH_SIZE.txt
  rm -rf $O/spmc $O/single_$mode.json
done
cat $O/${tag}_${wl}_single_trace.txt $O/${tag}_${wl}_single_pmc_FETCH_SIZE.txt
