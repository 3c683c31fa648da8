#!/bin/bash
# tools/kt_c5.sh "ENV=VAL ..."...: kernel trace of one C5 bench run per environment setting (PQV_BRUTE_OP, PQV_BRUTE_STAGE, PQV_BRUTE_TILE)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for envs in "$@"; do
  i=$((i+1)); O=/tmp/kt_c5_$i; rm -rf $O
  env $envs rocprofv3 --kernel-trace -d $O -- python $R/bench.py --workload ${WL:-c5} --steps 2 --no-cpu > /dev/null 2>&1
  echo "== $envs"; python $R/tools/rocpd_summary.py $(find $O -name "*.db" | head -1) --match brute | grep -v "^#" | cut -c1-80,95-175 | head -${HEAD:-4}
done
