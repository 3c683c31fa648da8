#!/bin/bash
# tools/kt_c5.sh OP...: kernel trace of one C5 bench run per operand form of the brute-force screen (PQV_BRUTE_OP)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for op in "$@"; do
  O=/tmp/kt_c5_$op; rm -rf $O
  PQV_BRUTE_OP=$op rocprofv3 --kernel-trace -d $O -- python $R/bench.py --workload ${WL:-c5} --steps 2 --no-cpu > /dev/null 2>&1
  echo "== $op"; python $R/tools/rocpd_summary.py $(find $O -name "*.db" | head -1) --match brute | grep -v "^#" | cut -c1-80,95-175
done
