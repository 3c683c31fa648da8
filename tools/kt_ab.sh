#!/bin/bash
# tools/kt_ab.sh LIB...: average duration of the filter instances on the serial C3 step, per library build
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in "$@"; do
  O=/tmp/kt_$lib; rm -rf $O
  PQV_LIB_PATH=$R/pq-vector_amd/libpqv_$lib.so rocprofv3 --kernel-trace -d $O -- python $R/bench.py --no-cpu --no-secondary --single 0 --recall 0 --parity-queries 0 --steps 3 --warmup 1 --streams 1 > /dev/null 2>&1
  echo "== $lib"; python $R/tools/rocpd_summary.py $(find $O -name "*.db" | head -1) --match wide_filter | grep -v "^#" | cut -c1-60,95-170
done
