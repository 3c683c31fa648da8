#!/bin/bash
# tools/phases_wl.sh WORKLOAD...: per-wave phase profile of the filter instances on one serial step of each workload (make phases)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for wl in "$@"; do
X=""; [ $wl = refbench ] && X="--k 100"
PQV_LIB_PATH=$R/pq-vector_amd/libpqv_hip_phases.so PQV_PHASES_OUT=$O/phases_$wl.bin python bench.py --workload $wl $X --steps 1 --warmup 1 --no-cpu --no-secondary --no-configs --single 0 --recall 0 --parity-queries 0 --streams 1 --no-timing > /dev/null 2>$O/ph.err
echo "## $wl"; python tools/phase_timeline.py $O/phases_$wl.bin 2>&1 | tee $O/phases_$wl.txt; rm -f $O/phases_$wl.bin
done
