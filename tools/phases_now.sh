#!/bin/bash
# tools/phases_now.sh: per-wave phase profile of the filter instances on one serial C3 step (diagnostic build: make phases)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for data in uniform mixture; do
PQV_LIB_PATH=$R/pq-vector_amd/libpqv_hip_phases.so PQV_PHASES_OUT=$O/phases_$data.bin python bench.py --workload c3 --data $data --steps 1 --warmup 1 --no-cpu --no-secondary --no-configs --single 0 --recall 0 --parity-queries 0 --streams 1 --no-timing > /dev/null 2>$O/ph.err
python tools/phase_timeline.py $O/phases_$data.bin > $O/phases_$data.txt 2>&1; rm -f $O/phases_$data.bin
cat $O/phases_$data.txt
done
