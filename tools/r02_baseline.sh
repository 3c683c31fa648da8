#!/bin/bash
# Round-2 starting point on C3: bench records (2 lanes / serial), per-wave phase timeline of the screen kernel,
# PMC passes of the screen kernel.   usage: tools/r02_baseline.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r02a}; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
python bench.py --workload c3 --steps 20 --no-cpu 2>$O/c3.err | tail -1 > $O/bench_c3.json
python bench.py --workload c3 --steps 20 --no-cpu --streams 1 2>$O/c3s.err | tail -1 > $O/bench_c3_serial.json
PQV_LIB_PATH=$R/pq-vector_amd/libpqv_hip_phases.so PQV_PHASES_OUT=$O/phases_c3.bin python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu --streams 1 --no-timing > /dev/null 2>$O/ph.err
python tools/phase_timeline.py $O/phases_c3.bin > $O/phases_c3.txt 2>&1; rm -f $O/phases_c3.bin
cd /tmp
bash $R/tools/pmc_kernel.sh ${tag}_c3 wide_filter --workload c3 --no-cpu --steps 2 --warmup 1 --streams 1 > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -- python $R/bench.py --workload c3 --no-cpu --steps 2 --warmup 1 --streams 1 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_$c -name "*.db" | head -1) --match pqv > $O/c3_pmc_$c.txt
  rm -rf $O/pmc_$c
done
ls -la $O
