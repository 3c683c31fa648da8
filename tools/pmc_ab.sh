#!/bin/bash
# tools/pmc_ab.sh TAG [bench args]: per-kernel durations + FETCH_SIZE of the serial C3 step (one rocprofv3 --pmc pass), for A/B runs
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
O=$R/gpurun_out/pmcab_$tag; mkdir -p $O
B="python $R/bench.py --no-cpu --no-secondary --single 0 --recall 0 --parity-queries 0 $*"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/p -- $B --steps 3 --warmup 1 --streams 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/p -name "*.db" | head -1) --match pqv | grep -E "^#|wide_|merge_kernel|probe_rows|quantize_|pair_|seed_" > $R/gpurun_out/pmcab_$tag.txt
rm -rf $O
cat $R/gpurun_out/pmcab_$tag.txt | cut -c1-60,95-250
