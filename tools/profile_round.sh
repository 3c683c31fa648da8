#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_${1:-r01c}; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --no-cpu --steps 10 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) --match pqv > $O/c2_kernel_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_$c -name "*.db" | head -1) --match pqv | grep -E "^#|wide_|merge_kernel|stream_kernel<32, 1, 0|pair_|fill_ones|seed_select" > $O/c2_pmc_$c.txt
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/pmc_sq1 -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/pmc_sq1 -name "*.db" | head -1) --match wide_ | grep -v "^#" > $O/c2_pmc_sq.txt
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $O/pmc_sq2 -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/pmc_sq2 -name "*.db" | head -1) --match wide_ | grep -v "^#" >> $O/c2_pmc_sq.txt
rocprofv3 --kernel-trace --stats -d $O/kt3 -- python $R/bench.py --workload c3 --no-cpu --steps 3 --warmup 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt3 -name "*.db" | head -1) --match pqv > $O/c3_kernel_trace.txt
rm -rf $O/kt $O/kt3 $O/pmc_*
ls -la $O
