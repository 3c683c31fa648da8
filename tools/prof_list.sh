#!/bin/bash
# round 6: list_filter_kernel against the filter instances on one workload -- serial kernel trace + the counters that say what it waits for
#   tools/prof_list.sh <tag> <data: uniform|mixture> [env assignments]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r06}; data=${2:-mixture}; shift 2
O=$R/gpurun_out/prof_$tag; mkdir -p $O
B="env $* python $R/bench.py --workload c3 --data $data --no-cpu --no-secondary --no-configs --single 0 --recall 0 --steps 3 --warmup 1 --streams 1"
rocprofv3 --kernel-trace --stats -d $O/kt -- $B > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) --match pqv > $O/${tag}_${data}_kernel_trace_serial.txt
: > $O/${tag}_${data}_pmc.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_x -- $B > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_x -name "*.db" | head -1) --match "filter_kernel" | grep -v "^#" >> $O/${tag}_${data}_pmc.txt
  rm -rf $O/pmc_x
done
rm -rf $O/kt
cat $O/${tag}_${data}_kernel_trace_serial.txt | head -30; cat $O/${tag}_${data}_pmc.txt
