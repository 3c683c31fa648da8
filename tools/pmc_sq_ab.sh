#!/bin/bash
# tools/pmc_sq_ab.sh TAG: SQ / LDS / TCC counters of the two filter instances on the serial C3 step
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
O=$R/gpurun_out/pmcsq_$tag; mkdir -p $O
B="python $R/bench.py --no-cpu --no-secondary --single 0 --recall 0 --parity-queries 0 $*"
: > $R/gpurun_out/pmcsq_$tag.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  rocprofv3 --pmc $set --kernel-trace -d $O/p -- $B --steps 3 --warmup 1 --streams 1 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $O/p -name "*.db" | head -1) --match wide_filter | grep -v "^#" | grep -E "SQ_|TCC_|TCP_" | cut -c1-60,95-200 >> $R/gpurun_out/pmcsq_$tag.txt
  rm -rf $O/p
done
cat $R/gpurun_out/pmcsq_$tag.txt
