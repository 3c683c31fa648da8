#!/bin/bash
# PMC counters of brute_f16_kernel on the 1 M-row C5 shape (separate passes, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_c5; mkdir -p $O; : > $O/c5s_pmc.txt; echo "# workload ${WL:-c5s}" >> $O/c5s_pmc.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" ; do
  rm -rf /tmp/pc5; rocprofv3 --pmc $set --kernel-trace -d /tmp/pc5 -- python $R/bench.py --workload ${WL:-c5s} --steps 2 --no-cpu > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find /tmp/pc5 -name "*.db" | head -1) --match brute_f16 | grep -v "^#" >> $O/c5s_pmc.txt
done
cut -c1-30,100-260 $O/c5s_pmc.txt
