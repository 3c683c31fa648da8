#!/bin/bash
# PMC passes for one kernel of a bench workload.  usage: tools/pmc_kernel.sh <tag> <kernel-substring> <bench args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; match=$2; shift 2
out=$R/gpurun_out/pmc_$tag.txt; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcd_${tag}_$i -- python $R/bench.py "$@" > /dev/null 2>$R/gpurun_out/pmcd_${tag}_$i.err
  f=$(find $R/gpurun_out/pmcd_${tag}_$i -name "*.db" | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_summary.py $f --match "$match" | grep -v "^#" >> $out
  rm -rf $R/gpurun_out/pmcd_${tag}_$i $R/gpurun_out/pmcd_${tag}_$i.err
done
grep -E "SQ_|TCC_|TCP_" $out | sed "s/.*x1 //"
