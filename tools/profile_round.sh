#!/bin/bash
# rocprofv3 summaries of one bench workload, as committed under profiles/:
#   tools/profile_round.sh <tag> <workload> [extra bench args]
#   -> gpurun_out/prof_<tag>/<tag>_<workload>_{kernel_trace,pmc_FETCH_SIZE,pmc_WRITE_SIZE,pmc_sq}.txt
# Kernel trace: the bench's default (pipelined) run.  PMC: separate passes (--pmc with --kernel-trace only), steps issued
# serially so a dispatch's counters belong to it alone.  bench.py reads the FETCH/WRITE summaries back for roofline.traffic.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r02}; wl=${2:-c3}; shift 2
O=$R/gpurun_out/prof_$tag; mkdir -p $O
B="python $R/bench.py --workload $wl --no-cpu --no-secondary --no-configs --single 0 --recall 0 $*"
rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 20 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) --match pqv > $O/${tag}_${wl}_kernel_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -- $B --steps 3 --warmup 1 --streams 1 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_$c -name "*.db" | head -1) --match pqv | grep -E "^#|wide_|list_|tile_|merge_kernel|stream_kernel|probe_rows|quantize_|pair_|seed_" > $O/${tag}_${wl}_pmc_$c.txt
done
: > $O/${tag}_${wl}_pmc_sq.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  rocprofv3 --pmc $set --kernel-trace -d $O/pmc_sq -- $B --steps 3 --warmup 1 --streams 1 > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $(find $O/pmc_sq -name "*.db" | head -1) --match wide_ | grep -v "^#" >> $O/${tag}_${wl}_pmc_sq.txt
  rm -rf $O/pmc_sq
done
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
