# tools/kpp_prof.sh: rocprofv3 kernel trace of one C3 bench run, the k-means++ kernels' lines -> gpurun_out/kpp/kpp_kernel_trace.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kpp; mkdir -p $O
B="python $R/bench.py --workload c3 --no-cpu --no-secondary --no-configs --single 0 --recall 0 --parity-queries 0"
rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) --match pqv > $O/kpp_kernel_trace.txt
grep -E "^#|kpp|minupd|stream_kernel" $O/kpp_kernel_trace.txt
rm -rf $O/kt
