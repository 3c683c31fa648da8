# tools/kpp_ab.sh: A/B of kernels_kpp.hip build variants on ONE box (launch times differ by +-5 % between boxes): the default library against
# every pq-vector_amd/libpqv_var_*.so (built with -DKPP_... by hand), kpp_pick_kernel's average from a rocprofv3 kernel trace of one C3 build each
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kpp; mkdir -p $O
B="python $R/bench.py --workload c3 --no-cpu --no-secondary --no-configs --single 0 --recall 0 --parity-queries 0"
for lib in libpqv_hip.so $(cd $R/pq-vector_amd && ls libpqv_var_*.so 2>/dev/null); do
  for rep in 1 2; do
    rm -rf $O/kt
    PQV_LIB_PATH=$R/pq-vector_amd/$lib rocprofv3 --kernel-trace --stats -d $O/kt -- $B --steps 2 --warmup 1 > /dev/null 2>&1
    echo "$lib $(python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) --match pqv | grep -E "kpp" | awk '{print $1, $4, $5, $6}')"
  done
done
rm -rf $O/kt
