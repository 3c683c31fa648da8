#!/bin/bash
# build-path check of a round-4 library: the build parity tests, the build phases of C3 (verbose), a kernel trace of the build
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r04c}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "assign or build or index or kmeans or hostile or config_scale or lloyd" > $O/tests_build.log 2>&1; tail -4 $O/tests_build.log
PQV_VERBOSE=1 timeout 600 python bench.py --no-secondary --no-configs --no-cpu --single 0 --recall 0 > $O/bench_c3.json 2> $O/bench_c3.err
grep -E "pqv\]|\[bench\] shard" $O/bench_c3.err | head -40
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -- python $R/bench.py --no-secondary --no-configs --no-cpu --single 0 --recall 0 --steps 5 > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) --match pqv > $O/c3_build_kernel_trace.txt
rm -rf $O/kt
head -40 $O/c3_build_kernel_trace.txt
