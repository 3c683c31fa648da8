#!/bin/bash
# Round-5 evidence in one GPU call: GPU tests, default bench line, rocprofv3 summaries (C3 uniform / mixture, C5, one-query), the
# from-Parquet benches (one GPU; two ranks sharing ONE file by row-group ranges), a fuzz batch.  -> gpurun_out/r05_ev/
R=${GRAFT_REPO_ROOT:-/root/repo}; E=$R/gpurun_out/r05_ev; mkdir -p $E; cd $R
tag=${1:-a}
python -m pytest tests -q -m gpu > $E/gpu_tests_$tag.log 2>&1; tail -2 $E/gpu_tests_$tag.log
python bench.py > $E/bench_default_line_$tag.json 2> $E/bench_default_line_$tag.err; echo "bench rc=$?"
python bench.py --from-parquet > $E/bench_from_parquet_$tag.json 2>/dev/null; echo "from-parquet rc=$?"
python bench.py --gpus 2 --backend gloo --from-parquet --rows-per-rank 1000000 --steps 10 --warmup 2 --parity-queries 64 > $E/bench_from_parquet_2ranks_$tag.json 2>/dev/null; echo "sharded from-parquet rc=$?"
bash tools/profile_round.sh r05 c3 > /dev/null 2>&1
bash tools/profile_round.sh r05mix c3 --data mixture > /dev/null 2>&1
bash tools/profile_round.sh r05 c5 > /dev/null 2>&1
bash tools/profile_single.sh r05 c3 > /dev/null 2>&1
FZ_LO=5000 FZ_HI=5150 FZ_TAG=default bash tools/fuzz_search.sh > $E/fuzz_$tag.txt 2>&1
FZ_LO=5150 FZ_HI=5250 FZ_TAG="drain 8, xcd 3, fork" PQV_DRAIN_MIN=8 PQV_XCD_ITEMS=3 PQV_FORK_WIDE=1 bash tools/fuzz_search.sh >> $E/fuzz_$tag.txt 2>&1
FZ_LO=5250 FZ_HI=5350 FZ_TAG="regular quads only" PQV_WIDE_QUADS=0 PQV_SEED_ROWS=64 PQV_WIDE_ROWS=256 bash tools/fuzz_search.sh >> $E/fuzz_$tag.txt 2>&1
grep "done" $E/fuzz_$tag.txt
ls $R/gpurun_out/prof_r05 $R/gpurun_out/prof_r05mix
