#!/bin/bash
# Round-6 evidence in one GPU call: GPU tests, the default bench line (compact: what the driver parses) + its full record, rocprofv3
# summaries (C3 uniform / mixture, C5, one-query), the from-Parquet benches (one GPU; two and EIGHT ranks sharing ONE file by
# row-group ranges on this GPU), the C4 path with eight small shards, a fuzz batch.  -> gpurun_out/r06_ev/
R=${GRAFT_REPO_ROOT:-/root/repo}; E=$R/gpurun_out/r06_ev; mkdir -p $E; cd $R
tag=${1:-a}
python -m pytest tests -q -m gpu > $E/gpu_tests_$tag.log 2>&1; tail -2 $E/gpu_tests_$tag.log
python bench.py > $E/bench_default_line_$tag.json 2> $E/bench_default_line_$tag.err; echo "bench rc=$? bytes=$(wc -c < $E/bench_default_line_$tag.json)"
cp bench_full.json $E/bench_default_full_$tag.json
python bench.py --from-parquet > $E/bench_from_parquet_$tag.json 2>/dev/null; echo "from-parquet rc=$?"
python bench.py --gpus 2 --backend gloo --from-parquet --rows-per-rank 1000000 --steps 10 --warmup 2 --parity-queries 64 > $E/bench_from_parquet_2ranks_$tag.json 2>/dev/null; echo "2-rank from-parquet rc=$?"
python bench.py --gpus 8 --backend gloo --from-parquet --rows-per-rank 250000 --steps 5 --warmup 1 --parity-queries 64 > $E/bench_from_parquet_8ranks_$tag.json 2>/dev/null; echo "8-rank from-parquet rc=$?"
python bench.py --gpus 8 --backend gloo --workload c4 --rows-per-rank 500000 --steps 5 --warmup 1 > $E/bench_c4_8ranks_small_$tag.json 2>/dev/null; echo "8-rank c4 rc=$?"
bash tools/profile_round.sh r06 c3 > /dev/null 2>&1
bash tools/profile_round.sh r06mix c3 --data mixture > /dev/null 2>&1
bash tools/profile_round.sh r06 c5 > /dev/null 2>&1
bash tools/profile_single.sh r06 c3 > /dev/null 2>&1
FZ_LO=7000 FZ_HI=7150 FZ_TAG=default bash tools/fuzz_search.sh > $E/fuzz_$tag.txt 2>&1
FZ_LO=7150 FZ_HI=7250 FZ_TAG="list_once 1, defer 0" PQV_LIST_ONCE=1 PQV_DEFER=0 bash tools/fuzz_search.sh >> $E/fuzz_$tag.txt 2>&1
FZ_LO=7250 FZ_HI=7350 FZ_TAG="regular quads only, loose seeds, small blocks" PQV_WIDE_QUADS=0 PQV_SEED_ROWS=64 PQV_WIDE_ROWS=256 bash tools/fuzz_search.sh >> $E/fuzz_$tag.txt 2>&1
FZ_LO=300 FZ_HI=360 bash tools/fuzz_build.sh >> $E/fuzz_$tag.txt 2>&1
grep "done" $E/fuzz_$tag.txt
ls $R/gpurun_out/prof_r06 $R/gpurun_out/prof_r06mix
