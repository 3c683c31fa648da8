#!/bin/bash
# fresh-process build time of C3 for several assignment chunk sizes (one process each: allocation costs included)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r04i}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "assign or build or index or kmeans or hostile or config_scale or lloyd" > $O/tests.log 2>&1; tail -2 $O/tests.log
for ch in ${2:-1048576 262144}; do
  for rep in 1 2; do
    PQV_ASSIGN_CHUNK=$ch PQV_VERBOSE=1 timeout 300 python bench.py --no-secondary --no-configs --no-cpu --single 0 --recall 0 --steps 5 > $O/b_$ch.json 2> $O/b_$ch.err
    echo "chunk $ch: $(grep -E 'k-means\+\+ rounds|k-means\+\+:|final assignment: 1|build: 0|enqueued' $O/b_$ch.err | sed 's/\[pqv\] //' | tr '\n' '|')"
  done
done
