#!/bin/bash
# tools/env_ab.sh OUT VAR VALUE...: the default C3 bench line once per value of an environment knob, twice round-robin (A/B on one box)
out=$1; var=$2; shift 2
rm -f gpurun_out/$out
for rep in 1 2; do
  for v in "$@"; do
    env $var=$v AB_TAG="$var=$v" AB_REPS=1 bash tools/ab.sh $out hip
  done
done
cat gpurun_out/$out
