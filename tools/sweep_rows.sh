#!/bin/bash
# tools/sweep_rows.sh: rows per block of the two filter instances (PQV_WIDE_ROWS x PQV_WIDE_QUAD_ROWS) on C3, uniform and mixture
for data in uniform mixture; do
  for wr in ${WR:-2048 3072 4096}; do
    for wq in ${WQ:-4096 6144 8192}; do
      PQV_WIDE_ROWS=$wr PQV_WIDE_QUAD_ROWS=$wq AB_TAG="$data rows=$wr wide=$wq" AB_REPS=1 AB_ARGS="--single 0 --data $data" bash tools/ab.sh sweep_rows.txt hip
    done
  done
done
cat gpurun_out/sweep_rows.txt
