#!/bin/bash
# tools/variant.sh NAME [-DFLAG ...]: an experiment build of the kernels under pq-vector_amd/libpqv_v_NAME.so (api.o / exchange.o
# of the regular build are reused); run with PQV_LIB_PATH=pq-vector_amd/libpqv_v_NAME.so.
set -e
cd "$(dirname "$0")/../pq-vector_amd/csrc"
name=$1; shift
make -s api.o exchange.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-fast-math -Wall -Wno-unused-result "$@" -c kernels_all.hip -o kernels_v_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libpqv_v_$name.so kernels_v_$name.o api.o exchange.o -ldl
rm -f kernels_v_$name.o
echo built libpqv_v_$name.so
