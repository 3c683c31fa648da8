#!/bin/bash
# tools/regs_c3.sh [-DFLAG ...]: registers / spills / LDS of C3's two screen instances alone (PQV_DEV_C3_ONLY: seconds instead of minutes),
# and their ISA in /tmp/regs/c3.s
mkdir -p /tmp/regs
cd "$(dirname "$0")/../pq-vector_amd/csrc"
bash ../../tools/kernel_regs.sh kernels_screen.hip -DPQV_DEV_C3_ONLY "$@" | grep wide_filter
if [ -n "$REGS_ASM" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -S --cuda-device-only \
      -DPQV_DEV_C3_ONLY "$@" kernels_screen.hip -o /tmp/regs/c3.s 2>/dev/null
fi
