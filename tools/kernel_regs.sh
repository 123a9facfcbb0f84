#!/bin/bash
# register / LDS / scratch use of the kernels of one translation unit (compiler remarks; no GPU needed)
# usage: tools/kernel_regs.sh k_lm_quad.hip [extra hipcc flags]
SRC=$1; shift
D=$(dirname "$0")/../2dliw-slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result --cuda-device-only -c "$D/$SRC" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' | awk '
/Function Name:/ {name=$NF}
/ VGPRs:/ {v=$NF} / AGPRs:/ {a=$NF} /TotalSGPRs:/ {s=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF} /VGPRs Spill/ {vs=$NF}
/LDS Size/ {printf "%-62s vgpr %3d agpr %3d sgpr %3d scratch %4d vspill %3d occ %d lds %6d\n", name, v, a, s, sc, vs, oc, $NF}'
