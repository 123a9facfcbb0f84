#!/bin/bash
# the two halves of the IMU role, each alone (probe builds, wrong results): bound of the split into a dual-number kernel and a matrix-core kernel
cd $GRAFT_REPO_ROOT
for ph in 0 1 2; do
  LIW_EXTRA_FLAGS="-DLIW_IMU_PROBE_PHASE=$ph" python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)"
  echo "--- LIW_IMU_PROBE_PHASE=$ph"
  bash tools/kernel_regs.sh k_linearize.hip -DLIW_IMU_PROBE_PHASE=$ph 2>/dev/null | grep k_lin_imu_chain
  LIW_EXTRA_FLAGS="-DLIW_IMU_PROBE_PHASE=$ph" python tools/ktimes.py 49152 2>&1 | grep -v amdgpu
  LIW_EXTRA_FLAGS="-DLIW_IMU_PROBE_PHASE=$ph" python tools/ktimes.py 49152 2>&1 | grep -v amdgpu
done
python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)"
