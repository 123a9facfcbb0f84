#!/bin/bash
cd $GRAFT_REPO_ROOT
for fl in "-DLIW_SMALL_PROBE_INCACHE=1" ""; do
  LIW_EXTRA_FLAGS="$fl" python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)"
  echo "--- $fl"
  LIW_EXTRA_FLAGS="$fl" python tools/ktimes.py 49152 2>&1 | grep -v amdgpu
  LIW_EXTRA_FLAGS="$fl" python tools/ktimes.py 49152 2>&1 | grep -v amdgpu
done
