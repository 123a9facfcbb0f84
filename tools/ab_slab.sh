#!/bin/bash
# A/B of build variants on the GPU box: bash tools/ab_slab.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("-" = the default build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
B=${B:-49152}
for v in "$@"; do
  [ "$v" = "-" ] && unset LIW_EXTRA_FLAGS || export LIW_EXTRA_FLAGS="$v"
  python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)" 2>&1 | tail -3
  echo "== variant: $v"
  python tools/ktimes.py $B 2>&1 | grep -v amdgpu.ids
done
