#!/bin/bash
# A/B of build variants on the tracking frame (GPU box): bash tools/ab_track.sh "<flags 1>" "<flags 2>" ...   ("-" = the default build)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for round in 1 2; do
for v in "$@"; do
  [ "$v" = "-" ] && unset LIW_EXTRA_FLAGS || export LIW_EXTRA_FLAGS="$v"
  python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)" 2>&1 | grep -i " error"
  echo "== variant: $v"
  for i in 1 2 3; do python tools/track_probe.py 2>&1 | grep -v amdgpu.ids; done
done
done
