#!/bin/bash
# prepare first (in the container, before the gpurun call):  git show 7f276e9:2dliw-slam_amd/csrc/k_laser_slab.hip > tools/_ab/k_laser_slab_r5.hip
# same-box A/B of the INIT lane-per-group laser kernel: round-6 source (templated on the topology) against the round-5 source
cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/ktimes.py 49152; done
cp 2dliw-slam_amd/csrc/k_laser_slab.hip /tmp/k_laser_slab_new.hip
cp tools/_ab/k_laser_slab_r5.hip 2dliw-slam_amd/csrc/k_laser_slab.hip
python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)"
echo "--- round-5 source"
for i in 1 2; do python tools/ktimes.py 49152; done
cp /tmp/k_laser_slab_new.hip 2dliw-slam_amd/csrc/k_laser_slab.hip
