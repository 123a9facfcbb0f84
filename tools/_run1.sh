cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
LIW_CLK=1 python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)" 2>&1 | tail -3
LIW_CLK=1 python tools/clk_probe_slab.py 2>&1 | grep -v amdgpu.ids
python -c "import importlib; b=importlib.import_module('2dliw-slam_amd.build'); b.build(force=True)" 2>&1 | tail -3
python tools/bracket_time.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_laser_slab.py tests/test_gpu_bench_shape.py tests/test_gpu_large_batch_paths.py -m gpu -q -x 2>&1 | tail -2
