cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()"
timeout 1500 python -m pytest tests/test_gpu_quad_step.py tests/test_gpu_large_batch_paths.py tests/test_gpu_laser_slab.py tests/test_gpu_bench_shape.py tests/test_gpu_batch.py -m gpu -q -x > gpurun_out/ab9_tests.log 2>&1
tail -3 gpurun_out/ab9_tests.log
python tools/ktimes.py 49152 2>&1 | grep -v amdgpu
python bench.py --no-cpu-baseline > gpurun_out/ab9_bench.json 2> gpurun_out/ab9_bench.err
python -c "
import json; d=json.load(open('gpurun_out/ab9_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_gate']['passed'], d['kernel_times']['k_lin_laser']['ms'], d['tracking_frame_latency'])"
export LIW_EXTRA_FLAGS="-DLIW_SLAB_WPLANE=0"
python -c "import __graft_entry__ as g; g.build()"
python bench.py --no-cpu-baseline > gpurun_out/ab9_bench_nowp.json 2> gpurun_out/ab9_bench_nowp.err
python -c "
import json; d=json.load(open('gpurun_out/ab9_bench_nowp.json')); print('no weight plane:', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_gate']['passed'], d['kernel_times']['k_lin_laser']['ms'])"
