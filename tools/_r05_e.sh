cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_bench_contract.py::test_eight_rank_control_flow_on_one_gpu 2>&1 | tail -16 > gpurun_out/r05_e_tests.log
python tools/track_probe.py > gpurun_out/r05_e_track.log 2>&1
tail -14 gpurun_out/r05_e_tests.log; tail -1 gpurun_out/r05_e_track.log
