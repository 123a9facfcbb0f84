# the three GPU soak sweeps that exercise the step / linearise kernels, side by side in one gpurun call (25 minutes each at most):
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/soak_round.sh'   -> gpurun_out/soak_r05_*.log
cd $GRAFT_REPO_ROOT
( timeout 1500 python tests/soak/soak_batch.py 0 3000 > gpurun_out/soak_r05_batch.log 2>&1 ) &
( timeout 1500 python tests/soak/soak_random_shapes.py 12 4000 > gpurun_out/soak_r05_shapes.log 2>&1 ) &
( timeout 1500 python tests/soak/soak_c2.py 0 40 15 > gpurun_out/soak_r05_c2.log 2>&1 ) &
wait
for f in batch shapes c2; do tail -n 1 gpurun_out/soak_r05_$f.log; done
