# third helping: seed ranges no earlier round has run (every tracking solve of soak_random_shapes.py is a two-frame window: k_lm_step_dense2)
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/soak_round3.sh'   -> gpurun_out/soak_r05c_*.log
cd $GRAFT_REPO_ROOT
( timeout 2400 python tests/soak/soak_batch.py 5000 8000 > gpurun_out/soak_r05c_batch.log 2>&1 ) &
( timeout 2400 python tests/soak/soak_random_shapes.py 9000 15000 > gpurun_out/soak_r05c_shapes.log 2>&1 ) &
( timeout 2400 python tests/soak/soak_random_shapes.py 15000 21000 > gpurun_out/soak_r05c_shapes2.log 2>&1 ) &
( timeout 2400 python tests/soak/soak_c2.py 40 100 15 > gpurun_out/soak_r05c_c2.log 2>&1 ) &
wait
for f in batch shapes shapes2 c2; do echo "== $f"; tail -n 1 gpurun_out/soak_r05c_$f.log; done
