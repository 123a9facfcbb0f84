cd $GRAFT_REPO_ROOT
( timeout 1500 python tests/soak/soak_batch.py 0 3000 > gpurun_out/soak_r04_batch.log 2>&1 ) &
( timeout 1500 python tests/soak/soak_random_shapes.py 12 4000 > gpurun_out/soak_r04_shapes.log 2>&1 ) &
( timeout 1500 python tests/soak/soak_c2.py 0 40 15 > gpurun_out/soak_r04_c2.log 2>&1 ) &
wait
tail -3 gpurun_out/soak_r04_batch.log gpurun_out/soak_r04_shapes.log gpurun_out/soak_r04_c2.log
