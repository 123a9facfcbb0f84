# second helping of GPU soak sweeps (extended seed ranges + the API fuzz / replay / pose-graph / pre-integration sweeps), side by side:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/soak_round2.sh'   -> gpurun_out/soak_r05b_*.log
cd $GRAFT_REPO_ROOT
( timeout 2000 python tests/soak/soak_batch.py 3000 5000 > gpurun_out/soak_r05b_batch.log 2>&1 ) &
( timeout 2000 python tests/soak/soak_random_shapes.py 4000 9000 > gpurun_out/soak_r05b_shapes.log 2>&1 ) &
( timeout 2000 python tests/soak/soak_api_fuzz.py 0 60 > gpurun_out/soak_r05b_fuzz.log 2>&1; timeout 900 python tests/soak/soak_replay.py 3 14 > gpurun_out/soak_r05b_replay.log 2>&1 ) &
( timeout 1200 python tests/soak/soak_posegraph.py 0 200 > gpurun_out/soak_r05b_pg.log 2>&1; timeout 1200 python tests/soak/soak_preint.py 0 600 > gpurun_out/soak_r05b_preint.log 2>&1 ) &
wait
for f in batch shapes fuzz replay pg preint; do echo "== $f"; tail -n 2 gpurun_out/soak_r05b_$f.log; done
