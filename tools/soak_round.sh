#!/bin/bash
# GPU soak sweeps side by side in one gpurun call: bash tools/soak_round.sh TAG SET   -> gpurun_out/soak_${TAG}_*.log
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/soak_round.sh r05d 4'
# SET 1: the three sweeps that exercise the step / linearise kernels (25 minutes at most); 2: extended seed ranges + API fuzz / replay /
# pose graph / pre-integration; 3, 4: seed ranges no earlier round has run (every tracking solve of soak_random_shapes.py is a two-frame
# window: k_lm_step_dense2, k_marg_schur4).  Results: tests/soak/README.md, profiles/*_soak*.txt.
TAG=${1:-r05}; SET=${2:-1}; T=${3:-2400}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift; ( timeout $T "$@" > gpurun_out/soak_${TAG}_$name.log 2>&1 ) & }
case $SET in
1) run batch python tests/soak/soak_batch.py 0 3000; run shapes python tests/soak/soak_random_shapes.py 12 4000; run c2 python tests/soak/soak_c2.py 0 40 15;;
2) run batch python tests/soak/soak_batch.py 3000 5000; run shapes python tests/soak/soak_random_shapes.py 4000 9000
   run fuzz python tests/soak/soak_api_fuzz.py 0 60; run replay python tests/soak/soak_replay.py 3 14
   run pg python tests/soak/soak_posegraph.py 0 200; run preint python tests/soak/soak_preint.py 0 600;;
3) run batch python tests/soak/soak_batch.py 5000 8000; run shapes python tests/soak/soak_random_shapes.py 9000 15000
   run shapes2 python tests/soak/soak_random_shapes.py 15000 21000; run c2 python tests/soak/soak_c2.py 40 100 15;;
4) run batch python tests/soak/soak_batch.py 8000 9500; run shapes python tests/soak/soak_random_shapes.py 21000 24500
   run shapes2 python tests/soak/soak_random_shapes.py 24500 28000; run fuzz python tests/soak/soak_api_fuzz.py 60 100;;
5) run batch python tests/soak/soak_batch.py 9500 11500; run shapes python tests/soak/soak_random_shapes.py 35000 38000
   run slab python tests/soak/soak_slab.py 0 400; run c2 python tests/soak/soak_c2.py 100 130 15;;
6) run batch python tests/soak/soak_batch.py 13600 15600; run slab python tests/soak/soak_slab.py 400 1000
   run fuzz python tests/soak/soak_api_fuzz.py 130 170; run replay python tests/soak/soak_replay.py 14 22
   run pg python tests/soak/soak_posegraph.py 200 400; run preint python tests/soak/soak_preint.py 600 1000;;
7) run batch python tests/soak/soak_batch.py 15600 17000; run slab python tests/soak/soak_slab.py 1000 1300
   run c2 python tests/soak/soak_c2.py 150 175 15; run shapes python tests/soak/soak_random_shapes.py 38000 39500;;
8) run batch python tests/soak/soak_batch.py 17000 19500; run slab python tests/soak/soak_slab.py 1300 2300
   run fuzz python tests/soak/soak_api_fuzz.py 170 220; run c2 python tests/soak/soak_c2.py 175 215 15;;
esac
wait
for f in gpurun_out/soak_${TAG}_*.log; do echo "== $f"; tail -n 2 $f; done
