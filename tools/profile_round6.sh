#!/bin/bash
# Round-6 evidence pass (one gpurun call): bash tools/profile_round6.sh TAG
#   1. tools/profile_round.sh TAG (C2: SQ issue / MFMA passes, kernel trace forked + serial roles, FETCH_SIZE / WRITE_SIZE -> profiles/pmc_traffic.json,
#      stand-alone kernel times, quad phases, tracking frame probe; its bench line is skipped here),
#   2. the batched TRACK leg: kernel trace + FETCH_SIZE / WRITE_SIZE passes of tools/track_batch_probe.py -> pmc_track.json,
#   3. the ragged leg (tools/ragged_probe.py), 4. the default bench line LAST.
TAG=${1:-r06_v1}; B=${2:-49152}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
SKIP_BENCH=1 bash tools/profile_round.sh $TAG $B > gpurun_out/${TAG}_profile_round.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tk -o track -- python $R/tools/track_batch_probe.py $B 8 --no-cpu > $R/gpurun_out/${TAG}_track_batch_under_rocprof.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_tf --output-format csv -- python $R/tools/track_batch_probe.py $B 4 --no-cpu > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_tw --output-format csv -- python $R/tools/track_batch_probe.py $B 4 --no-cpu > /dev/null 2>&1
cd $R
db=$(find gpurun_out/prof_tk -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_track_batch_kernel_stats_b${B}.csv > /dev/null
ff=$(find gpurun_out/prof_tf -name "*counter_collection.csv" | head -1); fw=$(find gpurun_out/prof_tw -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && cp $ff gpurun_out/${TAG}_track_pmc_fetch_size_b${B}.csv; [ -n "$fw" ] && cp $fw gpurun_out/${TAG}_track_pmc_write_size_b${B}.csv
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_track.py $ff $fw $B gpurun_out/${TAG}_pmc_track.json > gpurun_out/${TAG}_pmc_track.log 2>&1 && cp gpurun_out/${TAG}_pmc_track.json profiles/pmc_track.json
rm -rf gpurun_out/prof_tk gpurun_out/prof_tf gpurun_out/prof_tw
python tools/track_batch_probe.py $B 8 > gpurun_out/${TAG}_track_batch_probe.json 2> /dev/null
python tools/ragged_probe.py $B > gpurun_out/${TAG}_ragged_probe.json 2> /dev/null
python bench.py --batch $B > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err < /dev/null
ls -la gpurun_out | grep ${TAG}
