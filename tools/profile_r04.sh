#!/bin/bash
# Round-4 evidence pass (one gpurun call): tools/profile_round.sh (kernel-trace forked + serial roles, FETCH / WRITE PMC passes, traffic
# calibration, default bench line) + the SQ issue / MFMA passes + the quad-kernel occupancy and memory probes + the staging micro-benchmark.
TAG=${1:-r04_v1}; B=${2:-24576}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()"
B=$B bash tools/pmc_stall_passes.sh > gpurun_out/${TAG}_stall.log 2>&1
cp gpurun_out/pmc_st.csv gpurun_out/${TAG}_pmc_wave_cycles_b${B}.csv; cp gpurun_out/pmc_st2.csv gpurun_out/${TAG}_pmc_mfma_b${B}.csv
SKIP_BENCH=1 bash tools/profile_round.sh $TAG $B > gpurun_out/${TAG}_round.log 2>&1
python tools/pmc_issue_stats.py gpurun_out/${TAG}_pmc_wave_cycles_b${B}.csv gpurun_out/${TAG}_pmc_mfma_b${B}.csv gpurun_out/${TAG}_pmc_traffic.json "$TAG" > gpurun_out/${TAG}_issue.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
./tools/ubench/dma_rate > gpurun_out/${TAG}_ubench_dma_rate.log 2>&1
python tools/ktimes.py > gpurun_out/${TAG}_ktimes.log 2>&1
LIW_NO_LASER_SLAB=1 python tools/ktimes.py >> gpurun_out/${TAG}_ktimes.log 2>&1
python tools/two_stream_probe.py $B > gpurun_out/${TAG}_two_stream.log 2>&1
# (tools/quad_occ_probe2.sh belongs to the kernel of r04_v1 / v2: since the gather table and the staged second sweep a wave holds 38 kB of LDS,
#  four waves fill a CU and a two-waves-per-SIMD build no longer exists)
python tools/clk_probe_quad.py $B 100 15 3 > gpurun_out/${TAG}_quad_phases.log 2>&1
# the default bench line once more with the issue statistics of this build in place
python bench.py --batch $B > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err < /dev/null
ls gpurun_out | grep ${TAG}
