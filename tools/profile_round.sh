#!/bin/bash
# THE evidence pass of a round on the MI355X box (one gpurun call; usage: bash tools/profile_round.sh TAG [BATCH]):
#   1. SQ issue / MFMA counter passes (tools/pmc_stall_passes.sh; --pmc runs carry --kernel-trace only),
#   2. rocprofv3 --kernel-trace --stats of the bench command, role kernels forked and serial (LIW_SERIAL_ROLES=1) -> *_kernel_stats_*.csv,
#   3. FETCH_SIZE / WRITE_SIZE passes (separate runs) + per-kernel calibration (tools/pmc_traffic.py) -> profiles/pmc_traffic.json,
#   4. stand-alone kernel times (tools/ktimes.py), per-phase stamps of the quad step kernel, the staging micro-benchmark,
#   5. the default bench line LAST, with the traffic / issue statistics of this very build in place.
# Outputs under gpurun_out/${TAG}_*; copy what is to be judged into profiles/.
TAG=${1:-r05_v1}; B=${2:-49152}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
B=$B bash tools/pmc_stall_passes.sh > gpurun_out/${TAG}_stall.log 2>&1
cp gpurun_out/pmc_st.csv gpurun_out/${TAG}_pmc_wave_cycles_b${B}.csv; cp gpurun_out/pmc_st2.csv gpurun_out/${TAG}_pmc_mfma_b${B}.csv
for d in a s; do
  [ $d = s ] && export LIW_SERIAL_ROLES=1 || unset LIW_SERIAL_ROLES
  name=$([ $d = s ] && echo _serial_roles || echo "")
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$d -o $d -- python bench.py --batch $B --no-cpu-baseline --no-single --skip-sharded > gpurun_out/${TAG}_bench_b${B}${name}_under_rocprof.json 2> gpurun_out/prof_$d.err < /dev/null
  db=$(find gpurun_out/prof_$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_kernel_stats_b${B}${name}.csv > /dev/null
done
unset LIW_SERIAL_ROLES
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_f --output-format csv -- python bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > /dev/null 2> gpurun_out/prof_f.err < /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_w --output-format csv -- python bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > /dev/null 2> gpurun_out/prof_w.err < /dev/null
ff=$(find gpurun_out/prof_f -name "*counter_collection.csv" | head -1); fw=$(find gpurun_out/prof_w -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && cp $ff gpurun_out/${TAG}_pmc_fetch_size_b${B}.csv
[ -n "$fw" ] && cp $fw gpurun_out/${TAG}_pmc_write_size_b${B}.csv
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch_size_b${B}.csv gpurun_out/${TAG}_pmc_write_size_b${B}.csv $B 30 2000 gpurun_out/${TAG}_pmc_traffic.json
[ -f gpurun_out/${TAG}_kernel_stats_b${B}_serial_roles.csv ] && python tools/serial_roles_frac.py gpurun_out/${TAG}_kernel_stats_b${B}_serial_roles.csv gpurun_out/${TAG}_bench_b${B}_serial_roles_under_rocprof.json gpurun_out/${TAG}_pmc_traffic.json
rm -rf gpurun_out/prof_a gpurun_out/prof_s gpurun_out/prof_f gpurun_out/prof_w
python tools/pmc_issue_stats.py gpurun_out/${TAG}_pmc_wave_cycles_b${B}.csv gpurun_out/${TAG}_pmc_mfma_b${B}.csv gpurun_out/${TAG}_pmc_traffic.json "$TAG" > gpurun_out/${TAG}_issue.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_rate.hip -o tools/ubench/dma_rate 2> /dev/null && ./tools/ubench/dma_rate > gpurun_out/${TAG}_ubench_dma_rate.log 2>&1
python tools/ktimes.py $B > gpurun_out/${TAG}_ktimes.log 2>&1
LIW_NO_LASER_SLAB=1 python tools/ktimes.py $B >> gpurun_out/${TAG}_ktimes.log 2>&1
python tools/clk_probe_quad.py $B 100 15 3 > gpurun_out/${TAG}_quad_phases.log 2>&1
python tools/track_probe.py > gpurun_out/${TAG}_track_probe.log 2>&1
# the default bench line LAST, with the traffic / issue statistics of this very build in place
[ -z "$SKIP_BENCH" ] && python bench.py --batch $B > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err < /dev/null
ls -la gpurun_out | grep ${TAG}
