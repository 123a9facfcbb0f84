#!/bin/bash
# End-of-round evidence pass on the MI355X box (run through gpurun): default bench line, kernel-trace stats with forked and
# serial role kernels, PMC traffic passes (separate, kernel-trace only).  Outputs under gpurun_out/$TAG_*.
TAG=${1:-r01_v6}; B=${2:-6144}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_a -o a -- python bench.py --no-cpu-baseline --no-single --skip-sharded > gpurun_out/${TAG}_bench_b${B}_under_rocprof.json 2> gpurun_out/prof_a.err
LIW_SERIAL_ROLES=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o s -- python bench.py --no-cpu-baseline --no-single --skip-sharded > gpurun_out/${TAG}_bench_b${B}_serial_roles_under_rocprof.json 2> gpurun_out/prof_s.err
for d in a s; do db=$(find gpurun_out/prof_$d -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_stats_$d.csv; f=$(find gpurun_out/prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_kstats_$d.csv; done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_f --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > /dev/null 2> gpurun_out/prof_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_w --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > /dev/null 2> gpurun_out/prof_w.err
cp $(find gpurun_out/prof_f -name "*counter_collection.csv" | head -1) gpurun_out/${TAG}_pmc_fetch_size_b${B}.csv
cp $(find gpurun_out/prof_w -name "*counter_collection.csv" | head -1) gpurun_out/${TAG}_pmc_write_size_b${B}.csv
rm -rf gpurun_out/prof_a gpurun_out/prof_s gpurun_out/prof_f gpurun_out/prof_w
ls -la gpurun_out | grep ${TAG}
