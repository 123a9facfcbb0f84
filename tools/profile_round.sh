#!/bin/bash
# End-of-round evidence pass on the MI355X box (run through gpurun): default bench line, kernel-trace stats with forked and
# serial role kernels, PMC traffic passes (separate, kernel-trace only), per-kernel FETCH_SIZE calibration + consistency check
# (tools/pmc_traffic.py) and the serial-roles roofline fraction.  Outputs under gpurun_out/$TAG_* and profiles/pmc_traffic.json.
TAG=${1:-r03_v4}; B=${2:-24576}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
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
# the default bench line LAST, with the traffic file of this very build in place
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
[ -z "$SKIP_BENCH" ] && python bench.py --batch $B > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err < /dev/null
ls -la gpurun_out | grep ${TAG}
