#!/bin/bash
# kernel-trace passes only (forked + serial roles of the default workload, the batched TRACK leg): bash tools/profile_kernel_stats.sh TAG [BATCH]
TAG=${1:-r06_v3}; B=${2:-49152}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
for d in a s; do
  [ $d = s ] && export LIW_SERIAL_ROLES=1 || unset LIW_SERIAL_ROLES
  name=$([ $d = s ] && echo _serial_roles || echo "")
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$d -o $d -- python bench.py --batch $B --no-cpu-baseline --no-single --skip-sharded > gpurun_out/${TAG}_bench_b${B}${name}_under_rocprof.json 2> gpurun_out/prof_$d.err < /dev/null
  db=$(find gpurun_out/prof_$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_kernel_stats_b${B}${name}.csv > /dev/null
done
unset LIW_SERIAL_ROLES
R=$GRAFT_REPO_ROOT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tk -o track -- python $R/tools/track_batch_probe.py $B 8 --no-cpu > /dev/null 2>&1
cd $R; db=$(find gpurun_out/prof_tk -name "*.db" | head -1); [ -n "$db" ] && python tools/rocprof_summary.py $db gpurun_out/${TAG}_track_batch_kernel_stats_b${B}.csv > /dev/null
rm -rf gpurun_out/prof_a gpurun_out/prof_s gpurun_out/prof_tk
python tools/ktimes.py $B > gpurun_out/${TAG}_ktimes.log 2>&1
ls -la gpurun_out | grep ${TAG}
