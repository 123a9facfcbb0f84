#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the default workload only (separate runs, --kernel-trace only) -> profiles/pmc_traffic.json: bash tools/profile_pmc_traffic.sh TAG [BATCH]
TAG=${1:-r06_v3}; B=${2:-49152}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_f --output-format csv -- python bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > /dev/null 2> gpurun_out/prof_f.err < /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_w --output-format csv -- python bench.py --batch $B --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > /dev/null 2> gpurun_out/prof_w.err < /dev/null
ff=$(find gpurun_out/prof_f -name "*counter_collection.csv" | head -1); fw=$(find gpurun_out/prof_w -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && cp $ff gpurun_out/${TAG}_pmc_fetch_size_b${B}.csv
[ -n "$fw" ] && cp $fw gpurun_out/${TAG}_pmc_write_size_b${B}.csv
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch_size_b${B}.csv gpurun_out/${TAG}_pmc_write_size_b${B}.csv $B 30 2000 gpurun_out/${TAG}_pmc_traffic.json
rm -rf gpurun_out/prof_f gpurun_out/prof_w
