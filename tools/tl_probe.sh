# time line of the LM iterations of one bench step (run through gpurun): rocprofv3 --kernel-trace + tools/rocprof_timeline.py
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_tl -o x -- python bench.py --no-cpu-baseline --no-single --skip-sharded --steps 1 --warmup 1 > gpurun_out/bench_tl.json 2> gpurun_out/bench_tl.err < /dev/null
db=$(find gpurun_out/prof_tl -name "*.db" | head -1)
python tools/rocprof_timeline.py $db ${1:-3} ${2:-2}
rm -rf gpurun_out/prof_tl
