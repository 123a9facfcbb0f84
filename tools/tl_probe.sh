export TMPDIR=/tmp
for p in 0 1; do
export LIW_SIDE_PRIO=$p
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_x$p -o x -- python bench.py --no-cpu-baseline --no-single --skip-sharded --steps 1 --warmup 1 > gpurun_out/bench_tl$p.json 2> gpurun_out/bench_tl.err < /dev/null
db=$(find gpurun_out/prof_x$p -name "*.db" | head -1); echo PRIO $p; python tools/rocprof_timeline.py $db 3 2
python -c "
import json; d=json.loads(open('gpurun_out/bench_tl$p.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
rm -rf gpurun_out/prof_x$p
done
