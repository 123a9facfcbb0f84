# role kernels back to back (LIW_SERIAL_ROLES=1): per-kernel averages with an environment toggle on / off.  Usage: serial_probe.sh VAR
export TMPDIR=/tmp LIW_SERIAL_ROLES=1
V=${1:-LIW_NO_IMU_PACK}
for p in 0 1; do
[ $p = 1 ] && export $V=1 || unset $V
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_s$p -o x -- python bench.py --no-cpu-baseline --no-single --skip-sharded --steps 2 --warmup 1 > gpurun_out/bench_s$p.json 2> gpurun_out/bench_s.err < /dev/null
db=$(find gpurun_out/prof_s$p -name "*.db" | head -1); echo "$V=$p"; python tools/rocprof_summary.py $db | head -7 | cut -c1-150
python -c "
import json; d=json.loads(open('gpurun_out/bench_s$p.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
rm -rf gpurun_out/prof_s$p
done
