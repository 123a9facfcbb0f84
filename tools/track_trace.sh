cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/track_probe.py 100
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_t -o t -- python tools/track_probe.py 100 > gpurun_out/trk.log 2>&1
db=$(find gpurun_out/prof_t -name "*.db" | head -1); python tools/rocprof_summary.py $db gpurun_out/trk_stats.csv
python - <<'PY'
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/prof_t/**/*.db',recursive=True)[0])
rows=list(db.execute("select s.kernel_name,d.start,d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
# last full frame: take the last 60 dispatches
rows=rows[-45:]
t0=rows[0][1]
for n,s,e in rows: print("%-40s start %8.1f us dur %6.1f us"%(n[:40].replace('_ZN3liw',''),(s-t0)/1e3,(e-s)/1e3))
PY
rm -rf gpurun_out/prof_t
