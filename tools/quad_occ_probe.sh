# A/B of the quad step kernel compiled for 1 and 2 waves per SIMD: kernel time (bench) and SQ occupancy / issue counters (rocprofv3 --pmc)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for OCC in ${OCCS:-1 2}; do
  export LIW_QUAD_OCC=$OCC
  python -c "import __graft_entry__ as g; g.build()"
  python bench.py --no-cpu-baseline --no-single --skip-sharded --steps 3 > gpurun_out/qocc_bench_$OCC.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVES -d gpurun_out/pmc_q$OCC --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > gpurun_out/pmc_q$OCC.log 2>&1
  f=$(find gpurun_out/pmc_q$OCC -name "*counter_collection.csv" | head -1); cp $f gpurun_out/qocc_pmc_$OCC.csv; rm -rf gpurun_out/pmc_q$OCC
  python - <<P
import csv, collections, json
a=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open("gpurun_out/qocc_pmc_$OCC.csv")):
    if "k_lm_step_quad" in r["Kernel_Name"]:
        a[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("OCC=$OCC", {k:(round(v/max(n[k],1)),) for k,v in a.items()}, "launches", max(n.values()) if n else 0)
wc=a.get("SQ_WAVE_CYCLES",1)
print("  waves resident per busy cycle:", round(wc/max(a.get("SQ_BUSY_CYCLES",1),1),2), " wait_any %.3f wait_inst %.3f active %.3f valu %.3f" % tuple(a.get(k,0)/wc for k in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU")))
for l in open("gpurun_out/qocc_bench_$OCC.log"):
    if l.startswith("{"):
        d=json.loads(l); print("  bench value", d["value"], "lin", d["roofline"]["avg_launch_ms"], "step", d["roofline"]["lm_step_kernel_avg_ms"])
P
done
