# issue / wait counters of the laser kernels (rocprofv3 --pmc over tools/ktimes.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES -d gpurun_out/pmc_slab --output-format csv -- python tools/ktimes.py > gpurun_out/pmc_slab.log 2>&1
f=$(find gpurun_out/pmc_slab -name "*counter_collection.csv" | head -1); cp $f gpurun_out/slab_pmc.csv; rm -rf gpurun_out/pmc_slab
python - <<P
import csv, collections
a=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open("gpurun_out/slab_pmc.csv")):
    k=r["Kernel_Name"].split("(")[0][:40]
    a[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k,v in a.items():
    wc=v.get("SQ_WAVE_CYCLES",0)
    if wc<1e6: continue
    L=max(n[k].values())
    print(k, "launches",L, "waves/launch", v.get("SQ_WAVES",0)/L, "valu insts/wave", v.get("SQ_INSTS_VALU",0)/max(v.get("SQ_WAVES",1),1), " wait_any %.3f wait_inst %.3f active %.3f valu %.3f" % tuple(v.get(c,0)/wc for c in ("SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU")))
P
