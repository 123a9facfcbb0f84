# shared-resource counters of k_lm_step_quad (LDS conflicts / FIFOs, TA FIFOs, instruction fetch), three rocprofv3 --pmc passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$name --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded --gate-windows 0 > gpurun_out/pmc_$name.log 2>&1; f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1); cp $f gpurun_out/quadpmc_$name.csv; rm -rf gpurun_out/pmc_$name; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run b SQ_WAVE_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM
run c SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA
python - <<P
import csv, collections
for nm in "abc":
    a=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open("gpurun_out/quadpmc_%s.csv"%nm)):
        if "k_lm_step_quad" in r["Kernel_Name"]:
            a[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    wc=a.get("SQ_WAVE_CYCLES",1.0)
    print(nm, {k:"%.4g (%.3f of wave cycles)"%(v, v/wc) for k,v in a.items()})
P
