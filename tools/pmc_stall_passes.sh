cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --list-avail > gpurun_out/avail.txt 2>&1   # counter names of this box
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pmc_$name --output-format csv -- python bench.py --batch ${B:-4096} --steps 1 --warmup 0 --no-cpu-baseline --no-single --skip-sharded > gpurun_out/pmc_$name.log 2>&1; echo "$name rc=$?"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run st SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run st2 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
find gpurun_out/pmc_* -name "*counter_collection.csv" | while read f; do d=$(echo $f | cut -d/ -f2); cp $f gpurun_out/$d.csv; done
rm -rf gpurun_out/pmc_ic gpurun_out/pmc_st gpurun_out/pmc_st2
