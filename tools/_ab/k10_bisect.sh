#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "--- $*"; LIW_BENCH_K10_SPLIT=1 python bench.py --no-cpu-baseline --skip-sharded --track-batch 0 --ragged-batch 0 --replay-keep30-seconds 0 --replay-seconds 2 "$@" 2>&1 >/dev/null | grep k10; }
run --gate-windows 0
run --gate-windows 4
LIW_NO_EARLY_EXIT=1 run --gate-windows 4
