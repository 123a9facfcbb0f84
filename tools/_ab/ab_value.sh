#!/bin/bash
# prepare first (in the container, before the gpurun call):  mkdir -p tools/_ab/r5 && git archive 7f276e9 -- 2dliw-slam_amd oracle bench.py include __graft_entry__.py | tar -x -C tools/_ab/r5 && mkdir -p tools/_ab/r5/profiles && git show 7f276e9:profiles/pmc_traffic.json > tools/_ab/r5/profiles/pmc_traffic.json
# same-box A/B of the default bench workload: round-5 tree (tools/_ab/r5, commit 7f276e9) against this tree
cd $GRAFT_REPO_ROOT/tools/_ab/r5 && python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for i in 1 2; do
  cd $GRAFT_REPO_ROOT/tools/_ab/r5; python bench.py --no-cpu-baseline --no-single --skip-sharded 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5  ', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['lm_step_kernel_avg_ms'])"
  cd $GRAFT_REPO_ROOT; python bench.py --no-cpu-baseline --no-single --skip-sharded 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['lm_step_kernel_avg_ms'])"
done
