"""bench.py --gpus N without a launcher re-executes itself under torch.distributed.run with N ranks (the driver's command shape for
N = 1; for N > 1 the driver starts the ranks itself).  No GPU here: every rank must get as far as the loud "needs a GPU" refusal —
the estimator has no CPU fallback — and the launcher must report the failure; a mismatch between --gpus and the ranks started is
refused before anything else."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour (the GPU twin is tests/test_gpu_bench_contract.py)")


def run(args, env_extra=None, timeout=900):   # a fresh container pages torch in for a minute or two
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_spawns_two_ranks_that_refuse_to_run_without_a_gpu():
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                  # no JSON line without a measurement
    refusals = r.stderr.count("bench.py needs a GPU")
    assert refusals >= 1
    # two ranks were started: both refused, or the launcher reports rank 1 (it stops the others when the first rank fails)
    assert refusals >= 2 or "local_rank: 1" in r.stderr

def test_rank_count_mismatch_is_refused():
    r = run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 4 but the launcher started 2 rank(s)" in r.stderr
