"""bench.py's output contract: one JSON line with the required keys, single rank and — both ranks sharing the one GPU of the test
box over gloo (LIW_BENCH_SHARE_GPU=1) — the two-rank control flow incl. the factor-sharded C4 section."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "cpu_baseline"}


def last_json(out):
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_single_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "64", "--steps", "1", "--warmup", "0", "--cpu-reps", "1", "--cpu-procs", "2"],
                       capture_output=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and "model" not in d["config"]
    fs = d["factor_sharded"]                                     # N = 1: the un-sharded reference point of the factor-parallel curve
    assert "error" not in fs and fs["ranks"] == 1 and fs["scaling"] == "strong" and fs["solves_per_s"] > 0


def test_two_rank_control_flow_on_one_gpu():
    env = dict(os.environ, LIW_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and d["value"] > 0
    fs = d["factor_sharded"]
    assert "error" not in fs and fs["ranks"] == 2 and fs["scaling"] == "strong" and fs["solves_per_s"] > 0
    assert fs["allreduce_bytes_per_iteration"] == 256 * 30 * 128 * 8    # the laser partial region of the 256 C4 windows
