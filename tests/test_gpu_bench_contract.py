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
        "roofline", "cpu_baseline", "parity_gate", "roofline_lm_step"}


def last_json(out):
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.decode()[-2000:]
    return json.loads(lines[0])


def test_single_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "64", "--steps", "1", "--warmup", "0", "--cpu-reps", "2", "--cpu-warmup", "0",
                        "--cpu-procs", "2", "--distinct", "8"],
                       capture_output=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert {"achieved_counter_gbs", "frac_counter"} <= set(d["roofline"])
    pg = d["parity_gate"]                                        # the timed batch itself went through the oracle
    assert pg["passed"] and pg["iterations_equal"] and pg["max_rel_state_err_per_iteration"] <= 1e-6 and pg["windows"] >= 1
    assert {"median_ms", "p95_ms", "host_cpu_model"} <= set(d["cpu_baseline"])
    assert "lm_hits_iteration_cap_pct" in d["config"] and "lm_iterations_histogram" in d["config"]
    fs = d["factor_sharded"]                                     # N = 1: the un-sharded reference point of the factor-parallel curve
    assert "error" not in fs and fs["ranks"] == 1 and fs["scaling"] == "strong" and fs["solves_per_s"] > 0


def _check_two_ranks(d):
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None and d["value"] > 0
    assert d["parity_gate"]["passed"]
    fs = d["factor_sharded"]
    assert "error" not in fs, fs
    assert fs["ranks"] == 2 and fs["scaling"] == "strong" and fs["solves_per_s"] > 0
    assert fs["process_group"]["world_size"] == 2
    assert fs["allreduce_bytes_per_iteration"] == 8 * (256 * 30 * 45 + 1)    # compact record: 45 pair totals per (window, frame) + the active count
    assert fs["uncompacted_record_bytes"] == 256 * 30 * 128 * 8
    assert fs["allreduce_ms_per_iteration"] > 0 and fs["states_identical_across_ranks"]
    one = fs["oneshot_exchange"]
    assert one["solves_per_s"] > 0 and one["states_identical_across_ranks"] and one["exchange_ms_per_iteration"] > 0


def test_two_rank_control_flow_on_one_gpu():
    """launched the way the driver launches N > 1: torch.distributed.run around bench.py"""
    env = dict(os.environ, LIW_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1", "--warmup", "0", "--distinct", "4"],
                       capture_output=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check_two_ranks(last_json(r.stdout))


def test_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself and still print ONE line"""
    env = dict(os.environ, LIW_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "1", "--warmup", "0", "--distinct", "4"],
                       capture_output=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    _check_two_ranks(last_json(r.stdout))


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode != 0 and b"--gpus 2" in r.stderr


def test_eight_rank_control_flow_on_one_gpu():
    """`bench.py --gpus 8` the way the driver's 8-GPU node will run it, eight processes sharing this box's GPU over gloo: the batch is tiled
    on the device (eight host-side concatenations of the default batch did not fit a node's RAM), every rank's peak RSS is
    in the line, the factor-sharded C4 section runs with world 8 and checks itself."""
    env = dict(os.environ, LIW_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--batch", "512", "--steps", "1", "--warmup", "0", "--distinct", "4",
                        "--sharded-windows", "32"],
                       capture_output=True, cwd=ROOT, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["value"] > 0 and d["parity_gate"]["passed"]
    rss = d["host_peak_rss_mb_per_rank"]["end_of_run"]
    assert len(rss) == 8 and all(0 < v < 6000 for v in rss), rss
    fs = d["factor_sharded"]
    assert "error" not in fs, fs
    assert fs["ranks"] == 8 and fs["process_group"]["world_size"] == 8 and fs["states_identical_across_ranks"]
    assert fs["oneshot_exchange"]["states_identical_across_ranks"] and fs["self_check"]["passed"]
