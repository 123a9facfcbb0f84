"""The native peer-write exchange across PROCESSES: two processes on the test box's one GPU, gloo rendezvous, every
rank's receive area mapped into its peer with hipIpc (`BatchSolver.p2p_attach_ipc`), `liw_batch_solve_sharded(exchange = NULL)` pushing,
signalling and waiting through those mappings.  Statements: (1) both ranks end with bit-identical states, equal to the all-gather variant's
(same images, same rank order), and a second `liw_batch_p2p_setup` on the same areas continues the exchange counter from the flag words;
(2) a peer that attaches and then leaves is an error code (LIW_EHIP, rank named by liw_batch_p2p_status) after the bounded wait — also
when the timeout falls into the LAST chunk of iterations — not a hung device.  What one GPU cannot show: visibility across xGMI."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "workers", "p2p_ipc_worker.py")


def _port():
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    p = so.getsockname()[1]
    so.close()
    return p


def _launch(tmp_path, scenario, world=2, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    out = str(tmp_path / "r")
    ps = [subprocess.Popen([sys.executable, WORKER, out, scenario], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in ps:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    return out, ps, logs


def test_two_processes_push_into_each_others_ipc_areas(tmp_path):
    out, ps, logs = _launch(tmp_path, "solve")
    for p, lg in zip(ps, logs):
        assert p.returncode == 0, lg
    res = [json.load(open(out + "_rank%d.json" % r)) for r in range(2)]
    xp = [np.load(out + "_p2p_rank%d.npy" % r) for r in range(2)]
    xo = [np.load(out + "_oneshot_rank%d.npy" % r) for r in range(2)]
    assert np.array_equal(xp[0], xp[1]), "ranks must hold identical bits (identical sums in rank order)"
    assert np.array_equal(xo[0], xo[1])
    assert np.array_equal(xp[0], xo[0]), "peer-write exchange == all-gather variant, bit for bit"
    assert res[0]["p2p_iterations"] == res[1]["p2p_iterations"] == res[0]["oneshot_iterations"]
    assert max(res[0]["p2p_iterations"]) > 4, "the solve must run past the first chunk of iterations"
    assert res[0]["second_setup_identical"] and res[1]["second_setup_identical"]


def test_a_peer_that_leaves_is_an_error_not_a_hang(tmp_path):
    out, ps, logs = _launch(tmp_path, "dead_peer")
    assert ps[0].returncode == 0, logs[0]
    r0 = json.load(open(out + "_rank0.json"))
    assert r0["error"] is not None and r0["error"][0] == -5, r0      # LIW_EHIP
    assert r0["timed_out_rank_plus_1"] == 2, r0
    assert r0["seconds"] < 60.0, r0
