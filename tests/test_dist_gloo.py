"""world_size-2 CPU (gloo) test of the factor-parallel path: laser blocks of a window are sharded across ranks
(`shard_laser`), each rank linearises its shard (here with the oracle standing in for the HIP kernel — there is
no GPU in this container), the partial normal equations are sum-all-reduced with the same helper the GPU path
uses, and every rank must end up with the full H, g, cost."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    liw = importlib.import_module("2dliw-slam_amd")
    synth = importlib.import_module("2dliw-slam_amd.synth")
    from oracle import pyoracle
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    full = synth.make_window(orc, prm, seed=31, n=5, L=41)
    mine = liw.shard_laser(full, rank, world)
    # laser-only partial sums of this rank + the small factors evaluated redundantly on every rank:
    # H_shard(all factors) - H(no laser) is this rank's laser contribution
    none = dict(full)
    none["laser_frame"] = np.zeros(0, dtype=np.int32)
    none["laser_pts"] = np.zeros((0, 12))
    Hs, gs, cs = orc.linearize(pyoracle.Window(mine), 0)
    H0, g0, c0 = orc.linearize(pyoracle.Window(none), 0)
    part = torch.from_numpy(np.concatenate([(Hs - H0).reshape(-1), gs - g0, [cs - c0]]))
    liw.batch.allreduce_sum_(part)
    N = H0.shape[0]
    H = part[:N * N].numpy().reshape(N, N) + H0
    g = part[N * N:N * N + N].numpy() + g0
    c = float(part[-1]) + c0
    Hf, gf, cf = orc.linearize(pyoracle.Window(full), 0)
    ok = (np.abs(H - Hf).max() <= 1e-9 * np.abs(Hf).max() and np.abs(g - gf).max() <= 1e-9 * np.abs(gf).max() and abs(c - cf) <= 1e-9 * cf)
    nshard = int(np.asarray(mine["laser_frame"]).shape[0])
    gathered = [None] * world
    dist.all_gather_object(gathered, nshard)
    q.put((rank, bool(ok), gathered))
    dist.destroy_process_group()


def test_factor_sharded_allreduce_world2(pyoracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert sum(res[0][2]) == 41          # the shards partition the laser blocks


def test_shard_laser_partitions(liw, synth, pyoracle):
    prm = synth.office_params()
    d = synth.make_window(pyoracle.Oracle(prm), prm, seed=2, n=4, L=23)
    for world in (1, 2, 3, 8):
        parts = [liw.shard_laser(d, r, world) for r in range(world)]
        assert sum(len(p["laser_frame"]) for p in parts) == 23
        cat = np.concatenate([p["laser_pts"] for p in parts])
        assert np.array_equal(cat, d["laser_pts"])
        for p in parts:
            assert np.all(np.diff(p["laser_frame"]) >= 0)
