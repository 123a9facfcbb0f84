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


def _run(target, world, extra=()):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    return sorted(res, key=lambda r: r[0])


def test_factor_sharded_allreduce_world2(pyoracle):
    res = _run(_worker, 2)
    assert all(r[1] for r in res), res
    assert sum(res[0][2]) == 41          # the shards partition the laser blocks


def _worker_rank_order(rank, world, port, q):
    """the two transports of the factor-sharded exchange (2dliw-slam_amd/batch.py) at world sizes the GPU box cannot show: every rank's
    laser contribution to (H, g, cost) — here from the oracle — summed (a) by all-reduce, (b) by all-gather + sum in RANK ORDER"""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    liw = importlib.import_module("2dliw-slam_amd")
    synth = importlib.import_module("2dliw-slam_amd.synth")
    from oracle import pyoracle
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    full = synth.make_window(orc, prm, seed=57, n=6, L=83)
    none = dict(full)
    none["laser_frame"] = np.zeros(0, dtype=np.int32)
    none["laser_pts"] = np.zeros((0, 12))
    H0, g0, c0 = orc.linearize(pyoracle.Window(none), 0)

    def laser_part(r):
        Hs, gs, cs = orc.linearize(pyoracle.Window(liw.shard_laser(full, r, world)), 0)
        return np.concatenate([(Hs - H0).reshape(-1), gs - g0, [cs - c0]])
    mine = torch.from_numpy(laser_part(rank))
    comm = liw.batch.TorchComm()
    red = comm.all_reduce_sum_(mine.clone())
    allb = torch.zeros((world, mine.numel()), dtype=torch.float64)
    comm.all_gather_(allb, mine)
    tot = allb[0].clone()
    for r in range(1, world):                       # what liw_batch_exchange_unpack does with `world` images: sum in rank order
        tot += allb[r]
    # the same sum formed by ONE process from the same shards in the same order (a world-1 computation): bit-identical
    serial = laser_part(0)
    for r in range(1, world):
        serial = serial + laser_part(r)
    Hf, gf, cf = orc.linearize(pyoracle.Window(full), 0)
    N = H0.shape[0]
    Hsum = tot[:N * N].numpy().reshape(N, N) + H0
    close = bool(np.abs(Hsum - Hf).max() <= 1e-9 * np.abs(Hf).max() and np.abs(red.numpy() - tot.numpy()).max() <= 1e-9 * np.abs(tot.numpy()).max())
    q.put((rank, tot.numpy().tobytes(), red.numpy().tobytes(), bool(np.array_equal(tot.numpy(), serial)), close,
           int(np.asarray(liw.shard_laser(full, rank, world)["laser_frame"]).shape[0])))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_rank_order_sum_is_identical_on_every_rank_and_equals_the_serial_sum(pyoracle, world):
    """SURVEY 8e at the world sizes BASELINE names (4, 8): the one-shot exchange (all-gather + rank-order sum) gives every rank the SAME
    bits — and the bits a single process gets from the same shards — so states and `done` flags cannot part ways; the all-reduce gives
    every rank identical bits too (its sum may differ from the rank-order one in the last place)."""
    res = _run(_worker_rank_order, world)
    assert len({r[1] for r in res}) == 1, "rank-order sums differ between ranks"
    assert len({r[2] for r in res}) == 1, "all-reduce results differ between ranks"
    assert all(r[3] for r in res), "rank-order sum != serial sum of the same shards"
    assert all(r[4] for r in res)
    assert sum(r[5] for r in res) == 83


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
