"""The native one-shot exchange of the factor-sharded mode: every rank writes its packed laser record
straight into every peer's receive area, raises a flag there, waits for the flags of its own area and adds the images in rank order —
flag-synchronised kernels instead of an all-gather collective.  On this 1-GPU box: one rank onto itself, two rank objects of one
process on two HIP streams (bit-identical to the all-gather variant, which adds the same images in the same order), and the bounded
wait (a peer that never arrives is an error code, not a hung device).  Cross-device visibility has not run across xGMI."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(liw, synth, pyoracle):
    prm = synth.office_params()
    return prm, pyoracle.Oracle(prm)


def test_one_rank_pushing_onto_itself_is_the_plain_solve(liw, synth, env):
    import torch
    prm, orc = env
    ws = [synth.make_window(orc, prm, seed=190 + k, n=8, L=150 + 7 * k) for k in range(3)]
    plain = liw.BatchSolver(prm, ws)
    plain.solve(liw.LIW_MODE_INIT, 50)
    one = liw.BatchSolver(prm, ws, force_exchange=True, exchange="p2p")
    liw.BatchSolver.p2p_attach_local([one], liw.LIW_MODE_INIT)
    one.solve(liw.LIW_MODE_INIT, 50)
    torch.cuda.synchronize()
    assert np.array_equal(plain.states(), one.states())
    assert [s["iterations"] for s in plain.summaries()] == [s["iterations"] for s in one.summaries()]


def test_two_ranks_on_two_streams_equal_the_all_gather_variant(liw, synth, env):
    import torch
    prm, orc = env
    n, K = 10, 20
    ws = [synth.make_window(orc, prm, seed=4300 + k, n=n, L=900 + 100 * k) for k in range(5)]
    ref = {}
    comms = liw.batch.LockstepComm.make(2)
    gat = [liw.BatchSolver(prm, ws, rank=r, world=2, exchange="oneshot", comm=comms[r]) for r in range(2)]
    th = [threading.Thread(target=lambda rk=rk: rk.solve(liw.LIW_MODE_INIT, K)) for rk in gat]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    torch.cuda.synchronize()
    ref = gat[0].states()
    ranks = [liw.BatchSolver(prm, ws, rank=r, world=2, exchange="p2p") for r in range(2)]
    liw.BatchSolver.p2p_attach_local(ranks, liw.LIW_MODE_INIT)
    streams = [torch.cuda.Stream() for _ in range(2)]     # a rank waits (in a kernel) for its peer's push: the two must not share a stream
    errs = []

    def drive(rk, st):
        try:
            with torch.cuda.stream(st):
                rk.solve(liw.LIW_MODE_INIT, K)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=drive, args=(rk, st)) for rk, st in zip(ranks, streams)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    torch.cuda.synchronize()
    assert not errs, errs
    a, b = ranks[0].states(), ranks[1].states()
    assert np.array_equal(a, b)            # ranks bit-identical
    assert np.array_equal(a, ref)          # and bit-identical to the all-gather variant: same images, same order of the sum
    assert [s["iterations"] for s in ranks[0].summaries()] == [s["iterations"] for s in gat[0].summaries()]


def test_a_peer_that_never_arrives_is_an_error_not_a_hang(liw, synth, env):
    prm, orc = env
    ws = [synth.make_window(orc, prm, seed=77, n=4, L=40)]
    ranks = [liw.BatchSolver(prm, ws, rank=r, world=2, exchange="p2p") for r in range(2)]
    liw.BatchSolver.p2p_attach_local(ranks, liw.LIW_MODE_INIT)
    with pytest.raises(liw.LiwError):
        ranks[0].solve(liw.LIW_MODE_INIT, 8)      # rank 1 never runs: its flag stays 0, the bounded wait gives up
