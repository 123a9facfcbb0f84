"""Parity at the shapes bench.py measures: the benchmarked configuration itself goes through the oracle.

  * 4 421 windows of C2 (n = 30, L = 2 000; not a multiple of 64 or of 4) — bench.py's OWN launch shape since round 4: the lane-per-group
    laser kernel k_lin_laser_slab (needs >= 2 048 (slab, frame) pairs: >= 4 353 windows at n = 30), k_lin_imu_chain, k_lm_step_quad, forked
    role streams, the batch tiled on the device — per-iteration states, iteration counts and terminations of the first / last windows of the
    first and last slab and of jittered copies against the oracle;
  * 2 304 windows of C2 — the same step kernels behind the lane-per-BLOCK laser kernel k_lin_laser<true> (G = 8 groups per wave), which
    medium batches and 3-D scans still take;
  * C2- and C5-size batched marginalisation (Delta_H, Delta_g, prior J^T J) against the oracle at the same linearisation point;
  * n = 50 / L = 5 000 LM history;
  * a C4-size (n = 30, L = 20 000) 10-iteration factor-sharded solve: two rank objects on this one GPU driven in lock-step
    through the REAL sharded loop (pack, exchange, unpack, join, early exit) with both exchange variants, against the un-sharded
    solve and the oracle.
Reference: src/factor/solver.cpp:4-40 (marginalization_matrix), :50-169 (do_init_solve)."""
import threading

import numpy as np
import pytest

from parity_util import rel_inf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(liw, synth, pyoracle):
    prm = synth.office_params()
    return prm, pyoracle.Oracle(prm)


def test_bench_launch_shape_slab_chain_quad_per_iteration_parity(liw, synth, pyoracle, env):
    """The kernels behind bench.py's `value`: k_lin_laser_slab + k_lin_imu_chain + k_lm_step_quad at L = 2 000, on a batch whose last slab
    holds 5 windows and whose last quad wave holds one.  Windows {0, 63, 64, B-65, B-1} (first / last lane of slab 0, first lane of slab 1,
    last full slab, tail) + two more jittered copies: every LM iteration within 1e-6 of the oracle, equal iteration counts and
    terminations (solver.cpp:50-169, north_star's per-iteration bar)."""
    import ctypes as C
    import importlib
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    prm, orc = env
    B, n, L, K, nd = 4421, 30, 2000, 50, 8
    tw = bench.make_tiled(liw, synth, prm, B, n, L, seed0=20240, n_base=nd)
    bs = liw.BatchSolver(prm, tw.base, tile=tw.tile(), history_records=K + 1)
    bs.solve(liw.LIW_MODE_INIT, K)
    flags = C.c_int(0)
    assert bs.L.liw_batch_launch_paths(bs.h, C.byref(bs.b), bs._wsp(), C.byref(flags)) == 0
    assert flags.value == 3, flags.value          # large-batch format (chain + quad kernels) AND the lane-per-group laser kernel armed
    hist, summ, xg = bs.history(), bs.summaries(), bs.states()
    orc.set_max_iterations(K)
    sample = [0, 63, 64, B - 65, B - 1, nd + 3, B // 2 + 1]
    worst = 0.0
    for b in sample:
        w = pyoracle.Window(tw[b])
        orc.set_prior(None)
        orc.init_solve(w)
        so, its = orc.summary(), orc.iterations()
        assert summ[b]["iterations"] == so["iterations"] and summ[b]["termination"] == so["termination"], (b, summ[b], so)
        for it in range(so["iterations"] + 1):
            e = rel_inf(hist[it, b], its[it]["x"].reshape(n, 15))
            worst = max(worst, e)
            assert e <= 1e-6, (b, it, e)
        assert rel_inf(xg[b], w["states"].reshape(n, 15)) <= 1e-6
        assert abs(summ[b]["final_cost"] - so["final_cost"]) <= 1e-9 * so["final_cost"]
    orc.set_max_iterations(50)
    assert len({(s["iterations"], s["termination"], round(s["final_cost"], 6)) for s in summ[:nd]}) > 1
    print("bench launch shape (slab / chain / quad) per-iteration state error (max over %d windows): %.2e" % (len(sample), worst))


def test_bench_configuration_per_iteration_parity(liw, synth, pyoracle, env):
    """2 304 C2 windows (8 distinct seeds + jittered copies): the lane-per-block laser kernel k_lin_laser<true> (below the slab kernel's
    4 353-window threshold at n = 30) in front of k_lin_imu_chain / k_lm_step_quad."""
    import importlib
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    prm, orc = env
    B, n, L, K, nd = 2304, 30, 2000, 50, 8
    windows = bench.make_batch(liw, synth, prm, B, n, L, seed0=20240, n_base=nd)
    bs = liw.BatchSolver(prm, windows, history_records=K + 1)
    bs.solve(liw.LIW_MODE_INIT, K)
    hist, summ, xg = bs.history(), bs.summaries(), bs.states()
    orc.set_max_iterations(K)
    sample = list(range(nd)) + [nd + 3, B - 1]        # the generated windows + two jittered copies (their own oracle runs)
    worst = 0.0
    for b in sample:
        w = pyoracle.Window(windows[b])
        orc.set_prior(None)
        orc.init_solve(w)
        so, its = orc.summary(), orc.iterations()
        assert summ[b]["iterations"] == so["iterations"] and summ[b]["termination"] == so["termination"], (b, summ[b], so)
        for it in range(so["iterations"] + 1):
            e = rel_inf(hist[it, b], its[it]["x"].reshape(n, 15))
            worst = max(worst, e)
            assert e <= 1e-6, (b, it, e)
        assert rel_inf(xg[b], w["states"].reshape(n, 15)) <= 1e-6
        assert abs(summ[b]["final_cost"] - so["final_cost"]) <= 1e-9 * so["final_cost"]
    # distinct seeds really take distinct LM paths
    assert len({(s["iterations"], s["termination"], round(s["final_cost"], 6)) for s in summ[:nd]}) > 1
    print("bench-shape per-iteration state error (max over %d windows): %.2e" % (len(sample), worst))


def test_bench_launch_shape_marginalisation_chain_and_eigq_against_the_oracle(liw, synth, pyoracle, env):
    """VERDICT r5 missing 5 / next 1a: the marginalisation kernels behind bench.py's `value` — k_lin_laser<false> + k_lin_imu_chain +
    k_marg_schur_chain + k_marg_schur_eigq (batches above 256 windows) — meet the oracle at n = 30 / L = 2 000 on the 4 421-window launch
    shape, twice in a row (the second pass carries the prior the first one wrote).  Windows {0, 63, 64, B-65, B-1} + two jittered copies:
    Delta_H, Delta_g and what k_marg_schur_eigq writes as the NEW prior (solver.cpp:390-402): linearized_X, J^T J, J^T R (eigenvector signs
    and the order among equal eigenvalues are free, these are not), sqrt_H = the pose block of the prior Jacobian."""
    import importlib
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    prm, orc = env
    B, n, L, K, nd = 4421, 30, 2000, 8, 8
    tw = bench.make_tiled(liw, synth, prm, B, n, L, seed0=20240, n_base=nd)
    bs = liw.BatchSolver(prm, tw.base, tile=tw.tile())
    bs.solve(liw.LIW_MODE_INIT, K)
    xg = bs.states()
    mpg = bs.t["match_pose"].cpu().numpy().reshape(B, n, 12)
    got = []
    for _ in range(2):
        sH, dH, dg = bs.marginalize()
        got.append(dict(sH=sH.cpu().numpy().reshape(B, 6, 6), dH=dH.cpu().numpy().reshape(B, 15, 15), dg=dg.cpu().numpy().reshape(B, 15),
                        J=bs.t["prior_J"].cpu().numpy().reshape(B, 15, 15).copy(), R=bs.t["prior_R"].cpu().numpy().reshape(B, 15).copy(),
                        X=bs.t["prior_X"].cpu().numpy().reshape(B, 15).copy(), has=bs.t["has_prior"].cpu().numpy().copy()))
    assert np.array_equal(bs.states(), xg)                    # marginalisation moves no state
    worst = dict(dH=0.0, dg=0.0, JJ=0.0, JR=0.0)
    for b in [0, 63, 64, B - 65, B - 1, nd + 3, B // 2 + 1]:
        ref = bench.marg_reference(pyoracle, orc, tw[b], xg[b], mpg[b], 2)
        for p in range(2):
            g, o = got[p], ref[p]
            assert g["has"][b] == 1
            sc = np.abs(o["dH"]).max()
            eH = np.abs(g["dH"][b] - o["dH"]).max() / sc
            eg = np.abs(g["dg"][b] - o["dg"]).max() / o["g_scale"]
            assert np.array_equal(g["X"][b], xg[b, n - 1])                           # linearized_X = the newest frame's state (solver.cpp:385)
            assert np.array_equal(g["X"][b], o["X"])
            JJ, JJo = g["J"][b].T @ g["J"][b], o["J"].T @ o["J"]
            eJJ = np.abs(JJ - JJo).max() / sc
            JR, JRo = g["J"][b].T @ g["R"][b], o["J"].T @ o["R"]
            eJR = np.abs(JR - JRo).max() / o["g_scale"]
            assert np.array_equal(g["sH"][b], g["J"][b][:6, :6])                     # sqrt_H (solver.cpp:401)
            sHo = o["J"][:6, :6]
            assert np.abs(g["sH"][b].T @ g["sH"][b] - sHo.T @ sHo).max() / np.abs(sHo.T @ sHo).max() <= 1e-9, (b, p)
            for k, e in (("dH", eH), ("dg", eg), ("JJ", eJJ), ("JR", eJR)):
                worst[k] = max(worst[k], e)
            # Delta_H: relative to its own largest entry (BASELINE.md 3's 1e-10 on H, measured 1e-15); Delta_g, J^T R: relative to the round-off
            # scale of the gradient sums (marg_reference); the prior reproduces Delta_H above the 1e-8 eigenvalue floor
            assert eH <= 1e-12 and eg <= 1e-11 and eJJ <= 1e-11 and eJR <= 1e-10, (b, p, eH, eg, eJJ, eJR)
    print("bench launch shape marginalisation (chain + eigq), worst over 7 windows x 2 passes: Delta_H %.2e Delta_g %.2e (of its round-off scale) "
          "prior J^T J %.2e J^T R %.2e" % (worst["dH"], worst["dg"], worst["JJ"], worst["JR"]))


def test_ragged_slab_batch_per_iteration_parity(liw, synth, pyoracle, env):
    """VERDICT r5 missing 6 / next 1c: a RAGGED batch through the lane-per-group laser kernel — 4 421 windows cycling through 8 distinct
    ones whose L runs from ~950 to ~3 300 with per-frame groups from 0 (a quarter of the frames own no block at all) to > 500 blocks,
    so that the 64 lanes of a k_lin_laser_slab wave run out of blocks at different rows, whole lanes idle through a frame, and the packed
    rows would be 3.6x the data in batch order (k_laser_slab_order, round 6: each frame takes the windows in its own order by group
    length).  Per-iteration states against the oracle (solver.cpp:50-169).

    Frames without laser blocks hang on IMU / wheel / ground alone and five of the eight windows crawl into the 50-iteration cap, where the
    LM path is round-off chaotic: the ORACLE moves by 1e-4 when its IMU means are scaled by 1 + 1e-13 N(0,1), and the
    lane-per-block laser kernel leaves the oracle just as the lane-per-group kernel does (tools/ragged_diag.py, gpurun_out/r6_ragged_diag.log).
    So: every iteration within 1e-9 for the first 16 iterations (any kernel fault shows at iteration 1), and after that within
    max(1e-6, 3 x what that perturbation moves the oracle at the SAME iteration), never above 1e-3; iteration counts and terminations must be
    equal wherever the oracle's own end state is determined to 1e-7."""
    import ctypes as C
    import importlib
    import sys
    import os
    from parity_util import init_solve_sensitivity
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    prm, orc = env
    B, n, L, K, nd = 4421, 30, 2000, 50, 8
    tw = bench.make_tiled(liw, synth, prm, B, n, L, seed0=30240, n_base=nd, ragged=True)
    counts = np.stack([np.bincount(np.asarray(w["laser_frame"]), minlength=n) for w in tw.base])
    assert counts.sum() == nd * L and counts.sum(1).min() < 0.5 * L and counts.sum(1).max() > 1.3 * L
    assert (counts[:, 1:] == 0).sum() >= nd and counts.max() > 200
    pad = counts.max(0).sum() * nd / counts.sum()
    assert pad > 1.5, pad                                    # ragged: in batch order a (slab, frame) wave would pad 3.6 rows per row of data
    bs = liw.BatchSolver(prm, tw.base, tile=tw.tile(), history_records=K + 1)
    bs.solve(liw.LIW_MODE_INIT, K)
    lp = bs.launch_paths()
    assert lp["flags"] == 3, lp
    # round 6: the lanes of a (slab, frame) wave are the windows in the frame's own order by group length -> hardly any padding
    assert lp["blocks"] == bs.Ltot and 1.0 <= lp["padding_ratio"] < 1.25, lp
    pad = lp["padding_ratio"]
    hist, summ, xg = bs.history(), bs.summaries(), bs.states()
    orc.set_max_iterations(K)
    worst_early, worst_ratio, determined = 0.0, 0.0, 0
    for b in list(range(nd)) + [63, 64, B - 65, B - 1, nd + 3]:
        w = pyoracle.Window(tw[b])
        orc.set_prior(None)
        orc.init_solve(w)
        so, its = orc.summary(), orc.iterations()
        sens = init_solve_sensitivity(pyoracle, orc, tw[b], its)
        if sens[-1] <= 1e-7:
            determined += 1
            assert summ[b]["iterations"] == so["iterations"] and summ[b]["termination"] == so["termination"], (b, summ[b], so)
            assert rel_inf(xg[b], w["states"].reshape(n, 15)) <= 1e-6
        for it in range(min(so["iterations"], summ[b]["iterations"]) + 1):
            e = rel_inf(hist[it, b], its[it]["x"].reshape(n, 15))
            if it <= 16:
                worst_early = max(worst_early, e)
                assert e <= 1e-9, (b, it, e)
            bar = min(max(1e-6, 3.0 * sens[it]), 1e-3)
            worst_ratio = max(worst_ratio, e / bar)
            assert e <= bar, (b, it, e, sens[it])
    orc.set_max_iterations(50)
    assert determined >= 3
    print("ragged slab batch (padding ratio %.2f): per-iteration state error over 13 windows: first 16 iterations %.2e; worst fraction of the "
          "per-iteration bar max(1e-6, 3 x oracle sensitivity) %.2f; %d windows with a determined end state" % (pad, worst_early, worst_ratio, determined))


@pytest.mark.parametrize("n,L,seed", [(30, 2000, 20240), (50, 5000, 77)])
def test_batch_marginalization_at_bench_sizes(liw, synth, pyoracle, env, n, L, seed):
    """C2- / C5-size marginalisation (Hmm = 435^2 / 735^2 in the reference's dense form) after a short init solve, both sides
    at the same linearisation point; with and without a prior carried in from a previous marginalisation."""
    prm, orc = env
    ws = [synth.make_window(orc, prm, seed=seed + k, n=n, L=L) for k in range(2)]
    bs = liw.BatchSolver(prm, ws)
    bs.solve(liw.LIW_MODE_INIT, 6)
    xg = bs.states()
    mpg = bs.t["match_pose"].cpu().numpy().reshape(2, n, 12)
    for with_prior in (False, True):
        sH, dH, dg = bs.marginalize()             # second pass: the prior written by the first one is in the window
        dH, dg = dH.cpu().numpy().reshape(-1, 15, 15), dg.cpu().numpy()
        pJ = bs.t["prior_J"].cpu().numpy().reshape(-1, 15, 15)
        for k in range(2):
            w = pyoracle.Window(ws[k])
            w["states"][:] = xg[k].reshape(w["states"].shape)
            w["match_pose"][:] = mpg[k].reshape(w["match_pose"].shape)
            orc.set_prior(None)
            if with_prior:
                orc.marginalization(w)            # first pass on the oracle side
            orc.marginalization(w)
            m = orc.marg_pieces()
            eH, eg = rel_inf(dH[k], m["Delta_H"]), rel_inf(dg[k], m["Delta_g"])
            assert eH <= 1e-10 and eg <= 1e-9, (n, with_prior, k, eH, eg)      # measured: 6e-16 / 1e-11
            Xo, Jo, Ro = orc.get_prior()
            assert rel_inf(pJ[k].T @ pJ[k], Jo.T @ Jo) <= 1e-9      # eigenvector signs are free: compare J^T J
            print("marg n=%d prior=%s window %d: Delta_H %.2e Delta_g %.2e" % (n, with_prior, k, eH, eg))


def test_c5_window_lm_history(liw, synth, pyoracle, env):
    prm, orc = env
    n, L, K = 50, 5000, 25
    d = synth.make_window(orc, prm, seed=505, n=n, L=L)
    bs = liw.BatchSolver(prm, [d], history_records=K + 1)
    bs.solve(liw.LIW_MODE_INIT, K)
    hist, s = bs.history(), bs.summaries()[0]
    w = pyoracle.Window(d)
    orc.set_prior(None)
    orc.set_max_iterations(K)
    orc.init_solve(w)
    so, its = orc.summary(), orc.iterations()
    assert s["iterations"] == so["iterations"] and s["termination"] == so["termination"]
    for it in range(so["iterations"] + 1):
        assert rel_inf(hist[it, 0], its[it]["x"].reshape(n, 15)) <= 1e-6, it
    orc.set_max_iterations(50)


@pytest.mark.parametrize("exchange", ["allreduce", "oneshot"])
def test_c4_factor_sharded_solve_lockstep(liw, synth, pyoracle, env, exchange):
    """C4 (n = 30, L = 20 000), 10 LM iterations, laser blocks split over two rank objects on this GPU.  The ranks run the real
    sharded loop of BatchSolver.solve in two threads; LockstepComm stands in for RCCL only."""
    import torch
    prm, orc = env
    n, L, K = 30, 20000, 10
    ws = [synth.make_window(orc, prm, seed=4242 + k, n=n, L=L) for k in range(2)]
    ref = liw.BatchSolver(prm, ws)
    ref.solve(liw.LIW_MODE_INIT, K)
    comms = liw.batch.LockstepComm.make(2)
    ranks = [liw.BatchSolver(prm, ws, rank=r, world=2, exchange=exchange, comm=comms[r]) for r in range(2)]
    assert sum(rk.Ltot for rk in ranks) == ref.Ltot
    assert ranks[0].exchange_bytes(liw.LIW_MODE_INIT) == 8 * (2 * n * 45 + 1)        # compact record: 45 of 128 slots
    errs = []

    def drive(rk):
        try:
            rk.solve(liw.LIW_MODE_INIT, K)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
            comms[0].sh["bar"].abort()
    th = [threading.Thread(target=drive, args=(rk,)) for rk in ranks]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    torch.cuda.synchronize()
    a, b, r = ranks[0].states(), ranks[1].states(), ref.states()
    assert np.array_equal(a, b)                                   # ranks stay bit-identical
    assert rel_inf(a, r) <= 1e-9                                  # sharded sum == un-sharded sum up to summation order
    assert [s["iterations"] for s in ranks[0].summaries()] == [s["iterations"] for s in ref.summaries()]
    orc.set_max_iterations(K)
    for k in range(2):
        w = pyoracle.Window(ws[k])
        orc.set_prior(None)
        orc.init_solve(w)
        assert ranks[0].summaries()[k]["iterations"] == orc.summary()["iterations"]
        assert rel_inf(a[k], w["states"].reshape(n, 15)) <= 1e-6
    orc.set_max_iterations(50)


def test_exchange_variants_bit_identical_and_early_exit(liw, synth, pyoracle, env):
    """Both exchange variants give bit-identical states (rank-order sums of two images commute), the compact record round-trips
    the 128-slot record exactly (one rank, forced exchange == plain solve, bit for bit), and the sharded loop stops early when
    every window has terminated (same result as the full-length loop)."""
    import torch
    prm, orc = env
    ws = [synth.make_window(orc, prm, seed=90 + k, n=8, L=150 + 7 * k) for k in range(3)]
    plain = liw.BatchSolver(prm, ws)
    plain.solve(liw.LIW_MODE_INIT, 50)
    forced = liw.BatchSolver(prm, ws, force_exchange=True)
    forced.solve(liw.LIW_MODE_INIT, 50)
    torch.cuda.synchronize()
    assert np.array_equal(plain.states(), forced.states())
    assert [s["iterations"] for s in plain.summaries()] == [s["iterations"] for s in forced.summaries()]
    out = {}
    for xch in ("allreduce", "oneshot", "auto"):
        comms = liw.batch.LockstepComm.make(2)
        ranks = [liw.BatchSolver(prm, ws, rank=r, world=2, exchange=xch, comm=comms[r]) for r in range(2)]
        th = [threading.Thread(target=lambda rk=rk: rk.solve(liw.LIW_MODE_INIT, 50)) for rk in ranks]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        torch.cuda.synchronize()
        assert np.array_equal(ranks[0].states(), ranks[1].states())
        out[xch] = (ranks[0].states(), [s["iterations"] for s in ranks[0].summaries()])
        if xch == "auto":                               # timed both once, every rank kept the same (faster) transport
            assert ranks[0].exchange in ("allreduce", "oneshot") and ranks[0].exchange == ranks[1].exchange
            assert ranks[0].exchange_pick["picked"] == ranks[0].exchange and ranks[0].exchange_pick == ranks[1].exchange_pick
    assert np.array_equal(out["allreduce"][0], out["oneshot"][0]) and np.array_equal(out["allreduce"][0], out["auto"][0])
    assert out["allreduce"][1] == [s["iterations"] for s in plain.summaries()]
    assert rel_inf(out["allreduce"][0], plain.states()) <= 1e-9
