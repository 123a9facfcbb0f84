"""Host-side helpers of bench.py's round-6 legs (no GPU): the ragged window shapes, the 2-frame sub-windows of the batched TRACK leg, and the
oracle-side marginalisation reference with its round-off scales (the checker of the parity gate — the oracle checked against itself and numpy)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_ragged_shapes_keep_the_c2_totals(synth):
    bench = importlib.import_module("bench")
    for nb, n, L in ((8, 30, 2000), (64, 30, 2000), (5, 12, 300)):
        sh = bench.ragged_shapes(synth, n, L, nb, seed=123 + nb)
        st = np.stack(sh)
        assert st.shape == (nb, n) and st.sum() == nb * L and (st >= 0).all()
        assert (st[:, 0] == 0).all()                               # init topology: frame 0 owns no block (solver.cpp:93-106)
        assert st.sum(1).min() < 0.7 * L < 1.3 * L < st.sum(1).max()       # per-window L drawn wide
        assert (st[:, 1:] == 0).any() and st.max() > 3 * L / (n - 1)       # empty frames next to frames with several times the mean
    rng = np.random.default_rng(0)
    c = synth.ragged_frame_counts(rng, 6, 50, p_empty=1.0)                 # every frame "empty": the last one takes all (never an empty window)
    assert c.sum() == 50 and c[0] == 0 and c[5] == 50


def test_sub_window_is_the_two_frame_tracking_window(liw, synth, pyoracle):
    bench = importlib.import_module("bench")
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    d = synth.make_window(orc, prm, seed=31, n=5, frame_counts=[0, 7, 0, 11, 5])
    for lo in range(4):
        w = bench.sub_window(d, lo)
        assert w["n"] == 2 and np.asarray(w["states"]).shape == (2, 15) and np.asarray(w["match_pose"]).shape == (2, 12)
        assert np.array_equal(w["states"], np.asarray(d["states"])[lo:lo + 2]) and np.array_equal(w["has_match"], np.asarray(d["has_match"])[lo:lo + 2])
        assert np.asarray(w["imu_X"]).shape == (1, 15) and np.array_equal(w["imu_X"][0], np.asarray(d["imu_X"])[lo])
        assert np.array_equal(w["wheel_T"][0], np.asarray(d["wheel_T"])[lo])
        cnt = np.bincount(np.asarray(w["laser_frame"]), minlength=2)
        assert list(cnt) == [[0, 7, 0, 11, 5][lo], [0, 7, 0, 11, 5][lo + 1]]
        assert np.array_equal(w["laser_pts"], np.asarray(d["laser_pts"])[(np.asarray(d["laser_frame"]) >= lo) & (np.asarray(d["laser_frame"]) < lo + 2)])
        # the reference pose of every block stays the constant laser_match pose of the long window (p1, q1)
        assert np.array_equal(np.asarray(w["match_pose"])[:, 0:6], np.asarray(d["match_pose"])[lo:lo + 2, 0:6])
    # the oracle solves it as a tracking window: laser blocks of the newest frame only + the prior on the older one
    w = bench.sub_window(d, 2)
    wo = pyoracle.Window(w)
    orc.set_prior(None)
    orc.solve(wo)
    assert orc.summary()["iterations"] >= 1


def test_marg_reference_scales_and_prior_argument(liw, synth, pyoracle):
    bench = importlib.import_module("bench")
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    d = synth.make_window(orc, prm, seed=77, n=4, L=60)
    x, mp = np.asarray(d["states"]).reshape(4, 15), np.asarray(d["match_pose"]).reshape(4, 12)
    r2 = bench.marg_reference(pyoracle, orc, d, x, mp, 2)
    m = orc.marg_pieces()
    # pass 2 ran on the prior of pass 1: 15 more rows in J than without a prior
    orc.set_prior(None)
    orc.marginalization(pyoracle.Window(d))
    assert m["J"].shape[0] == orc.marg_pieces()["J"].shape[0] + 15
    # the Schur complement of the dense H is Delta_H / Delta_g (numpy), the new prior reproduces Delta_H above the 1e-8 floor
    H, g = orc.marg_pieces()["H"], orc.marg_pieces()["g"]
    N = H.shape[0]
    W = np.linalg.solve(H[:N - 15, :N - 15], H[N - 15:, :N - 15].T).T
    dH, dg = H[N - 15:, N - 15:] - W @ H[:N - 15, N - 15:], g[N - 15:] - W @ g[:N - 15]
    assert np.abs(dH - r2[0]["dH"]).max() <= 1e-9 * np.abs(dH).max() and np.abs(dg - r2[0]["dg"]).max() <= 1e-9 * r2[0]["g_scale"]
    assert np.abs(r2[0]["J"].T @ r2[0]["J"] - 0.5 * (dH + dH.T)).max() <= 1e-9 * np.abs(dH).max()
    # scales: the gradient's round-off scale dominates the gradient itself; H_rr bounds Delta_H
    assert r2[0]["g_scale"] >= np.abs(r2[0]["dg"]).max() and r2[0]["H_rr_scale"] >= 0.999 * np.abs(r2[0]["dH"]).max()
    # an explicit prior argument = the prior the first pass left
    r1 = bench.marg_reference(pyoracle, orc, d, x, mp, 1, prior=(r2[0]["X"], r2[0]["J"], r2[0]["R"]))
    assert np.array_equal(r1[0]["dH"], r2[1]["dH"]) and np.array_equal(r1[0]["dg"], r2[1]["dg"])
