"""Size-independent properties of the oracle's geometry / solver pieces."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def env(synth, pyoracle):
    prm = synth.office_params()
    return prm, pyoracle.Oracle(prm)


def test_exp_log_roundtrip_all_branches(env):
    prm, orc = env
    rng = np.random.default_rng(0)
    for scale in (1e-9, 1e-3, 0.5, 2.08, 3.0, 3.14):     # small angle, trace>0, trace<=0 (arg-max diagonal) branches
        for _ in range(20):
            a = rng.normal(size=3)
            a = a / np.linalg.norm(a) * scale
            R = orc.exp_so3(a)
            assert np.abs(R @ R.T - np.eye(3)).max() < 1e-13
            assert np.abs(orc.log_SO3(R) - a).max() < 1e-9 * max(1.0, scale) + 1e-7 * (scale > 3.1)
    assert np.abs(orc.exp_so3(np.zeros(3)) - np.eye(3)).max() == 0.0


def test_so3_plus_wraps_past_pi(env):
    prm, orc = env
    out, J = orc.so3_plus(np.array([0.1, 0.2, 0.3]), np.array([0.01, 0.0, 0.0]))
    assert np.allclose(out, [0.11, 0.2, 0.3]) and np.allclose(J, np.eye(3))
    x = np.array([0.0, 0.0, 3.5])     # |x| > pi : wraps to the equivalent rotation with |.| <= pi
    out, J = orc.so3_plus(x, np.zeros(3))
    assert abs(np.linalg.norm(out) - (2 * np.pi - 3.5)) < 1e-12
    assert np.abs(orc.exp_so3(out) - orc.exp_so3(x)).max() < 1e-12
    assert not np.allclose(J, np.eye(3))


def test_normal_equations_symmetric_psd_and_schur_identity(env, synth, pyoracle):
    prm, orc = env
    d = synth.make_window(orc, prm, seed=5, n=6, L=60)
    w = pyoracle.Window(d)
    H, g, c = orc.linearize(w, 0)
    assert np.abs(H - H.T).max() <= 1e-9 * np.abs(H).max()
    assert np.linalg.eigvalsh(H).min() > -1e-6 * np.abs(H).max()
    orc.set_prior(None)
    orc.marginalization(w)
    m = orc.marg_pieces()
    # Schur complement of the dense H onto the last 15 states == the oracle's Delta_H / Delta_g
    Hm, N = m["H"], m["H"].shape[0]
    mm = slice(0, N - 15)
    rr = slice(N - 15, N)
    dH = Hm[rr, rr] - Hm[rr, mm] @ np.linalg.solve(Hm[mm, mm], Hm[mm, rr])
    dg = m["g"][rr] - Hm[rr, mm] @ np.linalg.solve(Hm[mm, mm], m["g"][mm])
    assert np.abs(dH - m["Delta_H"]).max() <= 1e-7 * np.abs(dH).max()
    assert np.abs(dg - m["Delta_g"]).max() <= 1e-7 * max(1.0, np.abs(dg).max())
    # prior = eigen square root of Delta_H above the 1e-8 floor (solver.cpp:390-402)
    X, J, R = orc.get_prior()
    assert np.abs(J.T @ J - m["Delta_H"]).max() <= 1e-7 * np.abs(m["Delta_H"]).max()
    assert np.allclose(X, w["states"].reshape(-1, 15)[-1])


def test_lm_decreases_cost_and_matches_iteration_log(env, synth, pyoracle):
    prm, orc = env
    d = synth.make_window(orc, prm, seed=8, n=5, L=40)
    w = pyoracle.Window(d)
    orc.set_prior(None)
    orc.init_solve(w)
    s, its = orc.summary(), orc.iterations()
    assert len(its) == s["iterations"] + 1
    costs = [it["cost"] for it in its]
    assert all(b <= a for a, b in zip(costs, costs[1:]))
    assert s["final_cost"] < 0.05 * s["initial_cost"]
    err0 = np.abs(d["states"] - d["truth_states"])[:, 0:3].max()
    err1 = np.abs(w["states"].reshape(-1, 15) - d["truth_states"])[:, 0:3].max()
    assert err1 < err0


def test_tracking_solve_keeps_old_poses_constant(env, synth, pyoracle):
    prm, orc = env
    d = synth.make_window(orc, prm, seed=9, n=3, L=30, laser_on_frame0=False)
    w = pyoracle.Window(d)
    orc.set_prior(None)
    before = w["states"].reshape(-1, 15).copy()
    orc.solve(w)
    after = w["states"].reshape(-1, 15)
    assert np.array_equal(after[:2, 0:6], before[:2, 0:6])      # p,q of frames 0..n-2 are constant (solver.cpp:787-794)
    assert not np.array_equal(after[2, 0:6], before[2, 0:6])


def test_empty_and_ragged_windows(env, synth, pyoracle):
    prm, orc = env
    # no laser blocks at all; single frame; frame with a match but zero lines
    for n, L in ((3, 0), (1, 0), (2, 1)):
        d = synth.make_window(orc, prm, seed=4, n=n, L=L)
        w = pyoracle.Window(d)
        orc.set_prior(None)
        orc.init_solve(w)
        assert np.isfinite(w["states"]).all()
        H, g, c = orc.linearize(w, 0)
        assert np.isfinite(H).all() and np.isfinite(c)
