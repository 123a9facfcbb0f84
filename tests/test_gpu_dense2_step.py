"""k_lm_step_dense2 (round 5): every two-frame window — the window the reference's tracking loop solves each laser frame
(trajectory.cpp:525-560, solver.cpp:631-820) — is stepped as ONE dense 30 x 30 system instead of two chained frame eliminations.
Same normal equations, same Ceres logic, different elimination order: per-iteration states against the chained one-wave kernel
(LIW_STEP_VARIANT=0) and against the oracle, for the init topology (arrow folded into the coupling tile), the tracking topology with a
carried prior, fast mode (the older frame's biases constant too: 12 inert pivots), rotation vectors beyond pi (so3 Plus Jacobian) and a
batch of 300 such windows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-12, np.abs(np.asarray(b)).max()))


def _solve(liw, prm, wins, mode, cap, prior, variant, monkeypatch):
    import torch
    if variant is None:
        monkeypatch.delenv("LIW_STEP_VARIANT", raising=False)
    else:
        monkeypatch.setenv("LIW_STEP_VARIANT", variant)
    bs = liw.BatchSolver(prm, wins, history_records=cap + 2)
    if prior is not None:
        X, J, R = prior
        B = len(wins)
        bs.t["prior_X"].copy_(torch.from_numpy(np.tile(X, B)))
        bs.t["prior_J"].copy_(torch.from_numpy(np.tile(J.reshape(-1), B)))
        bs.t["prior_R"].copy_(torch.from_numpy(np.tile(R, B)))
        bs.t["has_prior"].fill_(1)
    bs.solve(mode, cap)
    torch.cuda.synchronize()
    out = (bs.history().copy(), bs.summaries(), bs.states().copy())
    bs.close()
    return out


@pytest.mark.parametrize("case", ["init", "track_prior", "track_fast", "init_beyond_pi"])
def test_dense_two_frame_step_matches_chained_kernel_and_oracle(liw, synth, pyoracle, monkeypatch, case):
    prm = dict(synth.office_params())
    if case == "track_fast":
        prm["fast_mode"] = True
    orc = pyoracle.Oracle(prm)
    d = synth.make_window(orc, prm, seed=515 + len(case), n=2, L=60)
    if case == "init_beyond_pi":
        st = np.array(d["states"], copy=True)
        q = st[1, 3:6]
        a = np.linalg.norm(q)
        st[1, 3:6] = q / a * (a - 2 * np.pi)                 # same rotation, |q| > pi
        assert np.linalg.norm(st[1, 3:6]) > np.pi
        d["states"] = st
        mp = np.array(d["match_pose"], copy=True)
        mp[:, 6:12] = st[:, 0:6]
        d["match_pose"] = mp
    track = case.startswith("track")
    mode = liw.LIW_MODE_TRACK if track else liw.LIW_MODE_INIT
    prior = None
    if case == "track_prior":
        w0 = pyoracle.Window(d)
        orc.set_prior(None)
        orc.set_max_iterations(3)
        orc.init_solve(w0)
        orc.marginalization(w0)
        prior = orc.get_prior()
    cap = 10 if case == "track_fast" else 12
    hd, sd, xd = _solve(liw, prm, [d], mode, cap, prior, None, monkeypatch)          # default: the dense kernel
    hc, sc, xc = _solve(liw, prm, [d], mode, cap, prior, "0", monkeypatch)           # the chained one-wave kernel
    assert (sd[0]["iterations"], sd[0]["termination"], sd[0]["successful"]) == (sc[0]["iterations"], sc[0]["termination"], sc[0]["successful"])
    for it in range(sd[0]["iterations"] + 1):
        assert rel(hd[it, 0], hc[it, 0]) <= 1e-9, (case, it)
    o2 = pyoracle.Oracle(prm)
    o2.set_prior(prior)
    o2.set_max_iterations(cap)
    wo = pyoracle.Window(d)
    (o2.solve if track else o2.init_solve)(wo)
    so, its = o2.summary(), o2.iterations()
    assert (sd[0]["iterations"], sd[0]["termination"]) == (so["iterations"], so["termination"]), (case, sd[0], so)
    for it in range(so["iterations"] + 1):
        assert rel(hd[it, 0], its[it]["x"].reshape(2, 15)) <= 1e-6, (case, it)
    assert rel(xd[0], wo["states"].reshape(2, 15)) <= 1e-6


def test_dense_two_frame_step_on_a_batch(liw, synth, pyoracle, monkeypatch):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    base = [synth.make_window(orc, prm, seed=900 + k, n=2, L=20 + 9 * k) for k in range(5)]
    wins = [base[b % 5] for b in range(300)]
    hd, sd, xd = _solve(liw, prm, wins, liw.LIW_MODE_INIT, 8, None, None, monkeypatch)
    hc, sc, xc = _solve(liw, prm, wins, liw.LIW_MODE_INIT, 8, None, "0", monkeypatch)
    assert [(s["iterations"], s["termination"]) for s in sd] == [(s["iterations"], s["termination"]) for s in sc]
    assert rel(xd, xc) <= 1e-9
    assert np.array_equal(xd[0], xd[295]) and np.array_equal(xd[4], xd[299])       # copies of a window take bit-identical paths
