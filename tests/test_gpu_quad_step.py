"""k_lm_step_quad (csrc/k_lm_quad.hip): four windows per wavefront, one 16-lane DPP row per window — the LM step of large INIT-topology
batches.  Forced here on small batches (LIW_STEP_VARIANT=3, read per launch) and checked window by window against the oracle's
per-iteration LM history (1e-6, north_star) and against the one-wave kernel it replaces; plus the windows it must hand over or stop:
a rotation vector outside the |theta| <= pi ball (stepped by k_lm_step in the same launch pair), an evaluation failure (Ceres FAILURE),
a ragged batch whose size is not a multiple of four."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()))


def oracle_history(pyoracle, orc, d, iters):
    wo = pyoracle.Window(d)
    orc.set_prior(None)
    orc.set_max_iterations(iters)
    orc.init_solve(wo)
    so, ho = orc.summary(), orc.iterations()
    orc.set_max_iterations(50)
    return wo, so, ho


# (a window without laser blocks crawls along the ground_factor_q cone and is round-off chaotic beyond ~35 iterations — DESIGN 6,
#  tools/quad_diag.py shows both step kernels leaving the oracle there — so those appear with the 20-iteration cap only)
@pytest.mark.parametrize("n,B,iters", [(1, 5, 20), (2, 6, 50), (3, 7, 20), (5, 6, 20), (7, 9, 20), (7, 10, 50), (8, 5, 20), (30, 5, 50)])
def test_quad_step_follows_the_oracle_iteration_by_iteration(liw, synth, pyoracle, monkeypatch, n, B, iters):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    base = [synth.make_window(orc, prm, seed=4100 + 17 * n + k, n=n, L=(0 if ((k == 3 and iters <= 20) or n == 1) else 20 * n + 37 * k)) for k in range(5)]
    wins = [base[b % 5] for b in range(B)]
    monkeypatch.setenv("LIW_STEP_VARIANT", "3")
    bs = liw.BatchSolver(prm, wins, history_records=iters + 1)
    bs.solve(liw.LIW_MODE_INIT, iters)
    got, summ, hist = bs.states(), bs.summaries(), bs.history()
    worst = 0.0
    for k in range(min(5, B)):
        wo, so, ho = oracle_history(pyoracle, orc, base[k], iters)
        for b in range(k, B, 5):
            assert (summ[b]["iterations"], summ[b]["termination"]) == (so["iterations"], so["termination"]), (n, k, b, summ[b], so)
            for it in range(len(ho)):
                xo = ho[it]["x"].reshape(n, 15)
                e = float(np.abs(hist[it, b] - xo).max() / max(np.abs(xo).max(), 1e-12))
                worst = max(worst, e)
                assert e <= 1e-6, (n, k, b, it, e)
            assert rel(got[b], wo["states"].reshape(n, 15)) <= 1e-6
            assert abs(summ[b]["final_cost"] - so["final_cost"]) <= 1e-6 * max(so["final_cost"], 1e-300)
    print("n=%d: worst per-iteration state error %.2e" % (n, worst))


def test_quad_step_matches_the_one_wave_kernel(liw, synth, pyoracle, monkeypatch):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B, K = 6, 23, 12
    base = [synth.make_window(orc, prm, seed=1310 + k, n=n, L=25 + 30 * k) for k in range(4)]
    wins = [base[b % 4] for b in range(B)]
    out = {}
    for variant in ("1", "3"):
        monkeypatch.setenv("LIW_STEP_VARIANT", variant)
        bs = liw.BatchSolver(prm, wins)
        bs.solve(liw.LIW_MODE_INIT, K)
        out[variant] = (bs.states(), bs.summaries())
        bs.close()
    for b in range(B):
        s1, s3 = out["1"][1][b], out["3"][1][b]
        assert (s1["iterations"], s1["termination"], s1["successful"]) == (s3["iterations"], s3["termination"], s3["successful"]), b
        assert rel(out["3"][0][b], out["1"][0][b]) <= 1e-9, b
        assert abs(s1["final_cost"] - s3["final_cost"]) <= 1e-9 * s1["final_cost"]


def test_quad_step_hands_wrapped_rotations_to_the_one_wave_kernel_and_fails_like_ceres(liw, synth, pyoracle, monkeypatch):
    """Batch of 6: windows 1 and 4 carry |theta| > pi (so3 Plus Jacobian != I: k_lm_step's job), window 2 is exactly stationary
    (NaN derivative in the wheel factor -> FAILURE at iteration 0, states untouched), the others are ordinary."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n = 4
    wins = [synth.make_window(orc, prm, seed=9 + k, n=n, L=40) for k in range(6)]
    for k in (1, 4):
        st = wins[k]["states"]
        for f in (0, 2):
            q = st[f, 3:6]
            a = np.linalg.norm(q)
            st[f, 3:6] = q / a * (a - 2 * np.pi)      # same rotation, |q| = 2 pi - a > pi
        assert np.linalg.norm(st[0, 3:6]) > np.pi
        wins[k]["match_pose"][:, 0:6] = st[0, 0:6]
        wins[k]["match_pose"][:, 6:12] = st[:, 0:6]
    wins[2] = synth.make_window(orc, prm, seed=3, n=n, L=40, motion="stationary", odom_noise=0.0, state_noise=0.0)
    monkeypatch.setenv("LIW_STEP_VARIANT", "3")
    bs = liw.BatchSolver(prm, wins)
    bs.solve(liw.LIW_MODE_INIT, 50)
    got, summ = bs.states(), bs.summaries()
    for k in range(6):
        wo, so, _ = oracle_history(pyoracle, orc, wins[k], 50)
        assert (summ[k]["iterations"], summ[k]["termination"]) == (so["iterations"], so["termination"]), (k, summ[k], so)
        assert rel(got[k], wo["states"].reshape(n, 15)) <= 1e-6, k
    assert summ[2]["termination"] == 6 and summ[2]["iterations"] == 0 and np.array_equal(got[2], wins[2]["states"])


@pytest.mark.parametrize("fast", [False, True])
def test_quad_step_tracking_topology_with_prior(liw, synth, pyoracle, monkeypatch, fast):
    """TRACK topology (solver.cpp:631-820) in the quad kernel: constant poses of the older frames (fast mode: their biases too), the prior
    block on frame n-2 (absent in fast mode), no arrow.  init solve -> marginalisation -> tracking solve on the stored prior, forced
    through k_lm_step_quad and through the one-wave kernel: same iteration counts, states 1e-8 apart, and both follow the oracle's chain."""
    prm = dict(synth.office_params(), fast_mode=fast)
    orc = pyoracle.Oracle(prm)
    n, B, K = 6, 11, 8                                 # (fast mode caps the tracking solve at 10 iterations: solver.cpp:800-801)
    base = [synth.make_window(orc, prm, seed=5310 + k, n=n, L=25 + 30 * k) for k in range(4)]
    wins = [base[b % 4] for b in range(B)]
    out = {}
    for variant in ("1", "3"):
        monkeypatch.setenv("LIW_STEP_VARIANT", variant)
        bs = liw.BatchSolver(prm, wins)
        bs.solve(liw.LIW_MODE_INIT, K)
        bs.marginalize()
        x = bs.states()
        x[:, n - 1, 0:3] += 0.01                      # something for the tracker to do
        bs.set_states(x)
        bs.solve(liw.LIW_MODE_TRACK, 0 if fast else K)       # fast mode: the reference's own cap of 10 (the oracle applies it regardless)
        out[variant] = (bs.states(), bs.summaries())
        bs.close()
    for b in range(B):
        s1, s3 = out["1"][1][b], out["3"][1][b]
        assert (s1["iterations"], s1["termination"]) == (s3["iterations"], s3["termination"]), (b, s1, s3)
        assert rel(out["3"][0][b], out["1"][0][b]) <= 1e-8, b
    for k in range(4):
        wo = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.set_max_iterations(K)
        orc.init_solve(wo)
        orc.marginalization(wo)
        wo["states"].reshape(n, 15)[n - 1, 0:3] += 0.01
        orc.solve(wo)
        so = orc.summary()
        assert (out["3"][1][k]["iterations"], out["3"][1][k]["termination"]) == (so["iterations"], so["termination"]), (k, out["3"][1][k], so)
        assert rel(out["3"][0][k], wo["states"].reshape(n, 15)) <= 1e-6, k
    orc.set_max_iterations(50)
