"""GPU parity on the stationary-robot arms of wheel_odom_factor (reference src/factor/wheel_factor.h:45, :58, :63).

A robot at rest, turning on the spot or driving straight takes the `else` sides of the three value-dependent branches of the
wheel factor; the moving arc every other test uses never does.  Each case below asserts WHICH arm every block takes (recomputed in
numpy, tests/parity_util.py::wheel_arms) and compares the HIP path with the oracle's Jets through the C-ABI: residuals / Jacobians
<= 1e-10 in both instantiations of the wheel role (one and three derivative directions per lane), the normal equations, and the LM
history of windows that contain such intervals.  An EXACTLY stationary interval (identical states, identity odometry increment)
makes norm() of a zero Jet: NaN derivatives -> Ceres rejects the evaluation -> FAILURE with the states untouched; both sides must
do the same.  (While INITIALIZING the reference drops scans of a robot at rest, trajectory.cpp:163; TRACKING has no such gate.)
"""
import numpy as np
import pytest

from parity_util import assert_normal_eq_close, rel_inf, wheel_arms

pytestmark = pytest.mark.gpu

TINY = dict(state_p_sigma=1e-6, state_q_sigma=1e-5)        # |dp| ~ 1e-5 < 1e-4 (0.9 m lever arm of the wheel frame), |dq| ~ 2e-5 < 1e-3
# name -> (make_window keywords, expected (moving45, moving58, moving63))
CASES = {
    "moving_arc": (dict(), (True, True, True)),
    "at_rest": (dict(motion="stationary", odom_noise=0.0, **TINY), (False, False, False)),
    "at_rest_noisy_odometry": (dict(motion="stationary", odom_noise=2e-4, **TINY), None),     # arm :45/:58 decided by the odometry noise
    "turning_on_the_spot": (dict(motion="rotate", odom_noise=0.0, state_p_sigma=1e-6, state_q_sigma=1e-5), (False, False, True)),
    "turning_states_off": (dict(motion="rotate", odom_noise=0.0), (False, False, True)),     # states 2 cm / 0.5 deg off: len > 1e-4, o_len = 0
    "straight_line": (dict(motion="translate", state_q_sigma=1e-5), (True, True, False)),
    "odometry_at_rest_states_moving": (dict(motion="stationary", state_motion="arc", odom_noise=0.0), (False, False, False)),
    "odometry_moving_states_at_rest": (dict(motion="arc", state_motion="stationary", **TINY), (False, False, False)),
    "odometry_turning_states_at_rest": (dict(motion="rotate", state_motion="stationary", odom_noise=0.0, **TINY), (False, False, False)),
    "odometry_straight_states_turning": (dict(motion="translate", state_motion="rotate", state_p_sigma=1e-6, state_q_sigma=1e-5), (False, False, False)),
}


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()))


@pytest.fixture(scope="module")
def setup(liw, synth, pyoracle):
    prm = synth.office_params()
    return prm, pyoracle.Oracle(prm), liw.Solver(prm)


@pytest.mark.parametrize("nd3", [False, True])
@pytest.mark.parametrize("case", sorted(CASES))
def test_wheel_factor_arms(liw, synth, pyoracle, setup, monkeypatch, case, nd3):
    prm, orc, slv = setup
    if nd3:
        monkeypatch.setenv("LIW_SMALL_ND3", "1")
    else:
        monkeypatch.delenv("LIW_SMALL_ND3", raising=False)
    kw, want = CASES[case]
    n, L = 7, 60
    d = synth.make_window(orc, prm, seed=31, n=n, L=L, **kw)
    slv.set_window(liw.Window(d))
    f = slv.eval_factors(liw.LIW_MODE_INIT)
    st = d["states"]
    seen = set()
    for k in range(n - 1):
        a = wheel_arms(synth, prm, d, k)
        arms = (a["moving45"], a["moving58"], a["moving63"])
        seen.add(arms)
        if want is not None:
            assert arms == want, (case, k, a)
        # keep clear of the thresholds: a value within round-off of 1e-4 / 1e-3 could flip an arm on one side only
        for v, thr in ((a["len"], 1e-4), (a["o_len"], 1e-4), (a["q"], 1e-3), (a["oq"], 1e-3)):
            assert abs(v - thr) > 1e-9 * thr, (case, k, a)
        r, J = orc.eval_wheel(d["wheel_T"][k], d["wheel_sqrtP"][k], st[k, 0:3], st[k, 3:6], st[k + 1, 0:3], st[k + 1, 3:6])
        assert np.isfinite(J).all() and np.isfinite(f["wheel_jac"][k]).all(), (case, k)
        assert rel(f["wheel_res"][k], r) < 1e-10, (case, k, a)
        assert rel(f["wheel_jac"][k], J) < 1e-10, (case, k, a)
        r, J = orc.eval_imu(d["imu_X"][k], d["imu_J"][k], d["imu_sqrtP"][k], d["imu_Dt"][k], st[k], st[k + 1])
        assert rel(f["imu_res"][k], r) < 1e-10 and rel(f["imu_jac"][k], J) < 1e-10, (case, k)
    print(case, "arms taken:", sorted(seen))


def test_every_arm_is_covered(synth, pyoracle, setup):
    """The union of the cases takes both sides of each of the three branches, and the else side of each one alone and together."""
    prm, orc, _ = setup
    taken = set()
    for case, (kw, _) in CASES.items():
        d = synth.make_window(orc, prm, seed=31, n=7, L=60, **kw)
        for k in range(6):
            a = wheel_arms(synth, prm, d, k)
            taken.add((a["moving45"], a["moving58"], a["moving63"]))
    for idx in range(3):
        assert {t[idx] for t in taken} == {True, False}
    assert {(True, True, True), (False, False, False), (False, False, True), (True, True, False)} <= taken


@pytest.mark.parametrize("case", ["at_rest", "turning_on_the_spot", "straight_line", "odometry_at_rest_states_moving"])
def test_normal_equations_on_stationary_windows(liw, synth, pyoracle, setup, case):
    prm, orc, slv = setup
    kw, _ = CASES[case]
    d = synth.make_window(orc, prm, seed=77, n=10, L=300, **kw)
    slv.set_window(liw.Window(d))
    H, g, c = slv.linearize(liw.LIW_MODE_INIT)
    Ho, go, co = orc.linearize(pyoracle.Window(d), 0)
    assert abs(c - co) <= 1e-12 * co
    assert_normal_eq_close(H, g, Ho, go, co, what=case)


def _history_check(liw, pyoracle, orc, slv, d, iters=50):
    n = d["n"]
    wo, wg = pyoracle.Window(d), liw.Window(d)
    orc.set_prior(None)
    slv.set_prior(None)
    orc.set_max_iterations(iters)
    orc.init_solve(wo)
    so, ho = orc.summary(), orc.iterations()
    slv.set_window(wg)
    sg = slv.init_solve(iters)
    hg = slv.history()
    assert (sg["iterations"], sg["termination"]) == (so["iterations"], so["termination"]), (sg, so)
    assert len(hg) == len(ho)
    worst = 0.0
    for k in range(len(ho)):
        xo = ho[k]["x"].reshape(n, 15)
        worst = max(worst, float(np.abs(hg[k] - xo).max() / max(np.abs(xo).max(), 1e-12)))
        assert worst <= 1e-6, "iteration %d" % k
    assert rel_inf(wg["states"], wo["states"]) <= 1e-6
    return so, worst


def test_lm_history_first_ten_frames_at_rest(liw, synth, pyoracle, setup):
    """The start of a log recorded from a parked robot: frames 0..10 at rest (odometry noise only), then the arc."""
    prm, orc, slv = setup
    d = synth.make_window(orc, prm, seed=404, n=30, L=2000, motion="standstill_then_go")
    arms = [wheel_arms(synth, prm, d, k) for k in range(29)]
    assert all(a["o_len"] < 2e-3 and a["oq"] < 1e-3 for a in arms[:9]) and all(a["moving45"] and a["moving63"] for a in arms[16:])
    assert not all(a["moving63"] for a in arms[:10])
    so, worst = _history_check(liw, pyoracle, orc, slv, d)
    print("standstill-then-go C2 window: %d iterations, termination %d, worst per-iteration state error %.2e" % (so["iterations"], so["termination"], worst))


@pytest.mark.parametrize("case", ["at_rest", "turning_on_the_spot", "straight_line"])
def test_lm_history_on_stationary_windows(liw, synth, pyoracle, setup, case):
    prm, orc, slv = setup
    kw, _ = CASES[case]
    kw = dict(kw)
    kw.update(state_p_sigma=2e-4 if case != "straight_line" else 0.02, state_q_sigma=2e-4)   # the solver has work to do; arms may change along the way
    d = synth.make_window(orc, prm, seed=99, n=8, L=160, **kw)
    so, worst = _history_check(liw, pyoracle, orc, slv, d, iters=30)
    print(case, so, "worst %.2e" % worst)


def _track_pair(synth, orc, prm, **kw):
    d = synth.make_window(orc, prm, seed=616, n=2, L=60, laser_on_frame0=False, **kw)
    return d


@pytest.mark.parametrize("case", ["at_rest", "at_rest_noisy_odometry", "turning_on_the_spot"])
def test_tracking_frame_of_a_robot_at_rest(liw, synth, pyoracle, setup, case):
    """The reference's steady state (2-frame TRACK solve + marginalisation, trajectory.cpp:525-560) while the robot stands still."""
    prm, orc, slv = setup
    kw, _ = CASES[case]
    # a prior for frame 0: marginalise a short moving window whose last frame becomes this window's first
    dm = synth.make_window(orc, prm, seed=615, n=4, L=120)
    wo = pyoracle.Window(dm)
    orc.set_prior(None)
    orc.set_max_iterations(50)
    orc.init_solve(wo)
    orc.marginalization(wo)
    Xo, Jo, Ro = orc.get_prior()
    d = _track_pair(synth, orc, prm, **kw)
    shift = Xo[0:15] - d["states"][0]            # move the pair onto the prior's frame: same relative geometry
    d["states"][:, 0:3] += shift[0:3]
    d["match_pose"][:, 0:3] += shift[0:3]
    d["match_pose"][:, 6:9] += shift[0:3]
    d["states"][1, 0:3] += 2e-5                   # something to do for the tracker, still inside the stationary arms
    wo2, wg2 = pyoracle.Window(d), liw.Window(d)
    orc.solve(wo2)
    so = orc.summary()
    slv.set_prior((Xo, Jo, Ro))
    slv.set_window(wg2)
    sg = slv.solve()
    assert (sg["iterations"], sg["termination"]) == (so["iterations"], so["termination"]), (sg, so)
    assert rel_inf(wg2["states"], wo2["states"]) <= 1e-6
    orc.marginalization(wo2)
    mo = orc.marg_pieces()
    mg = slv.marginalization()
    assert rel_inf(mg["Delta_H"], mo["Delta_H"]) <= 1e-6 and rel_inf(mg["Delta_g"], mo["Delta_g"]) <= 1e-6


def test_exactly_stationary_interval_fails_the_evaluation_like_ceres(liw, synth, pyoracle, setup):
    """Identical consecutive states + identity odometry increment: |q| = sqrt(0) on Jets -> NaN derivatives (wheel_factor.h:63).
    Ceres' IsEvaluationValid rejects the block, IterationZero fails: termination FAILURE, no iteration, states untouched."""
    prm, orc, slv = setup
    d = synth.make_window(orc, prm, seed=3, n=4, L=20, motion="stationary", odom_noise=0.0, state_noise=0.0)
    assert np.abs(np.diff(d["states"][:, 0:6], axis=0)).max() == 0.0
    st = d["states"]
    slv.set_window(liw.Window(d))
    f = slv.eval_factors(liw.LIW_MODE_INIT)
    for k in range(3):
        _, J = orc.eval_wheel(d["wheel_T"][k], d["wheel_sqrtP"][k], st[k, 0:3], st[k, 3:6], st[k + 1, 0:3], st[k + 1, 3:6])
        assert not np.isfinite(J).all()
        assert not np.isfinite(f["wheel_jac"][k]).all(), "the HIP wheel role must not hide the non-finite derivative"
    wo, wg = pyoracle.Window(d), liw.Window(d)
    orc.set_prior(None)
    slv.set_prior(None)
    orc.set_max_iterations(50)
    orc.init_solve(wo)
    so = orc.summary()
    slv.set_window(wg)
    sg = slv.init_solve()
    assert so["termination"] == 6 and so["iterations"] == 0
    assert (sg["iterations"], sg["termination"]) == (0, 6), sg
    assert np.array_equal(wg["states"], d["states"]) and np.array_equal(wo["states"], d["states"])


def _same_with_nan(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    fa, fb = np.isfinite(a), np.isfinite(b)
    assert np.array_equal(fa, fb), "non-finite entries in different places"
    if fa.any():
        assert np.abs(a[fa] - b[fa]).max() <= tol * max(1.0, np.abs(b[fb]).max())


@pytest.mark.parametrize("level", [False, True])
def test_merged_wheel_ground_role_of_batches_on_every_arm(liw, synth, pyoracle, level):
    """k_lin_small (batches: two lanes per wheel block, the translation directions as extra dual parts of the scalar tail, the ground factors
    evaluated by the wheel lanes) against the lane layouts of k_lin_all (a single window: three / nine lanes per block, ground waves of their
    own), which test_wheel_factor_arms / test_ground_tilt_exactly_level compare with the oracle's Jets factor by factor: every threshold arm
    of wheel_factor.h:36-70, an exactly stationary interval (NaN derivative of norm(), :63) and — `level` — a wheel frame whose z axis is
    exactly vertical in one frame (NaN derivative of the tilt, ground_factor.h:78).  Normal equations window by window, non-finite entries in
    the same places."""
    prm = synth.office_params()
    if level:
        T = np.eye(4)
        T[:3, 3] = [-0.061, 0.919, -0.224]
        prm["T_imu_to_wheel"] = list(T.reshape(16))
    orc = pyoracle.Oracle(prm)
    n, L = 7, 60
    base = [synth.make_window(orc, prm, seed=31, n=n, L=L, **CASES[c][0]) for c in sorted(CASES)]
    base.append(synth.make_window(orc, prm, seed=3, n=n, L=L, motion="stationary", odom_noise=0.0, state_noise=0.0))
    if level:
        for w in base[:3]:
            w["states"][2, 3:6] = 0.0
            w["match_pose"][2, 9:12] = 0.0
            w["states"][0, 3:6] = 0.0      # frame 0: evaluated by the first block's lane 0
            w["match_pose"][0, 9:12] = 0.0
    B = 96                                 # 672 laser waves: the roles run as kernels of their own (k_lin_small), 31 wheel blocks per wave
    wins = [base[b % len(base)] for b in range(B)]
    big = liw.BatchSolver(prm, wins)
    assert (B * n + 2 * B * (n - 1)) > 256
    for mode in (liw.LIW_MODE_INIT, liw.LIW_MODE_MARG):
        big.linearize(mode)
        Hb, gb, cb = [t.cpu().numpy() for t in big.export_dense(mode)]
        nonfinite = 0
        for k, w in enumerate(base):
            one = liw.BatchSolver(prm, [w])
            one.linearize(mode)
            H1, g1, c1 = [t.cpu().numpy() for t in one.export_dense(mode)]
            nonfinite += int(not np.isfinite(H1[0]).all())
            for b in (k, k + len(base) * ((B - 1 - k) // len(base))):      # first and last copy of this window in the batch
                _same_with_nan(Hb[b], H1[0], 1e-12)
                _same_with_nan(gb[b], g1[0], 1e-12)
                _same_with_nan(cb[b], c1[0], 1e-12)
        assert nonfinite >= (4 if level else 1), "the exactly stationary / exactly level windows must show up as non-finite entries"
