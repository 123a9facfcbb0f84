"""CPU: the oracle on the stationary-robot arms of wheel_odom_factor (reference src/factor/wheel_factor.h:45, :58, :63) against
tests/golden/wheel_stationary_golden.json (independent torch-autograd derivation, tests/golden/make_golden_stationary.py) and
central finite differences, the arm coverage of the motion cases, and Ceres' evaluation-failure semantics on an exactly
stationary interval.  PARITY UNPINNED w.r.t. the reference itself, as everywhere."""
import importlib
import json
import os

import numpy as np
import pytest

from parity_util import wheel_arms

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "wheel_stationary_golden.json")))


def close(a, b, tol=1e-8):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.fixture(scope="module")
def orc(pyoracle):
    return pyoracle.Oracle(G["params"])


def test_oracle_wheel_factor_on_every_arm(orc):
    arms = set()
    for case, vec in G["cases"].items():
        for c in vec:
            x = np.array(c["x"])
            r, J = orc.eval_wheel(c["T"], c["sqrtP"], x[0:3], x[3:6], x[6:9], x[9:12])
            assert close(r, c["res"]) and close(J, c["jac"]), case
            arms.add(tuple(c["arms"]))
    assert arms == {(True, True, True), (False, False, False), (False, False, True), (True, True, False)}


def test_oracle_wheel_jacobian_vs_finite_differences_on_every_arm(orc):
    for case, vec in G["cases"].items():
        for c in vec:
            x = np.array(c["x"])
            _, J = orc.eval_wheel(c["T"], c["sqrtP"], x[0:3], x[3:6], x[6:9], x[9:12])
            Jn = np.zeros((3, 12))
            for e in range(12):
                h = 1e-7 if min(np.hypot(*x[0:2] - x[6:8]), 1.0) > 1e-3 else 1e-9     # stay on the same arm
                xp, xm = x.copy(), x.copy()
                xp[e] += h
                xm[e] -= h
                rp, _ = orc.eval_wheel(c["T"], c["sqrtP"], xp[0:3], xp[3:6], xp[6:9], xp[9:12])
                rm, _ = orc.eval_wheel(c["T"], c["sqrtP"], xm[0:3], xm[3:6], xm[6:9], xm[9:12])
                Jn[:, e] = (rp - rm) / (2 * h)
            assert np.abs(J - Jn).max() <= 2e-4 * max(1.0, np.abs(J).max()), (case, np.abs(J - Jn).max())


def test_motion_cases_cover_both_sides_of_each_branch(synth, pyoracle):
    tst = importlib.import_module("test_gpu_stationary")
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    taken = set()
    for case, (kw, want) in tst.CASES.items():
        d = synth.make_window(orc, prm, seed=31, n=7, L=60, **kw)
        for k in range(6):
            a = wheel_arms(synth, prm, d, k)
            arms = (a["moving45"], a["moving58"], a["moving63"])
            if want is not None:
                assert arms == want, (case, k, a)
            taken.add(arms)
    for idx in range(3):
        assert {t[idx] for t in taken} == {True, False}


def test_exactly_stationary_interval_is_an_evaluation_failure(synth, pyoracle):
    """norm() of a zero Jet (wheel_factor.h:63) -> NaN derivatives -> Ceres' IsEvaluationValid fails the residual block ->
    IterationZero fails: termination FAILURE (6 here), zero iterations, states untouched."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    d = synth.make_window(orc, prm, seed=3, n=4, L=20, motion="stationary", odom_noise=0.0, state_noise=0.0)
    st = d["states"]
    for k in range(3):
        r, J = orc.eval_wheel(d["wheel_T"][k], d["wheel_sqrtP"][k], st[k, 0:3], st[k, 3:6], st[k + 1, 0:3], st[k + 1, 3:6])
        assert np.all(r == 0.0) and not np.isfinite(J).any()
    w = pyoracle.Window(d)
    orc.set_prior(None)
    orc.set_max_iterations(50)
    orc.init_solve(w)
    s = orc.summary()
    assert (s["iterations"], s["termination"], s["successful"]) == (0, 6, 0)
    assert np.array_equal(w["states"], d["states"])


def test_default_motion_is_bit_identical_to_the_round_2_generator(synth, pyoracle):
    """The motion switch must not move the C2 bench workload: digest of the seed-20240 recipe's arrays."""
    import hashlib
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    d = synth.make_window(orc, prm, seed=5, n=10, L=300)
    h = hashlib.sha256(d["states"].tobytes() + d["wheel_T"].tobytes() + d["laser_pts"].tobytes()).hexdigest()[:16]
    assert h == "2aa290fac84e0ef3"
