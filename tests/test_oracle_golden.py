"""Oracle vs the committed golden vectors (tests/golden/factors_golden.json, made by tests/golden/make_golden.py:
independent torch-autograd derivation) and vs central finite differences.  PARITY UNPINNED w.r.t. the reference
itself (it has no tests and cannot be built here) — these pin the oracle's math."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "factors_golden.json")))
TOL = 1e-8   # relative; the two derivations differ by round-off amplified by the 1e2..1e4 information weights


def close(a, b, tol=TOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.fixture(scope="module")
def orc(pyoracle):
    return pyoracle.Oracle(G["params"])


def test_laser_factor_golden(orc):
    for c in G["laser"]:
        x = np.array(c["x"])
        r, J = orc.eval_laser(c["pts"], x[0:3], x[3:6], x[6:9], x[9:12])
        assert close(r, c["res"]) and close(J, c["jac"])


def test_imu_factor_golden(orc):
    for c in G["imu"]:
        x = np.array(c["x"])
        r, J = orc.eval_imu(c["X"], c["J"], c["sqrtP"], c["Dt"], x[:15], x[15:])
        assert close(r, c["res"]) and close(J, c["jac"])


def test_wheel_factor_golden(orc):
    for c in G["wheel"]:
        x = np.array(c["x"])
        r, J = orc.eval_wheel(c["T"], c["sqrtP"], x[0:3], x[3:6], x[6:9], x[9:12])
        assert close(r, c["res"]) and close(J, c["jac"])


def test_ground_factor_golden(orc):
    for c in G["ground"]:
        x = np.array(c["x"])
        r, J = orc.eval_ground(x[0:3], x[3:6])
        assert close(r, c["res"]) and close(J, c["jac"])


def test_window_normal_equations_golden(orc, pyoracle):
    w = G["window_init"]
    win = pyoracle.Window({k: (np.array(v) if k != "n" else v) for k, v in w["window"].items()})
    H, g, cost = orc.linearize(win, 0)
    assert abs(cost - w["cost"]) <= 1e-10 * w["cost"]
    assert close(H, w["H"], 1e-8) and close(g, w["g"], 1e-8)


def _fd(f, x, h=1e-6):
    x = np.array(x, dtype=np.float64)
    r0 = f(x)
    J = np.zeros((len(r0), len(x)))
    for k in range(len(x)):
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        J[:, k] = (f(xp) - f(xm)) / (2 * h)
    return J


def test_factor_jacobians_vs_finite_differences(orc):
    c = G["laser"][0]
    f = lambda x: orc.eval_laser(c["pts"], x[0:3], x[3:6], x[6:9], x[9:12])[0]
    J = orc.eval_laser(c["pts"], *np.split(np.array(c["x"]), 4))[1]
    assert np.abs(_fd(f, c["x"]) - J).max() <= 1e-5 * max(1.0, np.abs(J).max())
    c = G["imu"][1]
    f = lambda x: orc.eval_imu(c["X"], c["J"], c["sqrtP"], c["Dt"], x[:15], x[15:])[0]
    J = orc.eval_imu(c["X"], c["J"], c["sqrtP"], c["Dt"], np.array(c["x"])[:15], np.array(c["x"])[15:])[1]
    assert np.abs(_fd(f, c["x"], 1e-7) - J).max() <= 1e-5 * max(1.0, np.abs(J).max())
    c = G["wheel"][2]
    f = lambda x: orc.eval_wheel(c["T"], c["sqrtP"], x[0:3], x[3:6], x[6:9], x[9:12])[0]
    J = orc.eval_wheel(c["T"], c["sqrtP"], *np.split(np.array(c["x"]), 4))[1]
    assert np.abs(_fd(f, c["x"]) - J).max() <= 1e-5 * max(1.0, np.abs(J).max())
    c = G["ground"][3]
    f = lambda x: orc.eval_ground(x[0:3], x[3:6])[0]
    J = orc.eval_ground(np.array(c["x"])[0:3], np.array(c["x"])[3:6])[1]
    assert np.abs(_fd(f, c["x"]) - J).max() <= 1e-5 * max(1.0, np.abs(J).max())
