"""Reference NaN semantics at the two remaining norm()-of-a-zero-Jet sites (the wheel factor's site is
tests/test_gpu_stationary.py::test_exactly_stationary_interval_fails_the_evaluation_like_ceres).

* e_laser::dis_from_line (reference src/utilies/common.h:86-95): a mapped end point EXACTLY on the matched line makes
  `(p2p - (l . p2p) l).norm()` the norm of a zero Jet -> 0 * inf = NaN derivative;
* ground_factor_q (reference src/factor/ground_factor.h:68-80): a wheel z axis EXACTLY equal to e3 makes `asin(|z x e3|)` the
  same thing.

In both cases the residual VALUE is a finite 0, the Jacobian is not finite, Ceres' ResidualBlock::Evaluate ->
IsEvaluationValid rejects the block, IterationZero fails and the solve ends with termination FAILURE, zero iterations, states
handed back untouched.  The oracle (faithful Jets + restated Ceres) does exactly that; the HIP path must do the same through the
C ABI — per-factor outputs (liw_eval_factors), the init-topology solve and the tracking-topology solve.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solvers(liw, pyoracle, prm):
    return pyoracle.Oracle(prm), liw.Solver(prm)


def _block_on_the_line(d, j):
    """block j: second segment's first end point := first segment's second end point (same laser-frame coordinates)"""
    d["laser_pts"][j, 6:9] = d["laser_pts"][j, 3:6]


def test_laser_point_exactly_on_the_line_init_topology(liw, synth, pyoracle):
    prm = synth.office_params()
    orc, slv = _solvers(liw, pyoracle, prm)
    n, L = 4, 24
    d = synth.make_window(orc, prm, seed=5, n=n, L=L)
    # frame 3 gets frame 0's pose: both segments of its blocks are then mapped by the very same transform, bit for bit
    d["states"][3, 0:6] = d["states"][0, 0:6]
    d["match_pose"][3, 6:12] = d["states"][3, 0:6]
    owned = np.flatnonzero(d["laser_frame"] == 3)
    j = int(owned[1])
    _block_on_the_line(d, j)
    st = d["states"]
    ro, Jo = orc.eval_laser(d["laser_pts"][j], st[0, 0:3], st[0, 3:6], st[3, 0:3], st[3, 3:6])
    assert ro[0] == 0.0 and np.isfinite(ro).all()
    assert not np.isfinite(Jo[0]).all(), "the oracle's Jet must yield a non-finite derivative here (common.h:94)"
    assert np.isfinite(Jo[1]).all()
    slv.set_window(liw.Window(d))
    f = slv.eval_factors(liw.LIW_MODE_INIT)
    assert f["laser_res"][j][0] == 0.0 and np.isfinite(f["laser_res"]).all()
    assert not np.isfinite(f["laser_jac"][j][0]).all(), "the HIP laser role must not hide the non-finite derivative"
    assert np.isfinite(f["laser_jac"][j][1]).all()
    others = [k for k in range(L) if k != j]
    assert np.isfinite(f["laser_jac"][others]).all()
    for k in owned:
        if k == j:
            continue
        r, J = orc.eval_laser(d["laser_pts"][k], st[0, 0:3], st[0, 3:6], st[3, 0:3], st[3, 3:6])
        assert np.abs(f["laser_res"][k] - r).max() <= 1e-10 * max(1.0, np.abs(r).max())
    wo, wg = pyoracle.Window(d), liw.Window(d)
    orc.set_prior(None)
    slv.set_prior(None)
    orc.set_max_iterations(50)
    orc.init_solve(wo)
    so = orc.summary()
    slv.set_window(wg)
    sg = slv.init_solve()
    assert (so["iterations"], so["termination"]) == (0, 6), so
    assert (sg["iterations"], sg["termination"]) == (0, 6), sg
    assert np.array_equal(wg["states"], d["states"]) and np.array_equal(wo["states"], d["states"])


def test_laser_point_exactly_on_the_line_tracking_topology(liw, synth, pyoracle):
    prm = synth.office_params()
    orc, slv = _solvers(liw, pyoracle, prm)
    n, L = 2, 12
    d = synth.make_window(orc, prm, seed=9, n=n, L=L)
    # tracking: the newest frame's blocks are tied to the constant laser_match pose (p1, q1) (solver.cpp:669-698)
    d["match_pose"][1, 0:6] = d["states"][1, 0:6]
    d["match_pose"][1, 6:12] = d["states"][1, 0:6]
    j = int(np.flatnonzero(d["laser_frame"] == 1)[3])
    _block_on_the_line(d, j)
    wo, wg = pyoracle.Window(d), liw.Window(d)
    orc.set_prior(None)
    slv.set_prior(None)
    orc.set_max_iterations(50)
    orc.solve(wo)
    so = orc.summary()
    slv.set_window(wg)
    f = slv.eval_factors(liw.LIW_MODE_TRACK)
    assert f["laser_res"][j][0] == 0.0 and not np.isfinite(f["laser_jac"][j][0]).all()
    sg = slv.solve()
    assert (so["iterations"], so["termination"]) == (0, 6), so
    assert (sg["iterations"], sg["termination"]) == (0, 6), sg
    assert np.array_equal(wg["states"], d["states"]) and np.array_equal(wo["states"], d["states"])


def _level_params(synth):
    """wheel frame = IMU frame rotated by the identity: a zero rotation vector then puts the wheel z axis exactly on e3"""
    prm = synth.office_params()
    T = np.eye(4)
    T[:3, 3] = [-0.061, 0.919, -0.224]
    prm["T_imu_to_wheel"] = list(T.reshape(16))
    return prm


@pytest.mark.parametrize("nd3", [False, True])
def test_ground_tilt_exactly_level(liw, synth, pyoracle, monkeypatch, nd3):
    if nd3:
        monkeypatch.setenv("LIW_SMALL_ND3", "1")
    else:
        monkeypatch.delenv("LIW_SMALL_ND3", raising=False)
    prm = _level_params(synth)
    orc, slv = _solvers(liw, pyoracle, prm)
    n, L = 4, 24
    d = synth.make_window(orc, prm, seed=11, n=n, L=L)
    d["states"][2, 3:6] = 0.0
    d["match_pose"][2, 9:12] = 0.0
    st = d["states"]
    ro, Jo = orc.eval_ground(st[2, 0:3], st[2, 3:6])
    assert ro[1] == 0.0 and np.isfinite(ro).all()
    assert not np.isfinite(Jo[1]).all(), "the oracle's Jet must yield a non-finite derivative here (ground_factor.h:78)"
    slv.set_window(liw.Window(d))
    f = slv.eval_factors(liw.LIW_MODE_INIT)
    assert f["ground_res"][2][1] == 0.0 and np.isfinite(f["ground_res"]).all()
    assert not np.isfinite(f["ground_jac"][2][1]).all(), "the HIP ground role must not hide the non-finite derivative"
    for i in (0, 1, 3):
        r, J = orc.eval_ground(st[i, 0:3], st[i, 3:6])
        assert np.isfinite(f["ground_jac"][i]).all()
        assert np.abs(f["ground_res"][i] - r).max() <= 1e-10 * max(1.0, np.abs(r).max())
        assert np.abs(f["ground_jac"][i] - J).max() <= 1e-10 * max(1.0, np.abs(J).max())
    wo, wg = pyoracle.Window(d), liw.Window(d)
    orc.set_prior(None)
    slv.set_prior(None)
    orc.set_max_iterations(50)
    orc.init_solve(wo)
    so = orc.summary()
    slv.set_window(wg)
    sg = slv.init_solve()
    assert (so["iterations"], so["termination"]) == (0, 6), so
    assert (sg["iterations"], sg["termination"]) == (0, 6), sg
    assert np.array_equal(wg["states"], d["states"]) and np.array_equal(wo["states"], d["states"])


def test_zero_norm_sites_in_a_batch_fail_only_their_own_window(liw, synth, pyoracle):
    """Batched path (k_lin_laser + the quad / one-wave step kernels): the poisoned window ends in FAILURE with its states untouched, its
    neighbours solve exactly as they do alone."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, L, B = 6, 60, 2304          # >= 1 024 windows: k_lm_step_quad
    base = [synth.make_window(orc, prm, seed=40 + k, n=n, L=L) for k in range(4)]
    bad = dict(base[1])
    bad = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in bad.items()}
    bad["states"][5, 0:6] = bad["states"][0, 0:6]
    bad["match_pose"][5, 6:12] = bad["states"][5, 0:6]
    _block_on_the_line(bad, int(np.flatnonzero(bad["laser_frame"] == 5)[0]))
    wins = [bad if (k % 4) == 1 else base[k % 4] for k in range(B)]
    x0 = [np.array(w["states"], copy=True) for w in wins[:8]]
    bs = liw.BatchSolver(prm, wins)
    bs.solve(liw.LIW_MODE_INIT, 50)
    info = bs.summaries()
    X = bs.states()
    ref = {}
    for k in (0, 2, 3):
        w = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.set_max_iterations(50)
        orc.init_solve(w)
        ref[k] = (orc.summary(), np.array(w["states"], copy=True))
    for b in range(8):
        k = b % 4
        if k == 1:
            assert (info[b]["iterations"], info[b]["termination"]) == (0, 6), (b, info[b])
            assert np.array_equal(X[b], x0[b])
        else:
            so, xs = ref[k]
            assert info[b]["iterations"] == so["iterations"] and info[b]["termination"] == so["termination"], (b, so, info[b])
            assert np.abs(X[b] - xs).max() <= 1e-6 * np.abs(xs).max()
    bs.close()
