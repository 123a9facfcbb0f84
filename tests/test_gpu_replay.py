"""End-to-end replay: a synthetic sensor log (IMU 200 Hz, wheel odometry 20 Hz, skewed LaserScans 10 Hz)
through lvio_2d::trajectory of include/lvio_2d_trajectory.hpp — dispatch merge, pre-integration, de-skew, line extraction,
matching, init window + init_solve, per-frame tracking solve + marginalisation on the MI355X, TUM output — against the
oracle's restatement of the same driver on the CPU.  Tolerance 1e-6 relative on every pose of the trajectory."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_replay(liw):
    src = os.path.join(ROOT, "tools", "replay_log.cpp")
    exe = os.path.join(ROOT, "tools", "replay_log")
    libdir = os.path.dirname(liw.LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lliw_window", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def oracle_replay(pyoracle, prm, lp, msgs, keep=1, backend=None, capture=False):
    """backend: None or (pg params, loop schedule, solve period)"""
    orc = pyoracle.TrajectoryOracle(prm, lp, keep_window_size=keep)
    orc.set_capture(capture)
    if backend is not None:
        orc.enable_backend(backend[0], backend[1], solve_period=backend[2])
    for m in msgs:
        if m["type"] == 0:
            orc.add_imu(m["time"], m["acc"], m["gyro"])
        elif m["type"] == 1:
            orc.add_wheel(m["time"], m["R"], m["t"])
        else:
            pts, ts = pyoracle.laser_to_points(m["ranges"], m["angle_min"], m["angle_increment"], m["time_increment"], m["time"])
            orc.add_laser(m["time"], pts, ts)
    return orc


@pytest.mark.parametrize("seed,duration", [(1, 4.0), (2, 6.0)])
def test_replay_matches_oracle_trajectory(liw, synth, pyoracle, tmp_path, seed, duration):
    import importlib
    replay = importlib.import_module("2dliw-slam_amd.replay")
    prm = synth.office_params()
    lp = liw.laser.office_laser_params(prm)
    msgs, truth = replay.make_log(prm, duration=duration, seed=seed)
    replay.write_log(str(tmp_path / "log.bin"), msgs)
    out = str(tmp_path) + "/"
    r = subprocess.run([build_replay(liw), str(tmp_path / "log.bin"), out], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    orc = oracle_replay(pyoracle, prm, lp, msgs)
    # counters / state machine
    raw = open(out + "result.bin", "rb").read()
    status, frames, tracked, inits, keyframes, sstat = struct.unpack("<6i", raw[:24])
    tm, state = struct.unpack("<d", raw[24:32])[0], np.frombuffer(raw[32:32 + 120], dtype=np.float64)
    c = orc.counters()
    assert (status, frames, tracked, inits, keyframes) == (c["status"], c["frames"], c["tracked"], c["initializations"], c["keyframes"])
    assert status == 1 and inits == 1 and tracked >= 20 and sstat == 0
    to, so = orc.current()
    assert tm == to
    assert np.abs(state - so).max() <= 1e-6 * max(1.0, np.abs(so).max())
    # TUM trajectory: same stamps, poses within 1e-6
    got = replay.read_tum(out + "fornt_end.txt")
    ref = np.array([ln.split() for ln in orc.tum().splitlines()[1:]], dtype=np.float64)
    assert got.shape == ref.shape == (tracked, 8)
    assert np.array_equal(got[:, 0], ref[:, 0])
    assert np.abs(got[:, 1:] - ref[:, 1:]).max() <= 1e-6 * max(1.0, np.abs(ref[:, 1:]).max())
    assert open(out + "fornt_end.txt").readline() == "#Time px py pz qx qy qz qw\n"
    # the estimate follows the truth (base pose; the estimator's world is the base frame at start-up)
    T = truth.T_w_o(got[-1, 0])
    assert np.linalg.norm(got[-1, 1:3] - T[:2, 3]) < 0.10
    # the record table carries the reference's labels
    md = open(out + "traj.md").read()
    for label in ("| solve |", "| marginalization |", "| spawn_scan |", "| match_line |", "| lines each frame |", "| match line size |"):
        assert label in md, label


def test_replay_look_ahead_is_order_preserving(liw, synth, tmp_path):
    """The 40-message look-ahead of the dispatcher only delays messages: with per-sensor sorted streams the hand-over
    order, hence the trajectory, does not depend on it."""
    import importlib
    replay = importlib.import_module("2dliw-slam_amd.replay")
    prm = synth.office_params()
    msgs, _ = replay.make_log(prm, duration=3.0, seed=5)
    replay.write_log(str(tmp_path / "log.bin"), msgs)
    exe = build_replay(liw)
    outs = []
    for la in (40, 1):
        d = tmp_path / ("la%d" % la)
        d.mkdir()
        r = subprocess.run([exe, str(tmp_path / "log.bin"), str(d) + "/", str(la)], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(open(str(d) + "/fornt_end.txt").read())
    assert outs[0] == outs[1] and outs[0].count("\n") > 10


def _check_free_running(out, orc, replay, min_tracked, lead):
    """Free-running product replay vs free-running oracle replay.  The driver itself is chaotic beyond the first frames (oracle vs
    oracle with 1e-13 input noise departs by 1e-3 within 46 frames at keep = 29: tests/soak/sensitivity_replay.py), so the
    bar here is: identical state machine, the first `lead` tracked poses within 1e-6, the rest within 5 cm of the oracle's."""
    raw = open(out + "result.bin", "rb").read()
    status, frames, tracked, inits, keyframes, sstat = struct.unpack("<6i", raw[:24])
    c = orc.counters()
    assert (status, frames, tracked, inits) == (c["status"], c["frames"], c["tracked"], c["initializations"])
    assert status == 1 and inits == 1 and tracked >= min_tracked and sstat == 0
    got = replay.read_tum(out + "fornt_end.txt")
    ref = np.array([ln.split() for ln in orc.tum().splitlines()[1:]], dtype=np.float64)
    assert got.shape == ref.shape == (tracked, 8) and np.array_equal(got[:, 0], ref[:, 0])
    err = np.abs(got[:, 1:] - ref[:, 1:]).max(axis=1) / max(1.0, np.abs(ref[:, 1:]).max())
    assert err[:lead].max() <= 1e-6, err[:lead]
    assert np.abs(got[:, 1:4] - ref[:, 1:4]).max() < 0.05
    # tracked number: the first pose of the free-running trajectory that is further than 1e-6 from the oracle twin's —
    # how long two correct implementations stay together before the driver's own sensitivity takes over
    beyond = np.nonzero(err > 1e-6)[0]
    print("free-running replay: %d tracked poses, first pose beyond 1e-6: %s (worst %.2e, position %.2e m)"
          % (tracked, int(beyond[0]) if len(beyond) else "none", float(err.max()), float(np.abs(got[:, 1:4] - ref[:, 1:4]).max())))
    return frames, tracked, got


def _teacher_forced_tracking(liw, pyoracle, prm, caps, tol=1e-6):
    """Every tracking solve of the oracle's replay re-run on the MI355X from the oracle's own input (window + carried prior):
    lvio_2d::solver::solve (TRACK topology) then ::marginalization; states, iteration count, termination, Delta_H / Delta_g and the
    new prior must match.  This is the per-solve parity statement for 30- / 50-frame tracking windows built from real line matches."""
    from parity_util import rel_inf
    slv = liw.Solver(prm)
    worst = dict(state=0.0, dH=0.0, dg=0.0, prior=0.0)
    referee = []
    for k, c in enumerate(caps):
        w = liw.Window(c)
        slv.set_prior((c["prior_X"], c["prior_J"].reshape(15, 15), c["prior_R"]) if c["has_prior"] else None)
        slv.set_window(w)
        s = slv.solve()
        assert (s["iterations"], s["termination"]) == (c["iterations"], c["termination"]), (k, c["n"], s, c["iterations"], c["termination"])
        e = rel_inf(w["states"].reshape(-1), c["states_after"])
        # The bar is 1e-6 on EVERY solve.  A solve cut off by the iteration cap (termination 4) is stopped in the middle of a crawl along the
        # ground_factor_q cone, where round-off differences between two correct implementations grow ~2.5x per iteration beyond iteration
        # ~35: one that misses the bar there is judged PER SOLVE by the oracle against itself — the same solve with its
        # pre-integrated IMU means scaled by 1 + 1e-13 N(0,1), the size of a round-off difference (the referee of
        # test_soak_outliers_are_within_the_problems_own_round_off_sensitivity).  No blanket slack (it was 10x until round 4).
        if e > tol:
            assert c["termination"] == 4, (k, c["n"], c["termination"], e)
            sens = oracle_round_off_sensitivity(pyoracle, prm, c)
            referee.append((k, e, sens))
            print("replay solve %d (n = %d, cap-terminated): product vs oracle %.2e, oracle vs itself with 1e-13 IMU noise %.2e" % (k, c["n"], e, sens))
            # absolute ceiling next to the referee: a chaotic oracle (sens ~ 1e-3) must not make ANY product result acceptable
            assert e <= min(3.0 * sens, 1e-4), (k, c["n"], e, sens)
        # marginalise at the oracle's post-solve point so that the comparison is at one linearisation point
        w["states"].reshape(-1)[:] = c["states_after"]
        w["match_pose"].reshape(-1)[:] = c["match_after"]
        slv.set_window(w)
        m = slv.marginalization()
        eH, eg = rel_inf(m["Delta_H"].reshape(-1), c["Delta_H"]), rel_inf(m["Delta_g"], c["Delta_g"])
        Xg, Jg, _ = slv.get_prior()
        Jo = c["post_J"].reshape(15, 15)
        eP = max(rel_inf(Xg, c["post_X"]), rel_inf(Jg.T @ Jg, Jo.T @ Jo))
        assert eH <= tol and eg <= tol and eP <= tol, (k, eH, eg, eP)
        worst = dict(state=max(worst["state"], e), dH=max(worst["dH"], eH), dg=max(worst["dg"], eg), prior=max(worst["prior"], eP))
    worst["solves"], worst["beyond_1e-6_judged_by_the_referee"] = len(caps), [(k, float("%.2e" % e), float("%.2e" % sn)) for k, e, sn in referee]
    return worst


def oracle_round_off_sensitivity(pyoracle, prm, c, trials=3):
    """how far the ORACLE's tracking solve of the captured input `c` moves when its pre-integrated IMU means are scaled by 1 + 1e-13 N(0,1)"""
    from parity_util import rel_inf

    def run(win):
        o = pyoracle.Oracle(prm)
        o.set_prior((win["prior_X"], win["prior_J"].reshape(15, 15), win["prior_R"]) if win["has_prior"] else None)
        w = pyoracle.Window(win)
        o.solve(w)
        return w["states"].reshape(-1).copy()
    x0 = run(c)
    rp, sens = np.random.default_rng(7), 0.0
    for _ in range(trials):
        alt = dict(c)
        alt["imu_X"] = np.asarray(c["imu_X"]) * (1.0 + 1e-13 * rp.standard_normal(np.asarray(c["imu_X"]).shape))
        sens = max(sens, rel_inf(run(alt), x0))
    return sens


@pytest.mark.parametrize("keep,duration,seed", [(29, 6.0, 1), (49, 8.0, 2)])
def test_keep_n_window_replay_matches_oracle(liw, synth, pyoracle, tmp_path, keep, duration, seed):
    """BASELINE configs C3 / C5 in shape: the explicit keep-N window policy of SURVEY 8 f3 — the window keeps `keep`
    frames after every tracking solve, so lvio_2d::solver::solve / marginalization run on 30- / 50-frame windows (TRACK topology:
    solver.cpp:631-820) for every laser frame once the window has filled.  (a) every tracking solve of the oracle twin's replay
    reproduced on the MI355X from the same input (1e-6, identical iteration counts); (b) the free-running C++ driver against the
    free-running oracle twin: identical state machine, leading poses within 1e-6."""
    import importlib
    replay = importlib.import_module("2dliw-slam_amd.replay")
    prm = synth.office_params()
    lp = liw.laser.office_laser_params(prm)
    msgs, truth = replay.make_log(prm, duration=duration, seed=seed)
    replay.write_log(str(tmp_path / "log.bin"), msgs)
    out = str(tmp_path) + "/"
    r = subprocess.run([build_replay(liw), str(tmp_path / "log.bin"), out, "--keep", str(keep)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    orc = oracle_replay(pyoracle, prm, lp, msgs, keep=keep, capture=True)
    frames, tracked, got = _check_free_running(out, orc, replay, min_tracked=keep + 5, lead=8)
    assert frames == keep                                   # the window really holds `keep` frames (keep + 1 at solve time)
    T = truth.T_w_o(got[-1, 0])
    assert np.linalg.norm(got[-1, 1:3] - T[:2, 3]) < 0.15
    caps = orc.captures()
    assert len(caps) == tracked and max(c["n"] for c in caps) == keep + 1
    worst = _teacher_forced_tracking(liw, pyoracle, prm, caps)
    print("keep=%d: %d tracking solves (up to %d frames, %d laser blocks) reproduced: %s" % (keep, len(caps), keep + 1, max(c["L"] for c in caps), worst))


@pytest.mark.parametrize("name,kw,duration", [
    ("parked_2s_then_drive", dict(motion="standstill_then_go", t_go=2.0), 6.0),
    ("stop_2s_while_tracking_noisy_odometry", dict(motion="stop_and_go", t_stop=3.0, pause=2.0), 7.0),
    ("stop_2s_while_tracking_identical_odometry", dict(motion="stop_and_go", t_stop=3.0, pause=2.0, odom_noise=0.0), 7.0)])
def test_replay_with_a_standstill(liw, synth, pyoracle, tmp_path, name, kw, duration):
    """Logs with a 2 s standstill.  Parked at the start: the INITIALIZING gate drops the scans of a robot at rest
    (reference trajectory.cpp:163), the state machine must agree.  A stop while TRACKING: the tracking solves of those frames take the
    stationary arms of wheel_odom_factor (wheel_factor.h:45/:58/:63; asserted from the captured inputs) — every one re-run on the
    MI355X from the oracle's input: states 1e-6, iteration counts, Delta_H / Delta_g, the new prior."""
    import importlib
    from parity_util import wheel_arms
    replay = importlib.import_module("2dliw-slam_amd.replay")
    prm = synth.office_params()
    lp = liw.laser.office_laser_params(prm)
    msgs, truth = replay.make_log(prm, duration=duration, seed=4, **kw)
    replay.write_log(str(tmp_path / "log.bin"), msgs)
    out = str(tmp_path) + "/"
    r = subprocess.run([build_replay(liw), str(tmp_path / "log.bin"), out], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    orc = oracle_replay(pyoracle, prm, lp, msgs, capture=True)
    frames, tracked, got = _check_free_running(out, orc, replay, min_tracked=20, lead=8)
    caps = orc.captures()
    assert len(caps) == tracked
    arms = []
    for c in caps:
        d = dict(c)
        d["states"], d["wheel_T"] = np.asarray(c["states"]).reshape(c["n"], 15), np.asarray(c["wheel_T"]).reshape(c["n"] - 1, 12)
        a = wheel_arms(synth, prm, d, c["n"] - 2)
        arms.append((a["moving45"], a["moving63"]))
    at_rest = sum(1 for a in arms if not a[1])
    if kw["motion"] == "stop_and_go":
        assert at_rest >= 15, arms                                    # ~2 s of 10 Hz scans with |oq| < 1e-3
        if kw.get("odom_noise", 1.0) == 0.0:
            assert sum(1 for a in arms if a == (False, False)) >= 15, arms   # identical odometry readings: o_len = 0 too
    worst = _teacher_forced_tracking(liw, pyoracle, prm, caps)
    print("%s: %d tracking solves (%d in the stationary arms) reproduced: %s" % (name, len(caps), at_rest, worst))


def test_c5_shape_replay_with_pose_graph_backend(liw, synth, pyoracle, tmp_path):
    """BASELINE C5 end to end in shape: 50-frame tracking windows, key frames leaving the window go to the back-end
    (include/lvio_2d_keyframe_manager.hpp: sequential edges, loop edges from a schedule standing in for loop detection,
    liw_posegraph_solve on the MI355X, modify_delta_tf, update_other_frame).  (a) the free-running C++ driver + back-end against the
    oracle twins (oracle/trajectory.h + oracle/keyframe_manager.h): identical state machine and back-end counters, bounded pose
    difference; (b) the back-end alone on the ORACLE's key frames (teacher forcing): every key-frame pose and modify_delta_tf within
    1e-6.  Reference: keyframe_manager.cpp:407 (add_keyframe), :419-482, :722-838."""
    import importlib
    replay = importlib.import_module("2dliw-slam_amd.replay")
    pgm = importlib.import_module("2dliw-slam_amd.posegraph")
    prm = synth.office_params()
    lp = liw.laser.office_laser_params(prm)
    pg = pgm.office_pg_params()
    msgs, truth = replay.make_log(prm, duration=11.0, seed=3)
    replay.write_log(str(tmp_path / "log.bin"), msgs)
    keep = 49
    # pass 1 (oracle, empty schedule): which key frames reach the back-end, when, and with which tracking pose
    first = oracle_replay(pyoracle, prm, lp, msgs, keep=keep, backend=(pg, [], 0.3))
    b1 = first.backend()
    times, kf_poses = b1["times"], b1["poses"]          # no loop, no solve: poses = tracking poses
    assert len(times) >= 12 and b1["solves"] == 0
    rng = np.random.default_rng(5)

    def rel_tf(i, j):       # tf12 of a loop edge (index1 = i newer, index2 = j older): T_i^-1 T_j from the simulated truth + a small error
        Ti, Tj = truth.T_w_i(times[i]), truth.T_w_i(times[j])
        E = np.eye(4)
        E[:3, :3] = synth.exp_so3(rng.normal(0.0, 4e-3, 3))
        E[:3, 3] = rng.normal(0.0, 0.02, 3)
        T = synth.inv_se3(Ti) @ Tj @ E
        return np.concatenate([T[:3, :3].reshape(9), T[:3, 3]])
    n_kf = len(times)
    loops = [(n_kf // 2, 1, rel_tf(n_kf // 2, 1)), (n_kf - 2, 3, rel_tf(n_kf - 2, 3))]
    with open(str(tmp_path / "loops.bin"), "wb") as f:
        f.write(struct.pack("<i", len(loops)))
        for trig, older, tf in loops:
            f.write(struct.pack("<ii", trig, older))
            f.write(np.asarray(tf, dtype=np.float64).tobytes())
    exe = build_replay(liw)

    def read_backend(d):
        raw = open(d + "backend.bin", "rb").read()
        cnt = struct.unpack("<4i", raw[:16])
        arr = np.frombuffer(raw[16:], dtype=np.float64)
        return cnt, arr[:12], arr[12:18], arr[18:].reshape(-1, 6)
    # (a) free running
    out = str(tmp_path) + "/"
    r = subprocess.run([exe, str(tmp_path / "log.bin"), out, "--keep", str(keep), "--loops", str(tmp_path / "loops.bin"), "--solve-period", "0.3"],
                       capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    orc = oracle_replay(pyoracle, prm, lp, msgs, keep=keep, backend=(pg, loops, 0.3))
    _check_free_running(out, orc, replay, min_tracked=keep + 5, lead=8)
    bo = orc.backend()
    (nk, nl, ns, its), modify, cur, poses = read_backend(out)
    assert (nk, nl, ns) == (bo["keyframes"], bo["loops"], bo["solves"]) and nl == 2 and ns >= 2 and nk == n_kf
    assert np.abs(poses[:, :3] - bo["poses"][:, :3]).max() < 0.05 and np.abs(cur[:3] - bo["current"][:3]).max() < 0.05
    assert np.abs(modify - np.concatenate([np.eye(3).reshape(9), np.zeros(3)])).max() > 1e-4      # the loop closures moved the map frame
    be = replay.read_tum(out + "back_end.txt")
    assert be.shape == (nk, 8) and np.allclose(be[:, 0], bo["times"], atol=1e-9)
    # (b) the back-end alone on the oracle's key frames
    with open(str(tmp_path / "kf.bin"), "wb") as f:
        f.write(struct.pack("<i", n_kf))
        for t, x in zip(times, kf_poses):
            f.write(struct.pack("<7d", t, *x))
    d2 = tmp_path / "be"
    d2.mkdir()
    # LM cap 20 on both sides: with the cone-shaped ground_factor_q the pose-graph LM path is round-off sensitive beyond ~35 iterations
    # (tests/test_gpu_posegraph.py), so the capped run is the per-iteration parity statement
    r = subprocess.run([exe, "--backend-only", str(tmp_path / "kf.bin"), str(d2) + "/", "--loops", str(tmp_path / "loops.bin"), "--solve-period", "0.3",
                        "--pg-iters", "20"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    ref = pyoracle.backend_run(pyoracle.Oracle(prm), pg, times, kf_poses, loops, solve_period=0.3, max_iterations=20)
    (nk, nl, ns, its), modify, _, poses = read_backend(str(d2) + "/")
    assert (nk, nl, ns, its) == (ref["keyframes"], ref["loops"], ref["solves"], ref["iterations"]), ((nk, nl, ns, its), ref)
    scale = max(1.0, np.abs(ref["poses"]).max())
    # The pose-graph LM path is round-off sensitive on some instances (tools/pg_diag.py on this one: product and oracle 1e-13 apart through
    # iteration 8, 4e-10 at 12, 1e-8 at 16, 1e-5 at 20 — the instance changes whenever the front-end replay that feeds it does).  The bar is
    # therefore the problem's OWN sensitivity: the oracle against itself with the key-frame poses perturbed by 1e-13 relative (the size of
    # the round-off difference between two correct implementations) must move at least a third as far as the product is from the oracle.
    rngp = np.random.default_rng(11)
    sens = 0.0
    for _ in range(3):
        kp = [np.asarray(x) * (1.0 + 1e-13 * rngp.standard_normal(6)) for x in kf_poses]
        alt = pyoracle.backend_run(pyoracle.Oracle(prm), pg, times, kp, loops, solve_period=0.3, max_iterations=20)
        sens = max(sens, float(np.abs(alt["poses"] - ref["poses"]).max()))
    err = float(np.abs(poses - ref["poses"]).max())
    print("back-end teacher forcing: product vs oracle %.2e, oracle vs oracle with 1e-13 input noise %.2e" % (err, sens))
    assert err <= max(1e-6 * scale, 3.0 * sens), (err, sens)
    assert np.abs(modify - ref["modify_delta_tf"]).max() <= max(1e-6, 3.0 * sens)
    # ... and with the LM cut at 8 iterations, before the amplification sets in, the plain bar holds
    d3 = tmp_path / "be8"
    d3.mkdir()
    r = subprocess.run([exe, "--backend-only", str(tmp_path / "kf.bin"), str(d3) + "/", "--loops", str(tmp_path / "loops.bin"), "--solve-period", "0.3",
                        "--pg-iters", "8"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    ref8 = pyoracle.backend_run(pyoracle.Oracle(prm), pg, times, kf_poses, loops, solve_period=0.3, max_iterations=8)
    (_, _, _, its8), modify8, _, poses8 = read_backend(str(d3) + "/")
    assert its8 == ref8["iterations"] and np.abs(poses8 - ref8["poses"]).max() <= 1e-9 * scale and np.abs(modify8 - ref8["modify_delta_tf"]).max() <= 1e-9
