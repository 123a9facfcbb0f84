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


def oracle_replay(pyoracle, prm, lp, msgs):
    orc = pyoracle.TrajectoryOracle(prm, lp)
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
