"""The C++ host mirror (include/lvio_2d_solver.hpp: lvio_2d::solver / frame_info / laser_match with the reference's
names) driven by a small C++ program (tests/cpp/host_api_driver.cpp) linked against libliw_window.so."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "host_api_driver")


def build_driver(liw):
    src = os.path.join(ROOT, "tests", "cpp", "host_api_driver.cpp")
    hdrs = [os.path.join(ROOT, "include", h) for h in ("lvio_2d_solver.hpp", "liw_window.h")]
    if (not os.path.exists(EXE)) or any(os.path.getmtime(p) > os.path.getmtime(EXE) for p in [src] + hdrs):
        libdir = os.path.dirname(liw.LIB_PATH)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                               "-L", libdir, "-lliw_window", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def dump_window(path, d):
    n, L = int(d["n"]), int(np.asarray(d["laser_frame"]).shape[0])
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", n, L))
        f.write(np.asarray(d["states"], dtype=np.float64).tobytes())
        f.write(np.asarray(d["laser_frame"], dtype=np.int32).tobytes())
        f.write(np.asarray(d["laser_pts"], dtype=np.float64).tobytes())
        f.write(np.asarray(d["match_pose"], dtype=np.float64).tobytes())
        f.write(np.asarray(d["has_match"], dtype=np.uint8).tobytes())
        for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
            f.write(np.asarray(d[k], dtype=np.float64).tobytes())


def test_cpp_driver_builds_and_fails_loudly_without_gpu(liw, synth, pyoracle, tmp_path):
    import torch
    exe = build_driver(liw)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    prm = synth.office_params()
    d = synth.make_window(pyoracle.Oracle(prm), prm, seed=3, n=3, L=9)
    dump_window(str(tmp_path / "w.bin"), d)
    r = subprocess.run([exe, str(tmp_path / "w.bin"), str(tmp_path / "o.bin"), "init"], capture_output=True)
    assert r.returncode == 19, r          # LIW_ENODEV: no CPU fallback behind the C++ class either
    assert b"no usable gfx950 device" in r.stderr or b"no HIP device" in r.stderr


@pytest.mark.gpu
def test_cpp_solver_class_matches_oracle(liw, synth, pyoracle, tmp_path):
    exe = build_driver(liw)
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, L = 8, 120
    d = synth.make_window(orc, prm, seed=33, n=n, L=L)
    dump_window(str(tmp_path / "w.bin"), d)
    subprocess.check_call([exe, str(tmp_path / "w.bin"), str(tmp_path / "o.bin"), "init"])
    raw = open(str(tmp_path / "o.bin"), "rb").read()
    states = np.frombuffer(raw[:n * 15 * 8], dtype=np.float64).reshape(n, 15)
    mp = np.frombuffer(raw[n * 15 * 8:n * 27 * 8], dtype=np.float64).reshape(n, 12)
    sqrt_H = np.frombuffer(raw[n * 27 * 8:n * 27 * 8 + 288], dtype=np.float64).reshape(6, 6)
    iters = struct.unpack("<i", raw[-4:])[0]
    wo = pyoracle.Window(d)
    orc.set_prior(None)
    orc.init_solve(wo)
    assert iters == orc.summary()["iterations"]
    assert np.abs(states - wo["states"].reshape(n, 15)).max() <= 1e-6 * np.abs(wo["states"]).max()
    m = np.asarray(d["has_match"]).astype(bool)
    assert np.abs(mp[m] - wo["match_pose"].reshape(n, 12)[m]).max() <= 1e-6 * np.abs(wo["match_pose"]).max()
    so = orc.marginalization(wo)
    assert np.abs(sqrt_H.T @ sqrt_H - so.T @ so).max() <= 1e-6 * max(1.0, np.abs(so.T @ so).max())


def test_cpp_laser_manager_class_matches_python_binding(liw, synth, tmp_path):
    """include/lvio_2d_laser.hpp (lvio_2d::laser_manager / scan with the reference's names) against the ctypes binding of the
    same C ABI and, through tests/test_laser_frontend.py, against the oracle.  CPU: the front-end is host code."""
    src = os.path.join(ROOT, "tests", "cpp", "laser_api_driver.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "laser_api_driver")
    libdir = os.path.dirname(liw.LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lliw_window", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    prm = synth.office_params()
    lp = liw.laser.office_laser_params(prm)
    room = liw.laser.room_segments(4)
    T_il = np.array(synth.normalize_extrinsic(prm["T_imu_to_laser"])).reshape(4, 4)
    poses = [np.array([0.1, 0.2, 0.0, 0.0, 0.0, 0.2]), np.array([0.22, 0.25, 0.0, 0.0, 0.0, 0.26])]
    pts = []
    for k, x in enumerate(poses):
        T = np.eye(4)
        T[:3, :3] = synth.exp_so3(x[3:6])
        T[:3, 3] = x[0:3]
        rg, amin, inc = liw.laser.cast_scan(room, T @ T_il, seed=40 + k)
        pts.append(liw.laser.laser_to_points(rg, amin, inc, 0.0, 0.0)[0])
    with open(str(tmp_path / "in.bin"), "wb") as f:
        f.write(struct.pack("<ii", len(pts[0]), len(pts[1])))
        f.write(np.concatenate(poses).tobytes())
        f.write(pts[0].tobytes()); f.write(pts[1].tobytes())
    subprocess.check_call([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = open(str(tmp_path / "out.bin"), "rb").read()
    nl1, nl2, nm = struct.unpack("<iii", raw[:12])
    arr = np.frombuffer(raw[12:], dtype=np.float64)
    l1, l2 = arr[:nl1 * 6].reshape(nl1, 6), arr[nl1 * 6:(nl1 + nl2) * 6].reshape(nl2, 6)
    mt, pose = arr[(nl1 + nl2) * 6:(nl1 + nl2) * 6 + nm * 12].reshape(nm, 12), arr[-12:]
    s1, s2 = liw.laser.Scan.spawn(lp, pts[0], 0.0), liw.laser.Scan.spawn(lp, pts[1], 0.1)
    mgr = liw.laser.LaserManager(lp)
    mgr.add_scan(s1, poses[0][0:3], poses[0][3:6])
    m = mgr.match_with_front(s2, poses[1][0:3], poses[1][3:6])
    assert np.array_equal(l1, s1.lines()[:, :6]) and np.array_equal(l2, s2.lines()[:, :6])
    assert nm == len(m) >= 4 and np.array_equal(mt, m.pts) and np.array_equal(pose, m.pose)


def test_replay_tool_builds_and_fails_loudly_without_gpu(liw, synth, tmp_path):
    """tools/replay_log.cpp (lvio_2d::trajectory + dispatch_queue of include/lvio_2d_trajectory.hpp) compiles against the public
    headers only; without an MI355X the first solve fails with LIW_ENODEV — the front-end driver has no CPU estimator to fall
    back to."""
    import importlib
    import torch
    replay = importlib.import_module("2dliw-slam_amd.replay")
    src = os.path.join(ROOT, "tools", "replay_log.cpp")
    exe = os.path.join(ROOT, "tools", "replay_log")
    libdir = os.path.dirname(liw.LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lliw_window", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    msgs, _ = replay.make_log(synth.office_params(), duration=2.5, seed=3)
    replay.write_log(str(tmp_path / "log.bin"), msgs)
    r = subprocess.run([exe, str(tmp_path / "log.bin"), str(tmp_path) + "/"], capture_output=True)
    assert r.returncode == 19, (r.returncode, r.stderr.decode())
    assert b"no usable gfx950 device" in r.stderr or b"no HIP device" in r.stderr


def _track_tool(liw):
    src = os.path.join(ROOT, "tools", "track_frame_cpp.cpp")
    exe = os.path.join(os.path.dirname(liw.LIB_PATH), "build", "track_frame_cpp_test")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir = os.path.dirname(liw.LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L", libdir, "-lliw_window", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_tracking_frame_tool_builds_and_fails_loudly_without_gpu(liw, synth, pyoracle, tmp_path):
    """tools/track_frame_cpp.cpp (bench.py's C++-caller leg of tracking_frame_latency): builds against the mirror header; no CPU fallback."""
    import torch
    exe = _track_tool(liw)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    prm = synth.office_params()
    d = synth.make_window(pyoracle.Oracle(prm), prm, seed=515, n=3, L=30, laser_on_frame0=False)
    dump_window(str(tmp_path / "w3.bin"), d)
    r = subprocess.run([exe, str(tmp_path / "w3.bin"), "2"], capture_output=True)
    assert r.returncode == 19, r


@pytest.mark.gpu
def test_cpp_tracking_frame_tool_runs_the_frame_the_python_mirror_runs(liw, synth, tmp_path):
    """Same window, same call sequence as bench.py's Python leg: the tool's last tracking solve takes the iterations liw.Solver takes."""
    exe = _track_tool(liw)
    prm = synth.office_params()
    hp = liw.HostPreint(prm)
    d3 = synth.make_window(hp, prm, seed=515, n=3, L=120, laser_on_frame0=False)
    dump_window(str(tmp_path / "w3.bin"), d3)
    r = subprocess.run([exe, str(tmp_path / "w3.bin"), "3"], capture_output=True, timeout=120)
    assert r.returncode == 0, r
    tok = r.stdout.decode().split()
    assert tok[0] == "ms_per_frame" and float(tok[1]) > 0.0

    def sub(lo):
        o = dict(d3)
        o["n"] = 2
        for k in ("states", "match_pose"):
            o[k] = np.asarray(d3[k]).reshape(3, -1)[lo:lo + 2].copy()
        o["has_match"] = np.asarray(d3["has_match"])[lo:lo + 2].copy()
        for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
            o[k] = np.asarray(d3[k])[lo:lo + 1].copy()
        m = (np.asarray(d3["laser_frame"]) >= lo) & (np.asarray(d3["laser_frame"]) < lo + 2)
        o["laser_frame"] = (np.asarray(d3["laser_frame"])[m] - lo).astype(np.int32)
        o["laser_pts"] = np.asarray(d3["laser_pts"])[m].copy()
        return o
    slv = liw.Solver(prm)
    slv.set_prior(None)
    slv.set_window(liw.Window(sub(0)))
    slv.solve()
    slv.marginalization()
    slv.set_window(liw.Window(sub(1)))
    sg = slv.solve()
    assert int(tok[3]) == sg["iterations"]
