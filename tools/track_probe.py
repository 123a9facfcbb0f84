"""Break the steady-state tracking frame (bench.py `tracking_frame_latency`) into its host-visible parts: liw_set_window, liw_solve,
liw_marginalize, each timed around the synchronous C-ABI call.  usage: python tools/track_probe.py [REPS]   (on the MI355X box)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prm = synth.office_params()
hp = liw.HostPreint(prm)
d3 = synth.make_window(hp, prm, seed=515, n=3, L=120, laser_on_frame0=False)


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
t = np.zeros(4)
its = 0
for rep in range(reps + 5):
    slv.set_prior(None)
    slv.set_window(liw.Window(sub(0)))
    slv.solve()
    slv.marginalization()
    t0 = time.perf_counter()
    w12 = liw.Window(sub(1))
    t1 = time.perf_counter()
    slv.set_window(w12)
    t2 = time.perf_counter()
    sg = slv.solve()
    t3 = time.perf_counter()
    slv.marginalization()
    t4 = time.perf_counter()
    if rep >= 5:
        t += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
        its = sg["iterations"]
t *= 1e6 / reps
print("tracking frame: Window() %.1f us | set_window %.1f us | solve %.1f us (%d LM iterations) | marginalization %.1f us | total w/o Window() %.1f us"
      % (t[0], t[1], t[2], its, t[3], t[1] + t[2] + t[3]))
