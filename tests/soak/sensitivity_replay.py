"""Intrinsic sensitivity of the replay driver: the ORACLE against itself with every IMU sample perturbed by a relative 1e-13 (argument 3) (CPU only).
If the oracle's own trajectory moves as much as product-vs-oracle does in soak_replay.py, that difference is the algorithm's
round-off amplification (thresholded line matching + long LM crawls), not a defect.  usage: sensitivity_replay.py FIRST LAST"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
replay = importlib.import_module("2dliw-slam_amd.replay")
from oracle import pyoracle
pyoracle.build()


def run(prm, lp, msgs):
    orc = pyoracle.TrajectoryOracle(prm, lp)
    for m in msgs:
        if m["type"] == 0:
            orc.add_imu(m["time"], m["acc"], m["gyro"])
        elif m["type"] == 1:
            orc.add_wheel(m["time"], m["R"], m["t"])
        else:
            pts, ts = pyoracle.laser_to_points(m["ranges"], m["angle_min"], m["angle_increment"], m["time_increment"], m["time"])
            orc.add_laser(m["time"], pts, ts)
    return np.array([ln.split() for ln in orc.tum().splitlines()[1:]], dtype=np.float64).reshape(-1, 8)


EPS = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-13
prm = synth.office_params()
lp = liw.laser.office_laser_params(prm)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    d = 4.0 + (seed % 5)
    msgs, _ = replay.make_log(prm, duration=d, seed=seed)
    a = run(prm, lp, msgs)
    rng = np.random.default_rng(seed)
    msgs = [dict(m, gyro=m["gyro"] * (1.0 + EPS * rng.standard_normal(3)), acc=m["acc"] * (1.0 + EPS * rng.standard_normal(3))) if m["type"] == 0 else m for m in msgs]
    b = run(prm, lp, msgs)
    if a.shape != b.shape:
        print("seed %3d: frame counts differ %s %s" % (seed, a.shape, b.shape)); continue
    dp = np.abs(a[:, 1:] - b[:, 1:]).max(axis=1)
    lead = int(np.argmax(dp > 1e-6)) if (dp > 1e-6).any() else len(dp)
    print("seed %3d duration %.0f s: %3d frames, first %3d within 1e-6, max |dpose| %.2e  (oracle vs oracle with every IMU sample x (1 + %.0e N(0,1)))" % (seed, d, len(dp), lead, dp.max(), EPS))
