"""CPU timing of the laser front-end (host C++ of libliw_window.so): spawn_scan on a synthetic scan and do_match between two
consecutive scans (ctypes binding included).  usage: python tools/bench_laser.py [reps]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    prm = synth.office_params()
    lp_dict = liw.laser.office_laser_params(prm)
    lp = liw.laser.laser_params_struct(lp_dict)   # built once: the timing below is the library, not the binding
    room = liw.laser.room_segments(1)
    T_il = np.array(synth.normalize_extrinsic(prm["T_imu_to_laser"])).reshape(4, 4)
    pts = []
    for k in range(2):
        T = np.eye(4)
        T[:3, :3] = synth.exp_so3(np.array([0, 0, 0.3 + 0.04 * k]))
        T[:2, 3] = [0.2 + 0.1 * k, -0.1]
        rg, amin, inc = liw.laser.cast_scan(room, T @ T_il, seed=k)
        pts.append(liw.laser.laser_to_points(rg, amin, inc, 0.0, 0.0)[0])
    out = {"points_per_scan": int(len(pts[0])), "reps": reps}
    t0 = time.perf_counter()
    for _ in range(reps):
        s = liw.laser.Scan.spawn(lp, pts[0])
    out["spawn_scan_us"] = (time.perf_counter() - t0) / reps * 1e6
    s1, s2 = liw.laser.Scan.spawn(lp, pts[0]), liw.laser.Scan.spawn(lp, pts[1])
    p1, q1, p2, q2 = [0.2, -0.1, 0], [0, 0, 0.3], [0.3, -0.1, 0], [0, 0, 0.34]
    t0 = time.perf_counter()
    for _ in range(reps):
        m = liw.laser.do_match(lp, s1, s2, p1, q1, p2, q2)
    out["do_match_us"] = (time.perf_counter() - t0) / reps * 1e6
    out["lines"], out["matches"] = int(s.lines().shape[0]), int(len(m))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
