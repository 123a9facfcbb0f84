"""Synthetic sensor logs for the front-end driver: a robot on synth._Truth's circle inside a walled
room, IMU / wheel-odometry / LaserScan messages with distinct stamps, written as a flat binary log that
tools/replay_log.cpp (lvio_2d::trajectory of include/lvio_2d_trajectory.hpp) replays.

Flat log (little-endian), a sequence of records  int32 type, then
  0 imu         float64 time, acc[3], gyro[3]
  1 wheel_odom  float64 time, R[9] (row-major), t[3]        (pose of the base in the odometry frame)
  3 laser_scan  float64 time, float32 angle_min, angle_increment, time_increment, int32 n, float32 ranges[n]
"""
import struct

import numpy as np

from . import synth


def replay_room():
    """Walls around the truth circle (centre (0, 5), radius 5): outer box, inner box, a few stubs."""
    segs = []

    def box(x0, y0, x1, y1):
        return [((x0, y0), (x1, y0)), ((x1, y0), (x1, y1)), ((x1, y1), (x0, y1)), ((x0, y1), (x0, y0))]
    segs += box(-8.0, -3.0, 8.0, 13.0)
    segs += box(-2.0, 3.2, 1.6, 6.9)
    segs += [((-8.0, 2.0), (-6.6, 2.0)), ((8.0, 7.5), (6.5, 7.9)), ((-3.0, 13.0), (-3.0, 11.6)), ((3.5, -3.0), (3.9, -1.7)),
             ((-0.5, 6.9), (-0.9, 8.2)), ((1.6, 4.0), (2.7, 4.4))]
    return [(np.array(a, dtype=np.float64), np.array(b, dtype=np.float64)) for a, b in segs]


def cast_scan_moving(segs, pose_of_time, t0, n_rays, fov, time_increment, noise, rng, max_range=30.0):
    """Ray i is cast from the laser pose at t0 + i * time_increment (a real spinning lidar): float32 ranges."""
    angle_min, inc = -fov / 2, fov / (n_rays - 1)
    P = np.stack([s[0] for s in segs])
    E = np.stack([s[1] - s[0] for s in segs])
    ranges = np.full(n_rays, np.inf, dtype=np.float32)
    # poses sampled on a coarse grid and interpolated per ray (the motion within one scan is a few centimetres)
    K = 16
    ts = t0 + np.linspace(0.0, time_increment * (n_rays - 1), K)
    ox, oy, yaw = np.zeros(K), np.zeros(K), np.zeros(K)
    for k, t in enumerate(ts):
        T = pose_of_time(t)
        ox[k], oy[k], yaw[k] = T[0, 3], T[1, 3], np.arctan2(T[1, 0], T[0, 0])
    yaw = np.unwrap(yaw)
    ti = t0 + time_increment * np.arange(n_rays)
    if time_increment > 0:
        o = np.stack([np.interp(ti, ts, ox), np.interp(ti, ts, oy)], axis=1)
        a = np.interp(ti, ts, yaw) + angle_min + inc * np.arange(n_rays)
    else:
        o = np.tile([ox[0], oy[0]], (n_rays, 1))
        a = yaw[0] + angle_min + inc * np.arange(n_rays)
    d = np.stack([np.cos(a), np.sin(a)], axis=1)
    den = d[:, None, 0] * E[None, :, 1] - d[:, None, 1] * E[None, :, 0]
    w = P[None, :, :] - o[:, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (w[:, :, 0] * E[None, :, 1] - w[:, :, 1] * E[None, :, 0]) / den
        u = (w[:, :, 0] * d[:, None, 1] - w[:, :, 1] * d[:, None, 0]) / den
    ok = (np.abs(den) > 1e-12) & (t > 0.05) & (u >= 0.0) & (u <= 1.0)
    t = np.where(ok, t, np.inf)
    best = t.min(axis=1)
    hit = best < max_range
    ranges[hit] = (best[hit] + rng.normal(0.0, noise, int(hit.sum()))).astype(np.float32)
    return ranges, np.float32(angle_min), np.float32(inc)


def make_log(prm, duration=3.0, seed=0, imu_rate=200.0, wheel_rate=20.0, laser_rate=10.0, n_rays=720, scan_time=0.04,
             motion="arc", odom_noise=2e-4, **truth_kw):
    """-> list of messages (dicts with 'type', 'time', ...), strictly increasing distinct stamps.
    motion / truth_kw: synth._Truth (e.g. "standstill_then_go" with t_go, "stop_and_go" with t_stop / pause); odom_noise = 0 gives the
    bit-identical odometry readings of a robot at rest."""
    rng = np.random.default_rng(seed)
    tr = synth._Truth(prm, motion=motion, **truth_kw)
    room = replay_room()
    bias = rng.normal(0.0, 1e-3, 6)
    msgs = []
    for k in range(int(duration * imu_rate)):
        t = k / imu_rate + 1e-4
        acc, gyro = tr.imu(t)
        msgs.append(dict(type=0, time=t, acc=acc + bias[:3] + rng.normal(0, 0.01, 3), gyro=gyro + bias[3:] + rng.normal(0, 0.001, 3)))
    for k in range(int(duration * wheel_rate)):
        t = k / wheel_rate + 3e-4
        T = tr.T_w_o(t)
        msgs.append(dict(type=1, time=t, R=T[:3, :3].copy(), t=T[:3, 3] + rng.normal(0, 1.0, 3) * odom_noise))
    T_i_l = tr.T_i_l
    tinc = scan_time / n_rays
    for k in range(int(duration * laser_rate)):
        t = 0.12 + k / laser_rate + 7e-4
        if t + scan_time >= duration:
            break
        rg, amin, inc = cast_scan_moving(room, lambda tt: tr.T_w_i(tt) @ T_i_l, t, n_rays, 2 * np.pi * 0.75, tinc, 0.004, rng)
        msgs.append(dict(type=3, time=t, angle_min=amin, angle_increment=inc, time_increment=np.float32(tinc), ranges=rg))
    msgs.sort(key=lambda m: m["time"])
    return msgs, tr


def write_log(path, msgs):
    with open(path, "wb") as f:
        for m in msgs:
            f.write(struct.pack("<i", m["type"]))
            if m["type"] == 0:
                f.write(struct.pack("<7d", m["time"], *m["acc"], *m["gyro"]))
            elif m["type"] == 1:
                f.write(struct.pack("<13d", m["time"], *np.asarray(m["R"]).reshape(9), *m["t"]))
            else:
                f.write(struct.pack("<dfffi", m["time"], float(m["angle_min"]), float(m["angle_increment"]), float(m["time_increment"]), len(m["ranges"])))
                f.write(np.asarray(m["ranges"], dtype=np.float32).tobytes())


def read_log(path):
    """the messages of a flat log written by write_log / rosbag_reader.bag_to_flatlog, in file order"""
    msgs = []
    with open(path, "rb") as f:
        raw = f.read()
    o = 0
    while o + 4 <= len(raw):
        (ty,) = struct.unpack_from("<i", raw, o)
        o += 4
        if ty == 0:
            v = struct.unpack_from("<7d", raw, o); o += 56
            msgs.append(dict(type=0, time=v[0], acc=np.array(v[1:4]), gyro=np.array(v[4:7])))
        elif ty == 1:
            v = struct.unpack_from("<13d", raw, o); o += 104
            msgs.append(dict(type=1, time=v[0], R=np.array(v[1:10]).reshape(3, 3), t=np.array(v[10:13])))
        else:
            t, amin, inc, tinc, nr = struct.unpack_from("<dfffi", raw, o); o += 24
            rg = np.frombuffer(raw, dtype=np.float32, count=nr, offset=o).copy(); o += 4 * nr
            msgs.append(dict(type=ty, time=t, angle_min=np.float32(amin), angle_increment=np.float32(inc), time_increment=np.float32(tinc), ranges=rg))
    return msgs


def read_tum(path):
    rows = [ln.split() for ln in open(path) if ln.strip() and not ln.startswith("#")]
    return np.array(rows, dtype=np.float64).reshape(-1, 8)
