#!/usr/bin/env python
"""Generates tests/golden/factors_golden.json — INDEPENDENT cross-check vectors for the oracle.

The reference (LittleDang/2DLIW-SLAM) ships no tests or golden vectors and cannot be built in this image
(no Ceres / Eigen / ROS), so parity is formally unpinned.  These vectors pin the oracle's MATH against a
second derivation written without looking at the oracle's code path: closed-form Rodrigues exp, arccos-based
log, residual formulas taken from the reference functors' definitions (src/factor/*_factor.h), and Jacobians
from PyTorch fp64 reverse-mode autograd (the oracle uses forward-mode dual numbers).  Run in the build
container only:  python tests/golden/make_golden.py
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_default_dtype(torch.float64)


def hat(w):
    z = torch.zeros((), dtype=w.dtype)
    return torch.stack([torch.stack([z, -w[2], w[1]]), torch.stack([w[2], z, -w[0]]), torch.stack([-w[1], w[0], z])])


def exp_so3(w):
    th = torch.sqrt((w * w).sum())
    K = hat(w)
    return torch.eye(3) + torch.sin(th) / th * K + (1.0 - torch.cos(th)) / (th * th) * (K @ K)


def log_so3(R):
    c = (torch.trace(R) - 1.0) * 0.5
    v = torch.stack([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = torch.sqrt((v * v).sum()) * 0.5
    th = torch.atan2(s, c)
    return v * (th / (2.0 * s))


def extrinsic(T16):
    T = np.asarray(T16, dtype=np.float64).reshape(4, 4)
    # the reference re-orthonormalises through a quaternion round trip; do the same with scipy-free math
    R = T[:3, :3]
    t = np.trace(R)
    assert t > 0
    s = np.sqrt(t + 1.0)
    w = 0.5 * s
    s = 0.5 / s
    x, y, z = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    Rn = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return torch.tensor(Rn), torch.tensor(T[:3, 3].copy())


def laser_res(prm, pts, x):
    """x = [p_i q_i p_j q_j]; reference src/factor/laser_factor.h:45-89."""
    Ril, til = prm["il"]
    pts = torch.tensor(pts)

    def world(p, q, pt):
        R = exp_so3(q) @ Ril
        t = exp_so3(q) @ til + p
        w = R @ pt + t
        return w[:2]
    A, B = world(x[0:3], x[3:6], pts[0:3]), world(x[0:3], x[3:6], pts[3:6])
    C1, C2 = world(x[6:9], x[9:12], pts[6:9]), world(x[6:9], x[9:12], pts[9:12])
    len1, len2 = torch.linalg.norm(pts[0:3] - pts[3:6]), torch.linalg.norm(pts[6:9] - pts[9:12])
    summ = torch.sqrt(torch.minimum(len1, len2) / 0.04)
    l = (B - A) / torch.linalg.norm(B - A)

    def dist(C):
        e = C - B
        return torch.abs(l[0] * e[1] - l[1] * e[0])
    w = summ / prm["line_to_line_sigma"]
    return torch.stack([w * dist(C1), w * dist(C2)])


def imu_res(prm, X, J, S, Dt, x):
    """x = [state_i(15) state_j(15)]; reference src/factor/imu_factor.h:13-89."""
    X, J, S = torch.tensor(X), torch.tensor(J).reshape(15, 15), torch.tensor(S).reshape(15, 15)
    pi, qi, vi, bai, bwi = x[0:3], x[3:6], x[6:9], x[9:12], x[12:15]
    pj, qj, vj, baj, bwj = x[15:18], x[18:21], x[21:24], x[24:27], x[27:30]
    g = torch.tensor([0.0, 0.0, prm["g"]])
    dba, dbw = bai - X[9:12], bwi - X[12:15]
    alpha = X[0:3] + J[0:3, 9:12] @ dba + J[0:3, 12:15] @ dbw
    beta = X[3:6] + J[3:6, 9:12] @ dba + J[3:6, 12:15] @ dbw
    gamma = X[6:9] + J[6:9, 12:15] @ dbw
    RiT = exp_so3(qi).T
    r_a = alpha - RiT @ (pj - pi + 0.5 * g * Dt * Dt - vi * Dt)
    r_b = beta - RiT @ (vj + g * Dt - vi)
    r_g = log_so3(exp_so3(gamma).T @ (RiT @ exp_so3(qj)))
    raw = torch.cat([r_a, r_b, r_g, baj - bai, bwj - bwi])
    return S @ raw


def wheel_res(prm, T12, sq9, x):
    """x = [p_i q_i p_j q_j]; reference src/factor/wheel_factor.h:12-73."""
    Riw, tiw = prm["iw"]
    T12, sq9 = torch.tensor(T12), torch.tensor(sq9)

    def tf(p, q):
        R = exp_so3(q)
        return R @ Riw, R @ tiw + p
    Ri, ti = tf(x[0:3], x[3:6])
    Rj, tj = tf(x[6:9], x[9:12])
    Rij, tij = Ri.T @ Rj, Ri.T @ (tj - ti)
    q = log_so3(Rij)
    op, oq = T12[9:12], log_so3(T12[0:9].reshape(3, 3))
    o_len, ln = torch.sqrt(op[0] ** 2 + op[1] ** 2), torch.sqrt(tij[0] ** 2 + tij[1] ** 2)
    if o_len > 1e-4 and ln > 1e-4:
        od, d = op[0:2] / o_len, tij[0:2] / ln
        angle = torch.asin(torch.abs(od[0] * d[1] - od[1] * d[0]))
    else:
        angle = ln
    r0 = sq9[0] * (ln if (ln < 1e-4 or o_len < 1e-4) else (o_len - ln))
    nq, noq = torch.linalg.norm(q), torch.linalg.norm(oq)
    r2 = sq9[8] * (nq if (nq < 1e-3 or noq < 1e-3) else (noq - nq))
    return torch.stack([r0, sq9[4] * angle, r2])


def ground_res(prm, x):
    """x = [p q]; reference src/factor/ground_factor.h:27-48, :59-82."""
    Riw, tiw = prm["iw"]
    R = exp_so3(x[3:6])
    h = (R @ tiw + x[0:3])[2]
    z = (R @ Riw)[:, 2]
    sinn = torch.sqrt(z[0] ** 2 + z[1] ** 2)   # |z x e3|
    return torch.stack([h / prm["manifold_p_sigma"], torch.asin(sinn) / prm["manifold_q_sigma"]])


def jac(f, x):
    x = torch.tensor(x)
    r = f(x)
    J = torch.autograd.functional.jacobian(f, x)
    return r.detach().numpy().tolist(), J.detach().numpy().tolist()


def main():
    synth = importlib.import_module("2dliw-slam_amd.synth")
    from oracle import pyoracle
    base = synth.office_params()
    prm = dict(base)
    prm["iw"] = extrinsic(base["T_imu_to_wheel"])
    prm["il"] = extrinsic(base["T_imu_to_laser"])
    orc = pyoracle.Oracle(base)   # only used as the pre-integration provider of the synthetic generator
    d = synth.make_window(orc, base, seed=99, n=5, L=12)
    st = d["states"]
    out = {"params": {k: base[k] for k in base}, "laser": [], "imu": [], "wheel": [], "ground": []}
    for j in range(12):
        k = int(d["laser_frame"][j])
        x = np.concatenate([st[0, 0:6], st[k, 0:6]])
        r, J = jac(lambda xx: laser_res(prm, d["laser_pts"][j], xx), x)
        out["laser"].append({"pts": d["laser_pts"][j].tolist(), "x": x.tolist(), "res": r, "jac": J})
    for k in range(4):
        x = np.concatenate([st[k], st[k + 1]])
        r, J = jac(lambda xx: imu_res(prm, d["imu_X"][k], d["imu_J"][k], d["imu_sqrtP"][k], float(d["imu_Dt"][k]), xx), x)
        out["imu"].append({"X": d["imu_X"][k].tolist(), "J": d["imu_J"][k].tolist(), "sqrtP": d["imu_sqrtP"][k].tolist(),
                           "Dt": float(d["imu_Dt"][k]), "x": x.tolist(), "res": r, "jac": J})
        x = np.concatenate([st[k, 0:6], st[k + 1, 0:6]])
        r, J = jac(lambda xx: wheel_res(prm, d["wheel_T"][k], d["wheel_sqrtP"][k], xx), x)
        out["wheel"].append({"T": d["wheel_T"][k].tolist(), "sqrtP": d["wheel_sqrtP"][k].tolist(), "x": x.tolist(), "res": r, "jac": J})
    for k in range(5):
        x = st[k, 0:6].copy()
        r, J = jac(lambda xx: ground_res(prm, xx), x)
        out["ground"].append({"x": x.tolist(), "res": r, "jac": J})
    # whole-window normal equations (init topology) from the stacked torch residual vector
    n = 4
    dw = synth.make_window(orc, base, seed=123, n=n, L=9)
    x0 = torch.tensor(dw["states"].reshape(-1))

    def stacked(xf):
        xs = xf.reshape(n, 15)
        rs = []
        for j in range(9):
            k = int(dw["laser_frame"][j])
            rs.append(laser_res(prm, dw["laser_pts"][j], torch.cat([xs[0, 0:6], xs[k, 0:6]])))
        for k in range(n - 1):
            rs.append(imu_res(prm, dw["imu_X"][k], dw["imu_J"][k], dw["imu_sqrtP"][k], float(dw["imu_Dt"][k]), torch.cat([xs[k], xs[k + 1]])))
            rs.append(wheel_res(prm, dw["wheel_T"][k], dw["wheel_sqrtP"][k], torch.cat([xs[k, 0:6], xs[k + 1, 0:6]])))
        for _ in range(n):   # the reference adds the whole ground set once per frame (solver.cpp:142-159)
            for k in range(n):
                rs.append(ground_res(prm, xs[k, 0:6]))
        return torch.cat(rs)
    r = stacked(x0)
    Jf = torch.autograd.functional.jacobian(stacked, x0)
    H = (Jf.T @ Jf).numpy()
    g = (Jf.T @ r).numpy()
    win = {k: np.asarray(dw[k]).tolist() for k in ("states", "laser_frame", "laser_pts", "match_pose", "has_match", "imu_X", "imu_J",
                                                   "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt")}
    win["n"] = n
    out["window_init"] = {"window": win, "H": H.tolist(), "g": g.tolist(), "cost": float(0.5 * (r * r).sum())}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "factors_golden.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
