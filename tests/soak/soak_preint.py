"""Soak run of the batched pre-integration kernels against the oracle's sequential accumulators over random ragged interval sets
(0 ... 300 samples per interval, irregular spacing, random biases; wheel samples denser or sparser than the 50 ms gate).
usage: python tests/soak/soak_preint.py FIRST LAST   (on the MI355X box)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
pyoracle.build()


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


prm = synth.office_params()
orc = pyoracle.Oracle(prm)
bp = liw.BatchPreint(prm)
bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(41000 + seed)
    msg = None
    try:
        M = int(rng.integers(1, 40))
        imu, wheel = [], []
        for m in range(M):
            cnt = int(rng.choice([1, 2, 3, 7, 20, 21, 64, 300]))
            span = cnt * float(rng.choice([0.002, 0.005, 0.01]))
            t = 5.0 + m + np.sort(rng.uniform(0.0, span, cnt))
            t += 1e-6 * np.arange(cnt)                                   # distinct stamps
            s = np.zeros((cnt, 7)); s[:, 0] = t
            s[:, 1:4] = rng.normal(0.0, 1.0, (cnt, 3)) + np.array([0.0, 0.0, 9.8])
            s[:, 4:7] = rng.normal(0.0, 0.5, (cnt, 3))
            imu.append((s, float(t[0] + rng.uniform(0.0, 0.002)), float(t[-1] + rng.uniform(0.0, 0.004)), rng.normal(0.0, 1e-2, 6)))
            wc = int(rng.choice([1, 2, 3, 6, 12]))
            tw = 5.0 + m + np.cumsum(rng.choice([0.02, 0.05, 0.051, 0.1], wc))
            w = np.zeros((wc, 13)); w[:, 0] = tw
            yaw = np.cumsum(rng.normal(0.0, 0.02, wc)); pos = np.cumsum(rng.normal(0.03, 0.01, (wc, 2)), axis=0)
            for k in range(wc):
                c, sn = np.cos(yaw[k]), np.sin(yaw[k])
                w[k, 1:10] = np.array([[c, -sn, 0.0], [sn, c, 0.0], [0.0, 0.0, 1.0]]).reshape(9)
                w[k, 10:12] = pos[k]; w[k, 12] = rng.normal(0.0, 1e-4)
            wheel.append((w, float(tw[0] + rng.uniform(-0.01, 0.01)), float(tw[-1] + rng.uniform(0.0, 0.03))))
        X, J, S, Dt = [t_.cpu().numpy() for t_ in bp.imu(imu)]
        for m, iv in enumerate(imu):
            Xo, Jo, So, Dto = orc.imu_preint(*iv)
            if abs(Dt[m] - Dto) > 1e-12 * max(1.0, abs(Dto)) or relerr(X[m], Xo) > 1e-11 or relerr(J[m].reshape(-1), np.asarray(Jo).reshape(-1)) > 1e-11:
                msg = "imu interval %d (%d samples): X %.2e J %.2e" % (m, len(iv[0]), relerr(X[m], Xo), relerr(J[m].reshape(-1), np.asarray(Jo).reshape(-1)))
            elif relerr(S[m].reshape(-1), np.asarray(So).reshape(-1)) > 1e-7:
                msg = "imu sqrt_info interval %d (%d samples): %.2e" % (m, len(iv[0]), relerr(S[m].reshape(-1), np.asarray(So).reshape(-1)))
        T, Sw, Dtw = [t_.cpu().numpy() for t_ in bp.wheel(wheel)]
        for m, iv in enumerate(wheel):
            To, So, Dto = orc.wheel_preint(*iv)
            if abs(Dtw[m] - Dto) > 1e-12 * max(1.0, abs(Dto)) or np.abs(T[m] - np.asarray(To)).max() > 1e-11 or relerr(Sw[m].reshape(-1), np.asarray(So).reshape(-1)) > 1e-9:
                msg = "wheel interval %d (%d samples)" % (m, len(iv[0]))
    except Exception as e:   # noqa: BLE001
        msg = repr(e)[:300]
    if msg:
        bad.append(seed)
        print("seed", seed, "FAILED:", msg)
print("seeds %s..%s: %d failures %s" % (sys.argv[1], int(sys.argv[2]) - 1, len(bad), bad))
