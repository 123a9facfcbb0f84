"""Replay one seed of tests/test_gpu_random_shapes.py and print, per LM iteration of both solves, the relative distance between
the product's and the oracle's state vectors (where do they part: early = a defect, late and gradual = round-off on a flat
cost).  usage: python tests/soak/diagnose_seed.py SEED"""
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


def cmp_hist(tag, hg, ho, n):
    for k in range(min(len(hg), len(ho))):
        xo = ho[k]["x"].reshape(n, 15)
        d = np.abs(hg[k] - xo)
        print("%s it %2d rel %.3e  worst entry (frame, var) %s  oracle cost %.9g radius %.3g %s" % (
            tag, k, d.max() / max(np.abs(xo).max(), 1e-12), np.unravel_index(d.argmax(), d.shape), ho[k]["cost"], ho[k]["radius"],
            "ok" if ho[k]["successful"] else "rejected"))
    print(tag, "records", len(hg), len(ho))


seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
prm = synth.office_params()
if seed % 4 == 3:
    prm = dict(prm, fast_mode=True)
orc, slv = pyoracle.Oracle(prm), liw.Solver(prm)
n = int(rng.integers(2, 25)); L = int(rng.integers(0, 301)); cap = int(rng.choice([1, 3, 8, 20]))
d = synth.make_window(orc, prm, seed=3000 + seed, n=n, L=L, state_noise=float(rng.choice([0.2, 1.0])))
if L > 20 and seed % 2:
    keep = d["laser_frame"] != int(rng.integers(1, n))
    d["laser_frame"], d["laser_pts"] = d["laser_frame"][keep], d["laser_pts"][keep]
print("seed", seed, "n", n, "L", L, "cap", cap, "fast", bool(prm.get("fast_mode")))
wo, wg = pyoracle.Window(d), liw.Window(d)
orc.set_prior(None); slv.set_prior(None)
orc.set_max_iterations(cap); orc.init_solve(wo); ho = orc.iterations()
slv.set_window(wg); sg = slv.init_solve(cap); hg = slv.history()
print("init", sg, orc.summary())
cmp_hist("init ", hg, ho, n)
orc.marginalization(wo); slv.marginalization()
sub = dict(d); sub["n"] = 2
for k in ("states", "match_pose"):
    sub[k] = np.asarray(wo[k]).reshape(n, -1)[n - 2:n].copy()
sub["has_match"] = np.asarray(d["has_match"])[n - 2:n].copy()
for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
    sub[k] = np.asarray(d[k])[n - 2:n - 1].copy()
m = np.asarray(d["laser_frame"]) >= n - 2
sub["laser_frame"] = (np.asarray(d["laser_frame"])[m] - (n - 2)).astype(np.int32)
sub["laser_pts"] = np.asarray(d["laser_pts"])[m].copy()
sub["states"][1, 0:3] += rng.normal(0, 0.01, 3)
if not prm.get("fast_mode"):
    slv.set_prior(orc.get_prior())
wo2, wg2 = pyoracle.Window(sub), liw.Window(sub)
orc.set_max_iterations(50); orc.solve(wo2); ho2 = orc.iterations()
slv.set_window(wg2); sg2 = slv.solve(); hg2 = slv.history()
print("track", sg2, orc.summary())
cmp_hist("track", hg2, ho2, 2)
