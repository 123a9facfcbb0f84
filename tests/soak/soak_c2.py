"""Soak run at the bench's window size (C2: n = 30, L = 2 000; also n = 50, L = 5 000 = C5's window): per-iteration LM states of the
single-window C ABI against the oracle for many synthetic seeds, then marginalisation.  usage: soak_c2.py FIRST LAST [CAP]   (MI355X box)"""
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

cap = int(sys.argv[3]) if len(sys.argv) > 3 else 15
prm = synth.office_params()
orc, slv = pyoracle.Oracle(prm), liw.Solver(prm)
bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    n, L = (50, 5000) if seed % 5 == 4 else (30, 2000)
    d = synth.make_window(orc, prm, seed=50000 + seed, n=n, L=L)
    wo, wg = pyoracle.Window(d), liw.Window(d)
    orc.set_prior(None); slv.set_prior(None)
    orc.set_max_iterations(cap)
    orc.init_solve(wo)
    so, ho = orc.summary(), orc.iterations()
    slv.set_window(wg)
    sg = slv.init_solve(cap)
    hg = slv.history()
    msg = None
    if (sg["iterations"], sg["termination"]) != (so["iterations"], so["termination"]) or len(hg) != len(ho):
        msg = "summary %s vs %s" % (sg, so)
    else:
        worst = max(np.abs(hg[k] - ho[k]["x"].reshape(n, 15)).max() / max(np.abs(ho[k]["x"]).max(), 1e-12) for k in range(len(ho)))
        if worst > 1e-6:
            msg = "per-iteration states rel %.3e" % worst
    # marginalisation at the oracle's linearisation point
    wg2 = liw.Window(dict(d, states=np.asarray(wo["states"]).reshape(n, 15), match_pose=np.asarray(wo["match_pose"]).reshape(n, 12)))
    slv.set_window(wg2)
    sH = slv.marginalization()["sqrt_H"]
    sHo = orc.marginalization(wo)
    if np.abs(sH.T @ sH - sHo.T @ sHo).max() > 1e-6 * max(1.0, np.abs(sHo.T @ sHo).max()):
        msg = "sqrt_H product %.3e" % (np.abs(sH.T @ sH - sHo.T @ sHo).max() / max(1.0, np.abs(sHo.T @ sHo).max()))
    if msg:
        bad.append(seed)
        print("seed", seed, "n", n, "L", L, "FAILED:", msg)
print("seeds %s..%s (cap %d): %d failures %s" % (sys.argv[1], int(sys.argv[2]) - 1, cap, len(bad), bad))
