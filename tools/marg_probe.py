"""Time the marginalisation kernel alone on one 30-frame window (BatchSolver.marginalize(), synchronised around each call).
usage: python tools/marg_probe.py [REPS]   (on the MI355X box; LIW_MARG_WAVES=1/4 selects the kernel variant)"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prm = synth.office_params()
hp = liw.HostPreint(prm)
for n in (4, 10, 30):
    s1 = liw.BatchSolver(prm, [synth.make_window(hp, prm, seed=20240, n=n, L=2000)])
    s1.solve(liw.LIW_MODE_TRACK, 5)
    t = 0.0
    for rep in range(reps + 5):
        s1.t["has_prior"].zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s1.marginalize()
        torch.cuda.synchronize()
        if rep >= 5:
            t += time.perf_counter() - t0
    print("n=%d marginalize (linearise + chain + eigen + tail): %.1f us" % (n, 1e6 * t / reps))
