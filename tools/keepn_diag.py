"""Per-solve state error of the teacher-forced keep-N tracking solves (the body of tests/test_gpu_replay.py::_teacher_forced_tracking, without the bar)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
replay = importlib.import_module("2dliw-slam_amd.replay")
from oracle import pyoracle
pyoracle.build()
import test_gpu_replay as TR
from parity_util import rel_inf
keep, duration, seed = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
prm = synth.office_params(); lp = liw.laser.office_laser_params(prm)
msgs, truth = replay.make_log(prm, duration=duration, seed=seed)
orc = TR.oracle_replay(pyoracle, prm, lp, msgs, keep=keep, capture=True)
slv = liw.Solver(prm)
for k, c in enumerate(orc.captures()):
    w = liw.Window(c)
    slv.set_prior((c["prior_X"], c["prior_J"].reshape(15, 15), c["prior_R"]) if c["has_prior"] else None)
    slv.set_window(w)
    s = slv.solve()
    h = slv.history()
    e = rel_inf(w["states"].reshape(-1), c["states_after"])
    print(k, "n", c["n"], "it", s["iterations"], c["iterations"], "term", s["termination"], c["termination"], "err %.2e" % e)
