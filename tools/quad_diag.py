"""Per-iteration distance to the oracle of the one-wave (LIW_STEP_VARIANT=1) and quad (=3) step kernels on one window (run on the GPU box)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
pyoracle.build()
n, k, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prm = synth.office_params()
orc = pyoracle.Oracle(prm)
d = synth.make_window(orc, prm, seed=4100 + 17 * n + k, n=n, L=(0 if (k == 3 or n == 1) else 20 * n + 37 * k))
wo = pyoracle.Window(d)
orc.set_prior(None); orc.set_max_iterations(iters); orc.init_solve(wo)
so, ho = orc.summary(), orc.iterations()
print("oracle", so)
for v in ("1", "3"):
    os.environ["LIW_STEP_VARIANT"] = v
    bs = liw.BatchSolver(prm, [d] * 5, history_records=iters + 1)
    bs.solve(liw.LIW_MODE_INIT, iters)
    h, s = bs.history(), bs.summaries()[0]
    errs = [float(np.abs(h[it, 0] - ho[it]["x"].reshape(n, 15)).max() / np.abs(ho[it]["x"]).max()) for it in range(min(len(ho), s["iterations"] + 1))]
    print("variant", v, s["iterations"], s["termination"], s["final_cost"], " ".join("%.1e" % e for e in errs))
