"""Per-iteration error growth of the ragged slab batch (tests/test_gpu_bench_shape.py::test_ragged_slab_batch...): product vs oracle through the
lane-per-group kernel and through the lane-per-block kernel (LIW_NO_LASER_SLAB=1), next to the oracle against itself with 1e-13 IMU noise."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
import bench
prm = synth.office_params(); orc = pyoracle.Oracle(prm)
B, n, L, K, nd = 4421, 30, 2000, 50, 8
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 30240
tw = bench.make_tiled(liw, synth, prm, B, n, L, seed0=seed0, n_base=nd, ragged=True)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
hist = {}
if "--cpu" not in sys.argv:
    for name, env in (("slab", None), ("block", "1")):
        if env: os.environ["LIW_NO_LASER_SLAB"] = env
        else: os.environ.pop("LIW_NO_LASER_SLAB", None)
        bs = liw.BatchSolver(prm, tw.base, tile=tw.tile(), history_records=K + 1)
        bs.solve(liw.LIW_MODE_INIT, K)
        hist[name] = (bs.history().copy(), bs.summaries())
        bs.close()
orc.set_max_iterations(K)
for b in range(nd):
    w = pyoracle.Window(tw[b]); orc.set_prior(None); orc.init_solve(w)
    so, its = orc.summary(), orc.iterations()
    sens = np.zeros(len(its)); rp = np.random.default_rng(7)
    for _ in range(3):
        alt = dict(tw[b]); alt["imu_X"] = np.asarray(alt["imu_X"]) * (1.0 + 1e-13 * rp.standard_normal(np.asarray(alt["imu_X"]).shape))
        wa = pyoracle.Window(alt); orc.set_prior(None); orc.init_solve(wa); ia = orc.iterations()
        for it in range(min(len(its), len(ia))): sens[it] = max(sens[it], rel(ia[it]["x"], its[it]["x"]))
    line = "w%d L=%d it=%d term=%d | sens: %s" % (b, len(tw[b]["laser_frame"]), so["iterations"], so["termination"], " ".join("%.0e" % s for s in sens[::5]))
    for name in hist:
        h, sm = hist[name]
        e = [rel(h[it, b], its[it]["x"].reshape(n, 15)) for it in range(so["iterations"] + 1)]
        line += "\n   %s (it=%d term=%d): %s" % (name, sm[b]["iterations"], sm[b]["termination"], " ".join("%.0e" % s for s in e[::5]))
    print(line)
