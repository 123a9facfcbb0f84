"""Soak run of the pose-graph relinearisation against the oracle over random graph sizes / loop counts / seeds: normal equations
(H, g, cost) and LM solves at iteration caps 3 / 10 with the full configuration and to convergence without ground_q.
usage: python tests/soak/soak_posegraph.py FIRST LAST   (on the MI355X box)"""
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

prm = synth.office_params()
pg = liw.posegraph.office_pg_params()
pgs, orc = liw.posegraph.PoseGraph(prm), pyoracle.Oracle(prm)
bad = []
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(31000 + seed)
    N = int(rng.choice([2, 3, 5, 11, 12, 30, 63, 64, 65, 90, 130]))
    n_loop = int(rng.integers(0, N // 6 + 1)) if N >= 6 else 0     # the generator needs room for a loop closure
    msg = None
    try:
        G = liw.posegraph.make_pose_graph(prm, N=N, seed=seed, n_loop=n_loop)
        args = (G["poses"], G["seq_idx"], G["seq_tf12"], G["loop_idx"], G["loop_tf12"])
        Hg, gg, cg = pgs.linearize(pg, *args)
        Ho, go, co, idx = pyoracle.posegraph_linearize(orc, pg, *args)
        if abs(cg - co) > 1e-12 * co or np.abs(gg[idx] - go).max() > 1e-10 * np.abs(go).max() or np.abs(Hg[np.ix_(idx, idx)] - Ho).max() > 1e-10 * np.abs(Ho).max():
            msg = "normal equations"
        for cfg, cap in ((pg, 3), (pg, 10), (dict(pg, use_ground_q_factor=False), 0)):
            xg, sg = pgs.solve(cfg, *args, max_iters=cap)
            xo, so = pyoracle.posegraph_solve(orc, cfg, *args, max_iters=cap)
            if (sg["iterations"], sg["termination"], sg["successful"]) != (so["iterations"], so["termination"], so["successful"]):
                msg = "summary cap %d: %s vs %s" % (cap, sg, so)
            elif np.abs(xg - xo).max() > 1e-6 * max(1.0, np.abs(xo).max()):
                msg = "poses cap %d rel %.3e" % (cap, np.abs(xg - xo).max() / max(1.0, np.abs(xo).max()))
    except Exception as e:   # noqa: BLE001
        msg = repr(e)[:300]
    if msg:
        bad.append(seed)
        print("seed", seed, "N", N, "loops", n_loop, "FAILED:", msg)
print("seeds %s..%s: %d failures %s" % (sys.argv[1], int(sys.argv[2]) - 1, len(bad), bad))
