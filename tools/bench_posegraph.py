"""Pose-graph relinearisation timing: LM iterations on synthetic key-frame graphs, dense blocked MFMA
Cholesky underneath.  usage: python tools/bench_posegraph.py [N ...]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")


def main():
    Ns = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [200, 1000, 2000]
    prm = synth.office_params()
    pg = dict(liw.posegraph.office_pg_params(), use_ground_q_factor=False)
    pgs = liw.posegraph.PoseGraph(prm)
    for N in Ns:
        G = liw.posegraph.make_pose_graph(prm, N=N, seed=N, n_loop=max(5, N // 40), laps=2.2)
        args = (G["poses"], G["seq_idx"], G["seq_tf12"], G["loop_idx"], G["loop_tf12"])
        pgs.solve(pg, *args, max_iters=1)           # warm-up (allocations, code objects)
        t0 = time.perf_counter()
        x, s = pgs.solve(pg, *args)
        dt = time.perf_counter() - t0
        n = 6 * N
        np_ = (n + 63) // 64 * 64
        # SPD solve alone
        rng = np.random.default_rng(0)
        A = np.eye(n) * n + 0.01 * rng.normal(size=(n, n)) if n <= 6000 else None
        out = {"key_frames": N, "unknowns": n, "loop_edges": int(len(G["loop_idx"])), "lm_iterations": s["iterations"], "termination": s["termination"],
               "solve_s": round(dt, 4), "ms_per_iteration": round(1e3 * dt / max(1, s["iterations"]), 3),
               "cholesky_gflop_per_factorisation": round(np_ ** 3 / 3 / 1e9, 2),
               "end_error_before_m": round(float(np.linalg.norm(G["poses"][-1, :3] - G["truth"][-1, :3])), 3),
               "end_error_after_m": round(float(np.linalg.norm(x[-1, :3] - G["truth"][-1, :3])), 3)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
