"""Does overlapping the LM step kernel of one half of the batch with the linearisation of the other half (two BatchSolver objects on
two HIP streams) beat one batch on one stream?  usage: python tools/two_stream_probe.py [B]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
prm = synth.office_params()
wins = bench.make_batch(liw, synth, prm, B, 30, 2000, seed0=20240, n_base=64)


def run(solvers, streams, reps=3):
    x0 = [s.t["x"].clone() for s in solvers]
    mp0 = [s.t["match_pose"].clone() for s in solvers]
    best = 1e9
    for rep in range(reps + 1):
        for s, a, b in zip(solvers, x0, mp0):
            s.t["x"].copy_(a); s.t["match_pose"].copy_(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s, st in zip(solvers, streams):
            with torch.cuda.stream(st):
                s.solve(liw.LIW_MODE_INIT, 50)
                s.marginalize()
        torch.cuda.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    return best


one = liw.BatchSolver(prm, wins)
t1 = run([one], [torch.cuda.Stream()])
print("one stream, %d windows: %.1f ms -> %.0f solves/s" % (B, 1e3 * t1, B / t1))
one.close()
for parts in (2, 3, 4):
    hs = [liw.BatchSolver(prm, wins[k::parts]) for k in range(parts)]
    t2 = run(hs, [torch.cuda.Stream() for _ in range(parts)])
    print("%d streams x %d windows: %.1f ms -> %.0f solves/s" % (parts, len(wins[0::parts]), 1e3 * t2, B / t2))
    for h in hs:
        h.close()
