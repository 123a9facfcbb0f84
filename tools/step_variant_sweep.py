"""solves/s of the C2 init solve (cap 12 LM iterations) per step-kernel variant and batch size: where does k_lm_step_quad start to pay?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench
prm = synth.office_params()
base = bench.make_batch(liw, synth, prm, 64, 30, 2000, seed0=20240, n_base=16)
for B in (256, 512, 1024, 2048, 4096):
    wins = [base[b % 64] for b in range(B)]
    row = []
    for v in ("0", "1", "2", "3"):
        if v == "2" and B > 512:
            row.append("   -  "); continue
        os.environ["LIW_STEP_VARIANT"] = v
        bs = liw.BatchSolver(prm, wins)
        x0 = bs.t["x"].clone()
        best = 1e9
        for rep in range(3):
            bs.t["x"].copy_(x0); torch.cuda.synchronize(); t0 = time.perf_counter()
            bs.solve(liw.LIW_MODE_INIT, 12); torch.cuda.synchronize()
            if rep: best = min(best, time.perf_counter() - t0)
        row.append("%6.0f" % (B / best)); bs.close()
    print("B=%5d  solves/s (12 iterations)  latency %s  throughput %s  four-wave %s  quad %s" % (B, *row))
