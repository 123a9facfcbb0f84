import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import torch
import bench
prm = synth.office_params()
B = int(sys.argv[1]); it = int(sys.argv[2]); nb = int(sys.argv[3])
wins = bench.make_batch(liw, synth, prm, B, 30, 2000, 20240, n_base=nb)
bs = liw.BatchSolver(prm, wins)
x0 = bs.t["x"].clone()
for rep in range(2):
    bs.t["x"].copy_(x0)
    bs.solve(liw.LIW_MODE_INIT, it)
    torch.cuda.synchronize()
    print("solve ok", rep, np.bincount([s["iterations"] for s in bs.summaries()])[-5:], flush=True)
