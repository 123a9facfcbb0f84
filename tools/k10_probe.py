"""fixed K = 10 solve + marginalisation of the default bench batch, a few repetitions: python tools/k10_probe.py [B]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
prm = synth.office_params()
tw = bench.make_tiled(liw, synth, prm, B, 30, 2000, seed0=20240, n_base=64)
bs = liw.BatchSolver(prm, tw.base, tile=tw.tile())
x0, mp0 = bs.t["x"].clone(), bs.t["match_pose"].clone()
for K in (10, 10, 10, 50, 10):
    bs.t["x"].copy_(x0); bs.t["match_pose"].copy_(mp0); bs.t["has_prior"].zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bs.solve(liw.LIW_MODE_INIT, K)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    bs.marginalize()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("K=%d: solve %.2f ms, marginalize %.2f ms -> %.0f solves/s" % (K, 1e3 * (t1 - t0), 1e3 * (t2 - t1), B / (t2 - t0)))
# after 12 s without GPU work (what the parity gate's oracle solves amount to in bench.py): does the first solve pay for the idle GPU?
for idle in (12.0, 0.0, 12.0):
    t_end = time.perf_counter() + idle
    while time.perf_counter() < t_end:
        pass
    bs.t["x"].copy_(x0); bs.t["match_pose"].copy_(mp0); bs.t["has_prior"].zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bs.solve(liw.LIW_MODE_INIT, 10); bs.marginalize()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("after %.0f s of host-only work: K=10 solve + marginalize %.2f ms -> %.0f solves/s" % (idle, 1e3 * (t2 - t0), B / (t2 - t0)))
