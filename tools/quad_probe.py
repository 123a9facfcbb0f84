"""What binds k_lm_step_quad?  Times the first LM steps of a full batch (every window active) with the kernel's diagnostic aliasing
(LIW_QUAD_PROBE, read once per process: bit 0 = every row reads window 0's partial records, bit 1 = every row's back-substitution record is
window 0's; results are wrong by design) — run once per probe value:  for p in 0 1 2 3; do LIW_QUAD_PROBE=$p python tools/quad_probe.py; done"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
prm = synth.office_params()
tw = bench.make_tiled(liw, synth, prm, B, 30, 2000, seed0=20240, n_base=64)      # (tiled on the device: no B-fold host concatenation)
bs = liw.BatchSolver(prm, tw.base, tile=tw.tile())
x0 = bs.t["x"].clone()
M = liw.LIW_MODE_INIT
res = []
for rep in range(4):
    bs.t["x"].copy_(x0)
    bs.lm_begin(M, 50)
    bs.lm_linearize(M, 0)
    ts = []
    for it in range(4):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); bs.lm_step(M); e1.record(); bs.lm_linearize(M, 1); e2.record()
        ts.append((e0, e1, e2))
    torch.cuda.synchronize()
    if rep:
        res.append([(a.elapsed_time(b), b.elapsed_time(c)) for a, b, c in ts])
r = np.array(res).mean(axis=0)
print("LIW_QUAD_PROBE=%s B=%d: step ms per launch (launch 1..4) %s | linearise %s" % (os.environ.get("LIW_QUAD_PROBE", "0"), B, np.round(r[:, 0], 3), np.round(r[:, 1], 3)))
