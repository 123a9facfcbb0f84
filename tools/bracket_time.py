"""The linearise bracket of one LM iteration (laser || IMU || wheel + ground on three streams) with every window active, next to the
role kernels alone: python tools/bracket_time.py [B]   (probe builds: LIW_EXTRA_FLAGS, tools/ab_slab.sh)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench, torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
prm = synth.office_params()
wins = bench.make_batch(liw, synth, prm, B, 30, 2000, seed0=20240, n_base=64)
bs = liw.BatchSolver(prm, wins)
M = liw.LIW_MODE_INIT
kt = bs.time_kernels(M, 3)
bs.lm_begin(M, 50); bs.lm_linearize(M, 0); bs.lm_step(M); bs.lm_linearize(M, 1)
torch.cuda.synchronize()
ts = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); bs.lm_linearize(M, 1); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("flags=%s serial_roles=%s" % (os.environ.get("LIW_EXTRA_FLAGS"), os.environ.get("LIW_SERIAL_ROLES")),
      {k: round(v, 3) for k, v in kt.items() if k.startswith("k_lin") and not k.endswith("marg")}, "bracket ms", [round(t, 3) for t in ts])
