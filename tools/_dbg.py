import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
prm = synth.office_params(); hp = liw.HostPreint(prm)
for mode_name in ("init", "track"):
    d = synth.make_window(hp, prm, seed=515, n=2, L=60)
    out = {}
    for var in ("0", "4"):
        os.environ["LIW_STEP_VARIANT"] = var
        bs = liw.BatchSolver(prm, [d], history_records=12)
        if mode_name == "track":
            orc = pyoracle.Oracle(prm); w = pyoracle.Window(d); orc.init_solve(w); orc.marginalization(w); X, J, R = orc.get_prior()
            import torch
            bs.t["prior_X"].copy_(torch.from_numpy(X)); bs.t["prior_J"].copy_(torch.from_numpy(J.reshape(-1))); bs.t["prior_R"].copy_(torch.from_numpy(R)); bs.t["has_prior"].fill_(1)
        bs.solve(liw.LIW_MODE_INIT if mode_name == "init" else liw.LIW_MODE_TRACK, 10)
        out[var] = (bs.history().copy(), bs.summaries()[0])
    h0, s0 = out["0"]; h4, s4 = out["4"]
    print(mode_name, s0, s4)
    for it in range(min(s0["iterations"], s4["iterations"]) + 1):
        print(" it", it, float(np.abs(h0[it] - h4[it]).max()))
