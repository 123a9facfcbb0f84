"""Scales of a marginalisation mismatch: Delta_g error next to |Delta_g|, |g| and cond(H_mm), with a 60-digit Schur complement of the
oracle's H, g as arbiter.  (Finding: compared after separate LM solves the two sides differ by |H| ~ 1e10 times their round-off-level state
difference; soak_batch.py therefore marginalises both sides at identical states.)
usage: python tests/soak/diagnose_marg.py SEED"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
pyoracle.build()
seed = int(sys.argv[1])
rng = np.random.default_rng(77000 + seed)
prm = synth.office_params(); orc = pyoracle.Oracle(prm)
n = int(rng.integers(2, 31)); nd = int(rng.integers(2, 7))
B = int(rng.choice([nd, 97, 1500, 2100, 2500])) if n <= 12 else int(rng.choice([nd, 97, 700]))
cap = int(rng.choice([1, 2, 5, 12]))
base = [synth.make_window(orc, prm, seed=88000 + 10 * seed + k, n=n, L=int(rng.integers(0, 260)), state_noise=float(rng.choice([0.2, 1.0]))) for k in range(nd)]
orc.set_max_iterations(cap)
wos = []
for k in range(nd):
    wo = pyoracle.Window(base[k]); orc.set_prior(None); orc.init_solve(wo); wos.append(wo)
bs = liw.BatchSolver(prm, base)
bs.solve(liw.LIW_MODE_INIT, cap)
# marginalise at the oracle's linearisation point (states and laser_match poses), as soak_batch.py does
bs.set_states(np.stack([np.asarray(w["states"]).reshape(n, 15) for w in wos]))
bs.t["match_pose"].copy_(bs.torch.from_numpy(np.concatenate([np.asarray(w["match_pose"]).reshape(-1) for w in wos])).to(bs.dev))
sH, dH, dg = bs.marginalize()
dH, dg = dH.cpu().numpy().reshape(-1, 15, 15), dg.cpu().numpy()
print("seed", seed, "n", n, "cap", cap)
for k in range(nd):
    wo = wos[k]; orc.set_prior(None); orc.marginalization(wo); m = orc.marg_pieces()
    H, g = m["H"], m["g"]
    N = H.shape[0]
    Hmm, Hrm, gm = H[:N - 15, :N - 15], H[N - 15:, :N - 15], g[:N - 15]
    cond = np.linalg.cond(Hmm) if N > 15 else 1.0
    # the same Schur complement in extended precision as an arbiter between the two fp64 results
    e = np.abs(dg[k] - m["Delta_g"])
    ref = None
    if 15 < N <= 90:   # 60-digit arbiter (slow: small windows only)
        import mpmath as mp
        mp.mp.dps = 60
        Hm = mp.matrix(Hmm.tolist()); x = mp.lu_solve(Hm, mp.matrix(gm.tolist()))
        ref = np.array([float(mp.mpf(g[N - 15 + i]) - sum(mp.mpf(Hrm[i, j]) * x[j] for j in range(N - 15))) for i in range(15)])
    print("k %d L %d |dg| %.3e |g| %.3e cond(Hmm) %.2e  gpu-oracle %.3e" % (k, len(base[k]["laser_frame"]), np.abs(m["Delta_g"]).max(), np.abs(g).max(), cond, e.max()),
          ("| gpu-exact %.3e oracle-exact %.3e" % (np.abs(dg[k] - ref).max(), np.abs(m["Delta_g"] - ref).max())) if ref is not None else "")
