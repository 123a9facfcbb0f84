"""Where does a Delta_g mismatch of tests/soak/soak_batch.py come from?  The GPU's dense H, g of the marginalisation topology (liw_batch_export_dense)
against the oracle's, entry by entry, next to the round-off scale a = |J|^T |R| of the gradient sums, then both Schur complements.
usage: python tests/soak/diagnose_marg_terms.py SEED"""
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
bs.set_states(np.stack([np.asarray(w["states"]).reshape(n, 15) for w in wos]))
bs.t["match_pose"].copy_(bs.torch.from_numpy(np.concatenate([np.asarray(w["match_pose"]).reshape(-1) for w in wos])).to(bs.dev))
bs.linearize(liw.LIW_MODE_MARG)
Hg, gg, cg = [t.cpu().numpy() for t in bs.export_dense(liw.LIW_MODE_MARG)]
sH, dH, dg = bs.marginalize()
dg = dg.cpu().numpy()
names = ["p", "q", "v", "ba", "bw"]
for k in range(nd):
    wo = wos[k]; orc.set_prior(None); orc.marginalization(wo); m = orc.marg_pieces()
    H, g, J, R = m["H"], m["g"], m["J"], m["R"]
    a = np.abs(J).T @ np.abs(R)
    N = H.shape[0]
    eg = np.abs(gg[k] - g)
    print("window %d (n=%d, L=%d): cost gpu %.6e oracle %.6e; |g|max %.3e, a max %.3e; max |g_gpu - g_oracle| %.3e = %.2e of a" % (k, n, len(base[k]["laser_frame"]), cg[k], 0.5 * float(R @ R), np.abs(g).max(), a.max(), eg.max(), (eg / np.maximum(a, 1e-300)).max()))
    for f in range(n):
        for bi, (lo, hi) in enumerate(((0, 3), (3, 6), (6, 9), (9, 12), (12, 15))):
            sl = slice(15 * f + lo, 15 * f + hi)
            print("   frame %d %-2s: |g| %.3e  a %.3e  |dg| %.3e (%.1e of a)   |dH|/|H| %.1e" % (f, names[bi], np.abs(g[sl]).max(), a[sl].max(), eg[sl].max(), (eg[sl] / np.maximum(a[sl], 1e-300)).max(),
                  np.abs(Hg[k][sl, :] - H[sl, :]).max() / max(np.abs(H[sl, :]).max(), 1e-300)))
    W = np.linalg.solve(H[:N - 15, :N - 15], H[N - 15:, :N - 15].T).T
    sc = (a[N - 15:] + np.abs(W) @ a[:N - 15]).max()
    dgo = g[N - 15:] - W @ g[:N - 15]
    dgg = gg[k][N - 15:] - np.linalg.solve(Hg[k][:N - 15, :N - 15], Hg[k][N - 15:, :N - 15].T).T @ gg[k][:N - 15]
    print("   Delta_g: kernel vs oracle %.3e = %.2e of its round-off scale %.3e; numpy Schur of the GPU's own H, g vs kernel %.3e; vs oracle %.3e" % (
        np.abs(dg[k] - m["Delta_g"]).max(), np.abs(dg[k] - m["Delta_g"]).max() / sc, sc, np.abs(dgg - dg[k]).max(), np.abs(dgg - m["Delta_g"]).max()))
