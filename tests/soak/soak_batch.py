"""Soak run of the batched path (both step-kernel instantiations: B below and above 2 048 windows) over random shapes against
the oracle: init solve with a random iteration cap, marginalisation (Delta_H, Delta_g), tracking solve on the stored prior.
usage: python tests/soak/soak_batch.py FIRST LAST   (on the MI355X box; test infrastructure: imports the oracle)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
pyoracle.build()


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()))


first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    rng = np.random.default_rng(77000 + seed)
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n = int(rng.integers(2, 31))
    nd = int(rng.integers(2, 7))
    B = int(rng.choice([nd, 97, 1500, 2100, 2500])) if n <= 12 else int(rng.choice([nd, 97, 700]))
    cap = int(rng.choice([1, 2, 5, 12]))
    base = [synth.make_window(orc, prm, seed=88000 + 10 * seed + k, n=n, L=int(rng.integers(0, 260)), state_noise=float(rng.choice([0.2, 1.0])))
            for k in range(nd)]
    msg = None
    try:
        bs = liw.BatchSolver(prm, [base[b % nd] for b in range(B)])
        bs.solve(liw.LIW_MODE_INIT, cap)
        got, summ = bs.states(), bs.summaries()
        # the oracle's init solves first: the marginalisation is compared at IDENTICAL linearisation points (states and the
        # laser_match poses): g = J^T r moves by |H| ~ 1e11 times a round-off-level difference, which would swamp the comparison
        orc.set_max_iterations(cap)
        wos, sos = [], []
        for k in range(nd):
            wo = pyoracle.Window(base[k])
            orc.set_prior(None)
            orc.init_solve(wo)
            wos.append(wo); sos.append(orc.summary())
        xo = np.stack([np.asarray(wos[b % nd]["states"]).reshape(n, 15) for b in range(B)])
        for k in range(nd):
            for b in sorted({k, ((B - 1 - k) // nd) * nd + k}):
                if rel(got[b], xo[b]) > 1e-6:
                    msg = "init states k=%d b=%d rel %.3e" % (k, b, rel(got[b], xo[b]))
        bs.set_states(xo)
        mo = np.concatenate([np.asarray(wos[b % nd]["match_pose"]).reshape(-1) for b in range(B)])
        bs.t["match_pose"].copy_(bs.torch.from_numpy(mo).to(bs.dev))   # the constant laser_match poses the init solve wrote back
        sH, dH, dg = bs.marginalize()
        dH, dg = dH.cpu().numpy().reshape(-1, 15, 15), dg.cpu().numpy()
        bs.solve(liw.LIW_MODE_TRACK, cap)
        got2, summ2 = bs.states(), bs.summaries()
        orc.set_max_iterations(cap)
        for k in range(nd):
            wo, so = wos[k], sos[k]
            orc.set_prior(None)
            orc.marginalization(wo)
            m = orc.marg_pieces()
            orc.solve(wo)
            so2 = orc.summary()
            for b in sorted({k, (B // nd // 2) * nd + k if B >= 2 * nd else k, ((B - 1 - k) // nd) * nd + k}):   # copies of window k: b % nd == k
                if (summ[b]["iterations"], summ[b]["termination"]) != (so["iterations"], so["termination"]):
                    msg = "init summary %s vs %s (k=%d b=%d)" % (summ[b], so, k, b)
                if np.abs(dH[b] - m["Delta_H"]).max() > 1e-7 * np.abs(m["Delta_H"]).max():
                    msg = "Delta_H k=%d b=%d %.3e" % (k, b, np.abs(dH[b] - m["Delta_H"]).max() / np.abs(m["Delta_H"]).max())
                # Delta_g = g_r - H_rm H_mm^-1 g_m: the two sides form H and g in different summation orders, and the Schur
                # complement amplifies that round-off by cond(H_mm) (1e6 ... 1e8 here) relative to the summands |g|
                Hm = m["H"][:-15, :-15]
                amp = (np.linalg.cond(Hm) if Hm.size else 1.0) * 2.2e-16 * np.abs(m["g"]).max()
                # ... and g = -J^T R itself is a cancelling sum: its round-off scales with a = |J|^T |R|, not with |g| (round 6, seed 18795: a two-frame window
                # stopped by the cap whose rotation gradient keeps 1e2 of 1e8 — 6.5e-5 off, 6e-13 of that scale; tests/soak/diagnose_marg_terms.py)
                a_ = np.abs(m["J"]).T @ np.abs(m["R"])
                Nn = m["H"].shape[0]
                Ws = np.linalg.solve(Hm, m["H"][Nn - 15:, :Nn - 15].T).T if Hm.size else np.zeros((15, 0))
                gsc = float((a_[Nn - 15:] + np.abs(Ws) @ a_[:Nn - 15]).max())
                if np.abs(dg[b] - m["Delta_g"]).max() > 1e-7 * max(1.0, np.abs(m["Delta_g"]).max()) + 1e-9 * np.abs(m["g"]).max() + 30.0 * amp + 1e-11 * gsc:
                    msg = "Delta_g k=%d b=%d" % (k, b)
                if (summ2[b]["iterations"], summ2[b]["termination"]) != (so2["iterations"], so2["termination"]):
                    msg = "track summary %s vs %s (k=%d b=%d)" % (summ2[b], so2, k, b)
                if rel(got2[b], np.asarray(wo["states"]).reshape(n, 15)) > 1e-6:
                    msg = "track states k=%d b=%d rel %.3e" % (k, b, rel(got2[b], np.asarray(wo["states"]).reshape(n, 15)))
        bs.close()
    except Exception as e:   # noqa: BLE001
        msg = repr(e)[:300]
    if msg:
        bad.append(seed)
        print("seed", seed, "n", n, "B", B, "cap", cap, "FAILED:", msg)
print("seeds %d..%d: %d failures %s" % (first, last - 1, len(bad), bad))
