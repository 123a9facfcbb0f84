"""One seed of tests/soak/soak_slab.py on both laser / IMU paths (default; LIW_NO_LASER_SLAB=1 LIW_NO_IMU_MULTI=1) with the oracle's own round-off
sensitivity beside it: python tests/soak/diagnose_slab.py SEED"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench
from oracle import pyoracle
rel = lambda a, b: float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()))
seed = int(sys.argv[1])
rng = np.random.default_rng(99000 + seed)
prm = synth.office_params(); orc = pyoracle.Oracle(prm)
track = seed % 3 == 2
for env in ({}, {"LIW_NO_LASER_SLAB": "1", "LIW_NO_IMU_MULTI": "1"}):
    rng = np.random.default_rng(99000 + seed)
    for k_, v_ in env.items(): os.environ[k_] = v_
    if track:
        nd = int(rng.integers(3, 9)); B = int(rng.integers(16384, 20000))
        tb = [bench.sub_window(synth.make_window(orc, prm, seed=98000 + 10 * seed + k, n=3, state_noise=float(rng.choice([0.2, 1.0])),
                                                 frame_counts=[0, int(rng.integers(0, 150)), int(rng.integers(0, 150))]), 1) for k in range(nd)]
        bs = liw.BatchSolver(prm, [tb[b % nd] for b in range(B)], history_records=52)
        bs.marginalize()
        bs.t["prior_X"].view(B, 15).copy_(bs.t["x"].view(B, 2, 15)[:, 0])
        prior = [bs.t[k_].cpu().numpy().copy() for k_ in ("prior_X", "prior_J", "prior_R")]
        bs.solve(liw.LIW_MODE_TRACK, 0)
        got, summ, hist = bs.states(), bs.summaries(), bs.history()
        print("env", env, bs.launch_paths())
        for k in range(nd):
            b = ((B - 1 - k) // nd) * nd + k
            pr = (prior[0].reshape(B, 15)[b], prior[1].reshape(B, 15, 15)[b], prior[2].reshape(B, 15)[b])
            wo = pyoracle.Window(tb[k]); orc.set_prior(pr); orc.set_max_iterations(50); orc.solve(wo); so = orc.summary(); its = orc.iterations()
            e = [rel(hist[i, b], its[i]["x"].reshape(2, 15)) for i in range(min(so["iterations"], summ[b]["iterations"]) + 1)]
            # the oracle against itself with 1e-13 noise on the IMU means
            sens = 0.0
            rp = np.random.default_rng(7)
            for _ in range(3):
                alt = dict(tb[k]); alt["imu_X"] = np.asarray(alt["imu_X"]) * (1 + 1e-13 * rp.standard_normal(np.asarray(alt["imu_X"]).shape))
                wa = pyoracle.Window(alt); orc.set_prior(pr); orc.solve(wa); sens = max(sens, rel(wa["states"], wo["states"]))
            print("  k=%d L=%s gpu %s oracle (%d, %d) final %.2e sens %.2e | per-it %s" % (k, np.bincount(tb[k]["laser_frame"], minlength=2), (summ[b]["iterations"], summ[b]["termination"]), so["iterations"], so["termination"], rel(got[b], wo["states"].reshape(2, 15)), sens, " ".join("%.0e" % v for v in e[::3])))
    else:
        n = int(rng.integers(8, 31)); nd = int(rng.integers(3, 9)); S = (2048 + n - 1) // n + int(rng.integers(0, 6)); B = 64 * S - int(rng.integers(0, 64))
        cap = int(rng.choice([2, 5, 12])); Lm = int(rng.integers(40, 900))
        base = [synth.make_window(orc, prm, seed=98000 + 10 * seed + k, n=n, state_noise=float(rng.choice([0.2, 1.0])),
                                  frame_counts=synth.ragged_frame_counts(rng, n, int(rng.integers(max(1, Lm // 4), 2 * Lm)), p_empty=float(rng.choice([0.0, 0.15, 0.4])), spread=float(rng.choice([0.5, 1.0, 2.5]))))
                for k in range(nd)]
        bs = liw.BatchSolver(prm, [base[b % nd] for b in range(B)])
        bs.solve(liw.LIW_MODE_INIT, cap)
        got = bs.states(); mpg = bs.t["match_pose"].cpu().numpy().reshape(B, n, 12)
        sH, dH, dg = bs.marginalize()
        dH = dH.cpu().numpy().reshape(-1, 15, 15); pJ = bs.t["prior_J"].cpu().numpy().reshape(B, 15, 15)
        print("env", env, bs.launch_paths(), "n", n, "cap", cap)
        for k in range(nd):
            b = ((B - 1 - k) // nd) * nd + k
            ref = bench.marg_reference(pyoracle, orc, base[k], got[b], mpg[b], 1)[0]
            H = orc.marg_pieces()["H"]; N = H.shape[0]
            print("  k=%d L=%d Delta_H %.2e prior %.2e cond(Hmm) %.2e |dH|max %.2e min eig %.2e" % (k, len(base[k]["laser_frame"]), np.abs(dH[b] - ref["dH"]).max() / np.abs(ref["dH"]).max(),
                  np.abs(pJ[b].T @ pJ[b] - ref["J"].T @ ref["J"]).max() / np.abs(ref["dH"]).max(), np.linalg.cond(H[:N - 15, :N - 15]), np.abs(ref["dH"]).max(), np.linalg.eigvalsh(ref["dH"]).min()))
    bs.close()
    for k_ in env: os.environ.pop(k_, None)
