"""The launch configurations only large batches reach — several (window, frame) groups per laser wave with chunks that
straddle group boundaries, empty groups in the middle of a wave, role kernels on forked streams — checked against the
single-window path (one group per wave, single-launch k_lin_all) and against the oracle, window by window."""
import numpy as np
import pytest

from parity_util import assert_normal_eq_close

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()))


@pytest.mark.parametrize("n,mode_name", [(9, "init"), (30, "init"), (13, "marg")])
def test_multi_group_laser_waves_match_single_window_path(liw, synth, pyoracle, n, mode_name):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    rng = np.random.default_rng(n)
    B = 1200 if n == 9 else (600 if n == 13 else 324)         # enough windows for G = 8 groups per wave
    base = []
    for k in range(6):
        L = int(rng.integers(0, 260))
        w = synth.make_window(orc, prm, seed=700 + 10 * n + k, n=n, L=L)
        if k % 2 == 1 and L > 10:                                # knock out every block of two frames: empty groups inside a wave
            keep = ~np.isin(w["laser_frame"], [2, n - 2])
            w["laser_frame"], w["laser_pts"] = w["laser_frame"][keep], w["laser_pts"][keep]
        base.append(w)
    windows = [base[b % 6] for b in range(B)]
    mode = liw.LIW_MODE_INIT if mode_name == "init" else liw.LIW_MODE_MARG
    big = liw.BatchSolver(prm, windows)
    big.linearize(mode)
    Hb, gb, cb = [t.cpu().numpy() for t in big.export_dense(mode)]
    for k in range(6):
        one = liw.BatchSolver(prm, [base[k]])
        one.linearize(mode)
        H1, g1, c1 = [t.cpu().numpy() for t in one.export_dense(mode)]
        for b in (k, k + 6 * 7, B - 6 + k):                      # first, middle and last copies of this window in the batch
            assert rel(Hb[b], H1[0]) <= 1e-12 and rel(gb[b], g1[0]) <= 1e-12 and abs(cb[b] - c1[0]) <= 1e-12 * max(1.0, abs(c1[0])), (k, b)
        if mode_name == "init":                                  # (the marginalisation topology is compared with the oracle in test_gpu_batch.py)
            wo = pyoracle.Window(base[k])
            orc.set_prior(None)
            Ho, go, co = orc.linearize(wo, 0)
            assert_normal_eq_close(H1[0], g1[0], Ho, go, co, what="window %d" % k)


def test_large_batch_solve_matches_oracle_on_sampled_windows(liw, synth, pyoracle):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B = 7, 1600
    base = [synth.make_window(orc, prm, seed=910 + k, n=n, L=30 + 41 * k) for k in range(5)]
    bs = liw.BatchSolver(prm, [base[b % 5] for b in range(B)])
    bs.solve(liw.LIW_MODE_INIT, 12)
    got, summ = bs.states(), bs.summaries()
    orc.set_max_iterations(12)
    for k in range(5):
        wo = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.init_solve(wo)
        so = orc.summary()
        for b in (k, B - 5 + k):
            assert summ[b]["iterations"] == so["iterations"] and summ[b]["termination"] == so["termination"], (k, b)
            assert rel(got[b], wo["states"].reshape(n, 15)) <= 1e-6, (k, b)
    orc.set_max_iterations(50)


def test_throughput_step_kernel_variant_matches_latency_variant_and_oracle(liw, synth, pyoracle):
    """Batches of 1 024 windows and more run k_lm_step_quad (four windows per wave; k_lm_step<true> — no look-ahead in
    the elimination sweep — right behind it for the windows it leaves out), smaller ones k_lm_step<false> / k_lm_step_tw.  Same arithmetic, different load schedule: states must agree window by window — init topology
    (arrow) and, after the on-device marginalisation, tracking topology with the stored prior — and follow the oracle."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B, K = 6, 2100, 10
    base = [synth.make_window(orc, prm, seed=1310 + k, n=n, L=25 + 30 * k) for k in range(4)]
    big = liw.BatchSolver(prm, [base[b % 4] for b in range(B)])
    small = liw.BatchSolver(prm, base)
    picks = [(k, b) for k in range(4) for b in (k, 4 * 300 + k, B - 4 + k)]
    big.solve(liw.LIW_MODE_INIT, K)
    small.solve(liw.LIW_MODE_INIT, K)
    gb, gs, sb, ss = big.states(), small.states(), big.summaries(), small.summaries()
    orc.set_max_iterations(K)
    for k, b in picks:
        assert sb[b]["iterations"] == ss[k]["iterations"] and sb[b]["termination"] == ss[k]["termination"], (k, b)
        assert rel(gb[b], gs[k]) <= 1e-10, (k, b)
    for k in range(4):
        wo = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.init_solve(wo)
        assert ss[k]["iterations"] == orc.summary()["iterations"], k
        assert rel(gs[k], wo["states"].reshape(n, 15)) <= 1e-6, k
    orc.set_max_iterations(50)
    big.marginalize()
    small.marginalize()
    big.solve(liw.LIW_MODE_TRACK, K)
    small.solve(liw.LIW_MODE_TRACK, K)
    gb, gs, sb, ss = big.states(), small.states(), big.summaries(), small.summaries()
    for k, b in picks:
        assert sb[b]["iterations"] == ss[k]["iterations"] and sb[b]["termination"] == ss[k]["termination"], (k, b)
        # the two schedules leave the init solve 1e-14 apart; the marginalisation (cond(H_mm) ~ 1e5 .. 1e7 on these windows) and the
        # tracking solve on its prior amplify that to ~1e-8 (measured 5e-10 .. 9e-9), the same factor by which either one moves
        # when its own summation order changes
        assert rel(gb[b], gs[k]) <= 1e-7, (k, b)


def test_packed_imu_records_are_the_callers_arrays(liw, synth, pyoracle, monkeypatch):
    """Large batches read the IMU block inputs from the packed records liw_batch_lm_begin builds (observation, Dt, the bias blocks of
    the pre-integration Jacobian, the upper triangle of sqrt_inverse_P): same numbers, so the solve is bit-identical to the one that
    reads the caller's arrays (LIW_NO_IMU_PACK=1).  A sqrt_inverse_P that is NOT upper triangular (not what
    imu_preintegraption.h:149 produces, but the arrays are the caller's) sends the role back to the full arrays on the device."""
    import copy
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B, K = 7, 800, 8                                        # 4 800 IMU blocks: above the packing threshold
    base = [synth.make_window(orc, prm, seed=2210 + k, n=n, L=20 + 25 * k) for k in range(4)]
    for w in base:
        S = np.asarray(w["imu_sqrtP"]).reshape(-1, 15, 15)
        assert np.all(np.tril(S, -1) == 0.0)                   # the synthetic windows follow the reference's construction

    def run(windows, no_pack):
        if no_pack:
            monkeypatch.setenv("LIW_NO_IMU_PACK", "1")
        else:
            monkeypatch.delenv("LIW_NO_IMU_PACK", raising=False)
        bs = liw.BatchSolver(prm, windows)
        bs.solve(liw.LIW_MODE_INIT, K)
        return bs.states().copy(), bs.summaries()

    wins = [base[b % 4] for b in range(B)]
    x_pk, s_pk = run(wins, False)
    x_full, s_full = run(wins, True)
    assert np.array_equal(x_pk, x_full)
    assert [s["iterations"] for s in s_pk] == [s["iterations"] for s in s_full]
    # oracle on the sampled windows
    orc.set_max_iterations(K)
    for k in range(4):
        wo = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.init_solve(wo)
        assert rel(x_pk[k], wo["states"].reshape(n, 15)) <= 1e-6, k
    orc.set_max_iterations(50)
    # one block of one window with a dense sqrt_inverse_P: the whole batch falls back, results = the full-array path
    odd = copy.deepcopy(base[1])
    S = np.array(odd["imu_sqrtP"], dtype=np.float64).reshape(-1, 15, 15)
    S[2] = S[2] + 1e-3 * np.tril(np.ones((15, 15)), -1) * np.abs(S[2]).max()
    odd["imu_sqrtP"] = S.reshape(np.asarray(odd["imu_sqrtP"]).shape)
    wins2 = list(wins)
    wins2[B - 3] = odd
    y_pk, _ = run(wins2, False)
    y_full, _ = run(wins2, True)
    assert np.array_equal(y_pk, y_full)
    assert not np.array_equal(y_pk[B - 3], x_pk[B - 3])         # (the dense block does change that window)


def test_laser_end_points_with_z_take_the_full_path(liw, synth, pyoracle):
    """A batch solve scans the laser end points once: 2-D scans (z = 0 in the laser frame, src/utilies/common.cpp:22-24) let the laser
    role skip the four z planes and their terms.  One end point with z != 0 anywhere in the batch switches the whole batch back to the
    full evaluation: the window that carries it follows the oracle (which always uses z), and so do its z = 0 neighbours."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B, K = 7, 800, 8
    base = [synth.make_window(orc, prm, seed=2310 + k, n=n, L=40 + 25 * k) for k in range(4)]
    for w in base:
        assert np.all(np.asarray(w["laser_pts"]).reshape(-1, 4, 3)[:, :, 2] == 0.0)
    tilted = dict(base[2])
    pts = np.array(tilted["laser_pts"], dtype=np.float64).reshape(-1, 4, 3)
    pts[5:40, :, 2] = 0.05 * np.random.default_rng(5).normal(size=(35, 4))     # end points off the scan plane
    tilted["laser_pts"] = pts.reshape(np.asarray(tilted["laser_pts"]).shape)
    orc.set_max_iterations(K)

    def oracle_states(w):
        wo = pyoracle.Window(w)
        orc.set_prior(None)
        orc.init_solve(wo)
        return wo["states"].reshape(n, 15).copy()

    flat = [base[b % 4] for b in range(B)]
    bs = liw.BatchSolver(prm, flat)
    bs.solve(liw.LIW_MODE_INIT, K)
    x_flat = bs.states().copy()
    for k in range(4):
        assert rel(x_flat[k], oracle_states(base[k])) <= 1e-6, k
    mixed = list(flat)
    mixed[B - 2] = tilted
    bs2 = liw.BatchSolver(prm, mixed)
    bs2.solve(liw.LIW_MODE_INIT, K)
    x_mixed = bs2.states().copy()
    ref_t = oracle_states(tilted)
    assert rel(x_mixed[B - 2], ref_t) <= 1e-6
    assert rel(x_mixed[B - 2], x_flat[B - 2]) > 1e-9                     # (the z components do matter to that window)
    for k in range(4):                                                   # z = 0 windows: skipping the planes is exact up to the order of the sums
        assert rel(x_mixed[k], x_flat[k]) <= 1e-9, k
    orc.set_max_iterations(50)


def _lockstep_solve(liw, prm, windows, exchange, mode, K):
    """two rank objects of this process through the real sharded loop (LockstepComm stands in for RCCL only)"""
    import threading
    import torch
    comms = liw.batch.LockstepComm.make(2)
    ranks = [liw.BatchSolver(prm, windows, rank=r, world=2, exchange=exchange, comm=comms[r]) for r in range(2)]
    errs = []

    def drive(rk):
        try:
            rk.solve(mode, K)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
            comms[0].sh["bar"].abort()
    th = [threading.Thread(target=drive, args=(rk,)) for rk in ranks]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    torch.cuda.synchronize()
    return ranks


@pytest.mark.parametrize("B,variant", [(1088, None), (3, "3")])
def test_factor_sharded_solve_in_the_large_batch_format(liw, synth, pyoracle, monkeypatch, B, variant):
    """in the large-batch format (>= 1 024 windows, or LIW_STEP_VARIANT=3) the prologue of k_lm_step_quad takes a
    candidate's laser cost from the compact cost array CS, which only the LOCAL laser role wrote — the exchange refreshed the group
    records and not CS, so every rank accepted / rejected on its own shard's cost.  Two ranks over a batch that takes that path:
    ranks bit-identical, same iteration counts / terminations as the un-sharded solve and the oracle (solver.cpp:93-106, :161-168)."""
    if variant:
        monkeypatch.setenv("LIW_STEP_VARIANT", variant)
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, K, nb = 6, 12, 4
    base = [synth.make_window(orc, prm, seed=5100 + k, n=n, L=40 + 23 * k) for k in range(nb)]
    windows = [base[b % nb] for b in range(B)]
    ref = liw.BatchSolver(prm, windows)
    ref.solve(liw.LIW_MODE_INIT, K)
    rs, rsum = ref.states(), ref.summaries()
    orc.set_max_iterations(K)
    want = []
    for k in range(nb):
        wo = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.init_solve(wo)
        want.append((orc.summary(), wo["states"].reshape(n, 15).copy()))
    orc.set_max_iterations(50)
    for xch in ("allreduce", "oneshot"):
        ranks = _lockstep_solve(liw, prm, windows, xch, liw.LIW_MODE_INIT, K)
        a, b = ranks[0].states(), ranks[1].states()
        assert np.array_equal(a, b), xch                                          # same sums -> same decisions -> same bits
        sa, sb = ranks[0].summaries(), ranks[1].summaries()
        assert [(s["iterations"], s["termination"]) for s in sa] == [(s["iterations"], s["termination"]) for s in sb]
        assert [(s["iterations"], s["termination"]) for s in sa] == [(s["iterations"], s["termination"]) for s in rsum], xch
        assert rel(a, rs) <= 1e-9, xch
        for k in range(min(nb, B)):
            for bb in {k, B - nb + k if B > nb else k}:
                assert sa[bb]["iterations"] == want[k][0]["iterations"] and sa[bb]["termination"] == want[k][0]["termination"], (xch, bb)
                assert rel(a[bb], want[k][1]) <= 1e-6, (xch, bb)


def test_quad_eigen_square_root_matches_one_wave_kernel_and_oracle(liw, synth, pyoracle, monkeypatch):
    """Batches above 256 windows marginalise with two kernels (k_marg_schur_chain: the chain by one wave per window;
    k_marg_schur_eigq: the Jacobi eigen square root of solver.cpp:390-402 for FOUR windows per wave, rows in registers).  Same rotations
    in the same order as the one-wave kernel it replaces (LIW_MARG_EIG=1): the new prior (X, J^T J, J^T R, sqrt_H^T sqrt_H — eigenvector
    order among equal eigenvalues and signs are free) must agree with that kernel to round-off and with the oracle as before.  B = 301:
    the last wave of the eigen kernel holds one window and three empty lane groups.  Windows with and without a carried prior."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B, K = 5, 301, 6
    base = [synth.make_window(orc, prm, seed=4410 + k, n=n, L=30 + 20 * k) for k in range(6)]
    nb = len(base)

    def run(eig_env):
        if eig_env:
            monkeypatch.setenv("LIW_MARG_EIG", "1")
        else:
            monkeypatch.delenv("LIW_MARG_EIG", raising=False)
        bs = liw.BatchSolver(prm, [base[b % nb] for b in range(B)])
        bs.solve(liw.LIW_MODE_INIT, K)
        out = []
        for _ in range(2):                               # second pass: with the prior the first one wrote
            sH, dH, dg = bs.marginalize()
            out.append(dict(sH=sH.cpu().numpy().reshape(B, 6, 6), dH=dH.cpu().numpy().reshape(B, 15, 15), dg=dg.cpu().numpy().reshape(B, 15),
                            J=bs.t["prior_J"].cpu().numpy().reshape(B, 15, 15).copy(), R=bs.t["prior_R"].cpu().numpy().reshape(B, 15).copy(),
                            X=bs.t["prior_X"].cpu().numpy().reshape(B, 15).copy(), has=bs.t["has_prior"].cpu().numpy().copy()))
        return bs.states(), bs.t["match_pose"].cpu().numpy().reshape(B, n, 12), out

    xq, mpq, new = run(False)
    xw, _, old = run(True)
    assert np.array_equal(xq, xw)
    # the same batch through the same kernels once more, in newly allocated buffers: bit-identical (nothing uninitialised is read)
    xq2, _, new2 = run(False)
    assert np.array_equal(xq, xq2)
    for p in range(2):
        for key in ("sH", "dH", "dg", "J", "R", "X"):
            assert np.array_equal(new[p][key], new2[p][key]), (p, key)
    worst = 0.0
    for p in range(2):
        a, o = new[p], old[p]
        assert np.array_equal(a["X"], o["X"]), p
        if p == 0:   # the chain is the same code on the same records
            assert np.array_equal(a["dH"], o["dH"]) and np.array_equal(a["dg"], o["dg"]), p
        else:        # second pass: the chain starts from the prior of the first, which the two eigen kernels leave a round-off apart
            sc = np.abs(o["dH"]).max(axis=(1, 2), keepdims=True)
            assert (np.abs(a["dH"] - o["dH"]) / sc).max() <= 1e-12, p
            assert (np.abs(a["dg"] - o["dg"]) / np.maximum(1.0, np.abs(o["dg"]).max(axis=1, keepdims=True))).max() <= 1e-9, p
        assert np.all(a["has"] == 1) and np.all(o["has"] == 1)
        for b in list(range(0, 12)) + [150, 151, 299, 300]:
            JJa, JJo = a["J"][b].T @ a["J"][b], o["J"][b].T @ o["J"][b]
            sc = np.abs(JJo).max()
            e1 = np.abs(JJa - JJo).max() / sc
            JRa, JRo = a["J"][b].T @ a["R"][b], o["J"][b].T @ o["R"][b]
            e2 = np.abs(JRa - JRo).max() / max(1.0, np.abs(JRo).max())
            # sqrt_H = the pose block of the prior Jacobian's rows in either kernel
            assert np.array_equal(a["sH"][b], a["J"][b][:6, :6])
            worst = max(worst, e1, e2)
            assert e1 <= 1e-12 and e2 <= 1e-9, (p, b, e1, e2)
            # J^T J is Delta_H up to the eigenvalue floor (1e-8 on a matrix of norm ~1e11)
            assert np.abs(JJa - 0.5 * (a["dH"][b] + a["dH"][b].T)).max() / sc <= 1e-12, (p, b)
    # oracle at the same linearisation point, first pass
    for k in range(nb):
        w = pyoracle.Window(base[k])
        w["states"][:] = xq[k].reshape(w["states"].shape)
        w["match_pose"][:] = mpq[k].reshape(w["match_pose"].shape)
        orc.set_prior(None)
        orc.marginalization(w)
        Xo, Jo, Ro = orc.get_prior()
        for b in (k, nb * 40 + k, B - 1 - ((B - 1 - k) % nb)):
            if b % nb != k:
                continue
            J, R = new[0]["J"][b], new[0]["R"][b]
            assert np.abs(J.T @ J - Jo.T @ Jo).max() / np.abs(Jo.T @ Jo).max() <= 1e-9, (k, b)
            assert np.abs(J.T @ R - Jo.T @ Ro).max() / max(1.0, np.abs(Jo.T @ Ro).max()) <= 1e-7, (k, b)
    print("quad eigen kernel vs one-wave kernel: worst relative difference of J^T J / J^T R over the sampled windows %.2e" % worst)
    # a window whose chain fails (a pivot that is not positive and finite: here NaN states) leaves its prior alone, in either form of the
    # kernel — the eigen kernel must skip it although its three neighbours in the wave (and, for the last window, nobody) go on
    for eig_env in (False, True):
        if eig_env:
            monkeypatch.setenv("LIW_MARG_EIG", "1")
        else:
            monkeypatch.delenv("LIW_MARG_EIG", raising=False)
        bs = liw.BatchSolver(prm, [base[b % nb] for b in range(B)])
        bs.solve(liw.LIW_MODE_INIT, K)
        x = bs.states().copy()
        for b in (7, 130, B - 1):
            x[b, 2, 3:6] = np.nan
        bs.set_states(x)
        bs.marginalize()
        has = bs.t["has_prior"].cpu().numpy()
        J = bs.t["prior_J"].cpu().numpy().reshape(B, 15, 15)
        bad = np.zeros(B, dtype=bool)
        bad[[7, 130, B - 1]] = True
        assert np.all(has[bad] == 0) and np.all(J[bad] == 0.0), eig_env
        assert np.all(has[~bad] == 1) and np.all(np.isfinite(J[~bad])), eig_env
        for b in (4, 5, 6, 128, 129, 131, B - 2):      # the neighbours are what they are without the poisoned windows
            assert np.array_equal(J[b], (old if eig_env else new)[0]["J"][b]), (eig_env, b)


@pytest.mark.parametrize("n,B,mode_name", [(2, 1101, "track"), (3, 1030, "init"), (5, 1101, "init"), (9, 1027, "init")])
def test_multi_window_imu_chain_is_bit_identical_to_one_window_per_wave(liw, synth, pyoracle, monkeypatch, n, B, mode_name):
    """k_lin_imu_chain_multi (round 6): windows of at most 8 IMU blocks share a wave (16 / (n - 1) windows: 16 two-frame tracking windows,
    8 three-frame, 4 five-frame, 2 nine-frame ones) instead of one window per wave.  Same arithmetic per block, the jj -> ii hand-over
    restarts at every window: the per-frame IMU records, the solve and its LM history must be BIT-identical to the one-window-per-wave
    kernel (LIW_NO_IMU_MULTI=1) — including a batch size that leaves the last wave partly empty and windows that finish early (the
    compacted list re-groups the windows of a wave from iteration to iteration)."""
    import torch
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    nb_ = 5
    base = [synth.make_window(orc, prm, seed=9300 + 10 * n + k, n=n, L=12 * (n - 1) + 9 * k, state_noise=(0.3 if k % 2 else 1.0)) for k in range(nb_)]
    wins = [base[b % nb_] for b in range(B)]
    mode = liw.LIW_MODE_TRACK if mode_name == "track" else liw.LIW_MODE_INIT

    def run(no_multi):
        if no_multi:
            monkeypatch.setenv("LIW_NO_IMU_MULTI", "1")
        else:
            monkeypatch.delenv("LIW_NO_IMU_MULTI", raising=False)
        bs = liw.BatchSolver(prm, wins, history_records=22)
        if mode_name == "track":
            bs.marginalize()
            bs.t["prior_X"].view(B, 15).copy_(bs.t["x"].view(B, n, 15)[:, n - 2])
        bs.solve(mode, 20)
        torch.cuda.synchronize()
        out = (bs.states().copy(), bs.summaries(), bs.history().copy())
        sH, dH, dg = bs.marginalize()
        out += (dH.cpu().numpy().copy(), dg.cpu().numpy().copy())
        bs.close()
        return out
    xa, sa, ha, dHa, dga = run(False)
    xb, sb, hb, dHb, dgb = run(True)
    assert [(s["iterations"], s["termination"]) for s in sa] == [(s["iterations"], s["termination"]) for s in sb]
    if n <= 3:
        assert len({s["iterations"] for s in sa}) > 1                  # windows finish at different iterations (larger windows all run into the cap of 20)
    assert np.array_equal(xa, xb) and np.array_equal(ha, hb)
    assert np.array_equal(dHa, dHb) and np.array_equal(dga, dgb)
    # and the oracle on the distinct windows (INIT) / the first window pair (TRACK is covered by tests/test_gpu_track_batch.py)
    if mode_name == "init":
        orc.set_max_iterations(20)
        for k in range(nb_):
            w = pyoracle.Window(base[k])
            orc.set_prior(None)
            orc.init_solve(w)
            so = orc.summary()
            for b in (k, B - nb_ + k if (B - nb_ + k) % nb_ == k else k):
                assert (sa[b]["iterations"], sa[b]["termination"]) == (so["iterations"], so["termination"]), (k, b)
                assert rel(xa[b], w["states"].reshape(n, 15)) <= 1e-6, (k, b)
        orc.set_max_iterations(50)


def test_factor_sharded_tracking_solve_and_marginalisation(liw, synth, pyoracle):
    """The factor-sharded loop in the TRACK topology (the steady state on several GPUs: laser blocks of the new frame split over the ranks, 21
    pair totals per group on the wire): 1 100 two-frame windows with a carried prior through two lock-step rank objects, both exchange
    variants — ranks bit-identical, same iteration counts / terminations as the un-sharded solve and the oracle, states to round-off; then
    the sharded marginalisation (MARG exchange) against the un-sharded one (solver.cpp:631-820, :257-442)."""
    import sys, os, importlib, threading
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    B, nb_ = 1100, 5
    base = [bench.sub_window(synth.make_window(orc, prm, seed=9500 + k, n=3, frame_counts=[0, 30 + 11 * k, 45 + 6 * k]), 1) for k in range(nb_)]
    wins = [base[b % nb_] for b in range(B)]
    ref = liw.BatchSolver(prm, wins)
    ref.marginalize()                                                     # a prior to carry: sit it on the older frame
    ref.t["prior_X"].view(B, 15).copy_(ref.t["x"].view(B, 2, 15)[:, 0])
    prior = {k: ref.t[k].clone() for k in ("prior_X", "prior_J", "prior_R", "has_prior")}
    ref.solve(liw.LIW_MODE_TRACK, 0)
    xr, sr = ref.states().copy(), ref.summaries()
    _, dHr, dgr = ref.marginalize()
    dHr, dgr = dHr.cpu().numpy(), dgr.cpu().numpy()
    for xch in ("allreduce", "oneshot"):
        comms = liw.batch.LockstepComm.make(2)
        ranks = [liw.BatchSolver(prm, wins, rank=r, world=2, exchange=xch, comm=comms[r]) for r in range(2)]
        assert sum(rk.Ltot for rk in ranks) == ref.Ltot
        assert ranks[0].exchange_bytes(liw.LIW_MODE_TRACK) == 8 * (B * 2 * 21 + 1)
        for rk in ranks:
            for k, v in prior.items():
                rk.t[k].copy_(v)
        errs, outs = [], [None, None]

        def drive(i):
            try:
                ranks[i].solve(liw.LIW_MODE_TRACK, 0)
                outs[i] = ranks[i].marginalize()
            except Exception as e:   # noqa: BLE001
                errs.append(e)
                comms[0].sh["bar"].abort()
        th = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not errs, errs
        torch.cuda.synchronize()
        a, b_ = ranks[0].states(), ranks[1].states()
        assert np.array_equal(a, b_), xch
        sa = ranks[0].summaries()
        assert [(s["iterations"], s["termination"]) for s in sa] == [(s["iterations"], s["termination"]) for s in sr], xch
        assert rel(a, xr) <= 1e-9, xch
        dH0, dH1 = outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()
        assert np.array_equal(dH0, dH1) and np.array_equal(outs[0][2].cpu().numpy(), outs[1][2].cpu().numpy()), xch
        assert np.abs(dH0 - dHr).max() <= 1e-9 * np.abs(dHr).max() and np.abs(outs[0][2].cpu().numpy() - dgr).max() <= 1e-7 * max(1.0, np.abs(dgr).max()), xch
        for rk in ranks:
            rk.close()
    for k in range(nb_):
        w = pyoracle.Window(base[k])
        orc.set_prior((prior["prior_X"].view(B, 15)[k].cpu().numpy(), prior["prior_J"].view(B, 15, 15)[k].cpu().numpy(), prior["prior_R"].view(B, 15)[k].cpu().numpy()))
        orc.solve(w)
        so = orc.summary()
        assert (sr[k]["iterations"], sr[k]["termination"]) == (so["iterations"], so["termination"]), k
        assert rel(xr[k], w["states"].reshape(2, 15)) <= 1e-6, k
    orc.set_prior(None)
    ref.close()
