"""The lane-per-group laser role of large 2-D batches (csrc/k_laser_slab.hip) against the lane-per-block kernel it replaces there
(k_lin_laser, LIW_NO_LASER_SLAB=1) and against the oracle: same group records to round-off (the sums run in block order instead of a
tree), same LM histories, the packed rows cover ragged groups / windows without blocks on a frame / a batch that is no multiple of 64 /
finished windows, 3-D end points and small batches stay on the old kernel, and a block exactly on its line still fails its own window only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(liw, synth, pyoracle, B=4480 + 37, n=30):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    base = [synth.make_window(orc, prm, seed=7100 + k, n=n, L=int(L)) for k, L in enumerate((29 * 3, 29 * 5 + 11, 40, 29 * 4 + 3, 200, 64))]
    # ragged groups: window 2 has frames without any block; window 4's blocks all sit on a few frames
    base[4]["laser_frame"] = np.sort(np.asarray(base[4]["laser_frame"]) % 7 + 1).astype(np.int32)
    rng = np.random.default_rng(5)
    wins = []
    for b in range(B):
        w = dict(base[b % len(base)])
        if b >= len(base):
            st = np.array(w["states"], copy=True)
            st[:, 0:3] += rng.normal(0.0, 2e-3, (n, 3))
            w["states"] = st
            mp = np.array(w["match_pose"], copy=True)
            mp[:, 0:6] = st[0, 0:6]; mp[:, 6:12] = st[:, 0:6]
            w["match_pose"] = mp
        wins.append(w)
    return prm, orc, base, wins


def _records(liw, bs):
    M = liw.LIW_MODE_INIT
    bs.lm_begin(M, 50)
    bs.lm_linearize(M, 0)
    import torch
    torch.cuda.synchronize()
    return bs.PL[0].cpu().numpy().reshape(bs.B, bs.n, 128).copy()


def test_group_records_and_lm_histories_match_the_block_kernel_and_the_oracle(liw, synth, pyoracle, monkeypatch):
    prm, orc, base, wins = _batch(liw, synth, pyoracle)
    B, n = len(wins), 30
    monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
    old = liw.BatchSolver(prm, wins, history_records=0)
    r_old = _records(liw, old)
    monkeypatch.delenv("LIW_NO_LASER_SLAB")
    new = liw.BatchSolver(prm, wins)
    r_new = _records(liw, new)
    # entry-scaled comparison of the 6x6 blocks / gradients of every (window, frame) record
    for name, sl in (("Haa", slice(0, 36)), ("Hbb", slice(36, 72)), ("Hab", slice(72, 108))):
        a, o = r_new[:, :, sl].reshape(B, n, 6, 6), r_old[:, :, sl].reshape(B, n, 6, 6)
        scale = np.abs(o).max(axis=(2, 3), keepdims=True) + 1e-300
        assert (np.abs(a - o) / scale).max() <= 1e-12, name
    for sl in (slice(108, 114), slice(114, 120), slice(120, 121)):
        a, o = r_new[:, :, sl], r_old[:, :, sl]
        assert (np.abs(a - o) / (np.abs(o).max(axis=2, keepdims=True) + 1e-300)).max() <= 1e-11
    assert np.all(r_new[:, :, 121:] == 0.0) and np.all(r_new[:, 0, :] == 0.0)      # padding slots; frame 0 owns no blocks in the init topology
    assert np.abs(r_new).max() > 0.0 and not np.array_equal(r_new, r_old)         # (a different summation order, not the same kernel)
    # whole solves
    K = 12
    old.set_states(np.stack([w["states"] for w in wins])); new.set_states(np.stack([w["states"] for w in wins]))
    monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
    old.solve(liw.LIW_MODE_INIT, K)
    monkeypatch.delenv("LIW_NO_LASER_SLAB")
    new.solve(liw.LIW_MODE_INIT, K)
    xo, xn, so, sn = old.states(), new.states(), old.summaries(), new.summaries()
    assert [s["iterations"] for s in so] == [s["iterations"] for s in sn]
    assert [s["termination"] for s in so] == [s["termination"] for s in sn]
    assert (np.abs(xn - xo).max(axis=(1, 2)) / np.abs(xo).max(axis=(1, 2))).max() <= 1e-9
    orc.set_max_iterations(K)
    for k in range(len(base)):
        w = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.init_solve(w)
        assert orc.summary()["iterations"] == sn[k]["iterations"], k
        assert np.abs(xn[k] - w["states"].reshape(n, 15)).max() <= 1e-6 * np.abs(w["states"]).max(), k
    orc.set_max_iterations(50)
    old.close(); new.close()


def test_three_dimensional_end_points_and_small_batches_keep_the_block_kernel(liw, synth, pyoracle, monkeypatch):
    prm, orc, base, wins = _batch(liw, synth, pyoracle)
    wz = [dict(w) for w in wins]
    pts = np.array(wz[11]["laser_pts"], copy=True)
    pts[3, 8] = 1e-3                          # one end point with a z component somewhere in the batch
    wz[11]["laser_pts"] = pts
    monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
    a = liw.BatchSolver(prm, wz)
    ra = _records(liw, a)
    monkeypatch.delenv("LIW_NO_LASER_SLAB")
    b = liw.BatchSolver(prm, wz)
    rb = _records(liw, b)
    assert np.array_equal(ra, rb), "a batch with 3-D end points must run the lane-per-block kernel either way"
    a.close(); b.close()
    small = wins[:2048 + 5]                    # 33 slabs x 30 frames < 2 048 waves: not worth a lane per group
    monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
    a = liw.BatchSolver(prm, small)
    ra = _records(liw, a)
    monkeypatch.delenv("LIW_NO_LASER_SLAB")
    b = liw.BatchSolver(prm, small)
    rb = _records(liw, b)
    assert np.array_equal(ra, rb)
    a.close(); b.close()


def test_a_block_on_its_line_fails_only_its_window_on_the_slab_path(liw, synth, pyoracle):
    prm, orc, base, wins = _batch(liw, synth, pyoracle)
    n = 30
    bad = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in wins[72].items()}
    bad["states"][9, 0:6] = bad["states"][0, 0:6]
    bad["match_pose"][9, 6:12] = bad["states"][9, 0:6]
    j = int(np.flatnonzero(bad["laser_frame"] == 9)[0])
    bad["laser_pts"][j, 6:9] = bad["laser_pts"][j, 3:6]
    ws = list(wins)
    ws[72] = bad
    bs = liw.BatchSolver(prm, ws)
    x0 = np.array(bad["states"], copy=True)
    bs.solve(liw.LIW_MODE_INIT, 8)
    info, X = bs.summaries(), bs.states()
    assert (info[72]["iterations"], info[72]["termination"]) == (0, 6), info[72]
    assert np.array_equal(X[72], x0)
    assert all(s["termination"] != 6 for k, s in enumerate(info) if k != 72)
    bs.close()


def test_a_zero_length_reference_segment_takes_the_general_form_on_the_slab_path(liw, synth, pyoracle, monkeypatch):
    """The lane-per-group kernel evaluates regular blocks with folded algebra and leaves blocks whose mapped reference segment has
    zero length (normalized() of a zero vector in the reference, src/utilies/common.h:86-95) to the general form in a pass of its own:
    the records of such a window must be those of the lane-per-block kernel (same NaN pattern, sums to round-off)."""
    prm, orc, base, wins = _batch(liw, synth, pyoracle)
    n = 30
    ws = list(wins)
    picks = []
    for k in (5, 70, len(wins) - 1):
        bad = {key: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for key, v in ws[k].items()}
        fr = int(np.bincount(bad["laser_frame"], minlength=n).argmax())     # (the windows are ragged: take a frame that owns blocks)
        picks.append((k, fr))
        js = np.flatnonzero(bad["laser_frame"] == fr)
        for j in (int(js[0]), int(js[-1])):
            bad["laser_pts"][j, 3:6] = bad["laser_pts"][j, 0:3]
        ws[k] = bad
    monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
    a = liw.BatchSolver(prm, ws, history_records=0)
    ra = _records(liw, a)
    monkeypatch.delenv("LIW_NO_LASER_SLAB")
    b = liw.BatchSolver(prm, ws)
    rb = _records(liw, b)
    assert not np.array_equal(ra, rb, equal_nan=True)             # (the lane-per-group kernel did run)
    assert np.array_equal(np.isnan(ra), np.isnan(rb))
    for k, fr in picks:
        o, v = ra[k, fr], rb[k, fr]
        fin = ~np.isnan(o)
        assert np.abs(v[fin] - o[fin]).max() <= 1e-12 * max(np.abs(o[fin]).max(), 1e-300), (k, fr)
    fin = ~np.isnan(ra)
    assert np.abs(rb[fin] - ra[fin]).max() <= 1e-11 * np.abs(ra[fin]).max()
    a.close(); b.close()


def test_one_pose_slab_kernel_marg_records_and_tracking_solves(liw, synth, pyoracle, monkeypatch):
    """k_lin_laser_slab1 (round 6): the one-free-pose topologies over the packed rows of the solve.
      * MARG: the group records bs.marginalize() linearises into (H_bb | g_b | cost of every frame incl. the zero-length-segment and
        on-the-line special cases) against the lane-per-block kernel k_lin_laser<false>, Delta_H / Delta_g / the new prior against it and
        against the oracle (solver.cpp:453-476, :257-442);
      * TRACK: 17 000 two-frame windows (S = 266 slabs >= the 256-slab arming threshold), the marginalisation behind it, against the
        lane-per-block path and the oracle (solver.cpp:669-698)."""
    import ctypes as C
    import torch
    prm, orc, base, wins = _batch(liw, synth, pyoracle)
    B, n = len(wins), 30
    ws = list(wins)
    for k in (5, 70):                                     # a zero-length reference segment (general form on either path)
        bad = {key: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for key, v in ws[k].items()}
        fr = int(np.bincount(bad["laser_frame"], minlength=n).argmax())
        j = int(np.flatnonzero(bad["laser_frame"] == fr)[0])
        bad["laser_pts"][j, 3:6] = bad["laser_pts"][j, 0:3]
        ws[k] = bad

    def run(no_slab):
        if no_slab:
            monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
        else:
            monkeypatch.delenv("LIW_NO_LASER_SLAB", raising=False)
        bs = liw.BatchSolver(prm, ws)
        bs.solve(liw.LIW_MODE_INIT, 5)
        flags = C.c_int(0)
        bs.L.liw_batch_launch_paths(bs.h, C.byref(bs.b), bs._wsp(), C.byref(flags))
        x = bs.states().copy()
        sH, dH, dg = bs.marginalize()
        torch.cuda.synchronize()
        rec = bs.PL[0].cpu().numpy().reshape(B, n, 128).copy()       # the marginalisation linearises into buffer 0
        out = dict(flags=flags.value, x=x, rec=rec, dH=dH.cpu().numpy().reshape(B, 15, 15), dg=dg.cpu().numpy().reshape(B, 15),
                   pJ=bs.t["prior_J"].cpu().numpy().reshape(B, 15, 15).copy(), mp=bs.t["match_pose"].cpu().numpy().reshape(B, n, 12).copy())
        bs.close()
        return out
    new, old = run(False), run(True)
    assert new["flags"] == 3 and old["flags"] == 1
    ra, rb = old["rec"], new["rec"]
    assert np.array_equal(np.isnan(ra), np.isnan(rb))
    assert np.all(rb[:, :, 0:36] == 0.0) and np.all(rb[:, :, 72:114] == 0.0) and np.all(rb[:, :, 121:] == 0.0)     # one pose free: H_aa, H_ab, g_a are structural zeros
    fin = ~np.isnan(ra)
    # the two runs linearise at states 1e-13 apart (different init-solve summation orders): compare group by group at 1e-9 of the group's scale
    sc = np.nanmax(np.abs(ra), axis=2, keepdims=True) + 1e-300
    assert np.nanmax(np.abs(rb - ra) / sc) <= 1e-9
    assert np.abs(rb[fin]).max() > 0 and not np.array_equal(ra, rb, equal_nan=True)
    good = ~np.isnan(old["dH"]).any(axis=(1, 2))
    assert good.sum() >= B - 4
    assert (np.abs(new["dH"][good] - old["dH"][good]).max(axis=(1, 2)) / np.abs(old["dH"][good]).max(axis=(1, 2))).max() <= 1e-9
    # oracle at the slab run's own linearisation point
    import importlib, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    for b in (0, 1, 2, 3, 4, B - 1):
        ref = bench.marg_reference(pyoracle, orc, ws[b], new["x"][b], new["mp"][b], 1)[0]
        assert np.abs(new["dH"][b] - ref["dH"]).max() <= 1e-11 * np.abs(ref["dH"]).max(), b
        assert np.abs(new["dg"][b] - ref["dg"]).max() <= 1e-10 * ref["g_scale"], b
        assert np.abs(new["pJ"][b].T @ new["pJ"][b] - ref["J"].T @ ref["J"]).max() <= 1e-10 * np.abs(ref["dH"]).max(), b

    # ---- TRACK: two-frame windows, prior from a first marginalisation
    Bt = 17000
    tb = [bench.sub_window(synth.make_window(orc, prm, seed=8100 + k, n=3, frame_counts=[0, 40 + 13 * k, 55 + 7 * k]), 1) for k in range(4)]
    tw = [tb[b % 4] for b in range(Bt)]

    def track(no_slab):
        if no_slab:
            monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
        else:
            monkeypatch.delenv("LIW_NO_LASER_SLAB", raising=False)
        bs = liw.BatchSolver(prm, tw)
        bs.marginalize()                                   # leaves a prior on the newest frame ... of THIS window pair: good enough as a carried prior
        bs.t["prior_X"].view(Bt, 15).copy_(bs.t["x"].view(Bt, 2, 15)[:, 0])     # sit it on the older frame, where a tracking solve expects it
        bs.solve(liw.LIW_MODE_TRACK, 0)
        flags = C.c_int(0)
        bs.L.liw_batch_launch_paths(bs.h, C.byref(bs.b), bs._wsp(), C.byref(flags))
        prior = [bs.t[k].cpu().numpy().copy() for k in ("prior_X", "prior_J", "prior_R")]
        x, sm = bs.states().copy(), bs.summaries()
        sH, dH, dg = bs.marginalize()
        out = dict(flags=flags.value, x=x, sm=sm, dH=dH.cpu().numpy().reshape(Bt, 15, 15), prior=prior)
        bs.close()
        return out
    tn, to = track(False), track(True)
    assert tn["flags"] == 3 and to["flags"] == 1, (tn["flags"], to["flags"])
    assert [(s["iterations"], s["termination"]) for s in tn["sm"]] == [(s["iterations"], s["termination"]) for s in to["sm"]]
    assert (np.abs(tn["x"] - to["x"]).max(axis=(1, 2)) / np.abs(to["x"]).max(axis=(1, 2))).max() <= 1e-9
    assert (np.abs(tn["dH"] - to["dH"]).max(axis=(1, 2)) / np.abs(to["dH"]).max(axis=(1, 2))).max() <= 1e-8
    for b in (0, 1, 2, 3, Bt - 1):
        w = pyoracle.Window(tw[b])
        orc.set_prior((tn["prior"][0].reshape(Bt, 15)[b], tn["prior"][1].reshape(Bt, 15, 15)[b], tn["prior"][2].reshape(Bt, 15)[b]))
        orc.solve(w)
        so = orc.summary()
        assert (tn["sm"][b]["iterations"], tn["sm"][b]["termination"]) == (so["iterations"], so["termination"]), (b, tn["sm"][b], so)
        assert np.abs(tn["x"][b] - w["states"].reshape(2, 15)).max() <= 1e-6 * np.abs(w["states"]).max(), b
    orc.set_prior(None)


def test_groups_longer_than_the_sort_bins_and_empty_windows(liw, synth, pyoracle, monkeypatch):
    """k_laser_slab_order sorts the windows of every frame by group length with 4 096 bins (longer groups share the last bin): a batch in which
    one distinct window carries 4 500 blocks on ONE frame, another none at all, next to ordinary ones — packed rows stay near the data volume,
    the solve matches the lane-per-block kernel and the oracle."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B, K = 30, 4480 + 11, 6
    fcs = [synth.ragged_frame_counts(np.random.default_rng(70 + k), n, 300) for k in range(4)]
    giant = np.zeros(n, dtype=np.int64); giant[17] = 4500; giant[3] = 40
    none = np.zeros(n, dtype=np.int64)
    base = [synth.make_window(orc, prm, seed=7300 + k, n=n, frame_counts=fc) for k, fc in enumerate(fcs + [giant, none])]
    wins = [base[b % len(base)] for b in range(B)]

    def run(no_slab):
        if no_slab:
            monkeypatch.setenv("LIW_NO_LASER_SLAB", "1")
        else:
            monkeypatch.delenv("LIW_NO_LASER_SLAB", raising=False)
        bs = liw.BatchSolver(prm, wins)
        bs.solve(liw.LIW_MODE_INIT, K)
        out = (bs.states().copy(), bs.summaries(), bs.launch_paths())
        bs.close()
        return out
    xn, sn, lp = run(False)
    xo, so, lpo = run(True)
    assert lp["flags"] == 3 and lpo["flags"] == 1
    assert 1.0 <= lp["padding_ratio"] < 1.1, lp          # equal lengths share slabs: the 4 500-block groups fill twelve slabs of their own
    assert [(s["iterations"], s["termination"]) for s in sn] == [(s["iterations"], s["termination"]) for s in so]
    assert (np.abs(xn - xo).max(axis=(1, 2)) / np.abs(xo).max(axis=(1, 2))).max() <= 1e-9
    orc.set_max_iterations(K)
    for k in range(len(base)):
        w = pyoracle.Window(base[k])
        orc.set_prior(None)
        orc.init_solve(w)
        for b in (k, B - 1 - ((B - 1 - k) % len(base))):
            assert b % len(base) == k
            assert orc.summary()["iterations"] == sn[b]["iterations"], (k, b)
            assert np.abs(xn[b] - w["states"].reshape(n, 15)).max() <= 1e-6 * np.abs(w["states"]).max(), (k, b)
    orc.set_max_iterations(50)
