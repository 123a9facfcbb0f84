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
