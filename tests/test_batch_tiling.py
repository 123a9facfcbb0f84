"""Device-side tiling of a batch (BatchSolver(tile=...), bench.py make_tiled): the tensors built from the distinct windows by
repeats are element for element the concatenation the host used to build (25 GB of host arrays per rank at
bench.py's 49 152 windows), and make_tiled() reproduces make_batch()'s windows (same seeds, same jitter stream).  CPU only:
torch on the CPU stands in for the device."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_tiled_tensors_equal_the_host_concatenation(liw, synth):
    prm = synth.office_params()
    hp = liw.HostPreint(prm)
    for n, nb, B in ((5, 3, 11), (4, 4, 8), (2, 2, 5), (3, 5, 3), (3, 1, 4)):
        base = [synth.make_window(hp, prm, seed=40 + k, n=n, L=17 + 5 * k) for k in range(nb)]
        if nb > 1:   # an empty window in the middle of the pattern
            base[1]["laser_frame"], base[1]["laser_pts"] = base[1]["laser_frame"][:0], base[1]["laser_pts"][:0]
        rng = np.random.default_rng(1)
        wins = []
        for b in range(B):
            w = dict(base[b % nb])
            w["states"] = np.asarray(w["states"]) + rng.normal(0, 1e-3, np.asarray(w["states"]).shape)
            w["match_pose"] = np.asarray(w["match_pose"]) + 1e-3 * b
            wins.append(w)
        full = liw.batch.host_arrays(wins)
        Lt = full.pop("_Ltot")
        tile = dict(B=B, states=np.stack([w["states"] for w in wins]), match_pose=np.stack([w["match_pose"] for w in wins]))
        t, B2, Lt2 = liw.batch.tiled_tensors(base[:B] if B < nb else base, tile, "cpu")
        assert (B2, Lt2) == (B, Lt)
        for k, a in full.items():
            if a.size:
                assert np.array_equal(t[k].numpy(), a), (n, nb, B, k)


def test_make_tiled_is_make_batch(liw, synth):
    bench = importlib.import_module("bench")
    prm = synth.office_params()
    B, n, L, nb = 23, 4, 30, 5
    wins = bench.make_batch(liw, synth, prm, B, n, L, seed0=77, n_base=nb)
    tw = bench.make_tiled(liw, synth, prm, B, n, L, seed0=77, n_base=nb)
    assert len(tw) == B
    for b in (0, nb - 1, nb, B // 2, B - 1, -1):
        w, v = wins[b], tw[b]
        for k in ("states", "match_pose", "laser_pts", "laser_frame", "imu_X", "imu_sqrtP", "wheel_T"):
            assert np.array_equal(np.asarray(w[k]).reshape(-1), np.asarray(v[k]).reshape(-1)), (b, k)
    assert [np.asarray(w["states"]).tobytes() for w in tw[3:6]] == [np.asarray(w["states"]).tobytes() for w in wins[3:6]]
