"""Soak run of the laser front-end (host C++) against the oracle over random rooms and poses: spawn_scan (lines, corners, grid
cells) and do_match (index lists, records).  CPU only.  usage: python tests/soak/soak_laser_frontend.py FIRST LAST"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
pyoracle.build()
import test_laser_frontend as t

prm = synth.office_params()
lp = liw.laser.office_laser_params(prm)
orc = pyoracle.LaserOracle(lp)
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    msg = None
    try:
        room = liw.laser.room_segments(seed)
        p1, q1 = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), 0.0]), np.array([0.0, 0.0, rng.uniform(-3, 3)])
        p2 = p1 + np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 0.0])
        q2 = q1 + np.array([0.0, 0.0, rng.uniform(-0.08, 0.08)])
        pts1, _ = t.scan_points(liw, synth, prm, room, p1, q1, 3 * seed)
        pts2, _ = t.scan_points(liw, synth, prm, room, p2, q2, 3 * seed + 1)
        s1, s2 = liw.laser.Scan.spawn(lp, pts1), liw.laser.Scan.spawn(lp, pts2)
        o1, o2 = orc.spawn_scan(pts1), orc.spawn_scan(pts2)
        for s, o, pts in ((s1, o1, pts1), (s2, o2, pts2)):
            la, lb = s.lines(), o.lines()
            if la.shape != lb.shape or (la.size and np.abs(la[:, :6] - lb[:, :6]).max() > 1e-9):
                msg = "lines %s vs %s" % (la.shape, lb.shape)
                break
            ca, cb = s.concers(), o.concers()
            if ca.shape != cb.shape or (ca.size and np.abs(ca - cb).max() > 1e-9):
                msg = "corners %s vs %s" % (ca.shape, cb.shape)
                break
            for x, y, _z in pts[::11]:
                ka, ia = s.cell_lines(x, y)
                kb, ib = o.cell_lines(x, y)
                if ka != kb or not np.array_equal(ia, ib):
                    msg = "cell (%g, %g)" % (x, y)
        if msg is None:
            for kk in (0, 1):
                pg, qg = p2 + np.array([rng.uniform(-0.02, 0.02), rng.uniform(-0.02, 0.02), 0.0]), q2 + np.array([0.0, 0.0, rng.uniform(-0.01, 0.01)])
                m, mo = liw.laser.do_match(lp, s1, s2, p1, q1, pg, qg, kk), orc.do_match(o1, o2, p1, q1, pg, qg, kk)
                if len(m) != len(mo) or not np.array_equal(m.idx1, mo.idx1) or not np.array_equal(m.idx2, mo.idx2):
                    msg = "match kk=%d sizes %d vs %d" % (kk, len(m), len(mo))
                elif len(m) and np.abs(m.pts - mo.pts).max() > 1e-9:
                    msg = "match records kk=%d" % kk
    except Exception as e:   # noqa: BLE001
        msg = repr(e)[:300]
    if msg:
        bad.append(seed)
        print("seed", seed, "FAILED:", msg)
print("seeds %d..%d: %d failures %s" % (first, last - 1, len(bad), bad))
