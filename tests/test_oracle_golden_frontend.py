"""Oracle restatements of the widened rows (pose-graph edge factors, laser line fit, TUM pose) vs the committed golden vectors
tests/golden/frontend_golden.json (tests/golden/make_golden_frontend.py: torch reverse-mode autograd over Rodrigues formulas,
LAPACK SVD, numpy quaternion algebra).  PARITY UNPINNED w.r.t. the reference itself — these pin the oracle's math; the product
is compared with the oracle elsewhere (tests/test_gpu_posegraph.py, test_laser_frontend.py, test_outputs.py)."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frontend_golden.json")))


def test_posegraph_normal_equations_golden(pyoracle):
    orc = pyoracle.Oracle(G["params"])
    c = G["posegraph"]
    H, g, cost, idx = pyoracle.posegraph_linearize(orc, G["pg_params"], c["poses"], c["seq_idx"], c["seq_tf12"], c["loop_idx"], c["loop_tf12"])
    assert list(idx) == c["idx"]
    Hg, gg = np.array(c["H"]), np.array(c["g"])
    assert abs(cost - c["cost"]) <= 1e-9 * c["cost"]
    assert np.abs(g - gg).max() <= 1e-8 * np.abs(gg).max()
    assert np.abs(H - Hg).max() <= 1e-8 * np.abs(Hg).max()


def test_laser_line_fit_golden(liw, pyoracle):
    lp = liw.laser.office_laser_params(G["params"])
    orc = pyoracle.LaserOracle(lp)
    for c in G["line_fit"]:
        pts = np.array(c["points"])
        for scan in (orc.spawn_scan(pts), liw.laser.Scan.spawn(lp, pts)):     # the oracle and, for good measure, the product
            L = scan.lines()
            assert L.shape[0] == 1, L.shape                                    # one straight wall -> one line
            abc = L[0, 6:9]
            sg = np.sign(abc @ np.array(c["abc"]))
            assert np.abs(sg * abc - np.array(c["abc"])).max() <= 1e-8
            assert np.abs(L[0, 0:2] - np.array(c["p1"])).max() <= 1e-8 and np.abs(L[0, 3:5] - np.array(c["p2"])).max() <= 1e-8
            assert c["max_dis"] <= lp["line_max_dis"]


def test_tum_pose_golden(liw, pyoracle):
    prm = G["params"]
    for c in G["tum"]:
        line = pyoracle.tum_line(prm["T_imu_to_wheel"], True, c["time"], c["p"], c["q"])
        got = np.array(line.split(), dtype=np.float64)
        ref = np.array(c["xyz_quat"])
        assert got[0] == c["time"]
        if got[7] * ref[6] < 0:
            ref[3:] = -ref[3:]
        # 1e-7: the extrinsic is re-orthonormalised through an UN-normalised quaternion (params.cpp:44-54 on 7-digit YAML values), so
        # the rotation is orthonormal to ~1e-7 only and Eigen's Quaternion(R) is not exactly unit; the golden one is
        assert np.abs(got[1:] - ref).max() <= 1e-7
        assert np.abs(liw.outputs.tum_pose(prm, c["p"], c["q"]) - got[1:]).max() <= 6e-11      # the product prints the same numbers (10 decimals)
