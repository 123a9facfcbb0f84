"""End to end over the widened boundary: synthetic scans -> laser front-end (host C++, include/liw_laser.h) ->
liw_window -> init-topology solve on the GPU, against the oracle's front-end + solver on the same scans
(1e-6 relative on the state vector, the north-star tolerance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_frontend_feeds_gpu_solver(liw, synth, pyoracle):
    prm = synth.office_params()
    lp = liw.laser.office_laser_params(prm)
    orc = pyoracle.Oracle(prm)
    lorc = pyoracle.LaserOracle(lp)
    n = 6
    w = synth.make_window(orc, prm, seed=77, n=n, L=0, state_noise=0.5)
    truth, est = w["truth_states"], w["states"]
    room = [(a + truth[0, 0:2], b + truth[0, 0:2]) for a, b in liw.laser.room_segments(3)]   # room centred on the first pose
    T_il = np.array(synth.normalize_extrinsic(prm["T_imu_to_laser"])).reshape(4, 4)

    def points_at(k):
        T = np.eye(4)
        T[:3, :3] = synth.exp_so3(truth[k, 3:6])
        T[:3, 3] = truth[k, 0:3]
        rg, amin, inc = liw.laser.cast_scan(room, T @ T_il, seed=500 + k)
        return liw.laser.laser_to_points(rg, amin, inc, 0.0, float(k))[0]

    mgr = liw.laser.LaserManager(lp)
    frames, pts_all, frames_o, pts_all_o = [], [], [], []
    match_pose, match_pose_o = np.zeros((n, 12)), np.zeros((n, 12))
    for k in range(n):
        pts = points_at(k)
        s, so = liw.laser.Scan.spawn(lp, pts, float(k)), lorc.spawn_scan(pts, float(k))
        if k == 0:
            mgr.add_scan(s, est[0, 0:3], est[0, 3:6])
            lorc.add_scan(so, est[0, 0:3], est[0, 3:6])
            match_pose[0] = np.concatenate([est[0, 0:6], est[0, 0:6]])
            match_pose_o[0] = match_pose[0]
            first = (s, so)
            continue
        m = mgr.match_with_front(s, est[k, 0:3], est[k, 3:6])           # init topology ties frame k to the front key frame
        mo = lorc.match_with_front(so, est[k, 0:3], est[k, 3:6])
        assert len(m) == len(mo) >= 4 and np.array_equal(m.idx2, mo.idx2)
        pts_all.append(m.pts); frames += [k] * len(m); match_pose[k] = m.pose
        pts_all_o.append(mo.pts); frames_o += [k] * len(mo); match_pose_o[k] = mo.pose
    has_match = np.ones(n, dtype=np.uint8)
    has_match[0] = 0

    def window(pts_list, fr, mp):
        d = dict(w)
        d.update(laser_pts=np.concatenate(pts_list, axis=0), laser_frame=np.asarray(fr, dtype=np.int32), match_pose=mp.copy(), has_match=has_match,
                 states=est.copy())
        return d
    wg, wo = window(pts_all, frames, match_pose), window(pts_all_o, frames_o, match_pose_o)
    bs = liw.BatchSolver(prm, [wg])
    bs.solve(liw.LIW_MODE_INIT, 50)
    got = bs.states()[0]
    ow = pyoracle.Window(wo)
    orc.set_prior(None)
    orc.init_solve(ow)
    ref = ow["states"].reshape(n, 15)
    assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    # and the estimate moved towards the truth: relative pose error of the last frame w.r.t. frame 0 (gauge-free)
    def rel_xy(x):
        R0 = synth.exp_so3(x[0, 3:6])
        return R0.T @ (x[n - 1, 0:3] - x[0, 0:3])
    assert np.linalg.norm(rel_xy(got) - rel_xy(truth)) < np.linalg.norm(rel_xy(est) - rel_xy(truth))
