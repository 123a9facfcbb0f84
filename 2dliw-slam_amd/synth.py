"""Synthetic sliding-window workloads (BASELINE.json configs C2/C4/C5 and small parity cases).

Deterministic (seeded) generator of ONE flat window in the layout of include/liw_window.h
(`liw_window`): n frame states, L `laser_factor` blocks (reference src/factor/laser_factor.h:26-100;
2 point-to-line rows each), n-1 pre-integrated IMU and wheel blocks produced by running a
pre-integrator (the caller supplies it: the product's host pre-integrator for bench.py, the oracle's for
tests) over synthetic 200 Hz IMU / ~20 Hz wheel-odometry samples, and per-frame `laser_match` poses.
Recipe: SURVEY.md §8(d) "Synthetic C2 input".  Extrinsics / noise: reference config/office.yaml:13-31,
:39-64, :90.
"""
import numpy as np

OFFICE_T_IMU_TO_WHEEL = [0.0040697, -0.9998940, -0.0139789, -0.061,
                         0.0099712, 0.0140189, -0.9998520, 0.919,
                         0.9999420, 0.0039297, 0.0100272, -0.224,
                         0.0, 0.0, 0.0, 1.0]
OFFICE_T_IMU_TO_LASER = [0.0019070, -0.9999900, 0.0040438, 0.024,
                         0.0459794, -0.0039519, -0.9989346, -0.078,
                         0.9989406, 0.0020909, 0.0459714, -0.071,
                         0.0, 0.0, 0.0, 1.0]


def office_params():
    """The parameters the hot path reads, values of reference config/office.yaml."""
    return dict(T_imu_to_wheel=list(OFFICE_T_IMU_TO_WHEEL), T_imu_to_laser=list(OFFICE_T_IMU_TO_LASER),
                g=9.8, line_to_line_sigma=0.001, manifold_p_sigma=0.01, manifold_q_sigma=0.0005,
                imu_noise_acc_sigma=[0.0163] * 3, imu_bias_acc_sigma=[0.00499] * 3,
                imu_noise_gyro_sigma=[0.003208] * 3, imu_bias_gyro_sigma=[0.000499] * 3,
                wheel_sigma=[0.5, 99999.0, 999.99], fast_mode=False, normalize_extrinsics=True)


# ---------------------------------------------------------------- small SO3/SE3 helpers (numpy)
def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def exp_so3(w):
    th = float(np.linalg.norm(w))
    K = hat(w)
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1.0 - np.cos(th)) / (th * th) * (K @ K)


def log_so3(R):
    c = max(-1.0, min(1.0, (np.trace(R) - 1.0) * 0.5))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * v
    if np.pi - th < 1e-6:  # near pi: take the axis from the symmetric part
        A = (R + np.eye(3)) * 0.5
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[:, k] / ax[k]
        ax = ax / np.linalg.norm(ax)
        if np.dot(ax, v) < 0:
            ax = -ax
        return ax * th
    return v * (th / (2.0 * np.sin(th)))


def se3(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def inv_se3(T):
    R, t = T[:3, :3], T[:3, 3]
    return se3(R.T, -R.T @ t)


def normalize_extrinsic(T16):
    """Projection of the 4x4 parameter onto SE3 used for data generation only."""
    T = np.asarray(T16, dtype=np.float64).reshape(4, 4).copy()
    U, _, Vt = np.linalg.svd(T[:3, :3])
    T[:3, :3] = U @ Vt
    return T


# ---------------------------------------------------------------- trajectory ground truth
MOTIONS = ("arc", "stationary", "rotate", "translate", "standstill_then_go", "stop_and_go")


class _Truth:
    """Ground-truth base (wheel / odom frame) trajectory T_w_o(t).

    motion: "arc" (default; circle of `radius`, yaw rate `yaw_rate`), "stationary" (the robot does not move at all),
    "rotate" (turning on the spot: the wheel frame's origin stays put, so the translation of the relative wheel pose is 0 — the arm of
    reference src/factor/wheel_factor.h:52,58 with |dp| < 1e-4), "translate" (straight line, no rotation: the arm of :63 with
    |dq| < 1e-3), "standstill_then_go" (at rest until t_go, then the arc with a smooth start: the first frames of a log recorded from a
    parked robot), "stop_and_go" (the arc, braking to a halt at t_stop, at rest for `pause` seconds, then on: a standstill while
    TRACKING).  The path parameter s(t) carries all time dependence, so a standstill is exactly constant."""

    def __init__(self, prm, yaw_rate=0.3, radius=5.0, motion="arc", t_go=2.0, speed=1.5, t_stop=3.0, pause=2.0):
        assert motion in MOTIONS, motion
        self.T_i_o = normalize_extrinsic(prm["T_imu_to_wheel"])
        self.T_o_i = inv_se3(self.T_i_o)
        self.T_i_l = normalize_extrinsic(prm["T_imu_to_laser"])
        self.w = yaw_rate
        self.r = radius
        self.g = prm["g"]
        self.motion = motion
        self.t_go = t_go
        self.speed = speed
        self.t_stop = t_stop
        self.pause = pause

    def s(self, t):
        if self.motion == "stationary":
            return 0.0
        if self.motion == "standstill_then_go":
            u, tau = t - self.t_go, 0.5
            if u <= 0.0:
                return 0.0
            return u * u / (2.0 * tau) if u < tau else u - 0.5 * tau
        if self.motion == "stop_and_go":      # speed 1 -> linear ramp to 0 over tau before t_stop -> 0 for `pause` -> ramp up over tau
            tau, a, b = 0.5, self.t_stop - 0.5, self.t_stop + self.pause
            if t <= a:
                return t
            s_stop = a + 0.5 * tau
            if t <= self.t_stop:
                u = t - a
                return a + u - u * u / (2.0 * tau)
            if t <= b:
                return s_stop
            u = t - b
            return s_stop + (u * u / (2.0 * tau) if u < tau else u - 0.5 * tau)
        return t

    def T_w_o(self, t):
        s = self.s(t)
        if self.motion == "translate":
            return se3(np.eye(3), np.array([self.speed * s, 0.0, 0.0]))
        psi = self.w * s
        if self.motion == "rotate":
            return se3(exp_so3(np.array([0.0, 0.0, psi])), np.zeros(3))
        roll = 1e-3 * np.sin(2.0 * s)
        pitch = 1e-3 * np.cos(1.5 * s)
        R = exp_so3(np.array([0.0, 0.0, psi])) @ exp_so3(np.array([roll, pitch, 0.0]))
        p = np.array([self.r * np.sin(psi), self.r * (1.0 - np.cos(psi)), 1e-3 * np.sin(3.0 * s)])
        return se3(R, p)

    def T_w_i(self, t):
        return self.T_w_o(t) @ self.T_o_i

    def imu(self, t):
        h = 1e-4
        Tm, T0, Tp = self.T_w_i(t - h), self.T_w_i(t), self.T_w_i(t + h)
        acc_w = (Tp[:3, 3] - 2.0 * T0[:3, 3] + Tm[:3, 3]) / (h * h)
        gyro = log_so3(Tm[:3, :3].T @ Tp[:3, :3]) / (2.0 * h)
        acc = T0[:3, :3].T @ (acc_w + np.array([0.0, 0.0, self.g]))   # +g z: reference imu_factor.h:41-42,79-80
        return acc, gyro

    def vel(self, t):
        h = 1e-5
        return (self.T_w_i(t + h)[:3, 3] - self.T_w_i(t - h)[:3, 3]) / (2.0 * h)


def ragged_frame_counts(rng, n, L, p_empty=0.15, spread=3.0):
    """Laser blocks per owning frame of a ragged window: frame 0 owns none (init topology), every other frame is empty with probability
    p_empty, the rest share L blocks with weights u ** spread (u uniform): a few frames with hundreds of matched lines next to frames with
    a handful — the shape real scans give do_match (laser_manager.cpp:316-345), unlike the even spread of the C2 recipe."""
    w = rng.uniform(0.0, 1.0, n) ** spread
    w[rng.uniform(0.0, 1.0, n) < p_empty] = 0.0
    w[0] = 0.0
    if w.sum() == 0.0:
        w[n - 1] = 1.0
    return rng.multinomial(int(L), w / w.sum()).astype(np.int64)


def make_window(preint, prm=None, seed=20240, n=30, L=2000, frame_dt=0.1, imu_rate=200.0, wheel_period=0.0505,
                laser_on_frame0=False, state_noise=1.0, motion="arc", state_motion=None, odom_noise=2e-4,
                state_p_sigma=0.02, state_q_sigma=np.deg2rad(0.5), t0=1.0, frame_counts=None):
    """Returns a dict of numpy arrays in the flat `liw_window` layout (+ 'truth_states').

    preint: object with imu_preint(samples[N,7], t_start, t_end, bias6) -> (X15, J15x15, sqrtP15x15, Dt)
            and wheel_preint(samples[N,13], t_start, t_end) -> (T12, sqrtP3x3, Dt)   (row-major matrices).
    laser blocks are spread over frames 1..n-1 (init topology ties them to frame 0; frame 0 owns none
    unless laser_on_frame0) and sorted by owning frame.

    motion / state_motion (see _Truth): the sensors (IMU, wheel odometry, matched lines) follow `motion`; the frame STATES the solver
    starts from are the truth of `state_motion` (default: the same) + N(0, state_noise * [state_p_sigma, state_q_sigma]).  Different
    values give the mixed arms of reference src/factor/wheel_factor.h:45-66: the odometry increment at rest while the states move
    (motion="stationary", state_motion="arc") and the mirror.  odom_noise: position noise of the odometry samples (0 for the
    identical readings of a parked robot).
    frame_counts: laser blocks PER owning frame (length n; overrides L and the even spread) — ragged scans: frames with no matched
    line at all next to frames with hundreds (real scans: the count is whatever do_match keeps, laser_manager.cpp:316-345).
    """
    prm = prm or office_params()
    rng = np.random.default_rng(seed)
    tr = _Truth(prm, motion=motion)
    trs = tr if state_motion in (None, motion) else _Truth(prm, motion=state_motion)
    times = t0 + frame_dt * np.arange(n)

    # ---- frame states: truth + perturbation
    true_bias = rng.normal(0.0, 1e-3, 6)
    truth = np.zeros((n, 15))
    states = np.zeros((n, 15))
    for k in range(n):
        T = trs.T_w_i(times[k])
        truth[k, 0:3] = T[:3, 3]
        truth[k, 3:6] = log_so3(T[:3, :3])
        truth[k, 6:9] = trs.vel(times[k])
        truth[k, 9:15] = true_bias
        Rn = T[:3, :3] @ exp_so3(rng.normal(0.0, state_q_sigma * state_noise, 3))
        states[k, 0:3] = T[:3, 3] + rng.normal(0.0, state_p_sigma * state_noise, 3)
        states[k, 3:6] = log_so3(Rn)
        states[k, 6:9] = truth[k, 6:9] + rng.normal(0.0, 0.05 * state_noise, 3)
        states[k, 9:15] = rng.normal(0.0, 1e-3, 6)

    # ---- IMU / wheel blocks (entry k: frames k -> k+1)
    imu_X, imu_J, imu_P, imu_Dt = np.zeros((max(n - 1, 0), 15)), np.zeros((max(n - 1, 0), 225)), np.zeros((max(n - 1, 0), 225)), np.zeros(max(n - 1, 0))
    wheel_T, wheel_P, wheel_Dt = np.zeros((max(n - 1, 0), 12)), np.zeros((max(n - 1, 0), 9)), np.zeros(max(n - 1, 0))
    n_imu = int(round(frame_dt * imu_rate))
    for k in range(n - 1):
        ts = times[k] + np.arange(n_imu) / imu_rate
        samples = np.zeros((n_imu, 7))
        for i, t in enumerate(ts):
            acc, gyro = tr.imu(t)
            samples[i, 0] = t
            samples[i, 1:4] = acc + true_bias[0:3] + rng.normal(0.0, 0.01, 3)
            samples[i, 4:7] = gyro + true_bias[3:6] + rng.normal(0.0, 0.001, 3)
        X, J, P, Dt = preint.imu_preint(samples, times[k], times[k + 1], states[k, 9:15])
        imu_X[k], imu_J[k], imu_P[k], imu_Dt[k] = X, np.asarray(J).reshape(225), np.asarray(P).reshape(225), Dt
        # wheel odometry: poses of the base in the odom(=world) frame, first sample a bit before the interval
        tw = np.arange(times[k] - 2.5 * wheel_period, times[k + 1], wheel_period)
        ws = np.zeros((len(tw), 13))
        for i, t in enumerate(tw):
            T = tr.T_w_o(t)
            ws[i, 0] = t
            ws[i, 1:10] = T[:3, :3].reshape(9)
            ws[i, 10:13] = T[:3, 3] + rng.normal(0.0, 1.0, 3) * odom_noise
        # samples before times[k] only establish the twist; the accumulator is reset at times[k] (trajectory.cpp:176-184)
        T12, P9, Dtw = preint.wheel_preint(ws, times[k], times[k + 1])
        wheel_T[k], wheel_P[k], wheel_Dt[k] = T12, np.asarray(P9).reshape(9), Dtw

    # ---- laser blocks
    first = 0 if laser_on_frame0 else 1
    owners = np.sort(first + (np.arange(L) % max(n - first, 1))) if (L > 0 and n > first) else np.zeros(0, dtype=np.int64)
    if frame_counts is not None:
        fc = np.asarray(frame_counts, dtype=np.int64)
        assert fc.shape == (n,) and (fc >= 0).all() and (laser_on_frame0 or fc[0] == 0)
        owners = np.repeat(np.arange(n), fc)
        L = int(fc.sum())
    laser_frame = owners.astype(np.int32)
    laser_pts = np.zeros((L, 12))
    T_w_l0 = tr.T_w_i(times[0]) @ tr.T_i_l
    for j in range(L):
        k = int(laser_frame[j])
        T_w_lk = tr.T_w_i(times[k]) @ tr.T_i_l
        rngd, bearing = rng.uniform(2.0, 10.0), rng.uniform(-np.pi, np.pi)
        mid = np.array([rngd * np.cos(bearing), rngd * np.sin(bearing), 0.0])
        ori, length = rng.uniform(-np.pi, np.pi), rng.uniform(0.2, 3.0)
        d = 0.5 * length * np.array([np.cos(ori), np.sin(ori), 0.0])
        l2_p1, l2_p2 = mid - d, mid + d
        # matched segment in frame 0's laser frame: same supporting line (true relative pose), end-points slid
        # along it, 5 mm end-point noise -> no residual is exactly 0.  The pose error the optimiser has to
        # remove is the N(0,[2 cm, 0.5 deg]) perturbation of the frame STATES above (one per frame, so the
        # blocks of a frame are mutually consistent, as matched lines of one real scan are).
        T_l0_lk = inv_se3(T_w_l0) @ T_w_lk
        s1, s2 = rng.uniform(-0.3, 0.3, 2)
        a = l2_p1 + s1 * (l2_p2 - l2_p1)
        b = l2_p2 + s2 * (l2_p2 - l2_p1)
        l1_p1 = T_l0_lk[:3, :3] @ a + T_l0_lk[:3, 3] + rng.normal(0.0, 0.005, 3)
        l1_p2 = T_l0_lk[:3, :3] @ b + T_l0_lk[:3, 3] + rng.normal(0.0, 0.005, 3)
        l1_p1[2] = 0.0   # 2D scans: z = 0 (reference src/utilies/common.cpp:22-24)
        l1_p2[2] = 0.0
        laser_pts[j] = np.concatenate([l1_p1, l1_p2, l2_p1, l2_p2])

    # ---- laser_match poses: p1,q1 = reference (frame 0) pose, p2,q2 = own pose (initial guesses)
    has_match = np.zeros(n, dtype=np.uint8)
    match_pose = np.zeros((n, 12))
    for k in range(n):
        if k >= first:
            has_match[k] = 1
        match_pose[k, 0:6] = states[0, 0:6]
        match_pose[k, 6:12] = states[k, 0:6]
    return dict(n=n, states=states, laser_frame=laser_frame, laser_pts=laser_pts, match_pose=match_pose,
                has_match=has_match, imu_X=imu_X, imu_J=imu_J, imu_sqrtP=imu_P, imu_Dt=imu_Dt,
                wheel_T=wheel_T, wheel_sqrtP=wheel_P, wheel_Dt=wheel_Dt, truth_states=truth, times=times)
