"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (2dliw-slam_amd/) never does.  PARITY UNPINNED: the reference has no golden vectors and
cannot be built here (see oracle/jet.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

dp = C.POINTER(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".h", ".cpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.oracle_create.restype = C.c_void_p
        _LIB.oracle_time_solves.restype = C.c_double
    return _LIB


class ParamsC(C.Structure):
    _fields_ = [("T_imu_to_wheel", C.c_double * 16), ("T_imu_to_laser", C.c_double * 16),
                ("g", C.c_double), ("line_to_line_sigma", C.c_double),
                ("manifold_p_sigma", C.c_double), ("manifold_q_sigma", C.c_double),
                ("imu_noise_acc_sigma", C.c_double * 3), ("imu_bias_acc_sigma", C.c_double * 3),
                ("imu_noise_gyro_sigma", C.c_double * 3), ("imu_bias_gyro_sigma", C.c_double * 3),
                ("wheel_sigma", C.c_double * 3), ("fast_mode", C.c_int), ("normalize_extrinsics", C.c_int)]


class WindowC(C.Structure):
    _fields_ = [("n", C.c_int), ("L", C.c_int), ("states", dp), ("laser_frame", C.POINTER(C.c_int)),
                ("laser_pts", dp), ("match_pose", dp), ("has_match", C.POINTER(C.c_ubyte)),
                ("imu_X", dp), ("imu_J", dp), ("imu_sqrtP", dp), ("imu_Dt", dp),
                ("wheel_T", dp), ("wheel_sqrtP", dp), ("wheel_Dt", dp)]


def _p(a):
    return a.ctypes.data_as(dp)


def params_struct(prm):
    """prm: dict with the keys of 2dliw-slam_amd Params (see synth.office_params())."""
    s = ParamsC()
    s.T_imu_to_wheel[:] = list(np.asarray(prm["T_imu_to_wheel"], dtype=np.float64).reshape(16))
    s.T_imu_to_laser[:] = list(np.asarray(prm["T_imu_to_laser"], dtype=np.float64).reshape(16))
    for k in ("g", "line_to_line_sigma", "manifold_p_sigma", "manifold_q_sigma"):
        setattr(s, k, float(prm[k]))
    for k in ("imu_noise_acc_sigma", "imu_bias_acc_sigma", "imu_noise_gyro_sigma", "imu_bias_gyro_sigma", "wheel_sigma"):
        getattr(s, k)[:] = [float(v) for v in prm[k]]
    s.fast_mode = int(bool(prm.get("fast_mode", False)))
    s.normalize_extrinsics = int(bool(prm.get("normalize_extrinsics", True)))
    return s


class Window:
    """Owns contiguous numpy arrays of one flat window and the matching WindowC view."""

    FIELDS = ("states", "laser_frame", "laser_pts", "match_pose", "has_match", "imu_X", "imu_J", "imu_sqrtP",
              "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt")

    def __init__(self, d):
        self.n = int(d["n"])
        self.a = {}
        for k in self.FIELDS:
            dt = np.int32 if k == "laser_frame" else (np.uint8 if k == "has_match" else np.float64)
            self.a[k] = np.ascontiguousarray(np.array(d[k], dtype=dt, copy=True))
        self.L = int(self.a["laser_frame"].shape[0])
        if self.n < 2:   # keep ctypes pointers valid for empty factor arrays
            for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
                if self.a[k].size == 0:
                    self.a[k] = np.zeros(1, dtype=np.float64)
        if self.L == 0:
            self.a["laser_frame"] = np.zeros(1, dtype=np.int32)
            self.a["laser_pts"] = np.zeros(12, dtype=np.float64)
        c = WindowC()
        c.n, c.L = self.n, self.L
        for k in self.FIELDS:
            if k == "laser_frame":
                c.laser_frame = self.a[k].ctypes.data_as(C.POINTER(C.c_int))
            elif k == "has_match":
                c.has_match = self.a[k].ctypes.data_as(C.POINTER(C.c_ubyte))
            else:
                setattr(c, k, _p(self.a[k]))
        self.c = c

    def __getitem__(self, k):
        return self.a[k]


class Oracle:
    def __init__(self, prm):
        self.L = lib()
        self._ps = params_struct(prm)
        self.h = C.c_void_p(self.L.oracle_create(C.byref(self._ps)))

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    def extrinsics(self):
        a, b = np.zeros(16), np.zeros(16)
        self.L.oracle_get_extrinsics(self.h, _p(a), _p(b))
        return a.reshape(4, 4), b.reshape(4, 4)

    # ---- single factors
    def eval_laser(self, pts12, pi, qi, pj, qj):
        pts12, pi, qi, pj, qj = [np.ascontiguousarray(v, dtype=np.float64) for v in (pts12, pi, qi, pj, qj)]
        res, jac = np.zeros(2), np.zeros(24)
        self.L.oracle_eval_laser(self.h, _p(pts12), _p(pi), _p(qi), _p(pj), _p(qj), _p(res), _p(jac))
        J = np.concatenate([jac[k * 6:(k + 1) * 6].reshape(2, 3) for k in range(4)], axis=1)
        return res, J

    def eval_imu(self, X, J, sqrtP, Dt, si, sj):
        X, J, sqrtP, si, sj = [np.ascontiguousarray(v, dtype=np.float64) for v in (X, J, sqrtP, si, sj)]
        res, jac = np.zeros(15), np.zeros(15 * 30)
        self.L.oracle_eval_imu(self.h, _p(X), _p(J), _p(sqrtP), C.c_double(Dt), _p(si), _p(sj), _p(res), _p(jac))
        return res, jac.reshape(15, 30)

    def eval_wheel(self, T12, sqrtP9, pi, qi, pj, qj):
        T12, sqrtP9, pi, qi, pj, qj = [np.ascontiguousarray(v, dtype=np.float64) for v in (T12, sqrtP9, pi, qi, pj, qj)]
        res, jac = np.zeros(3), np.zeros(36)
        self.L.oracle_eval_wheel(self.h, _p(T12), _p(sqrtP9), _p(pi), _p(qi), _p(pj), _p(qj), _p(res), _p(jac))
        return res, jac.reshape(3, 12)

    def eval_ground(self, p, q):
        p, q = [np.ascontiguousarray(v, dtype=np.float64) for v in (p, q)]
        res, jac = np.zeros(2), np.zeros(12)
        self.L.oracle_eval_ground(self.h, _p(p), _p(q), _p(res), _p(jac))
        return res, jac.reshape(2, 6)

    def exp_so3(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        R = np.zeros(9)
        self.L.oracle_exp_so3(_p(a), _p(R))
        return R.reshape(3, 3)

    def log_SO3(self, R):
        R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
        a = np.zeros(3)
        self.L.oracle_log_SO3(_p(R), _p(a))
        return a

    def so3_plus(self, x, d):
        x, d = [np.ascontiguousarray(v, dtype=np.float64) for v in (x, d)]
        out, jac = np.zeros(3), np.zeros(9)
        self.L.oracle_so3_plus(_p(x), _p(d), _p(out), _p(jac))
        return out, jac.reshape(3, 3)

    # ---- pre-integration
    def imu_preint(self, samples, t_start, t_end, bias6):
        samples = np.ascontiguousarray(samples, dtype=np.float64)
        bias6 = np.ascontiguousarray(bias6, dtype=np.float64)
        X, J, P, Dt = np.zeros(15), np.zeros(225), np.zeros(225), C.c_double(0)
        self.L.oracle_imu_preint(self.h, _p(samples), C.c_int(samples.shape[0]), C.c_double(t_start), C.c_double(t_end),
                                 _p(bias6), _p(X), _p(J), _p(P), C.byref(Dt))
        return X, J.reshape(15, 15), P.reshape(15, 15), Dt.value

    def wheel_preint(self, samples, t_start, t_end):
        samples = np.ascontiguousarray(samples, dtype=np.float64)
        T, P, Dt = np.zeros(12), np.zeros(9), C.c_double(0)
        self.L.oracle_wheel_preint(self.h, _p(samples), C.c_int(samples.shape[0]), C.c_double(t_start), C.c_double(t_end), _p(T), _p(P), C.byref(Dt))
        return T, P.reshape(3, 3), Dt.value

    # ---- window level
    def set_max_iterations(self, k):
        self.L.oracle_set_max_iterations(self.h, C.c_int(k))

    def set_threads(self, t):
        """OpenMP team for the residual-block evaluation (Ceres' num_threads) and the dense J^T J; 1 = the reference's configuration"""
        self.L.oracle_set_threads(self.h, C.c_int(int(t)))

    def init_solve(self, w):
        self.L.oracle_init_solve(self.h, C.byref(w.c))

    def solve(self, w):
        self.L.oracle_solve(self.h, C.byref(w.c))

    def marginalization(self, w):
        s = np.zeros(36)
        self.L.oracle_marginalization(self.h, C.byref(w.c), _p(s))
        return s.reshape(6, 6)

    def summary(self):
        t, s, c0, c1 = C.c_int(0), C.c_int(0), C.c_double(0), C.c_double(0)
        it = self.L.oracle_summary(self.h, C.byref(t), C.byref(s), C.byref(c0), C.byref(c1))
        return dict(iterations=it, termination=t.value, successful=s.value, initial_cost=c0.value, final_cost=c1.value)

    def iterations(self):
        out = []
        for k in range(self.L.oracle_iteration_count(self.h)):
            o7, x = np.zeros(7), np.zeros(4096)
            nx = self.L.oracle_iteration(self.h, C.c_int(k), _p(o7), _p(x), C.c_int(4096))
            out.append(dict(cost=o7[0], candidate_cost=o7[1], model_cost_change=o7[2], relative_decrease=o7[3],
                            radius=o7[4], valid=bool(o7[5]), successful=bool(o7[6]), x=x[:nx].copy()))
        return out

    def get_prior(self):
        X, J, R = np.zeros(15), np.zeros(225), np.zeros(15)
        has = self.L.oracle_get_prior(self.h, _p(X), _p(J), _p(R))
        return (X, J.reshape(15, 15), R) if has else None

    def set_prior(self, prior):
        if prior is None:
            z = np.zeros(225)
            self.L.oracle_set_prior(self.h, C.c_int(0), _p(z), _p(z), _p(z))
            return
        X, J, R = [np.ascontiguousarray(v, dtype=np.float64) for v in prior]
        self.L.oracle_set_prior(self.h, C.c_int(1), _p(X), _p(J.reshape(225)), _p(R))

    def marg_pieces(self):
        r, c = C.c_int(0), C.c_int(0)
        self.L.oracle_marg_dims(self.h, C.byref(r), C.byref(c))
        J, R = np.zeros((r.value, c.value)), np.zeros(r.value)
        H, g, dH, dg = np.zeros((c.value, c.value)), np.zeros(c.value), np.zeros((15, 15)), np.zeros(15)
        self.L.oracle_marg_get(self.h, _p(J), _p(R), _p(H), _p(g), _p(dH), _p(dg))
        return dict(J=J, R=R, H=H, g=g, Delta_H=dH, Delta_g=dg)

    def linearize(self, w, mode):
        N = 15 * w.n
        H, g, c = np.zeros((N, N)), np.zeros(N), C.c_double(0)
        self.L.oracle_linearize(self.h, C.byref(w.c), C.c_int(mode), _p(H), _p(g), C.byref(c))
        return H, g, c.value

    def time_solves(self, w, reps, max_iters=50, dense_product=True):
        it = C.c_int(0)
        sec = self.L.oracle_time_solves(self.h, C.byref(w.c), C.c_int(reps), C.c_int(max_iters), C.c_int(int(dense_product)), C.byref(it))
        return sec, it.value

    def time_solves_each(self, w, warmup, reps, max_iters=50, dense_product=True):
        """-> per-solve seconds [reps] (after `warmup` untimed solves), total LM iterations of the timed solves"""
        secs = np.zeros(max(reps, 1))
        it = self.L.oracle_time_solves_each(self.h, C.byref(w.c), C.c_int(warmup), C.c_int(reps), C.c_int(max_iters), C.c_int(int(dense_product)), _p(secs))
        return secs[:reps], int(it)


# ---------------------------------------------------------------------------------------------------
# laser front-end restatement (laser_frontend.h / laser_capi.cpp)
class LaserParamsC(C.Structure):
    _fields_ = [("w_laser_each_scan", C.c_double), ("h_laser_each_scan", C.c_double), ("laser_resolution", C.c_double),
                ("line_continuous_threshold", C.c_double), ("line_min_len", C.c_double), ("line_max_dis", C.c_double),
                ("line_max_tolerance_angle", C.c_double), ("ref_motion_filter_p", C.c_double), ("ref_motion_filter_q", C.c_double),
                ("ref_n_accumulation", C.c_int), ("T_imu_to_laser", C.c_double * 16), ("normalize_extrinsics", C.c_int)]


def _laser_lib():
    L = lib()
    if not getattr(L, "_laser_ready", False):
        for name in ("oracle_laser_create", "oracle_scan_spawn", "oracle_scan_create_empty", "oracle_laser_do_match", "oracle_laser_match_with",
                     "oracle_laser_ref_scan"):
            getattr(L, name).restype = C.c_void_p
        for name in ("oracle_laser_destroy", "oracle_scan_destroy", "oracle_laser_match_destroy"):
            getattr(L, name).argtypes = [C.c_void_p]
        L._laser_ready = True
    return L


class OracleScan:
    def __init__(self, h):
        self.h = C.c_void_p(h)

    def lines(self):
        L = _laser_lib()
        n = L.oracle_scan_num_lines(self.h)
        out = np.zeros((n, 10))
        if n:
            L.oracle_scan_get_lines(self.h, _p(out))
        return out

    def concers(self):
        L = _laser_lib()
        n = L.oracle_scan_num_concers(self.h)
        out = np.zeros((n, 3))
        if n:
            L.oracle_scan_get_concers(self.h, _p(out))
        return out

    def cell_lines(self, x, y, cap=16):
        ids = np.zeros(cap, dtype=np.int32)
        k = _laser_lib().oracle_scan_cell_lines(self.h, C.c_double(x), C.c_double(y), ids.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(cap))
        return k, ids[:max(0, min(k, cap))].copy()

    def add_segment(self, p1, p2, add_concers=False):
        a, b = np.ascontiguousarray(p1, dtype=np.float64), np.ascontiguousarray(p2, dtype=np.float64)
        return _laser_lib().oracle_scan_add_segment(self.h, _p(a), _p(b), C.c_int(int(add_concers)))

    def __del__(self):
        try:
            _laser_lib().oracle_scan_destroy(self.h)
        except Exception:
            pass


class OracleMatch:
    def __init__(self, h):
        L = _laser_lib()
        h = C.c_void_p(h)
        n = L.oracle_laser_match_size(h)
        self.pts, self.pose = np.zeros((n, 12)), np.zeros(12)
        self.idx1, self.idx2 = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        ipt = C.POINTER(C.c_int)
        L.oracle_laser_match_get(h, _p(self.pts), _p(self.pose), self.idx1.ctypes.data_as(ipt), self.idx2.ctypes.data_as(ipt))
        L.oracle_laser_match_destroy(h)

    def __len__(self):
        return self.pts.shape[0]


class LaserOracle:
    """laser_manager restatement; lp: dict with the keys of 2dliw-slam_amd.laser.office_laser_params()."""

    def __init__(self, lp):
        s = LaserParamsC()
        for k in ("w_laser_each_scan", "h_laser_each_scan", "laser_resolution", "line_continuous_threshold", "line_min_len", "line_max_dis",
                  "line_max_tolerance_angle", "ref_motion_filter_p", "ref_motion_filter_q"):
            setattr(s, k, float(lp[k]))
        s.ref_n_accumulation = int(lp["ref_n_accumulation"])
        s.T_imu_to_laser[:] = [float(v) for v in np.asarray(lp["T_imu_to_laser"], dtype=np.float64).reshape(16)]
        s.normalize_extrinsics = int(bool(lp.get("normalize_extrinsics", True)))
        self._ps = s
        self.h = C.c_void_p(_laser_lib().oracle_laser_create(C.byref(s)))

    def spawn_scan(self, points, time=0.0):
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        return OracleScan(_laser_lib().oracle_scan_spawn(self.h, _p(p), C.c_int(p.shape[0]), C.c_double(time)))

    def empty_scan(self, time=0.0):
        return OracleScan(_laser_lib().oracle_scan_create_empty(self.h, C.c_double(time)))

    def do_match(self, s1, s2, p1, q1, p2, q2, kk=0):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (p1, q1, p2, q2)]
        return OracleMatch(_laser_lib().oracle_laser_do_match(self.h, s1.h, s2.h, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), C.c_int(kk)))

    def add_scan(self, scan, p, q):
        a, b = np.ascontiguousarray(p, dtype=np.float64), np.ascontiguousarray(q, dtype=np.float64)
        _laser_lib().oracle_laser_add_scan(self.h, scan.h, _p(a), _p(b))

    def _match(self, which, scan, p, q):
        a, b = np.ascontiguousarray(p, dtype=np.float64), np.ascontiguousarray(q, dtype=np.float64)
        return OracleMatch(_laser_lib().oracle_laser_match_with(self.h, C.c_int(which), scan.h, _p(a), _p(b)))

    def match_with_front(self, scan, p, q):
        return self._match(0, scan, p, q)

    def match_with_back(self, scan, p, q):
        return self._match(1, scan, p, q)

    def match_with_ref(self, scan, p, q):
        return self._match(2, scan, p, q)

    def pop_scan(self):
        return _laser_lib().oracle_laser_pop_scan(self.h)

    def clear_all_scan(self):
        _laser_lib().oracle_laser_clear_all_scan(self.h)

    def num_keyframes(self):
        return _laser_lib().oracle_laser_num_keyframes(self.h)

    def ref_scan(self):
        p, q = np.zeros(3), np.zeros(3)
        h = _laser_lib().oracle_laser_ref_scan(self.h, _p(p), _p(q))
        if not h:
            return None
        return OracleScan(h), p, q

    def __del__(self):
        try:
            _laser_lib().oracle_laser_destroy(self.h)
        except Exception:
            pass


def laser_to_points(ranges, angle_min, angle_increment, time_increment, stamp):
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    pts, ts = np.zeros((len(r), 3)), np.zeros(len(r))
    m = _laser_lib().oracle_laser_to_points(r.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(len(r)), C.c_float(angle_min), C.c_float(angle_increment),
                                            C.c_float(time_increment), C.c_double(stamp), _p(pts), _p(ts))
    return pts[:m].copy(), ts[:m].copy()


def laser_correct(points, times, stamp, linear, angular):
    p = np.ascontiguousarray(points, dtype=np.float64).copy()
    t, lin, ang = (np.ascontiguousarray(v, dtype=np.float64) for v in (times, linear, angular))
    _laser_lib().oracle_laser_correct(_p(p), _p(t), C.c_int(len(t)), C.c_double(stamp), _p(lin), _p(ang))
    return p


# ---------------------------------------------------------------------------------------------------
# on-disk formats (io_formats.h)
def tum_line(T_imu_to_wheel16, normalize, time, p, q):
    buf = C.create_string_buffer(400)
    T, a, b = (np.ascontiguousarray(v, dtype=np.float64) for v in (T_imu_to_wheel16, p, q))
    n = lib().oracle_tum_line(_p(T), C.c_int(int(normalize)), C.c_double(time), _p(a), _p(b), buf, C.c_int(400))
    return buf.raw[:n].decode()


class OracleRecord:
    def __init__(self):
        L = lib()
        L.oracle_record_create.restype = C.c_void_p
        self.h = C.c_void_p(L.oracle_record_create())

    def add_time(self, name, us):
        lib().oracle_record_add_time(self.h, name.encode(), C.c_ulonglong(int(us)))

    def add_record(self, name, v):
        lib().oracle_record_add(self.h, name.encode(), C.c_ulonglong(int(v)))

    def dump(self):
        n = lib().oracle_record_dump(self.h, None, C.c_int(0))
        buf = C.create_string_buffer(n + 1)
        lib().oracle_record_dump(self.h, buf, C.c_int(n + 1))
        return buf.raw[:n].decode()


# ---------------------------------------------------------------------------------------------------
# trajectory restatement (trajectory.h)
class TrajParamsC(C.Structure):
    _fields_ = [("slide_window_size", C.c_int), ("p_motion_threshold", C.c_double), ("q_motion_threshold", C.c_double),
                ("key_frame_p_motion_threshold", C.c_double), ("key_frame_q_motion_threshold", C.c_double), ("min_delta_t", C.c_double),
                ("keep_window_size", C.c_int)]


class TrajectoryOracle:
    """lvio_2d::trajectory restatement; prm / lp dicts as for Oracle / LaserOracle."""

    def __init__(self, prm, lp, slide_window_size=10, p_motion_threshold=0.1, q_motion_threshold=0.05, key_frame_p_motion_threshold=0.05,
                 key_frame_q_motion_threshold=0.05, min_delta_t=0.001, keep_window_size=1):
        L = lib()
        L.oracle_traj_create.restype = C.c_void_p
        ps = params_struct(prm)
        ls = LaserOracle.__new__(LaserOracle)   # reuse the struct filling code
        s = LaserParamsC()
        for k in ("w_laser_each_scan", "h_laser_each_scan", "laser_resolution", "line_continuous_threshold", "line_min_len", "line_max_dis",
                  "line_max_tolerance_angle", "ref_motion_filter_p", "ref_motion_filter_q"):
            setattr(s, k, float(lp[k]))
        s.ref_n_accumulation = int(lp["ref_n_accumulation"])
        s.T_imu_to_laser[:] = [float(v) for v in np.asarray(lp["T_imu_to_laser"], dtype=np.float64).reshape(16)]
        s.normalize_extrinsics = int(bool(lp.get("normalize_extrinsics", True)))
        del ls
        tp = TrajParamsC(int(slide_window_size), p_motion_threshold, q_motion_threshold, key_frame_p_motion_threshold, key_frame_q_motion_threshold, min_delta_t,
                         int(keep_window_size))
        self.h = C.c_void_p(L.oracle_traj_create(C.byref(ps), C.byref(s), C.byref(tp)))

    def add_imu(self, t, acc, gyro):
        a, b = np.ascontiguousarray(acc, dtype=np.float64), np.ascontiguousarray(gyro, dtype=np.float64)
        lib().oracle_traj_add_imu(self.h, C.c_double(t), _p(a), _p(b))

    def add_wheel(self, t, R, tr):
        a, b = np.ascontiguousarray(R, dtype=np.float64).reshape(9), np.ascontiguousarray(tr, dtype=np.float64)
        lib().oracle_traj_add_wheel(self.h, C.c_double(t), _p(a), _p(b))

    def add_laser(self, t, points, times):
        p, ts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3), np.ascontiguousarray(times, dtype=np.float64)
        lib().oracle_traj_add_laser(self.h, C.c_double(t), _p(p), _p(ts), C.c_int(len(ts)))

    def counters(self):
        out = (C.c_int * 5)()
        lib().oracle_traj_counters(self.h, out)
        return dict(status=out[0], frames=out[1], tracked=out[2], initializations=out[3], keyframes=out[4])

    def current(self):
        t, s = C.c_double(0), np.zeros(15)
        lib().oracle_traj_current(self.h, C.byref(t), _p(s))
        return t.value, s

    def tum(self):
        n = lib().oracle_traj_tum(self.h, None, C.c_int(0))
        buf = C.create_string_buffer(n + 1)
        lib().oracle_traj_tum(self.h, buf, C.c_int(n + 1))
        return buf.raw[:n].decode()

    def last_iterations(self):
        return lib().oracle_traj_last_iterations(self.h)

    def set_capture(self, on=True):
        lib().oracle_traj_set_capture(self.h, C.c_int(int(on)))

    def captures(self):
        """-> list of dicts: the flat window + prior every tracking solve started from, and what it produced"""
        L = lib()
        names = ("states", "laser_pts", "match_pose", "imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt", "prior_X", "prior_J",
                 "prior_R", "states_after", "match_after", "Delta_H", "Delta_g", "post_X", "post_J", "post_R", "laser_frame", "has_match")
        out = []
        for k in range(L.oracle_traj_capture_count(self.h)):
            d5 = (C.c_int * 5)()
            L.oracle_traj_capture_dims(self.h, C.c_int(k), d5)
            rec = dict(n=d5[0], L=d5[1], has_prior=d5[2], iterations=d5[3], termination=d5[4])
            for fid, nm in enumerate(names):
                sz = L.oracle_traj_capture_field(self.h, C.c_int(k), C.c_int(fid), None, C.c_int(0))
                a = np.zeros(max(sz, 1))
                L.oracle_traj_capture_field(self.h, C.c_int(k), C.c_int(fid), _p(a), C.c_int(sz))
                rec[nm] = a[:sz].copy()
            rec["laser_frame"] = rec["laser_frame"].astype(np.int32)
            rec["has_match"] = rec["has_match"].astype(np.uint8)
            rec["laser_pts"] = rec["laser_pts"].reshape(-1, 12)
            out.append(rec)
        return out

    def enable_backend(self, pg, loops, solve_period=10.0, max_iterations=0):
        """pg: dict as posegraph.office_pg_params(); loops: list of (trigger key-frame index, older index, tf12[12])"""
        s = PgParamsC()
        s.loop_sigma_p[:] = [float(v) for v in pg["loop_sigma_p"]]
        s.loop_sigma_q[:] = [float(v) for v in pg["loop_sigma_q"]]
        s.loop_edge_k = float(pg["loop_edge_k"])
        s.use_ground_p_factor = int(bool(pg["use_ground_p_factor"]))
        s.use_ground_q_factor = int(bool(pg["use_ground_q_factor"]))
        idx = np.ascontiguousarray([[l[0], l[1]] for l in loops] or [[0, 0]], dtype=np.int32)
        tf = np.ascontiguousarray([l[2] for l in loops] or [np.zeros(12)], dtype=np.float64)
        lib().oracle_traj_enable_backend(self.h, C.byref(s), C.c_double(solve_period), C.c_int(max_iterations), C.c_int(len(loops)),
                                         idx.ctypes.data_as(C.POINTER(C.c_int)), _p(tf))

    def backend(self, cap=4096):
        out4, mod, poses, times, cur = (C.c_int * 4)(), np.zeros(12), np.zeros((cap, 6)), np.zeros(cap), np.zeros(6)
        n = lib().oracle_traj_backend(self.h, out4, _p(mod), _p(poses), C.c_int(cap), _p(times), _p(cur))
        return dict(keyframes=out4[0], loops=out4[1], solves=out4[2], iterations=out4[3], modify_delta_tf=mod, poses=poses[:n].copy(), times=times[:n].copy(),
                    current=cur)

    def __del__(self):
        try:
            lib().oracle_traj_destroy(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------
# pose-graph restatement (posegraph.h)
class PgParamsC(C.Structure):
    _fields_ = [("loop_sigma_p", C.c_double * 3), ("loop_sigma_q", C.c_double * 3), ("loop_edge_k", C.c_double),
                ("use_ground_p_factor", C.c_int), ("use_ground_q_factor", C.c_int)]


def posegraph_solve(orc, pg, poses, seq_idx, seq_tf12, loop_idx=None, loop_tf12=None, max_iters=0):
    """orc: Oracle (extrinsics / ground sigmas); pg: dict as 2dliw-slam_amd.posegraph.office_pg_params()."""
    s = PgParamsC()
    s.loop_sigma_p[:] = [float(v) for v in pg["loop_sigma_p"]]
    s.loop_sigma_q[:] = [float(v) for v in pg["loop_sigma_q"]]
    s.loop_edge_k = float(pg["loop_edge_k"])
    s.use_ground_p_factor = int(bool(pg["use_ground_p_factor"]))
    s.use_ground_q_factor = int(bool(pg["use_ground_q_factor"]))
    x = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6).copy()
    si, st = np.ascontiguousarray(seq_idx, dtype=np.int32).reshape(-1, 2), np.ascontiguousarray(seq_tf12, dtype=np.float64).reshape(-1, 12)
    nl = 0 if loop_idx is None else len(loop_idx)
    li = np.ascontiguousarray(loop_idx if nl else np.zeros((1, 2)), dtype=np.int32).reshape(-1, 2)
    lt = np.ascontiguousarray(loop_tf12 if nl else np.zeros((1, 12)), dtype=np.float64).reshape(-1, 12)
    out3, cost2 = (C.c_int * 3)(), (C.c_double * 2)()
    ipt = C.POINTER(C.c_int)
    lib().oracle_posegraph_solve(orc.h, C.byref(s), C.c_int(x.shape[0]), _p(x), C.c_int(si.shape[0]), si.ctypes.data_as(ipt), _p(st), C.c_int(nl),
                                 li.ctypes.data_as(ipt), _p(lt), C.c_int(max_iters), out3, cost2)
    return x, dict(iterations=out3[0], successful=out3[1], termination=out3[2], initial_cost=cost2[0], final_cost=cost2[1])


def backend_run(orc, pg, times, poses, loops, solve_period=10.0, max_iterations=0):
    """oracle/keyframe_manager.h alone on a list of key frames (time, [p q]); loops: (trigger, older, tf12[12]).
    -> dict(keyframes, loops, solves, iterations, modify_delta_tf, poses)"""
    s = PgParamsC()
    s.loop_sigma_p[:] = [float(v) for v in pg["loop_sigma_p"]]
    s.loop_sigma_q[:] = [float(v) for v in pg["loop_sigma_q"]]
    s.loop_edge_k = float(pg["loop_edge_k"])
    s.use_ground_p_factor = int(bool(pg["use_ground_p_factor"]))
    s.use_ground_q_factor = int(bool(pg["use_ground_q_factor"]))
    t = np.ascontiguousarray(times, dtype=np.float64)
    x = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
    idx = np.ascontiguousarray([[l[0], l[1]] for l in loops] or [[0, 0]], dtype=np.int32)
    tf = np.ascontiguousarray([l[2] for l in loops] or [np.zeros(12)], dtype=np.float64)
    out4, mod, po = (C.c_int * 4)(), np.zeros(12), np.zeros_like(x)
    lib().oracle_backend_run(orc.h, C.byref(s), C.c_double(solve_period), C.c_int(max_iterations), C.c_int(x.shape[0]), _p(t), _p(x), C.c_int(len(loops)),
                             idx.ctypes.data_as(C.POINTER(C.c_int)), _p(tf), out4, _p(mod), _p(po))
    return dict(keyframes=out4[0], loops=out4[1], solves=out4[2], iterations=out4[3], modify_delta_tf=mod, poses=po)


def posegraph_linearize(orc, pg, poses, seq_idx, seq_tf12, loop_idx=None, loop_tf12=None):
    """-> H [nt, nt], g [nt], cost, idx [nt] (flat index into the [N, 6] pose array of every tangent entry)"""
    s = PgParamsC()
    s.loop_sigma_p[:] = [float(v) for v in pg["loop_sigma_p"]]
    s.loop_sigma_q[:] = [float(v) for v in pg["loop_sigma_q"]]
    s.loop_edge_k = float(pg["loop_edge_k"])
    s.use_ground_p_factor = int(bool(pg["use_ground_p_factor"]))
    s.use_ground_q_factor = int(bool(pg["use_ground_q_factor"]))
    x = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
    si, st = np.ascontiguousarray(seq_idx, dtype=np.int32).reshape(-1, 2), np.ascontiguousarray(seq_tf12, dtype=np.float64).reshape(-1, 12)
    nl = 0 if loop_idx is None else len(loop_idx)
    li = np.ascontiguousarray(loop_idx if nl else np.zeros((1, 2)), dtype=np.int32).reshape(-1, 2)
    lt = np.ascontiguousarray(loop_tf12 if nl else np.zeros((1, 12)), dtype=np.float64).reshape(-1, 12)
    ipt = C.POINTER(C.c_int)
    n = 6 * x.shape[0]
    H, g, cost, idx = np.zeros((n, n)), np.zeros(n), C.c_double(0), np.zeros(n, dtype=np.int32)
    nt = lib().oracle_posegraph_linearize(orc.h, C.byref(s), C.c_int(x.shape[0]), _p(x), C.c_int(si.shape[0]), si.ctypes.data_as(ipt), _p(st), C.c_int(nl),
                                          li.ctypes.data_as(ipt), _p(lt), _p(H), _p(g), C.byref(cost), idx.ctypes.data_as(ipt))
    return H.reshape(-1)[:nt * nt].reshape(nt, nt).copy(), g[:nt].copy(), cost.value, idx[:nt].copy()
