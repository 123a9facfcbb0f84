"""Host mirror of the reference's 2D laser front-end over the C ABI of include/liw_laser.h:
`Scan` (lvio_2d::scan), `LaserMatch` (lvio_2d::laser_match), `LaserManager` (lvio_2d::laser_manager,
reference src/trajectory/laser_manager.h:9-49) and the LaserScan -> points helpers.  CPU code."""
import ctypes as C

import numpy as np

LASER_EXPORTS = [
    "liw_laser_to_points", "liw_laser_correct", "liw_scan_spawn", "liw_scan_create_empty", "liw_scan_add_segment", "liw_scan_destroy",
    "liw_scan_num_lines", "liw_scan_get_lines", "liw_scan_num_concers", "liw_scan_get_concers", "liw_scan_cell_lines",
    "liw_laser_do_match", "liw_laser_match_destroy", "liw_laser_match_size", "liw_laser_match_get", "liw_laser_manager_create",
    "liw_laser_manager_destroy", "liw_laser_manager_add_scan", "liw_laser_manager_match_with_front", "liw_laser_manager_match_with_back",
    "liw_laser_manager_match_with_ref", "liw_laser_manager_pop_scan", "liw_laser_manager_clear_all_scan", "liw_laser_manager_num_keyframes",
    "liw_laser_manager_ref_scan", "liw_laser_manager_set_keyframe_pose",
]


class LaserParamsC(C.Structure):
    _fields_ = [("w_laser_each_scan", C.c_double), ("h_laser_each_scan", C.c_double), ("laser_resolution", C.c_double),
                ("line_continuous_threshold", C.c_double), ("line_min_len", C.c_double), ("line_max_dis", C.c_double),
                ("line_max_tolerance_angle", C.c_double), ("ref_motion_filter_p", C.c_double), ("ref_motion_filter_q", C.c_double),
                ("ref_n_accumulation", C.c_int), ("T_imu_to_laser", C.c_double * 16), ("normalize_extrinsics", C.c_int)]


def office_laser_params(prm=None):
    """Laser front-end parameters of reference config/office.yaml:78-122 (+ the extrinsic of `prm`)."""
    from . import synth
    prm = prm or synth.office_params()
    return dict(w_laser_each_scan=100.0, h_laser_each_scan=100.0, laser_resolution=0.05, line_continuous_threshold=0.1, line_min_len=0.05,
                line_max_dis=0.03, line_max_tolerance_angle=175.0, ref_motion_filter_p=0.01, ref_motion_filter_q=0.01, ref_n_accumulation=2,
                T_imu_to_laser=list(prm["T_imu_to_laser"]), normalize_extrinsics=bool(prm.get("normalize_extrinsics", True)))


def laser_params_struct(lp, cls=LaserParamsC):
    if isinstance(lp, cls):
        return lp
    s = cls()
    for k in ("w_laser_each_scan", "h_laser_each_scan", "laser_resolution", "line_continuous_threshold", "line_min_len", "line_max_dis",
              "line_max_tolerance_angle", "ref_motion_filter_p", "ref_motion_filter_q"):
        setattr(s, k, float(lp[k]))
    s.ref_n_accumulation = int(lp["ref_n_accumulation"])
    s.T_imu_to_laser[:] = [float(v) for v in np.asarray(lp["T_imu_to_laser"], dtype=np.float64).reshape(16)]
    s.normalize_extrinsics = int(bool(lp.get("normalize_extrinsics", True)))
    return s


def _lib():
    from . import lib
    L = lib()
    if not getattr(L, "_laser_ready", False):
        for name in ("liw_scan_spawn", "liw_scan_create_empty", "liw_laser_do_match", "liw_laser_manager_create", "liw_laser_manager_match_with_front",
                     "liw_laser_manager_match_with_back", "liw_laser_manager_match_with_ref", "liw_laser_manager_ref_scan"):
            getattr(L, name).restype = C.c_void_p
        for name in ("liw_scan_destroy", "liw_laser_match_destroy", "liw_laser_manager_destroy"):
            getattr(L, name).argtypes = [C.c_void_p]
        L._laser_ready = True
    return L


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def laser_to_points(ranges, angle_min, angle_increment, time_increment, stamp):
    """convert::laser_to_point_times: -> (points [m,3], times [m])"""
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    pts, ts = np.zeros((len(r), 3)), np.zeros(len(r))
    m = _lib().liw_laser_to_points(r.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(len(r)), C.c_float(angle_min), C.c_float(angle_increment),
                                   C.c_float(time_increment), C.c_double(stamp), _pd(pts), _pd(ts))
    if m < 0:
        raise ValueError("angle_increment <= 0")
    return pts[:m].copy(), ts[:m].copy()


def laser_correct(points, times, stamp, linear, angular):
    """sensor::laser::correct (de-skew); returns the corrected copy"""
    p, t, lin, ang = _d(points).copy(), _d(times), _d(linear), _d(angular)
    _lib().liw_laser_correct(_pd(p), _pd(t), C.c_int(len(t)), C.c_double(stamp), _pd(lin), _pd(ang))
    return p


class Scan:
    def __init__(self, handle, lp_struct, owned=True):
        self.h, self._ps, self._owned = C.c_void_p(handle), lp_struct, owned

    @classmethod
    def spawn(cls, lp, points, time=0.0):
        ps = laser_params_struct(lp)
        p = _d(points).reshape(-1, 3)
        return cls(_lib().liw_scan_spawn(C.byref(ps), _pd(p), C.c_int(p.shape[0]), C.c_double(time)), ps)

    @classmethod
    def empty(cls, lp, time=0.0):
        ps = laser_params_struct(lp)
        return cls(_lib().liw_scan_create_empty(C.byref(ps), C.c_double(time)), ps)

    def add_segment(self, p1, p2, add_concers=False):
        a, b = _d(p1), _d(p2)
        return _lib().liw_scan_add_segment(self.h, _pd(a), _pd(b), C.c_int(int(add_concers)))

    def lines(self):
        n = _lib().liw_scan_num_lines(self.h)
        out = np.zeros((n, 10))
        if n:
            _lib().liw_scan_get_lines(self.h, _pd(out))
        return out

    def concers(self):
        n = _lib().liw_scan_num_concers(self.h)
        out = np.zeros((n, 3))
        if n:
            _lib().liw_scan_get_concers(self.h, _pd(out))
        return out

    def cell_lines(self, x, y, cap=16):
        ids = np.zeros(cap, dtype=np.int32)
        k = _lib().liw_scan_cell_lines(self.h, C.c_double(x), C.c_double(y), ids.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(cap))
        return k, ids[:max(0, min(k, cap))].copy()

    def __del__(self):
        try:
            if self._owned and self.h:
                _lib().liw_scan_destroy(self.h)
        except Exception:
            pass


class LaserMatch:
    def __init__(self, handle):
        L = _lib()
        h = C.c_void_p(handle)
        n = L.liw_laser_match_size(h)
        self.pts, self.pose = np.zeros((n, 12)), np.zeros(12)
        self.idx1, self.idx2 = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        ipt = C.POINTER(C.c_int)
        L.liw_laser_match_get(h, _pd(self.pts), _pd(self.pose), self.idx1.ctypes.data_as(ipt), self.idx2.ctypes.data_as(ipt))
        L.liw_laser_match_destroy(h)

    def __len__(self):
        return self.pts.shape[0]


def do_match(lp, scan1, scan2, p1, q1, p2, q2, kk=0):
    ps = laser_params_struct(lp)
    a, b, c, d = _d(p1), _d(q1), _d(p2), _d(q2)
    return LaserMatch(_lib().liw_laser_do_match(C.byref(ps), scan1.h, scan2.h, _pd(a), _pd(b), _pd(c), _pd(d), C.c_int(kk)))


class LaserManager:
    def __init__(self, lp):
        self._ps = laser_params_struct(lp)
        self.h = C.c_void_p(_lib().liw_laser_manager_create(C.byref(self._ps)))

    def add_scan(self, scan, p, q):
        a, b = _d(p), _d(q)
        _lib().liw_laser_manager_add_scan(self.h, scan.h, _pd(a), _pd(b))

    def _match(self, fn, scan, p, q):
        a, b = _d(p), _d(q)
        return LaserMatch(getattr(_lib(), fn)(self.h, scan.h, _pd(a), _pd(b)))

    def match_with_front(self, scan, p, q):
        return self._match("liw_laser_manager_match_with_front", scan, p, q)

    def match_with_back(self, scan, p, q):
        return self._match("liw_laser_manager_match_with_back", scan, p, q)

    def match_with_ref(self, scan, p, q):
        return self._match("liw_laser_manager_match_with_ref", scan, p, q)

    def pop_scan(self):
        return _lib().liw_laser_manager_pop_scan(self.h)

    def clear_all_scan(self):
        _lib().liw_laser_manager_clear_all_scan(self.h)

    def num_keyframes(self):
        return _lib().liw_laser_manager_num_keyframes(self.h)

    def ref_scan(self):
        p, q = np.zeros(3), np.zeros(3)
        h = _lib().liw_laser_manager_ref_scan(self.h, _pd(p), _pd(q))
        if not h:
            return None
        return Scan(h, self._ps, owned=False), p, q

    def __del__(self):
        try:
            if self.h:
                _lib().liw_laser_manager_destroy(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------- synthetic scans (tests / examples)
def room_segments(seed=0):
    """A closed room with a few inner walls: list of 2D segments (a, b) in the world frame."""
    rng = np.random.default_rng(seed)
    W, H = 9.0 + rng.uniform(-1, 1), 7.0 + rng.uniform(-1, 1)
    segs = [((-W / 2, -H / 2), (W / 2, -H / 2)), ((W / 2, -H / 2), (W / 2, H / 2)), ((W / 2, H / 2), (-W / 2, H / 2)), ((-W / 2, H / 2), (-W / 2, -H / 2))]
    for _ in range(4):
        c = rng.uniform([-W / 3, -H / 3], [W / 3, H / 3])
        ang, ln = rng.uniform(0, np.pi), rng.uniform(0.6, 2.0)
        d = 0.5 * ln * np.array([np.cos(ang), np.sin(ang)])
        segs.append((tuple(c - d), tuple(c + d)))
    return [(np.array(a, dtype=np.float64), np.array(b, dtype=np.float64)) for a, b in segs]


def cast_scan(segs, T_w_l, n_rays=1080, fov=2 * np.pi * 0.75, noise=0.004, seed=0, max_range=30.0):
    """Ranges of a planar lidar at pose T_w_l (4x4, laser in world) against the segments: float32 ranges + angle_min, increment."""
    rng = np.random.default_rng(seed)
    o = T_w_l[:2, 3]
    yaw = np.arctan2(T_w_l[1, 0], T_w_l[0, 0])
    angle_min, inc = -fov / 2, fov / (n_rays - 1)
    ranges = np.full(n_rays, np.inf, dtype=np.float32)
    for i in range(n_rays):
        a = yaw + angle_min + i * inc
        d = np.array([np.cos(a), np.sin(a)])
        best = np.inf
        for p, q in segs:
            e = q - p
            den = d[0] * e[1] - d[1] * e[0]
            if abs(den) < 1e-12:
                continue
            w = p - o
            t = (w[0] * e[1] - w[1] * e[0]) / den
            u = (w[0] * d[1] - w[1] * d[0]) / den
            if t > 0.05 and 0.0 <= u <= 1.0 and t < best:
                best = t
        if best < max_range:
            ranges[i] = np.float32(best + rng.normal(0.0, noise))
    return ranges, np.float32(angle_min), np.float32(inc)
