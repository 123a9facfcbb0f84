"""2dliw-slam_amd — MI355X-native sliding-window estimator for 2D lidar–inertial–wheel SLAM.

Python host layer over the C-ABI shared library (include/liw_window.h): ctypes bindings plus thin mirrors of
the reference interfaces the path sits behind —
  * `Solver`            lvio_2d::solver {init_solve, solve, marginalization}   (reference src/factor/solver.h:28-79)
  * `ImuPreintegration` imu_preintegraption      (reference src/factor/imu_preintegraption.h:105-208)
  * `WheelPreintegration` wheel_odom_preintegration (reference src/factor/wheel_odom_preintegration.h:44-152)
  * `BatchSolver`       many independent windows resident in HBM (throughput path, torch device buffers)
The package name is not a Python identifier; import it with importlib.import_module("2dliw-slam_amd").
There is no CPU fallback: compute calls raise LiwError(LIW_ENODEV) without a gfx950 device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libliw_window.so")

LIW_MODE_INIT, LIW_MODE_TRACK, LIW_MODE_MARG = 0, 1, 2
LIW_ENODEV = -19
LASER_PARTIAL = 128

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
up = C.POINTER(C.c_ubyte)


class LiwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("liw error %d: %s" % (code, msg))
        self.code = code


class ParamsC(C.Structure):
    _fields_ = [("T_imu_to_wheel", C.c_double * 16), ("T_imu_to_laser", C.c_double * 16), ("g", C.c_double),
                ("line_to_line_sigma", C.c_double), ("manifold_p_sigma", C.c_double), ("manifold_q_sigma", C.c_double),
                ("imu_noise_acc_sigma", C.c_double * 3), ("imu_bias_acc_sigma", C.c_double * 3),
                ("imu_noise_gyro_sigma", C.c_double * 3), ("imu_bias_gyro_sigma", C.c_double * 3),
                ("wheel_sigma", C.c_double * 3), ("fast_mode", C.c_int), ("normalize_extrinsics", C.c_int),
                ("device", C.c_int)]


class WindowC(C.Structure):
    _fields_ = [("n", C.c_int), ("L", C.c_int), ("states", dp), ("laser_frame", ip), ("laser_pts", dp),
                ("match_pose", dp), ("has_match", up), ("imu_X", dp), ("imu_J", dp), ("imu_sqrtP", dp), ("imu_Dt", dp),
                ("wheel_T", dp), ("wheel_sqrtP", dp), ("wheel_Dt", dp)]


class SummaryC(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful_steps", C.c_int), ("termination", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class BatchC(C.Structure):
    _fields_ = [("B", C.c_int), ("n", C.c_int), ("Ltot", C.c_int), ("x", C.c_void_p), ("laser_off", C.c_void_p),
                ("laser_frame", C.c_void_p), ("laser_pts", C.c_void_p), ("match_pose", C.c_void_p),
                ("has_match", C.c_void_p), ("imu_X", C.c_void_p), ("imu_J", C.c_void_p), ("imu_sqrtP", C.c_void_p),
                ("imu_Dt", C.c_void_p), ("wheel_T", C.c_void_p), ("wheel_sqrtP", C.c_void_p), ("prior_X", C.c_void_p),
                ("prior_J", C.c_void_p), ("prior_R", C.c_void_p), ("has_prior", C.c_void_p), ("eval_small", C.c_int),
                ("history_records", C.c_int)]


class WsLayoutC(C.Structure):
    _fields_ = [("bytes", C.c_size_t), ("laser_partial_off", C.c_size_t * 2), ("laser_partial_bytes", C.c_size_t),
                ("info_off", C.c_size_t), ("history_off", C.c_size_t)]


# every symbol include/liw_window.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "liw_create", "liw_destroy", "liw_last_error", "liw_get_extrinsics", "liw_set_window", "liw_clear_window", "liw_solve", "liw_get_history",
    "liw_linearize", "liw_eval_factors", "liw_marginalize", "liw_get_prior", "liw_set_prior", "liw_batch_ws_layout",
    "liw_batch_set_max_iters", "liw_batch_launch_paths", "liw_batch_packed_rows", "liw_batch_linearize", "liw_batch_time_kernels", "liw_batch_lm_begin", "liw_batch_lm_linearize", "liw_batch_lm_step",
    "liw_batch_lm_finish", "liw_batch_lm_linearize_async", "liw_batch_lm_join", "liw_batch_exchange_doubles", "liw_batch_exchange_pack",
    "liw_batch_exchange_unpack", "liw_batch_solve_sharded", "liw_batch_exchange_timing", "liw_batch_p2p_area_doubles", "liw_batch_p2p_setup", "liw_batch_p2p_status", "liw_batch_solve", "liw_batch_marg_linearize", "liw_batch_marg_schur", "liw_batch_export_dense",
    "liw_set_timing", "liw_get_timing", "liw_batch_imu_preint", "liw_batch_wheel_preint", "liw_imu_preint_create", "liw_imu_preint_destroy", "liw_imu_preint_reset",
    "liw_imu_preint_add", "liw_imu_preint_update_only_t", "liw_imu_preint_Dt", "liw_imu_preint_result",
    "liw_wheel_preint_create", "liw_wheel_preint_destroy", "liw_wheel_preint_reset", "liw_wheel_preint_add",
    "liw_wheel_preint_update_only_t", "liw_wheel_preint_result",
]

_LIB = None


def lib():
    """Load libliw_window.so (built in-tree by build.py / __graft_entry__.build()).  Fails loudly if missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LiwError(-2, "libliw_window.so is not built (run python 2dliw-slam_amd/build.py); there is no CPU fallback")
        # one HIP runtime per process: PyTorch-ROCm ships its own libamdhip64, and whichever copy is loaded second cannot
        # initialise the device ("No HIP GPUs are available").  Import torch first when it is installed, so that the
        # library's HIP calls and torch's device memory / streams share torch's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.liw_create.restype = C.c_void_p
        L.liw_create.argtypes = [C.POINTER(ParamsC)]
        L.liw_last_error.restype = C.c_char_p
        L.liw_last_error.argtypes = [C.c_void_p]
        L.liw_destroy.argtypes = [C.c_void_p]
        L.liw_imu_preint_create.restype = C.c_void_p
        L.liw_imu_preint_create.argtypes = [C.POINTER(ParamsC)]
        L.liw_wheel_preint_create.restype = C.c_void_p
        L.liw_wheel_preint_create.argtypes = [C.POINTER(ParamsC)]
        L.liw_imu_preint_Dt.restype = C.c_double
        for name in ("liw_imu_preint_destroy", "liw_wheel_preint_destroy"):
            getattr(L, name).argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def params_struct(prm, device=0):
    s = ParamsC()
    s.T_imu_to_wheel[:] = [float(v) for v in np.asarray(prm["T_imu_to_wheel"], dtype=np.float64).reshape(16)]
    s.T_imu_to_laser[:] = [float(v) for v in np.asarray(prm["T_imu_to_laser"], dtype=np.float64).reshape(16)]
    for k in ("g", "line_to_line_sigma", "manifold_p_sigma", "manifold_q_sigma"):
        setattr(s, k, float(prm[k]))
    for k in ("imu_noise_acc_sigma", "imu_bias_acc_sigma", "imu_noise_gyro_sigma", "imu_bias_gyro_sigma", "wheel_sigma"):
        getattr(s, k)[:] = [float(v) for v in prm[k]]
    s.fast_mode = int(bool(prm.get("fast_mode", False)))
    s.normalize_extrinsics = int(bool(prm.get("normalize_extrinsics", True)))
    s.device = int(device)
    return s


def _p(a):
    return a.ctypes.data_as(dp)


WINDOW_FIELDS = ("states", "laser_frame", "laser_pts", "match_pose", "has_match", "imu_X", "imu_J", "imu_sqrtP", "imu_Dt",
                 "wheel_T", "wheel_sqrtP", "wheel_Dt")


class Window:
    """Contiguous host arrays of one flat window (`liw_window`)."""

    def __init__(self, d):
        self.n = int(d["n"])
        self.a = {}
        for k in WINDOW_FIELDS:
            dt = np.int32 if k == "laser_frame" else (np.uint8 if k == "has_match" else np.float64)
            arr = np.ascontiguousarray(np.array(d[k], dtype=dt, copy=True))
            if arr.size == 0:
                arr = np.zeros(12 if k == "laser_pts" else 1, dtype=dt)
            self.a[k] = arr
        self.L = int(np.asarray(d["laser_frame"]).shape[0])
        c = WindowC()
        c.n, c.L = self.n, self.L
        c.states = _p(self.a["states"]); c.laser_frame = self.a["laser_frame"].ctypes.data_as(ip)
        c.laser_pts = _p(self.a["laser_pts"]); c.match_pose = _p(self.a["match_pose"])
        c.has_match = self.a["has_match"].ctypes.data_as(up)
        for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
            setattr(c, k, _p(self.a[k]))
        self.c = c

    def __getitem__(self, k):
        return self.a[k]


class Solver:
    """Mirror of lvio_2d::solver on a flat window: init_solve / solve / marginalization mutate the window's
    `states` and `match_pose` in place and keep the linearised prior inside the context."""

    def __init__(self, prm, device=0):
        self.L = lib()
        self._ps = params_struct(prm, device)
        self.h = C.c_void_p(self.L.liw_create(C.byref(self._ps)))
        if not self.h:
            raise LiwError(-22, "liw_create failed")
        self.win = None

    def close(self):
        if self.h:
            self.L.liw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r < 0:
            raise LiwError(r, self.L.liw_last_error(self.h).decode())
        return r

    def extrinsics(self):
        a, b = np.zeros(16), np.zeros(16)
        self._chk(self.L.liw_get_extrinsics(self.h, _p(a), _p(b)))
        return a.reshape(4, 4), b.reshape(4, 4)

    def set_window(self, win):
        self.win = win
        self._chk(self.L.liw_set_window(self.h, C.byref(win.c)))

    def _solve(self, mode, max_iters):
        s = SummaryC()
        self._chk(self.L.liw_solve(self.h, C.c_int(mode), C.c_int(max_iters), C.byref(s)))
        return dict(iterations=s.iterations, successful=s.successful_steps, termination=s.termination,
                    initial_cost=s.initial_cost, final_cost=s.final_cost)

    def init_solve(self, max_iters=0):
        return self._solve(LIW_MODE_INIT, max_iters)

    def solve(self, max_iters=0):
        return self._solve(LIW_MODE_TRACK, max_iters)

    def history(self, max_records=128):
        x = np.zeros((max_records, self.win.n, 15))
        r = self._chk(self.L.liw_get_history(self.h, _p(x), C.c_int(max_records)))
        return x[:r]

    def linearize(self, mode):
        N = 15 * self.win.n
        H, g, c = np.zeros((N, N)), np.zeros(N), C.c_double(0)
        self._chk(self.L.liw_linearize(self.h, C.c_int(mode), _p(H), _p(g), C.byref(c)))
        return H, g, c.value

    def eval_factors(self, mode):
        n, L = self.win.n, self.win.L
        out = dict(laser_res=np.zeros((max(L, 1), 2)), laser_jac=np.zeros((max(L, 1), 2, 12)), imu_res=np.zeros((max(n - 1, 1), 15)),
                   imu_jac=np.zeros((max(n - 1, 1), 15, 30)), wheel_res=np.zeros((max(n - 1, 1), 3)), wheel_jac=np.zeros((max(n - 1, 1), 3, 12)),
                   ground_res=np.zeros((n, 2)), ground_jac=np.zeros((n, 2, 6)))
        self._chk(self.L.liw_eval_factors(self.h, C.c_int(mode), *[_p(out[k]) for k in
                                          ("laser_res", "laser_jac", "imu_res", "imu_jac", "wheel_res", "wheel_jac", "ground_res", "ground_jac")]))
        out["laser_res"], out["laser_jac"] = out["laser_res"][:L], out["laser_jac"][:L]
        out["imu_res"], out["imu_jac"] = out["imu_res"][:n - 1], out["imu_jac"][:n - 1]
        out["wheel_res"], out["wheel_jac"] = out["wheel_res"][:n - 1], out["wheel_jac"][:n - 1]
        return out

    def marginalization(self):
        out = np.zeros(36 + 225 + 15)      # one allocation: the call is ~10 us of a 0.24 ms tracking frame
        base = out.ctypes.data
        self._chk(self.L.liw_marginalize(self.h, C.cast(base, dp), C.cast(base + 36 * 8, dp), C.cast(base + 261 * 8, dp)))
        return dict(sqrt_H=out[:36].reshape(6, 6), Delta_H=out[36:261].reshape(15, 15), Delta_g=out[261:])

    def get_prior(self):
        X, J, R = np.zeros(15), np.zeros(225), np.zeros(15)
        has = self._chk(self.L.liw_get_prior(self.h, _p(X), _p(J), _p(R)))
        return (X, J.reshape(15, 15), R) if has else None

    def set_prior(self, prior):
        if prior is None:
            self._chk(self.L.liw_set_prior(self.h, C.c_int(0), None, None, None))
            return
        X, J, R = [np.ascontiguousarray(v, dtype=np.float64) for v in prior]
        self._chk(self.L.liw_set_prior(self.h, C.c_int(1), _p(X), _p(J.reshape(225)), _p(R)))


class ImuPreintegration:
    """imu_preintegraption mirror (host, sequential): reset_imu_measure / add_imu_measure / update_only_t /
    get_preintegraption_result."""

    def __init__(self, prm):
        self.L = lib()
        self._ps = params_struct(prm)
        self.h = C.c_void_p(self.L.liw_imu_preint_create(C.byref(self._ps)))

    def __del__(self):
        try:
            self.L.liw_imu_preint_destroy(self.h)
        except Exception:
            pass

    def reset_imu_measure(self, time, acc_bias, gyr_bias):
        a, g = np.ascontiguousarray(acc_bias, dtype=np.float64), np.ascontiguousarray(gyr_bias, dtype=np.float64)
        self.L.liw_imu_preint_reset(self.h, C.c_double(time), _p(a), _p(g))

    def add_imu_measure(self, t, acc, gyro):
        a, g = np.ascontiguousarray(acc, dtype=np.float64), np.ascontiguousarray(gyro, dtype=np.float64)
        return bool(self.L.liw_imu_preint_add(self.h, C.c_double(t), _p(a), _p(g)))

    def update_only_t(self, t):
        self.L.liw_imu_preint_update_only_t(self.h, C.c_double(t))

    @property
    def Dt(self):
        return self.L.liw_imu_preint_Dt(self.h)

    def get_preintegraption_result(self):
        X, J, P, Dt = np.zeros(15), np.zeros(225), np.zeros(225), C.c_double(0)
        self.L.liw_imu_preint_result(self.h, _p(X), _p(J), _p(P), C.byref(Dt))
        return X, J.reshape(15, 15), P.reshape(15, 15), Dt.value


class WheelPreintegration:
    """wheel_odom_preintegration mirror (host, sequential)."""

    def __init__(self, prm):
        self.L = lib()
        self._ps = params_struct(prm)
        self.h = C.c_void_p(self.L.liw_wheel_preint_create(C.byref(self._ps)))

    def __del__(self):
        try:
            self.L.liw_wheel_preint_destroy(self.h)
        except Exception:
            pass

    def reset_wheel_odom_measure(self, time):
        self.L.liw_wheel_preint_reset(self.h, C.c_double(time))

    def add_wheel_odom_measure(self, t, R9, t3):
        r, tt = np.ascontiguousarray(R9, dtype=np.float64).reshape(9), np.ascontiguousarray(t3, dtype=np.float64)
        return bool(self.L.liw_wheel_preint_add(self.h, C.c_double(t), _p(r), _p(tt)))

    def update_only_t(self, t):
        self.L.liw_wheel_preint_update_only_t(self.h, C.c_double(t))

    def get_preintegraption_result(self):
        T, P, Dt = np.zeros(12), np.zeros(9), C.c_double(0)
        self.L.liw_wheel_preint_result(self.h, _p(T), _p(P), C.byref(Dt))
        return T, P.reshape(3, 3), Dt.value


class HostPreint:
    """Replays sample arrays through the product's host pre-integrators (the `preint` provider synth.make_window
    expects): the seed sample fixes last_info, the accumulator is reset at t_start, as the reference's trajectory
    does at every laser frame (src/trajectory/trajectory.cpp:176-184)."""

    def __init__(self, prm):
        self.prm = prm

    def imu_preint(self, samples, t_start, t_end, bias6):
        p = ImuPreintegration(self.prm)
        for i, s in enumerate(np.asarray(samples)):
            p.add_imu_measure(s[0], s[1:4], s[4:7])
            if i == 0:
                p.reset_imu_measure(t_start, bias6[0:3], bias6[3:6])
        p.update_only_t(t_end)
        return p.get_preintegraption_result()

    def wheel_preint(self, samples, t_start, t_end):
        p = WheelPreintegration(self.prm)
        did = False
        for s in np.asarray(samples):
            if not did and s[0] > t_start:
                p.update_only_t(t_start)
                p.reset_wheel_odom_measure(t_start)
                did = True
            p.add_wheel_odom_measure(s[0], s[1:10], s[10:13])
        if not did:
            p.update_only_t(t_start)
            p.reset_wheel_odom_measure(t_start)
        p.update_only_t(t_end)
        return p.get_preintegraption_result()


from .batch import BatchPreint, BatchSolver, shard_laser  # noqa: E402,F401
from . import laser, outputs, posegraph  # noqa: E402,F401
