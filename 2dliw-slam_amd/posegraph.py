"""Back-end pose-graph relinearisation over the C ABI of include/liw_posegraph.h: `posegraph_solve`
mirrors keyframe_manager::solve (reference src/trajectory/keyframe_manager.cpp:722-838), `dense_spd_solve` exposes the
blocked MFMA Cholesky underneath; `make_pose_graph` builds synthetic key-frame graphs with odometry drift and loop edges."""
import ctypes as C

import numpy as np

PG_EXPORTS = ["liw_posegraph_solve", "liw_posegraph_linearize", "liw_dense_spd_solve"]


class PgParamsC(C.Structure):
    _fields_ = [("loop_sigma_p", C.c_double * 3), ("loop_sigma_q", C.c_double * 3), ("loop_edge_k", C.c_double),
                ("use_ground_p_factor", C.c_int), ("use_ground_q_factor", C.c_int)]


def office_pg_params():
    """config/office.yaml:106-115"""
    return dict(loop_sigma_p=[0.1, 0.1, 0.1], loop_sigma_q=[0.01, 0.01, 0.01], loop_edge_k=10.0, use_ground_p_factor=True, use_ground_q_factor=True)


def pg_struct(pg, cls=PgParamsC):
    s = cls()
    s.loop_sigma_p[:] = [float(v) for v in pg["loop_sigma_p"]]
    s.loop_sigma_q[:] = [float(v) for v in pg["loop_sigma_q"]]
    s.loop_edge_k = float(pg["loop_edge_k"])
    s.use_ground_p_factor = int(bool(pg["use_ground_p_factor"]))
    s.use_ground_q_factor = int(bool(pg["use_ground_q_factor"]))
    return s


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class PoseGraph:
    """Holds a liw_ctx (device, extrinsics, ground sigmas come from `prm`)."""

    def __init__(self, prm, device=0):
        from . import lib, params_struct, LiwError
        self.L, self.LiwError = lib(), LiwError
        self._ps = params_struct(prm, device)
        self.h = C.c_void_p(self.L.liw_create(C.byref(self._ps)))

    def _chk(self, r):
        if r < 0:
            raise self.LiwError(r, self.L.liw_last_error(self.h).decode())

    def solve(self, pg, poses, seq_idx, seq_tf12, loop_idx=None, loop_tf12=None, max_iters=0):
        from . import SummaryC
        x = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6).copy()
        si, st = np.ascontiguousarray(seq_idx, dtype=np.int32).reshape(-1, 2), np.ascontiguousarray(seq_tf12, dtype=np.float64).reshape(-1, 12)
        nl = 0 if loop_idx is None else len(loop_idx)
        li = np.ascontiguousarray(loop_idx if nl else np.zeros((1, 2)), dtype=np.int32).reshape(-1, 2)
        lt = np.ascontiguousarray(loop_tf12 if nl else np.zeros((1, 12)), dtype=np.float64).reshape(-1, 12)
        ps, sm = pg_struct(pg), SummaryC()
        self._chk(self.L.liw_posegraph_solve(self.h, C.byref(ps), C.c_int(x.shape[0]), _pd(x), C.c_int(si.shape[0]), _pi(si), _pd(st), C.c_int(nl),
                                             _pi(li), _pd(lt), C.c_int(max_iters), C.byref(sm)))
        return x, dict(iterations=sm.iterations, successful=sm.successful_steps, termination=sm.termination, initial_cost=sm.initial_cost,
                       final_cost=sm.final_cost)

    def linearize(self, pg, poses, seq_idx, seq_tf12, loop_idx=None, loop_tf12=None):
        x = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 6)
        si, st = np.ascontiguousarray(seq_idx, dtype=np.int32).reshape(-1, 2), np.ascontiguousarray(seq_tf12, dtype=np.float64).reshape(-1, 12)
        nl = 0 if loop_idx is None else len(loop_idx)
        li = np.ascontiguousarray(loop_idx if nl else np.zeros((1, 2)), dtype=np.int32).reshape(-1, 2)
        lt = np.ascontiguousarray(loop_tf12 if nl else np.zeros((1, 12)), dtype=np.float64).reshape(-1, 12)
        n = 6 * x.shape[0]
        H, g, cost = np.zeros((n, n)), np.zeros(n), C.c_double(0)
        ps = pg_struct(pg)
        self._chk(self.L.liw_posegraph_linearize(self.h, C.byref(ps), C.c_int(x.shape[0]), _pd(x), C.c_int(si.shape[0]), _pi(si), _pd(st), C.c_int(nl),
                                                 _pi(li), _pd(lt), _pd(H), _pd(g), C.byref(cost)))
        return H, g, cost.value

    def dense_spd_solve(self, A, b):
        A, b = np.ascontiguousarray(A, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        self._chk(self.L.liw_dense_spd_solve(self.h, C.c_int(b.shape[0]), _pd(A), _pd(b), _pd(x)))
        return x

    def __del__(self):
        try:
            if self.h:
                self.L.liw_destroy(self.h)
        except Exception:
            pass


def make_pose_graph(prm, N=60, seed=0, n_loop=5, laps=1.15, odo_noise=(0.01, 0.004), loop_noise=(0.002, 0.0008)):
    """Key frames along synth._Truth's circle (a bit more than one lap, so the end overlaps the start): truth poses, drifting
    initial poses from noisy odometry edges, loop edges between the overlapping ends.  tf12 = T_1^-1 T_2 of the IMU poses."""
    from . import synth
    rng = np.random.default_rng(seed)
    tr = synth._Truth(prm)
    T_lap = 2 * np.pi / tr.w
    ts = np.linspace(0.0, laps * T_lap, N)
    Tt = [tr.T_w_i(t) for t in ts]

    def noisy(T, sp, sq):
        D = np.eye(4)
        D[:3, :3] = synth.exp_so3(rng.normal(0, sq, 3))
        D[:3, 3] = rng.normal(0, sp, 3)
        return T @ D

    def rec12(T):
        return np.concatenate([T[:3, :3].reshape(9), T[:3, 3]])
    seq_idx, seq_tf, est = [], [], [Tt[0].copy()]
    for i in range(N - 1):
        rel = noisy(synth.inv_se3(Tt[i]) @ Tt[i + 1], *odo_noise)
        seq_idx.append((i, i + 1)); seq_tf.append(rec12(rel))
        est.append(est[-1] @ rel)
    loop_idx, loop_tf = [], []
    first_overlap = int(np.searchsorted(ts, T_lap))
    for k in range(n_loop):
        j = min(N - 1, first_overlap + k)            # late key frame
        i = int(np.argmin([np.linalg.norm(Tt[m][:3, 3] - Tt[j][:3, 3]) for m in range(first_overlap // 2)]))   # early one nearby
        rel = noisy(synth.inv_se3(Tt[j]) @ Tt[i], *loop_noise)
        loop_idx.append((j, i)); loop_tf.append(rec12(rel))
    to6 = lambda T: np.concatenate([T[:3, 3], synth.log_so3(T[:3, :3])])
    return dict(N=N, truth=np.array([to6(T) for T in Tt]), poses=np.array([to6(T) for T in est]), seq_idx=np.array(seq_idx, dtype=np.int32),
                seq_tf12=np.array(seq_tf), loop_idx=np.array(loop_idx, dtype=np.int32).reshape(-1, 2), loop_tf12=np.array(loop_tf).reshape(-1, 12))
