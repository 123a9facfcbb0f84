"""Batched / multi-GPU host side of the estimator: device-resident windows (torch tensors as plain HBM
allocations), the LM launch sequence, and factor sharding with an RCCL all-reduce of the laser partial sums.

torch is plumbing here (device memory, streams, torch.distributed); every kernel is in libliw_window.so.
"""
import ctypes as C

import numpy as np


XFN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)   # liw_exchange_fn (include/liw_window.h)


def shard_laser(win, rank, world):
    """Factor-parallel partition of ONE window: rank r keeps a contiguous slice of the (frame-sorted) laser blocks;
    states, IMU / wheel blocks and poses are replicated.  Union over ranks = all blocks, slices are disjoint."""
    L = int(np.asarray(win["laser_frame"]).shape[0])
    lo, hi = (L * rank) // world, (L * (rank + 1)) // world
    out = dict(win)
    out["laser_frame"] = np.asarray(win["laser_frame"])[lo:hi].copy()
    out["laser_pts"] = np.asarray(win["laser_pts"])[lo:hi].copy()
    return out


def allreduce_sum_(t, group=None):
    """Sum-all-reduce a tensor in place over the process group (RCCL on GPUs, gloo on CPU); no-op single process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allgather_(out, t, group=None):
    """One-shot exchange: every rank's image of `t` lands in out[rank] on every rank (no reduction on the wire; the caller
    sums the images in rank order, which is deterministic and identical on all ranks)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if t.is_cuda and dist.get_backend(group) == "gloo":
            # gloo has no device all-gather (single-GPU testing aid of bench.py / the tests): stage through the host
            import torch
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host.view(-1), t.cpu(), group=group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out.view(-1), t, group=group)
    else:
        out.view(-1)[:t.numel()].copy_(t)
    return out


EXCHANGES = ("allreduce", "oneshot", "auto", "p2p")


class TorchComm:
    """The production transport: torch.distributed over the given process group (RCCL on GPUs)."""

    def __init__(self, group=None):
        self.group = group

    def all_reduce_sum_(self, t):
        return allreduce_sum_(t, self.group)

    def all_gather_(self, out, t):
        return allgather_(out, t, self.group)

    def all_reduce_max_(self, t):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t


class LockstepComm:
    """Testing aid for 1-GPU boxes: `world` rank objects of ONE process, each driven by its own Python thread, exchange through
    shared slots and a barrier.  All ranks enqueue on the same HIP stream, so stream order makes every rank's image complete
    before any rank's sum runs.  One instance per rank: LockstepComm.make(world) -> [comm_0, .., comm_{world-1}]."""

    def __init__(self, shared, rank):
        self.sh, self.rank = shared, rank

    @staticmethod
    def make(world):
        import threading
        shared = {"slots": [None] * world, "bar": threading.Barrier(world), "world": world}
        return [LockstepComm(shared, r) for r in range(world)]

    def _deposit(self, t):
        self.sh["slots"][self.rank] = t
        self.sh["bar"].wait()

    def all_reduce_sum_(self, t):
        self._deposit(t)
        tot = self.sh["slots"][0].clone()
        for r in range(1, self.sh["world"]):
            tot += self.sh["slots"][r]
        self.sh["bar"].wait()          # every rank has enqueued its read of the images before any image is overwritten
        t.copy_(tot)
        self.sh["bar"].wait()
        return t

    def all_gather_(self, out, t):
        self._deposit(t)
        for r in range(self.sh["world"]):
            out[r].copy_(self.sh["slots"][r])
        self.sh["bar"].wait()
        return out

    def all_reduce_max_(self, t):
        self._deposit(t)
        tot = self.sh["slots"][0].clone()
        for r in range(1, self.sh["world"]):
            tot = self.torch_max(tot, self.sh["slots"][r])
        self.sh["bar"].wait()
        t.copy_(tot)
        self.sh["bar"].wait()
        return t

    @staticmethod
    def torch_max(a, b):
        import torch
        return torch.maximum(a, b)


def host_arrays(windows):
    """concatenated host image of `windows` (the caller-side arrays of include/liw_window.h's liw_batch)"""
    B = len(windows)
    Ls = [int(np.asarray(w["laser_frame"]).shape[0]) for w in windows]
    Ltot = int(sum(Ls))
    off = np.zeros(B + 1, dtype=np.int32)
    off[1:] = np.cumsum(Ls)
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(w[k], dtype=dt).reshape(-1) for w in windows]))
    pts = np.concatenate([np.asarray(w["laser_pts"], dtype=np.float64).reshape(-1, 12) for w in windows], axis=0) if Ltot else np.zeros((1, 12))
    return dict(
        x=cat("states", np.float64), laser_off=off, laser_frame=cat("laser_frame", np.int32) if Ltot else np.zeros(1, np.int32),
        laser_pts=np.ascontiguousarray(pts.T).reshape(-1),   # component-major [12][Ltot]
        match_pose=cat("match_pose", np.float64), has_match=cat("has_match", np.uint8),
        imu_X=cat("imu_X", np.float64), imu_J=cat("imu_J", np.float64), imu_sqrtP=cat("imu_sqrtP", np.float64), imu_Dt=cat("imu_Dt", np.float64),
        wheel_T=cat("wheel_T", np.float64), wheel_sqrtP=cat("wheel_sqrtP", np.float64),
        prior_X=np.zeros(B * 15), prior_J=np.zeros(B * 225), prior_R=np.zeros(B * 15), has_prior=np.zeros(B, dtype=np.int32), _Ltot=Ltot)


def tiled_tensors(base, tile, device):
    """the batch of tile["B"] windows cycling through `base`, built in HBM from the uploaded distinct windows"""
    import torch
    B, nb, n = int(tile["B"]), len(base), int(base[0]["n"])
    dev = torch.device(device)
    assert B >= 1 and nb >= 1
    reps, rem = divmod(B, nb)
    host = host_arrays(base)
    Lb = int(host.pop("_Ltot"))
    Lrem = int(host["laser_off"][rem])
    Ltot = reps * Lb + Lrem
    assert Ltot < 2 ** 31, "laser_off is int32"
    d = {k: torch.from_numpy(a if a.size else np.zeros(1, dtype=a.dtype)).to(dev) for k, a in host.items()}

    t_ = {}

    def cyc(t, per):   # per-window records of `per` elements: base pattern `reps` times, then its first `rem` windows
        if per == 0 or t.numel() < nb * per:           # (empty role: n == 1 has no IMU / wheel blocks — a one-element placeholder)
            return t.clone()
        return torch.cat([t.repeat(reps), t[:rem * per]]) if rem else t.repeat(reps)
    for k, per in (("has_match", n), ("imu_X", (n - 1) * 15), ("imu_J", (n - 1) * 225), ("imu_sqrtP", (n - 1) * 225), ("imu_Dt", n - 1),
                   ("wheel_T", (n - 1) * 12), ("wheel_sqrtP", (n - 1) * 9)):
        t_[k] = cyc(d[k], per)
    if Lb:
        pb = d["laser_pts"].view(12, Lb)                     # component-major planes of the distinct windows
        t_["laser_pts"] = (torch.cat([pb.repeat(1, reps), pb[:, :Lrem]], dim=1) if Lrem else pb.repeat(1, reps)).contiguous().view(-1)
        fb = d["laser_frame"]
        t_["laser_frame"] = torch.cat([fb.repeat(reps), fb[:Lrem]]) if Lrem else fb.repeat(reps)
    else:
        t_["laser_pts"], t_["laser_frame"] = d["laser_pts"], d["laser_frame"]
    ob = d["laser_off"][:nb].to(torch.int64)
    off = (torch.arange(reps, device=dev, dtype=torch.int64)[:, None] * Lb + ob[None, :]).reshape(-1)
    tail = reps * Lb + d["laser_off"][:rem + 1].to(torch.int64)
    t_["laser_off"] = torch.cat([off, tail]).to(torch.int32)
    assert t_["laser_off"].numel() == B + 1
    st = np.ascontiguousarray(np.asarray(tile["states"], dtype=np.float64).reshape(-1))
    mp = np.ascontiguousarray(np.asarray(tile["match_pose"], dtype=np.float64).reshape(-1))
    assert st.size == B * n * 15 and mp.size == B * n * 12
    t_["x"] = torch.from_numpy(st).to(dev)
    t_["match_pose"] = torch.from_numpy(mp).to(dev)
    z = lambda m, dt=torch.float64: torch.zeros(m, dtype=dt, device=dev)
    t_.update(prior_X=z(B * 15), prior_J=z(B * 225), prior_R=z(B * 15), has_prior=z(B, torch.int32))
    return t_, B, Ltot


class BatchSolver:
    """B independent windows (uniform n) resident on one GPU.

    windows: list of window dicts (synth.make_window layout).  With `world > 1` each window's laser blocks are
    sharded across ranks (`shard_laser`), the small factors are evaluated on every rank, and `solve` all-reduces
    the laser partial sums after every linearisation (one exchange per LM iteration).

    tile = dict(B=, states=[B, n, 15], match_pose=[B, n, 12]) (host arrays): `windows` are the DISTINCT windows of a batch of B that
    cycles through them (window b = windows[b % len(windows)] with its own states / laser_match poses).  Only the distinct windows and
    the two small per-window arrays cross PCIe; the batch is laid out in HBM by device-side repeats, so the host never holds the
    B-fold concatenation (25 GB per rank at bench.py's 49 152 C2 windows: eight ranks of that did not fit a node's RAM)."""

    def __init__(self, prm, windows, device="cuda:0", history_records=0, rank=0, world=1, group=None, exchange="allreduce", force_exchange=False, comm=None, tile=None):
        import torch
        from . import BatchC, WsLayoutC, lib, params_struct, LiwError
        self.torch = torch
        self.LiwError = LiwError
        self.L = lib()
        self.dev = torch.device(device)
        dev_index = self.dev.index if self.dev.index is not None else 0
        self._ps = params_struct(prm, dev_index)
        self.h = C.c_void_p(self.L.liw_create(C.byref(self._ps)))
        self.rank, self.world, self.group = rank, world, group
        assert exchange in EXCHANGES, exchange
        # exchange: how the ranks of a factor-sharded run combine their laser partial sums:
        #   "allreduce": RCCL all-reduce(SUM) of the compact record; "oneshot": all-gather of every rank's compact record
        #   (each GPU pushes its image to all peers once) + local sum in rank order; "auto": both are timed once on the record size
        #   of this batch (pick_exchange, at the first sharded solve) and the faster one is used — SURVEY 8e's "pick per size", decided
        #   by measurement on the machine at hand and agreed between the ranks; "p2p": the native one-shot exchange (every rank writes
        #   its record straight into every peer's receive area, flag-synchronised kernels, no collective: include/liw_window.h) — the areas
        #   have to be attached first (p2p_attach_local for rank objects of one process, p2p_attach_ipc across processes).
        #   force_exchange runs the pack / exchange / unpack path even with one rank (tests).
        self.exchange, self.sharded = exchange, (world > 1 or force_exchange)
        self.comm = comm if comm is not None else TorchComm(group)
        self.exchange_ms, self.exchange_calls, self._xev, self._xbuf = 0.0, 0, [], {}
        self.time_exchange = False
        if world > 1:
            windows = [shard_laser(w, rank, world) for w in windows]
        self.n = int(windows[0]["n"])
        n = self.n
        for w in windows:
            assert int(w["n"]) == n, "uniform n per batch"
        if tile is None:
            self.B = len(windows)
            host = host_arrays(windows)
            self.Ltot = int(host.pop("_Ltot"))
            self.t = {}
            for k, a in host.items():
                if a.size == 0:
                    a = np.zeros(1, dtype=a.dtype)
                self.t[k] = torch.from_numpy(a).to(self.dev)
        else:
            self.t, self.B, self.Ltot = tiled_tensors(windows, tile, self.dev)
        B = self.B
        self.history_records = int(history_records)
        lay = WsLayoutC()
        r = self.L.liw_batch_ws_layout(C.c_int(B), C.c_int(n), C.c_int(self.history_records), C.byref(lay))
        if r < 0:
            raise LiwError(r, "liw_batch_ws_layout")
        self.lay = lay
        self.ws = torch.zeros(int(lay.bytes), dtype=torch.uint8, device=self.dev)
        b = BatchC()
        b.B, b.n, b.Ltot = B, n, self.Ltot
        for k in ("x", "laser_off", "laser_frame", "laser_pts", "match_pose", "has_match", "imu_X", "imu_J", "imu_sqrtP", "imu_Dt",
                  "wheel_T", "wheel_sqrtP", "prior_X", "prior_J", "prior_R", "has_prior"):
            setattr(b, k, self.t[k].data_ptr())
        b.eval_small = 1
        b.history_records = self.history_records
        self.b = b
        # views of the laser partial sums (what a factor-sharded run all-reduces)
        nd = int(lay.laser_partial_bytes) // 8
        self.PL = [self.ws[int(lay.laser_partial_off[k]):int(lay.laser_partial_off[k]) + nd * 8].view(torch.float64) for k in range(2)]

    INPUT_KEYS = ("x", "laser_off", "laser_frame", "laser_pts", "match_pose", "has_match", "imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP")

    def rebind(self, t, Ltot):
        """Point the batch at ANOTHER set of caller arrays of the same B and n (the next frame of B robots that track in lock-step:
        trajectory.cpp:525-560 calls solve + marginalization once per laser frame on a new 2-frame window).  The solver's persistent
        linearised block (prior_X / J / R, has_prior: solver.h:31-37) and the workspace stay."""
        for k in self.INPUT_KEYS:
            self.t[k] = t[k]
            setattr(self.b, k, t[k].data_ptr())
        self.Ltot = int(Ltot)
        self.b.Ltot = int(Ltot)

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
            raise self.LiwError(r, self.L.liw_last_error(self.h).decode())
        return r

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def _wsp(self):
        return C.c_void_p(self.ws.data_ptr())

    def launch_paths(self):
        """-> dict(large_batch_format, lane_per_group_laser, packed_rows, blocks, padding_ratio) of the solve opened last on this batch"""
        flags, rows, blocks = C.c_int(0), C.c_longlong(0), C.c_longlong(0)
        self._chk(self.L.liw_batch_launch_paths(self.h, C.byref(self.b), self._wsp(), C.byref(flags)))
        self._chk(self.L.liw_batch_packed_rows(self.h, C.byref(self.b), self._wsp(), C.byref(rows), C.byref(blocks)))
        return dict(flags=flags.value, large_batch_format=bool(flags.value & 1), lane_per_group_laser=bool(flags.value & 2), packed_rows=rows.value,
                    blocks=blocks.value, padding_ratio=(64.0 * rows.value / blocks.value) if (rows.value and blocks.value) else None)

    # ---- LM pieces
    def solve(self, mode, max_iters=0, use_graph=False):
        """Runs the whole LM loop.  Single rank: one native call (optionally a captured hipGraph).  Factor-sharded:
        the loop is driven here so the all-reduce sits between linearise and step."""
        if not self.sharded:
            if use_graph:
                # stream capture is not allowed on the legacy default stream: replay on a dedicated side stream
                torch = self.torch
                if getattr(self, "_side", None) is None:
                    self._side = torch.cuda.Stream(device=self.dev)
                cur = torch.cuda.current_stream(self.dev)
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self._chk(self.L.liw_batch_solve(self.h, C.byref(self.b), C.c_int(mode), C.c_int(max_iters), self._wsp(), self._stream(), C.c_int(1)))
                cur.wait_stream(self._side)
                return
            self._chk(self.L.liw_batch_solve(self.h, C.byref(self.b), C.c_int(mode), C.c_int(max_iters), self._wsp(), self._stream(), C.c_int(0)))
            return
        # factor-sharded: the loop lives in the library (liw_batch_solve_sharded, shared with C++ hosts: tests/cpp/sharded_driver.cpp); this
        # side supplies the collective as a callback.  Per LM iteration: step -> linearise (laser role on this stream, small roles on the
        # ctx's side streams) -> exchange of the compact laser record on this stream, overlapping the small roles -> join.  Every rank
        # holds bit-identical sums, so states and `done` flags stay identical across ranks; the number of windows still iterating rides
        # in the exchanged buffer (identical everywhere by construction) and is read back between growing chunks for the early exit.
        buf, allb, nd = self._xbuffers(mode)
        if self.exchange == "p2p":
            assert getattr(self, "_p2p", None) is not None and self._p2p["mode"] == mode, "attach the receive areas first (p2p_attach_local / p2p_attach_ipc)"
            self._chk(self.L.liw_batch_exchange_timing(self.h, C.c_int(1 if self.time_exchange else 0), None, None))
            self._chk(self.L.liw_batch_solve_sharded(self.h, C.byref(self.b), C.c_int(mode), C.c_int(max_iters), self._wsp(), self._stream(),
                                                     C.c_void_p(buf.data_ptr()), None, C.c_int(max(self.world, 1)), XFN(0), None))
            self._last_x = buf
            return
        oneshot = self.exchange == "oneshot"
        err = []

        def _cb(user, pbuf, pall, doubles, stream):
            try:
                if oneshot:
                    self.comm.all_gather_(allb, buf)
                    return max(self.world, 1)
                self.comm.all_reduce_sum_(buf)
                return 1
            except BaseException as e:   # never unwind through the C frames
                err.append(e)
                return -1
        cb = XFN(_cb)
        self._chk(self.L.liw_batch_exchange_timing(self.h, C.c_int(1 if self.time_exchange else 0), None, None))
        r = self.L.liw_batch_solve_sharded(self.h, C.byref(self.b), C.c_int(mode), C.c_int(max_iters), self._wsp(), self._stream(),
                                           C.c_void_p(buf.data_ptr()), C.c_void_p(allb.data_ptr()) if allb is not None else None,
                                           C.c_int(max(self.world, 1)), cb, None)
        if err:
            raise err[0]
        self._chk(r)
        self._last_x = allb[0] if oneshot else buf

    def pick_exchange(self, mode, reps=3):
        """exchange="auto": time `reps` all-reduces and `reps` all-gather + rank-order sums of this batch's compact record (collective +
        unpack, HIP events), take the maximum over the ranks of each (one small all-reduce: every rank sees the same two numbers)
        and keep the faster transport.  Returns {"allreduce": ms, "oneshot": ms, "picked": name}; the pick is kept in self.exchange /
        self.exchange_pick for the life of this object.  NOT run-to-run reproducible: a ring all-reduce and the rank-order sum of gathered
        images differ in the low-order bits, so a sharded solve is bit-identical across ranks but, with "auto", not necessarily across
        runs or machines — pass exchange="allreduce" / "oneshot" when that matters."""
        t = self.torch
        reps = max(1, int(reps))
        nd = int(self.L.liw_batch_exchange_doubles(C.c_int(self.B), C.c_int(self.n), C.c_int(mode)))
        buf = t.zeros(nd, dtype=t.float64, device=self.dev)
        allb = t.zeros((max(self.world, 1), nd), dtype=t.float64, device=self.dev)
        ms = {}
        for name in ("allreduce", "oneshot"):
            for rep in range(reps + 1):                    # first pass untimed (lazy channel set-up of the transport)
                if rep == 1:
                    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                    e0.record()
                if name == "allreduce":
                    self.comm.all_reduce_sum_(buf)
                else:
                    self.comm.all_gather_(allb, buf)
                    buf.copy_(allb.sum(dim=0))             # stands in for the P-term sum of liw_batch_exchange_unpack
            e1.record()
            t.cuda.synchronize(self.dev)
            ms[name] = e0.elapsed_time(e1) / reps
            buf.zero_()
        both = t.tensor([ms["allreduce"], ms["oneshot"]], dtype=t.float64, device=self.dev)
        self.comm.all_reduce_max_(both)
        ms = {"allreduce": float(both[0].item()), "oneshot": float(both[1].item())}
        ms["picked"] = "oneshot" if ms["oneshot"] < ms["allreduce"] else "allreduce"
        self.exchange = ms["picked"]
        self._xbuf = {}
        self.exchange_pick = ms
        return ms

    def _xbuffers(self, mode):
        if self.exchange == "auto":
            self.pick_exchange(mode)
        nd = int(self.L.liw_batch_exchange_doubles(C.c_int(self.B), C.c_int(self.n), C.c_int(mode)))
        if (mode, "buf") not in self._xbuf:
            t = self.torch
            self._xbuf[(mode, "buf")] = t.zeros(nd, dtype=t.float64, device=self.dev)
            self._xbuf[(mode, "all")] = t.zeros((max(self.world, 1), nd), dtype=t.float64, device=self.dev) if self.exchange == "oneshot" else None
        return self._xbuf[(mode, "buf")], self._xbuf[(mode, "all")], nd

    def exchange_bytes(self, mode):
        """bytes one rank contributes to one exchange (what crosses xGMI per rank: x 2(P-1)/P for a ring all-reduce, x (P-1) pushes
        for the one-shot variant)"""
        return 8 * int(self.L.liw_batch_exchange_doubles(C.c_int(self.B), C.c_int(self.n), C.c_int(mode)))

    def _exchange(self, mode, candidate):
        """pack -> all-reduce | all-gather -> unpack (sum in rank order) of the laser partial sums of buffer `candidate`"""
        buf, allb, nd = self._xbuffers(mode)
        s = self._stream()
        ev = None
        if self.time_exchange:
            ev = (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self._chk(self.L.liw_batch_exchange_pack(self.h, C.byref(self.b), C.c_int(mode), C.c_int(candidate), self._wsp(), C.c_void_p(buf.data_ptr()), s))
        if self.exchange == "oneshot":
            self.comm.all_gather_(allb, buf)
            self._chk(self.L.liw_batch_exchange_unpack(self.h, C.byref(self.b), C.c_int(mode), C.c_int(candidate), self._wsp(),
                                                       C.c_void_p(allb.data_ptr()), C.c_int(max(self.world, 1)), s))
            self._last_x = allb[0]
        else:
            self.comm.all_reduce_sum_(buf)
            self._chk(self.L.liw_batch_exchange_unpack(self.h, C.byref(self.b), C.c_int(mode), C.c_int(candidate), self._wsp(),
                                                       C.c_void_p(buf.data_ptr()), C.c_int(1), s))
            self._last_x = buf
        if ev:
            ev[1].record()
            self._xev.append(ev)

    def _lin_exchange(self, mode, candidate):
        self._chk(self.L.liw_batch_lm_linearize_async(self.h, C.byref(self.b), C.c_int(mode), C.c_int(candidate), self._wsp(), self._stream()))
        self._exchange(mode, candidate)
        self._chk(self.L.liw_batch_lm_join(self.h, self._stream()))

    def active_windows(self, mode):
        """windows still iterating at the last exchange (trailer of the exchanged buffer; blocking 8-byte read-back)"""
        buf, allb, nd = self._xbuffers(mode)
        if self.exchange == "oneshot":
            return int(round(float(allb[:, nd - 1].sum().item()))) // max(self.world, 1)
        return int(round(float(buf[nd - 1].item()))) // max(self.world, 1)

    # ---- native peer-write exchange (liw_batch_p2p_setup): receive areas
    def _p2p_alloc(self, mode):
        t = self.torch
        L = self.L
        L.liw_batch_p2p_area_doubles.restype = C.c_size_t
        nd = int(L.liw_batch_p2p_area_doubles(C.c_int(self.B), C.c_int(self.n), C.c_int(mode), C.c_int(max(self.world, 1))))
        area = t.zeros(nd, dtype=t.float64, device=self.dev)
        flags = t.zeros(max(self.world, 1), dtype=t.int64, device=self.dev)
        return area, flags

    def _p2p_set(self, mode, area_ptrs, flag_ptrs, keep):
        w = max(self.world, 1)
        A = (C.c_void_p * w)(*area_ptrs)
        F = (C.c_void_p * w)(*flag_ptrs)
        self._chk(self.L.liw_batch_p2p_setup(self.h, C.c_int(self.rank), C.c_int(w), A, F))
        self._p2p = {"mode": mode, "keep": keep}

    @staticmethod
    def p2p_attach_local(ranks, mode):
        """rank objects of ONE process (tests on a 1-GPU box; also `world` = 1): every rank's receive area is a torch allocation, all see all"""
        bufs = [rk._p2p_alloc(mode) for rk in ranks]
        for rk in ranks:
            rk.torch.cuda.synchronize(rk.dev)
            rk._p2p_set(mode, [a.data_ptr() for a, _ in bufs], [f.data_ptr() for _, f in bufs], bufs)

    def p2p_attach_ipc(self, mode):
        """one process per GPU: receive areas from hipMalloc, handles exchanged over torch.distributed, peers' areas mapped with hipIpc.
        Exercised by two PROCESSES sharing the test box's one GPU (tests/test_gpu_p2p_ipc.py; hipIpcOpenMemHandle only refuses a handle
        of its own process); across xGMI it has not run."""
        import torch.distributed as dist
        hip = C.CDLL("libamdhip64.so")

        class Handle(C.Structure):
            _fields_ = [("reserved", C.c_byte * 64)]
        L = self.L
        L.liw_batch_p2p_area_doubles.restype = C.c_size_t
        w = max(self.world, 1)
        nd = int(L.liw_batch_p2p_area_doubles(C.c_int(self.B), C.c_int(self.n), C.c_int(mode), C.c_int(w)))
        own = []
        for nbytes in (8 * nd, 8 * w):
            p = C.c_void_p()
            assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
            assert hip.hipMemset(p, 0, C.c_size_t(nbytes)) == 0
            own.append(p)
        hs = []
        for p in own:
            h = Handle()
            assert hip.hipIpcGetMemHandle(C.byref(h), p) == 0
            hs.append(bytes(h.reserved))
        allh = [None] * w
        dist.all_gather_object(allh, hs, group=self.group)
        areas, flags = [], []
        for r in range(w):
            if r == self.rank:
                areas.append(own[0].value); flags.append(own[1].value)
                continue
            ptrs = []
            for raw in allh[r]:
                h = Handle()
                C.memmove(h.reserved, raw, 64)
                q = C.c_void_p()
                assert hip.hipIpcOpenMemHandle(C.byref(q), h, C.c_uint(1)) == 0      # hipIpcMemLazyEnablePeerAccess
                ptrs.append(q.value)
            areas.append(ptrs[0]); flags.append(ptrs[1])
        dist.barrier(group=self.group)
        self._p2p_ptrs = (areas, flags)
        self._p2p_set(mode, areas, flags, own)

    def exchange_timing(self):
        """average device time (ms) of one exchange (pack + collective + unpack) since the last call, and the count"""
        self.torch.cuda.synchronize(self.dev)
        ms = [a.elapsed_time(b) for a, b in self._xev]
        self._xev = []
        avg, cnt = C.c_double(0.0), C.c_int(0)
        self._chk(self.L.liw_batch_exchange_timing(self.h, C.c_int(1 if self.time_exchange else 0), C.byref(avg), C.byref(cnt)))
        tot, num = sum(ms) + avg.value * cnt.value, len(ms) + cnt.value
        return (tot / num if num else 0.0), num

    # the launch pieces of one solve (what liw_batch_solve chains); a factor-sharded driver puts its exchange of
    # self.PL[candidate] between lm_linearize and lm_step
    def lm_begin(self, mode, max_iters=0):
        K = self._chk(self.L.liw_batch_set_max_iters(self.h, C.c_int(mode), C.c_int(max_iters)))
        self._chk(self.L.liw_batch_lm_begin(self.h, C.byref(self.b), C.c_int(mode), C.c_int(K), self._wsp(), self._stream()))
        return K

    def lm_linearize(self, mode, candidate):
        self._chk(self.L.liw_batch_lm_linearize(self.h, C.byref(self.b), C.c_int(mode), C.c_int(candidate), self._wsp(), self._stream()))

    def lm_step(self, mode):
        self._chk(self.L.liw_batch_lm_step(self.h, C.byref(self.b), C.c_int(mode), self._wsp(), self._stream()))

    def lm_finish(self, mode):
        self._chk(self.L.liw_batch_lm_finish(self.h, C.byref(self.b), C.c_int(mode), self._wsp(), self._stream()))

    def time_kernels(self, mode, reps=3):
        """liw_batch_time_kernels: ms of every kernel of one LM iteration launched alone over the whole batch (+ the marginalisation's)"""
        out = (C.c_double * 6)()
        self._chk(self.L.liw_batch_time_kernels(self.h, C.byref(self.b), C.c_int(mode), self._wsp(), self._stream(), C.c_int(reps), out))
        self.torch.cuda.synchronize(self.dev)
        return dict(zip(("k_lin_laser", "k_lin_imu", "k_lin_small", "k_lm_step", "k_marg_schur", "k_lin_laser_marg"), [float(v) for v in out]))

    def linearize(self, mode):
        self._chk(self.L.liw_batch_linearize(self.h, C.byref(self.b), C.c_int(mode), self._wsp(), self._stream()))
        if self.sharded:
            self._exchange_plain(mode)

    def _exchange_plain(self, mode):
        # stand-alone linearisations carry no LM state: exchange the full record region (rare path: tests, liw_linearize)
        self.comm.all_reduce_sum_(self.PL[0])

    def export_dense(self, mode):
        N = 15 * self.n
        torch = self.torch
        H = torch.zeros((self.B, N, N), dtype=torch.float64, device=self.dev)
        g = torch.zeros((self.B, N), dtype=torch.float64, device=self.dev)
        c = torch.zeros(self.B, dtype=torch.float64, device=self.dev)
        self._chk(self.L.liw_batch_export_dense(self.h, C.byref(self.b), C.c_int(mode), C.c_int(0), self._wsp(), C.c_void_p(H.data_ptr()),
                                                C.c_void_p(g.data_ptr()), C.c_void_p(c.data_ptr()), self._stream()))
        return H, g, c

    def marginalize(self):
        torch = self.torch
        sH = torch.zeros((self.B, 36), dtype=torch.float64, device=self.dev)
        dH = torch.zeros((self.B, 225), dtype=torch.float64, device=self.dev)
        dg = torch.zeros((self.B, 15), dtype=torch.float64, device=self.dev)
        s = self._stream()
        self._chk(self.L.liw_batch_marg_linearize(self.h, C.byref(self.b), self._wsp(), s))
        if self.sharded:
            from . import LIW_MODE_MARG
            self._exchange(LIW_MODE_MARG, 0)
        self._chk(self.L.liw_batch_marg_schur(self.h, C.byref(self.b), self._wsp(), C.c_void_p(sH.data_ptr()), C.c_void_p(dH.data_ptr()),
                                              C.c_void_p(dg.data_ptr()), s))
        return sH, dH, dg

    # ---- results
    def states(self):
        return self.t["x"].cpu().numpy().reshape(self.B, self.n, 15)

    def set_states(self, x):
        self.t["x"].copy_(self.torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).reshape(-1)).to(self.dev))

    def summaries(self):
        from . import SummaryC
        raw = self.ws[int(self.lay.info_off):int(self.lay.info_off) + C.sizeof(SummaryC) * self.B].cpu().numpy().tobytes()
        arr = (SummaryC * self.B).from_buffer_copy(raw)
        return [dict(iterations=a.iterations, successful=a.successful_steps, termination=a.termination,
                     initial_cost=a.initial_cost, final_cost=a.final_cost) for a in arr]

    def history(self):
        if not self.history_records:
            return None
        nd = self.history_records * self.B * self.n * 15
        o = int(self.lay.history_off)
        return self.ws[o:o + nd * 8].view(self.torch.float64).cpu().numpy().reshape(self.history_records, self.B, self.n, 15)

    def set_timing(self, on):
        self._chk(self.L.liw_set_timing(self.h, C.c_int(int(on))))

    def get_timing(self):
        a, b, c, d = C.c_double(0), C.c_int(0), C.c_double(0), C.c_int(0)
        self._chk(self.L.liw_get_timing(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(linearize_ms=a.value, linearize_launches=b.value, step_ms=c.value, step_launches=d.value)


class BatchPreint:
    """Batched pre-integration of M independent frame-to-frame intervals on the GPU (`liw_batch_imu_preint`,
    `liw_batch_wheel_preint`): the batch-replay form of imu_preintegraption / wheel_odom_preintegration.
    Interval tuples are the ones `HostPreint` takes one at a time: (samples, t_start, t_end[, bias6])."""

    def __init__(self, prm, device="cuda:0"):
        import torch
        from . import lib, params_struct, LiwError
        self.torch, self.LiwError, self.L = torch, LiwError, lib()
        self.dev = torch.device(device)
        self._ps = params_struct(prm, self.dev.index if self.dev.index is not None else 0)
        self.h = C.c_void_p(self.L.liw_create(C.byref(self._ps)))

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
            raise self.LiwError(r, self.L.liw_last_error(self.h).decode())

    def _pack(self, intervals, width):
        torch = self.torch
        M = len(intervals)
        cnt = [int(np.asarray(iv[0]).shape[0]) for iv in intervals]
        off = np.zeros(M + 1, dtype=np.int32)
        off[1:] = np.cumsum(cnt)
        smp = np.concatenate([np.asarray(iv[0], dtype=np.float64).reshape(-1, width) for iv in intervals], axis=0) if M else np.zeros((1, width))
        ts = np.array([iv[1] for iv in intervals], dtype=np.float64)
        te = np.array([iv[2] for iv in intervals], dtype=np.float64)
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        return M, d(off), d(smp.reshape(-1)), d(ts), d(te)

    def imu(self, intervals):
        """-> X [M,15], J [M,15,15], sqrt_inverse_P [M,15,15], Dt [M] (device tensors)"""
        torch = self.torch
        M, off, smp, ts, te = self._pack(intervals, 7)
        bias = torch.from_numpy(np.ascontiguousarray(np.array([iv[3] for iv in intervals], dtype=np.float64).reshape(-1))).to(self.dev)
        z = lambda *sh: torch.zeros(sh, dtype=torch.float64, device=self.dev)
        X, J, P, S, Dt = z(M, 15), z(M, 15, 15), z(M, 15, 15), z(M, 15, 15), z(M)
        p = lambda t: C.c_void_p(t.data_ptr())
        self._chk(self.L.liw_batch_imu_preint(self.h, C.c_int(M), p(off), p(smp), p(ts), p(te), p(bias), p(X), p(J), p(P), p(S), p(Dt),
                                              C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        self.last_P = P
        return X, J, S, Dt

    def wheel(self, intervals):
        """-> delta_Tij [M,12], sqrt_inverse_P [M,3,3], Dt [M] (device tensors)"""
        torch = self.torch
        M, off, smp, ts, te = self._pack(intervals, 13)
        z = lambda *sh: torch.zeros(sh, dtype=torch.float64, device=self.dev)
        T, S, Dt = z(M, 12), z(M, 3, 3), z(M)
        p = lambda t: C.c_void_p(t.data_ptr())
        self._chk(self.L.liw_batch_wheel_preint(self.h, C.c_int(M), p(off), p(smp), p(ts), p(te), p(T), p(S), p(Dt),
                                                C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        return T, S, Dt
