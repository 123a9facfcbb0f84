#!/usr/bin/env python
"""bench.py — sliding-window solves/second of the MI355X-native estimator (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input: for each of the B windows of the batch
(config C2: 30 key-frames, 2 000 laser_factor blocks, 29 IMU + 29 wheel pre-integrated blocks, 2*30^2 ground
blocks) one init-topology Levenberg–Marquardt solve to Ceres termination (cap 50 iterations) followed by one
marginalisation — i.e. what `lvio_2d::solver::init_solve` + `solver::marginalization` do for the reference's
trajectory (src/trajectory/trajectory.cpp:446,479).  Inputs are resident in HBM before the timed region.

N > 1: `python bench.py --gpus N` re-executes itself under torch.distributed.run (one rank per GPU, RCCL) when it is not already
running under a launcher; launched by torch.distributed.run directly it uses the ranks it was given (and fails loudly when
--gpus disagrees with WORLD_SIZE).  Windows are independent, so ranks hold disjoint
window batches (replicas, weak scaling, no data-path collective); the factor-sharded mode (laser blocks of each
window split across ranks, RCCL all-reduce of the laser partial sums per LM iteration) is measured separately on
the C4-shaped window and reported under "factor_sharded" at EVERY N (same total work: strong scaling; N = 1 is the
un-sharded reference point of that curve).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per residual block ("materialised-J" convention)
BYTES_LASER_BOTH, BYTES_LASER_ONE, BYTES_IMU, BYTES_WHEEL, BYTES_GROUND, BYTES_STATE = 312, 216, 7448, 520, 60, 120
PIS_BYTES, PWS_BYTES, LP_BYTES, PGS_BYTES, PIFS_BYTES = 496 * 8, 92 * 8, 128 * 8, 28 * 8, 376 * 8   # partial-sum records (csrc/liw_kernels.hpp; PIFS: per-frame IMU record of batches >= 1 024 windows)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes(n, L, both_free):
    return L * (BYTES_LASER_BOTH if both_free else BYTES_LASER_ONE) + (n - 1) * (BYTES_IMU + BYTES_WHEEL) + 2 * n * n * BYTES_GROUND + n * BYTES_STATE


def step_model(n, arrow=True):
    """HBM bytes and essential flops of ONE LM step of one window in k_lm_step_quad (csrc/k_lm_quad.hip), init topology.  Every read of
    the two sweeps is an LDS-DMA piece of 64 lanes (a wave = four windows), so the volume is known exactly.
    First sweep, per frame and wave: 12 pieces of 1 KiB for the four per-frame IMU records (3 008 B each), 4 for the laser group records
    (1 024 B each), 4 for the four wheel (736 B) + four ground (224 B) records, gathered; per window 3 x 120 B of Jacobi scale / LM diagonal /
    state entries.  Second sweep: 11 pieces for the four back-substitution records (2 640 B each), 6 four-byte pieces (256 B) for state / scale /
    diagonal.  Prologue: the compact cost array (eight 128-byte lines per window) and the current + candidate states.  Writes: the 22-column
    back-substitution record (2 640 B per frame, read again by the second sweep), LM diagonal, candidate states.
    Flops: Cholesky 15^3/3, 22 forward and 22 backward substitutions 2 x 22 x 15^2, Schur products (16x16 + 6x16 + 6x6/2) x 15 x 2, second
    sweep 21 x 15 x 2 — per frame."""
    sweep1 = n * ((12 + 4 + 4) * 1024 // 4 + 3 * 120)
    sweep2 = n * ((11 * 1024 + 6 * 256) // 4)
    rd = sweep1 + sweep2 + 8 * 128 + 2 * n * 120
    wr = n * 2640 + n * 120 + n * 120
    fl = n * (15 ** 3 / 3.0 + 2 * 22 * 15 * 15 + (16 * 16 + (6 * 16 + 18 if arrow else 0)) * 15 * 2 + 21 * 15 * 2)
    return {"read": int(rd), "write": int(wr), "flops": float(fl)}


def linearise_model(n, L):
    """HBM bytes ONE linearisation of one window moves through the role kernels of the large-batch format, from what each kernel
    addresses (csrc/k_laser_slab.hip, k_linearize.hip): an analytic model like step_model, checked against the PMC counters
    (profiles/pmc_traffic.json: 328 - 338 kB counted per window).
      laser role: the per-solve packed rows of a 2-D scan, 8 doubles per block (the z planes are skipped), the states of the window;
                  one 128-slot group record per frame written, its cost once more into the compact cost array;
      IMU role:   the per-solve packed block records (192 doubles), states; one per-frame record (376 doubles) per frame written;
      wheel / ground role: delta_Tij (12) + sqrt_inverse_P (9) per block, states; 92-double wheel records, 28-double ground records.
    The marginalisation's linearisation (one pose free per laser block) reads and writes the same records."""
    laser = {"read": 64 * L + 120 * n, "write": n * (LP_BYTES + 8)}
    imu = {"read": (n - 1) * 1536 + 120 * n, "write": n * PIFS_BYTES + 8 * (n - 1)}
    small = {"read": (n - 1) * 168 + 120 * n, "write": (n - 1) * (PWS_BYTES + 8) + n * (PGS_BYTES + 8)}
    tot = sum(r["read"] + r["write"] for r in (laser, imu, small))
    return {"laser": laser, "imu": imu, "small": small, "total": int(tot)}


def input_floor_bytes(n, L):
    """what an LM iteration must read at least when it reads the caller's factor records and the states (2 000 x 104 +
    29 x 3 728 + 29 x 208 + 30 x 120).  Since the end of round 3 the linearisation reads LESS than these arrays hold: the IMU role reads
    per-solve packed block records (1 536 of the 3 728 B: the entries the factor uses) and the laser role skips the z planes of 2-D scans
    (64 of 96 B per block) — input_floor_packed_bytes()."""
    return L * 104 + (n - 1) * (3728 + 208) + n * 120


def input_floor_packed_bytes(n, L):
    return L * 72 + (n - 1) * (1536 + 208) + n * 120


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def spawn_ranks(n_gpus):
    """--gpus N without a launcher: re-exec under torch.distributed.run, one rank per GPU, and pass rank 0's JSON line through."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_batch(liw, synth, prm, B, n, L, seed0, n_base=64):
    """B windows: n_base fully generated windows (distinct seeds: distinct LM paths, iteration counts and terminations), tiled with
    per-window state perturbations (so that every window's data is physically distinct in HBM)."""
    hp = liw.HostPreint(prm)
    base = [synth.make_window(hp, prm, seed=seed0 + k, n=n, L=L) for k in range(min(n_base, B))]
    rng = np.random.default_rng(seed0 + 1000)
    out = []
    for b in range(B):
        w = dict(base[b % len(base)])
        if b >= len(base):
            st = np.array(w["states"], copy=True)
            st[:, 0:3] += rng.normal(0.0, 2e-3, (n, 3))
            st[:, 6:9] += rng.normal(0.0, 2e-3, (n, 3))
            w["states"] = st
            mp = np.array(w["match_pose"], copy=True)
            mp[:, 0:6] = st[0, 0:6]
            mp[:, 6:12] = st[:, 0:6]
            w["match_pose"] = mp
        out.append(w)
    return out


class TiledWindows:
    """The batch make_batch() would build, WITHOUT building it: the distinct windows plus every window's own states / laser_match poses
    (B x n x 27 doubles); window b is put together on demand (parity gate, CPU legs).  BatchSolver(tile=...) lays the batch out in HBM
    from these with device-side repeats — the host of an 8-rank run holds 8 x ~0.4 GB instead of 8 x 25 GB."""

    def __init__(self, base, states, match_pose):
        self.base, self.states, self.match_pose = base, states, match_pose

    def __len__(self):
        return int(self.states.shape[0])

    def __getitem__(self, b):
        if isinstance(b, slice):
            return [self[i] for i in range(*b.indices(len(self)))]
        if b < 0:
            b += len(self)
        w = dict(self.base[b % len(self.base)])
        w["states"] = self.states[b].copy()
        w["match_pose"] = self.match_pose[b].copy()
        return w

    def tile(self):
        return dict(B=len(self), states=self.states, match_pose=self.match_pose)


def ragged_shapes(synth, n, L, nb, seed):
    """per-base-window laser block counts per frame of a RAGGED batch with the same total as nb even windows of L blocks: per-window L drawn
    from [L / 4, 2 L] and rescaled to the total, per-frame counts from synth.ragged_frame_counts (empty frames next to frames with hundreds)"""
    rng = np.random.default_rng(seed)
    Ls = rng.uniform(0.25 * L, 2.0 * L, nb)
    Ls = np.maximum(1, np.round(Ls * (L * nb / Ls.sum()))).astype(np.int64)
    Ls[-1] += L * nb - Ls.sum()
    return [synth.ragged_frame_counts(rng, n, int(Lk)) for Lk in Ls]


def make_tiled(liw, synth, prm, B, n, L, seed0, n_base=64, ragged=False):
    """the same windows as make_batch(...) (same seeds, same jitter stream), as a TiledWindows; ragged: the distinct windows get the laser
    block counts of ragged_shapes (same total) instead of L blocks spread evenly over frames 1 .. n-1"""
    hp = liw.HostPreint(prm)
    nb = min(n_base, B)
    if ragged:
        shapes = ragged_shapes(synth, n, L, nb, seed0 + 7)
        base = [synth.make_window(hp, prm, seed=seed0 + k, n=n, frame_counts=shapes[k]) for k in range(nb)]
    else:
        base = [synth.make_window(hp, prm, seed=seed0 + k, n=n, L=L) for k in range(nb)]
    rng = np.random.default_rng(seed0 + 1000)
    idx = np.arange(B) % nb
    st = np.stack([np.asarray(w["states"], dtype=np.float64).reshape(n, 15) for w in base])[idx]
    mp = np.stack([np.asarray(w["match_pose"], dtype=np.float64).reshape(n, 12) for w in base])[idx]
    if B > nb:
        jit = rng.normal(0.0, 2e-3, (B - nb, 2, n, 3))      # (window b: position jitter, then velocity jitter — make_batch's draw order)
        st[nb:, :, 0:3] += jit[:, 0]
        st[nb:, :, 6:9] += jit[:, 1]
        mp[nb:, :, 0:6] = st[nb:, 0:1, 0:6]
        mp[nb:, :, 6:12] = st[nb:, :, 0:6]
    return TiledWindows(base, st, mp)


def peak_rss_mb():
    import resource
    return round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)


def cpu_quota():
    """CPUs this process may use: the cgroup CPU quota (v2 cpu.max, v1 cfs_quota_us / cfs_period_us) and the affinity mask — a box
    can show 256 logical CPUs and grant 16"""
    q = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()[:2]
            q = None if a == "max" else float(a) / float(b)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                a, b = float(f.read()), float(g.read())
                q = None if a <= 0 else a / b
        except (OSError, ValueError):
            q = None
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    return {"cgroup_cpu_quota": round(q, 2) if q else None, "affinity_cpus": aff, "logical_cpus": os.cpu_count()}


def replay_leg(liw, synth, prm, path, seconds, keep, seed=11, teacher_forced=False):
    """BASELINE config C3 in one command: a flat sensor log (tools/replay_log's format; rosbag_reader.bag_to_flatlog
    converts an OpenLORIS bag) through the C++ driver on the GPU and through the oracle's twin on the host, same run: frames / s side by
    side, poses compared, the reference-shaped record table.  Default log: synthetic, corridor-rate sensors (IMU 200 Hz, odometry 20 Hz,
    LaserScan 10 Hz).  keep = 1 is the reference's window policy (trajectory.cpp:590-617)."""
    import struct
    import subprocess
    import tempfile
    from oracle import pyoracle
    replay = importlib.import_module("2dliw-slam_amd.replay")
    tmp = tempfile.mkdtemp(prefix="liw_replay_")
    if path:
        msgs = replay.read_log(path)
        src = "flat log %s" % os.path.basename(path)
    else:
        msgs, _ = replay.make_log(prm, duration=seconds, seed=seed)
        path = os.path.join(tmp, "log.bin")
        replay.write_log(path, msgs)
        src = "synthetic corridor-rate log, %.0f s, seed %d (2dliw-slam_amd/replay.py)" % (seconds, seed)
    exe = os.path.join(ROOT, "tools", "replay_log")
    libdir = os.path.dirname(liw.LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "replay_log.cpp"), "-o", exe,
                           "-L", libdir, "-lliw_window", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = tmp + "/"
    t0 = time.perf_counter()
    r = subprocess.run([exe, path, out, "--keep", str(keep)] if keep != 1 else [exe, path, out], capture_output=True)
    t_gpu = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": r.stderr.decode(errors="replace")[-300:]}
    raw = open(out + "result.bin", "rb").read()
    status, frames, tracked, inits, keyframes, sstat = struct.unpack("<6i", raw[:24])
    lp = liw.laser.office_laser_params(prm)
    orc = pyoracle.TrajectoryOracle(prm, lp, keep_window_size=keep)
    orc.set_capture(bool(teacher_forced))
    t0 = time.perf_counter()
    for m in msgs:
        if m["type"] == 0:
            orc.add_imu(m["time"], m["acc"], m["gyro"])
        elif m["type"] == 1:
            orc.add_wheel(m["time"], m["R"], m["t"])
        else:
            pts, ts = pyoracle.laser_to_points(m["ranges"], m["angle_min"], m["angle_increment"], m["time_increment"], m["time"])
            orc.add_laser(m["time"], pts, ts)
    t_cpu = time.perf_counter() - t0
    c = orc.counters()
    got = replay.read_tum(out + "fornt_end.txt")
    ref = np.array([ln.split() for ln in orc.tum().splitlines()[1:]], dtype=np.float64).reshape(-1, 8)
    m_ = min(len(got), len(ref))
    err = np.abs(got[:m_, 1:] - ref[:m_, 1:]).max(axis=1) / max(1.0, np.abs(ref[:m_, 1:]).max()) if m_ else np.zeros(0)
    beyond = np.nonzero(err > 1e-6)[0]
    table = {}
    try:
        for ln in open(out + "traj.md"):
            cells = [x.strip() for x in ln.strip().strip("|").split("|")]
            if len(cells) >= 5 and cells[0] in ("solve", "marginalization", "spawn_scan", "match_line", "init_solve"):
                table[cells[0]] = {"records": int(cells[1]), "max_us": float(cells[2]), "min_us": float(cells[3]), "aver_us": float(cells[4])}
    except (OSError, ValueError):
        pass
    scans = sum(1 for m in msgs if m["type"] not in (0, 1))
    tf = None
    if teacher_forced:
        # north_star's per-solve statement on THIS log: every tracking solve of the oracle twin's replay (the window + carried prior it
        # started from) re-run on the MI355X through lvio_2d::solver's C ABI; counted: solves whose states are within 1e-6 of the oracle's
        # with the same iteration count and termination (free-running trajectories part ways through the driver's own sensitivity, below)
        caps = orc.captures()
        slv = liw.Solver(prm)
        ok = same_it = 0
        worst, t_g = 0.0, 0.0
        for cp_ in caps:
            w = liw.Window(cp_)
            slv.set_prior((cp_["prior_X"], cp_["prior_J"].reshape(15, 15), cp_["prior_R"]) if cp_["has_prior"] else None)
            t0 = time.perf_counter()
            slv.set_window(w)
            sg = slv.solve()
            t_g += time.perf_counter() - t0
            e = float(np.abs(w["states"].reshape(-1) - cp_["states_after"]).max() / max(np.abs(cp_["states_after"]).max(), 1e-12))
            same = (sg["iterations"], sg["termination"]) == (cp_["iterations"], cp_["termination"])
            same_it += int(same)
            ok += int(same and e <= 1e-6)
            worst = max(worst, e)
        tf = {"solves": len(caps), "within_1e-6_with_equal_iterations_and_termination": ok, "equal_iterations_and_termination": same_it,
              "max_rel_state_err": float("%.3e" % worst), "largest_window_frames": int(max([q_["n"] for q_ in caps] or [0])),
              "largest_window_laser_blocks": int(max([q_["L"] for q_ in caps] or [0])), "gpu_ms_per_solve": round(1e3 * t_g / max(len(caps), 1), 3),
              "note": "each solve starts from the oracle's own input (window + prior), so this counts per-solve parity, not trajectory divergence"}
    return {"log": src, "teacher_forced_tracking_solves": tf, "messages": len(msgs), "laser_scans": scans, "keep_window_size": keep,
            "gpu": {"seconds": round(t_gpu, 3), "scans_per_s": round(scans / t_gpu, 1), "frames": frames, "tracked": tracked, "initializations": inits,
                    "note": "tools/replay_log: process start, log read, dispatch, pre-integration, laser front-end (host) and every solve / marginalisation (MI355X) included"},
            "cpu_oracle": {"seconds": round(t_cpu, 3), "scans_per_s": round(scans / t_cpu, 1), "frames": c["frames"], "tracked": c["tracked"], "initializations": c["initializations"],
                           "cores": 1},
            "state_machine_identical": bool((frames, tracked, inits) == (c["frames"], c["tracked"], c["initializations"])),
            "poses_compared": int(m_), "poses_within_1e-6_from_the_start": int(beyond[0]) if len(beyond) else int(m_),
            "max_rel_pose_err": float("%.3e" % (float(err.max()) if m_ else 0.0)),
            "max_position_difference_m": float("%.3e" % (float(np.abs(got[:m_, 1:4] - ref[:m_, 1:4]).max()) if m_ else 0.0)),
            "note": "free-running replays are chaotic beyond their first frames — the oracle against itself with 1e-13 input noise departs the same way "
                    "(tests/soak/sensitivity_replay.py, DESIGN 7); the 1e-6 statement on EVERY solve is teacher-forced in tests/test_gpu_replay.py",
            "record_table_us": table}


def _cpu_worker(job):
    """one host process = one oracle solving the same window `reps` times (window-parallel CPU throughput)"""
    prm, win, reps, iters = job
    from oracle import pyoracle
    orc = pyoracle.Oracle(prm)
    w = pyoracle.Window(win)
    sec, _ = orc.time_solves(w, reps, iters, dense_product=True)
    return sec


def track_frame_cpp(liw, d3, reps):
    """The steady-state tracking frame driven from C++ (tools/track_frame_cpp.cpp through include/lvio_2d_solver.hpp): builds the tool with
    g++ next to the library, hands it the three-frame window in the flat format of tests/test_cpp_host.py::dump_window."""
    import struct
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.abspath(__file__))
    libdir = os.path.dirname(liw.LIB_PATH)
    src = os.path.join(root, "tools", "track_frame_cpp.cpp")
    exe = os.path.join(libdir, "build", "track_frame_cpp")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    deps = [src, os.path.join(root, "include", "lvio_2d_solver.hpp"), os.path.join(root, "include", "liw_window.h")]
    if (not os.path.exists(exe)) or any(os.path.getmtime(q) > os.path.getmtime(exe) for q in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), src, "-o", exe,
                               "-L", libdir, "-lliw_window", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "w3.bin")
        n, L = int(d3["n"]), int(np.asarray(d3["laser_frame"]).shape[0])
        with open(path, "wb") as f:
            f.write(struct.pack("<ii", n, L))
            f.write(np.asarray(d3["states"], dtype=np.float64).tobytes())
            f.write(np.asarray(d3["laser_frame"], dtype=np.int32).tobytes())
            f.write(np.asarray(d3["laser_pts"], dtype=np.float64).tobytes())
            f.write(np.asarray(d3["match_pose"], dtype=np.float64).tobytes())
            f.write(np.asarray(d3["has_match"], dtype=np.uint8).tobytes())
            for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
                f.write(np.asarray(d3[k], dtype=np.float64).tobytes())
        r = subprocess.run([exe, path, str(reps)], capture_output=True, timeout=120)
    if r.returncode != 0:
        raise RuntimeError("track_frame_cpp rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-200:]))
    tok = r.stdout.decode().split()
    return {"ms_per_frame": float(tok[1]), "iterations": int(tok[3])}



def sub_window(d, lo, m=2):
    """frames lo .. lo + m - 1 of window `d` as an m-frame window: their states / laser_match poses / laser blocks, the IMU and wheel blocks
    between them — the window the reference's tracking holds after pop_frame_for_tracking (trajectory.cpp:590-617) for m = 2"""
    N = int(d["n"])
    o = dict(d)
    o["n"] = m
    for k in ("states", "match_pose", "truth_states"):
        o[k] = np.asarray(d[k]).reshape(N, -1)[lo:lo + m].copy()
    o["has_match"] = np.asarray(d["has_match"])[lo:lo + m].copy()
    for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
        o[k] = np.asarray(d[k])[lo:lo + m - 1].copy()
    lf = np.asarray(d["laser_frame"])
    msk = (lf >= lo) & (lf < lo + m)
    o["laser_frame"] = (lf[msk] - lo).astype(np.int32)
    o["laser_pts"] = np.asarray(d["laser_pts"])[msk].copy()
    return o


class TrackBatch:
    """The reference's STEADY STATE at scale: B robots tracking in lock-step.  Per laser frame and robot the front-end
    calls solver::solve on the 2-frame window (previous frame, new frame) — laser blocks of the new frame against the constant pose of their
    reference scan, one IMU and one wheel block, ground blocks, the prior of the last marginalisation on the older frame — and then
    solver::marginalization, whose result is the prior of the next frame (src/trajectory/trajectory.cpp:525-560, src/factor/solver.cpp:631-820,
    :257-442).  Here: K + 1 consecutive frames of nb distinct synthetic trajectories (60 - 70 matched line pairs per scan), tiled to B windows
    with 2 mm jitter on every new frame's initial guess; frame 0 only produces the first prior and is not timed.  The solved state and
    laser_match pose of frame k is the older frame of step k + 1 (device-to-device copies inside the timed region); inputs of every step are
    resident in HBM before the timed region starts."""

    def __init__(self, liw, synth, prm, B, K, nb, dev, seed0=60240, iters=0, blocks=(60, 71)):
        import torch
        self.liw, self.prm, self.B, self.K, self.dev, self.iters = liw, prm, int(B), int(K), dev, int(iters)
        self.torch = torch
        hp = liw.HostPreint(prm)
        nb = self.nb = min(int(nb), self.B)
        rng = np.random.default_rng(seed0 + 1)
        steps = self.steps = self.K + 1
        self.traj = [synth.make_window(hp, prm, seed=seed0 + k, n=steps + 1, frame_counts=np.concatenate([[0], rng.integers(blocks[0], blocks[1], steps)]))
                     for k in range(nb)]
        self.sub = [[sub_window(tr, k) for tr in self.traj] for k in range(steps)]
        idx = np.arange(self.B) % nb
        self.t, self.Ltot, self.x0, self.mp0 = [], [], [], []
        for k in range(steps):
            st = np.stack([np.asarray(w["states"], dtype=np.float64).reshape(2, 15) for w in self.sub[k]])[idx]
            mp = np.stack([np.asarray(w["match_pose"], dtype=np.float64).reshape(2, 12) for w in self.sub[k]])[idx]
            if self.B > nb:
                rows = (0, 1) if k == 0 else (1,)           # the older frame of every later step is the previous step's solved frame
                for r in rows:
                    st[nb:, r, 0:3] += rng.normal(0.0, 2e-3, (self.B - nb, 3))
                    st[nb:, r, 6:9] += rng.normal(0.0, 2e-3, (self.B - nb, 3))
                mp[nb:, :, 6:12] = st[nb:, :, 0:6]
            t, _, Ltot = liw.batch.tiled_tensors(self.sub[k], dict(B=self.B, states=st, match_pose=mp), dev)
            self.t.append(t)
            self.Ltot.append(Ltot)
            self.x0.append(t["x"].clone())
            self.mp0.append(t["match_pose"].clone())
        st0 = self.x0[0].cpu().numpy().reshape(self.B, 2, 15)
        mp0 = self.mp0[0].cpu().numpy().reshape(self.B, 2, 12)
        self.bs = liw.BatchSolver(prm, self.sub[0], device=dev, tile=dict(B=self.B, states=st0, match_pose=mp0))
        self.blocks_new = float(np.mean([(np.asarray(w["laser_frame"]) == 1).sum() for ws in self.sub[1:] for w in ws]))
        self.blocks_window = float(np.mean([len(w["laser_frame"]) for ws in self.sub[1:] for w in ws]))

    def window(self, k, b):
        return self.sub[k][b % self.nb]

    def run(self, capture_ids=None):
        """all K + 1 frames once -> (seconds of frames 1 .. K, per-frame LM iterations [K + 1][B] (None without capture), capture records)"""
        torch, bs, B, liw = self.torch, self.bs, self.B, self.liw
        bs.t["has_prior"].zero_()
        cap, its = [], []
        ids = None if capture_ids is None else torch.as_tensor(list(capture_ids), device=self.dev, dtype=torch.long)
        t0 = None
        for k in range(self.steps):
            if k == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            t = self.t[k]
            t["x"].copy_(self.x0[k])
            t["match_pose"].copy_(self.mp0[k])
            if k > 0:
                px, pm = self.t[k - 1]["x"].view(B, 2, 15), self.t[k - 1]["match_pose"].view(B, 2, 12)
                t["x"].view(B, 2, 15)[:, 0].copy_(px[:, 1])
                t["match_pose"].view(B, 2, 12)[:, 0, 6:12].copy_(pm[:, 1, 6:12])
            bs.rebind(t, self.Ltot[k])
            rec = None
            if ids is not None:
                g = lambda name, w: bs.t[name].view(B, *w)[ids].cpu().numpy().copy()
                rec = dict(x_in=g("x", (2, 15)), mp_in=g("match_pose", (2, 12)), pX_in=g("prior_X", (15,)), pJ_in=g("prior_J", (15, 15)), pR_in=g("prior_R", (15,)), has_in=g("has_prior", ()))
            bs.solve(liw.LIW_MODE_TRACK, self.iters)
            if ids is not None:
                rec.update(x_out=g("x", (2, 15)), mp_out=g("match_pose", (2, 12)))
                sm = bs.summaries()
                its.append(np.array([s_["iterations"] for s_ in sm]))
                rec["summ"] = [sm[int(b)] for b in capture_ids]
            sH, dH, dg = bs.marginalize()
            if ids is not None:
                rec.update(dH=dH.view(B, 15, 15)[ids].cpu().numpy(), dg=dg[ids].cpu().numpy(), sH=sH.view(B, 6, 6)[ids].cpu().numpy(),
                           pX_out=g("prior_X", (15,)), pJ_out=g("prior_J", (15, 15)), pR_out=g("prior_R", (15,)), has_out=g("has_prior", ()))
                cap.append(rec)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, (its if ids is not None else None), cap

    def teacher_forced_parity(self, capture_ids, cap):
        """every captured (window, frame >= 1): the oracle's solver::solve from the SAME inputs the GPU batch had (states, laser_match poses, the
        carried prior), then its marginalisation at the GPU's solved states -> counts and worst errors"""
        from oracle import pyoracle
        orc = pyoracle.Oracle(self.prm)
        orc.set_max_iterations(self.iters if self.iters > 0 else (10 if self.prm.get("fast_mode") else 50))
        rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
        out = dict(frames=0, within_1e_6=0, iterations_equal=0, terminations_equal=0, worst_rel_state=0.0, worst_rel_Delta_H=0.0, worst_Delta_g_of_roundoff_scale=0.0,
                   worst_rel_prior_JtJ=0.0, worst_prior_JtR_of_roundoff_scale=0.0)
        for k in range(1, self.steps):
            r = cap[k]
            for j, b in enumerate(capture_ids):
                w = dict(self.window(k, int(b)))
                w["states"], w["match_pose"] = r["x_in"][j], r["mp_in"][j]
                prior = (r["pX_in"][j], r["pJ_in"][j], r["pR_in"][j]) if r["has_in"][j] else None
                wo = pyoracle.Window(w)
                orc.set_prior(prior)
                orc.solve(wo)
                so = orc.summary()
                e = rel(r["x_out"][j], wo["states"].reshape(2, 15))
                out["frames"] += 1
                out["within_1e_6"] += int(e <= 1e-6)
                out["iterations_equal"] += int(r["summ"][j]["iterations"] == so["iterations"])
                out["terminations_equal"] += int(r["summ"][j]["termination"] == so["termination"])
                out["worst_rel_state"] = max(out["worst_rel_state"], e)
                if not self.prm.get("fast_mode"):
                    mo = marg_reference(pyoracle, orc, w, r["x_out"][j], r["mp_out"][j], 1, prior=prior)[0]
                    sc = float(np.abs(mo["dH"]).max())
                    pJ, pR = r["pJ_out"][j], r["pR_out"][j]
                    out["worst_rel_Delta_H"] = max(out["worst_rel_Delta_H"], rel(r["dH"][j], mo["dH"]))
                    out["worst_Delta_g_of_roundoff_scale"] = max(out["worst_Delta_g_of_roundoff_scale"], float(np.abs(r["dg"][j] - mo["dg"]).max() / mo["g_scale"]))
                    out["worst_rel_prior_JtJ"] = max(out["worst_rel_prior_JtJ"], float(np.abs(pJ.T @ pJ - mo["J"].T @ mo["J"]).max() / sc))
                    out["worst_prior_JtR_of_roundoff_scale"] = max(out["worst_prior_JtR_of_roundoff_scale"], float(np.abs(pJ.T @ pR - mo["J"].T @ mo["R"]).max() / mo["g_scale"]))
        for k_ in list(out):
            if isinstance(out[k_], float):
                out[k_] = float("%.3e" % out[k_])
        return out

    def cpu_oracle_frames_per_s(self, min_seconds=2.0):
        """trajectory 0 free-running through the oracle on one core: solve + marginalization per frame, the prior carried like the reference does"""
        from oracle import pyoracle
        orc = pyoracle.Oracle(self.prm)
        orc.set_max_iterations(self.iters if self.iters > 0 else (10 if self.prm.get("fast_mode") else 50))
        tot, frames, its = 0.0, 0, 0
        while tot < min_seconds:
            orc.set_prior(None)
            prev = None
            for k in range(self.steps):
                w = dict(self.window(k, 0))
                if prev is not None:
                    w["states"] = np.array(w["states"], copy=True)
                    w["match_pose"] = np.array(w["match_pose"], copy=True)
                    w["states"][0] = prev[0]
                    w["match_pose"][0, 6:12] = prev[1]
                wo = pyoracle.Window(w)
                t0 = time.perf_counter()
                orc.solve(wo)
                orc.marginalization(wo)
                dt = time.perf_counter() - t0
                if k > 0:
                    tot += dt
                    frames += 1
                    its += orc.summary()["iterations"]
                prev = (wo["states"].reshape(2, 15)[1].copy(), wo["match_pose"].reshape(2, 12)[1, 6:12].copy())
        return dict(frames_per_s=round(frames / tot, 1), ms_per_frame=round(1e3 * tot / frames, 4), cores=1, frames=frames, lm_iterations_mean=round(its / frames, 2),
                    sample="trajectory 0, %d frames free-running (solve + marginalization, prior carried), %.1f s" % (frames, tot))




def ragged_leg(liw, synth, prm, dev, B, n, L, iters, nb):
    """`ragged_c2`: B windows cycling through nb distinct RAGGED ones (ragged_shapes: the same total as nb x L blocks, per-window L from L / 4 to
    2 L, per-frame groups from 0 to several hundred) — init solve + marginalisation, one warm-up pass and one timed pass; which laser kernel
    ran and how much padding the packed rows carry; the same batch once more with the windows taken in batch order (LIW_SLAB_BATCH_ORDER=1:
    the round-5 layout) for the before / after of the per-frame order."""
    import torch
    tw = make_tiled(liw, synth, prm, B, n, L, seed0=40240, n_base=nb, ragged=True)
    counts = np.stack([np.bincount(np.asarray(w["laser_frame"]), minlength=n) for w in tw.base])
    out = {"windows": B, "distinct_windows": len(tw.base), "blocks_per_window_min_mean_max": [int(counts.sum(1).min()), float(counts.sum(1).mean()), int(counts.sum(1).max())],
           "blocks_per_frame_max": int(counts.max()), "empty_frames_pct": round(100.0 * float((counts[:, 1:] == 0).mean()), 1),
           "padding_ratio_in_batch_order": round(float(counts.max(0).sum() * len(tw.base) / counts.sum()), 3)}

    def run(env):
        for k, v in env.items():
            os.environ[k] = v
        try:
            bs = liw.BatchSolver(prm, tw.base, device=dev, tile=tw.tile())
            x0, mp0 = bs.t["x"].clone(), bs.t["match_pose"].clone()
            secs = []
            for rep in range(2):
                bs.t["x"].copy_(x0); bs.t["match_pose"].copy_(mp0); bs.t["has_prior"].zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                bs.solve(liw.LIW_MODE_INIT, iters)
                bs.marginalize()
                torch.cuda.synchronize()
                secs.append(time.perf_counter() - t0)
            lp = bs.launch_paths()
            kt = bs.time_kernels(liw.LIW_MODE_INIT, 3)
            sm = bs.summaries()
            r = {"solves_per_s": round(B / secs[-1], 1), "ms_per_step": round(1e3 * secs[-1], 2), "lane_per_group_laser_kernel": lp["lane_per_group_laser"],
                 "packed_rows_padding_ratio": round(lp["padding_ratio"], 3) if lp["padding_ratio"] else None,
                 "k_lin_laser_alone_ms": round(kt["k_lin_laser"], 4), "k_lin_laser_marg_alone_ms": round(kt["k_lin_laser_marg"], 4),
                 "lm_iterations_mean": round(float(np.mean([s_["iterations"] for s_ in sm])), 2)}
            bs.close()
            del bs
            torch.cuda.empty_cache()
            return r
        finally:
            for k in env:
                os.environ.pop(k, None)
    out["per_frame_order"] = run({})
    out["batch_order_round5_layout"] = run({"LIW_SLAB_BATCH_ORDER": "1"})
    out["note"] = ("same factors per window on average as C2; the lane-per-group laser kernel pads a (slab, frame) wave to its longest group: in batch order "
                   "that is padding_ratio_in_batch_order rows per row of data (above 4 the solve falls back to the lane-per-block kernel), in the per-frame order "
                   "by group length (round 6) packed_rows_padding_ratio")
    return out


def track_model(blocks_new, blocks_window):
    """HBM bytes per 2-frame window of the kernels of one batched tracking frame, from what each kernel addresses (the large-batch record
    format: linearise_model / step_model with n = 2).  TRACK linearisation: laser blocks of the NEW frame only (64 B per block of a 2-D scan:
    solver.cpp:669-698), one packed IMU record, one wheel block, two ground records.  The marginalisation's linearisation reads the blocks of
    both frames (solver.cpp:443-590)."""
    n = 2
    lin = {"read": 64 * blocks_new + 120 * n + 1536 + 120 * n + 168 + 120 * n, "write": n * (LP_BYTES + 8) + n * PIFS_BYTES + 8 + (PWS_BYTES + 8) + n * (PGS_BYTES + 8)}
    lin_marg = {"read": lin["read"] + 64 * (blocks_window - blocks_new), "write": lin["write"]}
    sm = step_model(n, arrow=False)
    pack = {"read": 3728 + 104 * blocks_window, "write": 1536}     # per solve: k_imu_pack reads the caller's IMU block (3 728 B), k_laser_z_scan the four z planes, k_group_offsets the frame ids
    return {"linearise": lin, "linearise_marg": lin_marg, "step": sm, "per_solve_packing": pack}


def track_batch_leg(liw, synth, prm, dev, B, K, nb, gate_windows, cpu=True):
    import ctypes as C
    import torch
    tb = TrackBatch(liw, synth, prm, B, K, nb, dev)
    ids = []
    for c_ in (0, B - 1, B // 2, min(nb, B) - 1, 63 % B, 64 % B, max(B - 65, 0), min(nb + 1, B - 1)):
        if c_ not in ids:
            ids.append(c_)
    ids = ids[:max(1, gate_windows)]
    _, its, cap = tb.run(capture_ids=ids)                       # warm-up pass with capture (the passes are bit-reproducible)
    flags = C.c_int(0)
    tb.bs.L.liw_batch_launch_paths(tb.bs.h, C.byref(tb.bs.b), tb.bs._wsp(), C.byref(flags))
    tb.bs.set_timing(True)
    secs = min(tb.run()[0] for _ in range(3))
    tm = tb.bs.get_timing()
    tb.bs.set_timing(False)
    kt = tb.bs.time_kernels(liw.LIW_MODE_TRACK, 3)
    itk = np.stack(its[1:])
    model = track_model(tb.blocks_new, tb.blocks_window)
    lin_b = model["linearise"]["read"] + model["linearise"]["write"]
    stp_b = model["step"]["read"] + model["step"]["write"]
    out = {"frames_per_s": round(B * K / secs, 1), "ms_per_frame_of_batch": round(1e3 * secs / K, 4), "robots": B, "frames_timed": K, "distinct_trajectories": tb.nb,
           "window": "n=2: %.1f laser blocks on the new frame (%.1f in the window: the marginalisation linearises both frames), 1 IMU + 1 wheel block, 8 ground blocks, prior on the older frame"
                     % (tb.blocks_new, tb.blocks_window),
           "per_frame": "x / laser_match carry (device copies) + solve(TRACK) + marginalization; reference call pattern trajectory.cpp:525-560, solver.cpp:631-820, :257-442",
           "lm_iterations_mean": round(float(itk.mean()), 3), "lm_iterations_histogram": {str(int(k)): int((itk == k).sum()) for k in np.unique(itk)},
           "launch_paths": {"flags": int(flags.value), "large_batch_record_format (k_lin_imu_chain_multi: 16 two-frame windows per wave, k_lm_step_quad, k_marg_schur_chain + k_marg_schur_eigq)": bool(flags.value & 1),
                            "lane_per_group_laser_kernel (k_lin_laser_slab1 on the solve's packed rows, also for the marginalisation)": bool(flags.value & 2)},
           "brackets": {"linearize_avg_ms": round(tm["linearize_ms"], 5), "linearize_launches": tm["linearize_launches"], "step_avg_ms": round(tm["step_ms"], 5), "step_launches": tm["step_launches"],
                        "share_of_frame_time": round((tm["linearize_ms"] * tm["linearize_launches"] + tm["step_ms"] * tm["step_launches"]) * 1e-3 / (3 * secs), 3),
                        "note": "HIP-event brackets of the LM linearisations / steps over three passes of the K frames; the rest of a frame is the per-solve packing, the marginalisation and the carry"},
           "kernel_times_alone_ms": {k: round(v, 4) for k, v in kt.items()},
           "model_bytes_per_window": {"linearise": lin_b, "step": stp_b, "linearise_marg": model["linearise_marg"]["read"] + model["linearise_marg"]["write"]}}
    # roofline of the dominant kernel, every window active (stand-alone kernel times).  `traffic`: FETCH_SIZE x 2 + WRITE_SIZE of that kernel's full
    # launch from the rocprofv3 --pmc passes of tools/track_batch_probe.py (tools/pmc_track.py -> profiles/pmc_track.json; separate runs, the
    # guide's gfx950 correction), scaled to this batch.  A frame's working set (~1.2 GB of records per launch) is only five times the 256 MiB
    # Infinity Cache, whose hits the counters include: `frac` is of the HBM peak and can exceed what HBM alone delivers.
    pmct = {}
    try:
        pmct = json.load(open(os.path.join(ROOT, "profiles", "pmc_track.json")))
    except Exception:
        pmct = {}

    def pmc_bytes(key):
        for k_, v_ in (pmct.get("kernels") or {}).items():
            if key in k_:
                return int(v_["hbm_bytes_per_robot_full_launch"] * B)
        return None
    lin_alone = kt["k_lin_laser"] + kt["k_lin_imu"] + kt["k_lin_small"]
    lin_traffic = [pmc_bytes(k_) for k_ in ("k_lin_laser_slab1" if (flags.value & 2) else "k_lin_laser<", "k_lin_imu_chain_multi", "k_lin_small")]
    cands = (("linearise (%s + k_lin_imu_chain_multi + k_lin_small, serial sum of the stand-alone times)" % ("k_lin_laser_slab1" if (flags.value & 2) else "k_lin_laser<false>"),
              lin_alone, lin_b, sum(lin_traffic) if all(lin_traffic) else None),
             ("k_lm_step_quad (TRACK topology: assembly of the two frames + prior, block elimination, back substitution; four windows per wave)", kt["k_lm_step"], stp_b, pmc_bytes("k_lm_step_quad")))
    dom = max(cands, key=lambda r: r[1])
    out["roofline"] = {"bound": "hbm (+ Infinity Cache: the per-launch working set is ~5x the 256 MiB cache)", "kernel": dom[0], "avg_launch_ms": round(dom[1], 4),
                       "achieved": round(B * dom[2] / (dom[1] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(B * dom[2] / (dom[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": dom[3],
                       "achieved_counter_gbs": round(dom[3] / (dom[1] * 1e-3) / 1e9, 1) if dom[3] else None,
                       "bytes_model": "analytic (track_model): bytes the kernel addresses per 2-frame window, all windows active",
                       "traffic_note": "FETCH_SIZE x 2 + WRITE_SIZE of a full launch (profiles/pmc_track.json, rocprofv3 --pmc passes of tools/track_batch_probe.py; not re-measured by this run); "
                                       "the fabric-side counters include Infinity Cache hits",
                       "other_kernel": {"kernel": [c_ for c_ in cands if c_ is not dom][0][0], "ms": round([c_ for c_ in cands if c_ is not dom][0][1], 4),
                                        "model_GBps": round(B * [c_ for c_ in cands if c_ is not dom][0][2] / ([c_ for c_ in cands if c_ is not dom][0][1] * 1e-3) / 1e9, 1),
                                        "traffic": [c_ for c_ in cands if c_ is not dom][0][3]}}
    out["parity_teacher_forced"] = tb.teacher_forced_parity(ids, cap)
    out["parity_teacher_forced"]["robots"] = [int(b) for b in ids]
    out["parity_teacher_forced"]["note"] = ("every frame of these robots: oracle solver::solve from the inputs the GPU batch had (states, laser_match poses, carried prior), bar 1e-6 on the "
                                            "state vector, then the oracle's marginalisation at the GPU's solved states (Delta_H relative; Delta_g / prior J^T R of the gradient sums' round-off scale)")
    if cpu:
        out["cpu_oracle"] = tb.cpu_oracle_frames_per_s()
        out["speedup_vs_cpu_1core"] = round(out["frames_per_s"] / out["cpu_oracle"]["frames_per_s"], 1)
    tb.bs.close()
    del tb
    torch.cuda.empty_cache()
    return out


def marg_reference(pyoracle, orc, win, x, mp, passes, prior=None):
    """The oracle's marginalisation (solver.cpp:257-442) at the linearisation point (x, mp) the GPU batch holds, `passes` times in a row
    (pass 2 carries the prior pass 1 wrote) -> per pass (Delta_H, Delta_g, prior X / J / R, the round-off scale of Delta_g).

    Scale of Delta_g ("state the bar that is physically right"): Delta_g = g_r - W g_m with W = H_rm H_mm^-1 and
    g = -J^T R a sum of ~6 300 signed terms that cancels towards 0 at the optimum, so its round-off does not scale with |Delta_g| but
    with a = |J|^T |R| (what one ulp of every term adds up to): a_r + |W| a_m.  |Delta_g_gpu - Delta_g_oracle| is compared with THAT."""
    w = pyoracle.Window(win)
    w["states"][:] = x.reshape(w["states"].shape)
    w["match_pose"][:] = mp.reshape(w["match_pose"].shape)
    orc.set_prior(prior)              # the linearised block the solver carries in (solver.h:31-37); None: no prior rows
    out = []
    # (single-threaded BLAS for the few dense products below: numpy's thread pool — one spinning thread per logical CPU of a 256-thread host inside a
    #  16-CPU cgroup — starved the launching thread of the NEXT GPU measurement: the fixed-10 side measurement behind the gate read 310 k instead of 515 k)
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    for _ in range(passes):
        orc.marginalization(w)
        m = orc.marg_pieces()
        J, R, H = m["J"], m["R"], m["H"]
        N = H.shape[0]
        a = np.abs(J).T @ np.abs(R)
        Hmm, Hrm = H[:N - 15, :N - 15], H[N - 15:, :N - 15]
        W = np.linalg.solve(Hmm, Hrm.T).T
        X, Jp, Rp = orc.get_prior()
        # H_rr_scale: |H_rr|max — Delta_H = H_rr - H_rm H_mm^-1 H_mr cancels against it (a window with little information on the newest frame keeps
        # 5e7 of 1e11): the round-off of Delta_H scales with the terms that cancel, not with what is left (tests/soak/soak_slab.py)
        out.append(dict(dH=m["Delta_H"].copy(), dg=m["Delta_g"].copy(), X=X.copy(), J=Jp.copy(), R=Rp.copy(), g_scale=float((a[N - 15:] + np.abs(W) @ a[:N - 15]).max()),
                        H_rr_scale=float(np.abs(H[N - 15:, N - 15:]).max())))
    if limit is not None:
        limit.restore_original_limits()
    return out


def parity_gate(liw, prm, windows, gate_ids, bs, marg_out, iters_cap, dev):
    """BASELINE.md 3 "equality gate": results of the TIMED batch (final states, LM iteration count, termination, marginalisation
    Delta_H / Delta_g of windows `gate_ids`) and the per-iteration state history of the same windows (re-solved with history
    recording) against the CPU oracle.  State bar: |x_gpu - x_cpu|_inf / |x_cpu|_inf <= 1e-6 after every iteration (north_star)."""
    from oracle import pyoracle
    import torch
    orc = pyoracle.Oracle(prm)
    orc.set_max_iterations(iters_cap)
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
    xg = bs.states()
    mpg = bs.t["match_pose"].cpu().numpy().reshape(bs.B, bs.n, 12)
    summ = bs.summaries()
    hs = liw.BatchSolver(prm, [windows[b] for b in gate_ids], device=dev, history_records=iters_cap + 1)
    hs.solve(liw.LIW_MODE_INIT, iters_cap)
    torch.cuda.synchronize()
    hist = hs.history()
    hsum = hs.summaries()
    out = {"windows": len(gate_ids), "window_ids": [int(b) for b in gate_ids], "max_rel_state_err_final": 0.0, "max_rel_state_err_per_iteration": 0.0, "iterations_equal": True,
           "terminations_equal": True, "max_rel_marg_Delta_H": 0.0, "max_rel_marg_Delta_g": 0.0, "max_marg_Delta_g_of_roundoff_scale": 0.0,
           "max_rel_prior_JtJ": 0.0, "max_prior_JtR_of_roundoff_scale": 0.0, "tolerance_state": 1e-6,
           "tolerance_marg": {"Delta_H": 1e-12, "Delta_g_of_roundoff_scale": 1e-11, "prior_JtJ": 1e-11, "prior_JtR_of_roundoff_scale": 1e-10},
           "tolerance_marg_note": "Delta_H and the new prior's J^T J relative to |Delta_H|max (BASELINE.md 3 asks 1e-10 on H; measured 1e-15).  Delta_g and the prior's J^T R: "
                                  "relative to the round-off scale of the gradient sums a_r + |H_rm H_mm^-1| a_m, a = |J|^T |R| — g = -J^T R cancels towards 0 at the optimum, so "
                                  "|Delta_g| itself is not the scale of its error (max_rel_marg_Delta_g, the old measure, is kept for comparison with earlier rounds)"}
    for k, b in enumerate(gate_ids):
        w = pyoracle.Window(windows[b])
        orc.set_prior(None)
        orc.init_solve(w)
        so = orc.summary()
        its = orc.iterations()
        out["max_rel_state_err_final"] = max(out["max_rel_state_err_final"], rel(xg[b], w["states"].reshape(xg[b].shape)))
        for srec in (summ[b], hsum[k]):
            out["iterations_equal"] &= bool(srec["iterations"] == so["iterations"])
            out["terminations_equal"] &= bool(srec["termination"] == so["termination"])
        for it in range(min(len(its), hist.shape[0], so["iterations"] + 1)):
            out["max_rel_state_err_per_iteration"] = max(out["max_rel_state_err_per_iteration"], rel(hist[it, k], its[it]["x"].reshape(hist[it, k].shape)))
        if marg_out is not None and not prm.get("fast_mode"):
            # same linearisation point on both sides (|H| ~ 1e11: a 1e-13 state difference alone moves g by 1e-2)
            w["states"][:] = xg[b].reshape(w["states"].shape)
            w["match_pose"][:] = mpg[b].reshape(w["match_pose"].shape)
            mo = marg_reference(pyoracle, orc, windows[b], xg[b], mpg[b], 1)[0]
            dH = marg_out[1][b].cpu().numpy().reshape(15, 15)
            dg = marg_out[2][b].cpu().numpy()
            pJ = bs.t["prior_J"][225 * b:225 * (b + 1)].cpu().numpy().reshape(15, 15)     # the prior k_marg_schur_eigq wrote (solver.cpp:390-402)
            pR = bs.t["prior_R"][15 * b:15 * (b + 1)].cpu().numpy()
            sc = float(np.abs(mo["dH"]).max())
            out["max_rel_marg_Delta_H"] = max(out["max_rel_marg_Delta_H"], rel(dH, mo["dH"]))
            out["max_rel_marg_Delta_g"] = max(out["max_rel_marg_Delta_g"], rel(dg, mo["dg"]))
            out["max_marg_Delta_g_of_roundoff_scale"] = max(out["max_marg_Delta_g_of_roundoff_scale"], float(np.abs(dg - mo["dg"]).max() / mo["g_scale"]))
            out["max_rel_prior_JtJ"] = max(out["max_rel_prior_JtJ"], float(np.abs(pJ.T @ pJ - mo["J"].T @ mo["J"]).max() / sc))
            out["max_prior_JtR_of_roundoff_scale"] = max(out["max_prior_JtR_of_roundoff_scale"], float(np.abs(pJ.T @ pR - mo["J"].T @ mo["R"]).max() / mo["g_scale"]))
    hs.close()
    tm_ = out["tolerance_marg"]
    out["passed"] = bool(out["iterations_equal"] and out["terminations_equal"] and out["max_rel_state_err_final"] <= 1e-6 and
                         out["max_rel_state_err_per_iteration"] <= 1e-6 and out["max_rel_marg_Delta_H"] <= tm_["Delta_H"] and
                         out["max_marg_Delta_g_of_roundoff_scale"] <= tm_["Delta_g_of_roundoff_scale"] and out["max_rel_prior_JtJ"] <= tm_["prior_JtJ"] and
                         out["max_prior_JtR_of_roundoff_scale"] <= tm_["prior_JtR_of_roundoff_scale"])
    for k in list(out):
        if isinstance(out[k], float):
            out[k] = float("%.3e" % out[k])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LIW_BENCH_BATCH", 49152)),
                    help="windows per GPU per step (49 152 = twelve full rounds of the quad step kernel, 12 288 waves on 1 024 SIMDs; round 4 on one "
                         "box: 24 576 -> 106.0 k, 32 768 -> 107.2 k, 49 152 -> 108.7 - 110.3 k, 65 536 -> 109.0 k solves/s: the late, partly finished "
                         "launches and the whole-round quantisation of the one-wave-per-SIMD kernels weigh less; ~1 MB of HBM per window)")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--laser", type=int, default=2000)
    ap.add_argument("--iters", type=int, default=50, help="LM iteration cap (Ceres default 50)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=200, help="timed solves of the 1-core CPU leg (after --cpu-warmup untimed ones; SURVEY 8d: >= 200)")
    ap.add_argument("--cpu-warmup", type=int, default=20, help="SURVEY 8d: >= 20")
    ap.add_argument("--replay", nargs="?", const="", default=None, metavar="FLATLOG",
                    help="C3 leg: replay a flat sensor log (default: a synthetic corridor-rate log of --replay-seconds) through the C++ driver on the "
                         "GPU and the oracle twin on the host; runs by default at N = 1 unless --no-single")
    ap.add_argument("--replay-seconds", type=float, default=60.0)
    ap.add_argument("--replay-keep", type=int, default=1, help="frames kept after a tracking solve (1 = the reference's policy; 29 = the C3 30-KF window)")
    ap.add_argument("--replay-keep30-seconds", type=float, default=20.0, help="length of the synthetic log of the second replay leg (keep = 29: BASELINE C3's 30-KF window; 0 = skip it)")
    ap.add_argument("--distinct", type=int, default=64, help="fully generated windows with distinct seeds per rank (the rest of the batch tiles them with state jitter)")
    ap.add_argument("--gate-windows", type=int, default=4, help="windows of the timed batch whose results are checked against the oracle (parity gate)")
    ap.add_argument("--cpu-procs", type=int, default=64, help="processes of the all-cores CPU baseline leg (capped at the core count)")
    ap.add_argument("--skip-sharded", action="store_true")
    ap.add_argument("--sharded-windows", type=int, default=256, help="windows of the factor-sharded C4 section (256; tests of the 8-rank control flow on one GPU use fewer)")
    ap.add_argument("--converging-scale", type=float, default=0.1, help="initial state error of the `converging_c2` side measurement, as a fraction of the C2 perturbation")
    ap.add_argument("--no-single", action="store_true", help="skip the B=1 latency measurement (clean per-kernel profiles)")
    ap.add_argument("--track-batch", type=int, default=int(os.environ.get("LIW_BENCH_TRACK_BATCH", 49152)), help="robots of the batched TRACK leg (the reference's steady state: 2-frame window, "
                    "solve + marginalization per laser frame, prior carried); 0 = skip")
    ap.add_argument("--ragged-batch", type=int, default=int(os.environ.get("LIW_BENCH_RAGGED_BATCH", 49152)), help="windows of the ragged-workload leg (same totals as C2, per-window L and per-frame "
                    "group sizes drawn wide); 0 = skip")
    ap.add_argument("--track-frames", type=int, default=8, help="timed consecutive frames of the batched TRACK leg")
    ap.add_argument("--track-gate-windows", type=int, default=6, help="robots of the TRACK leg whose every frame is checked teacher-forced against the oracle")
    ap.add_argument("--record-md", default=None, help="after the timed region, write a reference-shaped `record` table (labels "
                    "'solve' / 'marginalization', src/utilies/record.h) of per-batch durations to this path")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the estimator has no CPU fallback)")
    # LIW_BENCH_SHARE_GPU=1 (testing aid on a 1-GPU box): every rank on cuda:0 with the gloo backend, which exercises the
    # multi-rank control flow of this script (barriers, max-over-ranks timing, the factor-sharded loop) without RCCL
    share = os.environ.get("LIW_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device(dev), timeout=datetime.timedelta(minutes=5))

    liw = importlib.import_module("2dliw-slam_amd")
    synth = importlib.import_module("2dliw-slam_amd.synth")
    prm = synth.office_params()
    n, L, B = args.frames, args.laser, args.batch

    # the batch is tiled ON THE DEVICE from the distinct windows (the B-fold host concatenation was ~25 GB per rank)
    windows = make_tiled(liw, synth, prm, B, n, L, seed0=20240 + 7919 * rank, n_base=args.distinct)
    bs = liw.BatchSolver(prm, windows.base, device=dev, tile=windows.tile())
    rss_setup = peak_rss_mb()
    x0 = bs.t["x"].clone()
    mp0 = bs.t["match_pose"].clone()

    last_marg = [None]

    def one_step():
        bs.t["x"].copy_(x0)
        bs.t["match_pose"].copy_(mp0)
        bs.t["has_prior"].zero_()
        bs.solve(liw.LIW_MODE_INIT, args.iters)
        last_marg[0] = bs.marginalize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    bs.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    tm = bs.get_timing()
    bs.set_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if args.record_md and rank == 0:   # outside the timed region: the extra events would serialise solve and marginalisation
        rec = liw.outputs.Record()
        for _ in range(max(2, args.steps)):
            bs.t["x"].copy_(x0); bs.t["match_pose"].copy_(mp0); bs.t["has_prior"].zero_()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record(); bs.solve(liw.LIW_MODE_INIT, args.iters); ev[1].record(); bs.marginalize(); ev[2].record()
            torch.cuda.synchronize()
            rec.add_time("solve", ev[0].elapsed_time(ev[1]) * 1e3)
            rec.add_time("marginalization", ev[1].elapsed_time(ev[2]) * 1e3)
            rec.add_record("windows_per_batch", B)
        rec.write(args.record_md)
    summ = bs.summaries()
    iters = np.array([s["iterations"] for s in summ])
    term = np.array([s["termination"] for s in summ])

    # ---- parity gate on the timed batch itself (rank 0; the oracle is the checker, never the thing measured)
    gate = None
    if rank == 0 and args.gate_windows > 0:
        # first / last / middle window of the timed batch (the last two are jittered copies with their own LM paths: the tail wave and the
        # last slab of the launch are compared, not only slab 0), the last distinct window, then further jittered copies
        nb_ = min(args.distinct, B)
        gate_ids = []
        for c_ in (0, B - 1, B // 2, nb_ - 1, min(nb_ + 1, B - 1), max(B - 65, 0), 63 % B, 64 % B):
            if c_ not in gate_ids:
                gate_ids.append(c_)
        gate_ids = gate_ids[:args.gate_windows]
        gate = parity_gate(liw, prm, windows, gate_ids, bs, last_marg[0], args.iters, dev)

    # ---- roofline of the dominant kernel (k_linearize), from the HIP-event durations of the timed region
    # window-launches: every window takes part in the initial linearisation, one per LM iteration, and the
    # marginalisation linearisation (one pose free per laser block there)
    bytes_init = algorithmic_bytes(n, L, True)
    bytes_marg = algorithmic_bytes(n, L, False)
    lm_window_launches = int((iters + 1).sum())
    alg_bytes_total = args.steps * (lm_window_launches * bytes_init + B * bytes_marg)
    lin_time_s = tm["linearize_ms"] * tm["linearize_launches"] * 1e-3
    achieved = alg_bytes_total / lin_time_s / 1e9 if lin_time_s > 0 else 0.0
    # HBM traffic per full launch from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, calibrated as
    # tools/pmc_traffic.py documents), scaled from the profiled batch to this one: traffic is linear in the windows
    traffic = step_traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pmcj = {}
    if os.path.exists(pmc):
        try:
            pmcj = json.load(open(pmc))
            traffic = int(pmcj["k_linearize_hbm_bytes_per_window"] * B)
            step_traffic = int(pmcj["k_lm_step_hbm_bytes_per_launch"] / pmcj["windows"] * B)
        except Exception:
            traffic = step_traffic = None
    # `frac` is a BANDWIDTH: the bytes the role kernels address (linearise_model, an analytic count like step_model's; the PMC counters of
    # profiles/pmc_traffic.json agree within a few per cent) over the HIP-event time of the bracket.  SURVEY 8d's "materialised-J" convention
    # (312 B per laser block ... as if every Jacobian block were written to HBM) is kept beside it as frac_survey_convention: the kernels fuse
    # J^T J, no Jacobian ever reaches HBM, and that figure exceeds 1 — it measures the convention, not the memory system.
    lmod = linearise_model(n, L)
    model_bytes_total = args.steps * (lm_window_launches + B) * lmod["total"]
    achieved_model = model_bytes_total / lin_time_s / 1e9 if lin_time_s > 0 else 0.0
    roofline = {"schema_version": 2,
                "schema_note": "version 2 (rounds 5+): achieved / frac = analytic bytes of the role kernels (frac_model in ADVICE r5's wording) over the bracket time; "
                               "version 1 (BENCH_r01 .. r04): achieved / frac were what is now achieved_survey_convention / frac_survey_convention — compare those keys across rounds",
                "bound": "hbm", "kernel": "linearise = k_lin_laser_slab + k_lin_imu_chain + k_lin_small (Jacobian evaluation fused into J^T J partial sums, one HIP-event bracket)",
                "achieved": round(achieved_model, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_model / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_model": "analytic: bytes the role kernels address per window-linearisation (bench.py linearise_model: packed laser rows 64 B x L, packed IMU "
                               "records 1 536 B x (n-1), wheel inputs, states; group / per-frame / wheel / ground records and cost slots written) x window-linearisations of the timed region",
                "model_bytes_per_window": lmod["total"], "model_bytes_per_role": {k: lmod[k] for k in ("laser", "imu", "small")},
                # the same launch priced by the HBM bytes the PMC counters saw: rocprof HBM GB/s
                "achieved_counter_gbs": round(traffic / (tm["linearize_ms"] * 1e-3) / 1e9, 2) if (traffic and tm["linearize_ms"] > 0) else None,
                "frac_counter": round(traffic / (tm["linearize_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (traffic and tm["linearize_ms"] > 0) else None,
                "traffic_note": "traffic / frac_counter: FETCH_SIZE + WRITE_SIZE of the role kernels from the rocprofv3 --pmc passes recorded in profiles/pmc_traffic.json (build named there), per full launch of this batch; not re-measured by this run",
                "achieved_survey_convention": round(achieved, 2), "frac_survey_convention": round(achieved / HBM_PEAK_GBS, 4),
                "survey_convention_note": "SURVEY 8d algorithmic bytes (materialised-J: every Jacobian block written) over the same time; above 1 because no kernel writes a Jacobian",
                "avg_launch_ms": round(tm["linearize_ms"], 5), "launches": tm["linearize_launches"],
                "algorithmic_bytes_per_window": bytes_init,
                "algorithmic_bytes_per_full_launch": B * bytes_init,
                "lm_step_kernel_avg_ms": round(tm["step_ms"], 5), "lm_step_launches": tm["step_launches"],
                "linearize_only_windows_per_s": round(B / (tm["linearize_ms"] * 1e-3), 1) if tm["linearize_ms"] > 0 else None}
    # second kernel of the step: the LM step (assembly + elimination + back substitution).  Since round 3 batches above 2 048 windows run
    # k_lm_step_quad (four windows per wave, DPP row broadcasts, one-frame-ahead LDS-DMA): one wave per SIMD, streaming the partial sums
    # and its back-substitution record.  `achieved` prices the analytic bytes of the kernel (step_model) over
    # the windows that took the step, `achieved_counter_gbs` the PMC-counted bytes of profiles/pmc_traffic.json.
    sm = step_model(n)
    succ = np.array([s_["successful"] for s_ in summ])
    step_window_launches = int(iters.sum()) * args.steps            # window-iterations that ran both sweeps
    step_time_s = tm["step_ms"] * tm["step_launches"] * 1e-3
    step_bytes = (sm["read"] + sm["write"]) * step_window_launches
    step_roof = {"kernel": "k_lm_step_quad (normal-equation assembly + block-tridiagonal-arrow LM elimination + back substitution; four windows per wave)",
                 "bound": "hbm: both sweeps stage their records by LDS-DMA and move ~8 B per clock and CU (4.6 - 4.9 TB/s chip-wide, the practical rate of this read / write mix); one wave per SIMD, four per CU (38 kB of LDS each); DESIGN 4",
                 "avg_launch_ms": round(tm["step_ms"], 5), "launches": tm["step_launches"],
                 "analytic_bytes_per_window_iteration": sm["read"] + sm["write"],
                 "achieved": round(step_bytes / step_time_s / 1e9, 2) if step_time_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(step_bytes / step_time_s / 1e9 / HBM_PEAK_GBS, 4) if step_time_s > 0 else None,
                 "flops_per_window_iteration": sm["flops"],
                 "flops_frac": round(sm["flops"] * step_window_launches / step_time_s / 78.6e12, 4) if step_time_s > 0 else None,
                 "traffic": step_traffic,
                 "achieved_counter_gbs": round(step_traffic / (tm["step_ms"] * 1e-3) / 1e9, 2) if (step_traffic and tm["step_ms"] > 0) else None,
                 "frac_counter": round(step_traffic / (tm["step_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (step_traffic and tm["step_ms"] > 0) else None,
                 "window_iterations_per_s": round(B / (tm["step_ms"] * 1e-3), 1) if tm["step_ms"] > 0 else None,
                 "pmc_issue_stats": pmcj.get("k_lm_step_issue_stats")}
    # linearise + step together, per window and LM iteration, against the bytes an iteration cannot avoid reading
    floor = input_floor_bytes(n, L)
    lin_w = pmcj.get("k_linearize_hbm_bytes_per_window")
    stp_w = (pmcj.get("k_lm_step_hbm_bytes_per_launch", 0.0) / pmcj["windows"]) if pmcj.get("windows") else None
    hbm_iter = {"input_floor": floor, "input_floor_packed": input_floor_packed_bytes(n, L), "linearise_counter": int(lin_w) if lin_w else None, "step_counter": int(stp_w) if stp_w else None,
                "step_analytic": sm["read"] + sm["write"],
                "total_counter": int(lin_w + stp_w) if (lin_w and stp_w) else None,
                "ratio_to_floor": round((lin_w + stp_w) / floor, 3) if (lin_w and stp_w) else None,
                "source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the build named there; per-kernel calibration, tools/pmc_traffic.py)"}
    roofline["frac_recomputed_serial_roles"] = pmcj.get("roofline_frac_serial_roles")
    # Ceres evaluates a rejected candidate's residuals only; this path linearises every candidate speculatively (cost and Jacobians).  The
    # same launches priced the way Ceres would have done the work: accepted steps at the full algorithmic bytes, rejected ones at the
    # residual-only bytes (records read, residuals written)
    rejected = iters - succ
    res_only = L * (104 + 16) + (n - 1) * (3728 + 120 + 168 + 24) + 2 * n * n * 16 + n * 120
    ceres_bytes = args.steps * (int((succ + 1).sum()) * bytes_init + int(rejected.sum()) * res_only + B * bytes_marg)
    roofline["lm_rejected_steps_mean"] = round(float(rejected.mean()), 3)
    roofline["achieved_survey_convention_ceres_work"] = round(ceres_bytes / lin_time_s / 1e9, 2) if lin_time_s > 0 else None
    # fixed K = 10 LM iterations + marginalisation (SURVEY 8d asks for both stopping rules), untimed side measurement
    k10 = None
    if rank == 0 and world == 1 and not args.no_single:
        bs.t["x"].copy_(x0); bs.t["match_pose"].copy_(mp0); bs.t["has_prior"].zero_()
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        bs.solve(liw.LIW_MODE_INIT, 10)
        if os.environ.get("LIW_BENCH_K10_SPLIT"):
            torch.cuda.synchronize()
            sys.stderr.write("k10: solve %.2f ms\n" % (1e3 * (time.perf_counter() - t0_)))
        bs.marginalize()
        torch.cuda.synchronize()
        k10 = round(B / (time.perf_counter() - t0_), 1)
        if os.environ.get("LIW_BENCH_K10_SPLIT"):
            sys.stderr.write("k10: total %.2f ms, iterations histogram %s\n" % (1e3 * (time.perf_counter() - t0_), np.bincount([s_["iterations"] for s_ in bs.summaries()]).tolist()))

    # ---- "converging" C2 variant: the same windows started closer to the truth, so that the LM stops on Ceres' function
    #      tolerance instead of crawling along the ground_factor_q cone into the iteration cap; `value` stays on the workload above
    conv = None
    if rank == 0 and world == 1 and not args.no_single:
        truth = torch.from_numpy(np.stack([np.asarray(w_["truth_states"]) for w_ in windows.base])[np.arange(B) % len(windows.base)].reshape(-1)).to(dev)
        f = args.converging_scale
        xcv = truth + f * (x0 - truth)
        mpc = mp0.clone().reshape(B, n, 12)
        xv = xcv.reshape(B, n, 15)
        mpc[:, :, 0:6] = xv[:, 0:1, 0:6]
        mpc[:, :, 6:12] = xv[:, :, 0:6]
        bs.t["x"].copy_(xcv); bs.t["match_pose"].copy_(mpc.reshape(-1)); bs.t["has_prior"].zero_()
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        bs.solve(liw.LIW_MODE_INIT, args.iters)
        bs.marginalize()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0_
        sc = bs.summaries()
        itc, tmc = np.array([s_["iterations"] for s_ in sc]), np.array([s_["termination"] for s_ in sc])
        conv = {"initial_state_error_scale": f, "solves_per_s": round(B / dt_, 1), "lm_iterations_mean": round(float(itc.mean()), 2),
                "terminations": {str(int(k)): int((tmc == k).sum()) for k in np.unique(tmc)},
                "stopped_on_function_tolerance_pct": round(100.0 * float((tmc == 2).mean()), 1),
                "note": "states = truth + scale x (C2 perturbation); same factors, same kernels; untimed-region side measurement (one pass)"}

    # ---- every kernel of an LM iteration ALONE over the whole batch (all windows active), HIP events inside the library: the per-role
    #      bounds the concurrent bracket above cannot show.  MFMA counts are static properties of the kernels
    #      (20 v_mfma_f64_16x16x4 per IMU block: 8 whitening + 12 Gram; 4 per eliminated frame in the marginalisation chain), confirmed by
    #      the SQ_INSTS_MFMA pass in profiles/ (tools/pmc_stall_passes.sh)
    ktimes = None
    if rank == 0 and world == 1 and not args.no_single:
        bs.t["x"].copy_(x0); bs.t["match_pose"].copy_(mp0); bs.t["has_prior"].zero_()
        kt = bs.time_kernels(liw.LIW_MODE_INIT, 3)
        F64 = 78.6e12
        mf = 2048.0                                    # flops of one v_mfma_f64_16x16x4_f64
        lb, ib = B * L, B * (n - 1)
        ris = pmcj.get("role_issue_stats", {}) if pmcj else {}

        def issue(k):
            return (ris.get(k) or {}).get("active_inst_frac")
        t = {k: v * 1e-3 for k, v in kt.items()}
        lmod_ = linearise_model(n, L)

        def hbm_role(role, kern, frac=False):   # analytic bytes of the role (linearise_model) over its stand-alone time
            gbs = B * (lmod_[role]["read"] + lmod_[role]["write"]) / t[kern] / 1e9 if t[kern] > 0 else 0.0
            return round(gbs / HBM_PEAK_GBS, 4) if frac else round(gbs, 1)
        schur_bytes = B * ((n * PIFS_BYTES if B >= 1024 else (n - 1) * PIS_BYTES) + (n - 1) * PWS_BYTES + n * (LP_BYTES + PGS_BYTES))   # per-frame IMU records from 1 024 windows on
        laser_flops = 500.0                            # essential fp64 flops of one laser_factor block (2-D scan, both poses free): 32 world points,
        #                                                15 line direction, 81 its three rotation derivatives, 2 x 78 rows, 180 pair products, 28 lengths / weight
        ktimes = {
            "batch": B, "note": "each kernel launched alone, every window active (HIP events, 3 repeats); fractions are of the 78.6 TFLOP/s fp64 peak "
                                "(vector = matrix on MI355X, and they share the pipe: tools/ubench/mfma_valu_f64)",
            "k_lin_laser": {"ms": round(kt["k_lin_laser"], 4), "hbm_GBps": hbm_role("laser", "k_lin_laser"), "hbm_frac": hbm_role("laser", "k_lin_laser", True),
                            "survey_convention_GBps": round(lb * BYTES_LASER_BOTH / t["k_lin_laser"] / 1e9, 1),
                            "survey_convention_note": "SURVEY 8d's materialised-J bytes (312 B per block) over a kernel that never writes a Jacobian (8.4 TB/s-like values measure the convention, not HBM)",
                            "essential_flops_per_block": laser_flops,
                            "flops_frac": round(lb * laser_flops / t["k_lin_laser"] / F64, 4),
                            "kernel": "k_lin_laser_slab (a lane per (window, frame) group over per-solve packed rows: batches of >= 2 048 (slab, frame) waves of 2-D scans)" if (B + 63) // 64 * n >= 2048 and not os.environ.get("LIW_NO_LASER_SLAB") else "k_lin_laser<true> (a lane per block, wave reduction per group)",
                            "instructions": "lane-per-group kernel: ~274 VALU instructions per row of 64 blocks (446 until the block algebra was folded late in round 5), no cross-lane reduction, records staged through LDS and stored as 256-byte runs; lane-per-block kernel: ~1000 per 64-block chunk (per-group wave reduction ~260 per group end, second masked round of pair products where a chunk straddles two groups, transform reads from LDS, masks)",
                            "valu_issue_frac": issue("k_lin_laser"), "bound": "lane-per-group kernel alone: HBM at the practical rate of its read / write mix (8.0 GB per launch, probe builds: 1.30 ms cache-resident rows, 1.14 ms loads only; tools/ubench/hbm_stream: 5.3 - 5.7 TB/s for such mixes); inside the linearise bracket the three role kernels share the SIMDs and the bracket follows the SUM of their fp64 instruction streams; lane-per-block kernel: fp64 VALU issue"},
            "k_lin_imu": {"ms": round(kt["k_lin_imu"], 4), "hbm_GBps": hbm_role("imu", "k_lin_imu"), "hbm_frac": hbm_role("imu", "k_lin_imu", True), "mfma_insts": int(ib * 20), "mfma_util": round(ib * 20 * mf / t["k_lin_imu"] / F64, 4),
                          "flops_frac": round(ib * (20 * mf + 3 * 2600.0) / t["k_lin_imu"] / F64, 4), "valu_issue_frac": issue("k_lin_imu"),
                          "bound": "fp64 pipe shared by MFMA and VALU (their times add), two waves per SIMD"},
            "k_lin_small": {"ms": round(kt["k_lin_small"], 4), "hbm_GBps": hbm_role("small", "k_lin_small"), "hbm_frac": hbm_role("small", "k_lin_small", True), "valu_issue_frac": issue("k_lin_small"), "bound": "VALU issue / partial-sum writes"},
            "k_lm_step": {"ms": round(kt["k_lm_step"], 4), "flops_frac": round(B * step_model(n)["flops"] / t["k_lm_step"] / F64, 4),
                          "hbm_GBps": round(B * (step_model(n)["read"] + step_model(n)["write"]) / t["k_lm_step"] / 1e9, 1),
                          "hbm_frac": round(B * (step_model(n)["read"] + step_model(n)["write"]) / t["k_lm_step"] / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline_schur": {"kernel": "k_marg_schur_chain (chain Schur complement of frames 0..n-2 onto the newest frame, one wave per window) + k_marg_schur_eigq (15x15 Jacobi eigen square root, four windows per wave); one HIP-event bracket around both",
                               "avg_launch_ms": round(kt["k_marg_schur"], 4), "mfma_insts": int(ib * 4),
                               "mfma_util": round(ib * 4 * mf / t["k_marg_schur"] / F64, 5) if t["k_marg_schur"] > 0 else None,
                               "bytes": int(schur_bytes),
                               "GBps": round(schur_bytes / t["k_marg_schur"] / 1e9, 1) if t["k_marg_schur"] > 0 else None,
                               "note": "the reference's dense 6 337 x 450 J^T J (2.57 GFLOP) + 435^3 LU inverse is a 29-step chain of 15x15 eliminations here (~0.25 MFLOP per window): the Schur step is "
                                       "latency-bound on one wave per window and its matrix-core share is small by construction"},
            "k_lin_laser_marg": {"ms": round(kt["k_lin_laser_marg"], 4)},
            "iteration_serial_ms": round(kt["k_lin_laser"] + kt["k_lin_imu"] + kt["k_lin_small"] + kt["k_lm_step"], 4)}

    # ---- C3 leg: sensor-log replay, GPU and CPU oracle in the same run
    replay_out = replay30_out = None
    if rank == 0 and world == 1 and (args.replay is not None or not args.no_single):
        try:
            replay_out = replay_leg(liw, synth, prm, args.replay or None, args.replay_seconds, args.replay_keep, teacher_forced=args.replay_keep > 1)
        except Exception as e:   # a side measurement must never take the headline line down
            replay_out = {"error": repr(e)[:300]}
        # BASELINE C3 at its stated window: the same driver with keep = 29 (30 frames at solve time), GPU and oracle twin on the same policy in
        # the same run, every tracking solve also teacher-forced
        if args.replay_keep == 1 and args.replay_keep30_seconds > 0:
            try:
                replay30_out = replay_leg(liw, synth, prm, args.replay or None, args.replay_keep30_seconds, 29, teacher_forced=True)
            except Exception as e:
                replay30_out = {"error": repr(e)[:300]}

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores, rank 0, N = 1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        orc = pyoracle.Oracle(prm)
        w = pyoracle.Window(windows[0])
        secs, it = orc.time_solves_each(w, args.cpu_warmup, args.cpu_reps, args.iters, dense_product=True)
        med, p95 = float(np.median(secs)), float(np.percentile(secs, 95))
        cpu_model = ""
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
        except OSError:
            pass
        cpu = {"value": round(1.0 / med, 4), "unit": "solves/s", "cores": 1, "kind": "port",
               "sample": "%d warm-up + %d timed x (init_solve + marginalization) of window seed 20240 (n=%d, L=%d), %d LM iterations total, %.1f s; "
                         "value = 1 / median" % (args.cpu_warmup, args.cpu_reps, n, L, it, float(secs.sum())),
               "median_ms": round(1e3 * med, 2), "p95_ms": round(1e3 * p95, 2), "mean_ms": round(1e3 * float(secs.mean()), 2),
               "host_cpu_count": os.cpu_count(), "host_cpu_model": cpu_model, "host_cpu_quota": cpu_quota()}
        # ONE solve on many cores: the oracle with an OpenMP team over the residual blocks (what Ceres' num_threads does; the reference leaves
        # it at 1, solver.cpp:798) and over the rows of its dense Cholesky / J^T J — bit-identical results for any team size
        try:
            q = cpu["host_cpu_quota"]
            team = int(max(1, min(q["affinity_cpus"], q["cgroup_cpu_quota"] or q["affinity_cpus"], 64)))
            orc.set_threads(team)
            s2, it2 = orc.time_solves_each(pyoracle.Window(windows[0]), 2, 10, args.iters, dense_product=True)
            orc.set_threads(1)
            cpu["one_solve_all_cores"] = {"threads": team, "solves_per_s": round(1.0 / float(np.median(s2)), 3), "median_ms": round(1e3 * float(np.median(s2)), 2),
                                          "sample": "2 warm-up + 10 timed solves, OpenMP team over residual blocks / Cholesky rows / dense J^T J bands (oracle_set_threads)"}
        except Exception as e:
            cpu["one_solve_all_cores"] = {"error": str(e)[:160]}
        # the same port on ALL host cores, one independent window stream per core (the CPU analogue of the batched GPU run),
        # so that the batched ratio is not inflated by the reference's single-threadedness
        try:
            import concurrent.futures as cf
            ncpu = max(1, min(os.cpu_count() or 1, args.cpu_procs))
            jobs = [(prm, windows[0], 2, args.iters)] * ncpu
            import resource
            ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
            t0_ = time.perf_counter()
            import multiprocessing as mp
            with cf.ProcessPoolExecutor(max_workers=ncpu, mp_context=mp.get_context("spawn")) as ex:   # spawn: children never see the HIP runtime
                secs = list(ex.map(_cpu_worker, jobs))
            wall = time.perf_counter() - t0_
            ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            busy = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
            cpu["all_cores"] = {"processes": ncpu, "solves_per_s": round(sum(2.0 / t for t in secs), 2), "wall_s": round(wall, 1),
                                "effective_cores": round(busy / wall, 1),
                                "sample": "2 solves per process, window-parallel; rate = sum over processes of their own solve rates; "
                                          "effective_cores = children CPU time / wall (the box may cap the CPU quota below the core count: host_cpu_quota)"}
        except Exception as e:
            cpu["all_cores"] = {"error": str(e)[:160]}

    # ---- single-window latency (B = 1, the reference's own call pattern), hipGraph-captured launch sequence
    single = None
    if rank == 0 and world == 1 and not args.no_single:
        s1 = liw.BatchSolver(prm, windows[:1], device=dev)
        x1 = s1.t["x"].clone()
        reps = 5
        for rep in range(reps + 1):
            if rep == 1:
                torch.cuda.synchronize()
                ts = time.perf_counter()
            s1.t["x"].copy_(x1)
            s1.t["has_prior"].zero_()
            s1.solve(liw.LIW_MODE_INIT, args.iters, use_graph=True)
            s1.marginalize()
        torch.cuda.synchronize()
        ms1 = 1e3 * (time.perf_counter() - ts) / reps
        single = {"ms_per_solve": round(ms1, 3), "solves_per_s": round(1e3 / ms1, 1), "lm_iterations": s1.summaries()[0]["iterations"]}
        s1.close()

    # ---- steady-state tracking frame (the reference keeps a 2-frame window while tracking, trajectory.cpp:525-559):
    #      solver.solve + solver.marginalization on frames (k-1, k) with the prior of the previous marginalisation,
    #      through the single-window C ABI with HOST buffers (upload + synchronous calls included), CPU oracle beside it
    tracking = None
    if rank == 0 and world == 1 and not args.no_single:
        try:
            hp = liw.HostPreint(prm)
            d3 = synth.make_window(hp, prm, seed=515, n=3, L=120, laser_on_frame0=False)

            def sub(lo):
                o = dict(d3)
                o["n"] = 2
                for k in ("states", "match_pose"):
                    o[k] = np.asarray(d3[k]).reshape(3, -1)[lo:lo + 2].copy()
                o["has_match"] = np.asarray(d3["has_match"])[lo:lo + 2].copy()
                for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
                    o[k] = np.asarray(d3[k])[lo:lo + 1].copy()
                m = (np.asarray(d3["laser_frame"]) >= lo) & (np.asarray(d3["laser_frame"]) < lo + 2)
                o["laser_frame"] = (np.asarray(d3["laser_frame"])[m] - lo).astype(np.int32)
                o["laser_pts"] = np.asarray(d3["laser_pts"])[m].copy()
                return o
            slv = liw.Solver(prm)
            reps, tg, it_g = 200, 0.0, 0
            for rep in range(reps + 2):
                slv.set_prior(None)
                slv.set_window(liw.Window(sub(0)))
                slv.solve()
                slv.marginalization()
                w12 = liw.Window(sub(1))
                t0_ = time.perf_counter()
                slv.set_window(w12)
                sg = slv.solve()
                slv.marginalization()
                if rep >= 2:
                    tg += time.perf_counter() - t0_
                    it_g = sg["iterations"]
            tracking = {"ms_per_frame": round(1e3 * tg / reps, 3), "frames_per_s": round(reps / tg, 1), "lm_iterations": it_g,
                        "window": "n=2, %d laser blocks on the newest frame, prior on the older one" % int((np.asarray(d3["laser_frame"]) == 2).sum()),
                        "caller": "Python mirror (ctypes) of the C ABI; ms_per_frame_cpp_caller = the same frame, same window, same repetitions, driven from "
                                  "C++ through lvio_2d::solver (include/lvio_2d_solver.hpp: deque flattening + liw_set_window / liw_solve / liw_marginalize + "
                                  "scatter), the way the reference's trajectory.cpp:525-560 calls its solver"}
            try:
                cpp = track_frame_cpp(liw, d3, reps)
                tracking["ms_per_frame_cpp_caller"] = round(cpp["ms_per_frame"], 3)
                tracking["lm_iterations_cpp_caller"] = cpp["iterations"]
            except Exception as e:
                tracking["cpp_caller_error"] = str(e)[:200]
            if not args.no_cpu_baseline:
                from oracle import pyoracle
                orc2 = pyoracle.Oracle(prm)
                tc = 0.0
                for rep in range(reps):
                    orc2.set_prior(None)
                    w01 = pyoracle.Window(sub(0))
                    orc2.solve(w01)
                    orc2.marginalization(w01)
                    w12o = pyoracle.Window(sub(1))
                    t0_ = time.perf_counter()
                    orc2.solve(w12o)
                    orc2.marginalization(w12o)
                    tc += time.perf_counter() - t0_
                tracking["cpu_oracle_ms_per_frame"] = round(1e3 * tc / reps, 3)
        except Exception as e:   # a latency side-measurement must never take the headline line down
            tracking = {"error": str(e)[:200]}

    # ---- the same frame with the keep-N window policy of BASELINE config C3 (corridor replay, 30-KF window): tracking solve of a
    #      30-frame window (laser blocks of the newest frame, 29 IMU + 29 wheel blocks, prior on frame 28) + marginalisation of all 30
    #      frames (2 000 laser blocks), host buffers in and out, CPU oracle beside it
    keepn = None
    if rank == 0 and world == 1 and not args.no_single:
        try:
            hp = liw.HostPreint(prm)
            NK = 30
            dk = synth.make_window(hp, prm, seed=616, n=NK + 1, L=2000 + 67, laser_on_frame0=False)

            def subk(lo):
                o = dict(dk)
                o["n"] = NK
                for k in ("states", "match_pose"):
                    o[k] = np.asarray(dk[k]).reshape(NK + 1, -1)[lo:lo + NK].copy()
                o["has_match"] = np.asarray(dk["has_match"])[lo:lo + NK].copy()
                for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
                    o[k] = np.asarray(dk[k])[lo:lo + NK - 1].copy()
                m = (np.asarray(dk["laser_frame"]) >= lo) & (np.asarray(dk["laser_frame"]) < lo + NK)
                o["laser_frame"] = (np.asarray(dk["laser_frame"])[m] - lo).astype(np.int32)
                o["laser_pts"] = np.asarray(dk["laser_pts"])[m].copy()
                return o
            slv = liw.Solver(prm)
            reps, tg, it_g = 20, 0.0, 0
            for rep in range(reps + 2):
                slv.set_prior(None)
                slv.set_window(liw.Window(subk(0)))
                slv.solve()
                slv.marginalization()
                wk = liw.Window(subk(1))
                t0_ = time.perf_counter()
                slv.set_window(wk)
                sg = slv.solve()
                slv.marginalization()
                if rep >= 2:
                    tg += time.perf_counter() - t0_
                    it_g = sg["iterations"]
            keepn = {"ms_per_frame": round(1e3 * tg / reps, 3), "lm_iterations": it_g,
                     "window": "n=%d tracking topology (C3 shape), %d laser blocks in the window, prior on frame %d" % (NK, int(len(subk(1)["laser_frame"])), NK - 2)}
            if not args.no_cpu_baseline:
                from oracle import pyoracle
                orc3 = pyoracle.Oracle(prm)
                tc, rc = 0.0, 3
                for rep in range(rc):
                    orc3.set_prior(None)
                    w0 = pyoracle.Window(subk(0))
                    orc3.solve(w0)
                    orc3.marginalization(w0)
                    w1 = pyoracle.Window(subk(1))
                    t0_ = time.perf_counter()
                    orc3.solve(w1)
                    orc3.marginalization(w1)
                    tc += time.perf_counter() - t0_
                keepn["cpu_oracle_ms_per_frame"] = round(1e3 * tc / rc, 3)
                keepn["cpu_oracle_lm_iterations"] = orc3.summary()["iterations"]
        except Exception as e:   # a latency side-measurement must never take the headline line down
            keepn = {"error": str(e)[:200]}

    # ---- ragged workload: the C2 totals with per-window L drawn from [L / 4, 2 L] and per-frame groups from 0 to several
    #      hundred blocks — what real scans give do_match — through the same solve + marginalisation, beside the even C2 batch above
    ragged = None
    if rank == 0 and world == 1 and not args.no_single and args.ragged_batch > 0:
        try:
            ragged = ragged_leg(liw, synth, prm, dev, args.ragged_batch, n, L, args.iters, args.distinct)
        except Exception as e:   # a side measurement must never take the headline line down
            ragged = {"error": repr(e)[:300]}

    # ---- the reference's steady state at scale: batched 2-frame TRACK solves + marginalisation over consecutive frames
    track_batch = None
    if rank == 0 and world == 1 and not args.no_single and args.track_batch > 0:
        try:
            track_batch = track_batch_leg(liw, synth, prm, dev, args.track_batch, args.track_frames, args.distinct, args.track_gate_windows, cpu=not args.no_cpu_baseline)
        except Exception as e:   # a side measurement must never take the headline line down
            track_batch = {"error": repr(e)[:300]}

    # ---- factor-sharded mode (north_star's multi-GPU mode): C4-shaped windows, the laser blocks of every window split over the ranks,
    #      the compact laser record (45 pair totals per (window, frame)) exchanged once per LM iteration on the main stream while the
    #      IMU / wheel / ground roles still run on side streams.  Same total work at every N (strong scaling); N = 1 is the un-sharded
    #      reference point.  Two exchanges are measured at N > 1: RCCL all-reduce, and the one-shot all-gather + rank-order sum.
    sharded = None
    if not args.skip_sharded:
        try:
            Bs, Ls, Ks = args.sharded_windows, 20000, 10
            hp = liw.HostPreint(prm)
            wfull = [synth.make_window(hp, prm, seed=4242 + k, n=n, L=Ls) for k in range(2)]   # same seeds on every rank
            tile_s = dict(B=Bs, states=np.stack([np.asarray(wfull[k % 2]["states"], dtype=np.float64).reshape(n, 15) for k in range(Bs)]),
                          match_pose=np.stack([np.asarray(wfull[k % 2]["match_pose"], dtype=np.float64).reshape(n, 12) for k in range(Bs)]))
            sharded = {"workload": "C4: %d windows x (n=%d, L=%d) laser blocks split over %d rank(s), %d LM iterations, one exchange of the "
                                   "compact laser record per iteration%s" % (Bs, n, Ls, world, Ks, "" if world > 1 else " (none at 1 rank)"),
                       "scaling": "strong", "ranks": world,
                       "process_group": {"backend": dist.get_backend() if world > 1 else None, "world_size": dist.get_world_size() if world > 1 else 1}}
            # LIW_BENCH_P2P=1 (N > 1, real devices only): also time the native peer-write exchange (hipIpc areas + flag-synchronised kernels,
            # include/liw_window.h) — off by default: it has never run across xGMI
            xvars = ("allreduce", "oneshot") if world > 1 else ("allreduce",)
            if world > 1 and not share and os.environ.get("LIW_BENCH_P2P") == "1":
                xvars = xvars + ("p2p",)
            for xi, xch in enumerate(xvars):
                sb = liw.BatchSolver(prm, wfull, device=dev, rank=rank, world=world, exchange=xch, tile=tile_s)   # (laser blocks of the two distinct windows sharded, batch tiled on the device)
                if xch == "p2p":
                    sb.p2p_attach_ipc(liw.LIW_MODE_INIT)
                xs0 = sb.t["x"].clone()
                te = ts = 0.0
                for rep in range(3):
                    sb.t["x"].copy_(xs0)
                    sb.time_exchange = rep == 2           # third pass: HIP events around every exchange (not the pass that is timed end to end)
                    barrier()
                    ts_ = time.perf_counter()
                    sb.solve(liw.LIW_MODE_INIT, Ks)
                    barrier()
                    if rep == 1:
                        ts, te = ts_, time.perf_counter()
                xms, xn = sb.exchange_timing()
                same = True
                if world > 1:   # every rank must hold bit-identical states (identical sums -> identical LM paths)
                    ref = sb.t["x"].clone()
                    dist.broadcast(ref, src=0)
                    flag = torch.tensor([1.0 if torch.equal(ref, sb.t["x"]) else 0.0], dtype=torch.float64, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    same = bool(flag.item() == 1.0)
                res = {"solves_per_s": round(Bs / (te - ts), 3), "ms_per_lm_iteration": round(1e3 * (te - ts) / (Ks + 1), 3),
                       "exchange_ms_per_iteration": round(xms, 4) if world > 1 else 0.0, "exchanges": xn if world > 1 else 0,
                       "exchange_bytes_per_rank": sb.exchange_bytes(liw.LIW_MODE_INIT) if world > 1 else 0,
                       "uncompacted_record_bytes": int(sb.lay.laser_partial_bytes), "states_identical_across_ranks": same}
                if xi == 0:
                    sharded.update(res)
                    sharded["exchange"] = xch if world > 1 else None
                    sharded["allreduce_ms_per_iteration"] = res["exchange_ms_per_iteration"]
                    sharded["allreduce_bytes_per_iteration"] = res["exchange_bytes_per_rank"]
                elif xch == "p2p":
                    sharded["p2p_exchange"] = res
                else:
                    sharded["oneshot_exchange"] = res
                if world > 1:
                    # DESIGN 5's arithmetic for this record on one node (xGMI: ~153 GB/s per link and direction, ~10 us per collective launch + ~2 us
                    # per hop), printed beside the measurement so that the first multi-GPU run judges itself
                    xb = float(res["exchange_bytes_per_rank"])
                    link, launch, hop = 153e9, 10e-6, 2e-6
                    model = {"allreduce": launch + 2.0 * (world - 1) / world * xb / link + 2 * (world - 1) * hop,
                             "oneshot": launch + xb / link + hop + world * xb / 5e12, "p2p": xb / link + hop + world * xb / 5e12}[xch]
                    res["exchange_model_ms_per_iteration"] = round(1e3 * model, 4)
                    res["exchange_measured_over_model"] = round(res["exchange_ms_per_iteration"] / (1e3 * model), 2) if model > 0 else None
                    if xi == 0:
                        sharded["exchange_model_ms_per_iteration"] = res["exchange_model_ms_per_iteration"]
                        sharded["exchange_measured_over_model"] = res["exchange_measured_over_model"]
                if world > 1 and xi == 1:   # what exchange="auto" would keep on this machine at this record size (collective + sum, timed once)
                    sharded["auto_exchange_pick"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sb.pick_exchange(liw.LIW_MODE_INIT).items()}
                sb.close()
            if world > 1:
                ok_ws = dist.get_world_size() == args.gpus
                ok_id = bool(sharded.get("states_identical_across_ranks")) and all(
                    sharded[k]["states_identical_across_ranks"] for k in ("oneshot_exchange", "p2p_exchange") if k in sharded)
                sharded["self_check"] = {"process_group_world_size_equals_gpus": bool(ok_ws), "states_identical_across_ranks_every_exchange": bool(ok_id),
                                         "passed": bool(ok_ws and ok_id)}
        except Exception as e:   # never lose the headline line because of the secondary measurement
            import traceback
            sharded = {"error": repr(e)[:300], "trace": traceback.format_exc()[-600:]}
    rss_all = [peak_rss_mb()]
    if world > 1:
        rss_all = [None] * world
        dist.all_gather_object(rss_all, peak_rss_mb())
    if rank == 0:
        total = B * world * args.steps
        out = {"metric": "sliding-window solves/sec (30 KF, 2k scan pts)", "value": round(total / elapsed, 3), "unit": "solves/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "C2 synthetic: %d windows/GPU x (n=%d frames, L=%d laser_factor blocks, %d IMU + %d wheel, %d ground); "
                                      "step = init-topology LM solve (cap %d iters) + marginalization per window" % (B, n, L, n - 1, n - 1, 2 * n * n, args.iters),
                          "windows_per_gpu": B, "frames": n, "laser_blocks": L, "lm_iteration_cap": args.iters,
                          "parallelism": "windows replicated over %d GPU(s), no data-path collective" % world,
                          "lm_iterations_mean": float(iters.mean()), "terminations": {str(int(k)): int((term == k).sum()) for k in np.unique(term)}},
               "roofline": roofline, "roofline_lm_step": step_roof, "hbm_bytes_per_window_iteration": hbm_iter, "cpu_baseline": cpu, "parity_gate": gate}
        out["host_peak_rss_mb_per_rank"] = {"after_batch_setup_rank0": rss_setup, "end_of_run": rss_all,
                                            "note": "the batch is tiled in HBM from the distinct windows (BatchSolver(tile=...)); ru_maxrss of every rank"}
        capped = int((term == 4).sum())
        out["config"]["lm_iterations_histogram"] = {str(int(k)): int((iters == k).sum()) for k in np.unique(iters)}
        out["config"]["distinct_windows_per_gpu"] = min(args.distinct, B)
        out["config"]["lm_hits_iteration_cap_pct"] = round(100.0 * capped / len(term), 1)
        out["config"]["note"] = ("%.0f %% of the windows run into the %d-iteration cap on the GPU and on the CPU oracle alike (cone-shaped ground_factor_q "
                                 "residual, DESIGN 6); the marginalisation follows the init solve as in trajectory.cpp:446-479, where no prior rows exist yet "
                                 "(SURVEY 8d's 15 synthetic prior rows are exercised by tracking_frame_latency and the parity tests)" % (100.0 * capped / len(term), args.iters))
        if gate is not None and not gate["passed"]:
            # a timing is only accepted after the equality gate; no `value` without it
            out["value"] = None
            out["error"] = "parity gate failed"
            print(json.dumps(out))
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
        if cpu:
            out["speedup_vs_cpu_1core"] = round(out["value"] / cpu["value"], 1)
        if k10:
            out["solves_per_s_fixed_10_iterations"] = k10
        if conv:
            out["converging_c2"] = conv
        if single:
            out["single_window_latency"] = single
        if tracking:
            out["tracking_frame_latency"] = tracking
        if keepn:
            out["keep30_tracking_frame_latency"] = keepn
            out["keep30_tracking_frame_latency"]["note"] = "keep-N is this repository's window policy for BASELINE C3 / C5, not a reference behaviour (the reference keeps 1 frame)"
        if track_batch:
            out["tracking_batch"] = track_batch
        if ragged:
            out["ragged_c2"] = ragged
        if ktimes:
            out["kernel_times"] = ktimes
            out["roofline_schur"] = ktimes["roofline_schur"]
            out["roofline"]["mfma_util_k_lin_imu"] = ktimes["k_lin_imu"]["mfma_util"]
        if replay_out:
            out["c3_replay"] = replay_out
        if replay30_out:
            out["c3_replay_keep30"] = replay30_out
            if isinstance(replay30_out, dict) and "error" not in replay30_out:
                out["c3_replay_keep30"]["policy_note"] = "keep-N is this repository's window policy for BASELINE C3 / C5 (30 frames at solve time), not a reference behaviour (the reference keeps 1 frame, trajectory.cpp:590-617)"
        if sharded:
            out["factor_sharded"] = sharded
        print(json.dumps(out))
        if sharded and sharded.get("self_check") and not sharded["self_check"]["passed"]:
            # the factor-sharded ranks must agree bit for bit and the process group must be the one asked for: a multi-GPU line that fails
            # its own check is not a measurement
            sys.stderr.write("bench.py: factor_sharded.self_check failed: %s\n" % json.dumps(sharded["self_check"]))
            if world > 1:
                dist.destroy_process_group()
            sys.exit(5)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
