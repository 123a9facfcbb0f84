"""The batched steady state from C++ (INTEGRATION.md "Batched tracking"): tests/cpp/track_batch_driver.cpp drives
liw_batch_solve(TRACK) -> liw_batch_marg_linearize -> liw_batch_marg_schur frame after frame on hipMalloc'd arrays, carrying the solved
frame and the prior — no Python, no torch on that side.  It must reproduce bench.py's TrackBatch (the Python mirror over the same C ABI)
BIT FOR BIT: states, laser_match poses, iteration counts / terminations, Delta_H, Delta_g and the prior after every frame
(reference call pattern: src/trajectory/trajectory.cpp:525-560)."""
import importlib
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "track_batch_driver.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "track_batch_driver")


def build_driver(liw):
    libdir = os.path.dirname(liw.LIB_PATH)
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(liw.LIB_PATH)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                               SRC, "-o", EXE, "-L", libdir, "-lliw_window", "-L/opt/rocm/lib", "-lamdhip64",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_cpp_track_batch_driver_builds_and_needs_a_gpu(liw, tmp_path):
    """CPU container: the program links against the C ABI; without a device it stops at hipSetDevice / liw_create (no CPU fallback)."""
    import torch
    exe = build_driver(liw)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test below")
    r = subprocess.run([exe, str(tmp_path / "o.bin"), str(tmp_path / "missing.bin")], capture_output=True)
    assert r.returncode != 0


@pytest.mark.gpu
def test_cpp_track_batch_driver_reproduces_the_python_mirror_bit_for_bit(liw, synth, pyoracle, tmp_path):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from test_gpu_cpp_sharded import dump_batch
    prm = synth.office_params()
    B, K, nb = 1101, 3, 4                                        # 1 101 robots: the large-batch record format, 16-window IMU waves, a ragged last wave
    tb = bench.TrackBatch(liw, synth, prm, B, K, nb, "cuda:0", seed0=62240)
    ids = list(range(B))
    _, its, cap = tb.run(capture_ids=ids)
    # the frames as the C++ host gets them: every robot's window with ITS initial states / poses (the older frame of frames >= 1 is overwritten by the carry)
    paths = []
    for k in range(K + 1):
        x0 = tb.x0[k].cpu().numpy().reshape(B, 2, 15)
        mp0 = tb.mp0[k].cpu().numpy().reshape(B, 2, 12)
        wins = []
        for b in range(B):
            w = dict(tb.window(k, b))
            w["states"], w["match_pose"] = x0[b], mp0[b]
            wins.append(w)
        p = str(tmp_path / ("frame%d.bin" % k))
        dump_batch(p, wins)
        paths.append(p)
    out = str(tmp_path / "out.bin")
    r = subprocess.run([build_driver(liw), out] + paths, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    raw = open(out, "rb").read()
    assert struct.unpack("<i", raw[:4])[0] == K + 1
    o = 4

    def take(dt, cnt):
        nonlocal o
        a = np.frombuffer(raw, dtype=dt, count=cnt, offset=o).copy()
        o += a.nbytes
        return a
    for k in range(K + 1):
        x = take(np.float64, B * 30).reshape(B, 2, 15)
        mp = take(np.float64, B * 24).reshape(B, 2, 12)
        sm = take(np.int32, B * 3).reshape(B, 3)
        dH = take(np.float64, B * 225).reshape(B, 15, 15)
        dg = take(np.float64, B * 15).reshape(B, 15)
        pX, pJ, pR, pH = take(np.float64, B * 15).reshape(B, 15), take(np.float64, B * 225).reshape(B, 15, 15), take(np.float64, B * 15).reshape(B, 15), take(np.int32, B)
        c = cap[k]
        assert np.array_equal(x, c["x_out"]) and np.array_equal(mp, c["mp_out"]), k
        assert [tuple(r_) for r_ in sm] == [(s["iterations"], s["termination"], s["successful"]) for s in c["summ"]], k
        assert np.array_equal(dH, c["dH"]) and np.array_equal(dg, c["dg"]), k
        assert np.array_equal(pX, c["pX_out"]) and np.array_equal(pJ, c["pJ_out"]) and np.array_equal(pR, c["pR_out"]) and np.array_equal(pH, c["has_out"]), k
    assert o == len(raw)
    assert len({int(v) for v in np.stack(its[1:]).reshape(-1)}) > 1           # robots finish at different iterations
    tb.bs.close()
