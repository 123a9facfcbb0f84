"""The C++ multi-GPU sequence, compiled and run: tests/cpp/sharded_driver.cpp drives liw_batch_solve_sharded — the
chunked early-exit LM loop that lives in the C ABI since round 3 — with RCCL collectives (ncclAllReduce on the compact laser record, and
the one-shot ncclAllGather + rank-order sum).  World = 1: both exchanges must reproduce liw_batch_solve bit for bit and follow the oracle.
World = 2 on this 1-GPU box: two processes on device 0 — RCCL either runs (then both ranks must hold identical states, 1e-9 from the
un-sharded solve) or refuses the communicator, which the driver reports with exit code 3."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "sharded_driver.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "sharded_driver")


def build_driver(liw):
    libdir = os.path.dirname(liw.LIB_PATH)
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(liw.LIB_PATH)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                               SRC, "-o", EXE, "-L", libdir, "-lliw_window", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def dump_batch(path, wins):
    n = int(wins[0]["n"])
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", len(wins), n))
        for d in wins:
            f.write(struct.pack("<i", int(np.asarray(d["laser_frame"]).shape[0])))
            f.write(np.asarray(d["states"], dtype=np.float64).tobytes())
            f.write(np.asarray(d["laser_frame"], dtype=np.int32).tobytes())
            f.write(np.asarray(d["laser_pts"], dtype=np.float64).tobytes())
            f.write(np.asarray(d["match_pose"], dtype=np.float64).tobytes())
            f.write(np.asarray(d["has_match"], dtype=np.uint8).tobytes())
            for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
                f.write(np.asarray(d[k], dtype=np.float64).tobytes())


def read_out(path, B, n):
    raw = open(path, "rb").read()
    nv = struct.unpack("<i", raw[:4])[0]
    o, res = 4, {}
    for _ in range(nv):
        v = struct.unpack("<i", raw[o:o + 4])[0]
        o += 4
        x = np.frombuffer(raw[o:o + 8 * B * n * 15], dtype=np.float64).reshape(B, n, 15).copy()
        o += 8 * B * n * 15
        sm = np.frombuffer(raw[o:o + 12 * B], dtype=np.int32).reshape(B, 3).copy()
        o += 12 * B
        res[v] = (x, sm)
    return res


def test_cpp_sharded_driver_builds_and_needs_a_gpu(liw, synth, pyoracle, tmp_path):
    """CPU container: the C-ABI + RCCL program links; without a device it stops at hipSetDevice / liw_create (no CPU fallback)."""
    import torch
    exe = build_driver(liw)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests below")
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    dump_batch(str(tmp_path / "b.bin"), [synth.make_window(orc, prm, seed=1, n=3, L=20)])
    r = subprocess.run([exe, str(tmp_path / "b.bin"), str(tmp_path / "o.bin"), "0", "1", str(tmp_path / "id")], capture_output=True)
    assert r.returncode != 0 and not os.path.exists(str(tmp_path / "o.bin")) or os.path.getsize(str(tmp_path / "o.bin")) <= 4


@pytest.mark.gpu
def test_cpp_sharded_loop_world_1_is_bit_identical_to_the_plain_solve(liw, synth, pyoracle, tmp_path):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B = 7, 6
    wins = [synth.make_window(orc, prm, seed=8800 + k, n=n, L=60 + 45 * k) for k in range(B)]
    dump_batch(str(tmp_path / "b.bin"), wins)
    exe = build_driver(liw)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, str(tmp_path / "b.bin"), str(tmp_path / "o.bin"), "0", "1", str(tmp_path / "id"), "50"], capture_output=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    res = read_out(str(tmp_path / "o.bin"), B, n)
    assert sorted(res) == [0, 1, 2]
    for v in (1, 2):   # all-reduce and all-gather exchanges of one rank: the compact record round-trips the 128-slot record exactly
        assert np.array_equal(res[v][0], res[0][0]) and np.array_equal(res[v][1], res[0][1]), v
    orc.set_max_iterations(50)
    for k in range(B):
        wo = pyoracle.Window(wins[k])
        orc.set_prior(None)
        orc.init_solve(wo)
        so = orc.summary()
        assert (int(res[1][1][k, 0]), int(res[1][1][k, 1])) == (so["iterations"], so["termination"]), k
        xo = wo["states"].reshape(n, 15)
        assert np.abs(res[1][0][k] - xo).max() <= 1e-6 * np.abs(xo).max(), k


@pytest.mark.gpu
def test_cpp_sharded_loop_two_processes_on_one_gpu(liw, synth, pyoracle, tmp_path):
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    n, B = 6, 4
    wins = [synth.make_window(orc, prm, seed=8900 + k, n=n, L=120 + 40 * k) for k in range(B)]
    dump_batch(str(tmp_path / "b.bin"), wins)
    exe = build_driver(liw)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([exe, str(tmp_path / "b.bin"), str(tmp_path / ("o%d.bin" % r)), str(r), "2", str(tmp_path / "id"), "12"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for r in range(2)]
    outs = []
    for p in ps:
        try:
            outs.append(p.communicate(timeout=180))
        except subprocess.TimeoutExpired:
            for q in ps:
                q.kill()
            pytest.fail("two-rank RCCL run on one device hung")
    codes = [p.returncode for p in ps]
    if codes == [3, 3] or 3 in codes:
        msg = (outs[0][1] + outs[1][1]).decode()
        assert "refused" in msg or "unique id" in msg, msg[-500:]
        print("RCCL refuses two ranks on one device (graceful exit 3):", msg.strip().splitlines()[-1][:160])
        return
    assert codes == [0, 0], (codes, outs[0][1].decode()[-400:], outs[1][1].decode()[-400:])
    r0, r1 = read_out(str(tmp_path / "o0.bin"), B, n), read_out(str(tmp_path / "o1.bin"), B, n)
    for v in (1, 2):
        assert np.array_equal(r0[v][0], r1[v][0]) and np.array_equal(r0[v][1], r1[v][1]), "ranks must hold identical bits"
    assert np.array_equal(r0[1][0], r0[2][0]) or np.abs(r0[1][0] - r0[2][0]).max() <= 1e-9
    bs = liw.BatchSolver(prm, wins)
    bs.solve(liw.LIW_MODE_INIT, 12)
    assert np.abs(bs.states() - r0[2][0]).max() <= 1e-9 * np.abs(bs.states()).max()
