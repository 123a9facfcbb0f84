"""The C-ABI library loads and exports every symbol include/liw_window.h declares; without a GPU every compute
entry point fails loudly (no CPU fallback); the host pre-integrators agree with the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(liw):
    hdr = open(os.path.join(ROOT, "include", "liw_window.h")).read()
    declared = sorted(set(re.findall(r"\b(liw_[A-Za-z_0-9]+)\s*\(", hdr)))
    L = liw.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(set(liw.EXPORTS)) == declared


def test_laser_header_symbols_are_exported(liw):
    hdr = open(os.path.join(ROOT, "include", "liw_laser.h")).read()
    declared = sorted(set(re.findall(r"\b(liw_(?:laser|scan)_[A-Za-z_0-9]+)\s*\(", hdr)))
    L = liw.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(set(liw.laser.LASER_EXPORTS)) == declared


def test_io_header_symbols_are_exported(liw):
    hdr = open(os.path.join(ROOT, "include", "liw_io.h")).read()
    declared = sorted(set(re.findall(r"\b(liw_(?:tum|record)_[A-Za-z_0-9]+)\s*\(", hdr)))
    L = liw.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(set(liw.outputs.IO_EXPORTS)) == declared


def test_posegraph_header_symbols_are_exported(liw):
    hdr = open(os.path.join(ROOT, "include", "liw_posegraph.h")).read()
    declared = sorted(set(re.findall(r"\b(liw_(?:posegraph|dense)_[A-Za-z_0-9]+)\s*\(", hdr)))
    L = liw.lib()
    assert not [s for s in declared if not hasattr(L, s)]
    assert sorted(set(liw.posegraph.PG_EXPORTS)) == declared
    lie = sorted(set(re.findall(r"\b(liw_lie_[A-Za-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "liw_lie.h")).read())))
    assert lie and not [s for s in lie if not hasattr(L, s)]


def test_no_cpu_fallback(liw, synth):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    prm = synth.office_params()
    d = synth.make_window(liw.HostPreint(prm), prm, seed=1, n=3, L=8)
    s = liw.Solver(prm)
    with pytest.raises(liw.LiwError) as e:
        s.set_window(liw.Window(d))
    assert e.value.code == liw.LIW_ENODEV


def test_host_preintegrators_match_oracle(liw, synth, pyoracle):
    prm = synth.office_params()
    a = synth.make_window(liw.HostPreint(prm), prm, seed=21, n=6, L=5)
    b = synth.make_window(pyoracle.Oracle(prm), prm, seed=21, n=6, L=5)
    for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
        assert np.abs(a[k] - b[k]).max() <= 1e-12 * max(1.0, np.abs(b[k]).max()), k
    # sqrt_inverse_P is upper triangular with U^T U = P^-1 (imu_preintegraption.h:149)
    U = a["imu_sqrtP"][0].reshape(15, 15)
    assert np.abs(np.tril(U, -1)).max() == 0.0


def test_extrinsics_reorthonormalised(liw, synth):
    s = liw.Solver(synth.office_params())
    Tw, Tl = s.extrinsics()
    for T in (Tw, Tl):
        assert np.abs(T[:3, :3] @ T[:3, :3].T - np.eye(3)).max() < 1e-6


def test_log_so3_all_quaternion_branches_host(liw, synth, pyoracle):
    """Large relative rotations between two odometry samples drive lie::log_SO3 through the trace <= 0 branches
    (arg-max diagonal i = 0, 1, 2) of the shared host/device geometry header; compared with the oracle."""
    prm = synth.office_params()
    orc = pyoracle.Oracle(prm)
    rng = np.random.default_rng(3)
    hit = set()
    for trial in range(60):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        if trial < 3:
            axis = np.eye(3)[trial]
        ang = rng.uniform(2.2, 3.1)
        R = synth.exp_so3(axis * ang)
        hit.add(int(np.argmax(np.diag(R))) if np.trace(R) <= 0 else -1)
        samples = np.zeros((2, 13))
        samples[0, 0], samples[1, 0] = 0.0, 0.1
        samples[0, 1:10] = np.eye(3).reshape(9)
        samples[1, 1:10] = R.reshape(9)
        samples[1, 10:13] = [0.1, 0.02, 0.0]
        Ta, Pa, Da = liw.HostPreint(prm).wheel_preint(samples, 0.0, 0.2)
        Tb, Pb, Db = orc.wheel_preint(samples, 0.0, 0.2)
        assert np.abs(Ta - Tb).max() < 1e-12 and np.abs(Pa - Pb).max() <= 1e-10 * np.abs(Pb).max()
    assert {0, 1, 2} <= hit
