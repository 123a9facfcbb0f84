"""The reference's steady state as a batch: B robots tracking in lock-step, per laser frame one solver::solve on the
2-frame window (laser blocks of the new frame, one IMU + one wheel block, ground blocks, the carried prior on the older frame) and one
solver::marginalization whose result is the next frame's prior (src/trajectory/trajectory.cpp:525-560, src/factor/solver.cpp:631-820,
:257-442, :390-402).  bench.py's `tracking_batch` leg at a test-size batch (4 421 robots: the large-batch record format, k_lin_imu_chain,
k_lm_step_quad in the TRACK topology, k_marg_schur_chain + k_marg_schur_eigq; not a multiple of 64 or 4), four consecutive frames:

  * teacher-forced: every frame of the sampled robots re-solved by the oracle from the inputs the GPU batch had — states within 1e-6, equal
    iteration counts and terminations, the marginalisation (Delta_H, Delta_g, the new prior's J^T J / J^T R) at the GPU's solved states;
  * free-running: the oracle carries its OWN states and prior through the same frames — tracking solves converge in 3 - 6 iterations, so the
    two chains must still agree to 1e-6 after the last frame;
  * a second pass over the same frames is bit-identical (nothing of a previous frame leaks through the workspace)."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_batched_tracking_frames_teacher_forced_and_free_running(liw, synth, pyoracle):
    import ctypes as C
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    prm = synth.office_params()
    B, K, nb = 4421, 4, 8
    tb = bench.TrackBatch(liw, synth, prm, B, K, nb, "cuda:0", seed0=61240)
    ids = [0, 1, 2, 63, 64, B - 65, B - 1, nb + 3]
    _, its, cap = tb.run(capture_ids=ids)
    flags = C.c_int(0)
    assert tb.bs.L.liw_batch_launch_paths(tb.bs.h, C.byref(tb.bs.b), tb.bs._wsp(), C.byref(flags)) == 0
    assert flags.value & 1                                        # the large-batch record format: chain / quad / chain + eigq kernels
    assert all(int(r["has_out"].min()) == 1 for r in cap)         # every frame leaves a prior
    assert int(cap[0]["has_in"].max()) == 0 and all(int(r["has_in"].min()) == 1 for r in cap[1:])
    for k in range(1, K + 1):                                     # the carry: frame k's older frame is frame k-1's solved new frame
        assert np.array_equal(cap[k]["x_in"][:, 0], cap[k - 1]["x_out"][:, 1])
        assert np.array_equal(cap[k]["pX_in"], cap[k - 1]["pX_out"]) and np.array_equal(cap[k]["pJ_in"], cap[k - 1]["pJ_out"])
        assert np.array_equal(cap[k]["pX_in"], cap[k]["x_in"][:, 0])            # linearized_X = the frame the prior sits on
    par = tb.teacher_forced_parity(ids, cap)
    print("batched tracking, teacher-forced:", par)
    assert par["frames"] == K * len(ids)
    assert par["within_1e_6"] == par["frames"] and par["iterations_equal"] == par["frames"] and par["terminations_equal"] == par["frames"], par
    assert par["worst_rel_Delta_H"] <= 1e-11 and par["worst_Delta_g_of_roundoff_scale"] <= 1e-10, par
    assert par["worst_rel_prior_JtJ"] <= 1e-10 and par["worst_prior_JtR_of_roundoff_scale"] <= 1e-9, par
    itk = np.stack(its[1:])
    assert 2 <= itk.mean() <= 12 and itk.max() < 50, (itk.mean(), itk.max())
    # free-running oracle chains for the un-jittered robots
    orc = pyoracle.Oracle(prm)
    worst = 0.0
    for j, b in enumerate(ids[:3]):
        orc.set_prior(None)
        prev = None
        for k in range(K + 1):
            w = dict(tb.window(k, b))
            if prev is not None:
                w["states"] = np.array(w["states"], copy=True)
                w["match_pose"] = np.array(w["match_pose"], copy=True)
                w["states"][0] = prev[0]
                w["match_pose"][0, 6:12] = prev[1]
            wo = pyoracle.Window(w)
            orc.solve(wo)
            so = orc.summary()
            orc.marginalization(wo)
            prev = (wo["states"].reshape(2, 15)[1].copy(), wo["match_pose"].reshape(2, 12)[1, 6:12].copy())
            e = float(np.abs(cap[k]["x_out"][j] - wo["states"].reshape(2, 15)).max() / np.abs(wo["states"]).max())
            worst = max(worst, e)
            assert e <= 1e-6, (b, k, e)
            assert cap[k]["summ"][j]["iterations"] == so["iterations"] and cap[k]["summ"][j]["termination"] == so["termination"], (b, k)
    print("batched tracking, free-running oracle chains of 3 robots over %d frames: worst state error %.2e" % (K + 1, worst))
    # a second pass: bit-identical
    _, _, cap2 = tb.run(capture_ids=ids)
    for a, b_ in zip(cap, cap2):
        for key in ("x_out", "mp_out", "dH", "dg", "pJ_out", "pR_out"):
            assert np.array_equal(a[key], b_[key]), key
    tb.bs.close()
