"""Throughput of the batched pre-integration kernels (k_preint_imu + k_preint_imu_sqrt, k_preint_wheel) on M intervals
of the C2 shape (0.1 s at 200 Hz = 20 IMU samples, ~5 odometry samples), next to the host accumulators."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")


def main():
    import torch
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 29
    prm = synth.office_params()
    rng = np.random.default_rng(3)
    base_i, base_w = [], []
    for k in range(64):
        t = 1.0 + np.arange(20) / 200.0
        s = np.zeros((20, 7)); s[:, 0] = t
        s[:, 1:4] = rng.normal(0, 0.3, (20, 3)) + [0, 0, 9.8]; s[:, 4:7] = rng.normal(0, 0.2, (20, 3))
        base_i.append((s, 1.0, 1.1, rng.normal(0, 1e-3, 6)))
        tw = np.arange(1.0 - 0.12, 1.1, 0.0505)
        w = np.zeros((len(tw), 13)); w[:, 0] = tw
        for i, tt in enumerate(tw):
            w[i, 1:10] = synth.exp_so3(np.array([0, 0, 0.3 * tt])).reshape(9); w[i, 10:13] = [0.5 * tt, 0.01 * k * tt, 0]
        base_w.append((w, 1.0, 1.1))
    ivs_i = [base_i[m % 64] for m in range(M)]
    ivs_w = [base_w[m % 64] for m in range(M)]
    bp = liw.BatchPreint(prm)
    # pack once, time only the kernels
    Mi, off, smp, ts, te = bp._pack(ivs_i, 7)
    bias = torch.from_numpy(np.ascontiguousarray(np.array([iv[3] for iv in ivs_i]).reshape(-1))).cuda()
    z = lambda *sh: torch.zeros(sh, dtype=torch.float64, device="cuda")
    X, J, P, S, Dt = z(M, 15), z(M, 225), z(M, 225), z(M, 225), z(M)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run_i():
        bp._chk(bp.L.liw_batch_imu_preint(bp.h, C.c_int(M), p(off), p(smp), p(ts), p(te), p(bias), p(X), p(J), p(P), p(S), p(Dt), st))
    Mw, offw, smpw, tsw, tew = bp._pack(ivs_w, 13)
    T, Sw, Dtw = z(M, 12), z(M, 9), z(M)
    def run_w():
        bp._chk(bp.L.liw_batch_wheel_preint(bp.h, C.c_int(M), p(offw), p(smpw), p(tsw), p(tew), p(T), p(Sw), p(Dtw), st))
    out = {"intervals": M}
    for name, fn in (("imu", run_i), ("wheel", run_w)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[name + "_ms"] = ms
        out[name + "_intervals_per_s"] = M / ms * 1e3
    host = liw.HostPreint(prm)
    t0 = time.perf_counter()
    for iv in base_i:
        host.imu_preint(*iv)
    out["host_imu_intervals_per_s_python_driven"] = 64 / (time.perf_counter() - t0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
