"""GPU parity of the batched pre-integration kernels (liw_batch_imu_preint / liw_batch_wheel_preint, SURVEY §8 rows
a5 / a6 in their batch-replay form) against the oracle's sequential accumulators, through the C-ABI.

Tolerances: X, J, delta_Tij, Dt 1e-12 relative (same arithmetic, different summation order in F P F^T);
sqrt_inverse_P 1e-8 relative (inverse + Cholesky of a covariance with condition number ~1e8)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Recorder:
    """preint provider for synth.make_window that records every interval and delegates to the oracle."""

    def __init__(self, inner):
        self.inner, self.imu, self.wheel, self.imu_out, self.wheel_out = inner, [], [], [], []

    def imu_preint(self, samples, t_start, t_end, bias6):
        r = self.inner.imu_preint(samples, t_start, t_end, bias6)
        self.imu.append((np.array(samples), float(t_start), float(t_end), np.array(bias6)))
        self.imu_out.append(r)
        return r

    def wheel_preint(self, samples, t_start, t_end):
        r = self.inner.wheel_preint(samples, t_start, t_end)
        self.wheel.append((np.array(samples), float(t_start), float(t_end)))
        self.wheel_out.append(r)
        return r


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


@pytest.fixture(scope="module")
def recorded(liw, synth, pyoracle):
    prm = synth.office_params()
    rec = Recorder(pyoracle.Oracle(prm))
    for k in range(3):
        synth.make_window(rec, prm, seed=900 + k, n=9, L=0)
    # ragged intervals: different sample counts inside one wave (1, 2, 5, 41 samples), a long one (2 s at 200 Hz)
    rng = np.random.default_rng(7)
    for cnt, span in ((1, 0.004), (2, 0.011), (5, 0.03), (41, 0.2), (400, 2.0)):
        t = 3.0 + np.sort(rng.uniform(0.0, span, cnt))
        s = np.zeros((cnt, 7))
        s[:, 0] = t
        s[:, 1:4] = rng.normal(0.0, 1.0, (cnt, 3)) + np.array([0.0, 0.0, 9.8])
        s[:, 4:7] = rng.normal(0.0, 0.5, (cnt, 3))
        rec.imu_preint(s, t[0] + 0.001, t[-1] + 0.002, rng.normal(0.0, 1e-2, 6))
    return prm, rec


def test_batch_imu_preint_matches_oracle(liw, recorded):
    prm, rec = recorded
    bp = liw.BatchPreint(prm)
    X, J, S, Dt = [t.cpu().numpy() for t in bp.imu(rec.imu)]
    assert len(rec.imu) == 3 * 8 + 5
    for m, (Xo, Jo, So, Dto) in enumerate(rec.imu_out):
        assert abs(Dt[m] - Dto) <= 1e-12 * max(1.0, abs(Dto)), m
        assert relerr(X[m], Xo) <= 1e-12, (m, relerr(X[m], Xo))
        assert relerr(J[m].reshape(-1), np.asarray(Jo).reshape(-1)) <= 1e-12, m
        assert relerr(S[m].reshape(-1), np.asarray(So).reshape(-1)) <= 1e-8, (m, relerr(S[m].reshape(-1), np.asarray(So).reshape(-1)))
        # U^T U = P^-1 and U is upper triangular (imu_preintegraption.h:149)
        U = S[m]
        assert np.abs(np.tril(U, -1)).max() == 0.0
        P = bp.last_P[m].cpu().numpy()
        assert np.abs(U.T @ U @ P - np.eye(15)).max() <= 1e-6, m


def test_batch_imu_preint_matches_host_accumulator(liw, recorded):
    prm, rec = recorded
    host = liw.HostPreint(prm)
    bp = liw.BatchPreint(prm)
    X, J, S, Dt = [t.cpu().numpy() for t in bp.imu(rec.imu[:10])]
    for m, iv in enumerate(rec.imu[:10]):
        Xh, Jh, Sh, Dth = host.imu_preint(*iv)
        assert relerr(X[m], Xh) <= 1e-13 and relerr(J[m].reshape(-1), np.asarray(Jh).reshape(-1)) <= 1e-13, m
        assert relerr(S[m].reshape(-1), np.asarray(Sh).reshape(-1)) <= 1e-8, m


def test_batch_wheel_preint_matches_oracle(liw, recorded):
    prm, rec = recorded
    bp = liw.BatchPreint(prm)
    ivs = list(rec.wheel)
    outs = list(rec.wheel_out)
    # extra cases: no sample after t_start (reset at the end), a single seed sample, samples closer than 50 ms
    base = rec.wheel[0][0]
    extra = [(base[:2].copy(), float(base[1, 0] + 0.01), float(base[1, 0] + 0.08)),
             (base[:1].copy(), float(base[0, 0] + 0.01), float(base[0, 0] + 0.05))]
    dense = base.copy()
    dense[:, 0] = base[0, 0] + 0.02 * np.arange(len(base))
    extra.append((dense, float(dense[1, 0] + 0.001), float(dense[-1, 0] + 0.01)))
    for iv in extra:
        ivs.append(iv)
        outs.append(rec.inner.wheel_preint(*iv))
    T, S, Dt = [t.cpu().numpy() for t in bp.wheel(ivs)]
    for m, (To, So, Dto) in enumerate(outs):
        assert abs(Dt[m] - Dto) <= 1e-12 * max(1.0, abs(Dto)), m
        assert np.abs(T[m] - np.asarray(To)).max() <= 1e-12, (m, np.abs(T[m] - np.asarray(To)).max())
        assert relerr(S[m].reshape(-1), np.asarray(So).reshape(-1)) <= 1e-10, m


def test_batch_preint_feeds_the_solver(liw, synth, pyoracle, recorded):
    """A window whose IMU / wheel blocks come from the device pre-integration solves to the same states as the
    window built with the oracle's accumulators (1e-6 relative, the north-star tolerance)."""
    prm, _ = recorded
    rec = Recorder(pyoracle.Oracle(prm))
    w = synth.make_window(rec, prm, seed=31, n=8, L=200)
    bp = liw.BatchPreint(prm)
    X, J, S, Dt = [t.cpu().numpy() for t in bp.imu(rec.imu)]
    T, Sw, Dtw = [t.cpu().numpy() for t in bp.wheel(rec.wheel)]
    w2 = dict(w)
    w2.update(imu_X=X, imu_J=J.reshape(-1, 225), imu_sqrtP=S.reshape(-1, 225), imu_Dt=Dt, wheel_T=T, wheel_sqrtP=Sw.reshape(-1, 9), wheel_Dt=Dtw)
    out = []
    for ww in (w, w2):
        bs = liw.BatchSolver(prm, [ww])
        bs.solve(liw.LIW_MODE_INIT, 50)
        out.append(bs.states()[0])
    assert np.abs(out[0] - out[1]).max() <= 1e-6 * max(1.0, np.abs(out[0]).max())
