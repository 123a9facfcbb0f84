"""Differential fuzz of the single-window C ABI state machine: random sequences of liw_set_window / liw_solve (INIT, TRACK) /
liw_marginalize / liw_set_prior / liw_clear_window / re-attachment by content / liw_linearize / liw_get_history on two contexts —
the default one (marginalisation enqueued behind tracking solves, re-attachment by content), one without the speculative
marginalisation (LIW_NO_SPEC_MARG, read at liw_create), one with neither that nor the re-attachment (LIW_NO_REATTACH: every
liw_set_window uploads; its history calls are not compared, a new upload drops the history by design).  All run the same kernels on
the same inputs, so every output must agree bit for bit and every error must occur on every side."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(liw, prm, spec, reattach, monkeypatch):
    for name, on in (("LIW_NO_SPEC_MARG", not spec), ("LIW_NO_REATTACH", not reattach)):
        if on:
            monkeypatch.setenv(name, "1")
        else:
            monkeypatch.delenv(name, raising=False)
    return liw.Solver(prm)


def _call(fn):
    try:
        return ("ok", fn())
    except Exception as e:   # noqa: BLE001  (LiwError codes are compared)
        return ("err", getattr(e, "code", repr(e)))


def _same(a, b):
    if type(a) is not type(b):
        return False
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    if isinstance(a, float):
        return a == b or (a != a and b != b)
    return a == b


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_call_sequences_speculative_vs_plain(liw, synth, monkeypatch, seed):
    prm = synth.office_params()
    hp = liw.HostPreint(prm)
    rng = np.random.default_rng(4100 + seed)
    pool = [synth.make_window(hp, prm, seed=500 + 10 * seed + k, n=int(rng.integers(2, 5)), L=int(rng.integers(0, 120)), laser_on_frame0=False)
            for k in range(4)]
    slv = [_mk(liw, prm, True, True, monkeypatch), _mk(liw, prm, False, True, monkeypatch), _mk(liw, prm, False, False, monkeypatch)]
    win = [None, None, None]

    def both(f, upto=3):
        r = [_call(lambda s=s, i=i: f(s, i)) for i, s in enumerate(slv)]
        for k in range(1, upto):
            assert r[0][0] == r[k][0], (k, r[0], r[k], log[-8:])
            assert _same(r[0][1], r[k][1]), (k, r[0], r[k], log[-8:])
        return r[0]

    ops = ["set", "set", "reattach", "track", "track", "track", "init", "marg", "marg", "marg", "prior_none", "prior_copy", "clear", "lin", "hist", "edit"]
    log = []
    for step in range(60):
        op = ops[int(rng.integers(0, len(ops)))]
        log.append(op)
        if op == "set":
            k = int(rng.integers(0, len(pool)))

            def f(s, i, k=k):
                win[i] = liw.Window(pool[k])
                s.set_window(win[i])
            both(f)
        elif op == "reattach" and win[0] is not None:
            def f(s, i):
                d = {k: win[i][k].copy() for k in win[i].a}
                d["n"] = win[i].n
                win[i] = liw.Window(d)
                s.set_window(win[i])
            both(f)
        elif op == "edit" and win[0] is not None:        # same window, newest frame moved: must NOT re-attach
            dx = float(rng.normal(0.0, 1e-3))

            def f(s, i, dx=dx):
                d = {k: win[i][k].copy() for k in win[i].a}
                d["n"] = win[i].n
                d["states"].reshape(-1)[-15] += dx
                win[i] = liw.Window(d)
                s.set_window(win[i])
            both(f)
        elif op in ("track", "init"):
            cap = int(rng.choice([1, 3, 6, 50]))
            r = both(lambda s, i: (s.solve(cap) if op == "track" else s.init_solve(cap)))
            if r[0] == "ok":
                for k in (1, 2):
                    assert np.array_equal(win[0]["states"], win[k]["states"]) and np.array_equal(win[0]["match_pose"], win[k]["match_pose"])
        elif op == "marg":
            both(lambda s, i: s.marginalization())
            both(lambda s, i: s.get_prior())
        elif op == "prior_none":
            both(lambda s, i: s.set_prior(None))
        elif op == "prior_copy":
            def f(s, i):
                p = s.get_prior()
                if p is not None:
                    s.set_prior((p[0] + 1e-4, p[1], p[2]))
                return p is None
            both(f)
        elif op == "clear":
            both(lambda s, i: s.L.liw_clear_window(s.h))
            win = [None, None, None]
        elif op == "lin" and win[0] is not None:
            both(lambda s, i: s.linearize(liw.LIW_MODE_TRACK))
        elif op == "hist":
            both(lambda s, i: s.history(), upto=2)
    for s in slv:
        s.close()
