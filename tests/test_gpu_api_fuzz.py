"""Differential fuzz of the single-window C ABI state machine: random sequences of liw_set_window / liw_solve (INIT, TRACK) /
liw_marginalize / liw_set_prior / liw_clear_window / re-attachment by content / liw_linearize / liw_get_history on two contexts —
one with the marginalisation enqueued behind tracking solves (default), one without (LIW_NO_SPEC_MARG, read at liw_create).  Both
run the same kernels on the same inputs, so every output must agree bit for bit and every error must occur on both sides."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(liw, prm, spec, monkeypatch):
    if spec:
        monkeypatch.delenv("LIW_NO_SPEC_MARG", raising=False)
    else:
        monkeypatch.setenv("LIW_NO_SPEC_MARG", "1")
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
    slv = [_mk(liw, prm, True, monkeypatch), _mk(liw, prm, False, monkeypatch)]
    win = [None, None]

    def both(f):
        r = [_call(lambda s=s, i=i: f(s, i)) for i, s in enumerate(slv)]
        assert r[0][0] == r[1][0], (r[0], r[1])
        assert _same(r[0][1], r[1][1]), (r[0], r[1])
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
                assert np.array_equal(win[0]["states"], win[1]["states"]) and np.array_equal(win[0]["match_pose"], win[1]["match_pose"])
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
            win = [None, None]
        elif op == "lin" and win[0] is not None:
            both(lambda s, i: s.linearize(liw.LIW_MODE_TRACK))
        elif op == "hist":
            both(lambda s, i: s.history())
    for s in slv:
        s.close()
