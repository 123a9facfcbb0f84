"""s_memtime stamps of the two-wave step kernel (k_lm_step_tw, LIW_CLK=1 build): per wave, per step phases of LM iteration 3."""
import importlib, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
liw = importlib.import_module('2dliw-slam_amd'); synth = importlib.import_module('2dliw-slam_amd.synth')
prm = synth.office_params(); hp = liw.HostPreint(prm)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w = [synth.make_window(hp, prm, seed=20240, n=n, L=2000)]
bs = liw.BatchSolver(prm, w)
bs.solve(liw.LIW_MODE_INIT, 6); torch.cuda.synchronize()
clk = np.zeros(8192, dtype=np.int64)
liw.lib().liw_debug_clk(clk.ctypes.data_as(C.c_void_p), C.c_int(8192))
t0 = clk[4000]
for wv in (0, 1):
    c = clk[4000 + 1000 * wv:5000 + 1000 * wv]
    print('wave', wv, 'prologue+barrier', c[1] - t0, 'sweep', c[2] - c[1], 'to backsub', c[3] - c[2], 'backsub', c[4] - c[3], 'total', c[4] - t0)
    m = (n + 1) // 2
    ns = (n - 1 - m + 2) if wv == 0 else m
    for s in list(range(min(ns, 3))) + [ns - 2, ns - 1]:
        t = c[10 + s * 8:10 + s * 8 + 7]
        nxt = c[10 + (s + 1) * 8] if s + 1 < ns else c[2]
        print('  step', s, 'barrier', t[1] - t[0], 'commit', t[2] - t[1], 'diag', t[3] - t[2], 'col', t[4] - t[3], 'issue+chol', t[5] - t[4], 'tiles+rec', t[6] - t[5], 'schur', nxt - t[6])
