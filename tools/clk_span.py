"""Per-window start / end stamps of k_lm_step (LIW_CLK build): residency rounds, per-wave duration spread, effective clock."""
import importlib, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
liw = importlib.import_module('2dliw-slam_amd'); synth = importlib.import_module('2dliw-slam_amd.synth')
prm = synth.office_params()
hp = liw.HostPreint(prm)
w = [synth.make_window(hp, prm, seed=20240 + k, n=30, L=2000) for k in range(2)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bs = liw.BatchSolver(prm, [w[k % 2] for k in range(B)])
bs.solve(liw.LIW_MODE_INIT, 6); torch.cuda.synchronize()
sp = np.zeros(3 * 16384, dtype=np.int64)
liw.lib().liw_debug_span(sp.ctypes.data_as(C.c_void_p), C.c_int(3 * 16384))
sp = sp.reshape(-1, 3)[:B]
t0 = sp[:, 0].min()
st, en = sp[:, 0] - t0, sp[:, 1] - t0
dur = en - st
print('B', B, 'kernel span (cycles)', en.max(), 'dur min/med/max', dur.min(), int(np.median(dur)), dur.max())
late = st > 0.3 * en.max()
print('first-round waves', int((~late).sum()), 'dur med', int(np.median(dur[~late])), '| later waves', int(late.sum()), 'dur med', int(np.median(dur[late])) if late.any() else 0)
h, e = np.histogram(st, bins=10, range=(0, en.max()))
print('start histogram', h.tolist())
h, e = np.histogram(en, bins=10, range=(0, en.max()))
print('end histogram  ', h.tolist())
hw = sp[:, 2]
print('hw id sample', [hex(int(v)) for v in hw[:4]])
