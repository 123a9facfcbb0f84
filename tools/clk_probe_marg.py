"""Phase stamps (s_memtime) of the one-wave marginalisation kernel k_marg_schur on a C2 window (LIW_CLK=1 build; LIW_MARG_WAVES=1 forces the
one-wave kernel on a small batch): chain Schur complement | Jacobi eigen-decomposition | eigen square root + prior write-back."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LIW_MARG_WAVES", "1")
os.environ.setdefault("LIW_MARG_EIG", "1")   # the stamps are in k_marg_schur (chain + eigen by one wave), not in the two kernels large batches run since the end of round 5
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
prm = synth.office_params()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w = synth.make_window(liw.HostPreint(prm), prm, seed=20240, n=n, L=2000)
bs = liw.BatchSolver(prm, [w] * 8)
bs.solve(liw.LIW_MODE_INIT, 5)
for _ in range(3):
    bs.marginalize()
import torch
torch.cuda.synchronize()
clk = np.zeros(8192, dtype=np.int64)
liw.lib().liw_debug_clk(clk.ctypes.data_as(C.c_void_p), C.c_int(8192))
d = lambda a, b: int(clk[b] - clk[a])
print("k_marg_schur (one wave, n = %d): chain %d (%d per step) | eigen %d | tail %d cycles" % (n, d(5000, 5001), d(5000, 5001) // max(n - 1, 1), d(5001, 5002), d(5002, 5003)))
