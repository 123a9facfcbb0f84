"""s_memtime stamps of one k_lin_laser_slab wave (LIW_CLK=1 build): dependent index loads, transform records, block loop, record epilogue."""
import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
liw = importlib.import_module('2dliw-slam_amd'); synth = importlib.import_module('2dliw-slam_amd.synth')
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
prm = synth.office_params()
wins = bench.make_batch(liw, synth, prm, B, 30, 2000, seed0=20240, n_base=64)
bs = liw.BatchSolver(prm, wins)
M = liw.LIW_MODE_INIT
kt = bs.time_kernels(M, 2)
torch.cuda.synchronize()
c = np.zeros(16, dtype=np.int64)
liw.lib().liw_debug_clk_slab(c.ctypes.data_as(C.c_void_p), C.c_int(16))
print('k_lin_laser alone %.3f ms; one wave (cycles): index loads %d | transform records %d | block loop %d | record epilogue %d | total %d'
      % (kt['k_lin_laser'], c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], c[4] - c[0]))
