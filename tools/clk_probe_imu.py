"""s_memtime stamps of one k_lin_imu wave (LIW_CLK=1 build, middle block of the grid): dual-number part, matrix-core groups."""
import importlib, sys, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
liw = importlib.import_module('2dliw-slam_amd'); synth = importlib.import_module('2dliw-slam_amd.synth')
prm = synth.office_params()
hp = liw.HostPreint(prm)
w = [synth.make_window(hp, prm, seed=20240 + k, n=30, L=2000) for k in range(2)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
bs = liw.BatchSolver(prm, [w[k % 2] for k in range(B)])
bs.solve(liw.LIW_MODE_INIT, 3)      # (the linearisations of a solve read the packed IMU block records)
torch.cuda.synchronize()
clk = np.zeros(512, dtype=np.int64)
liw.lib().liw_debug_clk_lin(clk.ctypes.data_as(C.c_void_p), C.c_int(512))
c = clk
print('imu wave: alpha/beta rows', c[302] - c[300], ' gamma rows', c[303] - c[302], ' sync + codes + first sqrt_info group', c[305] - c[303])
g = [int(c[305 + k]) for k in range(7) if c[305 + k] > 0]
print('   matrix-core groups (k_lin_imu: 7 blocks each; k_lin_imu_chain: 4):', [g[k + 1] - g[k] for k in range(len(g) - 1)], ' last group + tail', int(c[311]) - g[-1], ' total', int(c[311] - c[300]))
