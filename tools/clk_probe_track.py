"""Phase stamps (s_memtime) of k_lm_step and k_marg_schur on the steady-state tracking window (n = 2).  Needs a stamp build:
LIW_CLK=1 LIW_CLK_IT=1 python -c 'import importlib; importlib.import_module("2dliw-slam_amd.build").build(force=True)'"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
prm = synth.office_params()
hp = liw.HostPreint(prm)
d3 = synth.make_window(hp, prm, seed=515, n=3, L=120, laser_on_frame0=False)


def sub(lo):
    o = dict(d3)
    o["n"] = 2
    for k in ("states", "match_pose"):
        o[k] = np.asarray(d3[k]).reshape(3, -1)[lo:lo + 2].copy()
    o["has_match"] = np.asarray(d3["has_match"])[lo:lo + 2].copy()
    for k in ("imu_X", "imu_J", "imu_sqrtP", "imu_Dt", "wheel_T", "wheel_sqrtP", "wheel_Dt"):
        o[k] = np.asarray(d3[k])[lo:lo + 1].copy()
    m = (np.asarray(d3["laser_frame"]) >= lo) & (np.asarray(d3["laser_frame"]) < lo + 2)
    o["laser_frame"] = (np.asarray(d3["laser_frame"])[m] - lo).astype(np.int32)
    o["laser_pts"] = np.asarray(d3["laser_pts"])[m].copy()
    return o


slv = liw.Solver(prm)
for rep in range(3):
    slv.set_prior(None)
    slv.set_window(liw.Window(sub(0)))
    slv.solve()
    slv.marginalization()
    slv.set_window(liw.Window(sub(1)))
    sg = slv.solve()
    slv.marginalization()
clk = np.zeros(8192, dtype=np.int64)
liw.lib().liw_debug_clk(clk.ctypes.data_as(C.c_void_p), C.c_int(8192))
d = lambda a, b: int(clk[b] - clk[a])
print("iterations", sg["iterations"])
print("k_lm_step: entry->cost %d | accept %d | barrier %d | to-sweep %d | sweep1 %d | term-check %d | barrier %d | sweep2 %d | tail %d | total %d"
      % (d(4000, 4001), d(4001, 4002), d(4002, 4003), d(4003, 0), d(0, 1), d(1, 4004), d(4004, 4005), d(4005, 3), d(3, 4006), d(4000, 4006)))
if clk[21] > clk[0] > 0:   # dense two-frame step (k_lm_step_dense2)
    print(" dense2: assemble frame 1 %d | frame 0 %d | entries + columns %d | chol30 %d | checks %d | solve + candidate %d" % (d(0, 20), d(20, 21), d(21, 22), d(22, 23), d(23, 4004), d(4005, 4006)))
for i in (1, 0):
    t = clk[10 + i * 8:10 + i * 8 + 6]
    print(" frame", i, "assemble", t[1] - t[0], "diag/gmax", t[2] - t[1], "colload", t[3] - t[2], "chol", t[4] - t[3], "ldsW", t[5] - t[4],
          "mfma+record+C", (clk[10 + (i - 1) * 8] if i else clk[1]) - t[5])
print("k_marg_schur: chain %d | eigen %d | tail %d | Jacobi sweeps %d" % (d(5000, 5001), d(5001, 5002), d(5002, 5003), int(clk[5010])))
if clk[5025] > clk[5020] > 0:   # one round of the four-wave Jacobi (thread 0, second sweep, fourth round)
    print(" Jacobi round: loads issued -> next round's indices %d | rotation %d | row rotation fetched %d | update + stores issued %d | barrier %d | (round = %d from the loads on)"
          % (d(5020, 5021), d(5021, 5022), d(5022, 5023), d(5023, 5024), d(5024, 5025), d(5020, 5025)))
# IMU role of the single-launch linearisation (k_lin_all: work-group n_laser + 0 is the IMU wave; LSTAMP picks gridDim / 2 = it at n = 2)
slv.linearize(liw.LIW_MODE_TRACK)
cl = np.zeros(512, dtype=np.int64)
liw.lib().liw_debug_clk_lin(cl.ctypes.data_as(C.c_void_p), C.c_int(512))
print("imu role (1 block): entry -> operand fetch issued %d | alpha/beta rows + dual exp %d | gamma rows (dE products, log_SO3) %d | sync %d | entry codes %d | MFMA + stores %d | total %d"
      % (cl[300] - cl[299], cl[302] - cl[300], cl[303] - cl[302], cl[304] - cl[303], cl[305] - cl[304], cl[311] - cl[305], cl[311] - cl[299]))
print("k_lin_all role waves (cycles, start-to-end; n = 2: laser f0, laser f1, imu, wheel, ground):", [int(cl[401 + 2 * v] - cl[400 + 2 * v]) for v in range(5)],
      "kernel span", int(max(cl[401 + 2 * v] for v in range(5)) - min(cl[400 + 2 * v] for v in range(5))))
