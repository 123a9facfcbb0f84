"""Per-step busy / barrier-wait cycles of the four waves of k_lm_step_tw (eliminators 0 / 1, producers 2 / 3) on one C2 window.
Needs a stamp build (LIW_CLK=1).  usage: python tools/clk_probe_tw.py [mode: init|track]"""
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
d = synth.make_window(hp, prm, seed=20240, n=30, L=2000)
slv = liw.Solver(prm)
slv.set_window(liw.Window(d))
slv.init_solve(6)
clk = np.zeros(8192, dtype=np.int64)
liw.lib().liw_debug_clk(clk.ctypes.data_as(C.c_void_p), C.c_int(8192))
for w, name in enumerate(("eliminator down", "eliminator up", "producer down", "producer up")):
    busy = clk[7000 + 100 * w:7000 + 100 * w + 40:2]
    wait = clk[7001 + 100 * w:7001 + 100 * w + 40:2]
    k = int((busy + wait > 0).sum())
    print("%-16s busy %s" % (name, [int(v) for v in busy[:k]]))
    print("%-16s wait %s  | sums: busy %d wait %d" % ("", [int(v) for v in wait[:k]], busy[:k].sum(), wait[:k].sum()))
