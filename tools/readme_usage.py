"""The 'usage in five lines' snippet of README.md, runnable as is on an MI355X box."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
liw   = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
prm = synth.office_params()
win = synth.make_window(liw.HostPreint(prm), prm, n=30, L=2000)
slv = liw.Solver(prm); slv.set_window(liw.Window(win))
print(slv.init_solve(), slv.marginalization()["sqrt_H"].shape)
bs = liw.BatchSolver(prm, [win] * 1024); bs.solve(liw.LIW_MODE_INIT); bs.marginalize()
print("batch ok", bs.summaries()[0])
