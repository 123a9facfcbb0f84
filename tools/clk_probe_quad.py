"""Phase stamps of one wave of k_lm_step_quad at one frame of one LM iteration (s_memtime; run on the GPU box):
python tools/clk_probe_quad.py [batch] [block] [frame] [iteration]"""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
blk = int(sys.argv[2]) if len(sys.argv) > 2 else 100
frame = int(sys.argv[3]) if len(sys.argv) > 3 else 15
it = int(sys.argv[4]) if len(sys.argv) > 4 else 3
prm = synth.office_params()
base = [synth.make_window(liw.HostPreint(prm), prm, seed=20240 + k, n=30, L=2000) for k in range(4)]
os.environ["LIW_STEP_VARIANT"] = "3"
bs = liw.BatchSolver(prm, [base[b % 4] for b in range(B)])
L = liw.lib()
out = (C.c_longlong * 64)()
L.liw_debug_quad_clk(1, blk, frame, it, None)
bs.solve(liw.LIW_MODE_INIT, 6)
import torch
torch.cuda.synchronize()
L.liw_debug_quad_clk(0, 0, 0, 0, out)
t = np.array(out[:16], dtype=np.int64)
names = ["frame start", "records landed (vmcnt)", "LDS read, pose blocks added", "(same)", "grad norm + LM diagonal", "scaled", "fold + prefetch issued", "eliminated", "Schur", "back-solved", "record stored"]
for k in range(10):
    print("%-34s -> %-34s %8d ticks" % (names[k], names[k + 1], t[k + 1] - t[k]))
print("whole frame %d ticks | kernel: prologue %d, sweep 1 (%d frames) %d, sweep 2 %d ticks" % (t[10] - t[0], t[13] - t[12], 30, t[14] - t[13], t[15] - t[14]))
