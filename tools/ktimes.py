"""per-kernel times of one LM iteration over a full C2 batch (liw_batch_time_kernels): python tools/ktimes.py [B]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
prm = synth.office_params()
tw = bench.make_tiled(liw, synth, prm, B, 30, 2000, seed0=20240, n_base=64)      # (tiled on the device: no B-fold host concatenation)
bs = liw.BatchSolver(prm, tw.base, tile=tw.tile())
kt = bs.time_kernels(liw.LIW_MODE_INIT, 3)
print("LIW_NO_LASER_SLAB=%s" % os.environ.get("LIW_NO_LASER_SLAB"), {k: round(v, 4) for k, v in kt.items()})
