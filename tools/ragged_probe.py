"""bench.py's ragged_c2 leg alone: python tools/ragged_probe.py [B]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
print(json.dumps(bench.ragged_leg(liw, synth, synth.office_params(), "cuda:0", B, 30, 2000, 50, 64)))
