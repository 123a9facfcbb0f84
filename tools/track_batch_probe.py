"""bench.py's tracking_batch leg alone (python tools/track_batch_probe.py [B] [K]) — for rocprofv3 passes of the batched TRACK path."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
print(json.dumps(bench.track_batch_leg(liw, synth, synth.office_params(), "cuda:0", B, K, 64, 6, cpu="--no-cpu" not in sys.argv)))
