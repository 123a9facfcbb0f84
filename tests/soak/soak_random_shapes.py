"""Soak run of tests/test_gpu_random_shapes.py over many more seeds than the committed parametrisation (test infrastructure:
imports the oracle).  usage: python tests/soak/soak_random_shapes.py FIRST LAST   (on the MI355X box)"""
import importlib
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle
pyoracle.build()
import test_gpu_random_shapes as t

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    try:
        t.test_random_window_init_then_track(liw, synth, pyoracle, seed)
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", repr(e)[:300])
        traceback.print_exc(limit=2)
print("seeds %d..%d: %d failures %s" % (first, last - 1, len(bad), bad))
