"""Soak run of tests/test_gpu_api_fuzz.py (differential fuzz of the single-window C ABI state machine) over many more seeds.
usage: python tests/soak/soak_api_fuzz.py FIRST LAST   (on the MI355X box; test infrastructure)"""
import importlib
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
liw = importlib.import_module("2dliw-slam_amd")
synth = importlib.import_module("2dliw-slam_amd.synth")
import test_gpu_api_fuzz as t


class Env:   # the two monkeypatch calls the test uses
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(first, last):
    try:
        t.test_random_call_sequences_speculative_vs_plain(liw, synth, Env(), seed)
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", repr(e)[:300])
        traceback.print_exc(limit=2)
print("seeds %d..%d: %d failures %s" % (first, last - 1, len(bad), bad))
