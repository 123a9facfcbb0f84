"""Two (or more) BatchSolver objects, each on its own HIP stream AND driven by its own host thread (liw_batch_solve blocks at its
active-window read-backs, so one thread serialises the solves: tools/two_stream_probe.py measures no overlap for that reason).
usage: python tools/two_thread_probe.py [B]"""
import importlib, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
prm = synth.office_params()
wins = bench.make_batch(liw, synth, prm, B, 30, 2000, seed0=20240, n_base=64)


def run(solvers, streams, reps=3):
    x0 = [s.t["x"].clone() for s in solvers]
    mp0 = [s.t["match_pose"].clone() for s in solvers]
    best = 1e9
    for rep in range(reps + 1):
        for s, a, b in zip(solvers, x0, mp0):
            s.t["x"].copy_(a); s.t["match_pose"].copy_(b)
        torch.cuda.synchronize()

        def work(s, st):
            with torch.cuda.stream(st):
                s.solve(liw.LIW_MODE_INIT, 50)
                s.marginalize()
            st.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(s, st)) for s, st in zip(solvers, streams)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    return best


for parts in (1, 2, 3, 4):
    hs = [liw.BatchSolver(prm, wins[k::parts]) for k in range(parts)]
    t2 = run(hs, [torch.cuda.Stream() for _ in range(parts)])
    print("%d threads / streams x %d windows: %.1f ms -> %.0f solves/s" % (parts, len(wins[0::parts]), 1e3 * t2, B / t2))
    for h in hs:
        h.close()
