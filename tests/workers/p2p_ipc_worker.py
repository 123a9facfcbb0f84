"""One rank of the two-PROCESS peer-write exchange test (tests/test_gpu_p2p_ipc.py): both processes share cuda:0, rendezvous over gloo,
map each other's receive areas with hipIpc (BatchSolver.p2p_attach_ipc) and run the factor-sharded solve with exchange="p2p".
argv: out_prefix scenario    (env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)
scenario "solve": every rank also solves with the all-gather variant (gloo, through the host) and saves both results.
scenario "dead_peer": the last rank attaches and leaves without ever pushing; the others must get LIW_EHIP, not a hang."""
import datetime
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    out, scenario = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
    liw = importlib.import_module("2dliw-slam_amd")
    synth = importlib.import_module("2dliw-slam_amd.synth")
    prm = synth.office_params()
    hp = liw.HostPreint(prm)
    n, K = 10, 20
    ws = [synth.make_window(hp, prm, seed=4300 + k, n=n, L=900 + 100 * k) for k in range(5)]   # same seeds on every rank
    M = liw.LIW_MODE_INIT
    res = {"rank": rank, "world": world}
    p2p = liw.BatchSolver(prm, ws, rank=rank, world=world, exchange="p2p")
    p2p.p2p_attach_ipc(M)
    if scenario == "dead_peer":
        if rank == world - 1:
            dist.barrier()
            res["left"] = True
        else:
            dist.barrier()
            t0 = time.perf_counter()
            try:
                p2p.solve(M, K)
                torch.cuda.synchronize()
                res["error"] = None
            except liw.LiwError as e:
                res["error"] = [int(e.code), str(e)]
            res["seconds"] = time.perf_counter() - t0
            st = importlib.import_module("ctypes").c_int(0)
            p2p.L.liw_batch_p2p_status(p2p.h, importlib.import_module("ctypes").byref(st))
            res["timed_out_rank_plus_1"] = int(st.value)
    else:
        p2p.solve(M, K)
        torch.cuda.synchronize()
        xp = p2p.states()
        res["p2p_iterations"] = [s["iterations"] for s in p2p.summaries()]
        one = liw.BatchSolver(prm, ws, rank=rank, world=world, exchange="oneshot")
        one.solve(M, K)
        torch.cuda.synchronize()
        xo = one.states()
        res["oneshot_iterations"] = [s["iterations"] for s in one.summaries()]
        np.save(out + "_p2p_rank%d.npy" % rank, xp)
        np.save(out + "_oneshot_rank%d.npy" % rank, xo)
        # a second solve on the SAME areas after a second setup: the exchange counter has to continue from the flags
        p2p.t["x"].copy_(torch.from_numpy(np.concatenate([np.asarray(w["states"]).reshape(-1) for w in ws])).to(p2p.dev))
        p2p._p2p_set(M, p2p._p2p_ptrs[0], p2p._p2p_ptrs[1], p2p._p2p["keep"])
        dist.barrier()
        p2p.solve(M, K)
        torch.cuda.synchronize()
        res["second_setup_identical"] = bool(np.array_equal(p2p.states(), xp))
        one.close()
    with open(out + "_rank%d.json" % rank, "w") as f:
        json.dump(res, f)
    dist.barrier() if scenario != "dead_peer" else None
    dist.destroy_process_group()
    os._exit(0)     # (mapped IPC areas of a peer that is already gone: skip the interpreter's tear-down order)


if __name__ == "__main__":
    main()
