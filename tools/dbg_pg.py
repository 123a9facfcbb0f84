import importlib, sys
sys.path.insert(0,'/root/repo')
import numpy as np
liw = importlib.import_module("2dliw-slam_amd"); synth = importlib.import_module("2dliw-slam_amd.synth")
from oracle import pyoracle as po
prm = synth.office_params(); pg = liw.posegraph.office_pg_params()
for (N, nl, seed) in ((12, 0, 1), (60, 5, 2)):
    G = liw.posegraph.make_pose_graph(prm, N=N, seed=seed, n_loop=nl)
    pgs = liw.posegraph.PoseGraph(prm); orc = po.Oracle(prm)
    for cap in (1, 3, 5, 10, 20, 30, 50):
        xg, sg = pgs.solve(pg, G["poses"], G["seq_idx"], G["seq_tf12"], G["loop_idx"], G["loop_tf12"], max_iters=cap)
        xo, so = po.posegraph_solve(orc, pg, G["poses"], G["seq_idx"], G["seq_tf12"], G["loop_idx"], G["loop_tf12"], max_iters=cap)
        print(N, cap, sg["iterations"], so["iterations"], sg["successful"], so["successful"], sg["final_cost"], so["final_cost"], np.abs(xg - xo).max())
    pg2 = dict(pg, use_ground_q_factor=False)
    xg, sg = pgs.solve(pg2, G["poses"], G["seq_idx"], G["seq_tf12"], G["loop_idx"], G["loop_tf12"])
    xo, so = po.posegraph_solve(orc, pg2, G["poses"], G["seq_idx"], G["seq_tf12"], G["loop_idx"], G["loop_tf12"])
    print('no ground_q', sg, so, np.abs(xg - xo).max())
