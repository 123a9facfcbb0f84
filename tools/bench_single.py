import importlib, sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
liw=importlib.import_module('2dliw-slam_amd'); synth=importlib.import_module('2dliw-slam_amd.synth')
prm=synth.office_params(); hp=liw.HostPreint(prm)
w=[synth.make_window(hp,prm,seed=20240,n=30,L=2000)]
s1=liw.BatchSolver(prm,w)
x1=s1.t["x"].clone()
for g in (False, True):
    for rep in range(4):
        if rep==1: torch.cuda.synchronize(); ts=time.perf_counter()
        s1.t["x"].copy_(x1); s1.t["has_prior"].zero_()
        s1.solve(liw.LIW_MODE_INIT, 50, use_graph=g); s1.marginalize()
    torch.cuda.synchronize(); print('graph',g,'ms/solve',1e3*(time.perf_counter()-ts)/3, s1.summaries()[0]['iterations'])
