import importlib, sys
sys.path.insert(0,'/root/repo')
import numpy as np
liw=importlib.import_module('2dliw-slam_amd'); synth=importlib.import_module('2dliw-slam_amd.synth')
from oracle import pyoracle
prm=synth.office_params(); orc=pyoracle.Oracle(prm)
for n,L in ((1,0),(2,0),(2,5)):
    d=synth.make_window(orc,prm,seed=5,n=n,L=L)
    wo,wg=pyoracle.Window(d),liw.Window(d)
    orc.set_prior(None); orc.init_solve(wo)
    slv=liw.Solver(prm); slv.set_window(wg); s=slv.init_solve()
    so=orc.summary(); ho=orc.iterations(); hg=slv.history()
    print(n,L,s,so)
    for k in range(min(len(ho),len(hg))):
        e=np.abs(hg[k]-ho[k]['x'].reshape(n,15)).max()
        if k<6 or e>1e-6: print('  it',k,'err %.3e'%e, 'cost', ho[k]['cost'], 'rho', ho[k]['relative_decrease'], ho[k]['successful'])
