"""Phase stamps (s_memtime) of k_lm_step on C2 windows of a batch (LIW_CLK=1 build: 2dliw-slam_amd/build.py): python tools/clk_probe.py [B]"""
import importlib, sys, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np, torch
liw=importlib.import_module('2dliw-slam_amd'); synth=importlib.import_module('2dliw-slam_amd.synth')
prm=synth.office_params()
hp=liw.HostPreint(prm)
w=[synth.make_window(hp,prm,seed=20240+k,n=30,L=2000) for k in range(2)]
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
bs=liw.BatchSolver(prm,[w[k%2] for k in range(B)])
bs.solve(liw.LIW_MODE_INIT, 6); torch.cuda.synchronize()
clk=np.zeros(8192,dtype=np.int64)
liw.lib().liw_debug_clk(clk.ctypes.data_as(C.c_void_p), C.c_int(8192))
print('sweep1 total', clk[1]-clk[0], 'gap', clk[2]-clk[1], 'sweep2 total', clk[3]-clk[2])
for i in (29,15,1):
    t=clk[10+i*8:10+i*8+6]
    print('frame',i,'assemble',t[1]-t[0],'diag/gmax',t[2]-t[1],'colload',t[3]-t[2],'chol',t[4]-t[3],'ldsW',t[5]-t[4],'mfma+record+C', clk[10+(i-1)*8]-t[5])
print('assemble stamps', [int(clk[2000+k+1]-clk[2000+k]) for k in range(4)])
for i in (1,15,29):
    print('bs frame',i,'loads+matvec',clk[300+i*4+1]-clk[300+i*4],'emit',clk[300+(i+1)*4]-clk[300+i*4+1] if i<29 else clk[3]-clk[300+i*4+1])
