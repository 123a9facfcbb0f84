import importlib, sys, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np, torch
liw=importlib.import_module('2dliw-slam_amd'); synth=importlib.import_module('2dliw-slam_amd.synth')
prm=synth.office_params()
hp=liw.HostPreint(prm)
w=[synth.make_window(hp,prm,seed=20240+k,n=30,L=2000) for k in range(2)]
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
bs=liw.BatchSolver(prm,[w[k%2] for k in range(B)])
for _ in range(3):
    bs.linearize(liw.LIW_MODE_INIT)
torch.cuda.synchronize()
clk=np.zeros(512,dtype=np.int64)
liw.lib().liw_debug_clk_lin(clk.ctypes.data_as(C.c_void_p), C.c_int(512))
print('total', clk[2]-clk[0], 'stage', clk[1]-clk[0])
for p in range(10):
    t=clk[8+p*8:8+p*8+8]
    print('pass',p,'wait+rows',t[1]-t[0],'prefetch-issue->prod0',t[2]-t[1],'prod0',t[3]-t[2],'emit0',t[4]-t[3] if t[4] else None,'prod1',t[5]-t[4] if t[5] else None, 'next', clk[8+(p+1)*8]-max(t[3],t[5]))

print('imu: prefetch', clk[300]-0 if False else 0, 'zero+sync', clk[301]-clk[300], 'alpha/beta', clk[302]-clk[301], 'closed', clk[303]-clk[302], 'gamma+sync', clk[304]-clk[303])
for g in range(6):
    print(' block', g, 'mfma', clk[310+2*g]-(clk[304] if g==0 else clk[309+2*g]), 'stores', clk[311+2*g]-clk[310+2*g])
print('imu total from 300', clk[321]-clk[300])
