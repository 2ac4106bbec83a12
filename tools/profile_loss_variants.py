import sys, time, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
dev = torch.device('cuda:0')
w = wl.build('cfg2', dev)
n=220
lrs = [0.005 * l / 2.0 for l in wl.lr_schedule(n - 1, 20, 0.1)]
for wts in [dict(rgb=0.7, mask=1.0), dict(rgb=0.7), dict(mask=1.0), dict(depth=1.0), dict(rgb=0.7,depth=1.0,mask=1.0)]:
    p = w['params0'].clone()
    eng = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], p, w['lr_mult'], lrs, wts, uv=w['uv'], tex=w['tex'], optimizer='adam')
    eng.run(20); torch.cuda.synchronize()
    t=time.time(); eng.run(200); torch.cuda.synchronize(); dt=time.time()-t
    eng2 = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], w['params0'].clone(), w['lr_mult'], lrs, wts, uv=w['uv'], tex=w['tex'], optimizer='adam')
    pr = eng2.profile(0, 20)
    print(sorted(wts), f'{200/dt:.0f} it/s', {k: round(v*1e3,1) for k,v in pr.items()})
