import sys, torch
sys.path.insert(0, '/root/repo')
import diffdope_amd as dd
from diffdope_amd import workloads as wl
dev = torch.device('cuda:0')
w = wl.build('cfg2', dev)
lrs = [0.0025] * 60
eng = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], w['params0'].clone(), w['lr_mult'], lrs, w['weights'], uv=w['uv'], tex=w['tex'], optimizer='adam')
print({k: round(v*1e3,1) for k,v in eng.profile(0, 30).items()})
