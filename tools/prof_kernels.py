"""Per-kernel HIP-event durations of the engine on cfg2 (default weights) -- ddx_engine_profile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
w = wl.build(sys.argv[1] if len(sys.argv) > 1 else 'cfg2', torch.device('cuda:0'))
eng = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], w['params0'].clone(), w['lr_mult'], [0.0] * 60, w['weights'], uv=w['uv'], tex=w['tex'], vtx_color=w['vtx_color'])
print({k: round(v * 1e3, 1) for k, v in eng.profile(0, 40).items()})
