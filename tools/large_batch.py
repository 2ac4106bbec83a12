"""Per-kernel durations and hypothesis-iterations/s at 512 hypotheses per GPU (cfg2 / cfg4 / cfg3 meshes), under both rasteriser rules
(both faces = dr.rasterize's rule, the default; culled = deviation D5)."""
import sys, time, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
for cfg, B, cull in [(c, 512, k) for c in ('cfg2', 'cfg4', 'cfg3') for k in (False, True)]:
    w = wl.build(cfg, torch.device('cuda:0'), B=B)
    n = 40
    lrs = [0.005 * l / 2.0 for l in wl.lr_schedule(n - 1, 20, 0.1)]
    p = w['params0'].clone()
    kw = dict(uv=w['uv'], tex=w['tex']) if w['tex'] is not None else dict(vtx_color=w['vtx_color'])
    eng = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], p, w['lr_mult'], lrs, w['weights'], optimizer='adam', cull_backfaces=cull, **kw)
    eng.run(10); torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run(20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = eng.profile(30, 5)
    print(f"{cfg} B={B} {'culled' if cull else 'both faces'}: {20/dt:.0f} it/s, {20*B/dt/1e6:.2f} M hyp-it/s", {k[:7]: round(v*1e3) for k, v in pr.items()})
