"""One batch against the same hypotheses split over 2 / 4 engines on as many streams (sub-batches never paid: DESIGN.md).
usage: [cfg2|cfg3|cfg5]"""
import sys, time, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
dev = torch.device('cuda:0')
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
NIT = 220
Btot = wl.CONFIGS[cfg]['B']
def make(lo, B):
    w = wl.build(cfg, dev, B=B, global_lo=lo, global_B=Btot)
    lrs = [0.005 * l / 2.0 for l in wl.lr_schedule(NIT - 1, 20, 0.1)]
    p = w['params0'].clone()
    kw = dict(uv=w['uv'], tex=w['tex']) if w['tex'] is not None else dict(vtx_color=w['vtx_color'])
    return dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], p, w['lr_mult'], lrs, w['weights'], optimizer='adam', global_batch=Btot, **kw), p
for parts in (1, 2, 4):
    Bp = Btot // parts
    es = [make(i * Bp, Bp) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in es]
    for (e, p), st in zip(es, streams):
        with torch.cuda.stream(st): e.run(20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for (e, p), st in zip(es, streams):
        with torch.cuda.stream(st): e.run(200)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{cfg}: {parts} x {Bp} hypotheses on {parts} stream(s): {200/dt:.0f} it/s")
