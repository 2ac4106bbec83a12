"""Several objects of one frame (independent engines) one after the other, on one stream each, and with interleaved
launches -- the measurement behind bop.refine_frame running its objects on separate streams.  usage: [cfg5|cfg2|...]"""
import sys, time, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
dev = torch.device('cuda:0')
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg5'
NOBJ, NIT = 4, 60
def make(seed):
    w = wl.build(cfg, dev, seed=seed)
    lrs = [0.005 * l / 2.0 for l in wl.lr_schedule(NIT - 1, 20, 0.1)]
    p = w['params0'].clone()
    kw = dict(uv=w['uv'], tex=w['tex']) if w['tex'] is not None else dict(vtx_color=w['vtx_color'])
    e = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], p, w['lr_mult'], lrs, w['weights'], optimizer='adam', **kw)
    return e, w, p
def fresh():
    es = [make(s) for s in range(NOBJ)]
    for e, w, p in es: e.run(2)
    torch.cuda.synchronize()
    return es
# sequential
es = fresh(); t0 = time.perf_counter()
for e, w, p in es: e.run(NIT - 2)
torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
ref = [p.clone() for e, w, p in es]
# concurrent: one stream per object, whole runs enqueued
es = fresh(); streams = [torch.cuda.Stream() for _ in es]; t0 = time.perf_counter()
for (e, w, p), st in zip(es, streams):
    with torch.cuda.stream(st): e.run(NIT - 2)
torch.cuda.synchronize(); t_par = time.perf_counter() - t0
same = all(torch.equal(a, p) for a, (e, w, p) in zip(ref, es))
# interleaved per iteration on separate streams
es = fresh(); t0 = time.perf_counter()
for it in range(NIT - 2):
    for (e, w, p), st in zip(es, streams):
        with torch.cuda.stream(st): e.run(1)
torch.cuda.synchronize(); t_int = time.perf_counter() - t0
print(f"{cfg}: {NOBJ} objects x {NIT-2} iterations: sequential {t_seq*1e3:.1f} ms, one stream per object {t_par*1e3:.1f} ms (identical results: {same}), interleaved launches {t_int*1e3:.1f} ms")
