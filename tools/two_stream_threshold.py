"""Measurement tool: from how many iterations on does a run as two half-batch chains on two streams (engine.hip engine_run_impl)
beat the single chain?  Same process, two engines on the same workload (single_stream on / off, DDX_TWO_MIN=2 so that every run of
the second one forks), alternating, median of 15 windows per length; run + fused selection + synchronisation as bench.py times it."""
import os, sys, time
os.environ["DDX_TWO_MIN"] = "2"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import dist as ddist, workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else None  # (hypotheses; default: the config's)
w = wl.build(cfg, torch.device("cuda"), **({"B": B} if B else {}))
cfg = f"{cfg}/B{B}" if B else cfg
N = 400
engs = {}
for single in (True, False):
    engs[single], _ = wl.engine_for(w, wl.bench_lr_schedule(N, "adam"), optimizer="adam", single_stream=single)
    engs[single].run(20)
torch.cuda.synchronize()
for _ in range(10):  # (steady state of the board first: tools/ramp_probe.py)
    for e in engs.values():
        e.rewind(20); e.run(100)
torch.cuda.synchronize()
print(f"{cfg}: n, single chain us, two chains us, difference us")
for n in (6, 10, 14, 16, 20, 24, 32, 48, 64, 100):
    ts = {True: [], False: []}
    for rep in range(15):
        for single in (True, False):
            e = engs[single]
            e.rewind(20)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ddist.run_and_select(e, n)
            torch.cuda.synchronize()
            ts[single].append((time.perf_counter() - t0) * 1e6)
    a, b = float(np.median(ts[True])), float(np.median(ts[False]))
    print(f"{cfg}: n={n:3d}  {a:8.1f}  {b:8.1f}  {b - a:+7.1f}")
