"""Experiment (round 6): how does the shading launch depend on the slices per hypothesis (tiles per workgroup)?  cfg2, one chain.
    python tools/experiments/shade_slices_sweep.py [slices ...]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffdope_amd import workloads as wl

S = [int(a) for a in sys.argv[1:]] or [0, 6, 8, 9, 10, 12, 16]
cfg = os.environ.get("CFG", "cfg2")
w = wl.build(cfg, torch.device("cuda:0"))
n, warm = 200, 20
for s in S:
    eng, _ = wl.engine_for(w, wl.bench_lr_schedule(n + warm, "adam"), optimizer="adam", single_stream=True, shade_slices=s)
    ts = []
    for _ in range(5):
        eng.new_observation(params=w["params0"])
        eng.run(warm); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.run(n); eng.finish(); ts.append((time.perf_counter() - t0) / n * 1e6)
    eng.new_observation(params=w["params0"])
    pr = eng.profile(5, 30)
    print(f"{cfg} shade_slices={s} (runs with {eng.slices}): {statistics.median(ts):7.2f} us/it; kernels {{{', '.join(f'{k[:6]} {v * 1e3:.1f}' for k, v in pr.items())}}}; active tiles {eng.status()['active_tiles']}", flush=True)
    del eng
