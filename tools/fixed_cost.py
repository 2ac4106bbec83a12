"""Measurement tool: the fixed cost of a timed window of bench.py (run(n) + arg-min selection + sync), by a linear fit over n."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import dist as ddist, workloads as wl
w = wl.build("cfg2", torch.device("cuda"))
N = 400
eng, params = wl.engine_for(w, wl.bench_lr_schedule(N, "adam"), optimizer="adam", single_stream=not os.environ.get("TWO_CHAINS"))  # (one regime for the fit)
eng.run(20); torch.cuda.synchronize()
rows = []
for n in (1, 2, 5, 10, 20, 40, 80):
    ts = []
    for rep in range(7):
        eng.rewind(20)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if os.environ.get("OLD_SELECT"):
            eng.run(n)
            t1 = time.perf_counter()
            best = ddist.global_argmin_fused(eng.loss_log[20 + n - 1], 0b0101, eng.mtx_log[20 + n - 1], lo=0)
        else:
            best = ddist.run_and_select(eng, n)  # (round 4: the selection inside the run's last kernel)
            t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t2 - t0, t1 - t0))
    ts.sort()
    rows.append((n, ts[len(ts) // 2][0] * 1e6, ts[len(ts) // 2][1] * 1e6))
    print(f"n={n:3d}: window {rows[-1][1]:8.1f} us (host enqueue {rows[-1][2]:7.1f} us) -> {rows[-1][1]/n:7.2f} us/step")
a = np.polyfit([r[0] for r in rows], [r[1] for r in rows], 1)
print(f"fit: {a[0]:.2f} us/iteration + {a[1]:.1f} us fixed")
