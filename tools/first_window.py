"""Measurement tool: where does the one-off cost of the FIRST long window of a fresh engine go?  (bench.py's first timed window
after a short warm-up measured 1.3 ms more than every later one on some workloads.)  Prints, for a fresh engine: warm-up of W
iterations, then windows of K iterations with the host's enqueue time and the time to completion apart; optionally with the
window cut into chunks with a synchronisation after each (in-flight depth bounded)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import dist as ddist, workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
w = wl.build(cfg, torch.device("cuda"))
eng, params = wl.engine_for(w, wl.bench_lr_schedule(W + K, "adam"), optimizer="adam")
sel = bool(os.environ.get("SELECT"))
if sel:
    ddist.run_and_select(eng, W)
else:
    eng.run(W)
torch.cuda.synchronize()
for k in range(5):
    t0 = time.perf_counter()
    if chunk:
        left = K
        while left > 0:
            eng.run(min(chunk, left)); torch.cuda.synchronize(); left -= min(chunk, left)
        t1 = time.perf_counter()
    else:
        if sel:
            ddist.run_and_select(eng, K)
        else:
            eng.run(K)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{cfg} W={W} K={K} chunk={chunk} window {k}: enqueue {1e6*(t1-t0):8.1f} us, done {1e6*(t2-t0):8.1f} us -> {1e6*(t2-t0)/K:6.2f} us/it")
    eng.rewind(W)
