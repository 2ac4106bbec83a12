"""Measurement tool: how the time of an identical 20-iteration window changes with what the GPU did just before it -- back-to-back
windows from a cold start, after an idle gap, and right after a burst of unrelated device work (is the slow start of a fresh
process the clock governor, the engine's own state, or the host?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
W, K = 5, 20
w = wl.build(cfg, torch.device("cuda"))
eng, params = wl.engine_for(w, wl.bench_lr_schedule(W + K, "adam"), optimizer="adam", single_stream=True)
eng.run(W)
torch.cuda.synchronize()
T0 = time.perf_counter()

def windows(n, label):
    out = []
    for k in range(n):
        t0 = time.perf_counter()
        eng.run(K)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e6 / K)
        eng.rewind(W)
    print(f"{cfg} {label:28s} t={1e3*(time.perf_counter()-T0):7.1f} ms:", " ".join(f"{x:5.1f}" for x in out), flush=True)

windows(16, "cold start")
windows(16, "continued")
time.sleep(0.05); windows(8, "after 50 ms idle")
time.sleep(0.5); windows(8, "after 500 ms idle")
x = torch.randn(8192, 8192, device="cuda")
for _ in range(400):
    x = x * 0.999 + 0.001
windows(8, "right behind 400 torch ops")
time.sleep(0.5)
a = torch.randn(4096, 4096, device="cuda")
for _ in range(100):
    a = (a @ a) * 1e-3
windows(8, "right behind 100 matmuls")
time.sleep(0.5)
for _ in range(50):
    eng.run(K); eng.rewind(W)
windows(8, "right behind 1000 iterations")
