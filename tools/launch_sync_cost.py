"""Measurement tool: what the host side of a timed window costs on this box -- one tiny kernel + synchronisation round trips, and
the same with the result awaited by polling pinned host memory (no runtime wait)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
p = torch.zeros(64, device=dev); g = torch.ones(64, device=dev)
pin = torch.zeros(64, dtype=torch.float32, pin_memory=True); pn = pin.numpy()
gp = torch.ones(64, device=dev)
s = torch.cuda.current_stream()
def launch(dst): lib.ddx_sgd_step(dst.data_ptr(), g.data_ptr(), ctypes.c_float(1.0), 64, _lib.stream_ptr())
for _ in range(20): launch(p); torch.cuda.synchronize()
def med(f, n=200):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6
def a(): launch(p); torch.cuda.synchronize()
def b(): launch(p); s.synchronize()
def c():
    v = pn[0]; launch(pin)
    while pn[0] == v: pass
def d():
    v = pn[0]; launch(pin)
    while pn[0] == v: pass
    torch.cuda.synchronize()
def e(): torch.cuda.synchronize()
def f():
    for _ in range(10): launch(p)
    torch.cuda.synchronize()
for name, fn in (("1 tiny kernel + device synchronize", a), ("1 tiny kernel + stream synchronize", b), ("1 tiny kernel writing pinned memory, host polls it", c),
                 ("... then device synchronize", d), ("device synchronize on an idle device", e), ("10 tiny kernels + device synchronize", f)):
    m, lo = med(fn)
    print(f"{name:60s} median {m:7.1f} us   p10 {lo:7.1f} us")
