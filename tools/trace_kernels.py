"""Measurement tool: per-workgroup phase stamps (s_memrealtime, 100 MHz) of step_kernel / shade_kernel.
    DDX_TRACE=1 python tools/trace_kernels.py [config] [distance]
step stamps: 0 start, 1 head done, 2 pose/matrices done, 3 first meshlet transformed (barrier passed), 4 its scatter issued,
5/6 the same for the second meshlet, 7 end.  shade: 0 start, 1 scan done, 2 tiles done, 3 end; [4] = grid z << 32 | tiles (z = 0 is the MASK role by default: engine_create)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DDX_TRACE", "1")
from diffdope_amd import _lib, workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dist = float(sys.argv[2]) if len(sys.argv) > 2 else None
w = wl.build(cfg, torch.device("cuda"), distance=dist)
eng, _ = wl.engine_for(w, wl.bench_lr_schedule(40, "adam"), optimizer="adam")
eng.run(30); eng.finish()
lib = _lib.load()
lib.ddx_engine_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
TW = 4096
buf = np.zeros(3 * TW * 8, np.uint64)
n = lib.ddx_engine_trace_read(eng.handle, buf.ctypes.data, buf.size)
assert n == buf.size, n
t = buf.reshape(3, TW, 8).astype(np.int64)
us = lambda x: x / 100.0
for k, name, cols in ((0, "step", 8), (1, "shade", 4)):
    a = t[k]
    live = a[:, 0] > 0
    a = a[live]
    if not len(a):
        continue
    t0 = a[:, 0].min()
    print(f"== {name}: {len(a)} workgroups, kernel span {us(a[:, :cols].max() - t0):.2f} us")
    rel = us(a[:, :cols] - t0)
    rel[a[:, :cols] == 0] = np.nan
    names = ["start", "head", "pose", "xfm1", "scat1", "xfm2", "scat2", "end"] if k == 0 else ["start", "scan", "tiles", "end"]
    for i in range(cols):
        c = rel[:, i][~np.isnan(rel[:, i])]
        if len(c):
            print(f"  {names[i]:6s} median {np.median(c):6.2f}  p10 {np.percentile(c,10):6.2f}  p90 {np.percentile(c,90):6.2f}  max {c.max():6.2f}   (n={len(c)})")
    if k == 1:
        role = a[:, 4] >> 32; nt = a[:, 4] & 0xffffffff
        for r in (0, 1):
            m = role == r
            if m.any():
                d = us(a[m][:, 3] - a[m][:, 0]); sc = us(a[m][:, 1] - a[m][:, 0])
                print(f"  grid z = {r} ({'mask role unless the set-up put the colour role first' if r == 0 else 'colour role unless ...'}): {m.sum()} wgs, tiles/wg mean {nt[m].mean():.2f}, in-kernel duration median {np.median(d):.2f} p90 {np.percentile(d,90):.2f}; scan median {np.median(sc):.2f}; start median {np.median(us(a[m][:,0]-t0)):.2f}")
