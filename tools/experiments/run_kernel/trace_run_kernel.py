"""Measurement tool: where an iteration of the one-launch run (engine.hip run_kernel) spends its time, per hypothesis' team.
    DDX_TRACE=1 DDX_RUN_TRACE=1 python tools/trace_run_kernel.py [config] [iters]
The stamps (s_memrealtime, 100 MHz) are those of step_wg / shade_wg, left by the LAST iteration of the run:
step 0 start, 1 head done, 2 pose done, 3 first meshlet transformed, 4 its scatter issued, 5/6 the last meshlet, 7 end;
shade 0 start, 1 scan done, 2 tiles done."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DDX_TRACE", "1")
os.environ.setdefault("DDX_RUN_TRACE", "1")
os.environ.setdefault("DDX_RUN_KERNEL", "1")
from diffdope_amd import _lib, workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w = wl.build(cfg, torch.device("cuda"))
eng, _ = wl.engine_for(w, wl.bench_lr_schedule(iters + 1, "adam"), optimizer="adam")
eng.run(iters); eng.finish()
print("run_form", eng.run_form)
lib = _lib.load()
TW = 4096
buf = np.zeros(3 * TW * 8, np.uint64)
n = lib.ddx_engine_trace_read(eng.handle, buf.ctypes.data, buf.size)
t = buf.reshape(3, TW, 8).astype(np.int64)
B = w["B"]
S, _ = eng.slices
st = t[0]; sh = t[1]
live = st[:, 0] > 0
G = int(live.sum()) // B
print(f"B {B}  team size {G}  shade slices {S}")
us = lambda x: x / 100.0
rows = []
for b in range(B):
    a = st[b * G:(b + 1) * G]            # step_wg: wg_id = b * SL + slot
    ids = [(z * S + sl) * B + b for z in range(2) for sl in range(S)]
    c = sh[ids]
    c = c[c[:, 0] > 0]
    s0, s7 = a[:, 0].min(), a[:, 7].max()
    h0, h2 = c[:, 0].min(), c[:, 2].max()
    rows.append(dict(step_span=us(s7 - s0), head=us(np.median(a[:, 1] - a[:, 0])), pose=us(np.median(a[:, 2] - a[:, 1])),
                     meshlets=us(np.median(a[:, 7] - a[:, 2])), step_wg_max=us((a[:, 7] - a[:, 0]).max()), step_wg_med=us(np.median(a[:, 7] - a[:, 0])),
                     start_skew=us(a[:, 0].max() - s0),
                     bar1=us(h0 - s7), shade_span=us(h2 - h0), scan=us(np.median(c[:, 1] - c[:, 0])), tiles_med=us(np.median(c[:, 2] - c[:, 1])),
                     tiles_max=us((c[:, 2] - c[:, 1]).max()), shade_skew=us(c[:, 0].max() - h0), total=us(h2 - s0)))
keys = list(rows[0].keys())
print("per hypothesis (us): median / p90 / max over the hypotheses")
for k in keys:
    v = np.array([r[k] for r in rows])
    print(f"  {k:12s} {np.median(v):7.2f} {np.percentile(v, 90):7.2f} {v.max():7.2f}")
hd = t[2]
if (hd[:, 0] > 0).any():  # built with -DDDX_TRACE_HEAD: the head's own stamps (update_head), workgroup b * G + k
    a = hd[:B * G]
    a = a[a[:, 0] > 0]
    names = ["loads back (sums held)", "re-arm issued, folded", "barrier 1", "barrier 2 (sums in LDS)", "tail (one lane)", "barrier 3"]
    print("head, us between stamps: median / p90 over the workgroups")
    for i in range(6):
        v = us(a[:, i + 1] - a[:, i])
        print(f"  {names[i]:26s} {np.median(v):6.2f} {np.percentile(v, 90):6.2f}")
    v = us(a[:, 6] - a[:, 0]); print(f"  {'whole head':26s} {np.median(v):6.2f} {np.percentile(v, 90):6.2f}")
    v = us(a[:, 0] - st[:B * G][st[:B * G][:, 0] > 0][:, 0]); print(f"  {'entry -> head':26s} {np.median(v):6.2f} {np.percentile(v, 90):6.2f}")
