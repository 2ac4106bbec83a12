"""Measurement tool: what makes the LAST workgroups of a step_kernel launch late?  One chain of full-batch launches (DDX_TWO_STREAMS=0),
per-workgroup stamps + where each ran (XCD, CU):   DDX_TRACE=1 DDX_TWO_STREAMS=0 python tools/trace_tail.py [config] [B]
Prints the end time by slot (which meshlets), by XCD, by the number of workgroups that shared the CU, and the head's duration by the
same keys."""
import ctypes, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DDX_TRACE", "1")
os.environ.setdefault("DDX_TWO_STREAMS", "0")
from diffdope_amd import _lib, workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else None
w = wl.build(cfg, torch.device("cuda"), B=B)
eng, _ = wl.engine_for(w, wl.bench_lr_schedule(40, "adam"), optimizer="adam", single_stream=True)
eng.run(30); eng.finish()
lib = _lib.load()
lib.ddx_engine_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
TW = 4096
buf = np.zeros(3 * TW * 8, np.uint64)
assert lib.ddx_engine_trace_read(eng.handle, buf.ctypes.data, buf.size) == buf.size
t = buf.reshape(3, TW, 8)
a = t[0].astype(np.int64); hw = t[2][:, 7]; hd = t[2].astype(np.int64)
live = a[:, 0] > 0
ids = np.nonzero(live)[0]
a = a[live]; hw = hw[live]; hd = hd[live]
SL = max(1, int(ids.max() + 1) // w["B"])   # step_kernel's workgroups per hypothesis
slot = ids % SL; hyp = ids // SL
t0 = a[:, 0].min()
rel = (a - t0) / 100.0
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
hwid = hw.astype(np.int64) & 0xffffffff
cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
cukey = xcc * 1000 + se * 100 + sh * 16 + cu
print(f"{cfg}: {len(a)} workgroups, {SL} slots x {len(set(hyp))} hypotheses; span {rel[:, 7].max():.2f} us; start median {np.median(rel[:, 0]):.2f} max {rel[:, 0].max():.2f}")
end = rel[:, 7]; head = rel[:, 1] - rel[:, 0]; work = rel[:, 7] - rel[:, 1]
def by(key, name, top=None):
    g = collections.defaultdict(list)
    for k, e, h, wk in zip(key, end, head, work): g[int(k)].append((e, h, wk))
    rows = sorted(g.items())
    print(f"-- by {name} ({len(rows)} groups): key n | end mean max | head mean | work mean")
    if top: rows = sorted(rows, key=lambda kv: -np.mean([x[0] for x in kv[1]]))[:top]
    for k, v in rows:
        v = np.array(v)
        print(f"   {k:6d} {len(v):4d} | {v[:, 0].mean():6.2f} {v[:, 0].max():6.2f} | {v[:, 1].mean():6.2f} | {v[:, 2].mean():6.2f}")
by(slot, "slot")
by(xcc, "XCD")
per_cu = collections.Counter(cukey.tolist())
by(np.array([per_cu[k] for k in cukey.tolist()]), "workgroups sharing the CU")
print("distinct CUs:", len(per_cu), " workgroups per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
late = end > np.percentile(end, 90)
print(f"late tenth: head mean {head[late].mean():.2f} (all {head.mean():.2f}); work mean {work[late].mean():.2f} (all {work.mean():.2f}); start mean {rel[late, 0].mean():.2f} (all {rel[:, 0].mean():.2f})")
print("late tenth by slot:", sorted(collections.Counter(slot[late].tolist()).items()))
print("late tenth by XCD:", sorted(collections.Counter(xcc[late].tolist()).items()))
print("late tenth by workgroups on its CU:", sorted(collections.Counter([per_cu[k] for k in cukey[late].tolist()]).items()))
for i, nm in enumerate(["start", "head", "pose", "xfm1", "scat1", "xfm2", "scat2", "end"]):
    c = rel[:, i]
    c = c[a[:, i] > 0]
    if not len(c): continue
    print(f"  {nm:6s} median {np.median(c):6.2f}  p10 {np.percentile(c, 10):6.2f}  p90 {np.percentile(c, 90):6.2f}  max {c.max():6.2f}")

if (hd[:, 0] > 0).any():  # a DDX_TRACE_HEAD build: the head's own phases (0 entry, 1 partial rows summed, 2 reduced, 3 barrier, 4 background term, 5 tail of wave 0, 6 barrier)
    hrel = (hd[:, :7] - a[:, :1]) / 100.0
    keep = hrel[:, 6] < rel[:, 7].max()   # (finish_kernel runs the same head later and overwrites the rows of its own workgroup ids)
    hrel, late, slot = hrel[keep], late[keep], slot[keep]
    print(f"  ({keep.sum()} of {len(keep)} rows are step_kernel's)")
    print("head phases, all workgroups / the late tenth / slot 0:")
    for i in range(7):
        c = hrel[:, i]
        print(f"  h{i} median {np.median(c):6.2f} p90 {np.percentile(c, 90):6.2f} max {c.max():6.2f} | late {np.median(c[late]):6.2f} | slot0 {np.median(c[slot == 0]):6.2f}")
