"""Exploration tool (needs a build with DDX_CXXFLAGS=-DDDX_TRACE): per-workgroup start/end cycle stamps of the
last shade_kernel launch -> distribution of workgroup durations, start skew, per-CU load."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl, _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
KIDX = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # 0 scatter, 1 compact/big-tile pass, 2 shade, 3 update
w = wl.build(cfg, torch.device('cuda:0'))
eng = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], w['params0'].clone(), w['lr_mult'], [0.0] * 60, w['weights'], uv=w['uv'], tex=w['tex'], vtx_color=w['vtx_color'])
eng.run(20)
torch.cuda.synchronize()
lib = _lib.load()
lib.ddx_engine_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
path = '/tmp/trace.bin'
assert lib.ddx_engine_trace_dump(eng.handle, path.encode()) == 0
t = np.fromfile(path, dtype=np.uint64).reshape(4, 8192, 4)[KIDX]
t = t[t[:, 1] > 0]
start, end, hw, info = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2], t[:, 3]
role, units = (info >> np.uint64(32)).astype(int), (info & np.uint64(0xffffffff)).astype(int)
xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(int)
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 0xf
se = (hwid >> 13) & 0x7
sh = (hwid >> 12) & 0x1
key = xcc * 1000 + se * 100 + sh * 50 + cu
dur = end - start
rel = start - start.min()  # s_memrealtime: 100 MHz, device-wide
relend = rel + dur
print('workgroups', len(t), 'xccs', sorted(set(xcc)), 'distinct CUs', len(set(key)))
print('kernel span per xcc (10 ns ticks):', {int(x): int(relend[xcc == x].max()) for x in sorted(set(xcc))})
for r in sorted(set(role)):
    for u in sorted(set(units[role == r])):
        m = (role == r) & (units == u)
        print(f'role {r} units {u}: n={m.sum():5d} dur mean {dur[m].mean():8.0f} p50 {np.median(dur[m]):8.0f} p95 {np.percentile(dur[m],95):8.0f} max {dur[m].max():8.0f}   start mean {rel[m].mean():8.0f} p95 {np.percentile(rel[m],95):8.0f} max {rel[m].max():8.0f}')
load = np.array([dur[key == k].sum() for k in sorted(set(key))])
cnt = np.array([(key == k).sum() for k in sorted(set(key))])
print('per-CU: workgroups min/mean/max', cnt.min(), cnt.mean(), cnt.max(), ' sum-dur min/mean/max', load.min(), int(load.mean()), load.max())
print('start percentiles:', [int(np.percentile(rel, p)) for p in (10, 25, 50, 75, 90, 99, 100)])
print('end percentiles:  ', [int(np.percentile(relend, p)) for p in (10, 25, 50, 75, 90, 99, 100)])
# concurrency profile: how many workgroups are in flight over time, chip-wide (per-xcc relative clocks)
T = int(relend.max()); grid = np.linspace(0, T, 21)
print('in-flight workgroups at 5% steps:', [int(((rel <= g) & (relend > g)).sum()) for g in grid])
