"""Exploration tool (build with DDX_CXXFLAGS="-DDDX_TRACE -DDDX_PHASES"): phase stamps of update_xfm_kernel (thread 0 of every
workgroup; phases separated by s_waitcnt 0)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl, _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
w = wl.build(cfg, torch.device('cuda:0'))
eng = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], w['params0'].clone(), w['lr_mult'], [0.0] * 60, w['weights'], uv=w['uv'], tex=w['tex'], vtx_color=w['vtx_color'])
eng.run(20); torch.cuda.synchronize()
lib = _lib.load(); lib.ddx_engine_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
assert lib.ddx_engine_trace_dump(eng.handle, b'/tmp/trace.bin') == 0
t = np.fromfile('/tmp/trace.bin', dtype=np.uint64).reshape(4, 8192 * 4)[1].reshape(4096, 8).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 7] > 0)]
names = ['prefetch + partial sums', 're-arm loop + lds sums', 'seg list term', 'tail (wave 0)', 'barrier', 'matrices', 'transform+stores']
print('workgroups:', len(t))
for i, n in enumerate(names):
    d = (t[:, i + 1] - t[:, i]) * 10
    print(f'{n:30s} mean {d.mean():7.0f} ns  p50 {np.median(d):7.0f}  p95 {np.percentile(d, 95):7.0f}')
print('total mean %.0f ns' % ((t[:, 7] - t[:, 0]).mean() * 10))
