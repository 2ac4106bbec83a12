"""Exploration tool (build with DDX_CXXFLAGS="-DDDX_TRACE -DDDX_PHASES -DDDX_PHASE_ROLE=1"): phase stamps of the mask role of
shade_kernel (wave 0 of every workgroup's first tile; phases separated by s_waitcnt 0)."""
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
order = [0, 1, 5, 2, 3, 4, 6]
names = ['halo staged', '(role-0 block skipped)', 'pairs detected + compacted', 'pair evaluation (trirec -> vertices -> unit gradient) + LDS deposit',
         'pixel phase (mask value, loss, d loss / d mask)', 'backward + reduction + store']
t = t[(t[:, 0] > 0) & (t[:, 6] > 0) & (t[:, 3] > 0)]
print('workgroups that reached the pair stage:', len(t))
for i, n in enumerate(names):
    d = (t[:, order[i + 1]] - t[:, order[i]]) * 10
    print(f'{n:75s} mean {d.mean():7.0f} ns  p50 {np.median(d):7.0f}  p95 {np.percentile(d, 95):7.0f}')
print('tile total mean %.0f ns' % ((t[:, 6] - t[:, 0]).mean() * 10))
