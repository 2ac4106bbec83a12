import sys, time, torch, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
w = wl.build(name, dev)
print('workload', name, 'V', w['V'], 'T', w['T'], 'coverage', w['coverage'])
def mk(lrs, optimizer='sgd', weights=None):
    p = w['params0'].clone()
    eng = dd.RefineEngine(w['pos'], w['tri'], w['proj'], [w['H'], w['W']], w['gt'], p, w['lr_mult'], lrs, weights or w['weights'],
                          uv=w['uv'], tex=w['tex'], vtx_color=w['vtx_color'], optimizer=optimizer)
    return eng, p
r0, t0 = wl.pose_errors(w['params0'], w['q_gt'], w['t_gt'])
print('init err: rot med %.4f max %.4f | trans med %.4f max %.4f' % (np.median(r0), r0.max(), np.median(t0), t0.max()))
for opt, scale in [('sgd', 1.0), ('sgd', 0.3), ('sgd', 0.1), ('adam', 0.005), ('adam', 0.002)]:
    n = 220
    lrs = [scale * l / 2.0 for l in wl.lr_schedule(n - 1, 20, 0.1)] if opt == 'sgd' else [scale * l / 2.0 for l in wl.lr_schedule(n - 1, 20, 0.1)]
    eng, p = mk(lrs, opt)
    eng.run(20); torch.cuda.synchronize()
    t = time.time(); eng.run(200); torch.cuda.synchronize(); dt = time.time() - t
    st = eng.check()
    lg = eng.losses().cpu().numpy()
    r, tr = wl.pose_errors(p, w['q_gt'], w['t_gt'])
    best = int(np.argmin(lg[-1].sum(0)))
    print(f'{opt} scale {scale}: {200/dt:.1f} it/s ({dt/200*1e3:.3f} ms/it) status {st} | loss first {lg[0].sum(0).mean():.5f} last {lg[-1].sum(0).mean():.5f} | best hyp {best}: rot {r[best]:.5f} trans {tr[best]:.5f} | med rot {np.median(r):.4f} trans {np.median(tr):.4f}')
eng, p = mk([0.1] * 40)
print({k: round(v * 1e3, 2) for k, v in eng.profile(0, 20).items()}, 'us per kernel')
