import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import diffdope_amd as dd
from diffdope_amd import workloads as wl, dist as ddist
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
w = wl.build('cfg2', dev)
n_it=220
lrs = [0.005 * l / 2.0 for l in wl.lr_schedule(n_it - 1, 20, 0.1)]
for trial in range(3):
    params = w["params0"].clone()
    eng = dd.RefineEngine(w["pos"], w["tri"], w["proj"], [w["H"], w["W"]], w["gt"], params, w["lr_mult"], lrs, w["weights"], uv=w["uv"], tex=w["tex"], optimizer='adam')
    eng.run(20); torch.cuda.synchronize()
    t0=time.perf_counter(); eng.run(200); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    last = eng.loss_log[n_it - 1]; per_hyp = last[[0,2]].mean(0)
    g = ddist.global_argmin(per_hyp, eng.mtx_log[n_it - 1].reshape(64, 4, 4), lo=0); torch.cuda.synchronize(); t3=time.perf_counter()
    print('launch %.1f ms, sync %.1f ms, argmin %.1f ms' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
