import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import workloads as wl
dev = torch.device('cuda:0')
for cfg in ('cfg2', 'cfg50k64', 'cfg1'):
    w = wl.build(cfg, dev)
    lrs = wl.bench_lr_schedule(100, 'sgd')
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng, params = wl.engine_for(w, lrs, optimizer='sgd')
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.run(1); torch.cuda.synchronize(); t2 = time.perf_counter()
        eng.run(99); torch.cuda.synchronize(); t3 = time.perf_counter()
        print(f'{cfg} rep{rep}: create {1e3*(t1-t0):.1f} ms, first iteration (incl. setup) {1e3*(t2-t1):.1f} ms, 99 iterations {1e3*(t3-t2):.1f} ms')
        del eng
