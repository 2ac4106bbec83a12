"""Measurement tool: the two-chain run repeated -- N runs of 60 iterations from the same start on each workload, every one compared
bit for bit (parameters, loss log, pose log, status, selected hypothesis) with the first and with the one-chain engine."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffdope_amd import dist as ddist, workloads as wl

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
for name, B in (("cfg2", 64), ("cfg4", 32), ("cfg5", 64), ("cfg50k64", 64), ("cfg2", 128)):
    w = wl.build(name, dev, B=B)
    lrs = wl.bench_lr_schedule(60, "adam")
    ref = None
    bad = 0
    for single in (True, False):
        eng, p = wl.engine_for(w, lrs, optimizer="adam", single_stream=single)
        for k in range(1 if single else N):
            eng.new_observation(params=w["params0"])
            best = ddist.run_and_select(eng, 60)
            got = (p.clone(), eng.losses().clone(), eng.mtx_log.clone(), best[0], best[1], eng.check())
            if ref is None:
                ref = got
            else:
                same = all(torch.equal(a, b) for a, b in zip(got[:3], ref[:3])) and got[3:] == ref[3:]
                bad += int(not same)
    print(f"{name} B={B}: {N} two-chain runs of 60 iterations against the one-chain run: {bad} differ", flush=True)
