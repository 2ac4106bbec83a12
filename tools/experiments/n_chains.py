"""Experiment (round 6): would MORE than two chains of smaller launches hide more of a launch's latency floor?  cfg2's 64 hypotheses as
N engines of 64 / N hypotheses each (same global batch, the slots per hypothesis of the full launch: DDX_STEP_RESIDENT), one chain
each, every engine on its own stream and enqueued by its own host thread, against the product's one engine (one chain / two chains).
    python tools/experiments/n_chains.py [n_iters]"""
import os, sys, threading, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffdope_amd import workloads as wl

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
warm = 20
dev = torch.device("cuda:0")
w = wl.build("cfg2", dev)
B = w["B"]
lrs = wl.bench_lr_schedule(n + warm, "adam")


def product(single):
    eng, _ = wl.engine_for(w, lrs, optimizer="adam", single_stream=single)
    ts = []
    for _ in range(5):
        eng.new_observation(params=w["params0"])
        eng.run(warm); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.run(n); eng.finish(); ts.append((time.perf_counter() - t0) / n * 1e6)
    return statistics.median(ts)


print(f"one engine, one chain : {product(True):7.2f} us/it", flush=True)
print(f"one engine, two chains: {product(False):7.2f} us/it", flush=True)
for N in (2, 4, 8):
    nb = B // N
    os.environ["DDX_STEP_RESIDENT"] = str(20 * nb)
    engs, streams = [], []
    for k in range(N):
        wk = dict(w)
        wk["params0"] = w["params0"][:, k * nb:(k + 1) * nb].contiguous()
        wk["lr_mult"] = w["lr_mult"][k * nb:(k + 1) * nb].contiguous()
        wk["B"] = nb
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            e, _ = wl.engine_for(wk, lrs, optimizer="adam", single_stream=True, global_batch=B)
            e.run(2); e.finish()
        engs.append((e, wk)); streams.append(s)
    ts = []
    for _ in range(5):
        for (e, wk), s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.new_observation(params=wk["params0"]); e.run(warm)
        torch.cuda.synchronize()
        bar = threading.Barrier(N + 1)

        def work(e, s):
            with torch.cuda.stream(s):
                bar.wait()
                e.run(n)
                s.synchronize()
        th = [threading.Thread(target=work, args=(e, s)) for (e, _), s in zip(engs, streams)]
        for t in th: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        ts.append((time.perf_counter() - t0) / n * 1e6)
    print(f"{N} engines of {nb} hypotheses, one stream and one host thread each: {statistics.median(ts):7.2f} us/it (min {min(ts):.2f})", flush=True)
    del engs
