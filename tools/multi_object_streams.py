"""BASELINE config 5's per-GPU share -- 4 objects x 64 hypotheses -- three ways: the objects one after the other, one stream per
object, and ONE engine group (one launch of each kernel per iteration for all objects).  usage: [cfg5|cfg2|cfg3] [objects] [iterations]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 58
dev = torch.device("cuda")
ws = [wl.build(cfg, dev, seed=k) for k in range(n_obj)]
lrs = wl.bench_lr_schedule(n_it + 10, "adam")

SS = int(os.environ.get("SS", "0"))  # shade slices per hypothesis (0 = the engine's own choice from ITS batch size)
ES = int(os.environ.get("ES", "0"))
def engines():
    return [wl.engine_for(w, lrs, optimizer="adam", shade_slices=SS, edge_slices=ES) for w in ws]

def timeit(fn, reps=5):
    ts = []
    for _ in range(reps):
        es = engines()
        fn(es, 10)  # warm-up iterations (set-up included)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(es, n_it)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], es

def sequential(es, n):
    for e, _ in es: e.run(n)

streams = [torch.cuda.Stream() for _ in range(n_obj)]
def on_streams(es, n):
    main = torch.cuda.current_stream()
    for (e, _), st in zip(es, streams):
        st.wait_stream(main)
        with torch.cuda.stream(st): e.run(n)
    for st in streams: main.wait_stream(st)

held = [None, None]
def grouped(es, n):
    if held[0] is not es: held[0], held[1] = es, dd.RefineEngineGroup([e for e, _ in es])
    held[1].run(n)

B = ws[0]["B"]
base = None
for name, fn in (("one after the other", sequential), ("one stream per object", on_streams), ("one engine group", grouped)):
    dt, es = timeit(fn)
    base = base or dt
    print(f"{cfg}: {n_obj} objects x {B} hypotheses, {n_it} iterations, {name:22s}: {dt*1e3:7.2f} ms  = {dt/n_it*1e6:7.1f} us per iteration of all objects"
          f"  ({n_obj*B*n_it/dt/1e6:.2f} M hypothesis-iterations/s, x{base/dt:.2f})")
    finals = [p.clone() for _, p in es]
    if name == "one after the other": ref = finals
    else: assert all(torch.equal(a, b) for a, b in zip(ref, finals)), "results differ"
