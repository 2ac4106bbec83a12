"""Experiment (round 4): the B = 64 iteration of cfg2 as (a) the product's run, (b) the same without any tile pass (the floor of what removing the launch can give),
(c) two 32-hypothesis halves in phase-staggered MIXED launches (step blocks of one half + shade blocks of the other in one grid),
(d) the two halves as independent chains on two streams, each sized for half the chip.  Needs the variant library:
    python tools/build_variant.py exp -DDDX_EXPERIMENTS ;  DDX_LIB=diffdope_amd/libddx_exp.so python tools/experiments/exp_stagger.py [cfg2]
Every variant's result is compared bit for bit with the unsplit run (a half with B_global = 64 and the unsplit slice counts is
shard-invariant)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DDX_LIB", os.path.join(ROOT, "diffdope_amd", "libddx_exp.so"))
from diffdope_amd import _lib, workloads as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dist = float(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
P, I = ctypes.c_void_p, ctypes.c_int
raw.ddx_exp_pair_staggered.argtypes = [P, P, I, I, I, I, P]
raw.ddx_exp_pair_streams.argtypes = [P, P, I, I, P, P]
raw.ddx_exp_run_no_big.argtypes = [P, I, I, P]
WARM, N, REPS = 20, 200, 5
Btot = wl.CONFIGS[cfg]["B"]
lrs = wl.bench_lr_schedule(WARM + N, "adam")
wfull = wl.build(cfg, dev, distance=dist)
halves = [wl.build(cfg, dev, B=Btot // 2, global_lo=k * (Btot // 2), global_B=Btot, distance=dist) for k in range(2)]

def full_engine():
    return wl.engine_for(wfull, lrs, optimizer="adam")

e0, p0 = full_engine()
SS, ES = e0.slices
def half_engines():
    return [wl.engine_for(w, lrs, optimizer="adam", global_batch=Btot, shade_slices=SS, edge_slices=ES) for w in halves]

def check(name, st):
    assert st["big_triangles"] == 0 and st["overflow"] == 0, (name, st)

def timeit(name, make, run, result):
    ts = []
    for _ in range(REPS):
        state = make()
        run(state, 0, WARM)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(state, WARM, N)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{name:64s} {med / N * 1e6:7.2f} us/iteration  {N / med:8.0f} it/s   (min {ts[0] / N * 1e6:.2f})", flush=True)
    return result(state)

s_cur = lambda: torch.cuda.current_stream().cuda_stream
def run_full(st, it0, n):
    st[0].rewind(it0); st[0].run(n)
ref = timeit(f"{cfg}: product run (step, shade with the tile-pass worker slab)", full_engine, run_full, lambda st: (st[1].clone(), st[0].loss_log.clone()))
def run_nobig(st, it0, n):
    _lib.check(raw.ddx_exp_run_no_big(st[0].handle, it0, n, s_cur()), "no_big")
got = timeit(f"{cfg}: without any tile pass (neither launch nor worker slab)", full_engine, run_nobig, lambda st: (st[1].clone(), st[0].loss_log.clone(), st[0].status()))
check("no_big", got[2])
print("   bit-identical to the product run:", torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]))

def halves_result(st):
    p = torch.cat([st[0][1], st[1][1]], dim=1)
    l = torch.cat([st[0][0].loss_log, st[1][0].loss_log], dim=2)
    for e, _ in st: check("half", e.status())
    return p, l

for SL in (10, 14, 20, 40):
    for step_first in (1, 0):
        def run_st(st, it0, n, SL=SL, sf=step_first):
            _lib.check(raw.ddx_exp_pair_staggered(st[0][0].handle, st[1][0].handle, it0, n, SL, sf, s_cur()), "staggered")
        got = timeit(f"{cfg}: 2 x {Btot // 2} staggered in mixed launches, {SL} slots, {'step' if step_first else 'shade'} blocks first", half_engines, run_st, halves_result)
        print("   bit-identical to the product run:", torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]))

streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for resident, grid, nobig in ((0, 0, 0), (640, 0, 0), (768, 0, 0), (640, 0, 1), (1024, 0, 1)):
    if resident: os.environ["DDX_STEP_RESIDENT"] = str(resident)
    else: os.environ.pop("DDX_STEP_RESIDENT", None)
    if nobig: os.environ["DDX_EXP_NO_BIG"] = "1"
    else: os.environ.pop("DDX_EXP_NO_BIG", None)
    def run_2s(st, it0, n):
        main = torch.cuda.current_stream()
        for s in streams: s.wait_stream(main)
        _lib.check(raw.ddx_exp_pair_streams(st[0][0].handle, st[1][0].handle, it0, n, streams[0].cuda_stream, streams[1].cuda_stream), "streams")
        for s in streams: main.wait_stream(s)
    got = timeit(f"{cfg}: 2 x {Btot // 2} on two streams, step_resident {resident or 'auto'}{', no tile-pass launch' if nobig else ''}", half_engines, run_2s, halves_result)
    print("   bit-identical to the product run:", torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]))
os.environ.pop("DDX_STEP_RESIDENT", None); os.environ.pop("DDX_EXP_NO_BIG", None)
