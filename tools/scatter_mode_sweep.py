"""Measurement tool: which fragment variant of the rasteriser (scatter_resolve MODE: 0 plain, 2 hybrid exchange, 3 compacting) is fastest at
which coverage, BOTH FACES drawn (the default since round 6; the round-3 crossover was measured with back faces culled, i.e. at half
the fragments per triangle).     python tools/scatter_mode_sweep.py [config] [distances...]"""
import os, statistics, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from diffdope_amd import workloads as wl
    cfg, dist, cull = sys.argv[2], float(sys.argv[3]), bool(int(sys.argv[4]))
    w = wl.build(cfg, torch.device("cuda"), distance=dist)
    n, warm = 100, 20
    eng, p = wl.engine_for(w, wl.bench_lr_schedule(n + warm, "adam"), optimizer="adam", cull_backfaces=cull, single_stream=True)
    ts = []
    for _ in range(5):
        eng.new_observation(params=w["params0"])
        eng.run(warm); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.run(n); eng.finish(); ts.append((time.perf_counter() - t0) / n * 1e6)
    per_tri = 2.0 * w["coverage"] * w["H"] * w["W"] / w["T"]
    print(f"{cfg} d={dist} cull={int(cull)} mode={os.environ.get('DDX_SCATTER_MODE', 'auto')}: {statistics.median(ts):7.2f} us/it  (coverage {100 * w['coverage']:.2f} %, {per_tri:.2f} centres per triangle both faces)", flush=True)
    sys.exit(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dists = [float(a) for a in sys.argv[2:]] or [7.5, 5.0, 3.75, 2.5, 1.8]
for d in dists:
    for cull in (0, 1):
        for mode in ("auto", "0", "2", "3"):
            env = dict(os.environ)
            if mode != "auto":
                env["DDX_SCATTER_MODE"] = mode
            subprocess.run([sys.executable, __file__, "--one", cfg, str(d), str(cull)], env=env, stderr=subprocess.DEVNULL)
