"""How much does the order of the mesh file matter?  cfg2 / cfg50k64 as built (row-major grid), with the triangle list
shuffled, and with vertex list + triangle list shuffled (the worst a mesh file can do).  The engine keeps an internal copy
with vertices renumbered and triangles processed in Morton order, so the three should run alike."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffdope_amd as dd
from diffdope_amd import workloads as wl
for cfg in ("cfg2", "cfg50k64"):
    for mode in ("as built", "triangles shuffled", "vertices + triangles shuffled"):
        w = wl.build(cfg, torch.device("cuda:0"))
        rng = np.random.RandomState(0)
        if mode != "as built":
            tri = w["tri"].cpu().numpy()
            tri = tri[rng.permutation(len(tri))]
            if mode.startswith("vertices"):
                V = w["pos"].shape[0]
                pv = rng.permutation(V)            # new position of old vertex v
                inv = np.empty(V, np.int64); inv[pv] = np.arange(V)
                w["pos"] = w["pos"][torch.tensor(inv, device="cuda")]
                w["uv"] = w["uv"][torch.tensor(inv, device="cuda")]
                tri = pv[tri]
            w["tri"] = torch.tensor(np.ascontiguousarray(tri), dtype=torch.int32, device="cuda")
        lrs = [0.005 * l / 2.0 for l in wl.lr_schedule(219, 20, 0.1)]
        p = w["params0"].clone()
        eng = dd.RefineEngine(w["pos"], w["tri"], w["proj"], [w["H"], w["W"]], w["gt"], p, w["lr_mult"], lrs, w["weights"], uv=w["uv"], tex=w["tex"], optimizer="adam")
        eng.run(20); torch.cuda.synchronize(); t0 = time.perf_counter(); eng.run(200); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        rot, tr = wl.pose_errors(p, w["q_gt"], w["t_gt"])
        print(f"{cfg:9s} {mode:30s} {1 / dt:8.0f} it/s   best rotation error {float(rot.min()):.2e} rad")
