"""Randomised parity sweep (exploration tool; the fixed-seed cases live in tests/): random mesh sizes, frame sizes (not
multiples of the tile), distances down to the camera plane, loss sets, textured / vertex colours -- the fused engine's
losses and pose gradients against the oracle, and the op-level rasteriser's triangle ids bit for bit, with every scatter
variant.  Usage: python tools/fuzz_parity.py [n_cases] [first_seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffdope_amd as dd
from oracle import oracle as orc
from tests.scenes import make_scene

KEYS = ("rgb", "depth", "mask", "edge")
T = lambda a, **k: torch.tensor(np.ascontiguousarray(a), device="cuda", **k)
def sweep(n_cases=60, seed0=1000, verbose=True):
    """Runs the cases seed0 .. seed0 + n_cases - 1; returns (number of mismatching cases, statistics)."""
    bad = 0
    stats = dict(max_grad_err=0.0, outside=0, big=0, covered=0, empty=0)
    t_start = time.time()
    for case in range(n_cases):
        rng = np.random.RandomState(seed0 + case)
        rows, cols = int(rng.randint(3, 40)), int(rng.randint(4, 48))
        big = rng.rand() < 0.1
        H, W = int(rng.randint(20, 400 if big else 150)), int(rng.randint(24, 520 if big else 200))
        if os.environ.get("FUZZ_BIG"):  # dense meshes on large frames (the regime of the benchmark workloads): slower oracle, fewer cases
            rows, cols = int(rng.randint(30, 110)), int(rng.randint(40, 140))
            H, W = int(rng.randint(200, 500)), int(rng.randint(250, 660))
        dist = float(np.exp(rng.uniform(np.log(0.9), np.log(9.0))))
        textured = bool(rng.randint(2))
        B = int(rng.randint(1, 6)) if not os.environ.get("FUZZ_BIG") else int(rng.randint(2, 40))
        names = [k for k in KEYS if rng.rand() < 0.6] or ["mask"]
        weights = {k: float(rng.uniform(0.3, 1.5)) for k in names}
        variant = ("0", "1", "2", None)[case % 4]
        if variant is None: os.environ.pop("DDX_SCATTER_EXCHANGE", None)
        else: os.environ["DDX_SCATTER_EXCHANGE"] = variant
        tag = f"case {case} seed {seed0 + case}: mesh {rows}x{cols} frame {H}x{W} dist {dist:.2f} B {B} {'tex' if textured else 'vcol'} {sorted(weights)} scatter {variant}"
        try:
            sc = make_scene(rows, cols, H, W, B=B, dist=dist, textured=textured, seed=seed0 + case, rot_deg=float(rng.uniform(1, 25)), trans=float(rng.uniform(0, 0.1)))
            if rng.rand() < float(os.environ.get('FUZZ_NEAR', '0.25')):  # push one hypothesis towards / through the camera plane
                sc["params"][6, 0] = -float(rng.uniform(0.05, 0.6))
            R = sc["oracle"]
            R.weights = {k: weights.get(k) for k in KEYS}
            total, logs, g_ref, _ = R.loss_and_grad(sc["params"], sc["lr_mult"])
            tex = dict(uv=T(sc["uv"]), tex=T(sc["tex"])) if textured else dict(vtx_color=T(sc["vtx_color"]))
            params = T(sc["params"])
            eng = dd.RefineEngine(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [H, W], {k: T(v) for k, v in sc["gt"].items()}, params, T(sc["lr_mult"]), [0.1], weights, **tex)
            losses, grad = eng.loss_and_grad()
            torch.cuda.synchronize()
            st = eng.check()
            lg, gg = losses.cpu().numpy(), grad.cpu().numpy()
            scale = max(np.abs(g_ref).max(), 1e-6)  # (below: round-off of the oracle on sub-pixel triangles, the kernels give exact zeros)
            ok = np.isfinite(gg).all()
            for i, k in enumerate(KEYS):
                if k in logs: ok &= np.allclose(lg[i], logs[k], rtol=1e-4, atol=2e-7)
            gerr = np.abs(gg - g_ref).max() / scale
            ok &= gerr < 1e-2
            stats['max_grad_err'] = max(stats['max_grad_err'], float(gerr)); stats['outside'] += int(st['outside_view_volume'] > 0); stats['big'] += int(st['big_triangles'] > 0)
            stats['covered'] += int(sc['coverage'] > 0); stats['empty'] += int(np.abs(g_ref).max() == 0)
            # op-level ids, both faces (nvdiffrast semantics)
            clip = orc.xfm_fwd(sc["pos"][None].repeat(B, 0), np.matmul(sc["proj"][None], orc.pose_fwd(sc["params"])).astype(np.float32), True)
            ref = orc.rasterize_fwd(clip, sc["tri"], H, W)
            rast, _ = dd.rasterize(dd.RasterizeGLContext(), T(clip), T(sc["tri"]), [H, W])
            ids_ok = np.array_equal(rast[..., 3].cpu().numpy(), ref[..., 3])
            uvz = float(np.abs(rast[..., :3].cpu().numpy() - ref[..., :3]).max())
            # every 4th case: the materialising path (render_texture_batch, fused or op by op) -- images and autograd gradient
            mat_ok = True
            if case % 4 == 1 and not (sc["params"][6] > -0.7).any():
                R.cull_backfaces = False
                R.weights = dict(rgb=0.7, depth=1.0, mask=1.0)
                tot2, _, g2, r2 = R.loss_and_grad(sc["params"], sc["lr_mult"])
                R.cull_backfaces = True
                pl = [T(sc["params"][i], requires_grad=True) for i in range(7)]
                q = torch.stack(pl[:4], dim=0).T
                q = q / torch.norm(q, dim=1).reshape(-1, 1)
                mtx = dd.matrix_batch_44_from_position_quat(p=torch.stack(pl[4:], dim=0).T, q=q)
                ex = lambda a: T(a)[None].expand(B, *a.shape)
                kw = dict(uv=ex(sc["uv"]), uv_idx=ex(sc["tri"]), tex=ex(sc["tex"])) if textured else dict(vtx_color=ex(sc["vtx_color"]))
                out = dd.render_texture_batch(dd.RasterizeGLContext(), ex(sc["proj"]), mtx, ex(sc["pos"]), ex(sc["tri"]), [H, W], return_rast_out=True,
                                              fused=bool(case % 8 == 1), **kw)
                from diffdope_amd.render import masked_l1_mean
                gtt = {k: T(v)[None] for k, v in sc["gt"].items()}
                lrm = T(sc["lr_mult"])
                loss = 0.7 * (masked_l1_mean(out["rgb"], gtt["rgb"], gtt["segmentation"]) * lrm).mean()
                loss = loss + (masked_l1_mean(out["depth"], gtt["depth"], gtt["segmentation"], mask_channel0=True) * lrm).mean()
                loss = loss + (masked_l1_mean(out["mask"], gtt["segmentation"]) * lrm).mean()
                loss.backward()
                gm = np.stack([p_.grad.cpu().numpy() for p_ in pl])
                mat_ok = np.array_equal(out["rast_out"][..., 3].detach().cpu().numpy(), r2["rast"][..., 3])
                for k in ("rgb", "depth", "mask"):
                    mat_ok &= np.allclose(out[k].detach().cpu().numpy(), r2[k], rtol=1e-4, atol=5e-5)
                mat_ok &= abs(float(loss.detach()) - tot2) < 2e-5 * max(1, abs(tot2))
                mat_ok &= np.abs(gm - g2).max() < 1e-2 * max(np.abs(g2).max(), 1e-7)
                stats["materialising"] = stats.get("materialising", 0) + 1
            # every 4th case: three fused SGD iterations against the oracle's loop
            traj_ok = True
            if case % 4 == 2:
                lrs3 = [0.05, 0.04, 0.03]
                R.weights = {k: weights.get(k) for k in KEYS}
                p_ref, _ = R.optimise(sc["params"], sc["lr_mult"], lrs3)[:2]
                p3 = T(sc["params"])
                e3 = dd.RefineEngine(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [H, W], {k: T(v) for k, v in sc["gt"].items()}, p3, T(sc["lr_mult"]), lrs3, weights, **tex)
                e3.run(); e3.finish()
                d3 = np.abs(p3.cpu().numpy() - p_ref).max()
                traj_ok = bool(d3 < 2e-3)
                stats["trajectories"] = stats.get("trajectories", 0) + 1
                stats["max_traj_diff"] = max(stats.get("max_traj_diff", 0.0), float(d3))
            if not (mat_ok and traj_ok):
                bad += 1
                print("MISMATCH (materialising path)" if not mat_ok else "MISMATCH (trajectory)", tag, stats.get("max_traj_diff"))
            if not (ok and ids_ok and uvz < 1e-5):
                bad += 1
                print("MISMATCH", tag, "| grad err", gerr, "ids", ids_ok, "uvz", uvz, "status", st, "| max |g_ref|", float(np.abs(g_ref).max()), "max |g_gpu|", float(np.abs(gg).max()), "| losses", lg[:, 0], {k: v[0] for k, v in logs.items()})
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e))
    if verbose:
        print(f"{n_cases} cases, {bad} bad, {time.time() - t_start:.0f} s", stats)
    return bad, stats


if __name__ == "__main__":
    sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
