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

# which rasteriser rule the engine-vs-oracle cases run under: FUZZ_CULL=0 (default) both faces, dr.rasterize's rule and the engine's
# default; FUZZ_CULL=1 deviation D5 (back faces of closed meshes culled), engine and oracle alike
CULL = bool(int(__import__("os").environ.get("FUZZ_CULL", "0")))
CULL_ENG = CULL


def _RE(*a, **k):
    import diffdope_amd as _dd

    k.setdefault("cull_backfaces", CULL_ENG)
    return _dd.RefineEngine(*a, **k)


KEYS = ("rgb", "depth", "mask", "edge")
T = lambda a, **k: torch.tensor(np.ascontiguousarray(a), device="cuda", **k)
def sweep(n_cases=60, seed0=1000, verbose=True):
    """Runs the cases seed0 .. seed0 + n_cases - 1; returns (number of mismatching cases, statistics)."""
    bad = 0
    stats = dict(max_grad_err=0.0, outside=0, big=0, covered=0, empty=0)
    t_start = time.time()
    first = int(os.environ.get("FUZZ_FIRST", "0"))  # (reproduce one reported case: FUZZ_FIRST=<case> with the sweep's own seed0 and n = case + 1)
    for case in range(first, n_cases):
        rng = np.random.RandomState(seed0 + case)
        rows, cols = int(rng.randint(3, 40)), int(rng.randint(4, 48))
        big = rng.rand() < 0.1
        H, W = int(rng.randint(20, 400 if big else 150)), int(rng.randint(24, 520 if big else 200))
        if os.environ.get("FUZZ_BIG"):  # dense meshes on large frames (the regime of the benchmark workloads): slower oracle, fewer cases
            rows, cols = int(rng.randint(30, 110)), int(rng.randint(40, 140))
            H, W = int(rng.randint(200, 500)), int(rng.randint(250, 660))
        dist = float(np.exp(rng.uniform(np.log(0.9), np.log(9.0))))
        if os.environ.get("FUZZ_FAR"):  # out to and beyond the far plane (zfar = 200): specks, far clipping per fragment
            dist = float(np.exp(rng.uniform(np.log(5.0), np.log(260.0))))
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
            R.cull_backfaces = CULL
            if textured and rng.rand() < 0.5:  # any texture size: non-square, not a power of two, down to 1 x 1
                sc["tex"] = rng.uniform(size=(int(rng.randint(1, 70)), int(rng.randint(1, 70)), 3)).astype(np.float32)
                R = orc.RenderOracle(sc["pos"], sc["tri"], sc["proj"], H, W, {}, dict(rgb=0.7, depth=1.0, mask=1.0), dtype=np.float32, cull_backfaces=CULL,
                                     uv=sc["uv"], tex=sc["tex"])
                R.gt = {k: v[None] for k, v in sc["gt"].items()}
                stats["odd_textures"] = stats.get("odd_textures", 0) + 1
            G = B + int(rng.randint(0, 20)) if rng.rand() < 0.3 else B  # the hypotheses are a shard of a larger global batch
            R.weights = {k: weights.get(k) for k in KEYS}
            total, logs, g_ref, _ = R.loss_and_grad(sc["params"], sc["lr_mult"], global_B=G)
            tex = dict(uv=T(sc["uv"]), tex=T(sc["tex"])) if textured else dict(vtx_color=T(sc["vtx_color"]))
            params = T(sc["params"])
            eng = _RE(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [H, W], {k: T(v) for k, v in sc["gt"].items()}, params, T(sc["lr_mult"]), [0.1], weights, global_batch=G, **tex)
            losses, grad = eng.loss_and_grad()
            torch.cuda.synchronize()
            st = eng.check()
            lg, gg = losses.cpu().numpy(), grad.cpu().numpy()
            scale = max(np.abs(g_ref).max(), 1e-6)  # (below: round-off of the oracle on sub-pixel triangles, the kernels give exact zeros)
            ok = np.isfinite(gg).all()
            for i, k in enumerate(KEYS):
                if k in logs: ok &= np.allclose(lg[i], logs[k], rtol=1e-4, atol=2e-7)
            gerr = np.abs(gg - g_ref).max() / scale
            if gerr >= 1e-2 and np.abs(g_ref).max() < 1e-6:
                # a gradient that is zero in exact arithmetic (a hypothesis through the camera plane whose object fills the frame): the
                # float32 oracle's op-by-op backward leaves round-off of 1e-8..2e-8 where the kernels give exact zeros (seeds 5005595,
                # 5100781, 5102773: its own float64 run gives 1e-17) -- the referee is then the float64 oracle
                kw64 = dict(uv=sc["uv"], tex=sc["tex"]) if textured else dict(vtx_color=sc["vtx_color"])
                R64 = orc.RenderOracle(sc["pos"], sc["tri"], sc["proj"], H, W, {}, dict(R.weights), dtype=np.float64, cull_backfaces=CULL, **kw64)
                R64.gt = {k: v.astype(np.float64) for k, v in R.gt.items()}
                R64.weights = R.weights
                g64 = R64.loss_and_grad(sc["params"].astype(np.float64), sc["lr_mult"].astype(np.float64), global_B=G)[2]
                gerr = np.abs(gg - g64).max() / max(np.abs(g64).max(), 1e-6)
                stats["f64_referee"] = stats.get("f64_referee", 0) + 1
            ok &= gerr < 1e-2
            if gerr > 2e-3 and verbose:
                print("note: gradient error", float(gerr), tag, "| max |g_ref|", float(np.abs(g_ref).max()))
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
                tot2, _, g2, r2 = R.loss_and_grad(sc["params"], sc["lr_mult"])  # (global batch = B: the torch expressions below)
                R.cull_backfaces = CULL
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
                # (ids: the matrices of this path come from the device's pose op and torch's matmul, the oracle's from numpy: a last-bit
                # difference in a clip coordinate can hand an exact-edge pixel to the neighbouring triangle -- seed 5001213, one pixel)
                id_diff = out["rast_out"][..., 3].detach().cpu().numpy() != r2["rast"][..., 3]
                n_id = int(id_diff.sum())
                mat_ok = n_id <= 2
                # (round 4, seeds 7057321 / 8015421 / 8022537, FUZZ_MAT_DIAG: 10-11 of the 16 entries of proj @ pose differ in the last
                # bit between the two sides, ONE pixel goes to the other triangle or to the background, and the oracle's rasteriser fed
                # the device path's own clip coordinates reproduces the device's ids exactly.  Such a pixel -- and its neighbours, which
                # the antialias blends with it -- is a different INPUT, not a different result: left out of the image comparison, and
                # the loss may differ by what that many pixels can contribute)
                excl = np.zeros_like(id_diff)
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        excl |= np.roll(np.roll(id_diff, dy, 1), dx, 2)
                for k in ("rgb", "depth", "mask"):
                    # (the clip-space vertices come from torch's proj @ mtx here, the oracle's from numpy's: last-bit differences that
                    # a sliver pixel's barycentrics amplify -- seed 702717: one pixel off by 1.6e-4; hence the median-tight, max-loose pair)
                    dk = np.abs(out[k].detach().cpu().numpy() - r2[k])
                    dk[excl] = 0.0
                    mat_ok &= bool(dk.max() < 2e-3 * max(1.0, float(np.abs(r2[k]).max())) and np.percentile(dk, 99.9) < 2e-4 * max(1.0, float(np.abs(r2[k]).max())))
                loss_slack = 9 * n_id * 8.0 * max(1.0, float(np.abs(r2["depth"]).max())) / (H * W)
                mat_ok &= abs(float(loss.detach()) - tot2) < 2e-5 * max(1, abs(tot2)) + loss_slack
                mat_ok &= np.abs(gm - g2).max() < 1e-2 * max(np.abs(g2).max(), 1e-7)
                stats["materialising"] = stats.get("materialising", 0) + 1
                if not mat_ok and os.environ.get("FUZZ_MAT_DIAG"):
                    # which criterion failed, and are the two sides even rasterising the same clip coordinates?
                    ids_d, ids_o = out["rast_out"][..., 3].detach().cpu().numpy(), r2["rast"][..., 3]
                    print("  diag", tag)
                    print("  diag ids differ at", int((ids_d != ids_o).sum()), "pixels of", ids_o.size)
                    for k in ("rgb", "depth", "mask"):
                        dk = np.abs(out[k].detach().cpu().numpy() - r2[k]); sk = max(1.0, float(np.abs(r2[k]).max()))
                        print(f"  diag {k}: max {dk.max():.3e} (limit {2e-3 * sk:.1e}), 99.9 % {np.percentile(dk, 99.9):.3e} (limit {2e-4 * sk:.1e}), pixels over the max limit {int((dk.reshape(dk.shape[0], H, W, -1).max(-1) >= 2e-3 * sk).sum())}")
                    print(f"  diag loss {float(loss.detach()):.8f} vs {tot2:.8f}; gradient rel {np.abs(gm - g2).max() / max(np.abs(g2).max(), 1e-7):.3e}")
                    m_o = np.matmul(sc["proj"][None], orc.pose_fwd(sc["params"])).astype(np.float32)
                    m_d = torch.matmul(T(sc["proj"])[None], mtx.detach()).cpu().numpy()
                    c_o = orc.xfm_fwd(sc["pos"][None].repeat(B, 0), m_o, True)
                    c_d = orc.xfm_fwd(sc["pos"][None].repeat(B, 0), m_d, True)
                    print(f"  diag proj @ pose: {int((m_o != m_d).sum())} of {m_o.size} entries differ (max {np.abs(m_o - m_d).max():.2e}); clip coordinates: "
                          f"{int((c_o != c_d).sum())} of {c_o.size} differ (max {np.abs(c_o - c_d).max():.2e})")
                    ref_d = orc.rasterize_fwd(c_d, sc["tri"], H, W)
                    print("  diag oracle rasteriser on the DEVICE path's clip coordinates: ids differ from the device's at", int((ref_d[..., 3] != ids_d).sum()), "pixels")
            # every 4th case: three fused SGD iterations, the oracle teacher-forced on the engine's own parameters before each (a free-
            # running comparison measures the chaos of L1 sign flips under large steps, not the kernels: seed 203758 differs by 3e-3
            # after three steps although every single gradient agrees to 3e-7).  State carried between iterations -- zbuf / tile-flag
            # re-arm, double-buffered matrices -- is what this adds over the single evaluation above.
            traj_ok = True
            if case % 4 == 2:
                lrs3 = [0.05, 0.04, 0.03]
                R.weights = {k: weights.get(k) for k in KEYS}
                p3 = T(sc["params"])
                e3 = _RE(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [H, W], {k: T(v) for k, v in sc["gt"].items()}, p3, T(sc["lr_mult"]), lrs3, weights, global_batch=G, **tex)
                for it3, lr3 in enumerate(lrs3):
                    before = p3.cpu().numpy().copy()
                    e3.run(1); e3.finish()
                    after = p3.cpu().numpy()
                    _, lg3, g3, _ = R.loss_and_grad(before, sc["lr_mult"], global_B=G)
                    # the step itself is compared (the gradient recovered from a float32 parameter difference is quantised at
                    # 6e-8 |param| / lr): within 1 % of the largest step plus that round-off
                    err = np.abs(after - (before - np.float32(lr3) * g3)).max()
                    d3 = float(err / (1e-2 * lr3 * max(np.abs(g3).max(), 1e-6) + 4e-7 * max(1.0, np.abs(before).max())))
                    stats["max_traj_diff"] = max(stats.get("max_traj_diff", 0.0), d3)
                    row = e3.losses()[it3].cpu().numpy()
                    for i, k in enumerate(KEYS):
                        if k in lg3: traj_ok &= bool(np.allclose(row[i], lg3[k], rtol=1e-4, atol=2e-7))
                    traj_ok &= d3 < 1.0  # (in units of the tolerance)
                stats["trajectories"] = stats.get("trajectories", 0) + 1
            if not (mat_ok and traj_ok):
                bad += 1
                print("MISMATCH (materialising path)" if not mat_ok else "MISMATCH (trajectory)", tag)
            if not (ok and ids_ok and uvz < 1e-5):
                bad += 1
                print("MISMATCH", tag, "| grad err", gerr, "ids", ids_ok, "uvz", uvz, "status", st, "| max |g_ref|", float(np.abs(g_ref).max()), "max |g_gpu|", float(np.abs(gg).max()), "| losses", lg[:, 0], {k: v[0] for k, v in logs.items()})
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e))
    if verbose:
        print(f"{n_cases} cases, {bad} bad, {time.time() - t_start:.0f} s", stats)
    return bad, stats


def sweep_soups(n_cases=200, seed0=0, verbose=True):
    """Random triangle soups in clip space through the op-level rasteriser (ids bit for bit, u / v / z-w to 2e-6) and its
    backward, antialias with random colours and interpolate forward / backward against the oracle: sizes from sub-pixel to far
    beyond the frame, w from 1e-5 to 10 and negative, z inside / outside [-w, w], shared and degenerate triangles."""
    bad = 0
    stats = dict(drawn=0, straddlers=0, near_eye=0, max_uvz=0.0)
    t_start = time.time()
    for case in range(n_cases):
        rng = np.random.RandomState(seed0 + case)
        H, W = int(rng.randint(1, 200)), int(rng.randint(1, 260))
        n_tri = int(rng.randint(1, 400))
        B = int(rng.randint(1, 4))
        nv = n_tri * 3
        cx, cy = rng.uniform(-0.5 * W, 1.5 * W, n_tri), rng.uniform(-0.5 * H, 1.5 * H, n_tri)
        size = np.exp(rng.uniform(np.log(0.2), np.log(4.0 * max(H, W, 8)), n_tri))
        ang = rng.uniform(0, 2 * np.pi, (n_tri, 3))
        rad = size[:, None] * rng.uniform(0.1, 1.0, (n_tri, 3))
        px, py = cx[:, None] + rad * np.cos(ang), cy[:, None] + rad * np.sin(ang)
        snap = rng.rand(n_tri) < 0.15
        px[snap] = np.round(px[snap] * 2) / 2
        py[snap] = np.round(py[snap] * 2) / 2
        z = np.round(rng.uniform(-0.9, 0.9, (n_tri, 1)) * 8) / 8 + rng.uniform(-0.05, 0.05, (n_tri, 3)) * (rng.rand(n_tri, 1) < 0.7)
        w = np.exp(rng.uniform(np.log(0.3), np.log(10.0), (n_tri, 3)))
        pos = np.zeros((B, nv, 4), np.float32)
        for b in range(B):
            sh = 0.37 * b
            ww = w.reshape(-1).copy()
            pos[b, :, 0], pos[b, :, 1] = ((px + sh) / W * 2 - 1).reshape(-1) * ww, ((py - sh) / H * 2 - 1).reshape(-1) * ww
            pos[b, :, 2], pos[b, :, 3] = z.reshape(-1) * ww, ww
        behind = rng.rand(nv) < 0.04
        pos[:, behind, 3] *= -1.0
        far = rng.rand(nv) < 0.03
        pos[:, far, 2] = 1.7 * pos[:, far, 3]
        eye = rng.rand(nv) < 0.04  # just in front of the eye plane, behind the near plane: projects far outside the guard band
        pos[:, eye, 3] = (10.0 ** rng.uniform(-5, -2, eye.sum())).astype(np.float32)
        pos[:, eye, 2] = -np.abs(pos[:, eye, 2]) - 0.01
        tri = np.arange(nv, dtype=np.int32).reshape(n_tri, 3)
        share = rng.rand(n_tri) < 0.3
        tri[share, 0] = tri[rng.randint(0, n_tri, share.sum()), 1]
        dup = rng.rand(n_tri) < 0.05
        tri[dup] = tri[rng.randint(0, n_tri, dup.sum())]
        deg = rng.rand(n_tri) < 0.03
        tri[deg, 2] = tri[deg, 1]
        tag = f"soup {case} seed {seed0 + case}: {n_tri} triangles, frame {H}x{W}, B {B}"
        try:
            ref = orc.rasterize_fwd(pos, tri, H, W)
            pos_t = T(pos, requires_grad=True)
            tri_t = T(tri)
            rast, _ = dd.rasterize(dd.RasterizeGLContext(), pos_t, tri_t, [H, W])
            got = rast.detach().cpu().numpy()
            ids_ok = np.array_equal(got[..., 3], ref[..., 3])
            uvz = float(np.abs(got[..., :3] - ref[..., :3]).max())
            ok = ids_ok and uvz < 3e-6
            stats["max_uvz"] = max(stats["max_uvz"], uvz)
            stats["drawn"] += int((ref[..., 3] > 0).sum())
            wt = pos[0, tri.reshape(-1), 3].reshape(-1, 3)
            stats["straddlers"] += int(((wt <= 0).any(1) & (wt > 0).any(1)).sum())
            stats["near_eye"] += int(eye.sum())
            if ok and case % 2 == 0:
                # backward of rasterize, and antialias / interpolate on this visibility
                g = rng.normal(size=ref.shape).astype(np.float32)
                rast.backward(T(g))
                dref = orc.rasterize_bwd(pos, tri, ref, g)
                ok &= bool(np.abs(pos_t.grad.cpu().numpy() - dref).max() <= 5e-3 * max(np.abs(dref).max(), 1e-6))
                rd = rast.detach()
                col = rng.uniform(size=(B, H, W, 3)).astype(np.float32)
                c_t, p_t = T(col, requires_grad=True), T(pos, requires_grad=True)
                out = dd.antialias(c_t, rd, p_t, tri_t)
                oref = orc.antialias_fwd(col, ref, pos, tri)
                ok &= bool(np.abs(out.detach().cpu().numpy() - oref).max() < 5e-5)
                go = rng.normal(size=oref.shape).astype(np.float32)
                out.backward(T(go))
                dcol, dpos = orc.antialias_bwd(col, ref, pos, tri, go)
                ok &= bool(np.abs(c_t.grad.cpu().numpy() - dcol).max() < 2e-4)
                dpe = np.abs(p_t.grad.cpu().numpy() - dpos).max()
                if dpe > 1e-2 * max(np.abs(dpos).max(), 1e-6):
                    # a vertex at w ~ 1e-5 turns the 1/w^2 of the w-gradient into a conditioning problem of the float32
                    # arithmetic itself (seed 961292: the float32 oracle is 2.07 from its own float64 run, the kernel 0.69): the
                    # referee is then the float64 oracle, and the kernel may be as far from it as the float32 oracle is
                    d64 = orc.antialias_bwd(col.astype(np.float64), ref.astype(np.float64), pos.astype(np.float64), tri, go.astype(np.float64))[1]
                    ok &= bool(np.abs(p_t.grad.cpu().numpy() - d64).max() <= max(1e-2 * np.abs(d64).max(), 1.5 * np.abs(dpos - d64).max()))
                    stats["f64_referee"] = stats.get("f64_referee", 0) + 1
                attr = rng.normal(size=(B, nv, 3)).astype(np.float32)
                a_t, r_t = T(attr, requires_grad=True), rd.clone().requires_grad_(True)
                io, _ = dd.interpolate(a_t, r_t, tri_t)
                iref = orc.interpolate_fwd(attr, ref, tri)
                ok &= bool(np.abs(io.detach().cpu().numpy() - iref).max() < 2e-5)
                gi = rng.normal(size=iref.shape).astype(np.float32)
                io.backward(T(gi))
                dattr, drast = orc.interpolate_bwd(attr, ref, tri, gi, True)
                ok &= bool(np.abs(r_t.grad.cpu().numpy() - drast).max() < 2e-4 * max(1.0, np.abs(drast).max()))
                ok &= bool(np.abs(a_t.grad.cpu().numpy() - dattr).max() < 2e-3 * max(1.0, np.abs(dattr).max()))
            if not ok:
                bad += 1
                print("MISMATCH", tag, "ids", ids_ok, "uvz", uvz, "pixels with other id", int((got[..., 3] != ref[..., 3]).sum()))
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e))
    if verbose:
        print(f"{n_cases} soups, {bad} bad, {time.time() - t_start:.0f} s", stats)
    return bad, stats


def sweep_engine_soups(n_cases=200, seed0=0, verbose=True):
    """The fused engine on arbitrary object-space triangle soups (open, self-intersecting, with degenerate, duplicated and
    unreferenced parts, sizes from specks to triangles larger than the frame, shared vertices): losses and pose gradients
    against the oracle.  No culling applies (not a closed surface); straddlers, the tile pass and depth ties all occur."""
    bad = 0
    stats = dict(max_grad_err=0.0, outside=0, big=0)
    t_start = time.time()
    for case in range(n_cases):
        rng = np.random.RandomState(seed0 + case)
        H, W = int(rng.randint(8, 160)), int(rng.randint(8, 220))
        n_tri = int(rng.randint(1, 250))
        nv = int(rng.randint(3, 3 * n_tri + 1))
        B = int(rng.randint(1, 5))
        ext = float(np.exp(rng.uniform(np.log(0.05), np.log(1.5))))
        pos = rng.normal(size=(nv, 3)).astype(np.float32) * np.float32(0.35)
        tri = rng.randint(0, nv, size=(n_tri, 3)).astype(np.int32)
        local = rng.rand(n_tri) < 0.7  # most triangles small: corners near the first one
        near = np.clip(tri[:, :1] + rng.randint(-4, 5, size=(n_tri, 3)), 0, nv - 1)
        tri[local] = near[local]
        pos = (pos * ext).astype(np.float32)
        if os.environ.get("FUZZ_NO_DEGENERATE"):
            keep = (tri[:, 0] != tri[:, 1]) & (tri[:, 1] != tri[:, 2]) & (tri[:, 0] != tri[:, 2])
            tri = tri[keep] if keep.any() else np.array([[0, 1, 2]], np.int32)
            n_tri = len(tri)
        textured = bool(rng.randint(2))
        uv = rng.uniform(-0.5, 1.5, size=(nv, 2)).astype(np.float32)
        tex = rng.uniform(size=(int(rng.randint(1, 40)), int(rng.randint(1, 40)), 3)).astype(np.float32)
        vcol = rng.uniform(size=(nv, 3)).astype(np.float32)
        from diffdope_amd import synthetic as syn
        proj = orc.projection_matrix(**syn.camera_intrinsics(W, H)).astype(np.float32)
        dist = float(np.exp(rng.uniform(np.log(0.4), np.log(6.0))))
        names = [k for k in KEYS if rng.rand() < 0.6] or ["depth"]
        weights = {k: float(rng.uniform(0.3, 1.5)) for k in names}
        kw = dict(uv=uv, tex=tex) if textured else dict(vtx_color=vcol)
        tag = f"engine soup {case} seed {seed0 + case}: {n_tri} triangles / {nv} vertices, frame {H}x{W}, dist {dist:.2f}, B {B}, {'tex' if textured else 'vcol'} {sorted(weights)}"
        try:
            R = orc.RenderOracle(pos, tri, proj, H, W, {}, dict(rgb=1.0, depth=1.0, mask=1.0), dtype=np.float32, cull_backfaces=CULL, **kw)
            q_gt, t_gt = syn.random_quat(rng), np.array([rng.uniform(-0.2, 0.2) * dist, rng.uniform(-0.2, 0.2) * dist, -dist])
            r = R.render(orc.pose_fwd(np.concatenate([q_gt, t_gt])[:, None].astype(np.float32)))
            cov = r["rast"][0, ..., 3] > 0
            gt = dict(rgb=r["rgb"][0].copy(), depth=r["depth"][0].copy(), segmentation=np.repeat(cov[..., None], 3, -1).astype(np.float32))
            R.gt = {k: v[None] for k, v in gt.items()}
            params = []
            for b in range(B):
                q, t = syn.perturb_pose(q_gt, t_gt, float(rng.uniform(0, 30)), float(rng.uniform(0, 0.3)), rng)
                params.append(np.concatenate([q * rng.uniform(0.7, 1.4), t]))
            params = np.stack(params, 1).astype(np.float32)
            lr_mult = rng.uniform(0.5, 2.0, size=B).astype(np.float32)
            R.weights = {k: weights.get(k) for k in KEYS}
            total, logs, g_ref, _ = R.loss_and_grad(params, lr_mult)
            texk = dict(uv=T(uv), tex=T(tex)) if textured else dict(vtx_color=T(vcol))
            eng = _RE(T(pos), T(tri), T(proj), [H, W], {k: T(v) for k, v in gt.items()}, T(params), T(lr_mult), [0.1], weights, **texk)
            losses, grad = eng.loss_and_grad()
            torch.cuda.synchronize()
            st = eng.check()
            lg, gg = losses.cpu().numpy(), grad.cpu().numpy()
            ok = bool(np.isfinite(gg).all()) and eng.cull_sign == (R._cull_sign if CULL else 0)
            for i, k in enumerate(KEYS):
                if k in logs: ok &= bool(np.allclose(lg[i], logs[k], rtol=1e-4, atol=2e-6))  # (atol: round-off of the frame sums)
            gerr = float(np.abs(gg - g_ref).max() / max(np.abs(g_ref).max(), 1e-5))  # (floor: oracle round-off where the true gradient vanishes)
            ok &= gerr < 1e-2
            stats["max_grad_err"] = max(stats["max_grad_err"], gerr); stats["outside"] += int(st["outside_view_volume"] > 0); stats["big"] += int(st["big_triangles"] > 0)
            if case % 3 == 0:  # the materialising path on the same soup: fused and op by op against the oracle's images (both faces drawn)
                R.cull_backfaces = False
                r2 = R.render(orc.pose_fwd(params))
                ex = lambda a: T(a)[None].expand(B, *a.shape)
                kw2 = dict(uv=ex(uv), uv_idx=ex(tri), tex=ex(tex)) if textured else dict(vtx_color=ex(vcol))
                for fused in (True, False):
                    out = dd.render_texture_batch(dd.RasterizeGLContext(), ex(proj), T(orc.pose_fwd(params)), ex(pos), ex(tri), [H, W], return_rast_out=True, fused=fused, **kw2)
                    mok = bool(np.array_equal(out["rast_out"][..., 3].cpu().numpy(), r2["rast"][..., 3]))
                    for k in ("rgb", "depth", "mask"):
                        mok &= bool(np.allclose(out[k].cpu().numpy(), r2[k], rtol=1e-4, atol=5e-5 * max(1.0, float(np.abs(r2[k]).max()))))
                    if not mok:
                        ok = False
                        print("MISMATCH (materialising path, fused =", fused, ")", tag)
                stats["materialising"] = stats.get("materialising", 0) + 1
            if not ok:
                bad += 1
                print("MISMATCH", tag, "| grad err", gerr, "cull", eng.cull_sign, R._cull_sign, "status", st, "| losses", lg[:, 0], {k: v[0] for k, v in logs.items()})
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e))
    if verbose:
        print(f"{n_cases} engine soups, {bad} bad, {time.time() - t_start:.0f} s", stats)
    return bad, stats


def sweep_ops(n_cases=300, seed0=0, verbose=True):
    """The small ops with random shapes and extreme values against the oracle: xfm_points / xfm_vectors forward and backward
    (any N, batch-1 broadcast of points or matrices), texture (uv far outside [0,1], negative, exactly on texel borders, 1x1 ..
    odd sizes), masked L1 (any element count, mask / no mask / channel-0 mask), the pose-matrix op, the fused Adam step
    (teacher-forced against the oracle's Adam over consecutive iterations)."""
    from diffdope_amd.render import masked_l1_mean
    bad = 0
    stats = dict(xfm=0, texture=0, masked_l1=0, pose=0, adam=0)
    t_start = time.time()
    for case in range(n_cases):
        rng = np.random.RandomState(seed0 + case)
        tag = f"ops case {case} seed {seed0 + case}"
        try:
            ok = True
            # ---- xfm
            B, N = int(rng.randint(1, 6)), int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, int(rng.randint(1, 5000))]))
            pb, mb = (1 if rng.rand() < 0.3 else B), B  # (points may be one broadcast copy; the matrix carries the batch, as in the reference)
            pts = (rng.normal(size=(pb, N, 3)) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
            mtx = rng.normal(size=(mb, 4, 4)).astype(np.float32)
            for is_points in (True, False):
                fn = dd.ops.xfm_points if is_points else dd.ops.xfm_vectors
                p_t, m_t = T(pts, requires_grad=True), T(mtx, requires_grad=True)
                out = fn(p_t, m_t)
                pe, me = np.broadcast_to(pts, (max(pb, mb),) + pts.shape[1:]), np.broadcast_to(mtx, (max(pb, mb), 4, 4))
                ref = orc.xfm_fwd(np.ascontiguousarray(pe), np.ascontiguousarray(me), is_points)
                ok &= bool(np.array_equal(out.detach().cpu().numpy(), ref))  # (bit-identical: the k-ordered fma chain)
                go = rng.normal(size=ref.shape).astype(np.float32)
                out.backward(T(go))
                dp, dm = orc.xfm_bwd(np.ascontiguousarray(pe), np.ascontiguousarray(me), go, is_points)
                if pb == 1 and max(pb, mb) > 1: dp = dp.sum(0, keepdims=True)
                if mb == 1 and max(pb, mb) > 1: dm = dm.sum(0, keepdims=True)
                ok &= bool(np.allclose(p_t.grad.cpu().numpy(), dp, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(dp).max())))
                ok &= bool(np.allclose(m_t.grad.cpu().numpy(), dm, rtol=2e-4, atol=2e-5 * max(1.0, np.abs(dm).max())))
            stats["xfm"] += 1
            if not ok: print("MISMATCH xfm", tag, B, N, pb, mb)
            # ---- texture
            ok2 = True
            Bt, H, W = int(rng.randint(1, 4)), int(rng.randint(1, 40)), int(rng.randint(1, 50))
            Th, Tw = int(rng.randint(1, 33)), int(rng.randint(1, 33))
            tb = 1 if rng.rand() < 0.5 else Bt
            tex = rng.uniform(size=(tb, Th, Tw, 3)).astype(np.float32)
            uv = (rng.normal(size=(Bt, H, W, 2)) * 10.0 ** rng.uniform(-1, 4)).astype(np.float32)
            snap = rng.rand(Bt, H, W, 2) < 0.2
            uv[snap] = (np.round(uv[snap] * Tw) / Tw).astype(np.float32)  # on texel borders / centres
            tx_t, uv_t = T(tex, requires_grad=True), T(uv, requires_grad=True)
            out = dd.texture(tx_t, uv_t, filter_mode="linear")
            ref = orc.texture_fwd(tex, uv)
            ok2 &= bool(np.allclose(out.detach().cpu().numpy(), ref, rtol=0, atol=2e-6))
            go = rng.normal(size=ref.shape).astype(np.float32)
            out.backward(T(go))
            duv, dtex = orc.texture_bwd(tex, uv, go, True)
            ok2 &= bool(np.allclose(uv_t.grad.cpu().numpy(), duv, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(duv).max())))
            ok2 &= bool(np.allclose(tx_t.grad.cpu().numpy(), dtex, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(dtex).max())))
            stats["texture"] += 1
            if not ok2: print("MISMATCH texture", tag, (Bt, H, W), (tb, Th, Tw), float(np.abs(uv).max()))
            # ---- masked L1
            ok3 = True
            Bm, Hm, Wm = int(rng.randint(1, 7)), int(rng.randint(1, 60)), int(rng.randint(1, 60))
            seg = (rng.rand(1, Hm, Wm, 3) > rng.uniform(0.1, 0.95)).astype(np.float32) * rng.choice([1.0, 0.5, -2.0])
            for shape, ch0 in (((Bm, Hm, Wm, 3), False), ((Bm, Hm, Wm), True)):
                x = rng.normal(size=shape).astype(np.float32)
                y = rng.normal(size=(1,) + shape[1:]).astype(np.float32)
                use_mask = rng.rand() < 0.8 or ch0
                x_t = T(x, requires_grad=True)
                yb, sb = T(y).expand(shape), T(seg).expand(Bm, Hm, Wm, 3)
                lr = T(rng.rand(Bm).astype(np.float32))
                o = masked_l1_mean(x_t, yb, sb if use_mask else None, mask_channel0=ch0)
                mk = (seg[..., 0] if ch0 else seg) if use_mask else 1.0
                refv = np.abs((x.astype(np.float64) - y) * mk).mean(axis=tuple(range(1, x.ndim)))
                ok3 &= bool(np.allclose(o.detach().cpu().numpy(), refv, rtol=3e-5, atol=1e-7))
                (o * lr).mean().backward()
                d = (x - y) * mk
                gref = np.sign(d) * mk * (lr.cpu().numpy().reshape((-1,) + (1,) * (x.ndim - 1)) / Bm / np.prod(shape[1:]))
                ok3 &= bool(np.allclose(x_t.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-10))
            stats["masked_l1"] += 1
            if not ok3: print("MISMATCH masked_l1", tag, (Bm, Hm, Wm))
            # ---- pose matrix op
            ok4 = True
            Bp = int(rng.randint(1, 300))
            q = (rng.normal(size=(Bp, 4)) * 10.0 ** rng.uniform(-2, 2)).astype(np.float32)
            t = (rng.normal(size=(Bp, 3)) * 10.0 ** rng.uniform(-2, 2)).astype(np.float32)
            qn = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
            q_t, t_t = T(qn, requires_grad=True), T(t, requires_grad=True)
            M = dd.matrix_batch_44_from_position_quat(q=q_t, p=t_t)
            pr = np.concatenate([qn, t], 1).T.astype(np.float32)
            Mref = orc.pose_fwd(pr)  # (normalises again: idempotent to round-off)
            ok4 &= bool(np.allclose(M.detach().cpu().numpy(), Mref, rtol=1e-5, atol=1e-6 * max(1.0, np.abs(t).max())))
            stats["pose"] += 1
            if not ok4: print("MISMATCH pose", tag, Bp)
            # ---- fused Adam: consecutive iterations, oracle Adam teacher-forced with the oracle's gradient at the engine's parameters
            ok5 = True
            if case % 3 == 0:
                sc = make_scene(int(rng.randint(6, 20)), int(rng.randint(8, 24)), int(rng.randint(30, 90)), int(rng.randint(40, 120)), B=int(rng.randint(1, 5)),
                                dist=float(rng.uniform(1.2, 4.0)), seed=seed0 + case)
                weights = dict(rgb=0.7, depth=1.0, mask=1.0)
                R = sc["oracle"]; R.weights = dict(weights, edge=None)
                R.cull_backfaces = CULL
                lrs = [0.01, 0.008, 0.006, 0.004]
                p_t = T(sc["params"])
                eng = _RE(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [sc["H"], sc["W"]], {k: T(v) for k, v in sc["gt"].items()}, p_t, T(sc["lr_mult"]), lrs,
                                      weights, uv=T(sc["uv"]), tex=T(sc["tex"]), optimizer="adam")
                m1 = np.zeros_like(sc["params"]); m2 = np.zeros_like(sc["params"])
                b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-8)
                for step, lr in enumerate(lrs, start=1):
                    before = p_t.cpu().numpy().copy()
                    eng.run(1); eng.finish()
                    _, _, g, _ = R.loss_and_grad(before, sc["lr_mult"])
                    m1 = (b1 * m1 + (np.float32(1) - b1) * g).astype(np.float32)
                    m2 = (b2 * m2 + (np.float32(1) - b2) * g * g).astype(np.float32)
                    c1, c2 = np.float32(1.0 - 0.9 ** step), np.float32(1.0 - 0.999 ** step)
                    upd = np.float32(lr) * (m1 / c1) / (np.sqrt(m2 / c2) + eps)
                    got = before - p_t.cpu().numpy()
                    # components whose gradient is round-off are excluded: Adam normalises them to a full-size step of either sign
                    big = np.abs(g) > 1e-4 * np.abs(g).max()
                    ok5 &= bool(np.allclose(got[big], upd[big], rtol=2e-3, atol=2e-3 * lr))
                stats["adam"] += 1
                if not ok5: print("MISMATCH adam", tag)
            if not (ok and ok2 and ok3 and ok4 and ok5): bad += 1
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e))
    if verbose:
        print(f"{n_cases} op cases, {bad} bad, {time.time() - t_start:.0f} s", stats)
    return bad, stats


def sweep_state(n_cases=100, seed0=0, verbose=True):
    """The engine as a state machine: a six-iteration run in one go against the same six iterations cut into random pieces with
    evaluation passes in between, replayed from captured graphs, restarted after a rewind, continued after a new observation
    and back, and computed as two shards with the unsharded run's slice counts -- everything bit for bit."""
    bad = 0
    stats = dict(graphs=0, shards=0, rewinds=0, observations=0)
    t_start = time.time()
    for case in range(n_cases):
        rng = np.random.RandomState(seed0 + case)
        rows, cols = int(rng.randint(5, 24)), int(rng.randint(6, 28))
        H, W = int(rng.randint(30, 110)), int(rng.randint(40, 140))
        B = int(rng.randint(2, 9))
        wide = bool(os.environ.get("FUZZ_WIDE"))  # batches wide enough for the two-chain runs (run with DDX_TWO_MIN=2): the reference
        if wide:                                  # run of a case is then ONE chain (single_stream), everything else may fork
            B = int(rng.choice([32, 48, 64]))
        names = [k for k in KEYS if rng.rand() < 0.6] or ["rgb"]
        weights = {k: float(rng.uniform(0.3, 1.5)) for k in names}
        optimizer = "adam" if rng.rand() < 0.5 else "sgd"
        sc = make_scene(rows, cols, H, W, B=B, dist=float(rng.uniform(1.2, 5.0)), seed=seed0 + case, textured=bool(rng.randint(2)))
        tag = f"state case {case} seed {seed0 + case}: mesh {rows}x{cols} frame {H}x{W} B {B} {sorted(weights)} {optimizer}"
        n = 6
        lrs = [float(x) for x in rng.uniform(0.002, 0.02, n)]
        try:
            tex = dict(uv=T(sc["uv"]), tex=T(sc["tex"])) if sc["textured"] else dict(vtx_color=T(sc["vtx_color"]))
            gt = {k: T(v) for k, v in sc["gt"].items()}
            def engine(params, lr_mult, lo=0, hi=B, **kw):
                return _RE(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [H, W], gt, params, T(lr_mult[lo:hi]), lrs, weights, optimizer=optimizer, **tex, **kw)
            p_ref = T(sc["params"])
            ref = engine(p_ref, sc["lr_mult"], single_stream=wide)
            ref.run(); ref.finish()
            want = (p_ref.clone(), ref.losses().clone(), ref.mtx_log.clone())
            same = lambda p, e: torch.equal(p, want[0]) and torch.equal(e.losses(), want[1]) and torch.equal(e.mtx_log, want[2])
            ok = True
            # (1) pieces + evaluation passes + graphs
            use_graph = int(rng.choice([0, 0, 1, 2, 3]))
            p1 = T(sc["params"])
            e1 = engine(p1, sc["lr_mult"])
            done = 0
            while done < n:
                k = int(rng.randint(1, n - done + 1))
                if rng.rand() < 0.5: e1.loss_and_grad()
                e1.run(k, use_graph=use_graph)
                done += k
            e1.finish()
            ok1 = same(p1, e1)
            stats["graphs"] += int(use_graph > 0)
            # (2) rewind: a few iterations, parameters put back, start again
            p2 = T(sc["params"])
            e2 = engine(p2, sc["lr_mult"])
            e2.run(int(rng.randint(1, n))); e2.finish()
            if rng.rand() < 0.5 or optimizer == "adam":  # (a plain rewind keeps Adam's moments: only new_observation restarts them)
                e2.new_observation(params=T(sc["params"]))
            else:
                p2.copy_(T(sc["params"])); e2.rewind(0)
            e2.run(); e2.finish()
            ok2 = same(p2, e2)
            stats["rewinds"] += 1
            # (3) another observation in between, then back
            sc_b = make_scene(rows, cols, H, W, B=B, dist=float(rng.uniform(1.2, 5.0)), seed=seed0 + case, textured=sc["textured"], rot_deg=3.0)
            e1.new_observation(gt={k: T(v) for k, v in sc_b["gt"].items()}, params=T(sc_b["params"]))
            e1.run(int(rng.randint(1, n + 1))); e1.finish()
            e1.new_observation(gt=gt, params=T(sc["params"]))
            e1.run(); e1.finish()
            ok3 = same(p1, e1)
            stats["observations"] += 1
            # (4) two shards with the unsharded run's slice counts
            cut = int(rng.randint(1, B))
            ss, es = ref.slices
            pa, pb = T(sc["params"][:, :cut]), T(sc["params"][:, cut:])
            ea = engine(pa, sc["lr_mult"], 0, cut, global_batch=B, shade_slices=ss, edge_slices=es)
            eb = engine(pb, sc["lr_mult"], cut, B, global_batch=B, shade_slices=ss, edge_slices=es)
            ea.run(); eb.run(); ea.finish(); eb.finish()
            ok4 = torch.equal(torch.cat([pa, pb], 1), want[0]) and torch.equal(torch.cat([ea.losses(), eb.losses()], 2), want[1])
            stats["shards"] += 1
            if not (ok1 and ok2 and ok3 and ok4):
                bad += 1
                print("MISMATCH", tag, "pieces/graphs", ok1, "(graph", use_graph, ") rewind", ok2, "observation", ok3, "shards", ok4, "cut", cut)
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e))
    if verbose:
        print(f"{n_cases} state cases, {bad} bad, {time.time() - t_start:.0f} s", stats)
    return bad, stats


def sweep_api(n_cases=60, seed0=0, verbose=True):
    """DiffDope.run_optimization through the fused engine and through the op-by-op autograd path (what a user loss function
    gets) on random small scenes and loss sets: the logged losses of iteration 0 agree tightly (same parameters), the run as a
    whole loosely (two different float orders driving L1 terms), engine reuse on a second run gives the first run's result."""
    bad = 0
    stats = dict(max_first_loss_diff=0.0, max_param_diff=0.0)
    t_start = time.time()
    for case in range(n_cases):
        rng = np.random.RandomState(seed0 + case)
        rows, cols = int(rng.randint(5, 22)), int(rng.randint(6, 26))
        H, W = int(rng.randint(30, 100)), int(rng.randint(40, 130))
        B = int(rng.randint(1, 7))
        losses = [k for k in ("rgb", "depth", "mask", "edge") if rng.rand() < 0.6] or ["mask"]
        sc = make_scene(rows, cols, H, W, B=1, dist=float(rng.uniform(1.3, 5.0)), seed=seed0 + case, rot_deg=float(rng.uniform(1, 12)), trans=float(rng.uniform(0, 0.05)))
        # even cases: the engine draws both faces like the op-by-op path (same visibility rule: the kernels are what is compared,
        # tight); odd cases: the engine's default, back faces of the closed mesh culled (deviation D5) -- equal in exact arithmetic,
        # but a sliver at a pole of the blob can win a pixel through its extrapolated depth (seed 940228: 600 triangles on 149
        # pixels, one pixel changes owner, the Sobel term moves by 0.4 %): loose
        cull = bool(case % 2)
        tag = f"api case {case} seed {seed0 + case}: mesh {rows}x{cols} frame {H}x{W} B {B} {losses} cull {cull}"
        try:
            def make():
                mesh = dd.Mesh.from_arrays(sc["pos"], sc["tri"], uv=sc["uv"], tex=sc["tex"])
                q, t = sc["params"][:4, 0], sc["params"][4:, 0]
                obj = dd.Object3D(position=list(t), rotation=list(q / np.linalg.norm(q)), batchsize=B, opencv2opengl=False, scale=1, mesh=mesh)
                scene = dd.Scene(tensor_rgb=dd.Image(img_tensor=torch.tensor(sc["gt"]["rgb"])), tensor_depth=dd.Image(img_tensor=torch.tensor(sc["gt"]["depth"])),
                                 tensor_segmentation=dd.Image(img_tensor=torch.tensor(sc["gt"]["segmentation"])))
                cam = dd.Camera(fx=1, fy=1, cx=0, cy=0, im_width=W, im_height=H)
                cam.cam_proj = torch.tensor(sc["proj"], dtype=torch.float64)
                cfg = dict(losses=dict(l1_rgb_with_mask="rgb" in losses, weight_rgb=0.7, l1_depth_with_mask="depth" in losses, weight_depth=1.0,
                                       l1_mask="mask" in losses, weight_mask=1.0, l1_edge="edge" in losses, weight_edge=0.6),
                           hyperparameters=dict(nb_iterations=3, batchsize=B, base_lr=0.05, learning_rates_bound=[0.5, 2.0], learning_rate_base=1, lr_decay=0.1, seed=3,
                                                cull_backfaces=cull))
                return dd.DiffDope(cfg=cfg, camera=cam, object3d=obj, scene=scene)
            a, b = make(), make()
            a.run_optimization(fused=True)
            b.run_optimization(fused=False)
            ok = set(a.losses_values) == set(b.losses_values)
            for k in a.losses_values:
                la, lb = a.losses_values[k].numpy(), b.losses_values[k].numpy()
                d0 = float(np.abs(la[0] - lb[0]).max() / max(np.abs(lb[0]).max(), 1e-6))
                # (culled cases: ONE pixel changing owner is within the declared deviation, and on a scene whose whole loss is a few
                # pixels' worth it is more than 1 % of it -- seed 9500533: 183 covered pixels, mask loss 9.1 pixel-units, one sliver
                # pixel = 0.5 of them = 5.5 %, while the un-culled run of the same case agrees to 4e-7 --: up to 1.5 pixel-units pass)
                px_units = float(np.abs(la[0] - lb[0]).max()) * H * W
                stats["max_first_loss_diff"] = max(stats["max_first_loss_diff"], d0 if not (cull and px_units < 1.5) else min(d0, 1e-2))
                ok &= d0 < (1e-2 if cull else 2e-3) or (cull and px_units < 1.5)
            pa, pb = a.object3d.params_tensor().cpu().numpy(), b.object3d.params_tensor().cpu().numpy()
            dp = float(np.abs(pa - pb).max())
            stats["max_param_diff"] = max(stats["max_param_diff"], dp)
            ok &= dp < 5e-3
            first = (pa.copy(), {k: v.clone() for k, v in a.losses_values.items()})
            a.object3d.reset_pose()
            a.run_optimization(fused=True)  # (reuses its engine)
            ok &= bool(np.array_equal(a.object3d.params_tensor().cpu().numpy(), first[0])) and all(torch.equal(a.losses_values[k], first[1][k]) for k in first[1])
            if not ok:
                bad += 1
                print("MISMATCH", tag, "first-iteration loss diff", stats["max_first_loss_diff"], "param diff", dp)
        except Exception as e:
            bad += 1
            print("ERROR", tag, repr(e))
    if verbose:
        print(f"{n_cases} api cases, {bad} bad, {time.time() - t_start:.0f} s", stats)
    return bad, stats


if __name__ == "__main__":
    if os.environ.get("FUZZ_STATE"):
        sweep_state(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        sys.exit(0)
    if os.environ.get("FUZZ_API"):
        sweep_api(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        sys.exit(0)
    if os.environ.get("FUZZ_OPS"):
        sweep_ops(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        sys.exit(0)
    if os.environ.get("FUZZ_ENGINE_SOUPS"):
        sweep_engine_soups(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        sys.exit(0)
    if os.environ.get("FUZZ_SOUPS"):
        sweep_soups(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        sys.exit(0)
    sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
