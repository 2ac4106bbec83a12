"""GPU parity of the fused engine and of the op-by-op autograd path against the oracle:
per-hypothesis losses, d loss / d pose (through one SGD step), and the optimised poses."""
import numpy as np
import pytest
import torch

from diffdope_amd import synthetic as syn
from tests.scenes import make_scene

pytestmark = pytest.mark.gpu

T = lambda a, **k: torch.tensor(np.ascontiguousarray(a), device="cuda", **k)
KEYS = ("rgb", "depth", "mask_selection", "edge")


def _engine(sc, weights, lrs, params=None, **kw):
    import diffdope_amd as dd

    params = T(sc["params"] if params is None else params)
    tex = dict(uv=T(sc["uv"]), tex=T(sc["tex"])) if sc["textured"] else dict(vtx_color=T(sc["vtx_color"]))
    gt = {k: T(v) for k, v in sc["gt"].items()}
    eng = dd.RefineEngine(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [sc["H"], sc["W"]], gt, params, T(sc["lr_mult"]), lrs,
                          weights, **tex, **kw)
    return eng, params


@pytest.mark.parametrize("weights", [dict(rgb=0.7), dict(depth=1.0), dict(mask=1.0), dict(rgb=0.7, depth=1.0, mask=1.0),
                                     dict(edge=1.0), dict(rgb=0.7, depth=1.0, edge=0.8), dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8)])
@pytest.mark.parametrize("textured", [True, False])
def test_engine_one_iteration_losses_and_gradients(weights, textured):
    sc = make_scene(16, 20, 60, 80, B=3, dist=1.8, textured=textured)
    R = sc["oracle"]
    R.weights = {k: weights.get(k) for k in ("rgb", "depth", "mask", "edge")}
    total, logs, g_ref, _ = R.loss_and_grad(sc["params"], sc["lr_mult"])
    lr = 0.5
    eng, params = _engine(sc, weights, [lr])
    eng.run(use_graph=False)
    eng.finish()
    st = eng.check()
    assert st["active_tiles"] > 0
    g_gpu = (sc["params"] - params.cpu().numpy()) / lr
    scale = np.abs(g_ref).max()
    assert scale > 0
    np.testing.assert_allclose(g_gpu, g_ref, rtol=2e-4, atol=2e-4 * scale)  # (round 5: a tenth of round 4's tolerance; passes at a third of this)
    lg = eng.losses()[0].cpu().numpy()
    for i, key in enumerate(KEYS):
        if key in logs:
            np.testing.assert_allclose(lg[i], logs[key], rtol=2e-5, atol=1e-7)
        else:
            assert np.all(lg[i] == 0)
    from oracle import oracle as orc

    np.testing.assert_allclose(eng.mtx_log[0].cpu().numpy().reshape(-1, 4, 4), orc.pose_fwd(sc["params"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("w", [dict(rgb=0.7, depth=1.0, mask=1.0), dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8)])
def test_engine_graph_replay_equals_stream_launches_and_is_deterministic(w):
    sc = make_scene(16, 20, 60, 80, B=4, dist=1.8)
    lrs = [0.05] * 6
    outs = []
    for use_graph in (False, True, True, 4):  # (4: graphs of 4 iterations + 2 iterations launched kernel by kernel)
        eng, params = _engine(sc, w, lrs)
        eng.run(use_graph=use_graph)
        eng.finish()
        eng.check()
        outs.append((params.cpu().numpy().copy(), eng.losses().cpu().numpy().copy()))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])  # no atomics on the float path


def test_engine_optimisation_matches_oracle_trajectory():
    """Reference mode (SGD, decayed LR, per-hypothesis multipliers): 30 iterations of the engine vs the
    oracle's op-by-op loop -- final poses within 1e-3 rad / 1e-3 m (1 unit = 0.1 m)."""
    from oracle import oracle as orc

    sc = make_scene(16, 20, 60, 80, B=4, dist=1.8, rot_deg=6.0, trans=0.02)
    w = dict(rgb=0.7, depth=1.0, mask=1.0)
    R = sc["oracle"]
    R.weights = w
    lrs = [l * 0.02 for l in orc.lr_schedule(29, 20, 0.1)]
    p_ref, logs_ref, mtx_ref = R.optimise(sc["params"], sc["lr_mult"], lrs)
    eng, params = _engine(sc, w, lrs)
    eng.run()
    eng.finish()
    eng.check()
    p_gpu = params.cpu().numpy()
    for b in range(sc["B"]):
        ang = syn.rotation_geodesic(p_gpu[:4, b], p_ref[:4, b])
        dt = np.linalg.norm(p_gpu[4:, b] - p_ref[4:, b]) * 0.1
        assert ang < 1e-3 and dt < 1e-3, (b, ang, dt)
    lg = eng.losses().cpu().numpy()
    np.testing.assert_allclose(lg[0, 0], logs_ref["rgb"][0], rtol=1e-4)
    np.testing.assert_allclose(eng.mtx_log.cpu().numpy().reshape(len(lrs), -1, 4, 4)[0], mtx_ref[0], rtol=1e-5, atol=1e-6)
    # and the loss went down for the best hypothesis
    tot = lg.sum(1)
    assert tot[-1].min() < tot[0].min()


def test_engine_recovers_known_pose():
    """K7: render the observation from a known pose, perturb, refine: the arg-min hypothesis ends within
    1e-3 rad / 1e-3 m of the generating pose (easy tier: ~2 deg, ~1 % translation)."""
    sc = make_scene(40, 64, 120, 160, B=8, dist=4.0, rot_deg=2.0, trans=0.01, tex_size=64)
    from oracle import oracle as orc

    w = dict(rgb=0.7, depth=1.0, mask=1.0)
    sc["lr_mult"] = np.linspace(0.5, 3.0, sc["B"]).astype(np.float32)
    lrs = [l * 0.02 for l in orc.lr_schedule(199, 20, 0.1)]
    eng, params = _engine(sc, w, lrs)
    eng.run()
    eng.finish()
    eng.check()
    lg = eng.losses().cpu().numpy()
    best = int(np.argmin(lg[-1].mean(0)))
    p = params.cpu().numpy()[:, best]
    ang = syn.rotation_geodesic(p[:4], sc["q_gt"])
    dt = np.linalg.norm(p[4:] - sc["t_gt"]) * 0.1
    assert lg[-1].sum(0)[best] < 0.2 * lg[0].sum(0)[best]
    assert ang < 1e-3 and dt < 1e-3, (ang, dt)


def test_render_texture_batch_autograd_matches_oracle():
    """The materialising path (render_texture_batch + torch losses + autograd) against the oracle: fused (ddx_gbuffer_fwd / _bwd
    between rasterize and antialias) and op by op (interpolate / texture / xfm_points like the reference)."""
    import diffdope_amd as dd
    from oracle import oracle as orc

    for textured, fused in ((True, True), (False, True), (True, False), (False, False)):
        sc = make_scene(16, 20, 60, 80, B=2, dist=1.8, textured=textured)
        R = sc["oracle"]
        R.cull_backfaces = False  # the op-level ops draw both faces, like nvdiffrast
        R.weights = dict(rgb=0.7, depth=1.0, mask=1.0)
        total, logs, g_ref, r_ref = R.loss_and_grad(sc["params"], sc["lr_mult"])
        B = sc["B"]
        params = [T(sc["params"][i], requires_grad=True) for i in range(7)]
        q = torch.stack(params[:4], dim=0).T
        q = q / torch.norm(q, dim=1).reshape(-1, 1)
        t = torch.stack(params[4:], dim=0).T
        mtx = dd.matrix_batch_44_from_position_quat(p=t, q=q)
        ex = lambda a: T(a)[None].expand(B, *a.shape)
        kw = dict(uv=ex(sc["uv"]), uv_idx=ex(sc["tri"]), tex=ex(sc["tex"])) if textured else dict(vtx_color=ex(sc["vtx_color"]))
        ctx = dd.RasterizeGLContext()
        out = dd.render_texture_batch(ctx, ex(sc["proj"]), mtx, ex(sc["pos"]), ex(sc["tri"]), [sc["H"], sc["W"]],
                                      return_rast_out=True, fused=fused, **kw)  # fused: one gbuffer pass each way; else op by op
        assert np.array_equal(out["rast_out"][..., 3].detach().cpu().numpy(), r_ref["rast"][..., 3])
        for k in ("rgb", "depth", "mask"):
            np.testing.assert_allclose(out[k].detach().cpu().numpy(), r_ref[k], rtol=1e-4, atol=2e-5)
        gt = {k: T(v)[None] for k, v in sc["gt"].items()}
        lrm = T(sc["lr_mult"])
        loss = 0.7 * (torch.mean(torch.abs((out["rgb"] - gt["rgb"]) * gt["segmentation"]), (1, 2, 3)) * lrm).mean()
        loss = loss + 1.0 * (torch.mean(torch.abs((out["depth"] - gt["depth"]) * gt["segmentation"][..., 0]), (1, 2)) * lrm).mean()
        loss = loss + 1.0 * (torch.mean(torch.abs(out["mask"] - gt["segmentation"]), (1, 2, 3)) * lrm).mean()
        assert abs(float(loss) - total) < 1e-5 * max(1, abs(total))
        loss.backward()
        g = np.stack([p.grad.cpu().numpy() for p in params])
        np.testing.assert_allclose(g, g_ref, rtol=1e-4, atol=1e-4 * np.abs(g_ref).max())


@pytest.mark.parametrize("rows,cols,H,W,B,dist", [(4, 6, 50, 70, 1, 1.2), (6, 8, 120, 160, 3, 1.0), (16, 20, 37, 53, 2, 1.8),
                                                  (24, 32, 1083, 1925, 2, 2.0)])  # 8228 tiles: more than one compaction pass
def test_engine_large_triangles_ragged_sizes_single_hypothesis(rows, cols, H, W, B, dist):
    """Low-poly meshes close to the camera (every triangle takes the tile pass), resolutions that are not
    multiples of the tile size, B = 1: losses and gradients still match the oracle."""
    sc = make_scene(rows, cols, H, W, B=B, dist=dist, tex_size=16)
    R = sc["oracle"]
    w = dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8)
    R.weights = w
    total, logs, g_ref, r_ref = R.loss_and_grad(sc["params"], sc["lr_mult"])
    eng, params = _engine(sc, w, [0.25])
    eng.run()
    eng.finish()
    st = eng.check()
    if rows <= 6:
        assert st["big_triangles"] == 1  # the tile pass ran
    g = (sc["params"] - params.cpu().numpy()) / 0.25
    np.testing.assert_allclose(g, g_ref, rtol=1e-4, atol=1e-4 * np.abs(g_ref).max())
    lg = eng.losses()[0].cpu().numpy()
    for i, key in enumerate(KEYS):
        np.testing.assert_allclose(lg[i], logs[key], rtol=3e-5, atol=1e-7)


def test_engine_hypothesis_leaving_the_frame():
    """One hypothesis looks away from the object: no active tile, the loss is the whole-frame background
    term (incl. the -t_z depth quirk of diffdope.py:204-209) and only t_z receives a gradient."""
    sc = make_scene(16, 20, 60, 80, B=2, dist=1.8)
    sc["params"][4, 1] = 50.0  # x far outside the frustum
    R = sc["oracle"]
    w = dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8)
    R.weights = w
    total, logs, g_ref, r_ref = R.loss_and_grad(sc["params"], sc["lr_mult"])
    assert (r_ref["rast"][1, ..., 3] > 0).sum() == 0
    eng, params = _engine(sc, w, [0.25])
    eng.run()
    eng.finish()
    g = (sc["params"] - params.cpu().numpy()) / 0.25
    np.testing.assert_allclose(g, g_ref, rtol=1e-4, atol=1e-4 * np.abs(g_ref).max())
    assert abs(g[6, 1]) > 0 and np.abs(g[:6, 1]).max() < 1e-9
    lg = eng.losses()[0].cpu().numpy()
    for i, key in enumerate(KEYS):
        np.testing.assert_allclose(lg[i], logs[key], rtol=3e-5, atol=1e-7)
    # and the engine keeps working on later iterations (tile flags / zbuf re-armed correctly)
    eng2, p2 = _engine(sc, w, [0.01] * 4)
    eng2.run()
    eng2.finish()
    p_ref, logs_ref, _ = R.optimise(sc["params"], sc["lr_mult"], [0.01] * 4)
    np.testing.assert_allclose(p2.cpu().numpy(), p_ref, rtol=0, atol=2e-5)


def test_engine_with_edge_extension_tracks_oracle_over_iterations():
    """rgb + depth + edge (BASELINE configs[2]; the edge term is this build's extension, checked against the
    oracle's definition of it, not against the reference): 20 SGD iterations vs the oracle's op-by-op loop."""
    from oracle import oracle as orc

    sc = make_scene(16, 20, 60, 80, B=3, dist=1.8, rot_deg=5.0, trans=0.02)
    w = dict(rgb=0.7, depth=1.0, edge=0.5)
    R = sc["oracle"]
    R.weights = w
    lrs = [l * 0.01 for l in orc.lr_schedule(19, 20, 0.1)]
    p_ref, logs_ref, _ = R.optimise(sc["params"], sc["lr_mult"], lrs)
    eng, params = _engine(sc, w, lrs)
    eng.run()
    eng.finish()
    eng.check()
    p_gpu = params.cpu().numpy()
    for b in range(sc["B"]):
        ang = syn.rotation_geodesic(p_gpu[:4, b], p_ref[:4, b])
        dt = np.linalg.norm(p_gpu[4:, b] - p_ref[4:, b]) * 0.1
        assert ang < 1e-3 and dt < 1e-3, (b, ang, dt)
    lg = eng.losses().cpu().numpy()
    np.testing.assert_allclose(lg[0, 3], logs_ref["edge"][0], rtol=1e-4)
    assert np.all(lg[:, 2] == 0)
    assert lg[-1, 3].min() < lg[0, 3].min()


@pytest.mark.parametrize("rows,cols,H,W,weights", [(160, 160, 480, 640, dict(rgb=0.7, depth=1.0, edge=1.0)),   # BASELINE config 3
                                                   (80, 128, 720, 1280, dict(rgb=0.7, depth=1.0, edge=1.0)),   # config 5 (one object)
                                                   (100, 150, 480, 640, dict(depth=1.0, mask=1.0)),            # config 4 mesh size
                                                   (12, 16, 480, 640, dict(depth=1.0, mask=1.0)),              # low-poly: every triangle takes the tile pass
                                                   (40, 64, 720, 1280, dict(rgb=0.7, mask=1.0))])              # 10-60 px triangles: both paths
def test_engine_full_size_one_hypothesis_against_oracle_and_batch_properties(rows, cols, H, W, weights):
    """Full BASELINE sizes.  The oracle renders ONE hypothesis (seconds); the batch of 16 is checked through
    size-independent properties: hypotheses with identical parameters and multipliers give bit-identical losses and
    updates, wherever they sit in the batch, and the per-hypothesis result does not depend on the batch around it
    beyond the global 1/B factor."""
    textured = "mask" not in weights or "rgb" in weights
    sc = make_scene(rows, cols, H, W, B=16, dist=7.5, textured=textured, tex_size=256)
    R = sc["oracle"]
    R.weights = {k: weights.get(k) for k in ("rgb", "depth", "mask", "edge")}
    params = sc["params"].copy()
    lrm = sc["lr_mult"].copy()
    params[:, 9] = params[:, 2]
    lrm[9] = lrm[2]
    sc = dict(sc, params=params, lr_mult=lrm)
    lr = 0.25
    eng, p = _engine(sc, weights, [lr])
    eng.run()
    eng.finish()
    eng.check()
    lg = eng.losses()[0].cpu().numpy()
    pn = p.cpu().numpy()
    assert np.array_equal(lg[:, 2], lg[:, 9]) and np.array_equal(pn[:, 2], pn[:, 9])
    # one hypothesis against the oracle (global_B keeps the 1/B factor of the batch mean)
    total, logs, g_ref, _ = R.loss_and_grad(params[:, 2:3], lrm[2:3], global_B=16)
    g_gpu = (params[:, 2] - pn[:, 2]) / lr
    # (the gradient is read back from the parameter update, (p' - p) / lr: the round-off of p' at its own magnitude shows in it -- 3e-4 .. 2e-3 measured)
    np.testing.assert_allclose(g_gpu, g_ref[:, 0], rtol=3e-3, atol=3e-3 * np.abs(g_ref).max())
    for i, key in enumerate(KEYS):
        if key in logs:
            np.testing.assert_allclose(lg[i, 2], logs[key][0], rtol=5e-5, atol=1e-7)
    # the same hypothesis alone in a batch of one: identical losses, gradient x16
    sc1 = dict(sc, params=params[:, 2:3].copy(), lr_mult=lrm[2:3].copy(), B=1)
    eng1, p1 = _engine(sc1, weights, [lr])
    eng1.run()
    eng1.finish()
    np.testing.assert_allclose(eng1.losses()[0].cpu().numpy()[:, 0], lg[:, 2], rtol=1e-6, atol=0)
    g1 = (params[:, 2] - p1.cpu().numpy()[:, 0]) / lr
    # (gradients are read back as parameter differences: resolution ulp(7.5) / lr = 2e-6, x16 for the batch of one)
    np.testing.assert_allclose(g1, g_gpu * 16, rtol=2e-4, atol=4e-5)


@pytest.mark.parametrize("weights", [dict(rgb=0.7, depth=1.0, mask=1.0), dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8)])
def test_engine_eval_pass_gradient_against_oracle_and_torch_optimizer(weights):
    """ddx_engine_eval: loss and d loss / d params handed out directly (no optimiser step, nothing mutated) --
    compared with the oracle without going through parameter differences -- and a torch optimiser driving the fused
    path through RefineEngine.loss()."""
    sc = make_scene(16, 20, 60, 80, B=3, dist=1.8)
    R = sc["oracle"]
    R.weights = {k: weights.get(k) for k in ("rgb", "depth", "mask", "edge")}
    total, logs, g_ref, _ = R.loss_and_grad(sc["params"], sc["lr_mult"])
    eng, params = _engine(sc, weights, [0.1] * 30)
    before = params.clone()
    losses, grad = eng.loss_and_grad()
    torch.cuda.synchronize()
    assert torch.equal(params, before) and eng.it == 0
    np.testing.assert_allclose(grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-5 * np.abs(g_ref).max())  # (round 5: a tenth of round 4's tolerance; passes at a third of this)
    for i, key in enumerate(KEYS):
        if key in logs:
            np.testing.assert_allclose(losses[i].cpu().numpy(), logs[key], rtol=2e-5, atol=1e-7)
    p = params.clone().requires_grad_(True)
    val = eng.loss(p)
    assert abs(float(val.detach()) - total) < 1e-5 * max(1.0, abs(total))
    val.backward()
    np.testing.assert_allclose(p.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-5 * np.abs(g_ref).max())  # (round 5: a tenth of round 4's tolerance; passes at a third of this)
    # the evaluation pass leaves the engine usable: a normal run afterwards equals a run on a fresh engine
    eng.run(3)
    eng2, params2 = _engine(sc, weights, [0.1] * 30)
    eng2.run(3)
    eng.finish()
    eng2.finish()
    assert torch.equal(params, params2) and torch.equal(eng.losses(), eng2.losses())
    # torch.optim on the fused path
    p = torch.tensor(sc["params"], device="cuda", requires_grad=True)
    eng3, _ = _engine(sc, weights, [0.1] * 30)
    opt = torch.optim.Adam([p], lr=5e-3)
    first = None
    for _ in range(25):
        opt.zero_grad()
        val = eng3.loss(p)
        val.backward()
        opt.step()
        first = float(val.detach()) if first is None else first
    assert float(val.detach()) < 0.7 * first


@pytest.mark.parametrize("rows,cols,H,W,B,dist", [(700, 720, 1080, 1920, 3, 2.2), (120, 160, 2160, 3840, 2, 1.6)])
def test_engine_at_the_large_end(rows, cols, H, W, B, dist):
    """A million triangles at 1080p, and a 4K frame (the upper end of what ddx_engine_create accepts is 4096 x 4096): all four
    loss terms and the pose gradient against the oracle, three iterations without a status flag, index arithmetic included."""
    sc = make_scene(rows, cols, H, W, B=B, dist=dist, tex_size=512)
    assert len(sc["tri"]) in (1008000, 38400) and sc["coverage"] > 0.1
    w = dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.5)
    R = sc["oracle"]
    R.weights = w
    total, logs, g_ref, _ = R.loss_and_grad(sc["params"], sc["lr_mult"])
    eng, params = _engine(sc, w, [0.05, 0.05, 0.05])
    losses, grad = eng.loss_and_grad()
    torch.cuda.synchronize()
    lg = losses.cpu().numpy()
    for i, key in enumerate(KEYS):
        np.testing.assert_allclose(lg[i], logs[key], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-5 * np.abs(g_ref).max())  # (round 5: a tenth of round 4's tolerance; passes at a third of this)
    eng.run()
    eng.finish()
    st = eng.check()
    assert st["it"] == 2 and st["active_tiles"] > 1000 and np.isfinite(params.cpu().numpy()).all()
    assert float(eng.losses()[2].sum()) < float(eng.losses()[0].sum())


def test_mesh_with_more_vertices_than_triangles():
    """Deviation D6: a mesh un-merged per wedge (V = 3 T > T, what a PLY with per-face uv becomes).  The reference's tensor of
    ones has T rows; here coverage is interpolated per vertex.  Fused engine, fused and op-by-op materialising paths all agree
    with the oracle, and the mask has no holes."""
    import diffdope_amd as dd
    from oracle import oracle as orc

    sc = make_scene(8, 10, 60, 80, B=2, dist=1.6)
    tri0 = sc["tri"]
    pos = sc["pos"][tri0.reshape(-1)].copy()          # every corner its own vertex
    uv = sc["uv"][tri0.reshape(-1)].copy()
    tri = np.arange(len(pos), dtype=np.int32).reshape(-1, 3)
    assert len(pos) == 3 * len(tri)
    w = dict(rgb=0.7, depth=1.0, mask=1.0)
    R = orc.RenderOracle(pos, tri, sc["proj"], sc["H"], sc["W"], sc["gt"], w, dtype=np.float32, cull_backfaces=False, uv=uv, tex=sc["tex"])
    total, logs, g_ref, r_ref = R.loss_and_grad(sc["params"], sc["lr_mult"])
    covered = r_ref["rast"][..., 3] > 0
    assert covered.sum() > 500 and r_ref["mask"][covered].min() > 0.4  # (no zero holes inside the silhouette)
    sc2 = dict(sc, pos=pos, uv=uv, tri=tri)
    eng, _ = _engine(sc2, w, [0.1])
    losses, grad = eng.loss_and_grad()
    torch.cuda.synchronize()
    lg = losses.cpu().numpy()
    for i, key in enumerate(KEYS):
        if key in logs:
            np.testing.assert_allclose(lg[i], logs[key], rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-4 * np.abs(g_ref).max())
    R.cull_backfaces = False
    _, _, _, r2 = R.loss_and_grad(sc["params"], sc["lr_mult"])
    B = sc["B"]
    ex = lambda a: T(a)[None].expand(B, *a.shape)
    mtx = T(orc.pose_fwd(sc["params"]))
    for fused in (True, False):
        out = dd.render_texture_batch(dd.RasterizeGLContext(), ex(sc["proj"]), mtx, ex(pos), ex(tri), [sc["H"], sc["W"]], uv=ex(uv), uv_idx=ex(tri),
                                      tex=ex(sc["tex"]), fused=fused)
        np.testing.assert_allclose(out["mask"].cpu().numpy(), r2["mask"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("use_graph", [False, 2])
def test_engine_new_observation_equals_a_fresh_engine(use_graph):
    """RefineEngine.new_observation (ddx_engine_new_observation): the same mesh against another observed frame, other initial
    poses and another schedule -- the engine that already ran on the first frame gives bit for bit what a fresh engine gives
    (also when it replays captured graphs: they are captured again for the new frame)."""
    w = dict(rgb=0.7, depth=1.0, mask=1.0)
    sc_a = make_scene(16, 20, 60, 80, B=3, dist=1.8)
    sc_b = make_scene(16, 20, 60, 80, B=3, dist=2.3, rot_deg=5.0)
    lrs_a, lrs_b = [0.2, 0.15, 0.1, 0.05], [0.3, 0.2, 0.1, 0.02]
    eng, p = _engine(sc_a, w, lrs_a)
    eng.run(use_graph=use_graph)
    eng.finish()
    eng.new_observation(gt={k: T(v) for k, v in sc_b["gt"].items()}, params=T(sc_b["params"]), lr_mult=T(sc_b["lr_mult"]), lr_sched=lrs_b)
    eng.run(use_graph=use_graph)
    eng.finish()
    fresh, pf = _engine(sc_b, w, lrs_b)
    fresh.run(use_graph=use_graph)
    fresh.finish()
    assert torch.equal(p, pf) and torch.equal(eng.losses(), fresh.losses()) and torch.equal(eng.mtx_log, fresh.mtx_log)
    first, _ = _engine(sc_a, w, lrs_a)
    first.run()
    first.finish()
    assert not torch.equal(first.losses(), fresh.losses())
    with pytest.raises(ValueError):
        eng.new_observation(gt=dict(segmentation=torch.zeros(10, 10, 3)))


def test_device_side_selection_matches_torch_argmin():
    """ddx_select_best / dist.global_argmin_fused == dist.global_argmin(loss_rows[used].mean(0), ...), ties -> lowest index."""
    from diffdope_amd import dist as ddist

    g = torch.Generator(device="cpu").manual_seed(0)
    for B, mask in ((1, 0b0001), (7, 0b0101), (64, 0b0101), (300, 0b1111), (513, 0b0010)):
        rows = torch.rand(4, B, generator=g).cuda()
        if B > 4:
            rows[:, B - 2] = rows[:, 1] = rows.min(1).values - 0.25   # an exact tie between two hypotheses: index 1 must win
        mtx = torch.rand(B, 16, generator=g).cuda()
        used = [r for r in range(4) if (mask >> r) & 1]
        ref = ddist.global_argmin(rows[used].mean(0), mtx.reshape(B, 4, 4), lo=10)
        got = ddist.global_argmin_fused(rows, mask, mtx, lo=10)
        assert got[0] == ref[0], (B, mask, got[0], ref[0])
        assert abs(got[1] - ref[1]) <= 1e-6 * max(1.0, abs(ref[1]))
        assert torch.equal(got[2].cpu(), ref[2].cpu())


def test_engine_result_does_not_depend_on_the_order_of_the_mesh_file():
    """The engine works on an internal copy with vertices renumbered and triangles processed in Morton order; shuffling the
    vertex list and the triangle list of the input must not change losses or gradients (beyond which of two coincident
    triangles wins an exact depth tie at the mesh seam)."""
    sc = make_scene(16, 20, 60, 80, B=3, dist=1.8)
    w = dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8)
    eng, params = _engine(sc, w, [0.1])
    l0, g0 = eng.loss_and_grad()
    rng = np.random.RandomState(1)
    V, T_ = sc["pos"].shape[0], sc["tri"].shape[0]
    pv = rng.permutation(V)                      # new position of old vertex v
    inv = np.empty(V, np.int64)
    inv[pv] = np.arange(V)
    sc2 = dict(sc, pos=sc["pos"][inv], uv=sc["uv"][inv], vtx_color=sc["vtx_color"][inv], tri=pv[sc["tri"]][rng.permutation(T_)].astype(np.int32))
    eng2, _ = _engine(sc2, w, [0.1])
    l1, g1 = eng2.loss_and_grad()
    torch.cuda.synchronize()
    np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(g0.abs().max()))  # (round 5: a tenth of round 4's tolerance; passes at a third of this)


def test_engine_close_up_dense_mesh_scatter_variants_agree(monkeypatch):
    """A dense mesh seen from close (about two covered pixel centres per triangle): the engine picks the fragment-exchange
    variant of scatter_kernel from the observed segmentation mask.  Forced on, forced off, with the compacting variant and
    chosen automatically the optimisation is bit-identical (visibility is an order-independent atomicMin of exact keys), and iteration 0 of one
    hypothesis matches the oracle."""
    weights = dict(rgb=0.7, depth=1.0, mask=1.0)
    sc = make_scene(80, 128, 480, 640, B=32, dist=3.0, textured=True, tex_size=256)
    per_tri = 2.0 * float(sc["gt"]["segmentation"][..., 0].sum()) / sc["tri"].shape[0]
    assert per_tri > 1.5
    lrs = [0.2, 0.15, 0.1]
    runs = {}
    for mode in ("0", "1", "2", None):  # (2: the compacting variant, normally picked for long launches in the micro-polygon regime)
        if mode is None:
            monkeypatch.delenv("DDX_SCATTER_EXCHANGE", raising=False)
        else:
            monkeypatch.setenv("DDX_SCATTER_EXCHANGE", mode)
        eng, p = _engine(sc, weights, lrs)
        eng.run()
        eng.finish()
        eng.check()
        runs[mode] = (eng.losses().cpu().numpy().copy(), p.cpu().numpy().copy())
    for mode in ("1", "2", None):
        assert np.array_equal(runs["0"][0], runs[mode][0]) and np.array_equal(runs["0"][1], runs[mode][1])
    R = sc["oracle"]
    R.weights = {k: weights.get(k) for k in ("rgb", "depth", "mask", "edge")}
    total, logs, g_ref, _ = R.loss_and_grad(sc["params"][:, 5:6], sc["lr_mult"][5:6], global_B=32)
    for i, key in enumerate(KEYS):
        if key in logs:
            np.testing.assert_allclose(runs[None][0][0, i, 5], logs[key][0], rtol=5e-5, atol=1e-7)


def test_engine_large_batch_not_a_multiple_of_eight():
    """300 hypotheses (the update kernel then runs 2 slices per hypothesis instead of 8, the shade grid 3 slices): losses and
    the SGD step of sampled hypotheses match the oracle, which renders them with the batch-of-300 mean factor."""
    sc = make_scene(16, 20, 60, 80, B=300, dist=1.8)
    weights = dict(rgb=0.7, depth=1.0, mask=1.0)
    R = sc["oracle"]
    R.weights = {k: weights.get(k) for k in ("rgb", "depth", "mask", "edge")}
    lr = 0.25
    eng, p = _engine(sc, weights, [lr])
    eng.run()
    eng.finish()
    eng.check()
    lg = eng.losses()[0].cpu().numpy()
    pn = p.cpu().numpy()
    for j in (0, 7, 151, 299):
        total, logs, g_ref, _ = R.loss_and_grad(sc["params"][:, j:j + 1], sc["lr_mult"][j:j + 1], global_B=300)
        for i, key in enumerate(KEYS):
            if key in logs:
                np.testing.assert_allclose(lg[i, j], logs[key][0], rtol=5e-5, atol=1e-7)
        g_gpu = (sc["params"][:, j] - pn[:, j]) / lr
        np.testing.assert_allclose(g_gpu, g_ref[:, 0], rtol=5e-4, atol=5e-4 * np.abs(g_ref).max() + 2e-6)  # (round 5: a tenth of round 4's tolerance; passes at a third of this)


def test_forward_backward_pair_and_standalone_optimiser_steps():
    """ddx_render_loss_fwd / ddx_render_loss_bwd give the two outputs of ddx_engine_eval; ddx_sgd_step / ddx_adam_step applied to the
    evaluation pass's gradient reproduce the engine's own fused update bit for bit (SGD) / to rounding (Adam vs torch.optim.Adam)."""
    from diffdope_amd import _lib

    lib = _lib.load()
    sc = make_scene(16, 20, 60, 80, B=4, dist=1.8)
    weights = dict(rgb=0.7, depth=1.0, mask=1.0)
    lrs = [0.2, 0.15, 0.1]
    eng, p = _engine(sc, weights, lrs)
    l_ref, g_ref = eng.loss_and_grad()
    g = torch.empty_like(g_ref)
    l = torch.empty_like(l_ref)
    _lib.check(lib.ddx_render_loss_fwd(eng.handle, 0, l.data_ptr(), _lib.stream_ptr()), "fwd")
    _lib.check(lib.ddx_render_loss_bwd(eng.handle, 0, g.data_ptr(), _lib.stream_ptr()), "bwd")
    torch.cuda.synchronize()
    assert torch.equal(g, g_ref) and torch.equal(l, l_ref)
    # SGD: evaluation pass + stand-alone step == the engine's fused iteration
    p_manual = p.clone()
    _lib.check(lib.ddx_sgd_step(p_manual.data_ptr(), g.data_ptr(), lrs[0], p_manual.numel(), _lib.stream_ptr()), "sgd")
    eng.run(1)
    eng.finish()
    assert torch.equal(p_manual, p)
    # Adam against torch.optim.Adam on the same gradients
    x = torch.randn(7, 4, device="cuda")
    x_t = x.clone().requires_grad_(True)
    opt = torch.optim.Adam([x_t], lr=0.01, betas=(0.9, 0.999), eps=1e-8)
    m, v = torch.zeros_like(x), torch.zeros_like(x)
    for step in range(1, 6):
        gr = torch.randn(7, 4, device="cuda")
        x_t.grad = gr.clone()
        opt.step()
        _lib.check(lib.ddx_adam_step(x.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), 0.01, 0.9, 0.999, 1e-8, step, x.numel(),
                                     _lib.stream_ptr()), "adam")
    torch.cuda.synchronize()
    np.testing.assert_allclose(x.cpu().numpy(), x_t.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)



def test_backface_culling_of_closed_meshes_is_invisible_and_conditional():
    """Deviation D5: on a closed mesh the engine skips back-facing triangles of hypotheses that lie inside the view volume.
    (a) culling on / off give the same losses and gradients (bit-identical here: no pixel changes owner); (b) an OPEN mesh
    (some triangles removed) is never culled; (c) a hypothesis that pokes through the far plane draws both faces again and
    still matches the oracle, which applies the same rule."""
    sc = make_scene(16, 20, 60, 80, B=3, dist=1.8)
    w = dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8)
    res = []
    for cull in (True, False):
        eng, p = _engine(sc, w, [0.1], cull_backfaces=cull)
        losses, grad = eng.loss_and_grad()
        torch.cuda.synchronize()
        assert eng.cull_sign == (-1 if cull else 0)
        res.append((losses.cpu().numpy(), grad.cpu().numpy()))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-4, atol=1e-6 * np.abs(res[1][1]).max())
    # (b) open mesh: drop a band of triangles
    keep = np.ones(len(sc["tri"]), bool)
    keep[100:140] = False
    sc_open = dict(sc, tri=sc["tri"][keep])
    eng, _ = _engine(sc_open, w, [0.1], cull_backfaces=True)
    l_open, g_open = eng.loss_and_grad()
    torch.cuda.synchronize()
    assert eng.cull_sign == 0
    from oracle import oracle as orc

    kw = dict(uv=sc["uv"], tex=sc["tex"])
    Ro = orc.RenderOracle(sc["pos"], sc_open["tri"], sc["proj"], sc["H"], sc["W"], sc["gt"], w, dtype=np.float32, cull_backfaces=True, **kw)
    assert orc.mesh_cull_sign(sc["pos"], sc_open["tri"], sc["proj"]) == 0
    total, logs, g_ref, _ = Ro.loss_and_grad(sc["params"], sc["lr_mult"])
    np.testing.assert_allclose(g_open.cpu().numpy(), g_ref, rtol=2e-4, atol=2e-4 * np.abs(g_ref).max())  # (round 5: a tenth of round 4's tolerance; passes at a third of this)
    # (c) hypothesis 1 straddles the far plane (zfar = 200): its back faces are drawn again; losses still match the oracle
    far = sc["params"].copy()
    far[6, 1] = -200.0
    sc_far = dict(sc, params=far)
    eng, _ = _engine(sc_far, w, [0.1], cull_backfaces=True)
    l_far, g_far = eng.loss_and_grad()
    torch.cuda.synchronize()
    assert eng.cull_sign == -1 and eng.status()["outside_view_volume"] == 1  # (reported: hypothesis 1 is not entirely inside)
    R = sc["oracle"]
    R.cull_backfaces = True
    R.weights = {k: w.get(k) for k in ("rgb", "depth", "mask", "edge")}
    total, logs, g_ref, r_ref = R.loss_and_grad(far, sc["lr_mult"])
    lg = l_far.cpu().numpy()
    for i, key in enumerate(KEYS):
        np.testing.assert_allclose(lg[i], logs[key], rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(g_far.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-4 * np.abs(g_ref).max())
    # (d) two shells: culled when both are outward, not when the second is inside-out (same decision as the oracle's rule)
    n = len(sc["pos"])
    pos2 = np.concatenate([sc["pos"], sc["pos"] * np.float32(0.25) + np.float32(0.02)]).astype(np.float32)
    uv2 = np.concatenate([sc["uv"], sc["uv"]]) if sc["textured"] else None
    for second, want in ((sc["tri"] + n, -1), ((sc["tri"] + n)[:, [0, 2, 1]], 0)):
        tri2 = np.concatenate([sc["tri"], second]).astype(np.int32)
        sc2 = dict(sc, pos=pos2, tri=tri2, uv=uv2)
        eng, _ = _engine(sc2, w, [0.1], cull_backfaces=True)
        eng.loss_and_grad()  # (the mesh is analysed by the first run)
        torch.cuda.synchronize()
        assert eng.cull_sign == want == orc.mesh_cull_sign(pos2, tri2, sc["proj"])
    # (e) a flat two-sided patch next to the solid (edges pair up, the enclosed volume is round-off): no culling
    tri3 = np.concatenate([sc["tri"], sc["tri"][:30] + n, (sc["tri"][:30] + n)[:, [0, 2, 1]]]).astype(np.int32)
    eng, _ = _engine(dict(sc, pos=pos2, tri=tri3, uv=uv2), w, [0.1], cull_backfaces=True)
    eng.loss_and_grad()
    torch.cuda.synchronize()
    assert eng.cull_sign == 0 == orc.mesh_cull_sign(pos2, tri3, sc["proj"])


@pytest.mark.parametrize("deg,frac,tol_rad", [(1.0, 0.01, 1e-3), (10.0, 0.04, 5e-3), (40.0, 0.16, None)])
def test_pose_recovery_on_the_perturbation_tiers_of_the_dataset(deg, frac, tol_rad):
    """K7 on the three perturbation tiers the reference's dataset files are named after
    (data/*/*/scene_error_deg_{001,010,040}_trans_{001,004,016}.json, examples/run_bop_scene.py:27-89): the observation is rendered
    from a known pose, ONE initial guess is that pose rotated by `deg` about a random axis and moved by `frac` of its distance,
    and -- as in the reference -- every hypothesis starts from that same guess and differs only in its learning-rate
    multiplier (here 32 of them, log-spaced over 0.05 .. 500, the spread of the reference's learning_rates_bound [0.01, 100];
    SGD, the reference's decayed schedule, rgb + depth + mask).  On
    the 1 deg / 1 % tier the arg-min hypothesis must come back to the generating pose within 1e-3 rad / 1e-3 m (north_star
    tolerance); on the 10 deg / 4 % tier within 5e-3 rad / 1e-3 m (measured 3.4e-3 rad / 3e-4 m: the winner there is a x150
    multiplier, whose last steps are still coarse at the end of the 10x decay); the 40 deg / 16 % tier is outside the basin
    of a local method (measured: it settles in a neighbouring minimum with a lower loss, rotation error unchanged) -- only the
    loss decrease and a finite pose are asserted."""
    from oracle import oracle as orc

    sc = make_scene(40, 64, 120, 160, B=1, dist=4.0, tex_size=64)
    rng = np.random.RandomState(11)
    q0, t0 = syn.perturb_pose(sc["q_gt"], sc["t_gt"], deg, frac, rng)
    B = 32
    params = np.tile(np.concatenate([q0, t0])[:, None], (1, B)).astype(np.float32)
    lrm = np.geomspace(0.05, 500.0, B).astype(np.float32)
    sc = dict(sc, params=params, lr_mult=lrm, B=B)
    w = dict(rgb=0.7, depth=1.0, mask=1.0)
    lrs = [l * 0.02 for l in orc.lr_schedule(299, 20, 0.1)]
    eng, p = _engine(sc, w, lrs)
    eng.run()
    eng.finish()
    eng.check()
    lg = eng.losses().cpu().numpy()
    best = int(np.argmin(lg[-1].mean(0)))
    pb = p.cpu().numpy()[:, best]
    ang0, dt0 = syn.rotation_geodesic(q0, sc["q_gt"]), float(np.linalg.norm(t0 - sc["t_gt"])) * 0.1
    ang, dt = syn.rotation_geodesic(pb[:4], sc["q_gt"]), float(np.linalg.norm(pb[4:] - sc["t_gt"])) * 0.1
    print(f"tier {deg} deg / {100 * frac:.0f} %: start {ang0:.4f} rad {dt0:.4f} m -> arg-min hypothesis {best} (x{lrm[best]:.2f}): {ang:.2e} rad {dt:.2e} m, "
          f"loss {lg[0].sum(0)[best]:.4f} -> {lg[-1].sum(0)[best]:.4f}")
    assert abs(ang0 - np.radians(deg)) < 1e-6
    assert lg[-1].sum(0)[best] < lg[0].sum(0)[best]
    if tol_rad is not None:
        assert ang < tol_rad and dt < 1e-3, (ang, dt)
    else:
        assert np.all(np.isfinite(pb)) and ang < 1.2 * ang0


def test_engine_hypothesis_cut_by_the_camera_plane_is_clipped_like_the_oracle():
    """One hypothesis of the batch sits so close that the camera plane cuts the object (vertices at w <= 0): its straddling
    triangles are clipped at the near plane (tile pass), its back faces are drawn (D5 off for it), the status reports it, and
    losses and pose gradients of the whole batch match the oracle."""
    sc = make_scene(6, 8, 96, 128, B=3, dist=1.6, tex_size=16)
    near = sc["params"].copy()
    near[4:, 1] = [0.05, -0.02, -0.35]
    sc = dict(sc, params=near)
    w = dict(rgb=0.7, depth=1.0, mask=1.0)
    R = sc["oracle"]
    R.weights = {k: w.get(k) for k in ("rgb", "depth", "mask", "edge")}
    total, logs, g_ref, r_ref = R.loss_and_grad(near, sc["lr_mult"])
    assert (r_ref["pos_clip"][1, :, 3] <= 0).any() and (r_ref["rast"][1, ..., 3] > 0).mean() > 0.3
    eng, _ = _engine(sc, w, [0.1])
    losses, grad = eng.loss_and_grad()
    torch.cuda.synchronize()
    st = eng.status()
    assert st["outside_view_volume"] == 1 and st["big_triangles"] == 1
    lg = losses.cpu().numpy()
    for i, key in enumerate(KEYS):
        if key in logs:
            np.testing.assert_allclose(lg[i], logs[key], rtol=5e-5, atol=1e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-4 * np.abs(g_ref).max())
