"""The compat switch for deviation D2 (ddx.h DDX_COMPAT_UNCLAMPED_BARY_GRAD): op-level rasterize backward, the fused G-buffer
backward and the fused engine against the oracle under the same switch, and the DiffDope(cfg) route to it and to the culling
switch (D5)."""
import numpy as np
import pytest
import torch

from tests.scenes import clip_from_pixels, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture
def nvdiffrast_compat():
    from diffdope_amd import render
    from oracle import oracle as orc

    old_lib = render.set_compat("nvdiffrast")
    old_orc = orc.set_compat(True)
    yield
    render.set_compat(None)
    from diffdope_amd import _lib

    _lib.load().ddx_set_compat(old_lib)
    orc.set_compat(old_orc)


def test_rasterize_backward_unclamped_matches_oracle_and_differs_only_on_saturated_pixels(nvdiffrast_compat):
    import diffdope_amd as dd
    from diffdope_amd import render
    from oracle import oracle as orc

    H, W = 8, 8
    xe = 4.5 - 0.4 / 256  # (the saturated pixel of tests/test_oracle_deviations.py::test_d2...)
    P = clip_from_pixels(np.array([[0.5, 4.0], [xe, 0.5], [xe, 7.5]]), H, W, z=0.0)[None]
    tri = np.array([[0, 1, 2]], np.int32)
    rast = orc.rasterize_fwd(P, tri, H, W)
    assert rast[0, 3, 4, 3] == 1 and rast[0, 3, 4, 0] == 0.0
    rng = np.random.RandomState(0)
    d = np.zeros_like(rast)
    d[..., :2] = rng.normal(size=rast[..., :2].shape)
    g_ref = orc.rasterize_bwd(P, tri, rast, d)
    pos = torch.tensor(P, device="cuda", requires_grad=True)
    r, _ = dd.rasterize(dd.RasterizeGLContext(), pos, torch.tensor(tri, device="cuda"), [H, W])
    (r * torch.tensor(d, device="cuda")).sum().backward()
    np.testing.assert_allclose(pos.grad.cpu().numpy(), g_ref, rtol=1e-4, atol=1e-6)
    # without the switch the saturated pixel passes nothing: the two gradients differ
    render.set_compat(None)
    orc.set_compat(False)
    g_own = orc.rasterize_bwd(P, tri, rast, d)
    assert np.abs(g_own - g_ref).max() > 1e-3


@pytest.mark.parametrize("fused", [True, False])
def test_engine_and_gbuffer_with_compat_match_the_oracle_with_compat(nvdiffrast_compat, fused):
    """Thin triangles (a 40 x 48 mesh on 48x64 pixels: many pixels with a saturated barycentric): the engine built with
    compat="nvdiffrast" against the oracle under the same switch, and the materialising path through the process-wide switch."""
    import diffdope_amd as dd

    sc = make_scene(40, 48, 48, 64, B=3, dist=2.0)
    R = sc["oracle"]
    w = dict(rgb=0.7, depth=1.0)
    R.weights = dict(rgb=0.7, depth=1.0, mask=None, edge=None)
    total, logs, g_ref, _ = R.loss_and_grad(sc["params"], sc["lr_mult"])
    T = lambda a, **k: torch.tensor(np.ascontiguousarray(a), device="cuda", **k)
    if fused:
        params = T(sc["params"])
        eng = dd.RefineEngine(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [sc["H"], sc["W"]], {k: T(v) for k, v in sc["gt"].items()}, params,
                              T(sc["lr_mult"]), [0.1], w, uv=T(sc["uv"]), tex=T(sc["tex"]), compat="nvdiffrast")
        losses, grad = eng.loss_and_grad()
        np.testing.assert_allclose(grad.cpu().numpy(), g_ref, rtol=2e-4, atol=2e-5 * np.abs(g_ref).max())  # (round 5: a tenth of round 4's tolerance; passes at a third of this)
        eng0 = dd.RefineEngine(T(sc["pos"]), T(sc["tri"]), T(sc["proj"]), [sc["H"], sc["W"]], {k: T(v) for k, v in sc["gt"].items()}, params,
                               T(sc["lr_mult"]), [0.1], w, uv=T(sc["uv"]), tex=T(sc["tex"]))
        _, grad0 = eng0.loss_and_grad()
        assert not torch.equal(grad, grad0)  # (the clamped rule gives another gradient on this scene: the switch does something)
    else:
        from oracle import oracle as orc

        R.cull_backfaces = False  # the op-level ops draw both faces, like nvdiffrast
        total, logs, g_ref, _ = R.loss_and_grad(sc["params"], sc["lr_mult"])
        B = sc["B"]
        params = [T(sc["params"][i], requires_grad=True) for i in range(7)]
        q = torch.stack(params[:4], dim=0).T
        q = q / torch.norm(q, dim=1).reshape(-1, 1)
        mtx = dd.matrix_batch_44_from_position_quat(p=torch.stack(params[4:], dim=0).T, q=q)
        ex = lambda a: T(a)[None].expand(B, *a.shape)
        grads = []
        for fused_ops in (True, False):  # ddx_gbuffer_bwd, and ddx_rasterize_bwd behind the separate ops
            for p_ in params:
                p_.grad = None
            out = dd.render_texture_batch(dd.RasterizeGLContext(), ex(sc["proj"]), mtx, ex(sc["pos"]), ex(sc["tri"]), [sc["H"], sc["W"]],
                                          uv=ex(sc["uv"]), uv_idx=ex(sc["tri"]), tex=ex(sc["tex"]), fused=fused_ops)
            gt = {k: T(v)[None] for k, v in sc["gt"].items()}
            lrm = T(sc["lr_mult"])
            loss = 0.7 * (torch.mean(torch.abs((out["rgb"] - gt["rgb"]) * gt["segmentation"]), (1, 2, 3)) * lrm).mean()
            loss = loss + 1.0 * (torch.mean(torch.abs((out["depth"] - gt["depth"]) * gt["segmentation"][..., 0]), (1, 2)) * lrm).mean()
            assert abs(float(loss.detach()) - total) < 2e-5 * max(1.0, abs(total))
            loss.backward(retain_graph=True)
            g = np.stack([p_.grad.cpu().numpy() for p_ in params])
            np.testing.assert_allclose(g, g_ref, rtol=3e-4, atol=3e-4 * np.abs(g_ref).max())  # (round 5: a tenth of round 4's tolerance; passes at a third of this)
            grads.append(g)


def test_diffdope_cfg_reaches_the_culling_and_compat_switches():
    """cfg.hyperparameters.cull_backfaces / .compat (D5 / D2 from the config).  The DiffDope API draws both faces by default, like
    dr.rasterize (round 5); culling is RefineEngine's own default and an option here."""
    import diffdope_amd as dd
    from diffdope_amd import synthetic as syn

    H, W = 60, 80
    pos, tri, uv = syn.blob_mesh(16, 20, seed=0)
    mesh = dd.Mesh.from_arrays(pos * 100.0, tri, uv=uv, tex=syn.texture(32, seed=1), scale=0.01)
    cam = dd.Camera(**syn.camera_intrinsics(W, H))
    seg = torch.zeros(H, W, 3)
    seg[20:40, 30:50] = 1.0
    for hp_extra, want_cull, want_compat in ((dict(), False, 0), (dict(cull_backfaces=True), True, 0), (dict(compat="nvdiffrast"), False, 1)):
        cfg = dict(losses=dict(l1_rgb_with_mask=False, weight_rgb=0.7, l1_depth_with_mask=False, weight_depth=1.0, l1_mask=True, weight_mask=1.0),
                   hyperparameters=dict(nb_iterations=2, batchsize=2, base_lr=0.1, learning_rates_bound=[0.5, 3.0], learning_rate_base=1,
                                        lr_decay=0.1, seed=2, **hp_extra))
        obj = dd.Object3D(position=[0.0, 0.0, 300.0], rotation=[0.0, 0.0, 0.0, 1.0], batchsize=2, scale=0.01, mesh=mesh)
        sc = dd.Scene(tensor_segmentation=dd.Image(img_tensor=seg))
        d = dd.DiffDope(cfg=cfg, camera=cam, object3d=obj, scene=sc)
        d.run_optimization()
        e = d.last_engine
        assert e.desc.no_backface_cull == int(not want_cull) and e.desc.compat == want_compat
        assert (e.cull_sign != 0) == want_cull
