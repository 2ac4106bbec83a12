"""GPU tests of the DiffDope class: the fused path and the op-by-op autograd path (user loss functions) give
the same optimisation, the results API (losses_values, optimization_results, get_argmin, get_pose) behaves
like the reference's, and a file-based run (PLY + PNGs + yaml-style config) recovers a known pose."""
import os

import numpy as np
import pytest
import torch

from tests.scenes import make_scene

pytestmark = pytest.mark.gpu


def _ddope(sc, losses, B, nb=5, **hp):
    import diffdope_amd as dd

    mesh = dd.Mesh.from_arrays(sc["pos"], sc["tri"], uv=sc["uv"], tex=sc["tex"])
    q, t = sc["params"][:4, 0], sc["params"][4:, 0]
    obj = dd.Object3D(position=list(t), rotation=list(q / np.linalg.norm(q)), batchsize=B, opencv2opengl=False, scale=1, mesh=mesh)
    scene = dd.Scene(tensor_rgb=dd.Image(img_tensor=torch.tensor(sc["gt"]["rgb"])), tensor_depth=dd.Image(img_tensor=torch.tensor(sc["gt"]["depth"])),
                     tensor_segmentation=dd.Image(img_tensor=torch.tensor(sc["gt"]["segmentation"])))
    cam = dd.Camera(fx=1, fy=1, cx=0, cy=0, im_width=sc["W"], im_height=sc["H"])
    cam.cam_proj = torch.tensor(sc["proj"], dtype=torch.float64)
    cfg = dict(losses=dict(l1_rgb_with_mask="rgb" in losses, weight_rgb=0.7, l1_depth_with_mask="depth" in losses, weight_depth=1.0,
                           l1_mask="mask" in losses, weight_mask=1.0, l1_edge="edge" in losses, weight_edge=0.6),
               hyperparameters=dict(nb_iterations=nb, batchsize=B, base_lr=hp.get("base_lr", 0.4), learning_rates_bound=[0.5, 2.0],
                                    learning_rate_base=1, lr_decay=0.1, seed=3))
    return dd.DiffDope(cfg=cfg, camera=cam, object3d=obj, scene=scene)


def test_fused_and_autograd_paths_agree_and_results_api():
    sc = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    B = 4
    a = _ddope(sc, ("rgb", "depth", "mask"), B)
    b = _ddope(sc, ("rgb", "depth", "mask"), B)
    assert torch.equal(a.learning_rates, b.learning_rates)  # seeded
    a.run_optimization(fused=True)
    b.run_optimization(fused=False)
    assert set(a.losses_values) == {"rgb", "depth", "mask_selection"} == set(b.losses_values)
    for k in a.losses_values:
        assert tuple(a.losses_values[k].shape) == (6, B)
        np.testing.assert_allclose(a.losses_values[k].numpy(), b.losses_values[k].numpy(), rtol=2e-3, atol=1e-6)
    pa, pb = a.object3d.params_tensor().cpu().numpy(), b.object3d.params_tensor().cpu().numpy()
    np.testing.assert_allclose(pa, pb, rtol=0, atol=2e-4)
    assert int(a.get_argmin()) == int(b.get_argmin())
    np.testing.assert_allclose(a.get_pose(), b.get_pose(), atol=2e-4)
    assert len(a.optimization_results) == 6 and tuple(a.optimization_results[0]["mtx"].shape) == (B, 4, 4)
    # images are rendered on demand from the stored poses and agree between the two paths
    ra, rb = a.optimization_results[-1]["rgb"], b.optimization_results[-1]["rgb"]
    assert tuple(ra.shape) == (B, 60, 80, 3) and float((ra - rb).abs().max()) < 5e-2
    img = a.render_img(batch_index=0)
    assert img.shape == (60, 80, 3) and img.dtype == np.uint8
    grid = a.render_img()
    assert grid.ndim == 3 and grid.shape[0] > 60 and grid.shape[1] > 80 * 3
    curves = a.plot_losses()
    assert curves.ndim == 3 and curves.shape[2] == 3
    import os, tempfile
    with tempfile.TemporaryDirectory() as td:
        out = a.make_animation(os.path.join(td, "anim.gif"))
        assert os.path.getsize(out) > 1000


def test_engine_is_reused_across_runs_and_frames():
    """A second run_optimization on the same DiffDope -- same frame, or a new observation of the same object -- keeps the fused
    engine (mesh half of its set-up is not repeated: RefineEngine.new_observation / ddx_engine_new_observation) and gives
    bit for bit what a fresh DiffDope gives; changing the loss set builds a new engine."""
    import diffdope_amd as dd

    sc_a = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    sc_b = make_scene(16, 20, 60, 80, B=1, dist=2.1)  # same mesh and texture, another observation (and initial pose)
    B = 4
    scene_of = lambda sc: dd.Scene(tensor_rgb=dd.Image(img_tensor=torch.tensor(sc["gt"]["rgb"])), tensor_depth=dd.Image(img_tensor=torch.tensor(sc["gt"]["depth"])),
                                   tensor_segmentation=dd.Image(img_tensor=torch.tensor(sc["gt"]["segmentation"])))

    def result(d):
        return d.object3d.params_tensor().cpu().clone(), {k: v.clone() for k, v in d.losses_values.items()}, d.get_pose().copy()

    def same(x, y):
        return torch.equal(x[0], y[0]) and set(x[1]) == set(y[1]) and all(torch.equal(x[1][k], y[1][k]) for k in x[1]) and np.array_equal(x[2], y[2])

    d = _ddope(sc_a, ("rgb", "depth", "mask"), B)
    d.run_optimization(fused=True)
    first = result(d)
    eng = d.last_engine
    d.object3d.reset_pose()
    d.run_optimization(fused=True)  # the same frame again
    assert d.last_engine is eng and same(result(d), first)
    # the next frame of the same object
    d.scene = scene_of(sc_b)
    d.scene.cuda()
    d.scene.set_batchsize(B)
    d.object3d.reset_pose()
    d.run_optimization(fused=True)
    assert d.last_engine is eng
    fresh = _ddope(sc_a, ("rgb", "depth", "mask"), B)
    fresh.scene = scene_of(sc_b)
    fresh.scene.cuda()
    fresh.scene.set_batchsize(B)
    fresh.run_optimization(fused=True)
    assert fresh.last_engine is not eng and same(result(d), result(fresh))
    assert not same(result(d), first)
    # another loss set: not the same engine
    d.loss_functions = d.loss_functions[:2]
    d.object3d.reset_pose()
    d.run_optimization(fused=True)
    assert d.last_engine is not eng and set(d.losses_values) == {"rgb", "depth"}


def test_edge_extension_fused_and_autograd_paths_agree():
    """cfg.losses.l1_edge (this build's extension): the fused engine's edge role and the torch conv2d loss of the
    op-by-op path optimise alike."""
    sc = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    B = 3
    a = _ddope(sc, ("rgb", "edge"), B, nb=4, base_lr=0.2)
    b = _ddope(sc, ("rgb", "edge"), B, nb=4, base_lr=0.2)
    a.run_optimization(fused=True)
    b.run_optimization(fused=False)
    assert set(a.losses_values) == {"rgb", "edge"} == set(b.losses_values)
    for k in a.losses_values:
        np.testing.assert_allclose(a.losses_values[k].numpy(), b.losses_values[k].numpy(), rtol=3e-3, atol=1e-6)
    pa, pb = a.object3d.params_tensor().cpu().numpy(), b.object3d.params_tensor().cpu().numpy()
    np.testing.assert_allclose(pa, pb, rtol=0, atol=3e-4)


def test_user_loss_function_forces_the_autograd_path():
    import diffdope_amd as dd

    sc = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    d = _ddope(sc, ("mask",), 2, nb=2)

    def my_loss(ddope):  # the documented extension point: f(ddope) -> scalar, reading ddope.renders / gt_tensors
        v = torch.mean(torch.abs(ddope.renders["depth"] - ddope.gt_tensors["depth"]) * ddope.gt_tensors["segmentation"][..., 0], (1, 2))
        ddope.add_loss_value("my_depth", v)
        return (v * ddope.learning_rates).mean()

    d.loss_functions.append(my_loss)
    with pytest.raises(RuntimeError):
        d.run_optimization(fused=True)
    p0 = d.object3d.params_tensor().clone()
    d.run_optimization()
    assert "my_depth" in d.losses_values and tuple(d.losses_values["my_depth"].shape) == (3, 2)
    assert float((d.object3d.params_tensor() - p0).abs().max()) > 0


def test_file_based_run_recovers_pose(tmp_path):
    """PLY + PNG files + yaml-style config, like examples/simple_scene.py, at a small resolution."""
    import diffdope_amd as dd
    from diffdope_amd import api, synthetic as syn
    from PIL import Image as PILImage

    H, W = 120, 160
    pos, tri, uv = syn.blob_mesh(24, 32, seed=0)
    tex = (syn.texture(64, seed=1) * 255).astype(np.uint8)
    PILImage.fromarray(tex).save(tmp_path / "blob.png")
    with open(tmp_path / "blob.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment TextureFile blob.png\n")
        f.write(f"element vertex {len(pos)}\nproperty float x\nproperty float y\nproperty float z\nproperty float texture_u\nproperty float texture_v\n")
        f.write(f"element face {len(tri)}\nproperty list uchar int vertex_indices\nend_header\n")
        for p, t in zip(pos * 100.0, uv):
            f.write(f"{p[0]:.5f} {p[1]:.5f} {p[2]:.5f} {t[0]:.6f} {1 - t[1]:.6f}\n")
        for t in tri:
            f.write(f"3 {t[0]} {t[1]} {t[2]}\n")
    intr = syn.camera_intrinsics(W, H)
    t_cv, q_cv = np.array([5.0, -8.0, 300.0]), syn.quat_from_axis_angle([0.3, 1.0, 0.2], 0.4)
    cam = dd.Camera(**intr)
    mesh = dd.Mesh(str(tmp_path / "blob.ply"), scale=0.01)
    obj = dd.Object3D(position=list(t_cv), rotation=list(q_cv), batchsize=1, scale=0.01, mesh=mesh)
    obj.cuda(); cam.cuda(); cam.set_batchsize(1); obj.set_batchsize(1)
    with torch.no_grad():
        r = obj()
        mtx_gt = dd.matrix_batch_44_from_position_quat(p=r["trans"], q=r["quat"])
        o = dd.render_texture_batch(dd.RasterizeGLContext(), cam.cam_proj, mtx_gt, r["pos"], r["pos_idx"], [H, W], uv=r["uv"],
                                    uv_idx=r["uv_idx"], tex=r["tex"], return_rast_out=True)
    cov = (o["rast_out"][0, ..., 3] > 0).cpu().numpy()[::-1]
    PILImage.fromarray((o["rgb"][0].clamp(0, 1).cpu().numpy()[::-1] * 255).round().astype(np.uint8)).save(tmp_path / "rgb.png")
    PILImage.fromarray((o["depth"][0].cpu().numpy()[::-1] * 100.0 * cov).round().astype(np.uint16)).save(tmp_path / "depth.png")
    PILImage.fromarray((cov * 255).astype(np.uint8), mode="L").save(tmp_path / "seg.png")
    rng = np.random.RandomState(5)
    q0, t0 = syn.perturb_pose(q_cv, t_cv, 3.0, 0.01, rng)
    cfg = dict(camera=intr, scene=dict(path_img=str(tmp_path / "rgb.png"), path_depth=str(tmp_path / "depth.png"),
                                       path_segmentation=str(tmp_path / "seg.png"), image_resize=1.0),
               object3d=dict(position=list(t0), rotation=list(api.matrix_from_quat(q0).reshape(-1)), scale=0.01, model_path=str(tmp_path / "blob.ply")),
               losses=dict(l1_rgb_with_mask=True, weight_rgb=0.7, l1_depth_with_mask=True, weight_depth=1.0, l1_mask=True, weight_mask=1.0),
               hyperparameters=dict(nb_iterations=150, batchsize=8, base_lr=0.1, learning_rates_bound=[0.5, 3.0], learning_rate_base=1,
                                    lr_decay=0.1, seed=1))
    d = dd.DiffDope(cfg=cfg)
    assert d.resolution == [H, W]
    d.run_optimization(optimizer="adam")
    pose = d.get_pose()
    gt = mtx_gt[0].cpu().numpy()
    ang = syn.matrix_rotation_geodesic(pose[:3, :3], gt[:3, :3])
    dt = np.linalg.norm(pose[:3, 3] - gt[:3, 3]) * 0.1
    first, last = d.losses_values["rgb"][0].mean(), d.losses_values["rgb"][-1].min()
    assert last < first
    assert ang < 5e-3 and dt < 2e-3, (ang, dt)


def test_deferred_runs_on_separate_streams_equal_blocking_runs():
    """run_optimization(wait=False) + finish_optimization(): two objects enqueued on one stream each give the results of two
    blocking runs, bit for bit."""
    sc = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    ref = []
    for losses in (("rgb", "mask"), ("depth", "mask")):
        d = _ddope(sc, losses, 4)
        d.run_optimization(fused=True)
        ref.append((d.object3d.params_tensor().clone(), {k: v.clone() for k, v in d.losses_values.items()}))
    runs = []
    main = torch.cuda.current_stream()
    for losses in (("rgb", "mask"), ("depth", "mask")):
        d = _ddope(sc, losses, 4)
        st = torch.cuda.Stream()
        st.wait_stream(main)
        with torch.cuda.stream(st):
            d.run_optimization(fused=True, wait=False)
        runs.append(d)
    for d, (p, lv) in zip(runs, ref):
        assert not d.losses_values  # nothing fetched yet
        d.finish_optimization()
        assert torch.equal(d.object3d.params_tensor(), p)
        assert set(d.losses_values) == set(lv) and all(torch.equal(d.losses_values[k], lv[k]) for k in lv)


@pytest.mark.parametrize("H,W", [(37, 53), (36, 52)])  # (element counts not divisible / divisible by 4: scalar and 4-wide kernels)
def test_masked_l1_mean_matches_the_torch_expression(H, W):
    """render.masked_l1_mean (one forward + one backward kernel) against the reference's torch expressions
    (diffdope.py:547-613), values and gradients, incl. batched stride-0 views of the observed image and the
    channel-0 mask of the depth term."""
    from diffdope_amd import render
    from diffdope_amd.render import masked_l1_mean

    hits = []
    orig = render._masked_l1_func.apply
    monkey = lambda *a: (hits.append(a[3]), orig(*a))[1]
    g = torch.Generator(device="cuda").manual_seed(0)
    B = 5
    seg = (torch.rand(1, H, W, 3, device="cuda", generator=g) > 0.4).float()
    render._masked_l1_func.apply = monkey
    try:
        for shape, ch0 in (((B, H, W, 3), False), ((B, H, W), True)):
            x = torch.randn(shape, device="cuda", generator=g)
            y = torch.randn((1,) + shape[1:], device="cuda", generator=g)
            masked_l1_mean(x, y.expand(shape), seg.expand(B, H, W, 3), mask_channel0=ch0)
    finally:
        render._masked_l1_func.apply = orig
    assert hits == [1, 3]  # both the rgb-shaped and the depth-shaped (channel-0 mask) calls took the single-kernel path
    for shape, ch0 in (((B, H, W, 3), False), ((B, H, W), True)):
        x = torch.randn(shape, device="cuda", generator=g, requires_grad=True)
        y = torch.randn((1,) + shape[1:], device="cuda", generator=g)
        yb, sb = y.expand(shape), seg.expand(B, H, W, 3)  # batch stride 0, as DiffDope holds its gt tensors
        lr = torch.rand(B, device="cuda", generator=g)
        ref = torch.mean(torch.abs((x - yb) * (sb[..., 0] if ch0 else sb)), tuple(range(1, x.dim())))
        out = masked_l1_mean(x, yb, sb, mask_channel0=ch0)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6)
        g_ref, = torch.autograd.grad((ref * lr).mean(), x)
        g_out, = torch.autograd.grad((out * lr).mean(), x)
        np.testing.assert_allclose(g_out.cpu().numpy(), g_ref.cpu().numpy(), rtol=1e-6, atol=1e-12)
    x = torch.rand((B, H, W, 3), device="cuda", generator=g, requires_grad=True)
    ref = torch.mean(torch.abs(x - seg), (1, 2, 3))
    out = masked_l1_mean(x, seg)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6)


def test_pose_matrix_kernel_matches_the_reference_goldens():
    """matrix_batch_44_from_position_quat on ROCm tensors (ddx_pose_matrix_fwd / _bwd, one kernel each way) against the golden
    vectors generated from the reference's own function (tests/golden/g2_pose.npz: values and gradients w.r.t. the 7 parameters
    through the q / |q| normalisation of Object3D.forward), and against the torch-expression fallback."""
    import diffdope_amd as dd
    from diffdope_amd import pose

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g2_pose.npz"))
    outs = []
    for dev in ("cuda", "cpu"):
        params = [torch.tensor(g["params"][i], device=dev, requires_grad=True) for i in range(7)]
        q = torch.stack(params[:4], dim=0).T
        q = q / torch.norm(q, dim=1).reshape(-1, 1)
        mtx = dd.matrix_batch_44_from_position_quat(p=torch.stack(params[4:], dim=0).T, q=q)
        np.testing.assert_allclose(mtx.detach().cpu().numpy(), g["mtx"], rtol=1e-6, atol=1e-6)
        mtx.backward(torch.tensor(g["dmtx"], device=dev))
        grads = np.stack([p.grad.cpu().numpy() for p in params])
        np.testing.assert_allclose(grads, g["dparams"], rtol=1e-4, atol=1e-5)
        outs.append((mtx.detach().cpu().numpy(), grads))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6)
    assert pose._pose_matrix_func is not None


def test_pose_head_kernel_matches_the_reference_goldens():
    """pose.quat_trans_from_parameters (ddx_pose_pack_fwd / _bwd: Object3D.forward's stack / norm / divide / stack, diffdope.py:
    1085-1098, one kernel each way) followed by matrix_batch_44_from_position_quat: the matrices and the gradients of the seven
    parameters of tests/golden/g2_pose.npz (made by the reference's own functions), and the torch-expression fallback on the CPU."""
    import diffdope_amd as dd
    from diffdope_amd import pose

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g2_pose.npz"))
    hits = []
    orig = pose._pose_pack_func.apply
    pose._pose_pack_func.apply = lambda *a: (hits.append(len(a)), orig(*a))[1]
    try:
        outs = []
        for dev in ("cuda", "cpu"):
            params = [torch.tensor(g["params"][i], device=dev, requires_grad=True) for i in range(7)]
            q, t = pose.quat_trans_from_parameters(*params)
            assert tuple(q.shape) == (params[0].shape[0], 4) and tuple(t.shape) == (params[0].shape[0], 3)
            np.testing.assert_allclose(q.detach().norm(dim=1).cpu().numpy(), 1.0, rtol=0, atol=2e-7)
            mtx = dd.matrix_batch_44_from_position_quat(p=t, q=q)
            np.testing.assert_allclose(mtx.detach().cpu().numpy(), g["mtx"], rtol=1e-6, atol=1e-6)
            mtx.backward(torch.tensor(g["dmtx"], device=dev))
            grads = np.stack([p.grad.cpu().numpy() for p in params])
            np.testing.assert_allclose(grads, g["dparams"], rtol=1e-4, atol=1e-5)
            outs.append((mtx.detach().cpu().numpy(), grads))
    finally:
        pose._pose_pack_func.apply = orig
    assert hits == [7]  # (the ROCm tensors took the kernel, the CPU ones the expressions)
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-6)
    # only the translation is used downstream: the quaternion's gradient is zero, not missing
    params = [torch.tensor(g["params"][i], device="cuda", requires_grad=True) for i in range(7)]
    q, t = pose.quat_trans_from_parameters(*params)
    t.sum().backward()
    assert all(float(p.grad.abs().max()) == 0.0 for p in params[:4]) and all(torch.equal(p.grad, torch.ones_like(p)) for p in params[4:])


def test_captured_iteration_of_the_op_by_op_path_equals_the_eager_loop():
    """run_optimization(fused=False, graph=True): two eager iterations, then ONE captured iteration replayed for the rest of the
    schedule (learning rate, loss rows and pose log indexed by a device-side counter inside the graph) -- against the eager loop
    (diffdope/diffdope.py:1656-1714 as the reference runs it): losses per iteration, poses per iteration, final parameters, the
    selected hypothesis and the images left in `renders`; with a user loss function in the list (capture-safe by declaration:
    graph=True); and the same object again, after its logs have been read and host memory has churned."""
    import diffdope_amd as dd

    sc = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    B, nb = 4, 9

    def my_loss(ddope):  # a user loss: silhouette area against the observed one
        v = (ddope.renders["mask"].mean((1, 2, 3)) - ddope.gt_tensors["segmentation"].mean((1, 2, 3))).abs()
        ddope.add_loss_value("area", v.detach())
        return (v * ddope.learning_rates).mean() * 0.3

    runs = {}
    for graph in (False, True):
        d = _ddope(sc, ("rgb", "depth", "mask"), B, nb=nb)
        d.loss_functions.append(my_loss)
        d.run_optimization(fused=False, graph=graph)
        runs[graph] = d
    a, b = runs[False], runs[True]
    assert set(a.losses_values) == set(b.losses_values) == {"rgb", "depth", "mask_selection", "area"}
    for k in a.losses_values:
        assert tuple(b.losses_values[k].shape) == (nb + 1, B)
        # (two EAGER runs of this loop differ by up to 4e-5 in the later rows: the backward passes accumulate with floating-point
        # atomics, and the difference of one step is carried through the rest of the schedule)
        np.testing.assert_allclose(a.losses_values[k].numpy(), b.losses_values[k].numpy(), rtol=3e-4, atol=1e-7)
    assert len(b.optimization_results) == nb + 1
    for ra, rb in zip(a.optimization_results, b.optimization_results):
        np.testing.assert_allclose(ra["mtx"].numpy(), rb["mtx"].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(a.object3d.params_tensor().cpu().numpy(), b.object3d.params_tensor().cpu().numpy(), rtol=0, atol=1e-5)
    assert int(a.get_argmin()) == int(b.get_argmin())
    # (a pose that differs in its sixth digit moves the texture lookups of a sharp texture by 1e-4 of a texel)
    np.testing.assert_allclose(a.renders["rgb"].detach().cpu().numpy(), b.renders["rgb"].detach().cpu().numpy(), atol=3e-3)
    # the same object again from its start pose (its logs have been read, host memory has churned): the same numbers
    first = {k: v.clone() for k, v in b.losses_values.items()}
    junk = [np.zeros(n) for n in (10, 1000, 100000)]
    b.object3d.load_params_tensor(_ddope(sc, ("rgb",), B).object3d.params_tensor())
    b.run_optimization(fused=False, graph=True)
    assert len(b.optimization_results) == nb + 1 and len(junk) == 3
    for k in first:
        np.testing.assert_allclose(first[k].numpy(), b.losses_values[k].numpy(), rtol=3e-4, atol=1e-7)


def test_captured_iteration_called_again_and_again_with_the_results_read_in_between(monkeypatch):
    """run_optimization(fused=False, graph=True) three times on one object with get_argmin / the logs / a lazily rendered image
    (through the rasteriser context the captured kernels point into) read in between: every call reproduces the eager loop from
    the same start.  Round 6: the captured iteration indexes its learning-rate table, loss-row buffers and
    pose log with a DEVICE-SIDE counter that the graph increments itself -- it must stand at the first iteration to replay before
    the first replay and at the schedule's length after the last (one replay more and index_select / index_copy_ run past their
    buffers: tools/graph_fault_repro.py 8 6, the queue aborts with 0x1016).  DDX_DEBUG_KEEP_GRAPH keeps the tables for the check."""
    monkeypatch.setenv("DDX_DEBUG_KEEP_GRAPH", "1")
    sc = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    B, nb = 4, 9
    ref = _ddope(sc, ("rgb", "depth", "mask"), B, nb=nb)
    ref.run_optimization(fused=False)
    p_ref = ref.object3d.params_tensor().cpu().numpy()
    l_ref = {k: v.numpy().copy() for k, v in ref.losses_values.items()}
    d = _ddope(sc, ("rgb", "depth", "mask"), B, nb=nb)
    p0 = [p.detach().clone() for p in d.object3d.parameters()]
    for call in range(3):
        with torch.no_grad():
            for p, q in zip(d.object3d.parameters(), p0):
                p.copy_(q)
        d.run_optimization(fused=False, graph=True)
        k = d._kept_graph
        assert int(k["cap"]["it"]) == nb + 1 == k["lr_table"].numel() == k["mtx_log"].shape[0]  # (the counter: exactly at the end)
        np.testing.assert_allclose(d.object3d.params_tensor().cpu().numpy(), p_ref, rtol=0, atol=2e-4)
        for key in l_ref:
            assert tuple(d.losses_values[key].shape) == (nb + 1, B)
            np.testing.assert_allclose(d.losses_values[key].numpy(), l_ref[key], rtol=2e-3, atol=1e-6)
        assert int(d.get_argmin()) == int(ref.get_argmin())
        assert tuple(d.optimization_results[-1]["rgb"].shape) == (B, 60, 80, 3)
        # the kept graph of THIS call replays once more when its counter is put back first (the mechanism a kept graph needs)
        k["cap"]["it"].fill_(k["n_eager"])
        k["g"].replay()
        torch.cuda.synchronize()
        assert int(k["cap"]["it"]) == k["n_eager"] + 1
        d._kept_graph = None
