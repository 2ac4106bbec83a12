"""The silhouette kept as ONE channel on the materialising path (`-m gpu`; round 5: ddx_gbuffer_fwd_rows_c, ddx_silhouette_*_rows_c,
ddx_masked_l1_bc3_*).

The reference's `mask` output is dr.antialias of the interpolation of a [T,3] tensor of ones (diffdope/diffdope.py:212-214): three
equal channels.  render_texture_batch stores one and returns its expand(..., 3); l1_mask (diffdope.py:583-613) compares that one
channel with the three of the observed segmentation in one kernel each way.  Held here against the three-copy form of the same
passes and against the reference's torch expressions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(name, B=6):
    import diffdope_amd as dd
    from diffdope_amd import workloads as wl

    dev = torch.device("cuda")
    w = wl.build(name, dev, B=B)
    p = w["params0"].clone()
    p[5, 1] += 2.2   # towards the upper border
    p[4, 3] += 40.0  # out of the frame: draws nothing
    ex = lambda t: t[None].expand(B, *t.shape)
    kw = dict(uv=ex(w["uv"]), uv_idx=ex(w["tri"]), tex=ex(w["tex"])) if w["uv"] is not None else dict(vtx_color=ex(w["vtx_color"]))
    q = p[:4].T / torch.norm(p[:4].T, dim=1, keepdim=True)
    mtx = dd.matrix_batch_44_from_position_quat(q=q, p=p[4:].T).detach()
    return w, ex, kw, mtx


@pytest.mark.parametrize("name", ["cfg2", "cfg4"])
def test_one_stored_channel_equals_three(name):
    """rgb, depth and every channel of mask: the bits of the three-copy passes; the gradient of a loss that weighs the three
    channels differently: equal up to the order of the floating-point additions (autograd sums the channels of the view, the
    three-copy backward accumulates them per pair)."""
    from diffdope_amd.render import RasterizeContext, render_texture_batch

    w, ex, kw, mtx0 = _scene(name)
    B, H, W = mtx0.shape[0], w["H"], w["W"]
    outs = []
    for compact in (False, True):
        mtx = mtx0.clone().requires_grad_(True)
        r = render_texture_batch(RasterizeContext(), ex(w["proj"]), mtx, ex(w["pos"]), ex(w["tri"]), [H, W], fused=True, compact_mask=compact, **kw)
        assert tuple(r["mask"].shape) == (B, H, W, 3)
        assert (r["mask"].stride(-1) == 0) == compact
        g = torch.Generator(device="cpu").manual_seed(11)
        wr, wd, wm = (torch.rand(r[k].shape, generator=g).to(mtx.device) for k in ("rgb", "depth", "mask"))
        loss = (r["rgb"] * wr).sum() + (r["depth"] * wd).sum() + (r["mask"] * wm).sum()
        (grad,) = torch.autograd.grad(loss, mtx)
        outs.append((r["rgb"].detach().clone(), r["depth"].detach().clone(), r["mask"].detach().contiguous().clone(), grad.clone()))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert torch.equal(a, b)
    assert float(outs[0][2].sum()) > 0
    ga, gb = outs[0][3], outs[1][3]
    assert torch.allclose(ga, gb, rtol=2e-4, atol=2e-4 * float(ga.abs().max()))


def test_default_mask_is_an_ordinary_tensor_as_the_reference_returns_it():
    """render_texture_batch(...)["mask"] without asking for anything (diffdope.py:212-214 returns an ordinary
    tensor): contiguous [B,H,W,3], mask.view(B, -1) works, an in-place write works and does not reach rgb / depth; the values are
    those of the compact form, which only DiffDope's built-in loss loop asks for."""
    from diffdope_amd.render import RasterizeContext, render_texture_batch

    w, ex, kw, mtx = _scene("cfg2")
    B, H, W = mtx.shape[0], w["H"], w["W"]
    r = render_texture_batch(RasterizeContext(), ex(w["proj"]), mtx, ex(w["pos"]), ex(w["tri"]), [H, W], **kw)
    m = r["mask"]
    assert tuple(m.shape) == (B, H, W, 3) and m.is_contiguous() and m.stride(-1) == 1
    flat = m.view(B, -1)
    assert flat.shape == (B, H * W * 3)
    c = render_texture_batch(RasterizeContext(), ex(w["proj"]), mtx, ex(w["pos"]), ex(w["tri"]), [H, W], compact_mask=True, **kw)
    assert c["mask"].stride(-1) == 0 and torch.equal(c["mask"].contiguous(), m)
    with pytest.raises(RuntimeError):
        c["mask"].view(B, -1)
    before, rgb0 = m.clone(), r["rgb"].clone()
    m.mul_(0.5)
    m[..., 1] += 1.0
    assert torch.equal(m[..., 0], before[..., 0] * 0.5) and torch.equal(m[..., 1], before[..., 1] * 0.5 + 1.0) and torch.equal(r["rgb"], rgb0)


def test_outputs_without_rgb():
    """outputs=("depth", "mask"): no colour image, depth and mask and their gradient as with it; outputs=("rgb",): the colour image
    alone, the same bits."""
    from diffdope_amd.render import RasterizeContext, render_texture_batch

    w, ex, kw, mtx0 = _scene("cfg2")
    H, W = w["H"], w["W"]
    outs = []
    ctx = RasterizeContext()  # (one context for all calls: from the second on its depth buffer is not cleared again)
    r = render_texture_batch(ctx, ex(w["proj"]), mtx0, ex(w["pos"]), ex(w["tri"]), [H, W], outputs=("rgb",), **kw)
    assert r["depth"] is None and r["mask"] is None and tuple(r["rgb"].shape) == (mtx0.shape[0], H, W, 3)
    rgb_only = r["rgb"].clone()
    for outputs in (None, ("depth", "mask")):
        mtx = mtx0.clone().requires_grad_(True)
        r = render_texture_batch(ctx, ex(w["proj"]), mtx, ex(w["pos"]), ex(w["tri"]), [H, W], outputs=outputs, **kw)
        assert (r["rgb"] is None) == (outputs is not None)
        if outputs is None:
            assert torch.equal(r["rgb"], rgb_only)
        g = torch.Generator(device="cpu").manual_seed(5)
        wd, wm = (torch.rand(r[k].shape, generator=g).to(mtx.device) for k in ("depth", "mask"))
        (grad,) = torch.autograd.grad((r["depth"] * wd).sum() + (r["mask"] * wm).sum(), mtx)
        outs.append((r["depth"].detach().clone(), r["mask"].detach().contiguous().clone(), grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(outs[0][2], outs[1][2], rtol=2e-4, atol=2e-4 * float(outs[0][2].abs().max()))


@pytest.mark.parametrize("H,W", [(37, 53), (36, 52)])  # (pixel counts not divisible / divisible by 4: one and four pixels per lane)
@pytest.mark.parametrize("masked", [False, True])
def test_masked_l1_of_the_one_channel_view_matches_the_torch_expression(H, W, masked):
    """masked_l1_mean(x.expand(..., 3), y[, m]) takes ddx_masked_l1_bc3_* on the stored channel: values and the gradient with
    respect to the stored channel against torch.mean(torch.abs((x - y) * m)) (diffdope.py:583-613) on the expanded tensor."""
    from diffdope_amd import render
    from diffdope_amd.render import masked_l1_mean

    g = torch.Generator(device="cuda").manual_seed(3)
    B = 5
    base = torch.rand((B, H, W, 1), device="cuda", generator=g, requires_grad=True)
    y = torch.rand((1, H, W, 3), device="cuda", generator=g)
    y = torch.where(y > 0.5, torch.ones_like(y), y)  # (ties x == y do not occur; saturated pixels do)
    m = (torch.rand((1, H, W, 3), device="cuda", generator=g) > 0.6).float() * torch.rand((1, H, W, 3), device="cuda", generator=g) if masked else None
    lr = torch.rand(B, device="cuda", generator=g)
    hits = []
    orig = render._masked_l1_bc3_func.apply
    render._masked_l1_bc3_func.apply = lambda *a: (hits.append(tuple(a[0].shape)), orig(*a))[1]
    try:
        out = masked_l1_mean(base.expand(B, H, W, 3), y.expand(B, H, W, 3), None if m is None else m.expand(B, H, W, 3))
    finally:
        render._masked_l1_bc3_func.apply = orig
    assert hits == [(B, H, W, 1)]
    xe = base.expand(B, H, W, 3)
    ref = torch.mean(torch.abs((xe - y) * (1.0 if m is None else m)), (1, 2, 3))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6)
    (g_ref,) = torch.autograd.grad((ref * lr).mean(), base)
    (g_out,) = torch.autograd.grad((out * lr).mean(), base)
    # (three signed terms per pixel, added in another order than autograd's sum over the expanded axis: they may cancel)
    np.testing.assert_allclose(g_out.cpu().numpy(), g_ref.cpu().numpy(), rtol=2e-6, atol=1e-6 * float(g_ref.abs().max()))
    # a view without a history (detached, or made under no_grad) gives the same value and no gradient path to the stored channel
    out_d = masked_l1_mean(base.expand(B, H, W, 3).detach(), y.expand(B, H, W, 3), None if m is None else m.expand(B, H, W, 3))
    with torch.no_grad():
        out_n = masked_l1_mean(base.expand(B, H, W, 3), y.expand(B, H, W, 3), None if m is None else m.expand(B, H, W, 3))
    assert not out_d.requires_grad and not out_n.requires_grad
    for o in (out_d, out_n):  # (whichever path they take: the stored channel's, or -- torch does not report them as views of it -- the general one)
        np.testing.assert_allclose(o.cpu().numpy(), out.detach().cpu().numpy(), rtol=2e-6)
    n_hits = len(hits)
    # a zero-stride view that is NOT the expand of a contiguous [...,1] base takes the general path, with the same value
    odd = torch.rand((B, H, W, 2), device="cuda", generator=g)[..., :1]
    out2 = masked_l1_mean(odd.expand(B, H, W, 3), y.expand(B, H, W, 3), None if m is None else m.expand(B, H, W, 3))
    ref2 = torch.mean(torch.abs((odd.expand(B, H, W, 3) - y) * (1.0 if m is None else m)), (1, 2, 3))
    np.testing.assert_allclose(out2.cpu().numpy(), ref2.cpu().numpy(), rtol=2e-6)
    assert len(hits) == n_hits == 1  # (the monkeypatch was removed after the first call)


def test_the_api_loop_renders_what_its_losses_read():
    """DiffDope.run_optimization(fused=False) with the depth and mask terms: the loop's renders carry no colour image, the result
    is that of a loop that renders everything (a user loss function makes it: api.DiffDope._loop_outputs), and the complete images
    of the last iteration are in ddope.renders afterwards, as the reference leaves them (diffdope.py:1656-1714)."""
    from diffdope_amd import api
    from tests.scenes import make_scene
    from tests.test_gpu_api import _ddope

    sc = make_scene(16, 20, 60, 80, B=1, dist=1.8)
    B = 4
    seen = []

    def user_mask_loss(ddope):  # (not a built-in: everything is rendered for it, and `mask` is the reference's ordinary tensor)
        m = ddope.renders["mask"]
        seen.append(ddope.renders["rgb"] is not None and m.is_contiguous() and m.view(m.shape[0], -1).shape[1] == m[0].numel())
        return api.l1_mask(ddope)

    a = _ddope(sc, ("depth", "mask"), B)
    assert a._loop_outputs() == ("depth", "mask") and a._builtin_losses_only()
    a.run_optimization(fused=False)
    b = _ddope(sc, ("depth", "mask"), B)
    b.loss_functions = [f if f is not api.l1_mask else user_mask_loss for f in b.loss_functions]
    assert b._loop_outputs() is None
    b.run_optimization(fused=False)
    assert seen and all(seen)
    for d in (a, b):
        assert set(d.renders) >= {"rgb", "depth", "mask"} and tuple(d.renders["rgb"].shape) == (B, 60, 80, 3)
        assert tuple(d.renders["mask"].shape) == (B, 60, 80, 3)
    assert set(a.losses_values) == set(b.losses_values) == {"depth", "mask_selection"}
    for k in a.losses_values:
        np.testing.assert_allclose(a.losses_values[k].numpy(), b.losses_values[k].numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(a.object3d.params_tensor().cpu().numpy(), b.object3d.params_tensor().cpu().numpy(), rtol=0, atol=1e-5)
    assert _ddope(sc, ("rgb", "mask"), B)._loop_outputs() == ("mask", "rgb")
    assert _ddope(sc, ("rgb", "depth", "mask"), B)._loop_outputs() is None
    c = _ddope(sc, ("rgb",), B)
    assert c._loop_outputs() == ("rgb",)
    c.run_optimization(fused=False)  # (no silhouette pass in the loop; all three images afterwards)
    assert all(c.renders[k] is not None for k in ("rgb", "depth", "mask"))


@pytest.mark.parametrize("H,W", [(37, 53), (36, 52)])
def test_batch_weighted_sum_of_masked_l1(H, W):
    """masked_l1_mean(..., batch_weights=bw) -> (v, sum_b v[b] bw[b]) (ddx_masked_l1_fwd_sum / _bwd_sum): v has the bits of the call
    without weights; the sum and the gradient -- through the sum alone, as the built-in losses use it, and through both outputs --
    match the reference's expression (v * learning_rates).mean() * weight (diffdope.py:534-544, :562, :580, :613); for the
    colour-shaped, the depth-shaped (channel-0 mask) and the one-stored-channel operands."""
    from diffdope_amd.render import masked_l1_mean

    g = torch.Generator(device="cuda").manual_seed(9)
    B, weight = 6, 0.7
    lr = torch.rand(B, device="cuda", generator=g) + 0.5
    bw = lr * (weight / B)
    seg = (torch.rand(1, H, W, 3, device="cuda", generator=g) > 0.4).float()
    extra = torch.rand(B, device="cuda", generator=g)
    cases = []
    x = torch.randn((B, H, W, 3), device="cuda", generator=g, requires_grad=True)
    cases.append((x, x, torch.randn((1, H, W, 3), device="cuda", generator=g).expand(B, H, W, 3), seg.expand(B, H, W, 3), False))
    x = torch.randn((B, H, W), device="cuda", generator=g, requires_grad=True)
    cases.append((x, x, torch.randn((1, H, W), device="cuda", generator=g).expand(B, H, W), seg.expand(B, H, W, 3), True))
    base = torch.rand((B, H, W, 1), device="cuda", generator=g, requires_grad=True)
    cases.append((base, base.expand(B, H, W, 3), seg.expand(B, H, W, 3), None, False))
    for leaf, xin, y, m, ch0 in cases:
        v0 = masked_l1_mean(xin, y, m, mask_channel0=ch0)
        v, s = masked_l1_mean(xin, y, m, mask_channel0=ch0, batch_weights=bw)
        assert torch.equal(v0.detach(), v.detach()) and s.dim() == 0
        mk = 1.0 if m is None else (m[..., 0] if ch0 else m)
        ref_v = torch.mean(torch.abs((xin - y) * mk), tuple(range(1, xin.dim())))
        ref_s = (ref_v * lr).mean() * weight
        np.testing.assert_allclose(float(s), float(ref_s), rtol=3e-6)
        (g_ref,) = torch.autograd.grad(ref_s, leaf, retain_graph=True)
        (g_out,) = torch.autograd.grad(s, leaf, retain_graph=True)
        tol = dict(rtol=3e-6, atol=1e-6 * float(g_ref.abs().max()))
        np.testing.assert_allclose(g_out.cpu().numpy(), g_ref.cpu().numpy(), **tol)
        (g_ref2,) = torch.autograd.grad(ref_s + (ref_v * extra).sum(), leaf)
        (g_out2,) = torch.autograd.grad(s + (v * extra).sum(), leaf)
        np.testing.assert_allclose(g_out2.cpu().numpy(), g_ref2.cpu().numpy(), rtol=3e-6, atol=1e-6 * float(g_ref2.abs().max()))


def test_a_kept_context_needs_no_clear_of_its_depth_buffer():
    """ddx_rasterize_fwd_rows_clean: the pass that reads the depth buffer puts back what it finds, so a context's second and later
    calls skip the clear -- the same rast, bit for bit, as a fresh context gives, for a sequence of different poses (each leaves
    its own pixels behind if the restore misses one), through both the row-restricted and the whole-frame emit."""
    import diffdope_amd as dd
    from diffdope_amd import render
    from diffdope_amd.render import RasterizeContext

    w, ex, kw, mtx0 = _scene("cfg2")
    B, H, W = mtx0.shape[0], w["H"], w["W"]
    kept = RasterizeContext()
    g = torch.Generator(device="cuda").manual_seed(2)
    for it in range(6):
        mtx = mtx0.clone()
        mtx[:, :3, 3] += (torch.rand((B, 3), device="cuda", generator=g) - 0.5) * torch.tensor([1.5, 1.5, 3.0], device="cuda")
        clip = dd.xfm_points(ex(w["pos"]).contiguous(), torch.matmul(ex(w["proj"]), mtx))
        emit_all = it % 2 == 1
        a, ra = render._rasterize_rows(kept, clip, w["tri"], [H, W], emit_all=emit_all)
        b, rb = render._rasterize_rows(RasterizeContext(), clip, w["tri"], [H, W], emit_all=True)
        assert kept._zbuf_clean and torch.equal(ra, rb)
        if emit_all:
            assert torch.equal(a, b)
        else:  # (only the rows of the active tiles are defined)
            for h in range(B):
                lo, hi = int(ra[h, 0]), int(ra[h, 1])
                if lo <= hi:
                    assert torch.equal(a[h, lo:hi + 1], b[h, lo:hi + 1])
        assert float(b[..., 3].max()) > 0
