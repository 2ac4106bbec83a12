"""K6: finite-difference checks (float64 oracle) of every analytic backward and of the whole
pose -> loss chain.  CPU only."""
import numpy as np
import pytest

from diffdope_amd import synthetic as syn
from oracle import oracle as orc

H, W = 48, 64


@pytest.fixture(scope="module")
def scene():
    pos, tri, uv = syn.blob_mesh(10, 14, seed=0)
    tex = syn.texture(32, seed=1)
    proj = orc.projection_matrix(**syn.camera_intrinsics(W, H))
    rng = np.random.RandomState(2)
    q, t = syn.random_quat(rng), np.array([0.1, -0.05, -2.0])
    qg, tg = syn.perturb_pose(q, t, 8.0, 0.03, rng)
    R = orc.RenderOracle(pos, tri, proj, H, W, {}, dict(rgb=0.7, depth=1.0, mask=1.0), uv=uv, tex=tex, dtype=np.float64)
    r = R.render(orc.pose_fwd(np.concatenate([qg, tg])[:, None]))
    cov = r["rast"][..., 3] > 0
    assert 0.05 < cov.mean() < 0.5
    R.gt = {"rgb": r["rgb"], "depth": r["depth"], "segmentation": np.repeat(cov[..., None], 3, -1).astype(np.float64)}
    params = np.stack([np.concatenate([q * 1.3, t]), np.concatenate([qg * 0.9, tg + 0.01])], 1)
    return R, params


@pytest.mark.parametrize("weights", [
    dict(rgb=0.7, depth=None, mask=None), dict(rgb=None, depth=1.0, mask=None), dict(rgb=None, depth=None, mask=1.0),
    dict(rgb=0.7, depth=1.0, mask=1.0),
    dict(rgb=None, depth=None, mask=None, edge=1.0), dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.8),  # edge: this build's extension
])
def test_pose_to_loss_chain_matches_finite_differences(scene, weights):
    R, params = scene
    R.weights = weights
    lrm = np.array([0.7, 2.0])
    _, _, g, _ = R.loss_and_grad(params, lrm)
    num = np.zeros_like(g)
    eps = 1e-6
    for i in range(7):
        for b in range(params.shape[1]):
            pp, pm = params.copy(), params.copy()
            pp[i, b] += eps
            pm[i, b] -= eps
            num[i, b] = (R.loss_and_grad(pp, lrm, want_grad=False)[0] - R.loss_and_grad(pm, lrm, want_grad=False)[0]) / (2 * eps)
    assert np.abs(g - num).max() <= 1e-6 * np.abs(num).max() + 1e-10


def test_vertex_color_path_chain(scene):
    R, params = scene
    R2 = orc.RenderOracle(R.pos, R.tri, R.proj, H, W, {}, dict(rgb=1.0, depth=None, mask=None),
                          vtx_color=syn.vertex_colors(R.pos), dtype=np.float64)
    R2.gt = R.gt
    _, _, g, _ = R2.loss_and_grad(params[:, :1])
    eps = 1e-6
    for i in (0, 3, 4, 6):
        pp, pm = params[:, :1].copy(), params[:, :1].copy()
        pp[i] += eps
        pm[i] -= eps
        num = (R2.loss_and_grad(pp, want_grad=False)[0] - R2.loss_and_grad(pm, want_grad=False)[0]) / (2 * eps)
        assert abs(g[i, 0] - num) <= 1e-6 * abs(num) + 1e-10


def test_interpolate_attr_and_texture_texel_gradients():
    rng = np.random.RandomState(3)
    pos, tri, _ = syn.blob_mesh(6, 8, seed=1)
    proj = orc.projection_matrix(**syn.camera_intrinsics(32, 24))
    mtx = orc.pose_fwd(np.array([0.1, 0.2, 0.3, 0.9, 0, 0, -2.0])[:, None])
    pc = orc.xfm_fwd(pos[None].astype(np.float64), proj[None] @ mtx)
    rast = orc.rasterize_fwd(pc, tri, 24, 32)
    attr = rng.normal(size=(1, pos.shape[0], 2))
    G = rng.normal(size=(1, 24, 32, 2))
    dattr, _ = orc.interpolate_bwd(attr, rast, tri, G, True)
    used = np.unique(tri[(rast[0, ..., 3][rast[0, ..., 3] > 0] - 1).astype(int)])
    for v in used[:5]:
        a = attr.copy()
        a[0, v, 1] += 1e-6
        num = ((G * orc.interpolate_fwd(a, rast, tri)).sum() - (G * orc.interpolate_fwd(attr, rast, tri)).sum()) / 1e-6
        assert abs(dattr[0, v, 1] - num) < 1e-6 * max(1, abs(num))
    tex = rng.uniform(size=(1, 4, 5, 3))
    uv = rng.uniform(-1, 2, size=(1, 3, 4, 2))
    Gt = rng.normal(size=(1, 3, 4, 3))
    duv, dtex = orc.texture_bwd(tex, uv, Gt, True)
    t2 = tex.copy()
    t2[0, 2, 3, 1] += 1e-6
    num = ((Gt * orc.texture_fwd(t2, uv)).sum() - (Gt * orc.texture_fwd(tex, uv)).sum()) / 1e-6
    assert abs(dtex[0, 2, 3, 1] - num) < 1e-6
    for c in (0, 1):
        u2 = uv.copy()
        u2[0, 1, 2, c] += 1e-7
        num = ((Gt * orc.texture_fwd(tex, u2)).sum() - (Gt * orc.texture_fwd(tex, uv)).sum()) / 1e-7
        assert abs(duv[0, 1, 2, c] - num) < 1e-4 * max(1, abs(num))


def test_antialias_color_gradient():
    rng = np.random.RandomState(4)
    pos, tri, _ = syn.blob_mesh(6, 8, seed=1)
    proj = orc.projection_matrix(**syn.camera_intrinsics(32, 24))
    mtx = orc.pose_fwd(np.array([0.1, 0.2, 0.3, 0.9, 0, 0, -2.0])[:, None])
    pc = orc.xfm_fwd(pos[None].astype(np.float64), proj[None] @ mtx)
    rast = orc.rasterize_fwd(pc, tri, 24, 32)
    col = rng.uniform(size=(1, 24, 32, 3))
    G = rng.normal(size=col.shape)
    dcol, dpos = orc.antialias_bwd(col, rast, pc, tri, G)
    base = (G * orc.antialias_fwd(col, rast, pc, tri)).sum()
    ys, xs = np.nonzero(np.abs(dcol[0] - G[0]).sum(-1) > 0)
    assert len(ys) > 5
    for k in range(0, len(ys), max(1, len(ys) // 6)):
        c2 = col.copy()
        c2[0, ys[k], xs[k], 1] += 1e-6
        num = ((G * orc.antialias_fwd(c2, rast, pc, tri)).sum() - base) / 1e-6
        assert abs(dcol[0, ys[k], xs[k], 1] - num) < 1e-6 * max(1, abs(num))
    # position gradient (general colours, including covered-vs-covered pairs)
    vs = np.argsort(-np.abs(dpos[0]).sum(-1))[:6]
    checked = 0

    def fd(v, c, eps):
        a, b = pc.copy(), pc.copy()
        a[0, v, c] += eps
        b[0, v, c] -= eps
        # hold rast fixed (its z only selects the nearer surface)
        return ((G * orc.antialias_fwd(col, rast, a, tri)).sum() - (G * orc.antialias_fwd(col, rast, b, tri)).sum()) / (2 * eps)

    for v in vs:
        for c in (0, 1, 3):
            n1, n2 = fd(v, c, 1e-7), fd(v, c, 1e-8)
            if abs(n1 - n2) > 1e-4 * max(1, abs(n1)):
                continue  # a discrete antialias decision flips inside the stencil: not differentiable here
            assert abs(dpos[0, v, c] - n1) < 1e-5 * max(1, abs(n1))
            checked += 1
    assert checked >= 10


def test_edge_loss_image_gradient_matches_finite_differences():
    """Extension (no reference counterpart): d edge / d rgb of orc_loss_edge against central differences."""
    rng = np.random.default_rng(5)
    B, h, w = 2, 7, 9
    rgb, gt = rng.random((B, h, w, 3)), rng.random((1, h, w, 3))
    seg = np.repeat((rng.random((1, h, w, 1)) > 0.4).astype(np.float64), 3, -1)
    sc = np.array([0.6, 1.7])
    per, d = orc.loss_edge(rgb, gt, seg, sc, True)
    eps = 1e-6
    for idx in [(0, 0, 0, 0), (0, 3, 4, 1), (1, 6, 8, 2), (1, 2, 0, 0), (0, 0, 8, 1)]:
        p, m = rgb.copy(), rgb.copy()
        p[idx] += eps
        m[idx] -= eps
        num = ((orc.loss_edge(p, gt, seg)[0] - orc.loss_edge(m, gt, seg)[0]) * sc).sum() / (2 * eps)
        assert abs(d[idx] - num) < 1e-7, (idx, d[idx], num)
