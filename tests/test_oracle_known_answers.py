"""Known-answer tests that define the renderer semantics of this build:
the nvdiffrast-side ops have no importable reference, so the oracle is held to hand-computed
answers.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc


def clip_from_pixels(xy, H, W, z=0.0, w=1.0):
    """Pixel-space (continuous, pixel centre = i+0.5) -> clip coordinates with the given w."""
    xy = np.asarray(xy, np.float64)
    x = (xy[:, 0] / W * 2 - 1) * w
    y = (xy[:, 1] / H * 2 - 1) * w
    return np.stack([x, y, np.full(len(xy), z) * w, np.full(len(xy), w)], axis=1)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_k1_single_triangle_pixel_set_and_barycentrics(dt):
    H, W = 8, 8
    pos = clip_from_pixels([[1, 1], [7, 1], [1, 7]], H, W, z=0.25)[None].astype(dt)
    tri = np.array([[0, 1, 2]], np.int32)
    rast = orc.rasterize_fwd(pos, tri, H, W)
    ids = rast[0, ..., 3]
    # centre (px+.5, py+.5) inside x>=1, y>=1, x+y<=8 ; the hypotenuse passes through centres
    # with px+py == 7 and is NOT owned by this triangle under the ownership rule chosen here
    # (checked against the two-triangle watertightness test below).
    for py in range(H):
        for px in range(W):
            inside = px >= 1 and py >= 1 and (px + py + 1) < 8
            on_edge = px >= 1 and py >= 1 and (px + py + 1) == 8
            if not on_edge:
                assert (ids[py, px] == 1) == inside, (px, py)
    # barycentrics at pixel (2,3): centre (2.5,3.5); u = weight of v0, v = weight of v1
    u, v = rast[0, 3, 2, 0], rast[0, 3, 2, 1]
    # solve: c = u*v0 + v*v1 + (1-u-v)*v2  -> x: 1u+7v+1(1-u-v)=2.5 -> v=0.25 ; y: u+v+7(1-u-v)=3.5 -> u+v=7/12
    assert abs(v - 0.25) < 1e-6 and abs(u - (7 / 12 - 0.25)) < 1e-6
    assert abs(rast[0, 3, 2, 2] - 0.25) < 1e-6
    # background is all-zero
    assert np.all(rast[0, 0, 0] == 0)
    # row 0 is NDC y=-1: the triangle touches y_pix in [1,7] only
    assert ids[0].sum() == 0


def test_k1_perspective_correct_barycentrics():
    H, W = 16, 16
    # a triangle with different w per vertex: u,v must be the perspective-correct weights
    pix = np.array([[2.0, 2.0], [14.0, 3.0], [4.0, 13.0]])
    ws = np.array([1.0, 2.0, 4.0])
    pos = np.concatenate([clip_from_pixels(pix[i : i + 1], H, W, z=0.1, w=ws[i]) for i in range(3)])[None]
    rast = orc.rasterize_fwd(pos, np.array([[0, 1, 2]], np.int32), H, W)
    py, px = 6, 6
    assert rast[0, py, px, 3] == 1
    # screen-space barycentrics
    c = np.array([px + 0.5, py + 0.5])
    A = np.array([[pix[0, 0] - pix[2, 0], pix[1, 0] - pix[2, 0]], [pix[0, 1] - pix[2, 1], pix[1, 1] - pix[2, 1]]])
    l0, l1 = np.linalg.solve(A, c - pix[2])
    lam = np.array([l0, l1, 1 - l0 - l1])
    pc = lam / ws
    pc /= pc.sum()
    assert abs(rast[0, py, px, 0] - pc[0]) < 1e-9 and abs(rast[0, py, px, 1] - pc[1]) < 1e-9


def test_k2_depth_order_and_watertight_shared_edge():
    H, W = 12, 12
    # two triangles forming a quad that covers pixels [2,10)^2, shared diagonal through pixel centres
    quad = clip_from_pixels([[2, 2], [10, 2], [10, 10], [2, 10]], H, W, z=0.5)
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    rast = orc.rasterize_fwd(quad[None], tri, H, W)
    ids = rast[0, ..., 3]
    inside = np.zeros((H, W), bool)
    inside[2:10, 2:10] = True
    assert np.all((ids > 0) == inside)  # every pixel owned exactly once, none missing on the diagonal
    assert set(np.unique(ids[inside])) == {1.0, 2.0}
    # same quad with opposite winding for the second triangle: still watertight (no culling)
    tri2 = np.array([[0, 1, 2], [0, 3, 2]], np.int32)
    ids2 = orc.rasterize_fwd(quad[None], tri2, H, W)[0, ..., 3]
    assert np.all((ids2 > 0) == inside)
    # a nearer triangle wins regardless of index order; equal depth -> lower index
    near = clip_from_pixels([[4, 4], [8, 4], [4, 8]], H, W, z=-0.5)
    pos = np.concatenate([quad, near])[None]
    tri3 = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6]], np.int32)
    ids3 = orc.rasterize_fwd(pos, tri3, H, W)[0, ..., 3]
    assert ids3[5, 5] == 3
    dup = np.array([[0, 1, 2], [0, 1, 2]], np.int32)
    ids4 = orc.rasterize_fwd(quad[None], dup, H, W)[0, ..., 3]
    assert set(np.unique(ids4)) == {0.0, 1.0}
    # clipping: fragments beyond the far plane disappear, w<=0 triangles are dropped
    far = clip_from_pixels([[2, 2], [10, 2], [2, 10]], H, W, z=1.5)
    assert orc.rasterize_fwd(far[None], np.array([[0, 1, 2]], np.int32), H, W)[0, ..., 3].sum() == 0
    neg = quad.copy()
    neg[0, 3] = -1.0
    assert orc.rasterize_fwd(neg[None], np.array([[0, 1, 2]], np.int32), H, W)[0, ..., 3].sum() == 0


def test_k3_interpolate_constant_and_linear_fields_exact():
    H, W = 16, 16
    pix = np.array([[1.0, 1.0], [15.0, 2.0], [3.0, 15.0]])
    pos = clip_from_pixels(pix, H, W, z=0.0)[None]
    tri = np.array([[0, 1, 2]], np.int32)
    rast = orc.rasterize_fwd(pos, tri, H, W)
    const = np.full((1, 3, 2), 3.25)
    out = orc.interpolate_fwd(const, rast, tri)
    cov = rast[0, ..., 3] > 0
    np.testing.assert_allclose(out[0][cov], 3.25, rtol=0, atol=1e-12)
    assert np.all(out[0][~cov] == 0)
    # attribute = pixel-space position (affine, w=1) -> reproduces the pixel centre
    lin = pix[None]
    out = orc.interpolate_fwd(lin, rast, tri)
    ys, xs = np.nonzero(cov)
    np.testing.assert_allclose(out[0][cov], np.stack([xs + 0.5, ys + 0.5], 1), atol=1e-9)
    # the "mask" call: ones [1,T,3] indexed as per-vertex (diffdope.py:212) -> 1 on covered
    ones = np.ones((1, 5, 3))
    m = orc.interpolate_fwd(ones, rast, tri)
    assert np.all(np.abs(m[0][cov] - 1) < 1e-12) and np.all(m[0][~cov] == 0)  # u+v+(1-u-v) rounds


def test_k4_bilinear_texture_wrap():
    tex = np.array([[[0.0], [1.0]], [[2.0], [3.0]]])[None]  # [1,2,2,1]; tex[y,x]
    # texel centres are at (x+.5)/Tw
    uv = np.array([[[[0.25, 0.25], [0.75, 0.25], [0.25, 0.75], [0.75, 0.75], [0.5, 0.5], [0.0, 0.25], [1.25, -0.25]]]])
    out = orc.texture_fwd(tex, uv)[0, 0, :, 0]
    np.testing.assert_allclose(out[:4], [0, 1, 2, 3], atol=1e-12)
    assert abs(out[4] - 1.5) < 1e-12
    assert abs(out[5] - 0.5) < 1e-12  # u=0 sits between texel 1 (wrapped) and texel 0
    # uv=(1.25,-0.25) wraps to (0.25,0.75)
    assert abs(out[6] - 2.0) < 1e-12
    # 4x4 gradient texture: interior sample equals the analytic bilinear value
    t4 = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    uvp = np.array([[[[0.4, 0.6]]]])
    x, y = 0.4 * 4 - 0.5, 0.6 * 4 - 0.5
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    fx, fy = x - x0, y - y0
    ref = (t4[0, y0, x0, 0] * (1 - fx) + t4[0, y0, x0 + 1, 0] * fx) * (1 - fy) + \
          (t4[0, y0 + 1, x0, 0] * (1 - fx) + t4[0, y0 + 1, x0 + 1, 0] * fx) * fy
    assert abs(orc.texture_fwd(t4, uvp)[0, 0, 0, 0] - ref) < 1e-12


def _halfplane_mesh(edge_x, H, W):
    """A big quad covering x < edge_x (pixel units) over the full height and beyond."""
    pix = np.array([[-4.0, -4.0], [edge_x, -4.0], [edge_x, H + 4.0], [-4.0, H + 4.0]])
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return pix, tri


@pytest.mark.parametrize("edge_x,expect_l,expect_r", [(5.25, 0.75, 0.0), (5.75, 1.0, 0.25), (5.5, 1.0, 0.0)])
def test_k5_antialias_vertical_edge_coverage(edge_x, expect_l, expect_r):
    """Pixel 5 (centre 5.5) and 6: a vertical silhouette at edge_x leaves coverage =
    analytic area fraction in the boundary pixel pair."""
    H, W = 8, 12
    pix, tri = _halfplane_mesh(edge_x, H, W)
    pos = clip_from_pixels(pix, H, W, z=0.0)[None]
    rast = orc.rasterize_fwd(pos, tri, H, W)
    cov = orc.interpolate_fwd(np.ones((1, 4, 1)), rast, tri)
    out = orc.antialias_fwd(cov, rast, pos, tri)
    row = out[0, 4, :, 0]
    # box-filter coverage of a vertical edge: pixel p gets clamp(edge_x - p, 0, 1)
    for p in range(W):
        assert abs(row[p] - min(max(edge_x - p, 0.0), 1.0)) < 1e-9, (p, row)
    assert abs(row[5] - expect_l) < 1e-9 or abs(row[5] - (edge_x - 5)) < 1e-9
    assert abs(row[6] - expect_r) < 1e-9 or row[6] == 0


def test_k5_antialias_leaves_interior_and_nonsilhouette_edges_alone():
    H, W = 12, 12
    quad = clip_from_pixels([[2.3, 2.2], [9.6, 2.4], [9.7, 9.8], [2.1, 9.5]], H, W, z=0.0)
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    rast = orc.rasterize_fwd(quad[None], tri, H, W)
    col = np.random.RandomState(0).uniform(size=(1, H, W, 3))
    out = orc.antialias_fwd(col, rast, quad[None], tri)
    ids = rast[0, ..., 3]
    # pixels whose 4-neighbourhood is entirely inside the quad keep their colour although the
    # internal diagonal separates triangle ids 1 and 2 there
    interior = np.zeros((H, W), bool)
    for y in range(1, H - 1):
        for x in range(1, W - 1):
            interior[y, x] = ids[y, x] > 0 and ids[y - 1, x] > 0 and ids[y + 1, x] > 0 and ids[y, x - 1] > 0 and ids[y, x + 1] > 0
    assert interior.sum() > 10
    np.testing.assert_array_equal(out[0][interior], col[0][interior])
    # some silhouette pixel did change
    assert np.abs(out - col).max() > 1e-3


def test_opposite_vertex_topology():
    tri = np.array([[0, 1, 2], [2, 1, 3], [4, 5, 6]], np.int32)
    opp = orc.build_opposite(tri)
    # edge k joins vertices (k+1)%3,(k+2)%3: tri0 edge0 = (1,2) shared with tri1 whose opposite vertex is 3
    assert opp[0, 0] == 3 and opp[0, 1] == -1 and opp[0, 2] == -1
    assert opp[1, 2] == 0  # tri1 edge2 = (2,1)
    assert np.all(opp[2] == -1)


def test_edge_extension_known_answers():
    """Edge term (this build's extension, no reference counterpart): Sobel/8 of the luminance, zero padding.
    A vertical unit step gives |Gx| = 1/2 on the two columns next to the step (3/8 on the border rows), Gy = 0
    away from the border rows; the loss equals an independent numpy restatement."""
    h, w = 6, 8
    img = np.zeros((1, h, w, 3))
    img[:, :, 4:, :] = 1.0
    zero = np.zeros((1, h, w, 3))
    ones = np.ones((1, h, w, 3))
    per, _ = orc.loss_edge(img, zero, ones)

    def sob(l):
        p = np.pad(l, 1)
        gx = ((p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])) / 8
        gy = ((p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])) / 8
        return gx, gy

    gx, gy = sob(img[0].mean(-1))
    assert np.allclose(gx[1:-1, 3], 0.5) and np.allclose(gx[1:-1, 4], 0.5) and np.allclose(gx[0, 3], 0.375)
    assert np.allclose(gy[1:-1, :7], 0.0)
    assert np.isclose(per[0], (np.abs(gx) + np.abs(gy)).sum() / (2 * h * w))
    rng = np.random.default_rng(3)
    a, g = rng.random((2, h, w, 3)), rng.random((1, h, w, 3))
    seg = np.repeat((rng.random((1, h, w, 1)) > 0.5).astype(np.float64), 3, -1)
    per, _ = orc.loss_edge(a, g, seg)
    gx0, gy0 = sob((g * seg)[0].mean(-1))
    for b in range(2):
        gx, gy = sob(a[b].mean(-1))
        assert np.isclose(per[b], (np.abs(gx - gx0) + np.abs(gy - gy0)).sum() / (2 * h * w), rtol=1e-12)
    # identical images: zero loss, zero gradient
    per, d = orc.loss_edge(g * seg, g, seg, None, True)
    assert per[0] == 0 and np.all(d == 0)
