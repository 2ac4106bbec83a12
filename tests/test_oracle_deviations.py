"""The four DECLARED deviations of this build from nvdiffrast at the unpinned boundary (nvdiffrast is
un-vendored and un-pinned).  Each test states, for a hand-built case, what nvdiffrast's published algorithm
yields and what this build yields, so the size and the trigger of every deviation is on record.  CPU only (the HIP
kernels are held to this oracle bit-for-bit / to tolerance by the -m gpu tests).

  D1  (closed in round 2) a triangle with a vertex at clip w <= 0 was dropped; it is now clipped at the near plane as GL does;
      (round 3) its fully visible silhouette edges are antialiased, the edges that run through the eye plane are not
  D2  rasterize backward: a barycentric saturated by the [0,1] clamp passes no gradient, nvdiffrast differentiates
      the unclamped expression
  D3  antialias: of two triangle edges that cross the pixel-pair segment the one with the LARGER crossing parameter is
      used (explicit max), ties resolved to the lower edge index
  D4  coverage of a pixel centre lying exactly on an (unshared) triangle edge: own ownership rule on 1/256-pixel snapped
      coordinates; GL hardware applies the top-left rule of its own fixed-point grid
"""
import numpy as np

from oracle import oracle as orc
from tests.test_oracle_known_answers import clip_from_pixels


def _homogeneous_inside(P, px, py, H, W):
    """Reference semantics of clipped rasterisation without clipping (Olano & Greer homogeneous 2-D rasterisation, float64):
    pixel centre inside the triangle's visible part <=> the three homogeneous edge functions share the sign of their sum
    and the interpolated w is positive and -w <= z <= w."""
    fx, fy = (px + 0.5) / W * 2 - 1, (py + 0.5) / H * 2 - 1
    q = [(p[0] - fx * p[3], p[1] - fy * p[3]) for p in P]
    a = [q[1][0] * q[2][1] - q[1][1] * q[2][0], q[2][0] * q[0][1] - q[2][1] * q[0][0], q[0][0] * q[1][1] - q[0][1] * q[1][0]]
    s = sum(a)
    if s == 0 or not all(ai * s >= 0 for ai in a):
        return False
    w = sum(ai * p[3] for ai, p in zip(a, P)) / s
    z = sum(ai * p[2] for ai, p in zip(a, P)) / s
    return w > 0 and -w <= z <= w


def test_d1_triangle_straddling_the_camera_plane_is_clipped_at_the_near_plane():
    """Round 1 dropped any triangle with a vertex at w <= 0; since round 2 it is clipped against the near plane like
    dr.rasterize does: the drawn pixels are exactly those of homogeneous rasterisation (no clipping needed there: Olano & Greer),
    and (u, v, z/w) are the perspective-correct values of the ORIGINAL triangle."""
    from diffdope_amd import synthetic as syn

    H, W = 16, 16
    proj = orc.projection_matrix(**syn.camera_intrinsics(W, H))
    cam = np.array([[-0.6, -0.4, -1.0, 1.0], [-0.6, 0.4, -1.0, 1.0], [0.5, 0.0, 0.3, 1.0]])  # the third vertex is BEHIND the camera
    P = (proj @ cam.T).T
    assert P[2, 3] < 0 < P[0, 3]
    tri = np.array([[0, 1, 2]], np.int32)
    want = np.array([[_homogeneous_inside(P, x, y, H, W) for x in range(W)] for y in range(H)])
    for dt in (np.float64, np.float32):
        rast = orc.rasterize_fwd(P[None].astype(dt), tri, H, W)
        got = rast[0, ..., 3] > 0
        assert want.sum() > 100 and np.array_equal(got, want)
        for y, x in ((0, 1), (8, 1), (15, 15), (3, 9)):
            u, v, zw = (float(c) for c in rast[0, y, x, :3])
            c = u * P[0] + v * P[1] + (1 - u - v) * P[2]  # the point of the original triangle's plane these weights name
            fx, fy = (x + 0.5) / W * 2 - 1, (y + 0.5) / H * 2 - 1
            tol = 1e-9 if dt is np.float64 else 2e-5
            assert abs(c[0] / c[3] - fx) < tol and abs(c[1] / c[3] - fy) < tol and abs(c[2] / c[3] - zw) < tol
    # all three vertices behind the near plane: nothing; the same triangle reversed: the same pixels (no culling at this level)
    behind = (proj @ np.array([[-0.6, -0.4, 0.2, 1.0], [-0.6, 0.4, 0.2, 1.0], [0.5, 0.0, 0.3, 1.0]]).T).T
    assert not orc.rasterize_fwd(behind[None], tri, H, W)[..., 3].any()
    assert np.array_equal(orc.rasterize_fwd(P[None], tri[:, [0, 2, 1]], H, W)[0, ..., 3] > 0, want)
    # two straddlers sharing an edge whose ends are on opposite sides of the near plane: the clipped polygons meet without
    # gap or overlap (the intersection is computed from the inside end of the edge in both)
    cam4 = np.array([[-0.6, -0.4, -1.0, 1.0], [-0.6, 0.4, -1.0, 1.0], [0.5, 0.0, 0.3, 1.0], [0.45, 0.9, 0.3, 1.0]])
    P4 = (proj @ cam4.T).T
    r2 = orc.rasterize_fwd(P4[None], np.array([[0, 1, 2], [1, 3, 2]], np.int32), H, W)[0, ..., 3]
    a = np.array([[_homogeneous_inside(P4[[0, 1, 2]], x, y, H, W) for x in range(W)] for y in range(H)])
    bq = np.array([[_homogeneous_inside(P4[[1, 3, 2]], x, y, H, W) for x in range(W)] for y in range(H)])
    assert np.array_equal(r2 > 0, a | bq) and (r2[a & ~bq] == 1).all() and (r2[bq & ~a] == 2).all()


def test_d1_pose_gradient_when_part_of_a_mesh_crosses_the_camera_plane():
    """On a whole mesh: a hypothesis so close that the camera plane cuts the object; the straddling triangles are clipped, loss
    and pose gradient stay finite, and the analytic gradient is the derivative of what is drawn (finite differences, float64)."""
    from diffdope_amd import synthetic as syn

    pos, tri, uv = syn.blob_mesh(6, 8, seed=0)
    H, W = 24, 32
    proj = orc.projection_matrix(**syn.camera_intrinsics(W, H))
    R = orc.RenderOracle(pos, tri, proj, H, W, {}, dict(rgb=None, depth=1.0, mask=1.0), vtx_color=syn.vertex_colors(pos, seed=5), dtype=np.float64)
    p_far = np.array([[0.1], [0.2], [0.05], [0.97], [0.0], [0.0], [-1.6]])
    r = R.render(orc.pose_fwd(p_far))
    cov = r["rast"][0, ..., 3] > 0
    R.gt = {"depth": r["depth"].copy(), "segmentation": np.repeat(cov[None, ..., None], 3, -1).astype(np.float64)}
    p_cut = p_far.copy()
    p_cut[6, 0] = -0.25  # the camera plane now cuts the object: some vertices have w = -z_cam <= 0
    clip = R.render(orc.pose_fwd(p_cut))["pos_clip"][0]
    n_behind = int((clip[:, 3] <= 0).sum())
    straddlers = sum(1 for t in tri if 0 < sum(clip[v, 3] <= 0 for v in t) < 3)
    assert n_behind > 0 and straddlers > 0
    total, logs, g, _ = R.loss_and_grad(p_cut, np.ones(1))
    assert np.isfinite(total) and np.all(np.isfinite(g))
    eps = 1e-6
    for i in (4, 6):
        pp, pm = p_cut.copy(), p_cut.copy()
        pp[i, 0] += eps
        pm[i, 0] -= eps
        fd = (R.loss_and_grad(pp, np.ones(1), want_grad=False)[0] - R.loss_and_grad(pm, np.ones(1), want_grad=False)[0]) / (2 * eps)
        assert abs(fd - g[i, 0]) < 1e-4 * max(1.0, abs(fd)), (i, fd, g[i, 0])


def _straddling_quad():
    """Two triangles sharing the edge (0, 2); vertex 2 lies BEHIND the eye plane (w < 0).  Generic coordinates: no edge passes
    through a pixel centre."""
    pos = np.array([[[-0.6137, -0.5219, 0.2, 1.013], [0.5531, -0.4473, 0.1, 0.917], [0.2071, 0.9113, -0.3, -0.4219], [-0.9043, 0.6171, 0.3, 1.2331]]])
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    return pos, tri


def test_d1_antialias_on_the_visible_edges_of_a_triangle_cut_by_the_camera_plane():
    """Round 3 (what was left of D1): the pair analysis used to return early for a triangle with a vertex at w <= 0, so a
    hypothesis cut by the camera plane got no silhouette blend / gradient on those triangles.  Now the orientation tests are done in
    homogeneous form (sign of det[x y w]) and every edge with BOTH endpoints in front of the eye plane is analysed as usual; an
    edge that itself crosses the eye plane, and the cut, are not antialiased.  Known answers: blended pixels exist along the two
    fully visible silhouette edges (0,1) and (3,0), none along the cut edges; the vertex behind the plane gets no gradient; the
    gradient is the derivative (float64 finite differences, rast held fixed)."""
    H, W = 48, 64
    pos, tri = _straddling_quad()
    rast = orc.rasterize_fwd(pos, tri, H, W)
    ids = rast[0, ..., 3]
    assert (ids == 1).sum() > 500 and (ids == 2).sum() > 20  # both clipped triangles are drawn
    col = (ids > 0).astype(np.float64)[None, ..., None].repeat(3, -1)
    out = orc.antialias_fwd(col, rast, pos, tri)
    changed = np.abs(out - col)[0, ..., 0] > 1e-9
    assert changed.sum() >= 20
    # every blended pixel lies within a pixel of the projected edge (0,1) or (3,0)
    px = lambda v: ((pos[0, v, 0] / pos[0, v, 3] * 0.5 + 0.5) * W, (pos[0, v, 1] / pos[0, v, 3] * 0.5 + 0.5) * H)
    def dist_to_segment(x, y, a, b):
        a, b, q = np.array(px(a)), np.array(px(b)), np.array([x + 0.5, y + 0.5])
        t = np.clip(np.dot(q - a, b - a) / np.dot(b - a, b - a), 0, 1)
        return np.linalg.norm(q - (a + t * (b - a)))
    for y, x in np.argwhere(changed):
        assert min(dist_to_segment(x, y, 0, 1), dist_to_segment(x, y, 3, 0)) < 1.5
    rng = np.random.RandomState(0)
    G = rng.normal(size=out.shape)
    _, dpos = orc.antialias_bwd(col, rast, pos, tri, G)
    assert np.all(dpos[0, 2] == 0) and np.abs(dpos[0, [0, 1, 3]]).max() > 1
    f = lambda q: float((orc.antialias_fwd(col, rast, q, tri) * G).sum())
    checked = 0
    for v in (0, 1, 3):
        for c in (0, 1, 3):
            def fd(eps):
                a, b = pos.copy(), pos.copy()
                a[0, v, c] += eps
                b[0, v, c] -= eps
                return (f(a) - f(b)) / (2 * eps)
            n1, n2 = fd(1e-7), fd(1e-8)
            if abs(n1 - n2) > 1e-4 * max(1, abs(n1)):
                continue
            assert abs(dpos[0, v, c] - n1) < 1e-5 * max(1, abs(n1))
            checked += 1
    assert checked >= 8
    # float32 and float64 take the same discrete decisions on this case
    o32 = orc.antialias_fwd(col.astype(np.float32), rast.astype(np.float32), pos.astype(np.float32), tri)
    assert np.abs(o32 - out).max() < 1e-5


def test_d2_saturated_barycentric_passes_no_gradient():
    H, W = 8, 8
    # The right edge of the triangle runs 0.4/256 pixel LEFT of the centre of pixel (4,3).  The fixed-point rule snaps the
    # edge onto the centre and owns it (upward edge), so the pixel is covered; the float barycentric of the vertex opposite
    # that edge (vertex 0 here, so it is `u`) comes out slightly negative and the forward clamps it to 0, as nvdiffrast's
    # output is clamped.
    xe = 4.5 - 0.4 / 256
    pix = np.array([[0.5, 4.0], [xe, 0.5], [xe, 7.5]])
    P = clip_from_pixels(pix, H, W, z=0.0)[None]
    tri = np.array([[0, 1, 2]], np.int32)
    rast = orc.rasterize_fwd(P, tri, H, W)
    assert rast[0, 3, 4, 3] == 1 and rast[0, 3, 4, 0] == 0.0  # covered, u saturated at 0
    u_unclamped = (xe - 4.5) / (xe - 0.5)  # weight of vertex 0 at the centre: distance past the edge / triangle width
    assert -2e-3 < u_unclamped < 0
    # backward of d loss / d u = 1 at that pixel: this build passes nothing through the saturated component ...
    d = np.zeros_like(rast)
    d[0, 3, 4, 0] = 1.0
    g_sat = orc.rasterize_bwd(P, tri, rast, d)
    assert np.all(g_sat == 0)
    # ... nvdiffrast differentiates the unclamped expression: the same 1/width-sized gradient an interior pixel carries
    d2 = np.zeros_like(rast)
    d2[0, 3, 3, 0] = 1.0  # its left neighbour, clear interior (u = 0.25)
    g_int = orc.rasterize_bwd(P, tri, rast, d2)
    assert 0 < rast[0, 3, 3, 0] < 1 and np.abs(g_int[0, :, 0]).max() > 0.2  # d u / d x_clip ~ (W/2) / (width in pixels) = 4 / 4
    # the compat switch (ddx.h DDX_COMPAT_UNCLAMPED_BARY_GRAD / RefineEngine(compat="nvdiffrast") / oracle.set_compat): the saturated
    # pixel then carries the derivative of the unclamped u -- the same formula as an interior pixel -- and nothing else changes
    old = orc.set_compat(True)
    try:
        g_nv = orc.rasterize_bwd(P, tri, rast, d)
        assert np.array_equal(orc.rasterize_bwd(P, tri, rast, d2), g_int)
    finally:
        orc.set_compat(old)
    Pd = P.astype(np.float64)
    def u_at(Pq):  # unclamped barycentric of vertex 0 at the centre of pixel (4,3): signed areas in clip space (w = 1)
        x, y = Pq[0, :, 0], Pq[0, :, 1]
        fx, fy = (4 + 0.5) / W * 2 - 1, (3 + 0.5) / H * 2 - 1
        a = lambda i, j: (x[i] - fx) * (y[j] - fy) - (y[i] - fy) * (x[j] - fx)
        return a(1, 2) / (a(1, 2) + a(2, 0) + a(0, 1))
    for v in range(3):
        for c in range(2):
            Pp, Pm = Pd.copy(), Pd.copy()
            Pp[0, v, c] += 1e-6
            Pm[0, v, c] -= 1e-6
            assert abs((u_at(Pp) - u_at(Pm)) / 2e-6 - g_nv[0, v, c]) < 1e-4 * max(1.0, np.abs(g_nv).max())
    assert np.abs(g_nv).max() > 0.2
    # the deviation is confined to saturated pixels: the analytic gradient at the interior pixel is the true derivative
    eps = 1e-6
    Pp, Pm = P.copy(), P.copy()
    Pp[0, 0, 0] += eps
    Pm[0, 0, 0] -= eps
    fd = (orc.rasterize_fwd(Pp, tri, H, W)[0, 3, 3, 0] - orc.rasterize_fwd(Pm, tri, H, W)[0, 3, 3, 0]) / (2 * eps)
    assert abs(fd - g_int[0, 0, 0]) < 1e-6 * max(1.0, abs(fd))


def test_d3_antialias_edge_choice_by_largest_crossing_parameter():
    """A sliver narrower than a pixel covers the centre of pixel (5,4) and not that of (6,4): BOTH of its steep edges cross
    the row of centres inside the (-eps, 1+eps) window around the pair segment -- the entry edge 0.002 px left of centre 5,
    the exit edge 0.372 px right of it.  This build takes the crossing with the larger parameter along the direction
    covered -> uncovered (explicit max), i.e. the exit edge, so pixel 5 ends with the analytic box coverage of that edge."""
    H, W = 8, 12
    pix = np.array([[5.47, -10.0], [5.9, -10.0], [5.685, 100.0]])
    tri = np.array([[0, 1, 2]], np.int32)
    P = clip_from_pixels(pix, H, W, z=0.0)[None]
    rast = orc.rasterize_fwd(P, tri, H, W)
    assert rast[0, 4, 5, 3] == 1 and rast[0, 4, 6, 3] == 0 and rast[0, 4, 4, 3] == 0
    cov = orc.interpolate_fwd(np.ones((1, 3, 1)), rast, tri)
    out = orc.antialias_fwd(cov, rast, P, tri)
    x_exit = 5.9 - (4.5 + 10.0) / 110.0 * (5.9 - 5.685)
    x_entry = 5.47 + (4.5 + 10.0) / 110.0 * (5.685 - 5.47)
    assert -0.0625 < x_entry - 5.5 < 0 < x_exit - 5.5 < 1  # both crossings inside the acceptance window of the (5,6) pair
    # pair (5,6): exit edge chosen -> pixel 5 keeps 1 - (0.5 - 0.372) = 0.872 (choosing the entry edge instead, crossing
    # parameter -0.002 clamped to 0, would have blended half of pixel 5 away); pair (4,5) then takes the 0.498 that lies left of
    # the entry edge: pixel 5 ends with exactly the sliver's analytic width inside it
    np.testing.assert_allclose(out[0, 4, 5, 0], x_exit - x_entry, atol=1e-6)
    assert out[0, 4, 6, 0] == 0.0 and out[0, 4, 4, 0] == 0.0
    # a plain vertical silhouette for scale (K5): the pixel pair straddling an edge at x = 5.8 leaves 0.8 in pixel 5
    pix_q = np.array([[-4.0, -4.0], [5.8, -4.0], [5.8, H + 4.0], [-4.0, H + 4.0]])
    tri_q = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    Pq = clip_from_pixels(pix_q, H, W, z=0.0)[None]
    rq = orc.rasterize_fwd(Pq, tri_q, H, W)
    oq = orc.antialias_fwd(orc.interpolate_fwd(np.ones((1, 4, 1)), rq, tri_q), rq, Pq, tri_q)
    np.testing.assert_allclose(oq[0, 4, 5:8, 0], [0.8, 0.0, 0.0], atol=1e-9)


def test_d4_pixel_centre_exactly_on_an_unshared_edge():
    """Ownership of pixel centres lying exactly on triangle edges (after the 1/256-pixel snap).  Own rule (oracle edge_inside):
    with the triangle oriented counter-clockwise in bottom-up window coordinates a centre on an edge is covered iff the edge
    runs upward (dy > 0: the triangle's right side) or is horizontal running leftward (its top side).  GL / D3D hardware use
    the top-LEFT rule in top-down window coordinates.  The two agree on every centre not exactly on an edge."""
    H, W = 8, 8
    # axis-aligned right triangle with vertices on pixel centres: legs on column x = 1.5 and row y = 1.5, hypotenuse x + y = 8
    pix = np.array([[1.5, 1.5], [6.5, 1.5], [1.5, 6.5]])
    rast = orc.rasterize_fwd(clip_from_pixels(pix, H, W)[None], np.array([[0, 1, 2]], np.int32), H, W)
    ids = rast[0, ..., 3] > 0
    own = {"bottom_row": bool(ids[1, 3]), "left_col": bool(ids[3, 1]), "hypotenuse": bool(ids[3, 4]), "interior": bool(ids[2, 2])}
    assert own["interior"]
    # own rule: bottom edge runs in +x (dy = 0, dx > 0) -> not owned; left edge runs downward (dy < 0) -> not owned;
    # hypotenuse runs up-left (dy > 0) -> owned
    assert own == {"bottom_row": False, "left_col": False, "hypotenuse": True, "interior": True}
    # top-left rule in top-down coordinates (rows flipped): "top" = this bottom row, "left" = the left column -> it would own
    # exactly the two legs and not the hypotenuse: the complement on these measure-zero centres
    gl = {"bottom_row": True, "left_col": True, "hypotenuse": False, "interior": True}
    assert all(own[k] != gl[k] for k in ("bottom_row", "left_col", "hypotenuse"))
    # watertightness, the property both rules exist for: adding the mirrored triangle covers every centre of the square once
    quad = np.array([[1.5, 1.5], [6.5, 1.5], [1.5, 6.5], [6.5, 6.5]])
    r2 = orc.rasterize_fwd(clip_from_pixels(quad, H, W)[None], np.array([[0, 1, 2], [1, 3, 2]], np.int32), H, W)
    ids2 = r2[0, ..., 3]
    assert ids2[3, 4] in (1, 2) and (ids2[2:6, 2:6] > 0).all()
    # and a centre NOT on an edge is decided identically by any rule: nudge the triangle by 1/64 pixel
    r3 = orc.rasterize_fwd(clip_from_pixels(pix + 1 / 64, H, W)[None], np.array([[0, 1, 2]], np.int32), H, W)
    assert not r3[0, 1, 3, 3] and not r3[0, 3, 1, 3] and r3[0, 2, 2, 3] and r3[0, 3, 3, 3]


def test_d5_backface_culling_only_on_closed_meshes_and_invisible_there():
    """D5 (fused engine only): on a CLOSED, consistently oriented mesh, hypotheses that lie entirely inside the view volume skip
    their back-facing triangles.  nvdiffrast draws both faces; on a closed surface every pixel centre is covered by as many
    front as back faces and the nearest is a front face, so the drawn image is the same: zero pixels change owner over a
    sweep of poses here.  The rule switches itself off for open / inconsistently oriented meshes, non-pinhole projections and
    for any hypothesis whose object-space bounding box has a corner outside the view volume (w <= 0 or |z| > w; inside at the 8
    corners means inside at every vertex: the drawn surface is whole)."""
    from diffdope_amd import synthetic as syn

    pos, tri, uv = syn.blob_mesh(12, 16, seed=0)
    H, W = 48, 64
    proj = orc.projection_matrix(**syn.camera_intrinsics(W, H))
    assert orc.mesh_cull_sign(pos, tri, proj) == -1  # outward-oriented mesh under the y-up pinhole: back faces snap to negative area
    assert orc.mesh_cull_sign(pos, tri[:, [0, 2, 1]], proj) == 1  # the same surface oriented inward
    assert orc.mesh_cull_sign(pos, tri[:-3], proj) == 0  # a hole
    flipped = tri.copy()
    flipped[37] = flipped[37][[0, 2, 1]]
    assert orc.mesh_cull_sign(pos, flipped, proj) == 0  # one triangle wound the other way
    # two shells: fine when both are outward, refused when the smaller one is inside-out (its visible faces would go)
    two_pos = np.concatenate([pos, pos * 0.5 + np.float32(2.0)]).astype(np.float32)
    small = tri + len(pos)
    assert orc.mesh_cull_sign(two_pos, np.concatenate([tri, small]), proj) == -1
    assert orc.mesh_cull_sign(two_pos, np.concatenate([tri, small[:, [0, 2, 1]]]), proj) == 0
    # a flat two-sided patch (the same triangles wound both ways) has every edge shared by two opposite triangles, but its
    # volume is round-off of either sign: not a solid, never culled
    flat = np.concatenate([tri[:40], tri[:40][:, [0, 2, 1]]])
    assert orc.mesh_cull_sign(pos, flat, proj) == 0
    assert orc.mesh_cull_sign(two_pos, np.concatenate([tri, flat + len(pos)]), proj) == 0  # (also next to a real solid)
    bad_proj = proj.copy()
    bad_proj[3, 3] = 1.0  # orthographic-style w row: not the pinhole the sign argument needs
    assert orc.mesh_cull_sign(pos, tri, bad_proj) == 0
    rng = np.random.RandomState(3)
    changed = 0
    for _ in range(24):
        q = syn.random_quat(rng)
        t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), -rng.uniform(1.2, 3.0)])
        mtx = orc.pose_fwd(np.concatenate([q, t])[:, None].astype(np.float32))
        final = np.matmul(proj[None].astype(np.float32), mtx)
        clip = orc.xfm_fwd(pos[None], final, True)
        cull = orc.view_volume_cull(pos, final, -1)
        assert cull.tolist() == [-1]  # the whole bounding box is in front of the near plane: the rule is on
        a = orc.rasterize_fwd(clip, tri, H, W)
        b = orc.rasterize_fwd(clip, tri, H, W, cull)
        assert (a[..., 3] > 0).sum() > 50
        changed += int((a[..., 3] != b[..., 3]).sum())
        np.testing.assert_allclose(a[..., 2], b[..., 2], rtol=0, atol=0)  # same depth image
    assert changed == 0
    # a hypothesis poking through the near plane is NOT culled (its drawn surface is open: the inside is visible)
    mtx = orc.pose_fwd(np.array([[0.0], [0.0], [0.0], [1.0], [0.0], [0.0], [-0.3]], np.float32))
    final = np.matmul(proj[None].astype(np.float32), mtx)
    clip = orc.xfm_fwd(pos[None], final, True)
    assert (clip[0, :, 3] <= 0).any() or (np.abs(clip[0, :, 2]) > clip[0, :, 3]).any()
    assert orc.view_volume_cull(pos, final, -1).tolist() == [0]  # a vertex outside => a box corner outside: both faces drawn
    # the box test is conservative: an object whose box pokes through the near plane while its vertices do not is not culled either
    near = orc.pose_fwd(np.array([[0.0], [0.0], [0.0], [1.0], [0.0], [0.0], [-(float(np.abs(pos).max()) + 0.012)]], np.float32))
    fin2 = np.matmul(proj[None].astype(np.float32), near)
    c2 = orc.xfm_fwd(pos[None], fin2, True)[0]
    if ((c2[:, 3] > 0) & (c2[:, 2] >= -c2[:, 3]) & (c2[:, 2] <= c2[:, 3])).all():
        assert orc.view_volume_cull(pos, fin2, -1).tolist() in ([0], [-1])
    assert orc.view_volume_cull(np.concatenate([pos, [[np.nan, 0, 0]]]).astype(np.float32), final, -1).tolist() == [0]  # non-finite mesh: never
    # what the rule avoids: FORCING it on an open surface removes what is seen through the hole
    open_tri = tri[np.linalg.norm(pos[tri].mean(1) - pos[tri].mean(1)[0], axis=1) > 0.35]  # cut a cap off
    mtx = orc.pose_fwd(np.concatenate([syn.random_quat(np.random.RandomState(0)), [0, 0, -2.0]])[:, None].astype(np.float32))
    clip = orc.xfm_fwd(pos[None], np.matmul(proj[None].astype(np.float32), mtx), True)
    both = np.stack([orc.rasterize_fwd(clip, open_tri, H, W, None if s == 0 else np.array([s], np.int32))[0, ..., 3] > 0 for s in (0, -1)])
    assert orc.mesh_cull_sign(pos, open_tri, proj) == 0 and both[0].sum() >= both[1].sum()


def test_d6_coverage_interpolates_one_per_vertex_not_per_index_row():
    """D6: the reference builds the silhouette's tensor of ones with the shape of the INDEX buffer
    (`torch.ones(pos_idx.shape)`, diffdope.py:212: [B,T,3]) and hands it to dr.interpolate as per-vertex attributes.  With more
    vertices than triangles -- any mesh un-merged along its uv seams or per wedge, V up to 3T -- vertex ids run past that tensor:
    dr.interpolate's kernel leaves such a pixel at zero (an index check), so the reference's mask has holes there.  This build
    interpolates one 1 per VERTEX: the mask is the coverage of the drawn triangles whatever V and T are."""
    H, W = 32, 40
    proj = orc.projection_matrix(fx=60.0, fy=60.0, cx=W / 2, cy=H / 2, im_width=W, im_height=H)
    # three triangles, nine vertices (nothing shared: V = 9 > T = 3)
    pos = np.array([[-0.3, -0.2, 0.0], [0.1, -0.25, 0.0], [-0.1, 0.25, 0.0], [0.05, -0.1, 0.1], [0.4, -0.05, 0.1], [0.2, 0.3, 0.1],
                    [-0.45, 0.1, 0.05], [-0.2, 0.15, 0.05], [-0.4, 0.35, 0.05]], np.float32)
    tri = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.int32)
    vcol = np.ones((9, 3), np.float32)
    R = orc.RenderOracle(pos, tri, proj, H, W, {}, dict(mask=1.0), vtx_color=vcol, dtype=np.float32)
    mtx = orc.pose_fwd(np.array([[0.0], [0.0], [0.0], [1.0], [0.0], [0.0], [-1.5]], np.float32))
    r = R.render(mtx)
    ids = r["rast"][0, ..., 3]
    assert (ids == 1).sum() > 20 and (ids == 2).sum() > 20 and (ids == 3).sum() > 5
    inner = (ids > 0) & np.roll(ids > 0, 1, 0) & np.roll(ids > 0, -1, 0) & np.roll(ids > 0, 1, 1) & np.roll(ids > 0, -1, 1)
    np.testing.assert_allclose(r["mask"][0][inner], 1.0, atol=1e-6)  # full coverage inside BOTH triangles
    # what the [T,3]-shaped tensor gives when the attribute lookup checks its bounds (only the vertex ids 0..2 lie inside its 3 rows)
    literal = orc.interpolate_fwd(np.ones((1, 3, 3), np.float32), r["rast"], tri)
    assert literal[0][ids == 1].min() > 0.999 and literal[0][ids == 2].max() == 0.0 and literal[0][ids == 3].max() == 0.0
