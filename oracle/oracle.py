"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/ctypes front end of the CPU oracle (oracle/ddx_oracle.c) plus the pieces that are
simplest to restate in numpy (projection matrix, LR schedule, argmin, the op-by-op render graph
of diffdope/diffdope.py:156-234 and the optimiser loop :1634-1714).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status per function is stated in the header of ddx_oracle.c ("parity unpinned" for the
nvdiffrast-side ops).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the C oracle with gcc (both precisions)."""
    outs = [os.path.join(_HERE, "_build", n) for n in ("liborc_f32.so", "liborc_f64.so")]
    src = os.path.join(_HERE, "ddx_oracle.c")
    stale = force or any((not os.path.exists(o)) or os.path.getmtime(o) < os.path.getmtime(src) for o in outs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "all"], check=True, capture_output=True)
    return outs


def _lib(dtype):
    dtype = np.dtype(dtype)
    key = dtype.name
    if key not in _LIBS:
        name = {"float32": "liborc_f32.so", "float64": "liborc_f64.so"}[key]
        path = os.path.join(_HERE, "_build", name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        assert lib.orc_sizeof_real() == dtype.itemsize
        _LIBS[key] = lib
    return _LIBS[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------------------------
# xfm (ops.py:128-175)
def xfm_fwd(points, matrix, is_points=True):
    dt = matrix.dtype
    points, matrix = _c(points, dt), _c(matrix, dt)
    B, N = matrix.shape[0], points.shape[1]
    out = np.empty((B, N, 4 if is_points else 3), dt)
    _lib(dt).orc_xfm_fwd(_p(points), points.shape[0], _p(matrix), B, N, int(is_points), _p(out))
    return out


def xfm_bwd(points, matrix, dout, is_points=True):
    dt = matrix.dtype
    points, matrix, dout = _c(points, dt), _c(matrix, dt), _c(dout, dt)
    B, N = matrix.shape[0], points.shape[1]
    dp = np.empty((B, N, 3), dt)
    dm = np.empty((B, 4, 4), dt)
    _lib(dt).orc_xfm_bwd(_p(points), points.shape[0], _p(matrix), B, N, int(is_points), _p(dout), _p(dp), _p(dm))
    return dp, dm


# --------------------------------------------------------------------------------------------
# pose (diffdope.py:46-89, 1085-1098); params [7,B] = qx,qy,qz,qw,x,y,z
def pose_fwd(params):
    dt = params.dtype
    params = _c(params, dt)
    B = params.shape[1]
    mtx = np.empty((B, 4, 4), dt)
    _lib(dt).orc_pose_fwd(_p(params), B, _p(mtx))
    return mtx


def pose_bwd(params, dmtx):
    dt = params.dtype
    params, dmtx = _c(params, dt), _c(dmtx, dt)
    B = params.shape[1]
    dpar = np.empty((7, B), dt)
    _lib(dt).orc_pose_bwd(_p(params), B, _p(dmtx), _p(dpar))
    return dpar


# --------------------------------------------------------------------------------------------
# Camera.get_projection_matrix (diffdope.py:679-742), "y_down" window convention
def projection_matrix(fx, fy, cx, cy, im_width, im_height, znear=0.01, zfar=200.0):
    w, h = im_width, im_height
    depth = float(zfar - znear)
    q = -(zfar + znear) / depth
    qn = -2 * (zfar * znear) / depth
    return np.array(
        [
            [2 * fx / w, -2 * 0 / w, (-2 * cx + w) / w, 0],
            [0, 2 * fy / h, (2 * cy - h) / h, 0],
            [0, 0, q, qn],
            [0, 0, -1, 0],
        ],
        dtype=np.float64,
    )


def lr_schedule(nb_iterations, base_lr, lr_decay):
    """diffdope.py:1656-1661 -- one lr per iteration, nb_iterations+1 of them."""
    return [base_lr * lr_decay ** (it / nb_iterations + 1) for it in range(nb_iterations + 1)]


def argmin_losses(losses_values):
    """diffdope.py:1488-1513: mean over keys of the last-step per-hypothesis loss, argmin."""
    stacked = np.stack([np.asarray(v)[-1] for v in losses_values.values()], axis=0)
    return int(np.argmin(stacked.mean(axis=0)))


# --------------------------------------------------------------------------------------------
# renderer ops (nvdiffrast semantics at diffdope.py:143-231) -- parity unpinned
def rasterize_fwd(pos, tri, H, W, cull=None):
    """cull: None (both faces, dr.rasterize) or int32 [B]: per hypothesis 0 or the snapped-area sign of the back faces to skip
    (deviation D5, see view_volume_cull)."""
    dt = pos.dtype
    pos, tri = _c(pos, dt), _i32(tri)
    B, V = pos.shape[:2]
    rast = np.empty((B, H, W, 4), dt)
    cb = None if cull is None else np.ascontiguousarray(np.broadcast_to(np.asarray(cull, np.int32), (B,)))
    _lib(dt).orc_rasterize_fwd(_p(pos), B, V, _p(tri), tri.shape[0], H, W, _p(rast), _p(cb) if cb is not None else None)
    return rast


def view_volume_cull(pos, final, cull_sign):
    """Per-hypothesis culling decision of the fused engine (deviation D5): cull_sign where the 8 corners of the object-space
    bounding box of ALL vertices, taken through the same k-ordered fma transform as the vertices (xfm_fwd), have w > 0 and
    -w <= z <= w -- the three conditions are half-spaces of object space, so they then hold at every vertex and the closed
    surface is drawn whole -- else 0.  A mesh with a non-finite coordinate never culls."""
    dt = final.dtype
    B = final.shape[0]
    if not cull_sign or not np.all(np.isfinite(pos)):
        return np.zeros(B, np.int32)
    lo, hi = pos.min(0), pos.max(0)
    corners = np.array([[hi[0] if c & 1 else lo[0], hi[1] if c & 2 else lo[1], hi[2] if c & 4 else lo[2]] for c in range(8)], dt)
    cc = xfm_fwd(corners[None], final, True)  # [B,8,4]
    ok = (cc[..., 3] > 0) & (cc[..., 2] >= -cc[..., 3]) & (cc[..., 2] <= cc[..., 3])
    return np.where(ok.all(1), np.int32(cull_sign), np.int32(0)).astype(np.int32)


def mesh_cull_sign(pos, tri, proj):
    """Deviation D5 (fused engine only): the sign of the snapped screen area of BACK-facing triangles when `pos`/`tri` is a closed,
    consistently oriented surface and `proj` a pinhole projection, else 0 (draw both faces).  Vertices are welded by position
    (bit-equal float32 coordinates: uv seams duplicate vertices); triangles with two welded corners equal are ignored; every
    welded edge must belong to exactly two triangles running through it in opposite directions, and every connected shell must
    have the sign of volume of the whole.  A camera-space triangle has
    det[p0;p1;p2] > 0 exactly when its counter-clockwise normal points away from the camera; the projection multiplies that
    orientation by det A (A = rows x, y, w of proj, whose 4th column must vanish), and counter-clockwise is outward when the
    signed volume is positive: sign = sign(volume) * sign(det A)."""
    pos32 = np.ascontiguousarray(pos, np.float32) + np.float32(0)  # (-0 -> +0)
    tri = np.asarray(tri, np.int64)
    V = pos32.shape[0]
    if tri.size == 0 or tri.min() < 0 or tri.max() >= V:
        return 0
    _, canon = np.unique(pos32.view(np.uint32).reshape(V, 3), axis=0, return_inverse=True)
    c = np.asarray(canon).reshape(-1)[tri]
    keep = (c[:, 0] != c[:, 1]) & (c[:, 1] != c[:, 2]) & (c[:, 0] != c[:, 2])
    c, t = c[keep], tri[keep]
    if len(c) == 0:
        return 0
    a = np.concatenate([c[:, 0], c[:, 1], c[:, 2]])
    b = np.concatenate([c[:, 1], c[:, 2], c[:, 0]])
    lo, hi, fwd = np.minimum(a, b), np.maximum(a, b), (a < b)
    key = lo.astype(np.int64) * (V + 1) + hi
    order = np.argsort(key, kind="stable")
    key, fwd = key[order], fwd[order]
    if len(key) % 2:
        return 0
    k0, k1, f0, f1 = key[0::2], key[1::2], fwd[0::2], fwd[1::2]
    if not (np.all(k0 == k1) and np.all(f0 != f1) and np.all(k0[1:] != k0[:-1])):
        return 0
    p = np.asarray(pos, np.float64)
    p0, p1, p2 = p[t[:, 0]], p[t[:, 1]], p[t[:, 2]]
    tvol = np.einsum("ij,ij->i", p0, np.cross(p1, p2))
    vol6 = float(tvol.sum())
    # every shell (connected component over welded vertices) must be oriented like the whole
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    nc = int(np.asarray(canon).max()) + 1
    g = coo_matrix((np.ones(len(a), np.int8), (a, b)), shape=(nc, nc))
    _, label = connected_components(g, directed=False)
    svol = np.bincount(label[c[:, 0]], weights=tvol)
    sabs = np.bincount(label[c[:, 0]], weights=np.abs(tvol))
    # ... and a real volume: a flat two-sided patch (the same triangles wound both ways) is "closed" with a volume that is round-off
    if np.any((sabs > 0) & (~(np.abs(svol) > 1e-9 * sabs) | ((svol > 0) != (vol6 > 0)))):
        return 0
    P = np.asarray(proj, np.float64)
    if not (P[0, 3] == 0 and P[1, 3] == 0 and P[3, 3] == 0):
        return 0
    detA = float(np.linalg.det(P[[0, 1, 3]][:, :3]))
    if vol6 == 0 or detA == 0 or not np.isfinite(vol6 * detA):
        return 0
    return 1 if (vol6 > 0) == (detA > 0) else -1


def set_compat(nvdiffrast):
    """Deviation D2 switch for both precisions: True = the rasterize backward differentiates the unclamped barycentrics
    (nvdiffrast's published rule); False = the true derivative of the clamped forward (this build).  Returns the old setting."""
    old = 0
    for dt in (np.float32, np.float64):
        old = _lib(dt).orc_set_unclamped_bary_grad(int(bool(nvdiffrast)))
    return bool(old)


def rasterize_bwd(pos, tri, rast, drast):
    dt = pos.dtype
    pos, tri, rast, drast = _c(pos, dt), _i32(tri), _c(rast, dt), _c(drast, dt)
    B, V = pos.shape[:2]
    H, W = rast.shape[1:3]
    dpos = np.zeros((B, V, 4), dt)
    _lib(dt).orc_rasterize_bwd(_p(pos), B, V, _p(tri), tri.shape[0], H, W, _p(rast), _p(drast), _p(dpos))
    return dpos


def interpolate_fwd(attr, rast, tri):
    dt = rast.dtype
    attr, rast, tri = _c(attr, dt), _c(rast, dt), _i32(tri)
    B, H, W = rast.shape[:3]
    Ba, Va, A = attr.shape
    out = np.empty((B, H, W, A), dt)
    _lib(dt).orc_interpolate_fwd(_p(attr), Ba, Va, A, _p(rast), B, H, W, _p(tri), _p(out))
    return out


def interpolate_bwd(attr, rast, tri, dout, want_dattr=False):
    dt = rast.dtype
    attr, rast, tri, dout = _c(attr, dt), _c(rast, dt), _i32(tri), _c(dout, dt)
    B, H, W = rast.shape[:3]
    Ba, Va, A = attr.shape
    dattr = np.zeros_like(attr) if want_dattr else None
    drast = np.zeros((B, H, W, 4), dt)
    _lib(dt).orc_interpolate_bwd(_p(attr), Ba, Va, A, _p(rast), B, H, W, _p(tri), _p(dout), _p(dattr), _p(drast))
    return dattr, drast


def texture_fwd(tex, uv):
    dt = uv.dtype
    tex, uv = _c(tex, dt), _c(uv, dt)
    Bt, Th, Tw, C = tex.shape
    B, H, W = uv.shape[:3]
    out = np.empty((B, H, W, C), dt)
    _lib(dt).orc_texture_fwd(_p(tex), Bt, Th, Tw, C, _p(uv), B, H, W, _p(out))
    return out


def texture_bwd(tex, uv, dout, want_dtex=False):
    dt = uv.dtype
    tex, uv, dout = _c(tex, dt), _c(uv, dt), _c(dout, dt)
    Bt, Th, Tw, C = tex.shape
    B, H, W = uv.shape[:3]
    duv = np.empty((B, H, W, 2), dt)
    dtex = np.zeros_like(tex) if want_dtex else None
    _lib(dt).orc_texture_bwd(_p(tex), Bt, Th, Tw, C, _p(uv), B, H, W, _p(dout), _p(duv), _p(dtex))
    return duv, dtex


def build_opposite(tri):
    tri = _i32(tri)
    opp = np.empty_like(tri)
    _lib(np.float32).orc_build_opposite(_p(tri), tri.shape[0], _p(opp))
    return opp


def antialias_fwd(color, rast, pos, tri, opp=None):
    dt = rast.dtype
    color, rast, pos, tri = _c(color, dt), _c(rast, dt), _c(pos, dt), _i32(tri)
    opp = build_opposite(tri) if opp is None else _i32(opp)
    B, H, W, C = color.shape
    out = np.empty_like(color)
    _lib(dt).orc_antialias_fwd(_p(color), C, _p(rast), _p(pos), B, pos.shape[1], H, W, _p(tri), _p(opp), _p(out))
    return out


def antialias_bwd(color, rast, pos, tri, dout, opp=None):
    dt = rast.dtype
    color, rast, pos, tri, dout = _c(color, dt), _c(rast, dt), _c(pos, dt), _i32(tri), _c(dout, dt)
    opp = build_opposite(tri) if opp is None else _i32(opp)
    B, H, W, C = color.shape
    dcolor = np.empty_like(color)
    dpos = np.zeros_like(pos)
    _lib(dt).orc_antialias_bwd(
        _p(color), C, _p(rast), _p(pos), B, pos.shape[1], H, W, _p(tri), _p(opp), _p(dout), _p(dcolor), _p(dpos)
    )
    return dcolor, dpos


# --------------------------------------------------------------------------------------------
# losses (diffdope.py:534-613)
def _loss(fn, img, gts, seg, scale, want_grad):
    dt = img.dtype
    B, H, W = img.shape[:3]
    img, seg = _c(img, dt), _c(seg, dt)
    scale = np.ones(B, dt) if scale is None else _c(scale, dt)
    per = np.empty(B, dt)
    dimg = np.empty_like(img) if want_grad else None
    lib = _lib(dt)
    if fn == "mask":
        lib.orc_loss_mask(_p(img), _p(seg), seg.shape[0], B, H, W, _p(scale), _p(per), _p(dimg))
    else:
        gt = _c(gts, dt)
        getattr(lib, "orc_loss_" + fn)(_p(img), _p(gt), _p(seg), seg.shape[0], B, H, W, _p(scale), _p(per), _p(dimg))
    return per, dimg


def loss_rgb(rgb, gt_rgb, seg, scale=None, want_grad=False):
    return _loss("rgb", rgb, gt_rgb, seg, scale, want_grad)


def loss_depth(depth, gt_depth, seg, scale=None, want_grad=False):
    return _loss("depth", depth, gt_depth, seg, scale, want_grad)


def loss_mask(mask, seg, scale=None, want_grad=False):
    return _loss("mask", mask, None, seg, scale, want_grad)


def loss_edge(rgb, gt_rgb, seg, scale=None, want_grad=False):
    """EXTENSION, no reference counterpart (see orc_loss_edge): L1 between the Sobel gradients of the
    luminance of the rendered colour image and of gt_rgb * seg."""
    return _loss("edge", rgb, gt_rgb, seg, scale, want_grad)


# --------------------------------------------------------------------------------------------
# The op-by-op render graph of render_texture_batch (diffdope.py:156-234) and its reverse.
class RenderOracle:
    """Holds one mesh + camera + observed images and evaluates loss and d loss / d pose.

    pos [V,3], tri [T,3], and either (uv [V,2], tex [Th,Tw,3]) or vtx_color [V,3];
    gt: dict with 'rgb' [H,W,3], 'depth' [H,W], 'segmentation' [H,W,3] (any subset);
    weights: dict(rgb=, depth=, mask=, edge=) of floats or None to disable that term ("edge" is this
    build's extension, see orc_loss_edge).
    """

    def __init__(self, pos, tri, proj, H, W, gt, weights, uv=None, tex=None, vtx_color=None, dtype=np.float32, cull_backfaces=False):
        self.dt = np.dtype(dtype)
        self.pos = _c(pos, self.dt)
        self.tri = _i32(tri)
        self.opp = build_opposite(self.tri)
        self.proj = _c(proj, self.dt)
        self.H, self.W = H, W
        self.uv = None if uv is None else _c(uv, self.dt)
        self.tex = None if tex is None else _c(tex, self.dt)
        self.vtx_color = None if vtx_color is None else _c(vtx_color, self.dt)
        self.gt = {k: _c(v, self.dt)[None] for k, v in gt.items()}
        self.weights = weights
        # cull_backfaces=True: the fused engine's visibility (deviation D5: back faces of a closed mesh skipped while a hypothesis
        # lies inside the view volume); False: nvdiffrast's (both faces), what the op-level renderer ops implement
        self.cull_backfaces = cull_backfaces
        self._cull_sign = None

    def render(self, mtx):
        """Forward graph; returns dict of intermediates (all [B,...])."""
        dt = self.dt
        mtx = _c(mtx, dt)
        B = mtx.shape[0]
        final = np.matmul(self.proj[None], mtx).astype(dt)  # diffdope.py:195
        pos_clip = xfm_fwd(self.pos[None], final, True)  # :196
        if self._cull_sign is None:
            self._cull_sign = mesh_cull_sign(self.pos, self.tri, self.proj)
        cull = view_volume_cull(self.pos, final, self._cull_sign) if self.cull_backfaces else None
        rast = rasterize_fwd(pos_clip, self.tri, self.H, self.W, cull)  # :198
        posw = np.concatenate([self.pos, np.ones((self.pos.shape[0], 1), dt)], axis=1)[None]
        gb_pos = interpolate_fwd(posw, rast, self.tri)  # :203
        gb3 = np.ascontiguousarray(gb_pos[..., :3]).reshape(B, -1, 3)
        depth = -xfm_fwd(gb3, mtx, True).reshape(B, self.H, self.W, 4)[..., 2]  # :208-209
        # :212 -- the reference builds the tensor of ones with the SHAPE OF THE INDEX BUFFER ([T,3]) and hands it to dr.interpolate
        # as per-vertex attributes: with more vertices than triangles (a mesh un-merged along its uv seams) the vertex ids run
        # past it.  Meant, and restated here, is the interpolation of one 1 per VERTEX = the coverage image (deviation D6).
        ones = np.ones((1, self.pos.shape[0], 3), dt)
        cov = interpolate_fwd(ones, rast, self.tri)
        mask = antialias_fwd(cov, rast, pos_clip, self.tri, self.opp)  # :214
        cov1 = np.clip(rast[..., 3:], 0, 1)
        r = dict(final=final, pos_clip=pos_clip, rast=rast, gb3=gb3, depth=depth, cov=cov, mask=mask)
        if self.vtx_color is None:
            texc = interpolate_fwd(self.uv[None], rast, self.tri)  # :218
            col = texture_fwd(self.tex[None], texc)  # :221
            r.update(texc=texc, col=col, rgb=col * cov1)  # :228
        else:
            col = interpolate_fwd(self.vtx_color[None], rast, self.tri)  # :230
            r.update(col=col, rgb=col * cov1)
        return r

    def loss_and_grad(self, params, lr_mult=None, want_grad=True, global_B=None):
        """params [7,B].  Returns (total, per_key_per_hyp dict (weighted, no LR), dparams, renders)."""
        dt = self.dt
        params = _c(params, dt)
        B = params.shape[1]
        GB = B if global_B is None else global_B
        lr_mult = np.ones(B, dt) if lr_mult is None else _c(lr_mult, dt)
        mtx = pose_fwd(params)
        r = self.render(mtx)
        w = self.weights
        seg = self.gt.get("segmentation")
        logs = {}
        total = 0.0
        d_rgb = d_depth = d_mask = None
        if w.get("rgb") is not None:
            per, d_rgb = loss_rgb(r["rgb"], self.gt["rgb"], seg, lr_mult * dt.type(w["rgb"] / GB), want_grad)
            logs["rgb"] = per * w["rgb"]
            total += float(np.sum(per.astype(np.float64) * lr_mult) / GB * w["rgb"])
        if w.get("depth") is not None:
            per, d_depth = loss_depth(r["depth"], self.gt["depth"], seg, lr_mult * dt.type(w["depth"] / GB), want_grad)
            logs["depth"] = per * w["depth"]
            total += float(np.sum(per.astype(np.float64) * lr_mult) / GB * w["depth"])
        if w.get("mask") is not None:
            per, d_mask = loss_mask(r["mask"], seg, lr_mult * dt.type(w["mask"] / GB), want_grad)
            logs["mask_selection"] = per * w["mask"]
            total += float(np.sum(per.astype(np.float64) * lr_mult) / GB * w["mask"])
        if w.get("edge") is not None:
            per, d_edge = loss_edge(r["rgb"], self.gt["rgb"], seg, lr_mult * dt.type(w["edge"] / GB), want_grad)
            logs["edge"] = per * w["edge"]
            total += float(np.sum(per.astype(np.float64) * lr_mult) / GB * w["edge"])
            if want_grad:
                d_rgb = d_edge if d_rgb is None else d_rgb + d_edge
        if not want_grad:
            return total, logs, None, r
        rast, tri = r["rast"], self.tri
        drast = np.zeros_like(rast)
        dpos_clip = np.zeros_like(r["pos_clip"])
        dmtx = np.zeros((B, 4, 4), dt)
        if d_rgb is not None:
            dcol = d_rgb * np.clip(rast[..., 3:], 0, 1)
            if self.vtx_color is None:
                duv, _ = texture_bwd(self.tex[None], r["texc"], dcol)
                _, dr_ = interpolate_bwd(self.uv[None], rast, tri, duv)
            else:
                _, dr_ = interpolate_bwd(self.vtx_color[None], rast, tri, dcol)
            drast += dr_
        if d_depth is not None:
            dout = np.zeros((B, self.H * self.W, 4), dt)
            dout[..., 2] = -d_depth.reshape(B, -1)
            dgb3, dm = xfm_bwd(r["gb3"], mtx, dout, True)
            dmtx += dm
            dgb = np.zeros((B, self.H, self.W, 4), dt)
            dgb[..., :3] = dgb3.reshape(B, self.H, self.W, 3)
            posw = np.concatenate([self.pos, np.ones((self.pos.shape[0], 1), dt)], axis=1)[None]
            _, dr_ = interpolate_bwd(posw, rast, tri, dgb)
            drast += dr_
        if d_mask is not None:
            dcov, dp = antialias_bwd(r["cov"], rast, r["pos_clip"], tri, d_mask, self.opp)
            dpos_clip += dp
            # d cov -> interpolate(ones) -> (u,v): identically zero (all attributes equal)
        dpos_clip += rasterize_bwd(r["pos_clip"], tri, rast, drast)
        _, dfinal = xfm_bwd(self.pos[None], r["final"], dpos_clip, True)
        dmtx += np.matmul(self.proj.T[None], dfinal).astype(dt)  # final = proj @ mtx
        dparams = pose_bwd(params, dmtx)
        return total, logs, dparams, r

    def optimise(self, params0, lr_mult, lrs, optimizer="sgd", adam=(0.9, 0.999, 1e-8), global_B=None):
        """diffdope.py:1634-1714: SGD (the reference's optimiser, :1642-1644) or Adam (north_star; torch.optim.Adam's
        update, step counted from 1) -- returns final params, loss logs, mtx history."""
        dt = self.dt.type
        params = _c(params0, self.dt).copy()
        logs_hist = {}
        mtx_hist = []
        m1 = np.zeros_like(params)
        m2 = np.zeros_like(params)
        b1, b2, eps = (dt(x) for x in adam)
        for step, lr in enumerate(lrs, start=1):
            total, logs, g, r = self.loss_and_grad(params, lr_mult, global_B=global_B)
            mtx_hist.append(pose_fwd(params))
            for k, v in logs.items():
                logs_hist.setdefault(k, []).append(v.copy())
            if optimizer == "sgd":
                params = (params - dt(lr) * g).astype(self.dt)
            else:
                m1 = (b1 * m1 + (dt(1) - b1) * g).astype(self.dt)
                m2 = (b2 * m2 + (dt(1) - b2) * g * g).astype(self.dt)
                c1, c2 = dt(1.0 - float(b1) ** step), dt(1.0 - float(b2) ** step)
                params = (params - dt(lr) * (m1 / c1) / (np.sqrt(m2 / c2) + eps)).astype(self.dt)
        return params, {k: np.stack(v) for k, v in logs_hist.items()}, np.stack(mtx_hist)
