"""Seeded scenes shared by the tests (numpy only)."""
import numpy as np

from diffdope_amd import synthetic as syn
from oracle import oracle as orc


def make_scene(rows=10, cols=14, H=48, W=64, B=2, tex_size=32, seed=0, dist=2.0, textured=True, rot_deg=8.0, trans=0.03):
    """Returns a dict with mesh, camera, GT images rendered by the f32 oracle at the GT pose, and
    B initial parameter vectors (perturbed copies of the GT pose, non-unit quaternions)."""
    pos, tri, uv = syn.blob_mesh(rows, cols, seed=seed)
    tex = syn.texture(tex_size, seed=seed + 1)
    vcol = syn.vertex_colors(pos, seed=seed + 5)
    proj = orc.projection_matrix(**syn.camera_intrinsics(W, H)).astype(np.float32)
    rng = np.random.RandomState(seed + 2)
    q_gt, t_gt = syn.random_quat(rng), np.array([0.05 * dist, -0.03 * dist, -dist])
    weights = dict(rgb=0.7, depth=1.0, mask=1.0)
    kw = dict(uv=uv, tex=tex) if textured else dict(vtx_color=vcol)
    # both faces drawn: dr.rasterize's rule (diffdope.py:198-200), the default of RefineEngine, the DiffDope API and the op-level
    # ops alike; tests of deviation D5 (back faces of the closed mesh culled) set R.cull_backfaces = True and ask the engine for it
    R = orc.RenderOracle(pos, tri, proj, H, W, {}, weights, dtype=np.float32, cull_backfaces=False, **kw)
    p_gt = np.concatenate([q_gt, t_gt])[:, None].astype(np.float32)
    r = R.render(orc.pose_fwd(p_gt))
    cov = r["rast"][0, ..., 3] > 0
    gt = dict(rgb=r["rgb"][0].copy(), depth=r["depth"][0].copy(), segmentation=np.repeat(cov[..., None], 3, -1).astype(np.float32))
    R.gt = {k: v[None] for k, v in gt.items()}
    params = []
    for b in range(B):
        q, t = syn.perturb_pose(q_gt, t_gt, rot_deg * (0.5 + 0.5 * rng.uniform()), trans * rng.uniform(), rng)
        params.append(np.concatenate([q * rng.uniform(0.7, 1.4), t]))
    params = np.stack(params, 1).astype(np.float32)
    lr_mult = rng.uniform(0.5, 2.0, size=B).astype(np.float32)
    return dict(pos=pos, tri=tri, uv=uv, tex=tex, vtx_color=vcol, proj=proj, H=H, W=W, B=B, gt=gt, params=params,
                lr_mult=lr_mult, p_gt=p_gt, q_gt=q_gt, t_gt=t_gt, oracle=R, textured=textured, coverage=float(cov.mean()))


def clip_from_pixels(xy, H, W, z=0.0, w=1.0):
    xy = np.asarray(xy, np.float64)
    x = (xy[:, 0] / W * 2 - 1) * w
    y = (xy[:, 1] / H * 2 - 1) * w
    return np.stack([x, y, np.full(len(xy), z) * w, np.full(len(xy), w)], axis=1).astype(np.float32)
