"""The oracle against the golden vectors captured from the reference's own Python
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np

from oracle import oracle as orc


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_xfm_forward_backward(golden_dir):
    g = _load(golden_dir, "g1_xfm.npz")
    n = int(g["n_cases"])
    assert n == 40
    for k in range(n):
        pre = f"c{k}_"
        pts, mtx, isp = g[pre + "points"], g[pre + "matrix"], bool(g[pre + "is_points"])
        out = orc.xfm_fwd(pts, mtx, isp)
        np.testing.assert_allclose(out, g[pre + "out"], rtol=1e-5, atol=1e-5)
        dp, dm = orc.xfm_bwd(pts, mtx, g[pre + "dout"], isp)
        if pts.shape[0] == 1 and mtx.shape[0] > 1:
            dp = dp.sum(axis=0, keepdims=True)  # autograd reduces the broadcast batch
        np.testing.assert_allclose(dp, g[pre + "dpoints"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(dm, g[pre + "dmatrix"], rtol=1e-4, atol=1e-4)


def test_g2_pose_matrix_and_grads(golden_dir):
    g = _load(golden_dir, "g2_pose.npz")
    mtx = orc.pose_fwd(g["params"])
    np.testing.assert_allclose(mtx, g["mtx"], rtol=1e-5, atol=1e-6)
    dpar = orc.pose_bwd(g["params"], g["dmtx"])
    np.testing.assert_allclose(dpar, g["dparams"], rtol=2e-4, atol=2e-5)
    # float64 build agrees too
    mtx64 = orc.pose_fwd(g["params"].astype(np.float64))
    np.testing.assert_allclose(mtx64, g["mtx"], rtol=1e-5, atol=1e-6)


def test_g3_projection(golden_dir):
    g = _load(golden_dir, "g3_proj.npz")
    for i in range(int(g["n"])):
        a = g[f"cam{i}_args"]
        proj = orc.projection_matrix(a[0], a[1], a[2], a[3], int(a[4]), int(a[5]), a[6], a[7])
        np.testing.assert_allclose(proj, g[f"cam{i}_proj"], rtol=1e-12, atol=1e-12)


def test_g4_losses(golden_dir):
    g = _load(golden_dir, "g4_losses.npz")
    lrs, w = g["learning_rates"], g["weights"]
    B = lrs.shape[0]
    seg = g["gt_segmentation"]
    cases = [
        ("rgb", lambda s, wg: orc.loss_rgb(g["render_rgb"], g["gt_rgb"], seg, s, wg), w[0]),
        ("depth", lambda s, wg: orc.loss_depth(g["render_depth"], g["gt_depth"], seg, s, wg), w[1]),
        ("mask", lambda s, wg: orc.loss_mask(g["render_mask"], seg, s, wg), w[2]),
    ]
    for name, fn, wk in cases:
        scale = (lrs * wk / B).astype(np.float32)  # d/d img of  w * mean_b(lr_b * mean_px)
        per, dimg = fn(scale, True)
        np.testing.assert_allclose(per * wk, g[f"logged_{name}"], rtol=1e-5, atol=1e-7)
        total = float(np.mean(per.astype(np.float64) * lrs) * wk)
        np.testing.assert_allclose(total, float(g[f"loss_{name}"]), rtol=1e-5)
        np.testing.assert_allclose(dimg, g[f"grad_{name}"], rtol=1e-4, atol=1e-7)


def test_g5_lr_schedule(golden_dir):
    g = _load(golden_dir, "g5_lr.npz")
    for i in range(int(g["n"])):
        nb, base, decay = g[f"s{i}_args"]
        lrs = orc.lr_schedule(int(nb), base, decay)
        np.testing.assert_allclose(lrs, g[f"s{i}_lr"], rtol=1e-14)
    # the yaml default goes 2.0 -> 0.2
    lrs = orc.lr_schedule(60, 20, 0.1)
    assert abs(lrs[0] - 2.0) < 1e-12 and abs(lrs[-1] - 0.2) < 1e-12 and len(lrs) == 61


def test_g6_argmin(golden_dir):
    g = _load(golden_dir, "g6_argmin.npz")
    vals = {k: g[f"v_{k}"] for k in ("rgb", "depth", "mask_selection")}
    assert orc.argmin_losses(vals) == int(g["argmin"])
    for k in vals:
        np.testing.assert_array_equal(vals[k], g[f"stored_{k}"])
