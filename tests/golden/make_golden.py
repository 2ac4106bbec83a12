"""Generate tests/golden/*.npz by IMPORTING the reference's Python in the build container.

Run here only (needs /root/reference):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
The reference source never leaves this container; only the numeric input/output vectors below
are committed.  What each fixture pins:
  G1 xfm      ops.xfm_points / xfm_vectors(use_python=True) fwd + autograd grads  (ops.py:128-175)
  G2 pose     Object3D.forward quaternion normalisation + matrix_batch_44_from_position_quat
              and grads w.r.t. the 7 parameters                                  (diffdope.py:46-89,1085-1098)
  G3 proj     Camera.get_projection_matrix                                         (diffdope.py:679-742)
  G4 losses   l1_rgb_with_mask / l1_depth_with_mask / l1_mask + dist_batch_lr     (diffdope.py:534-613)
  G5 lr       the LR schedule expression                                           (diffdope.py:1657-1661)
  G6 argmin   DiffDope.get_argmin / add_loss_value                                 (diffdope.py:1488-1513,1554-1571)
The non-use_python path of ops.py is never called (it would try to JIT the CUDA plugin).
"""
import importlib.util
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load_ref_ops():
    spec = importlib.util.spec_from_file_location("ref_ops", os.path.join(REF, "diffdope", "ops.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def load_ref_diffdope():
    for name in ["cv2", "hydra", "hydra.utils", "imageio", "nvdiffrast", "nvdiffrast.torch", "pyrr", "trimesh",
                 "icecream", "omegaconf"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    pkg = types.ModuleType("diffdope")
    pkg.__path__ = [os.path.join(REF, "diffdope")]
    sys.modules["diffdope"] = pkg
    torch.Tensor.cuda = lambda self, *a, **k: self  # reference hard-codes .cuda()
    spec = importlib.util.spec_from_file_location("diffdope.diffdope", os.path.join(REF, "diffdope", "diffdope.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["diffdope.diffdope"] = m
    spec.loader.exec_module(m)
    return m


def g1_xfm(ops):
    rng = np.random.RandomState(10)
    out = {}
    k = 0
    for B in (1, 3):
        for N in (1, 63, 64, 65, 257):
            for bcast in (False, True):
                pts = rng.normal(size=(1 if bcast else B, N, 3)).astype(np.float32)
                mtx = rng.normal(size=(B, 4, 4)).astype(np.float32)
                for is_points in (True, False):
                    p = torch.tensor(pts, requires_grad=True)
                    m = torch.tensor(mtx, requires_grad=True)
                    fn = ops.xfm_points if is_points else ops.xfm_vectors
                    o = fn(p, m, use_python=True)
                    g = torch.tensor(rng.normal(size=tuple(o.shape)).astype(np.float32))
                    o.backward(g)
                    pre = f"c{k}_"
                    out[pre + "points"] = pts
                    out[pre + "matrix"] = mtx
                    out[pre + "is_points"] = np.array(is_points)
                    out[pre + "out"] = o.detach().numpy()
                    out[pre + "dout"] = g.numpy()
                    out[pre + "dpoints"] = p.grad.numpy()  # summed over b when broadcast (autograd)
                    out[pre + "dmatrix"] = m.grad.numpy()
                    k += 1
    out["n_cases"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "g1_xfm.npz"), **out)


def g2_pose(dd):
    rng = np.random.RandomState(11)
    B = 5
    raw = rng.normal(size=(7, B)).astype(np.float32)
    raw[:4, 0] = [0, 0, 0, 1]  # identity
    raw[:4, 1] *= 3.0  # far from unit norm
    params = [torch.tensor(raw[i], requires_grad=True) for i in range(7)]
    # Object3D.forward (diffdope.py:1090-1096)
    q = torch.stack(params[:4], dim=0).T
    q = q / torch.norm(q, dim=1).reshape(-1, 1)
    t = torch.stack(params[4:], dim=0).T
    mtx = dd.matrix_batch_44_from_position_quat(p=t, q=q)
    g = torch.tensor(rng.normal(size=(B, 4, 4)).astype(np.float32))
    mtx.backward(g)
    np.savez_compressed(
        os.path.join(OUT, "g2_pose.npz"), params=raw, mtx=mtx.detach().numpy(), dmtx=g.numpy(),
        dparams=np.stack([p.grad.numpy() for p in params]),
    )


def g3_proj(dd):
    cams = [
        dict(fx=1390.53, fy=1386.99, cx=964.957, cy=522.586, im_width=1920, im_height=1080),
        dict(fx=463.51, fy=616.44, cx=321.652, cy=232.26, im_width=640, im_height=480),
        dict(fx=927.02, fy=924.66, cx=643.305, cy=348.39, im_width=1280, im_height=720),
        dict(fx=115.88, fy=154.11, cx=80.41, cy=58.07, im_width=160, im_height=120, znear=0.1, zfar=50.0),
    ]
    out = {}
    for i, c in enumerate(cams):
        cam = dd.Camera(**c)
        out[f"cam{i}_args"] = np.array([c[k] for k in ("fx", "fy", "cx", "cy", "im_width", "im_height")] +
                                       [c.get("znear", 0.01), c.get("zfar", 200)], np.float64)
        out[f"cam{i}_proj"] = cam.cam_proj.numpy()
    out["n"] = np.array(len(cams))
    np.savez_compressed(os.path.join(OUT, "g3_proj.npz"), **out)


def g4_losses(dd):
    rng = np.random.RandomState(12)
    B, H, W = 3, 6, 7
    seg1 = (rng.uniform(size=(H, W, 1)) > 0.4).astype(np.float32).repeat(3, axis=2)
    gt = dict(
        rgb=np.broadcast_to(rng.uniform(size=(H, W, 3)).astype(np.float32), (B, H, W, 3)).copy(),
        depth=np.broadcast_to(rng.uniform(1, 9, size=(H, W)).astype(np.float32), (B, H, W)).copy(),
        segmentation=np.broadcast_to(seg1, (B, H, W, 3)).copy(),
    )
    renders = dict(
        rgb=rng.uniform(size=(B, H, W, 3)).astype(np.float32),
        depth=rng.uniform(1, 9, size=(B, H, W)).astype(np.float32),
        mask=rng.uniform(size=(B, H, W, 3)).astype(np.float32),
    )
    lrs = rng.uniform(0.01, 100, size=B).astype(np.float32)
    weights = dict(weight_rgb=0.7, weight_depth=1.0, weight_mask=1.3)

    logged = {}

    class Fake:
        pass

    dd_obj = Fake()
    dd_obj.renders = {k: torch.tensor(v, requires_grad=True) for k, v in renders.items()}
    dd_obj.gt_tensors = {k: torch.tensor(v) for k, v in gt.items()}
    dd_obj.learning_rates = torch.tensor(lrs)
    dd_obj.cfg = types.SimpleNamespace(losses=types.SimpleNamespace(**weights))
    dd_obj.optimization_results = [{}]
    dd_obj.add_loss_value = lambda key, values, values_weighted=None: logged.__setitem__(key, values.detach().numpy())
    out = {f"gt_{k}": v for k, v in gt.items()}
    out.update({f"render_{k}": v for k, v in renders.items()})
    out["learning_rates"] = lrs
    out["weights"] = np.array([weights["weight_rgb"], weights["weight_depth"], weights["weight_mask"]])
    for name, fn, key, rk in [("rgb", dd.l1_rgb_with_mask, "rgb", "rgb"), ("depth", dd.l1_depth_with_mask, "depth", "depth"),
                              ("mask", dd.l1_mask, "mask_selection", "mask")]:
        loss = fn(dd_obj)
        loss.backward()
        out[f"loss_{name}"] = loss.detach().numpy()
        out[f"logged_{name}"] = logged[key]
        out[f"grad_{name}"] = dd_obj.renders[rk].grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g4_losses.npz"), **out)


def g5_lr():
    # the expression at diffdope.py:1657-1661, evaluated as the reference does (python floats)
    out = {}
    for i, (nb, base, decay) in enumerate([(60, 20, 0.1), (100, 5.0, 0.5), (10, 1.0, 0.9)]):
        lrs = []
        for iteration_now in range(nb + 1):
            itf = iteration_now / nb + 1
            lrs.append(base * decay**itf)
        out[f"s{i}_args"] = np.array([nb, base, decay], np.float64)
        out[f"s{i}_lr"] = np.array(lrs, np.float64)
    out["n"] = np.array(3)
    np.savez_compressed(os.path.join(OUT, "g5_lr.npz"), **out)


def g6_argmin(dd):
    rng = np.random.RandomState(13)
    its, B = 4, 9
    obj = dd.DiffDope.__new__(dd.DiffDope)
    obj.losses_values = {}
    vals = {}
    for key in ("rgb", "depth", "mask_selection"):
        v = rng.uniform(size=(its, B)).astype(np.float32)
        vals[key] = v
        for i in range(its):
            dd.DiffDope.add_loss_value(obj, key, torch.tensor(v[i]))
    am = int(dd.DiffDope.get_argmin(obj))
    np.savez_compressed(os.path.join(OUT, "g6_argmin.npz"), argmin=np.array(am),
                        **{f"v_{k}": v for k, v in vals.items()},
                        **{f"stored_{k}": obj.losses_values[k].numpy() for k in vals})


if __name__ == "__main__":
    assert os.path.isdir(REF), "golden vectors can only be generated where /root/reference exists"
    ops = load_ref_ops()
    g1_xfm(ops)
    dd = load_ref_diffdope()
    g2_pose(dd)
    g3_proj(dd)
    g4_losses(dd)
    g5_lr()
    g6_argmin(dd)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
