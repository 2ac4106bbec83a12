#!/usr/bin/env python
"""Write a small self-contained example scene in the reference's file formats (PLY with texture_u/v +
TextureFile comment in millimetres, 8-bit rgb/seg PNGs, 16-bit depth PNG in millimetres) under
examples/data/, rendered with the HIP renderer at a known pose.  Needs a GPU.  The config that goes with it
is configs/diffdope.yaml (pose there = the generating pose perturbed)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffdope_amd as dd  # noqa: E402
from diffdope_amd import synthetic as syn  # noqa: E402


def main(out=os.path.join(ROOT, "examples", "data"), W=1920, H=1080):
    from PIL import Image as PILImage

    os.makedirs(os.path.join(out, "mesh"), exist_ok=True)
    os.makedirs(os.path.join(out, "scene"), exist_ok=True)
    pos, tri, uv = syn.blob_mesh(40, 64, seed=0)  # ~1 unit across
    pos_mm = pos * 100.0  # object3d.scale 0.01 brings millimetres to scene units (10 cm = 1 unit)
    tex = (syn.texture(512, seed=1) * 255).astype(np.uint8)
    PILImage.fromarray(tex).save(os.path.join(out, "mesh", "blob.png"))
    with open(os.path.join(out, "mesh", "blob.ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment TextureFile blob.png\n")
        f.write(f"element vertex {len(pos)}\nproperty float x\nproperty float y\nproperty float z\nproperty float texture_u\nproperty float texture_v\n")
        f.write(f"element face {len(tri)}\nproperty list uchar int vertex_indices\nend_header\n")
        for p, t in zip(pos_mm, uv):
            f.write(f"{p[0]:.4f} {p[1]:.4f} {p[2]:.4f} {t[0]:.6f} {1 - t[1]:.6f}\n")  # PLY v is bottom-up; Mesh flips it back
        for t in tri:
            f.write(f"3 {t[0]} {t[1]} {t[2]}\n")
    # ground-truth pose in the OpenCV frame (millimetres), like the yaml of the reference
    t_cv = np.array([10.0, -25.0, 747.0])
    q_cv = syn.quat_from_axis_angle([0.3, 1.0, 0.2], 0.4)
    cam = dd.Camera(**syn.YAML_CAMERA)
    mesh = dd.Mesh(os.path.join(out, "mesh", "blob.ply"), scale=0.01)
    obj = dd.Object3D(position=list(t_cv), rotation=list(q_cv), batchsize=1, opencv2opengl=True, scale=0.01, mesh=mesh)
    obj.cuda(); cam.cuda(); cam.set_batchsize(1); obj.set_batchsize(1)
    with torch.no_grad():
        r = obj()
        mtx = dd.matrix_batch_44_from_position_quat(p=r["trans"], q=r["quat"])
        o = dd.render_texture_batch(dd.RasterizeGLContext(), cam.cam_proj, mtx, r["pos"], r["pos_idx"], [H, W], uv=r["uv"],
                                    uv_idx=r["uv_idx"], tex=r["tex"], return_rast_out=True)
    cov = (o["rast_out"][0, ..., 3] > 0).cpu().numpy()[::-1]  # files are top-down, tensors bottom-up
    rgb = (o["rgb"][0].clamp(0, 1).cpu().numpy()[::-1] * 255).round().astype(np.uint8)
    rgb[~cov] = 90  # flat background
    depth_mm = (o["depth"][0].cpu().numpy()[::-1] * 100.0 * cov).round().astype(np.uint16)  # depth PNG / depth_scale(100) = units
    PILImage.fromarray(rgb).save(os.path.join(out, "scene", "rgb.png"))
    PILImage.fromarray(depth_mm).save(os.path.join(out, "scene", "depth.png"))
    PILImage.fromarray((cov * 255).astype(np.uint8), mode="L").save(os.path.join(out, "scene", "seg.png"))
    np.savetxt(os.path.join(out, "scene", "gt_pose_opencv_mm.txt"), np.concatenate([t_cv, q_cv])[None], header="x y z qx qy qz qw")
    print("wrote", out, "coverage", cov.mean())


if __name__ == "__main__":
    main()
