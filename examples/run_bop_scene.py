#!/usr/bin/env python
"""The reference's examples/run_bop_scene.py flow on this build (BASELINE config 5 in miniature): one frame, several
objects, each with its own visible mask and a noisy initial pose in the BOP json format, refined independently with B
hypotheses each; objects shard over ranks and ONE all_reduce hands every rank every object's best pose.

    python examples/run_bop_scene.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 examples/run_bop_scene.py

No BOP data ships with the repository (and there is no network), so the frame is synthetic: three textured blob
meshes rendered by the library's own renderer into one rgb / depth image with per-object masks, and the initial
poses are the generating poses disturbed by 3 deg / 1 % and written as a scene_error_*.json file first -- the
driver then reads that file exactly as it would read the reference's data/<scene>/scene_error_deg_*_trans_*.json.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffdope as dd  # noqa: E402  (alias of diffdope_amd)
from diffdope_amd import api, bop, synthetic as syn  # noqa: E402


def main():
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    share = bool(os.environ.get("DDX_BENCH_SHARE_GPU"))  # (tests: two ranks on the one GPU of the test box, over gloo)
    torch.cuda.set_device(0 if share else int(os.environ.get("LOCAL_RANK", 0)))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("gloo" if share else "nccl")
    H, W = 240, 320
    intr = syn.camera_intrinsics(W, H)
    cam = dd.Camera(**intr)
    cam.cuda(); cam.set_batchsize(1)
    rng = np.random.RandomState(0)
    meshes, gts, masks, rgb, depth = {}, [], [], None, None
    for k, c in enumerate([(-60.0, -20.0, 420.0), (10.0, 35.0, 400.0), (75.0, -30.0, 440.0)]):
        pos, tri, uv = syn.blob_mesh(24, 32, seed=k)
        meshes[k + 1] = dd.Mesh.from_arrays(pos * 100.0 * 0.6, tri, uv=uv, tex=syn.texture(128, seed=10 + k), scale=0.01)
        q = syn.random_quat(rng)
        obj = dd.Object3D(position=list(c), rotation=list(q), batchsize=1, scale=0.01, mesh=meshes[k + 1])
        obj.cuda(); obj.set_batchsize(1)
        with torch.no_grad():
            r = obj()
            mtx = dd.matrix_batch_44_from_position_quat(p=r["trans"], q=r["quat"])
            o = dd.render_texture_batch(dd.RasterizeGLContext(), cam.cam_proj, mtx, r["pos"], r["pos_idx"], [H, W], uv=r["uv"],
                                        uv_idx=r["uv_idx"], tex=r["tex"], return_rast_out=True)
        cov = (o["rast_out"][0, ..., 3:] > 0).float()
        masks.append(dd.Image(img_tensor=cov.expand(H, W, 3).contiguous().cpu()))
        rgb = o["rgb"][0] if rgb is None else rgb + o["rgb"][0]
        depth = o["depth"][0] * cov[..., 0] if depth is None else depth + o["depth"][0] * cov[..., 0]
        gts.append((np.array(c), q, mtx[0].cpu().numpy()))
    frame = []
    for k, (c, q, _) in enumerate(gts):
        q0, t0 = syn.perturb_pose(q, c, 3.0, 0.01, rng)
        frame.append({"cam_R_m2c": list(api.matrix_from_quat(q0).reshape(-1)), "cam_t_m2c": list(t0), "obj_id": k + 1})
    path = os.path.join(tempfile.gettempdir(), f"scene_error_deg_3_trans_1_rank{rank}.json")
    with open(path, "w") as f:
        json.dump({"0": frame}, f)

    objs = bop.load_scene_poses(path)["0"]
    objs[1]["losses"] = dict(l1_mask=False, l1_edge=True, weight_edge=1.0)   # mixed loss sets per object (config 5)
    objs[2]["losses"] = dict(l1_rgb_with_mask=False)
    scene = dd.Scene(tensor_rgb=dd.Image(img_tensor=rgb.cpu()), tensor_depth=dd.Image(img_tensor=depth.cpu()))
    cfg = dict(losses=dict(l1_rgb_with_mask=True, weight_rgb=0.7, l1_depth_with_mask=True, weight_depth=1.0, l1_mask=True, weight_mask=1.0),
               hyperparameters=dict(nb_iterations=120, batchsize=16, base_lr=0.1, learning_rates_bound=[0.5, 3.0], learning_rate_base=1,
                                    lr_decay=0.1, seed=2))
    table, _ = bop.refine_frame(cfg, dd.Camera(**intr), scene, objs, meshes, masks, rank=rank, world=world, optimizer="adam")
    if rank == 0:
        for k, (_, _, mtx_gt) in enumerate(gts):
            pose = table[k, 2:].reshape(4, 4).cpu().numpy()
            ang = syn.matrix_rotation_geodesic(pose[:3, :3], mtx_gt[:3, :3])
            dt = np.linalg.norm(pose[:3, 3] - mtx_gt[:3, 3]) * 0.1
            print(f"object {k + 1}: best hypothesis {int(table[k, 1])}, loss {float(table[k, 0]):.3e}, "
                  f"rotation error {ang:.2e} rad, translation error {dt:.2e} m (owner rank {bop.owner_of(k, world)})")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
