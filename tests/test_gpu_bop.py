"""GPU test of the multi-object driver (BASELINE config 5 in miniature): three objects in one frame, each with
its own visible mask and a noisy initial pose in the BOP json format, refined with 8 hypotheses each."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_three_objects_one_frame(tmp_path):
    import diffdope_amd as dd
    from diffdope_amd import api, bop, synthetic as syn

    H, W = 120, 160
    intr = syn.camera_intrinsics(W, H)
    cam = dd.Camera(**intr)
    cam.cuda(); cam.set_batchsize(1)
    rng = np.random.RandomState(0)
    meshes, gts, rgb_sum, depth_sum, masks = {}, [], None, None, []
    centers = [(-60.0, -20.0, 420.0), (10.0, 35.0, 400.0), (75.0, -30.0, 440.0)]
    for k, c in enumerate(centers):
        pos, tri, uv = syn.blob_mesh(16, 24, seed=k)
        meshes[k + 1] = dd.Mesh.from_arrays(pos * 100.0 * 0.6, tri, uv=uv, tex=syn.texture(64, seed=10 + k), scale=0.01)
        q = syn.random_quat(rng)
        gts.append((np.array(c), q))
        obj = dd.Object3D(position=list(c), rotation=list(q), batchsize=1, scale=0.01, mesh=meshes[k + 1])
        obj.cuda(); obj.set_batchsize(1)
        with torch.no_grad():
            r = obj()
            mtx = dd.matrix_batch_44_from_position_quat(p=r["trans"], q=r["quat"])
            o = dd.render_texture_batch(dd.RasterizeGLContext(), cam.cam_proj, mtx, r["pos"], r["pos_idx"], [H, W], uv=r["uv"],
                                        uv_idx=r["uv_idx"], tex=r["tex"], return_rast_out=True)
        cov = (o["rast_out"][0, ..., 3:] > 0).float()
        masks.append(dd.Image(img_tensor=cov.expand(H, W, 3).contiguous().cpu()))
        rgb_sum = o["rgb"][0] if rgb_sum is None else rgb_sum + o["rgb"][0]
        depth_sum = o["depth"][0] * cov[..., 0] if depth_sum is None else depth_sum + o["depth"][0] * cov[..., 0]
        gts[-1] = gts[-1] + (mtx[0].cpu().numpy(),)
    # noisy initial poses in the reference's json format
    frame = []
    for k, (c, q, _) in enumerate(gts):
        q0, t0 = syn.perturb_pose(q, c, 3.0, 0.01, rng)
        frame.append({"cam_R_m2c": list(api.matrix_from_quat(q0).reshape(-1)), "cam_t_m2c": list(t0), "obj_id": k + 1})
    with open(tmp_path / "scene_error.json", "w") as f:
        json.dump({"0": frame}, f)
    objs = bop.load_scene_poses(str(tmp_path / "scene_error.json"))["0"]
    assert len(objs) == 3 and objs[1]["obj_id"] == 2 and objs[0]["R"].shape == (3, 3)
    scene = dd.Scene(tensor_rgb=dd.Image(img_tensor=rgb_sum.cpu()), tensor_depth=dd.Image(img_tensor=depth_sum.cpu()))
    cfg = dict(losses=dict(l1_rgb_with_mask=True, weight_rgb=0.7, l1_depth_with_mask=True, weight_depth=1.0, l1_mask=True, weight_mask=1.0),
               hyperparameters=dict(nb_iterations=120, batchsize=8, base_lr=0.1, learning_rates_bound=[0.5, 3.0], learning_rate_base=1,
                                    lr_decay=0.1, seed=2))
    # mixed per-object loss sets (BASELINE config 5): object 2 rgb+depth+edge (extension), object 3 depth+mask
    objs[1]["losses"] = dict(l1_mask=False, l1_edge=True, weight_edge=1.0)
    objs[2]["losses"] = dict(l1_rgb_with_mask=False)
    cam2 = dd.Camera(**intr)
    table, handles = bop.refine_frame(cfg, cam2, scene, objs, meshes, masks, optimizer="adam")
    assert tuple(table.shape) == (3, 18) and len(handles) == 3
    assert set(handles[0].losses_values) == {"rgb", "depth", "mask_selection"}
    assert set(handles[1].losses_values) == {"rgb", "depth", "edge"} and set(handles[2].losses_values) == {"depth", "mask_selection"}
    for k, (_, _, mtx_gt) in enumerate(gts):
        pose = table[k, 2:].reshape(4, 4).cpu().numpy()
        ang = syn.matrix_rotation_geodesic(pose[:3, :3], mtx_gt[:3, :3])
        dt = np.linalg.norm(pose[:3, 3] - mtx_gt[:3, 3]) * 0.1
        assert ang < 1e-2 and dt < 3e-3, (k, ang, dt)
