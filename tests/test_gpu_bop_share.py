"""BASELINE configs[4], one GPU's share: 32 objects x 64 hypotheses at 1280x720 over 8 GPUs = FOUR different meshes x 64 hypotheses per
GPU, refined through bop.refine_frame (the reference's examples/run_bop_scene.py:48-89 flow) with mixed rgb / depth / edge / mask
loss sets.  Checked here: the frame as ONE engine group is bit-identical to one stream per object and to four engines run one
after the other; and every object's engine,
as refine_frame built it (mesh of 20 480 triangles, 2048-px frame crop semantics of the API, its own mask and loss set), against the
oracle on two hypotheses (losses rtol 5e-5, pose gradient 1e-4 of its largest component) with duplicated hypotheses bit-identical."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEYS = ("rgb", "depth", "mask_selection", "edge")


def _frame(tmp_path, H, W, rows, cols, tex_size):
    import diffdope_amd as dd
    from diffdope_amd import api, bop, synthetic as syn

    intr = syn.camera_intrinsics(W, H)
    cam = dd.Camera(**intr)
    cam.cuda(); cam.set_batchsize(1)
    rng = np.random.RandomState(7)
    meshes, gts, masks, rgb, depth = {}, [], [], None, None
    centers = [(-150.0, -60.0, 750.0), (150.0, -55.0, 760.0), (-145.0, 85.0, 740.0), (155.0, 80.0, 755.0)]
    for k, c in enumerate(centers):
        pos, tri, uv = syn.blob_mesh(rows, cols, seed=20 + k)
        meshes[k + 1] = dd.Mesh.from_arrays(pos * 100.0, tri, uv=uv, tex=syn.texture(tex_size, seed=30 + k), scale=0.01)
        q = syn.random_quat(rng)
        obj = dd.Object3D(position=list(c), rotation=list(q), batchsize=1, scale=0.01, mesh=meshes[k + 1])
        obj.cuda(); obj.set_batchsize(1)
        with torch.no_grad():
            r = obj()
            mtx = dd.matrix_batch_44_from_position_quat(p=r["trans"], q=r["quat"])
            o = dd.render_texture_batch(dd.RasterizeGLContext(), cam.cam_proj, mtx, r["pos"], r["pos_idx"], [H, W], uv=r["uv"],
                                        uv_idx=r["uv_idx"], tex=r["tex"], return_rast_out=True)
        cov = (o["rast_out"][0, ..., 3:] > 0).float()
        assert 0.002 < float(cov.mean()) < 0.05  # the object is in the frame, a BOP-crop-sized blob
        masks.append(dd.Image(img_tensor=cov.expand(H, W, 3).contiguous().cpu()))
        rgb = o["rgb"][0] if rgb is None else rgb + o["rgb"][0]
        depth = o["depth"][0] * cov[..., 0] if depth is None else depth + o["depth"][0] * cov[..., 0]
        gts.append((np.array(c), q))
    frame = []
    for k, (c, q) in enumerate(gts):
        q0, t0 = syn.perturb_pose(q, c, 4.0, 0.01, rng)
        frame.append({"cam_R_m2c": list(api.matrix_from_quat(q0).reshape(-1)), "cam_t_m2c": list(t0), "obj_id": k + 1})
    path = tmp_path / "scene_error_deg_4_trans_1.json"
    with open(path, "w") as f:
        json.dump({"0": frame}, f)
    objs = bop.load_scene_poses(str(path))["0"]
    # mixed loss sets (config 5: "mixed rgb/depth/edge losses"; the edge term is this build's extension)
    objs[0]["losses"] = dict(l1_mask=False, l1_edge=True, weight_edge=1.0)                    # rgb + depth + edge
    objs[1]["losses"] = dict()                                                              # rgb + depth + mask
    objs[2]["losses"] = dict(l1_rgb_with_mask=False)                                         # depth + mask
    objs[3]["losses"] = dict(l1_depth_with_mask=False, l1_mask=False, l1_edge=True, weight_edge=0.5)  # rgb + edge
    scene = dd.Scene(tensor_rgb=dd.Image(img_tensor=rgb.cpu()), tensor_depth=dd.Image(img_tensor=depth.cpu()))
    return intr, meshes, masks, objs, scene


@pytest.mark.parametrize("H,W,rows,cols,B", [(720, 1280, 80, 128, 64)])
def test_config5_share_four_objects_64_hypotheses_1280x720(tmp_path, H, W, rows, cols, B):
    import diffdope_amd as dd
    from diffdope_amd import bop
    from oracle import oracle as orc

    intr, meshes, masks, objs, scene = _frame(tmp_path, H, W, rows, cols, 512)
    n_it = 3
    cfg = dict(losses=dict(l1_rgb_with_mask=True, weight_rgb=0.7, l1_depth_with_mask=True, weight_depth=1.0, l1_mask=True, weight_mask=1.0),
               hyperparameters=dict(nb_iterations=n_it, batchsize=B, base_lr=0.05, learning_rates_bound=[0.5, 3.0], learning_rate_base=1,
                                    lr_decay=0.1, seed=5))
    # ---- the four local objects as ONE engine group (one launch of each kernel per iteration)
    table, handles = bop.refine_frame(cfg, dd.Camera(**intr), scene, objs, meshes, masks, optimizer="sgd", mode="group")
    # ... one stream per object gives the same bits
    table_s, handles_s = bop.refine_frame(cfg, dd.Camera(**intr), scene, objs, meshes, masks, optimizer="sgd", mode="streams")
    assert torch.equal(table, table_s)
    for i in range(4):
        assert torch.equal(handles[i].last_engine.mtx_log, handles_s[i].last_engine.mtx_log)
        assert torch.equal(handles[i].object3d.params_tensor(), handles_s[i].object3d.params_tensor())
    del handles_s
    assert tuple(table.shape) == (4, 18) and sorted(handles) == [0, 1, 2, 3]
    want = [{"rgb", "depth", "edge"}, {"rgb", "depth", "mask_selection"}, {"depth", "mask_selection"}, {"rgb", "edge"}]
    for i in range(4):
        h = handles[i]
        assert set(h.losses_values) == want[i]
        e = h.last_engine
        assert (e.B, e.H, e.W, e.desc.T, e.desc.B_global) == (B, H, W, 2 * rows * cols, B)
        assert e.status()["overflow"] == 0 and e.status()["active_tiles"] > 0
        assert all(torch.isfinite(v).all() for v in h.losses_values.values())
    # ---- ... and is bit-identical to four engines run one after the other on the default stream
    e0 = handles[0].last_engine
    ss, es = int(e0.desc.shade_slices), int(e0.desc.edge_slices)
    for i, o in enumerate(objs):
        obj = dd.Object3D(position=list(o["t_mm"]), rotation=list(np.asarray(o["R"]).reshape(-1)), batchsize=B, scale=0.01, mesh=meshes[o["obj_id"]])
        sc = dd.Scene(tensor_rgb=scene.tensor_rgb, tensor_depth=scene.tensor_depth, tensor_segmentation=masks[i])
        seq = dd.DiffDope(cfg={**cfg, "losses": {**cfg["losses"], **o["losses"]}}, camera=dd.Camera(**intr), object3d=obj, scene=sc)
        p0 = seq.object3d.params_tensor().clone()
        seq.prepare_optimization(optimizer="sgd", shade_slices=ss, edge_slices=es).run()
        seq.finish_optimization()
        h = handles[i]
        assert set(seq.losses_values) == set(h.losses_values)
        for k in seq.losses_values:
            assert torch.equal(seq.losses_values[k], h.losses_values[k]), (i, k)
        assert torch.equal(seq.object3d.params_tensor(), h.object3d.params_tensor())
        assert torch.equal(seq.last_engine.mtx_log, h.last_engine.mtx_log)
        best = int(seq.get_argmin())
        assert best == int(table[i, 1]) and np.array_equal(seq.get_pose(best).reshape(16), table[i, 2:].cpu().numpy())
        # ---- the engine of this object against the oracle: two hypotheses, and a duplicated one
        e = seq.last_engine
        lrm = e.lr_mult.clone()
        p = p0.clone()
        rng = np.random.RandomState(100 + i)
        p += torch.tensor(rng.normal(scale=[[0.01]] * 4 + [[0.004]] * 3, size=(7, B)), dtype=torch.float32, device=p.device)  # distinct hypotheses
        p[:, B - 1] = p[:, 1]
        lrm[B - 1] = lrm[1]
        e.new_observation(params=p, lr_mult=lrm)
        losses, grad = e.loss_and_grad()
        torch.cuda.synchronize()
        e.check()
        lg, g = losses.cpu().numpy(), grad.cpu().numpy()
        assert np.array_equal(lg[:, 1], lg[:, B - 1]) and np.array_equal(g[:, 1], g[:, B - 1])
        npy = lambda t: t.detach().cpu().numpy()
        r = seq.object3d.mesh()
        lw = seq.cfg.losses
        wts = dict(rgb=lw.weight_rgb if lw.l1_rgb_with_mask else None, depth=lw.weight_depth if lw.l1_depth_with_mask else None,
                   mask=lw.weight_mask if lw.l1_mask else None, edge=lw.get("weight_edge", 1.0) if lw.get("l1_edge", False) else None)
        R = orc.RenderOracle(npy(r["pos"][0]), npy(r["pos_idx"][0]), npy(seq.camera.cam_proj[0]), H, W, {k: npy(v[0]) for k, v in seq.gt_tensors.items()},
                             wts, dtype=np.float32, cull_backfaces=not e.desc.no_backface_cull, uv=npy(r["uv"][0]), tex=npy(r["tex"][0]))
        pn, ln = npy(p), npy(lrm)
        for b in (0, B // 2):
            total, logs, g_ref, _ = R.loss_and_grad(pn[:, b:b + 1], ln[b:b + 1], global_B=B)
            for j, key in enumerate(KEYS):
                if key in logs:
                    np.testing.assert_allclose(lg[j, b], logs[key][0], rtol=5e-5, atol=1e-7, err_msg=f"object {i} hypothesis {b} {key}")
                else:
                    assert lg[j, b] == 0
            np.testing.assert_allclose(g[:, b], g_ref[:, 0], rtol=1e-4, atol=1e-4 * np.abs(g_ref).max(), err_msg=f"object {i} hypothesis {b}")
