"""__graft_entry__.smoke(): one tiny pass of the hot path on cuda:0, checked against the CPU oracle."""
import numpy as np
import torch


def run():
    import diffdope_amd as dd
    from diffdope_amd import synthetic as syn
    from oracle import oracle as orc  # the checker; the product path above never touches it

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda:0")
    H, W, B = 60, 80, 3
    pos, tri, uv = syn.blob_mesh(16, 20, seed=0)
    tex = syn.texture(32, seed=1)
    proj = orc.projection_matrix(**syn.camera_intrinsics(W, H)).astype(np.float32)
    rng = np.random.RandomState(2)
    q_gt, t_gt = syn.random_quat(rng), np.array([0.09, -0.05, -1.8])
    weights = dict(rgb=0.7, depth=1.0, mask=1.0, edge=0.5)
    R = orc.RenderOracle(pos, tri, proj, H, W, {}, weights, uv=uv, tex=tex, dtype=np.float32, cull_backfaces=False)  # (both faces: dr.rasterize's rule, the engine's default)
    r = R.render(orc.pose_fwd(np.concatenate([q_gt, t_gt])[:, None].astype(np.float32)))
    cov = r["rast"][0, ..., 3] > 0
    gt = dict(rgb=r["rgb"][0], depth=r["depth"][0], segmentation=np.repeat(cov[..., None], 3, -1).astype(np.float32))
    R.gt = {k: v[None] for k, v in gt.items()}
    params = np.stack([np.concatenate(syn.perturb_pose(q_gt, t_gt, 5.0, 0.02, rng)) for _ in range(B)], 1).astype(np.float32)
    lr_mult = np.array([0.5, 1.0, 2.0], np.float32)
    _, logs, g_ref, r_ref = R.loss_and_grad(params, lr_mult)

    T = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    # 1. op-level: transform + rasterize, triangle ids must be bit-identical
    final = torch.matmul(T(proj)[None], T(orc.pose_fwd(params)))
    clip = dd.xfm_points(T(pos)[None].expand(B, -1, -1).contiguous(), final)
    rast, _ = dd.rasterize(dd.RasterizeGLContext(), clip, T(tri), [H, W])
    assert np.array_equal(rast[..., 3].cpu().numpy(), r_ref["rast"][..., 3]), "triangle ids differ from the oracle"
    # 2. fused engine: one iteration forward + backward + SGD step
    p = T(params)
    eng = dd.RefineEngine(T(pos), T(tri), T(proj), [H, W], {k: T(v) for k, v in gt.items()}, p, T(lr_mult), [0.5], weights,
                          uv=T(uv), tex=T(tex))
    eng.run()
    torch.cuda.synchronize()
    eng.check()
    g = (params - p.cpu().numpy()) / 0.5
    np.testing.assert_allclose(g, g_ref, rtol=2e-3, atol=2e-3 * np.abs(g_ref).max())
    lg = eng.losses()[0].cpu().numpy()
    for i, k in enumerate(("rgb", "depth", "mask_selection", "edge")):
        np.testing.assert_allclose(lg[i], logs[k], rtol=2e-5, atol=1e-7)
    print("smoke ok: ids bit-identical, losses and pose gradients match the oracle; max |grad| =", float(np.abs(g).max()))
