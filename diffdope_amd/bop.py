"""BOP-style multi-object refinement (the reference's examples/run_bop_scene.py:27-89 and BASELINE config 5):
one frame, several objects, each with its own visible-mask and noisy initial pose, refined independently with
B hypotheses.  Objects are independent, so they shard over ranks (object i -> rank i % world) and ONE
all_reduce at the end gives every rank every object's best pose.

Pose files: the reference's data/*/*/scene_error_deg_*_trans_*.json --
{frame_id: [{"cam_R_m2c": [9 row-major], "cam_t_m2c": [3, millimetres], "obj_id": int}, ...]}.
"""
import json

import numpy as np
import torch

from .api import Camera, DiffDope, Image, Mesh, Object3D, Scene


def load_scene_poses(path):
    """{frame(str): [dict(obj_id, R [3,3] f64, t_mm [3] f64)]} from a scene_error_*.json / BOP scene_gt.json."""
    with open(path) as f:
        raw = json.load(f)
    out = {}
    for frame, objs in raw.items():
        out[str(frame)] = [dict(obj_id=int(o["obj_id"]), R=np.asarray(o["cam_R_m2c"], np.float64).reshape(3, 3),
                                t_mm=np.asarray(o["cam_t_m2c"], np.float64).reshape(3)) for o in objs]
    return out


def owner_of(obj_index, world):
    return obj_index % world


def refine_frame(cfg, camera, scene, objects, meshes, masks, rank=0, world=1, optimizer="sgd", scale=0.01):
    """Refine every object of one frame.

    cfg: config mapping (losses / hyperparameters as configs/diffdope.yaml); camera: Camera; scene: Scene with the
    shared rgb/depth; objects: list of dict(obj_id, R, t_mm[, losses]) (load_scene_poses(...)[frame]; an optional
    per-object "losses" mapping overrides cfg["losses"] entries -- BASELINE config 5's mixed loss sets); meshes: {obj_id: Mesh};
    masks: list of Image (mask_visib of object i).  Returns (table [n_obj,18] float64 tensor identical on every
    rank: loss, arg-min hypothesis, 4x4 pose row-major, and per-object DiffDope handles for the local objects).
    """
    n = len(objects)
    B = cfg["hyperparameters"]["batchsize"]
    dev = torch.device("cuda", torch.cuda.current_device())
    table = torch.zeros((n, 18), dtype=torch.float32, device=dev)
    handles = {}
    # the local objects run on one stream each: every engine is a chain of four latency-bound kernels per iteration, and
    # the kernels of another object fill their launch tails (4 objects of BASELINE config 5: 22.6 -> 20.8 ms per frame)
    main = torch.cuda.current_stream()
    for i, o in enumerate(objects):
        if owner_of(i, world) != rank:
            continue
        obj = Object3D(position=list(o["t_mm"]), rotation=list(np.asarray(o["R"]).reshape(-1)), batchsize=B, scale=scale,
                       mesh=meshes[o["obj_id"]])
        sc = Scene(tensor_rgb=scene.tensor_rgb, tensor_depth=scene.tensor_depth, tensor_segmentation=masks[i])
        cfg_i = cfg if "losses" not in o else {**cfg, "losses": {**cfg["losses"], **o["losses"]}}
        dd = DiffDope(cfg=cfg_i, camera=camera, object3d=obj, scene=sc)
        st = torch.cuda.Stream()
        st.wait_stream(main)
        with torch.cuda.stream(st):
            dd.run_optimization(optimizer=optimizer, wait=False)
        handles[i] = dd
    for i, dd in handles.items():
        dd.finish_optimization()
        best = int(dd.get_argmin())
        stacked = torch.stack([t[-1] for t in dd.losses_values.values()], dim=0).mean(0)
        table[i, 0] = float(stacked[best])
        table[i, 1] = best
        table[i, 2:] = torch.as_tensor(dd.get_pose(best)).reshape(16).to(dev)
    if world > 1:
        from .dist import merge_object_tables

        merge_object_tables(table)  # ONE all_reduce: every row has exactly one non-zero contributor
    return table, handles
