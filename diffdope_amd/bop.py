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


def refine_frame(cfg, camera, scene, objects, meshes, masks, rank=0, world=1, optimizer="sgd", scale=0.01, mode="streams"):
    """Refine every object of one frame.

    cfg: config mapping (losses / hyperparameters as configs/diffdope.yaml); camera: Camera; scene: Scene with the
    shared rgb/depth; objects: list of dict(obj_id, R, t_mm[, losses]) (load_scene_poses(...)[frame]; an optional
    per-object "losses" mapping overrides cfg["losses"] entries -- BASELINE config 5's mixed loss sets); meshes: {obj_id: Mesh};
    masks: list of Image (mask_visib of object i).  Returns (table [n_obj,18] float64 tensor identical on every
    rank: loss, arg-min hypothesis, 4x4 pose row-major, and per-object DiffDope handles for the local objects).

    mode: how the local objects (4 per GPU in config 5) share the GPU.  "streams" (default): one HIP stream per object (every
    engine a chain of latency-bound launches -- two chains of half-batch launches each since the end of round 4 --, the other
    objects' kernels fill its gaps); "group": ONE engine group -- one launch of each kernel per iteration for all of them
    (RefineEngineGroup; a 64-hypothesis launch is latency-bound and fills a fraction of the chip); "sequential": one after the
    other.  End of round 4, config 5's share, streams / group / sequential: 15.7-15.8 / 15.9-16.4 / 17.1 ms per 58 iterations; four
    cfg2-sized objects: 7.3 / 7.2 / 9.1 ms (tools/multi_object_streams.py).  Whatever the mode, an object's result is the same bits."""
    n = len(objects)
    B = cfg["hyperparameters"]["batchsize"]
    dev = torch.device("cuda", torch.cuda.current_device())
    table = torch.zeros((n, 18), dtype=torch.float32, device=dev)
    handles = {}
    local = [i for i in range(n) if owner_of(i, world) == rank]
    # slices of the shading / edge launches per hypothesis: every engine's own choice from ITS batch size -- not from how many
    # objects happen to share this GPU -- so that an object's result does not depend on the number of ranks (the gradient sum is
    # grouped by slices).  (Fewer slices for a fuller GPU measured 5 % faster on 4 x 64 hypotheses at 640x480, equal at 1280x720.)
    ss, es = int(cfg["hyperparameters"].get("shade_slices", 0)), int(cfg["hyperparameters"].get("edge_slices", 0))
    main = torch.cuda.current_stream()
    engines = []
    for i in local:
        o = objects[i]
        obj = Object3D(position=list(o["t_mm"]), rotation=list(np.asarray(o["R"]).reshape(-1)), batchsize=B, scale=scale,
                       mesh=meshes[o["obj_id"]])
        sc = Scene(tensor_rgb=scene.tensor_rgb, tensor_depth=scene.tensor_depth, tensor_segmentation=masks[i])
        cfg_i = cfg if "losses" not in o else {**cfg, "losses": {**cfg["losses"], **o["losses"]}}
        dd = DiffDope(cfg=cfg_i, camera=camera, object3d=obj, scene=sc)
        handles[i] = dd
        if mode == "group":
            engines.append(dd.prepare_optimization(optimizer=optimizer, shade_slices=ss, edge_slices=es))
        elif mode == "streams":
            # every engine is a chain of latency-bound kernels per iteration, and the kernels of another object fill their launch tails
            st = torch.cuda.Stream()
            st.wait_stream(main)
            with torch.cuda.stream(st):
                eng = dd.prepare_optimization(optimizer=optimizer, shade_slices=ss, edge_slices=es)
                eng.run()
        elif mode == "sequential":
            dd.prepare_optimization(optimizer=optimizer, shade_slices=ss, edge_slices=es).run()
        else:
            raise ValueError(f"refine_frame: unknown mode {mode!r}")
    if mode == "group" and engines:
        from .engine import RefineEngineGroup

        group = RefineEngineGroup(engines)
        group.run()
        group.finish()  # (synchronises and validates: a run whose in-launch tile pass timed out is repeated here, by the group)
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
