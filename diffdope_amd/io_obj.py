"""Wavefront OBJ reader (numpy only) -- the reference loads any mesh format through trimesh.load(path, force="mesh")
(diffdope/diffdope.py:784); PLY (io_ply.py) covers its own data set, OBJ is what HOPE / YCB object models ship as.

Read: `v x y z [r g b]`, `vt u v [w]`, `vn x y z`, `f` with the four corner forms (v, v/vt, v//vn, v/vt/vn), 1-based and
negative (relative) indices, polygons (fan triangulation), `mtllib` -> the first `map_Kd` of the material library as the
texture file.  Corners that share a position but not a texture coordinate become separate vertices (what trimesh does), so the
result has one uv per vertex like io_ply.read_ply.  Groups, smoothing groups, material switches and lines are ignored: the
renderer takes one texture per object, as the reference does."""
import os

import numpy as np


def _material_texture(mtl_path):
    try:
        with open(mtl_path, "r", errors="replace") as f:
            for line in f:
                tok = line.split()
                if tok and tok[0].lower() == "map_kd":
                    name = line.split(None, 1)[1].strip().split()[-1]  # (options such as -s 1 1 1 come before the file name)
                    return os.path.join(os.path.dirname(os.path.abspath(mtl_path)), name)
    except OSError:
        pass
    return None


def read_obj(path):
    """Returns dict(pos [V,3] f32, faces [T,3] i32, normals [V,3]|None, uv [V,2]|None, colors [V,3] u8|None,
    texture_file str|None) -- the layout of io_ply.read_ply."""
    v, vt, vn, vc = [], [], [], []
    polys = []  # per face: list of (vi, ti, ni), 0-based, -1 = absent
    mtllib = None
    with open(path, "r", errors="replace") as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            key = tok[0]
            if key == "v":
                v.append((float(tok[1]), float(tok[2]), float(tok[3])))
                if len(tok) >= 7:
                    vc.append((float(tok[4]), float(tok[5]), float(tok[6])))
            elif key == "vt":
                vt.append((float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0))
            elif key == "vn":
                vn.append((float(tok[1]), float(tok[2]), float(tok[3])))
            elif key == "f":
                corners = []
                for c in tok[1:]:
                    parts = c.split("/")
                    idx = [-1, -1, -1]
                    for i, (p, n) in enumerate(zip(parts[:3], (len(v), len(vt), len(vn)))):
                        if p:
                            k = int(p)
                            idx[i] = k - 1 if k > 0 else n + k
                    corners.append(tuple(idx))
                if len(corners) >= 3:
                    polys.append(corners)
            elif key == "mtllib":
                mtllib = line.split(None, 1)[1].strip()
    if not v:
        raise ValueError(f"{path}: no vertices")
    pos = np.asarray(v, np.float32)
    n_v = len(pos)
    tri = []
    for corners in polys:
        for i in range(1, len(corners) - 1):
            tri.append((corners[0], corners[i], corners[i + 1]))
    tri = np.asarray(tri, np.int64).reshape(-1, 3, 3)  # [T, corner, (v, vt, vn)]
    if tri.size and (tri[..., 0].min() < 0 or tri[..., 0].max() >= n_v):
        raise ValueError(f"{path}: face refers to a vertex that does not exist")
    colors = None
    if len(vc) == n_v:
        c = np.asarray(vc, np.float64)
        colors = np.clip(np.rint(c * (255.0 if c.max() <= 1.0 else 1.0)), 0, 255).astype(np.uint8)
    normals = None
    uv = None
    faces = tri[..., 0].astype(np.int32)
    has_vt = len(vt) > 0 and tri.size and (tri[..., 1] >= 0).all() and tri[..., 1].max() < len(vt)
    has_vn = len(vn) > 0 and tri.size and (tri[..., 2] >= 0).all() and tri[..., 2].max() < len(vn)
    if has_vt:
        # one vertex per distinct (position, texture coordinate) pair
        pairs = tri[..., :2].reshape(-1, 2)
        uniq, inverse = np.unique(pairs, axis=0, return_inverse=True)
        src = uniq[:, 0]
        pos = pos[src]
        uv = np.asarray(vt, np.float32)[uniq[:, 1]]
        colors = None if colors is None else colors[src]
        faces = np.asarray(inverse, np.int32).reshape(-1, 3)
        if has_vn:  # the normal of the first corner that produced the vertex
            first = np.full(len(uniq), -1, np.int64)
            flat_n = tri[..., 2].reshape(-1)
            inv = np.asarray(inverse).reshape(-1)
            order = np.arange(len(inv))[::-1]
            first[inv[order]] = flat_n[order]
            normals = np.asarray(vn, np.float32)[first]
    elif has_vn and len(vn) == n_v and (tri[..., 2] == tri[..., 0]).all():
        normals = np.asarray(vn, np.float32)
    texture_file = None
    if mtllib is not None:
        texture_file = _material_texture(os.path.join(os.path.dirname(os.path.abspath(path)), mtllib))
    return dict(pos=pos, faces=faces, normals=normals, uv=uv, colors=colors, texture_file=texture_file)
