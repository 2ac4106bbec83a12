"""Minimal PLY reader (ASCII and binary little/big endian) -- stands in for trimesh.load(path, force="mesh")
at diffdope/diffdope.py:784 (trimesh is not a dependency of this build).  Reads vertex positions, optional
normals, texture coordinates (per vertex: texture_u/texture_v, s/t or u/v; or per face corner: the face element's
`property list uchar float texcoord` that MeshLab writes for wedge UVs -- vertices are then un-merged per distinct
(vertex, uv) pair, as trimesh does), vertex colours, polygon faces (fan triangulated) and the
`comment TextureFile <name>` header MeshLab/Blender write."""
import os

import numpy as np

_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


def _list_dtype(item_type):
    """numpy dtype list items are held in: the declared item type decides (indices int64, texcoords float64)."""
    return np.float64 if item_type[0] == "f" else np.int64


def read_ply(path):
    """Returns dict(pos [V,3] f32, faces [T,3] i32, normals [V,3]|None, uv [V,2]|None, colors [V,3] u8|None,
    texture_file str|None)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    header = data[:end].decode("ascii", "replace").splitlines()
    body_start = data.find(b"\n", end) + 1
    fmt, elements, texture_file = None, [], None
    for line in header[1:]:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "comment" and len(tok) >= 3 and tok[1].lower() == "texturefile":
            texture_file = " ".join(tok[2:])
        elif tok[0] == "element":
            elements.append(dict(name=tok[1], count=int(tok[2]), props=[]))
        elif tok[0] == "property":
            if tok[1] == "list":
                elements[-1]["props"].append(dict(name=tok[4], list=(_TYPES[tok[2]], _TYPES[tok[3]])))
            else:
                elements[-1]["props"].append(dict(name=tok[2], type=_TYPES[tok[1]]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"{path}: unsupported PLY format {fmt}")
    out = {}
    if fmt == "ascii":
        tokens = data[body_start:].split()
        pos = 0
        for el in elements:
            has_list = any("list" in p for p in el["props"])
            if not has_list:
                n = len(el["props"])
                arr = np.array(tokens[pos:pos + el["count"] * n], dtype=np.float64).reshape(el["count"], n)
                pos += el["count"] * n
                out[el["name"]] = {p["name"]: arr[:, i] for i, p in enumerate(el["props"])}
            else:
                rows = {p["name"]: [] for p in el["props"]}
                for _ in range(el["count"]):
                    for p in el["props"]:
                        if "list" in p:
                            k = int(tokens[pos]); pos += 1
                            rows[p["name"]].append(np.array(tokens[pos:pos + k], dtype=_list_dtype(p["list"][1]))); pos += k
                        else:
                            rows[p["name"]].append(float(tokens[pos])); pos += 1
                out[el["name"]] = rows
    else:
        bo = "<" if fmt == "binary_little_endian" else ">"
        off = body_start
        for el in elements:
            has_list = any("list" in p for p in el["props"])
            if not has_list:
                dt = np.dtype([(p["name"], bo + p["type"]) for p in el["props"]])
                arr = np.frombuffer(data, dtype=dt, count=el["count"], offset=off)
                off += dt.itemsize * el["count"]
                out[el["name"]] = {p["name"]: arr[p["name"]].astype(np.float64) for p in el["props"]}
            else:
                rows = {p["name"]: [] for p in el["props"]}
                for _ in range(el["count"]):
                    for p in el["props"]:
                        if "list" in p:
                            ct, it = np.dtype(bo + p["list"][0]), np.dtype(bo + p["list"][1])
                            k = int(np.frombuffer(data, ct, 1, off)[0]); off += ct.itemsize
                            rows[p["name"]].append(np.frombuffer(data, it, k, off).astype(_list_dtype(p["list"][1]))); off += it.itemsize * k
                        else:
                            t = np.dtype(bo + p["type"])
                            rows[p["name"]].append(float(np.frombuffer(data, t, 1, off)[0])); off += t.itemsize
                out[el["name"]] = rows
    v = out["vertex"]
    pos = np.stack([v["x"], v["y"], v["z"]], 1).astype(np.float32)
    normals = np.stack([v["nx"], v["ny"], v["nz"]], 1).astype(np.float32) if "nx" in v else None
    uv = None
    for a, b in (("texture_u", "texture_v"), ("s", "t"), ("u", "v")):
        if a in v and b in v:
            uv = np.stack([v[a], v[b]], 1).astype(np.float32)
            break
    colors = np.stack([v["red"], v["green"], v["blue"]], 1).astype(np.uint8) if "red" in v else None
    faces = []
    fe = out.get("face", {})
    key = "vertex_indices" if "vertex_indices" in fe else ("vertex_index" if "vertex_index" in fe else None)
    if key is not None:
        for poly in fe[key]:
            for i in range(1, len(poly) - 1):  # fan triangulation
                faces.append((poly[0], poly[i], poly[i + 1]))
    faces = np.array(faces, dtype=np.int32).reshape(-1, 3)
    # per-face-corner ("wedge") texture coordinates: un-merge the vertices, one per distinct (vertex, u, v)
    tkey = next((k for k in ("texcoord", "texcoords", "uv") if k in fe), None)
    if tkey is not None and key is not None and len(faces):
        corner_v, corner_uv = [], []
        for poly, tc in zip(fe[key], fe[tkey]):
            tc = np.asarray(tc, np.float64).reshape(-1, 2)
            if len(tc) != len(poly):
                raise ValueError(f"{path}: face with {len(poly)} vertices carries {len(tc)} texture coordinates")
            for i in range(1, len(poly) - 1):
                for c in (0, i, i + 1):
                    corner_v.append(int(poly[c]))
                    corner_uv.append(tc[c])
        corner_v = np.asarray(corner_v, np.int64)
        corner_uv = np.asarray(corner_uv, np.float32).reshape(-1, 2)
        rec = np.concatenate([corner_v[:, None].astype(np.float64), corner_uv.astype(np.float64)], 1)
        uniq, inverse = np.unique(rec, axis=0, return_inverse=True)
        src = uniq[:, 0].astype(np.int64)
        pos, uv = pos[src], uniq[:, 1:].astype(np.float32)
        normals = None if normals is None else normals[src]
        colors = None if colors is None else colors[src]
        faces = np.asarray(inverse, np.int32).reshape(-1, 3)
    if texture_file is not None:
        texture_file = os.path.join(os.path.dirname(os.path.abspath(path)), texture_file)
    return dict(pos=pos, faces=faces, normals=normals, uv=uv, colors=colors, texture_file=texture_file)


def vertex_normals(pos, faces):
    """Area-weighted vertex normals (what trimesh derives when the file carries none)."""
    n = np.zeros_like(pos, dtype=np.float64)
    fn = np.cross(pos[faces[:, 1]] - pos[faces[:, 0]], pos[faces[:, 2]] - pos[faces[:, 0]])
    for k in range(3):
        np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return (n / np.maximum(ln, 1e-20)).astype(np.float32)
