"""Randomised round trip of the PLY reader (diffdope_amd/io_ply.py; SURVEY 8(f2): meshes without trimesh): files written by an
independent writer below in the three encodings, with random property sets, orders and scalar types, polygons of 3..5 corners,
optional per-corner texture coordinates, CRLF headers, comment / obj_info lines and unknown extra properties and elements."""
import struct

import numpy as np

from diffdope_amd.io_ply import read_ply

_FMT = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d"}


def _write(path, rng, enc, pos, polys, normals, uv, colors, wedge):
    ftype = lambda: str(rng.choice(["float", "double"]))
    vprops = [("x", ftype()), ("y", ftype()), ("z", ftype())]
    cols = {"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2]}
    if normals is not None:
        vprops += [("nx", ftype()), ("ny", ftype()), ("nz", ftype())]
        cols.update(nx=normals[:, 0], ny=normals[:, 1], nz=normals[:, 2])
    if uv is not None:
        a, b = [("texture_u", "texture_v"), ("s", "t"), ("u", "v")][rng.randint(3)]
        vprops += [(a, ftype()), (b, ftype())]
        cols[a], cols[b] = uv[:, 0], uv[:, 1]
    if colors is not None:
        vprops += [("red", "uchar"), ("green", "uchar"), ("blue", "uchar")]
        cols.update(red=colors[:, 0], green=colors[:, 1], blue=colors[:, 2])
        if rng.rand() < 0.5:
            vprops.append(("alpha", "uchar")); cols["alpha"] = np.full(len(pos), 255)
    if rng.rand() < 0.3:
        vprops.append(("quality", "float")); cols["quality"] = rng.rand(len(pos))
    head, tail = vprops[:3], vprops[3:]
    rng.shuffle(tail)  # (x, y, z stay first as every exporter writes them; the rest in any order)
    vprops = head + tail
    ct, it = str(rng.choice(["uchar", "ushort", "int"])), str(rng.choice(["int", "uint", "short" if len(pos) < 30000 else "int"]))
    idx_name = str(rng.choice(["vertex_indices", "vertex_index"]))
    fprops = [("list", ct, it, idx_name)]
    if wedge is not None:
        fprops.append(("list", "uchar", "float", "texcoord"))
    if rng.rand() < 0.3:
        fprops.append(("scalar", "int", None, "flags"))
    rng.shuffle(fprops)
    nl = "\r\n" if rng.rand() < 0.3 else "\n"
    hdr = ["ply", f"format {enc} 1.0", "comment written by the fuzz test"]
    if rng.rand() < 0.5:
        hdr.append("comment TextureFile tex.png")
    if rng.rand() < 0.3:
        hdr.append("obj_info something 1 2 3")
    extra_first = rng.rand() < 0.2
    if extra_first:
        hdr += ["element camera 1", "property float view_px"]
    hdr.append(f"element vertex {len(pos)}")
    hdr += [f"property {t} {n}" for n, t in vprops]
    hdr.append(f"element face {len(polys)}")
    for p in fprops:
        hdr.append(f"property list {p[1]} {p[2]} {p[3]}" if p[0] == "list" else f"property {p[1]} {p[3]}")
    hdr.append("end_header")
    with open(path, "wb") as f:
        f.write((nl.join(hdr) + nl).encode("ascii"))
        if enc == "ascii":
            lines = []
            if extra_first:
                lines.append("0.5")
            for i in range(len(pos)):
                lines.append(" ".join(str(int(cols[n][i])) if t in ("uchar",) else repr(float(cols[n][i])) for n, t in vprops))
            for k, poly in enumerate(polys):
                parts = []
                for p in fprops:
                    if p[0] == "scalar":
                        parts.append("7")
                    elif p[3] == "texcoord":
                        parts.append(str(2 * len(poly)) + " " + " ".join(repr(float(x)) for x in wedge[k].reshape(-1)))
                    else:
                        parts.append(str(len(poly)) + " " + " ".join(str(int(v)) for v in poly))
                lines.append(" ".join(parts))
            f.write(("\n".join(lines) + "\n").encode("ascii"))
        else:
            bo = "<" if enc == "binary_little_endian" else ">"
            if extra_first:
                f.write(struct.pack(bo + "f", 0.5))
            for i in range(len(pos)):
                for n, t in vprops:
                    f.write(struct.pack(bo + _FMT[t], int(cols[n][i]) if t == "uchar" else float(cols[n][i])))
            for k, poly in enumerate(polys):
                for p in fprops:
                    if p[0] == "scalar":
                        f.write(struct.pack(bo + "i", 7))
                    elif p[3] == "texcoord":
                        f.write(struct.pack(bo + "B", 2 * len(poly)))
                        f.write(struct.pack(bo + "f" * (2 * len(poly)), *[float(x) for x in wedge[k].reshape(-1)]))
                    else:
                        f.write(struct.pack(bo + _FMT[p[1]], len(poly)))
                        f.write(struct.pack(bo + _FMT[p[2]] * len(poly), *[int(v) for v in poly]))
    return vprops


def test_ply_reader_round_trips_random_files(tmp_path):
    n_wedge = 0
    for case in range(150):
        rng = np.random.RandomState(9000 + case)
        V = int(rng.randint(5, 60))
        pos = rng.normal(size=(V, 3)).astype(np.float32)
        normals = rng.normal(size=(V, 3)).astype(np.float32) if rng.rand() < 0.5 else None
        uv = rng.uniform(size=(V, 2)).astype(np.float32) if rng.rand() < 0.5 else None
        colors = rng.randint(0, 256, size=(V, 3)).astype(np.uint8) if rng.rand() < 0.5 else None
        polys = [rng.choice(V, size=int(rng.randint(3, 6)), replace=False) for _ in range(int(rng.randint(1, 40)))]
        wedge = [rng.uniform(size=(len(p), 2)).astype(np.float32) for p in polys] if rng.rand() < 0.3 else None
        enc = ["ascii", "binary_little_endian", "binary_big_endian"][case % 3]
        path = str(tmp_path / f"m{case}.ply")
        vprops = _write(path, rng, enc, pos, polys, normals, uv, colors, wedge)
        m = read_ply(path)
        tris = np.array([(p[0], p[i], p[i + 1]) for p in polys for i in range(1, len(p) - 1)], np.int64)
        assert m["faces"].shape == tris.shape and m["faces"].dtype == np.int32
        double_pos = dict(vprops)["x"] == "double"  # (float32 data written as double or float reads back exactly either way)
        if wedge is None:
            np.testing.assert_array_equal(m["pos"], pos)
            np.testing.assert_array_equal(m["faces"], tris)
            for key, ref in (("normals", normals), ("uv", uv)):
                assert (m[key] is None) == (ref is None)
                if ref is not None:
                    np.testing.assert_array_equal(m[key], ref)
            assert (m["colors"] is None) == (colors is None)
            if colors is not None:
                np.testing.assert_array_equal(m["colors"], colors)
        else:
            # un-merged per corner: every corner keeps its position and its own texture coordinate
            n_wedge += 1
            corner_pos = pos[tris.reshape(-1)]
            corner_uv = np.concatenate([w[[0, i, i + 1]] for w in wedge for i in range(1, len(w) - 1)])
            np.testing.assert_array_equal(m["pos"][m["faces"].reshape(-1)], corner_pos)
            np.testing.assert_array_equal(m["uv"][m["faces"].reshape(-1)], corner_uv)
            assert len(m["pos"]) <= 3 * len(tris) and len(np.unique(np.concatenate([m["pos"], m["uv"]], 1), axis=0)) == len(m["pos"])
            if colors is not None:
                np.testing.assert_array_equal(m["colors"][m["faces"].reshape(-1)], colors[tris.reshape(-1)])
        assert double_pos in (True, False)
    assert n_wedge > 20
