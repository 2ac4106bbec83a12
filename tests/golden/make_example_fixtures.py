#!/usr/bin/env python
"""Derive the small real-data fixtures under tests/golden/example/ from the reference's example scene
(/root/reference/data/example + configs/diffdope.yaml: BASELINE configs[0], the AlphabetSoup HOPE object).
Run in the build container only (the reference tree does not travel); DATA only -- no reference source text.

    python tests/golden/make_example_fixtures.py

Writes
  example/mesh/AlphabetSoup.ply   the full mesh (8 240 vertices with normals + texture_u/v, 13 860 faces), re-encoded as
                                  binary little-endian PLY from the values diffdope_amd.io_ply parses out of the reference's ASCII file
  example/mesh/AlphabetSoup.png   the 2048x2048 texture reduced to 512x512 (box filter)
  example/scene/{rgb,depth,seg}.png  a 640x360 window of the 1920x1080 observation (columns 358..997, rows 714..1073, around the
                                  object; pixels untouched: 8-bit rgb, uint16 depth in 1/100 units, 8-bit mask) -- with the
                                  reference's image_resize 0.5 the pipeline works at 320x180 at the reference's own pixel scale
                                  (at a coarser scale the 13 860-triangle mesh is far below one pixel per triangle and the
                                  antialiased mask carries no gradient)
  example/diffdope.yaml           the reference's configuration values with the principal point shifted to the window
  example/expected.json           parse check values of the ORIGINAL files (vertex/face counts, bounding box, first vertices,
                                  depth/seg statistics) that tests compare the readers against
"""
import json
import os
import struct
import sys

import numpy as np
import yaml
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(HERE, "example")

from diffdope_amd import io_img, io_ply  # noqa: E402

m = io_ply.read_ply(os.path.join(REF, "data/example/mesh/AlphabetSoup.ply"))
V, T = m["pos"].shape[0], m["faces"].shape[0]
os.makedirs(os.path.join(OUT, "mesh"), exist_ok=True)
os.makedirs(os.path.join(OUT, "scene"), exist_ok=True)
with open(os.path.join(OUT, "mesh", "AlphabetSoup.ply"), "wb") as f:
    f.write(("ply\nformat binary_little_endian 1.0\ncomment TextureFile AlphabetSoup.png\n"
             f"element vertex {V}\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\n"
             "property float nz\nproperty float texture_u\nproperty float texture_v\n"
             f"element face {T}\nproperty list uchar int vertex_indices\nend_header\n").encode("ascii"))
    rows = np.concatenate([m["pos"], m["normals"], m["uv"]], 1).astype("<f4")
    f.write(rows.tobytes())
    for t in m["faces"]:
        f.write(struct.pack("<Biii", 3, *[int(i) for i in t]))
tex = Image.open(os.path.join(REF, "data/example/mesh/AlphabetSoup.png")).convert("RGB")
tex.resize((512, 512), Image.BOX).save(os.path.join(OUT, "mesh", "AlphabetSoup.png"), optimize=True)

X0, Y0, CW, CH = 358, 714, 640, 360
rgb = Image.open(os.path.join(REF, "data/example/scene/rgb.png")).convert("RGB")
rgb.crop((X0, Y0, X0 + CW, Y0 + CH)).save(os.path.join(OUT, "scene", "rgb.png"), optimize=True)
depth = np.asarray(Image.open(os.path.join(REF, "data/example/scene/depth.png")))
seg = np.asarray(Image.open(os.path.join(REF, "data/example/scene/seg.png")))
Image.fromarray(depth[Y0:Y0 + CH, X0:X0 + CW].astype(np.uint16)).save(os.path.join(OUT, "scene", "depth.png"), optimize=True)
Image.fromarray(seg[Y0:Y0 + CH, X0:X0 + CW].astype(np.uint8), mode="L").save(os.path.join(OUT, "scene", "seg.png"), optimize=True)

cfg = yaml.safe_load(open(os.path.join(REF, "configs/diffdope.yaml")))
cam = cfg["camera"]
cfg["camera"] = dict(fx=cam["fx"], fy=cam["fy"], cx=cam["cx"] - X0, cy=cam["cy"] - Y0, im_width=CW, im_height=CH)
cfg["scene"] = dict(path_img="scene/rgb.png", path_depth="scene/depth.png", path_segmentation="scene/seg.png", image_resize=0.5)
cfg["object3d"]["model_path"] = "mesh/AlphabetSoup.ply"
with open(os.path.join(OUT, "diffdope.yaml"), "w") as f:
    f.write(f"# values of the reference's configs/diffdope.yaml; principal point shifted by ({X0}, {Y0}) to the {CW}x{CH} window of the fixture images, paths relative to this file\n")
    yaml.safe_dump(cfg, f, sort_keys=False)

d_full = io_img.imread_depth(os.path.join(REF, "data/example/scene/depth.png"))
expected = dict(
    V=int(V), T=int(T), texture_file="AlphabetSoup.png", has_normals=m["normals"] is not None,
    bbox_min=[float(x) for x in m["pos"].min(0)], bbox_max=[float(x) for x in m["pos"].max(0)],
    first_vertices=[[float(x) for x in r] for r in m["pos"][:3]], first_uv=[[float(x) for x in r] for r in m["uv"][:3]],
    first_faces=[[int(x) for x in r] for r in m["faces"][:3]], uv_min=float(m["uv"].min()), uv_max=float(m["uv"].max()),
    pos_sum=float(m["pos"].astype(np.float64).sum()), faces_sum=int(m["faces"].astype(np.int64).sum()),
    window=[X0, Y0, CW, CH], seg_pixels_in_window=int((seg[Y0:Y0 + CH, X0:X0 + CW] > 0).sum()), seg_pixels=int((seg > 0).sum()),
    scene=dict(width=1920, height=1080, seg_fraction=float((seg > 0).mean()), depth_max_raw=int(depth.max()),
               depth_at_seg_mean_units=float((d_full[seg > 0] / 100.0).mean()), rgb_mean_in_window=float(np.asarray(rgb.crop((X0, Y0, X0 + CW, Y0 + CH))).mean() / 255.0)),
)
json.dump(expected, open(os.path.join(OUT, "expected.json"), "w"), indent=1)
print(json.dumps(expected)[:400])
os.system(f"du -sh {OUT}/* {OUT}/*/*")
