"""The diff-dope Python API (diffdope/diffdope.py) on top of the MI355X engine: same class names,
constructor arguments, attributes and tensor layouts -- Camera, Mesh, Object3D, Image, Scene, DiffDope,
the loss functions and pose helpers -- without trimesh / cv2 / pyrr / hydra / nvdiffrast.

What differs on purpose ("Logging semantics"):
  * `set_batchsize` makes stride-0 batch VIEWS (`expand`) instead of B physical copies of the mesh,
    texture and images (diffdope.py:875-893,1176); shapes are unchanged;
  * `run_optimization` takes the fused engine when every loss function is a built-in one (four kernel
    launches per iteration, no per-iteration device->host copies); any user loss
    function switches to the op-by-op autograd path, which behaves like the reference's loop;
  * `optimization_results[i]` always holds "mtx"; "rgb"/"depth"/"mask" are rendered on first access from
    the stored pose instead of being copied to the host every iteration (diffdope.py:1698-1703).
"""
import logging
import math
import random
from dataclasses import dataclass
from typing import Optional

import os

import numpy as np
import torch

from . import io_img, io_ply
from . import ops as dd_ops
from .engine import RefineEngine
from .pose import matrix_batch_44_from_position_quat, quat_trans_from_parameters
from .render import RasterizeGLContext, masked_l1_mean, render_texture_batch

log = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------------
# config: any attribute/item mapping with the keys of configs/diffdope.yaml works; this one needs no hydra
class Cfg(dict):
    """dict with attribute access, usable as `**cfg.camera` like an omegaconf DictConfig."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return Cfg({k: Cfg.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return [Cfg.wrap(v) for v in obj]
        return obj


def load_config(path):
    """Read a diffdope.yaml (configs/diffdope.yaml layout) into a Cfg."""
    import yaml

    with open(path) as f:
        return Cfg.wrap(yaml.safe_load(f))


# ------------------------------------------------------------------------------------------------
# pose helpers (diffdope.py:92-140 without pyrr)
def quat_from_matrix(m):
    """xyzw quaternion of a 3x3 rotation matrix (column-vector convention, what pyrr.Matrix33(m).quaternion gives)."""
    m = np.asarray(m, np.float64).reshape(3, 3)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = 0.5 / math.sqrt(tr + 1.0)
        q = [(m[2, 1] - m[1, 2]) * s, (m[0, 2] - m[2, 0]) * s, (m[1, 0] - m[0, 1]) * s, 0.25 / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = 2.0 * math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2])
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = 2.0 * math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2])
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = 2.0 * math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1])
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    q = np.array(q, np.float64)
    return q / np.linalg.norm(q)


def matrix_from_quat(q):
    x, y, z, w = np.asarray(q, np.float64) / np.linalg.norm(q)
    return np.array([
        [1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
        [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
        [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y],
    ])


def opencv_2_opengl(p, q):
    """OpenCV camera frame -> OpenGL camera frame (diffdope.py:92-140): [R|t] -> diag(1,-1,-1) [R|t].
    The reference's trailing "legacy" quaternion product Rz(90) Ry(-90) Rz(-90) Rx(-90) is the identity
    (checked numerically), so it is not reproduced.  q is xyzw; returns (p', q')."""
    flip = np.diag([1.0, -1.0, -1.0])
    R = flip @ matrix_from_quat(q)
    return flip @ np.asarray(p, np.float64), quat_from_matrix(R)


# ------------------------------------------------------------------------------------------------
# losses (diffdope.py:534-613): f(ddope) -> scalar, reading ddope.renders / gt_tensors / learning_rates / cfg
def dist_batch_lr(tensor, learning_rates, channels=[1, 2, 3]):
    return torch.mean(tensor, channels) * learning_rates


def _lr_weights(ddope, weight):
    """learning_rates * weight / B: with it, (v * learning_rates).mean() * weight (dist_batch_lr and the term's weight, diffdope.py:
    534-544) is one weighted sum of v, which masked_l1_mean folds into its own launches.  Cached per (the learning_rates TENSOR
    OBJECT, its version counter, weight): the entry holds the tensor itself, so a replaced tensor can neither be mistaken for the
    old one at a recycled address nor be freed under the cache.  (An edit through `learning_rates.data` bumps no version counter
    and is not seen: assign a new tensor or write in place through the tensor itself, as set_batchsize and the examples do.)"""
    lr = ddope.learning_rates
    cache = getattr(ddope, "_lr_weights_cache", None)
    if cache is None or cache[0] is not lr or cache[1] != lr._version:
        cache = (lr, lr._version, {})
        ddope._lr_weights_cache = cache
    if float(weight) not in cache[2]:
        cache[2][float(weight)] = (lr.detach() * (float(weight) / lr.shape[0])).contiguous()
    return cache[2][float(weight)]


def l1_rgb_with_mask(ddope):
    """diffdope.py:547-562 (image-space part and the weighting by the learning rates fused for ROCm tensors: render.masked_l1_mean)."""
    w = ddope.cfg.losses.weight_rgb
    v, term = masked_l1_mean(ddope.renders["rgb"], ddope.gt_tensors["rgb"], ddope.gt_tensors["segmentation"], batch_weights=_lr_weights(ddope, w))
    ddope.add_loss_value("rgb", v.detach() * w)
    return term


def l1_depth_with_mask(ddope):
    """diffdope.py:565-580."""
    w = ddope.cfg.losses.weight_depth
    v, term = masked_l1_mean(ddope.renders["depth"], ddope.gt_tensors["depth"], ddope.gt_tensors["segmentation"], mask_channel0=True,
                             batch_weights=_lr_weights(ddope, w))
    ddope.add_loss_value("depth", v.detach() * w)
    return term


def l1_mask(ddope):
    """diffdope.py:583-613."""
    w = ddope.cfg.losses.weight_mask
    v, term = masked_l1_mean(ddope.renders["mask"], ddope.gt_tensors["segmentation"], batch_weights=_lr_weights(ddope, w))
    ddope.add_loss_value("mask_selection", v.detach() * w)
    return term


_SOBEL = None


def _sobel_xy(img):
    """[B,H,W,3] -> [B,2,H,W]: Sobel gradients (coefficients / 8, zero padding) of the luminance (r+g+b)/3."""
    global _SOBEL
    if _SOBEL is None or _SOBEL.device != img.device:
        kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], device=img.device) / 8.0
        _SOBEL = torch.stack([kx, kx.t()])[:, None].contiguous()
    lum = ((img[..., 0] + img[..., 1]) + img[..., 2]) * (1.0 / 3.0)
    return torch.nn.functional.conv2d(lum[:, None], _SOBEL.to(img.dtype), padding=1)


def l1_edge(ddope):
    """EXTENSION -- the reference has no edge loss (BASELINE configs 3/5 name one).  L1 between the Sobel
    gradients of the luminance of the rendered colour image and of the observed image masked by its
    segmentation (both "object on black"); definition shared with oracle/ddx_oracle.c:orc_loss_edge and the
    fused engine.  Enabled by cfg.losses.l1_edge / weight_edge."""
    w = float(ddope.cfg.losses.get("weight_edge", 1.0))
    diff = torch.abs(_sobel_xy(ddope.renders["rgb"]) - _sobel_xy(ddope.gt_tensors["rgb"] * ddope.gt_tensors["segmentation"]))
    lr_diff = dist_batch_lr(diff, ddope.learning_rates)
    ddope.add_loss_value("edge", torch.mean(diff.detach(), (1, 2, 3)) * w)
    return lr_diff.mean() * w


_BUILTIN_LOSSES = {l1_rgb_with_mask: "rgb", l1_depth_with_mask: "depth", l1_mask: "mask", l1_edge: "edge"}
_LOG_KEYS = {"rgb": "rgb", "depth": "depth", "mask": "mask_selection", "edge": "edge"}


# ------------------------------------------------------------------------------------------------
@dataclass
class Camera:
    """Intrinsics -> OpenGL projection (diffdope.py:621-742)."""

    fx: float
    fy: float
    cx: float
    cy: float
    im_width: int
    im_height: int
    znear: Optional[float] = 0.01
    zfar: Optional[float] = 200

    def __post_init__(self):
        self.cam_proj = self.get_projection_matrix()

    def set_batchsize(self, batchsize):
        base = self.cam_proj if self.cam_proj.dim() == 2 else self.cam_proj[0]
        self.cam_proj = base[None].expand(batchsize, 4, 4)

    def cuda(self):
        self.cam_proj = self.cam_proj.cuda().float()

    def resize(self, percentage):
        """Scale the intrinsics with the image (diffdope.py:665-677): focal lengths exactly, principal point and image size
        truncated to whole pixels."""
        k = float(percentage)
        self.fx, self.fy = self.fx * k, self.fy * k
        for name in ("cx", "cy", "im_width", "im_height"):
            setattr(self, name, int(k * getattr(self, name)))

    def get_projection_matrix(self):
        """'y_down' window convention of diffdope.py:726-740."""
        w, h, nc, fc = self.im_width, self.im_height, self.znear, self.zfar
        depth = float(fc - nc)
        q = -(fc + nc) / depth
        qn = -2 * (fc * nc) / depth
        proj = np.array([
            [2 * self.fx / w, -2 * 0 / w, (-2 * self.cx + w) / w, 0],
            [0, 2 * self.fy / h, (2 * self.cy - h) / h, 0],
            [0, 0, q, qn],
            [0, 0, -1, 0],
        ])
        return torch.tensor(proj)


class Mesh(torch.nn.Module):
    """Mesh tensors for the renderer (diffdope.py:746-935).  `path_model` is a PLY file (with `texture_u/texture_v` or
    per-face texture coordinates + `comment TextureFile` for a textured model, or per-vertex colours) or a Wavefront OBJ
    (`vt` + `mtllib` -> `map_Kd`); or build one from arrays with Mesh.from_arrays."""

    def __init__(self, path_model=None, scale=1, _arrays=None):
        super().__init__()
        self.path_model = path_model
        self.to_process = ["pos", "pos_idx", "vtx_color", "tex", "uv", "uv_idx", "vtx_normals"]
        if _arrays is None:
            if str(path_model).lower().endswith(".obj"):
                from . import io_obj

                m = io_obj.read_obj(path_model)
            else:
                m = io_ply.read_ply(path_model)
            tex = None
            if m["uv"] is not None and m["texture_file"] is not None:
                from PIL import Image as PILImage

                tex = np.asarray(PILImage.open(m["texture_file"]).convert("RGB")).astype(np.float64) / 255.0
            colors = None if m["colors"] is None else m["colors"].astype(np.float64) / 255.0
            _arrays = dict(pos=m["pos"], faces=m["faces"], normals=m["normals"], uv=m["uv"], tex=tex, colors=colors)
        a = _arrays
        pos_idx = torch.from_numpy(np.ascontiguousarray(a["faces"]).astype(np.int32))
        vtx_pos = torch.from_numpy(np.ascontiguousarray(a["pos"]).astype(np.float32)) * scale
        normals = a.get("normals")
        if normals is None:
            normals = io_ply.vertex_normals(np.asarray(a["pos"], np.float64), np.asarray(a["faces"]))
        self.pos_idx = pos_idx
        self.pos = vtx_pos
        self.vtx_normals = torch.from_numpy(np.ascontiguousarray(normals).astype(np.float32))
        lo, hi = vtx_pos.min(0).values, vtx_pos.max(0).values
        self.bounding_volume = [[lo[0], lo[1], lo[2]], [hi[0], hi[1], hi[2]]]
        self.dimensions = [hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]]
        self.center_point = [((lo[i] + hi[i]) / 2).item() for i in range(3)]
        if a.get("tex") is not None and a.get("uv") is not None:
            uv = np.array(a["uv"], np.float32, copy=True)
            if a.get("flip_v", True):
                uv[:, 1] = 1 - uv[:, 1]  # diffdope.py:822
            self.tex = torch.from_numpy(np.ascontiguousarray(a["tex"]).astype(np.float32))
            self.uv = torch.from_numpy(uv)
            self.uv_idx = pos_idx.clone()
            self.has_textured_map = True
        else:
            col = a.get("colors")
            if col is None:
                col = np.full((vtx_pos.shape[0], 3), 0.5)
            self.vtx_color = torch.from_numpy(np.ascontiguousarray(col).astype(np.float32))
            self.has_textured_map = False
        log.info("mesh %s: %d vertices, %d faces, %s", self.path_model, int(vtx_pos.shape[0]), int(pos_idx.shape[0]), "texture map" if self.has_textured_map else "vertex colours")
        self._batchsize_set = False

    @classmethod
    def from_arrays(cls, pos, faces, uv=None, tex=None, vtx_color=None, normals=None, scale=1, flip_v=False):
        """Mesh from numpy arrays (pos [V,3], faces [T,3], and uv [V,2] + tex [Th,Tw,3] or vtx_color [V,3] in 0..1).
        flip_v=False: uv is already in the renderer's convention."""
        return cls(None, scale, _arrays=dict(pos=pos, faces=faces, uv=uv, tex=tex, colors=vtx_color, normals=normals, flip_v=flip_v))

    def __str__(self):
        return f"mesh @{self.path_model}. vtx:{self.pos.shape} on {self.pos.device}"

    __repr__ = __str__

    def set_batchsize(self, batchsize):
        """Batch VIEWS of every array in `to_process` ([B,...], stride 0 on the batch axis)."""
        for key in list(vars(self).keys()):
            if key not in self.to_process:
                continue
            v = vars(self)[key]
            base = v[0] if self._batchsize_set else v
            vars(self)[key] = base[None].expand(batchsize, *base.shape)
        self._batchsize_set = True

    def cuda(self):
        super().cuda()
        for key in list(vars(self).keys()):
            if key in self.to_process:
                vars(self)[key] = vars(self)[key].cuda()

    def enable_gradients_texture(self):
        if self.has_textured_map:
            self.tex = torch.nn.Parameter(self.tex.contiguous(), requires_grad=True).to(self.tex.device)
        else:
            self.vtx_color = torch.nn.Parameter(self.vtx_color.contiguous(), requires_grad=True).to(self.vtx_color.device)

    def forward(self):
        return {key: vars(self)[key] for key in vars(self) if key in self.to_process}


class Object3D(torch.nn.Module):
    """The 7 pose parameters, each an nn.Parameter of shape [B] (diffdope.py:938-1098)."""

    def __init__(self, position, rotation, batchsize=32, opencv2opengl=True, model_path=None, scale=1, mesh=None):
        super().__init__()
        self.qx = None
        self.mesh = mesh if mesh is not None else (None if model_path is None else Mesh(path_model=model_path, scale=scale))
        self.set_pose(position, rotation, batchsize, scale=scale, opencv2opengl=opencv2opengl)

    def _make_params(self, batchsize, rotation, position):
        for name, val in zip(("qx", "qy", "qz", "qw"), rotation):
            setattr(self, name, torch.nn.Parameter(torch.ones(batchsize) * float(val)))
        for name, val in zip(("x", "y", "z"), position):
            setattr(self, name, torch.nn.Parameter(torch.ones(batchsize) * float(val)))

    def set_pose(self, position, rotation, batchsize=32, opencv2opengl=True, scale=1):
        assert len(position) == 3
        position = np.array(position, np.float64) * scale
        rotation = np.asarray(rotation, np.float64)
        assert rotation.size in (4, 9)
        rotation = rotation.reshape(-1) if rotation.size == 4 else quat_from_matrix(rotation)
        if opencv2opengl:
            position, rotation = opencv_2_opengl(position, rotation)
        log.info("Object3D pose set: t = %s, q (xyzw) = %s", np.round(np.asarray(position, np.float64), 6).tolist(), np.round(np.asarray(rotation, np.float64), 6).tolist())
        self._position, self._rotation = position, rotation
        device = "cpu" if self.qx is None else self.qx.device
        self._make_params(batchsize, rotation, position)
        self.to(device)
        if self.mesh is not None and torch.cuda.is_available():
            self.mesh.cuda()

    def set_batchsize(self, batchsize):
        device = self.qx.device
        self._make_params(batchsize, self._rotation, self._position)
        self.to(device)
        if self.mesh is not None:
            self.mesh.set_batchsize(batchsize=batchsize)
            if torch.cuda.is_available():
                self.mesh.cuda()

    def __repr__(self):
        return f"Object3D( \n (pos): {self.x.shape} ,[0]:[{self.x[0].item(), self.y[0].item(), self.z[0].item()}] on {self.x.device}\n (mesh): {self.mesh} \n)"

    def cuda(self):
        super().cuda()
        if self.mesh is not None:
            self.mesh.cuda()

    def reset_pose(self):
        device = self.qx.device
        self._make_params(self.qx.shape[0], self._rotation, self._position)
        self.to(device)

    def params_tensor(self):
        """[7,B] contiguous copy: qx,qy,qz,qw,x,y,z (the engine's parameter layout)."""
        return torch.stack([self.qx, self.qy, self.qz, self.qw, self.x, self.y, self.z]).detach().contiguous()

    def load_params_tensor(self, p):
        with torch.no_grad():
            for i, name in enumerate(("qx", "qy", "qz", "qw", "x", "y", "z")):
                getattr(self, name).copy_(p[i])

    def forward(self):
        """The mesh dictionary plus the pose of every hypothesis: `quat` [B,4] (x, y, z, w, normalised here -- the seven parameters
        are free, so the optimiser may leave the unit sphere) and `trans` [B,3] (semantics of diffdope.py:1085-1098)."""
        out = dict(self.mesh())
        out["quat"], out["trans"] = quat_trans_from_parameters(self.qx, self.qy, self.qz, self.qw, self.x, self.y, self.z)
        return out


@dataclass
class Image:
    """rgb / segmentation / depth image tensor, stored bottom-up (diffdope.py:1101-1180)."""

    img_path: Optional[str] = None
    img_tensor: Optional[torch.Tensor] = None
    img_resize: Optional[float] = 1
    flip_img: Optional[bool] = True
    depth: Optional[bool] = False
    depth_scale: Optional[float] = 100

    def __post_init__(self):
        if self.img_path is not None:
            if self.depth:
                im = io_img.imread_depth(self.img_path) / self.depth_scale
            else:
                im = io_img.imread_rgb(self.img_path)
            if self.flip_img:
                im = im[::-1]
            if self.img_resize is not None and self.img_resize < 1.0:
                ow, oh = int(im.shape[1] * self.img_resize), int(im.shape[0] * self.img_resize)
                im = io_img.resize_nearest(im, ow, oh) if self.depth else io_img.resize_linear(im, ow, oh)
            self.img_tensor = torch.tensor(np.ascontiguousarray(im)).float()
            log.info("image %s read as %s", self.img_path, tuple(self.img_tensor.shape))
        self._batchsize_set = False

    def __repr__(self):
        return f"{self.img_tensor.shape} @ {self.img_path} on {self.img_tensor.device}"

    __str__ = __repr__

    def cuda(self):
        self.img_tensor = self.img_tensor.cuda().float()

    def set_batchsize(self, batchsize):
        base = self.img_tensor[0] if self._batchsize_set else self.img_tensor
        self.img_tensor = base[None].expand(batchsize, *base.shape)
        self._batchsize_set = True


@dataclass
class Scene:
    """Observed images (diffdope.py:1183-1264)."""

    path_img: Optional[str] = None
    path_depth: Optional[str] = None
    path_segmentation: Optional[str] = None
    image_resize: Optional[float] = None
    tensor_rgb: Optional[Image] = None
    tensor_depth: Optional[Image] = None
    tensor_segmentation: Optional[Image] = None

    def __post_init__(self):
        if self.path_img is not None:
            self.tensor_rgb = Image(self.path_img, img_resize=self.image_resize)
        if self.path_depth is not None:
            self.tensor_depth = Image(self.path_depth, img_resize=self.image_resize, depth=True)
        if self.path_segmentation is not None:
            self.tensor_segmentation = Image(self.path_segmentation, img_resize=self.image_resize)

    def _images(self):
        return [t for t in (self.tensor_rgb, self.tensor_depth, self.tensor_segmentation) if t is not None]

    def set_batchsize(self, batchsize):
        for t in self._images():
            t.set_batchsize(batchsize)

    def get_resolution(self):
        """[H, W] of the optimisation images (diffdope.py:1231-1252)."""
        if self.tensor_rgb is not None:
            return [self.tensor_rgb.img_tensor.shape[-3], self.tensor_rgb.img_tensor.shape[-2]]
        if self.tensor_depth is not None:
            return [self.tensor_depth.img_tensor.shape[-2], self.tensor_depth.img_tensor.shape[-1]]
        if self.tensor_segmentation is not None:
            return [self.tensor_segmentation.img_tensor.shape[-3], self.tensor_segmentation.img_tensor.shape[-2]]

    def cuda(self):
        for t in self._images():
            t.cuda()


class _LossLog(dict):
    """losses_values of the op-by-op path: {key: [iterations, B] CPU tensor} like the reference's (diffdope.py:1600-1616), but
    the rows a loss function logs stay on the device until somebody reads the log -- a host copy per loss term and iteration
    is a synchronisation per term and iteration, and the host could never run ahead of the GPU."""

    def __init__(self):
        super().__init__()
        self._pending = {}

    def add(self, key, row):
        self._pending.setdefault(key, []).append(row)

    def _flush(self):
        if not self._pending:
            return
        pending, self._pending = self._pending, {}
        for key, rows in pending.items():
            v = torch.stack(rows, dim=0).cpu()
            dict.__setitem__(self, key, v if not dict.__contains__(self, key) else torch.cat((dict.__getitem__(self, key), v), dim=0))

    def _reader(name):
        def f(self, *a, **k):
            self._flush()
            return getattr(dict, name)(self, *a, **k)
        f.__name__ = name
        return f

    # (writers too: setdefault / update / item assignment on a key with un-flushed rows would otherwise see a dict without them)
    for _n in ("__getitem__", "__iter__", "__len__", "__contains__", "__repr__", "__eq__", "__reversed__", "__or__", "__ror__", "__ior__", "keys",
               "values", "items", "get", "copy", "pop", "popitem", "setdefault", "update", "__setitem__", "__delitem__", "clear"):
        locals()[_n] = _reader(_n)
    del _n, _reader


class _LazyResult(dict):
    """optimization_results entry: "mtx" is stored (moved to the host on first access); "rgb"/"depth"/"mask" are rendered on
    first access."""

    def __init__(self, mtx, render_fn):
        super().__init__(mtx=mtx)
        self._render_fn = render_fn

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if key == "mtx" and v.is_cuda:
            v = v.cpu()
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        try:
            return self[key]  # (through __getitem__ / __missing__: "mtx" comes back on the host, the images are rendered)
        except KeyError:
            return default

    def __missing__(self, key):
        if key in ("rgb", "depth", "mask"):
            for k, v in self._render_fn(self["mtx"]).items():
                self[k] = v
            return dict.__getitem__(self, key)
        raise KeyError(key)


@dataclass
class DiffDope:
    """The optimisation driver (diffdope.py:1267-1725).  `cfg` is any attribute mapping with the keys of
    configs/diffdope.yaml (see load_config); camera / object3d / scene may be passed ready-made."""

    cfg: Optional[dict] = None
    camera: Optional[Camera] = None
    object3d: Optional[Object3D] = None
    scene: Optional[Scene] = None
    resolution: Optional[list] = None
    batchsize: Optional[int] = 16

    def __post_init__(self):
        self.cfg = Cfg.wrap(self.cfg)
        if self.camera is None:
            self.camera = Camera(**self.cfg.camera)
        if self.object3d is None:
            self.object3d = Object3D(**self.cfg.object3d)
        if self.scene is None:
            self.scene = Scene(**self.cfg.scene)
        self.batchsize = self.cfg.hyperparameters.batchsize
        self.glctx = RasterizeGLContext()
        self.cuda()
        self.resolution = self.scene.get_resolution()
        self.optimization_results = []
        self.gt_tensors = {}
        self._refresh_gt()
        self.set_batchsize(self.batchsize)
        self.losses_values = _LossLog()
        self.loss_functions = []
        if self.cfg.losses.l1_rgb_with_mask:
            self.loss_functions.append(l1_rgb_with_mask)
        if self.cfg.losses.l1_depth_with_mask:
            self.loss_functions.append(l1_depth_with_mask)
        if self.cfg.losses.l1_mask:
            self.loss_functions.append(l1_mask)
        if self.cfg.losses.get("l1_edge", False):  # extension, absent from the reference's yaml
            self.loss_functions.append(l1_edge)
        self.last_engine = None
        log.info("DiffDope ready: %d hypotheses, losses %s", self.batchsize, [f.__name__ for f in self.loss_functions])

    def _refresh_gt(self):
        if self.scene.tensor_rgb is not None:
            self.gt_tensors["rgb"] = self.scene.tensor_rgb.img_tensor
        if self.scene.tensor_depth is not None:
            self.gt_tensors["depth"] = self.scene.tensor_depth.img_tensor
        if self.scene.tensor_segmentation is not None:
            self.gt_tensors["segmentation"] = self.scene.tensor_segmentation.img_tensor

    def set_batchsize(self, batchsize):
        self.batchsize = batchsize
        self.scene.set_batchsize(batchsize)
        self.object3d.set_batchsize(batchsize)
        self.camera.set_batchsize(batchsize)
        self._refresh_gt()
        self.optimizer = torch.optim.SGD(self.object3d.parameters(), lr=self.cfg.hyperparameters.learning_rate_base)
        lo, hi = self.cfg.hyperparameters.learning_rates_bound
        rng = random.Random(self.cfg.hyperparameters.get("seed")) if self.cfg.hyperparameters.get("seed") is not None else random
        self.learning_rates = torch.tensor([rng.uniform(lo, hi) for _ in range(batchsize)]).float().cuda()

    # ---- results ---------------------------------------------------------------------------------
    def get_argmin(self):
        """argmin over hypotheses of the mean over loss keys of the last-step losses (diffdope.py:1488-1513)."""
        stacked = torch.stack([t[-1] for t in self.losses_values.values()], dim=0)
        return torch.argmin(stacked.mean(dim=0), dim=-1)

    def get_pose(self, batch_index=-1):
        """4x4 numpy pose (OpenGL camera frame) of hypothesis `batch_index` (argmin if -1), diffdope.py:1618-1632."""
        if batch_index == -1:
            batch_index = self.get_argmin()
        return self.optimization_results[-1]["mtx"][batch_index].numpy()

    def add_loss_value(self, key, values, values_weighted=None):
        """diffdope.py:1600-1616 (the copy to the host is deferred to the first read of losses_values, see _LossLog)."""
        if not isinstance(self.losses_values, _LossLog):  # (a caller replaced the log by a plain dict)
            log_, self.losses_values = self.losses_values, _LossLog()
            self.losses_values.update(log_)
        cap = getattr(self, "_capture", None)
        if cap is not None:
            # (inside the captured iteration of run_optimization(graph=True): the row goes to a [iterations, B] device buffer at the
            # row a device counter names -- a Python-side append would happen once, at capture time)
            buf = cap["logs"].get(key)
            if buf is None:
                raise RuntimeError(f"run_optimization(graph=True): the loss functions logged '{key}' for the first time inside the captured "
                                   "iteration (a key must appear in every iteration: the log buffers are laid out from the eager ones)")
            buf.index_copy_(0, cap["it"], values.detach()[None])
            return
        shapes = getattr(self, "_log_shapes", None)
        if shapes is not None:
            shapes[key] = (tuple(values.shape), values.dtype)
        self.losses_values.add(key, values.detach().clone())

    # ---- rendering -------------------------------------------------------------------------------
    def _render(self, mtx, outputs=None, compact_mask=False):
        r = self.object3d.mesh()
        kw = dict(uv=r["uv"], uv_idx=r["uv_idx"], tex=r["tex"]) if self.object3d.mesh.has_textured_map else dict(vtx_color=r["vtx_color"])
        return render_texture_batch(glctx=self.glctx, proj_cam=self.camera.cam_proj, mtx=mtx, pos=r["pos"], pos_idx=r["pos_idx"],
                                    resolution=self.resolution, outputs=outputs, compact_mask=compact_mask, **kw)

    def _builtin_losses_only(self):
        """Every loss function of this run is one of this module's own (they read `mask` through masked_l1_mean, which works on the
        one stored channel of a compact mask): only then does the loop ask render_texture_batch for the compact form -- a user
        function gets the reference's ordinary [B,H,W,3] tensor (diffdope.py:212-214)."""
        return bool(self.loss_functions) and all(f in _BUILTIN_LOSSES for f in self.loss_functions)

    def _loop_outputs(self):
        """What the loss functions of this run read from self.renders: known for the built-in ones (None = everything: a user
        function may read anything).  An image no term reads is not rendered in the loop (render_texture_batch(outputs=...));
        the complete images of the last iteration are rendered once more when the loop is done, as the reference leaves them."""
        if not self.loss_functions or not all(f in _BUILTIN_LOSSES for f in self.loss_functions):
            return None
        need = {"rgb" if _BUILTIN_LOSSES[f] == "edge" else _BUILTIN_LOSSES[f] for f in self.loss_functions}
        return None if need == {"rgb", "depth", "mask"} else tuple(sorted(need))

    def _render_cpu(self, mtx_cpu):
        with torch.no_grad():
            out = self._render(mtx_cpu.cuda())
        return {k: out[k].detach().cpu() for k in ("rgb", "depth", "mask")}

    def render_img(self, index=None, batch_index=None, render_selection="rgb"):
        """Overlay of the render of iteration `index` (default: last) on the observed image, as an upright uint8
        RGB numpy image: one hypothesis (`batch_index`) or, with batch_index=None, a grid of all of them.
        Honours cfg.render_images.{nrow, add_background, alpha_overlay, flip_result, crop_around_mask} when the
        config has them, plus add_countour / color_countour (sic) for the silhouette contour (diffdope.py:1377-1486)."""
        from . import viz

        ri = self.cfg.get("render_images", {}) if isinstance(self.cfg, dict) else {}
        index = -1 if index is None else index
        entry = self.optimization_results[index]
        fg_all = entry[render_selection]
        bg_all = self.gt_tensors.get(render_selection)
        which = range(fg_all.shape[0]) if batch_index is None else [int(batch_index)]
        crop = None
        if ri.get("crop_around_mask", False):
            src = self.gt_tensors["segmentation"][0].cpu() if "segmentation" in self.gt_tensors else fg_all[0]
            crop = viz.find_crop(src.numpy())
        tiles = []
        for b in which:
            fg = fg_all[b].numpy()
            bg = None if bg_all is None else bg_all[b].cpu().numpy()
            if crop is not None:
                r0, c0, sz = crop
                fg = fg[r0:r0 + sz + 1, c0:c0 + sz + 1]
                bg = None if bg is None else bg[r0:r0 + sz + 1, c0:c0 + sz + 1]
            im = viz.overlay(bg, fg, alpha=ri.get("alpha_overlay", 0.7), add_background=ri.get("add_background", True))
            if ri.get("add_countour", False) and fg.ndim == 3:  # (sic: the reference's config key, configs/diffdope.yaml:40)
                im = im.copy()
                im[viz.contour(fg.sum(-1) > 0)] = np.asarray(ri.get("color_countour", [0.46, 0.73, 0]), np.float32)
            tiles.append(im[::-1] if ri.get("flip_result", True) else im)
        img = tiles[0] if len(tiles) == 1 else viz.make_grid(tiles, nrow=ri.get("nrow", 4))
        return viz.to_uint8(img)

    def make_animation(self, output_file_path="animation.gif", frame_rate=10, batch_index=-1):
        """One frame per iteration of hypothesis `batch_index` (argmin if -1), written as a GIF through PIL (the
        reference writes mp4 through imageio/libx264, diffdope.py:1515-1552)."""
        from . import viz

        if batch_index == -1:
            batch_index = int(self.get_argmin())
        frames = [self.render_img(index=i, batch_index=batch_index) for i in range(len(self.optimization_results))]
        return viz.save_gif(frames, output_file_path, fps=frame_rate)

    def plot_losses(self, keys=None, batch_index=-1):
        """Loss curves of one hypothesis as an RGB image array (diffdope.py:1573-1616); None before a run."""
        from . import viz

        if len(self.losses_values) == 0:
            return None
        if batch_index == -1:
            batch_index = int(self.get_argmin())
        return viz.plot_losses(self.losses_values, batch_index, keys)

    # ---- optimisation ----------------------------------------------------------------------------
    def lr_schedule(self):
        hp = self.cfg.hyperparameters
        return [hp.base_lr * hp.lr_decay ** (it / hp.nb_iterations + 1) for it in range(hp.nb_iterations + 1)]

    def prepare_optimization(self, optimizer="sgd", global_batch=None, shade_slices=0, edge_slices=0, separate_big_pass=False):
        """First half of run_optimization(wait=False) for callers that run several objects as ONE engine group
        (bop.refine_frame): resets the logs, builds / refreshes the fused engine and returns it WITHOUT launching anything; after
        the group has run, finish_optimization() collects the results as usual.  Only for the built-in losses."""
        self.losses_values = _LossLog()
        self.optimization_results = []
        self._refresh_gt()
        if not (all(f in _BUILTIN_LOSSES for f in self.loss_functions) and len(self.loss_functions) > 0):
            raise RuntimeError("the fused engine only knows l1_rgb_with_mask / l1_depth_with_mask / l1_mask / l1_edge")
        eng, params, weights = self._fused_prepare(optimizer, global_batch, shade_slices, edge_slices, separate_big_pass)
        self._pending = (eng, params, weights, torch.cuda.current_stream())
        return eng

    def run_optimization(self, fused=None, optimizer="sgd", global_batch=None, wait=True, graph=None):
        """diffdope.py:1634-1714.  fused=None picks the fused engine when every loss function is a built-in.
        wait=False (fused path only) enqueues the whole optimisation on the current stream and returns; call
        finish_optimization() to synchronise and fetch the results -- several objects can then run on one stream each
        and fill each other's launch tails (bop.refine_frame).
        graph (op-by-op path, fused=False): the loop body -- Object3D.forward, the pose matrices, render_texture_batch, every loss
        function, backward, the SGD step -- is ~60 framework launches per iteration and host-bound; with graph=True the first two
        iterations run eagerly and the third is captured ONCE into a hipGraph that is replayed for the rest of this call (the learning
        rate, the loss rows and the pose log are indexed by a device-side iteration counter inside the graph; the graph is destroyed
        before the call returns).  A loss function must
        then be capture-safe: tensor operations and ddope.add_loss_value only -- no .item() / .cpu(), and no Python-side state
        that has to change every iteration (it would change once).  Default (None / False): eager, as the reference runs it.
        Measured on cfg2 (64 hypotheses, 640x480, tools/bench_opbyop.py --api): 1.1-1.2 k it/s eager, 1.5 k captured over
        101 iterations; no gain at 41 -- capture and instantiation cost about 15 ms per call.
        graph=True is EXPERIMENTAL: it is covered by the tests (tests/test_gpu_api.py) and used by tools/bench_opbyop.py, but a graph
        kept ACROSS calls faulted in one call sequence for a reason that was not found (HISTORY.md round 6, item 4), which is why the
        graph lives for one call only."""
        self.losses_values = _LossLog()
        self.optimization_results = []
        self._refresh_gt()
        builtin = all(f in _BUILTIN_LOSSES for f in self.loss_functions) and len(self.loss_functions) > 0
        if fused is None:
            fused = builtin
        if fused and not builtin:
            raise RuntimeError("the fused engine only knows l1_rgb_with_mask / l1_depth_with_mask / l1_mask / l1_edge")
        if fused:
            self._fused_enqueue(optimizer, global_batch)
            if wait:
                self.finish_optimization()
        else:
            self._run_autograd(bool(graph))

    def finish_optimization(self):
        """Synchronise with a run_optimization(wait=False) and fetch its results (no-op otherwise)."""
        if getattr(self, "_pending", None) is not None:
            self._fused_collect()

    def _fused_prepare(self, optimizer, global_batch, shade_slices=0, edge_slices=0, separate_big_pass=False):
        """The fused engine for this object, observation and schedule, ready at iteration 0 (built, or the previous run's engine
        with the new observation copied in); nothing is launched.  Returns (engine, params tensor, weights)."""
        r = self.object3d.mesh()
        lw = self.cfg.losses
        weights = {}
        for f in self.loss_functions:
            k = _BUILTIN_LOSSES[f]
            weights[k] = float({"rgb": lw.weight_rgb, "depth": lw.weight_depth, "mask": lw.weight_mask,
                                "edge": lw.get("weight_edge", 1.0)}[k])
        params = self.object3d.params_tensor()
        gt = {k: v[0] for k, v in self.gt_tensors.items()}
        tex = dict(uv=r["uv"][0], tex=r["tex"][0]) if self.object3d.mesh.has_textured_map else dict(vtx_color=r["vtx_color"][0])
        # the engine of the previous run is kept while everything but the observation and the initial poses is the same (the
        # same object in the next frame, or a second run on the same frame): the mesh half of its set-up -- sorted copies,
        # meshlets, triangle / texel records, closedness analysis, ~half of a 100-iteration call -- is then not repeated
        from .render import _buffer_key

        mesh_t = [r["pos"], r["pos_idx"], self.camera.cam_proj] + list(tex.values())
        sig = (tuple(_buffer_key(t) for t in mesh_t), tuple(self.resolution), self.batchsize, tuple(sorted(weights.items())), optimizer,
               global_batch, len(self.lr_schedule()), tuple(sorted((k, tuple(v.shape)) for k, v in gt.items())), shade_slices, edge_slices,
               bool(self.cfg.hyperparameters.get("cull_backfaces", False)), self.cfg.hyperparameters.get("compat"), bool(separate_big_pass))
        cached = getattr(self, "_engine_cache", None)
        if cached is not None and cached[0] == sig and getattr(self, "_pending", None) is None:
            eng = cached[1]
            eng.new_observation(gt=gt, params=params, lr_mult=self.learning_rates, lr_sched=self.lr_schedule())
            params = eng.params
        else:
            # cfg.hyperparameters.cull_backfaces and .compat reach the engine from the config.  THIS API draws both faces of every
            # triangle unless asked, as dr.rasterize does (diffdope.py:198-200): skipping the back faces of a closed mesh (deviation
            # D5, RefineEngine's own default: +9 % on cfg2) is invisible in exact arithmetic, but in float32 a back-facing sliver on
            # the silhouette can win a pixel -- tools/cull_sweep.py: 1 351 of 8 000 random hypotheses differ in some bit of their
            # losses or gradient between the two rules (profiles/r5c_cull_sweep.json).  .compat default None; "nvdiffrast": deviation
            # D2 switched to nvdiffrast's rule
            hp = self.cfg.hyperparameters
            eng = RefineEngine(r["pos"][0], r["pos_idx"][0], self.camera.cam_proj[0], self.resolution, gt, params, self.learning_rates,
                               self.lr_schedule(), weights, optimizer=optimizer, global_batch=global_batch, shade_slices=shade_slices,
                               edge_slices=edge_slices, cull_backfaces=bool(hp.get("cull_backfaces", False)), compat=hp.get("compat"),
                               separate_big_pass=separate_big_pass, **tex)
            self._engine_cache = (sig, eng, mesh_t)  # (mesh_t: keeps the keyed buffers alive, see render._buffer_key)
        return eng, params, weights

    def _fused_enqueue(self, optimizer, global_batch):
        eng, params, weights = self._fused_prepare(optimizer, global_batch)
        eng.run()
        self._pending = (eng, params, weights, torch.cuda.current_stream())

    def _fused_collect(self):
        eng, params, weights, stream = self._pending
        self._pending = None
        stream.synchronize()
        eng.check()
        self.object3d.load_params_tensor(params)
        losses = eng.losses().cpu()
        for i, k in enumerate(("rgb", "depth", "mask", "edge")):
            if k in weights:
                self.losses_values[_LOG_KEYS[k]] = losses[:, i].clone()
        mtx = eng.mtx_log.reshape(eng.max_iters, self.batchsize, 4, 4).cpu()
        self.optimization_results = [_LazyResult(mtx[i], self._render_cpu) for i in range(mtx.shape[0])]
        self.last_engine = eng

    def _iteration(self, lr):
        """One pass of the loop body (diffdope.py:1656-1714).  `lr`: a Python float (eager: through torch.optim.SGD, as the reference
        steps) or a one-element device tensor (captured: the same update p <- p - lr g written as tensor operations)."""
        prms = self._sgd_params()
        for prm in prms:  # (optimizer.zero_grad(): set_to_none)
            prm.grad = None
        result = self.object3d()
        mtx_gu = matrix_batch_44_from_position_quat(p=result["trans"], q=result["quat"])
        self.renders = self._render(mtx_gu, outputs=self._loop_outputs(), compact_mask=self._builtin_losses_only())
        loss = None  # (the reference starts from torch.zeros(1): a fill, an addition and a reshaping in the backward for nothing)
        for loss_function in self.loss_functions:
            l = loss_function(self)
            if l is None:
                continue
            loss = l if loss is None else loss + l
        if loss is not None:
            loss.backward()
        if torch.is_tensor(lr):
            with torch.no_grad():  # (two launches for the seven parameters; p + (-1) (g lr), the rounding of the eager step)
                prms = [prm for prm in self.object3d.parameters() if prm.grad is not None]
                lr_b = self._capture["lr_b"]
                lr_b.copy_(lr.expand_as(lr_b))
                torch._foreach_addcmul_(prms, [prm.grad for prm in prms], [lr_b] * len(prms), value=-1.0)
        else:
            for g in self.optimizer.param_groups:
                g["lr"] = lr
            # torch.optim.SGD.step() without momentum, weight decay or nesterov is p <- p + (-lr) g in one multi-tensor launch
            # (torch/optim/sgd.py _multi_tensor_sgd); issued directly: the optimizer's Python wrapper costs 0.1 ms of an iteration
            # whose GPU work takes 0.6.  Any other optimizer state goes through step().
            g0 = self.optimizer.param_groups[0]
            plain = (type(self.optimizer) is torch.optim.SGD and len(self.optimizer.param_groups) == 1 and not g0.get("momentum") and not g0.get("weight_decay") and not g0.get("nesterov")
                     and not g0.get("maximize") and prms and prms[0].is_cuda)
            if plain:
                with torch.no_grad():
                    live = [prm for prm in prms if prm.grad is not None]
                    if live:
                        torch._foreach_add_(live, [prm.grad for prm in live], alpha=-float(lr))
            else:
                self.optimizer.step()
        return mtx_gu

    def _sgd_params(self):
        return [prm for g in self.optimizer.param_groups for prm in g["params"]]

    def _run_autograd(self, graph=False):
        hp = self.cfg.hyperparameters
        self.optimizer = torch.optim.SGD(self.object3d.parameters(), lr=hp.learning_rate_base)
        lrs = self.lr_schedule()
        n_it = len(lrs)
        if not graph or n_it <= 2:
            for lr in lrs:
                mtx_gu = self._iteration(lr)
                self.optimization_results.append(_LazyResult(mtx_gu.detach(), self._render_cpu))  # (copied to the host when read)
            if n_it and self._loop_outputs() is not None:
                with torch.no_grad():  # (the loop's renders were partial: the last iteration's images once more, complete)
                    self.renders = self._render(mtx_gu.detach())
            return
        # ---- two eager iterations (the allocator, the lazily built tables and the autograd nodes of the parameters settle) ON THE
        # STREAM THE CAPTURE WILL USE -- an AccumulateGrad node made on another stream would synchronise with it inside the capture --,
        # then the rest of the schedule as replays of ONE captured iteration
        n_eager = 2
        dev = self.object3d.qx.device
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.renders = None  # (an earlier run's autograd graph, made on another stream, goes with it)
        self._log_shapes = {}
        with torch.cuda.stream(side):
            try:
                for lr in lrs[:n_eager]:
                    mtx_gu = self._iteration(lr)
                    self.optimization_results.append(_LazyResult(mtx_gu.detach(), self._render_cpu))
            finally:
                shapes, self._log_shapes = self._log_shapes, None
            # everything that outlives an iteration is allocated HERE, outside the capture: a tensor created inside it may be given
            # the memory of a temporary freed earlier in the iteration, which every replay then scribbles over
            cap = dict(n_it=n_it, it=torch.full((1,), n_eager, dtype=torch.long, device=dev),
                       logs={k: torch.zeros((n_it,) + shp, dtype=dt, device=dev) for k, (shp, dt) in shapes.items()})
            cap["lr_b"] = torch.zeros_like(self.object3d.qx)  # (this iteration's learning rate, one copy per hypothesis)
            lr_table = torch.tensor(lrs, dtype=torch.float32, device=dev)
            mtx_log = torch.zeros((n_it,) + tuple(mtx_gu.shape), dtype=mtx_gu.dtype, device=dev)
            del mtx_gu
            self.renders = None  # (the last autograd graph goes with it)
        g = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)  # (the gradients become tensors of the graph's own pool)
        self._capture = cap
        try:
            with torch.cuda.graph(g, stream=side):
                mtx_gu = self._iteration(lr_table.index_select(0, cap["it"]))
                mtx_log.index_copy_(0, cap["it"], mtx_gu.detach()[None])
                cap["it"].add_(1)
        finally:
            self._capture = None
        torch.cuda.current_stream().wait_stream(side)
        # THE COUNTER.  The captured iteration indexes the learning-rate table, the loss-row buffers and the pose log with the
        # DEVICE-SIDE counter cap["it"], which the graph itself increments: exactly n_it - n_eager replays fit, and one more runs
        # index_select / index_copy_ past the ends of those buffers -- the index kernels answer with a trap (the queue aborts with
        # HSA_STATUS_ERROR_EXCEPTION 0x1016; tools/graph_fault_repro.py 8 6).  Nothing ran during the capture: the counter still
        # names the first iteration to replay.
        for _ in range(n_eager, n_it):
            g.replay()
        # THE GRAPH LIVES FOR THIS CALL ONLY (HISTORY.md round 6 item 4 has the whole investigation).  A graph kept for the next
        # call replays correctly when its counter is put back first -- three calls with the results read in between, six orders of
        # host operations, 0.75-0.83 ms per iteration on cfg2 (tools/graph_fault_repro.py, graph_fault_repro2.py) -- but in the call
        # sequence of tools/bench_opbyop.py the FIRST replay of a freshly captured graph faults on an address 15.8 GB above the
        # rasteriser's scratch whenever the graph OR the tensors it was captured with (tables, counter, capture stream) are stored on
        # this object between the capture and that replay (five variants; storing an unrelated object does not do it; without the
        # store, and in the round-5 tree, the same sequence passes; the same iteration run eagerly in its captured form passes; no
        # host-to-device copy and no BLAS call is among the captured nodes).  A fault that depends on which host objects stay
        # referenced is not explained, so nothing is replayed once this call has returned.
        torch.cuda.current_stream().synchronize()
        mtx_rows = mtx_log[n_eager:].clone()
        for i in range(n_it - n_eager):
            self.optimization_results.append(_LazyResult(mtx_rows[i], self._render_cpu))
        if not isinstance(self.losses_values, _LossLog):
            log_, self.losses_values = self.losses_values, _LossLog()
            self.losses_values.update(log_)
        for key, buf in cap["logs"].items():
            rows = buf[n_eager:].clone()
            for i in range(n_it - n_eager):
                self.losses_values.add(key, rows[i])
        if os.environ.get("DDX_DEBUG_KEEP_GRAPH"):  # (tools/graph_fault_repro.py only: the graph and everything it was captured with stay alive)
            self._kept_graph = dict(g=g, cap=cap, lr_table=lr_table, mtx_log=mtx_log, side=side, renders=self.renders, n_eager=n_eager)
            return
        self.renders = None
        self.optimizer.zero_grad(set_to_none=True)
        del g
        with torch.no_grad():  # (the reference leaves the last iteration's images in self.renders: rendered again, outside the graph)
            self.renders = self._render(mtx_rows[-1])

    def cuda(self):
        self.object3d.cuda()
        self.scene.cuda()
        self.camera.cuda()
