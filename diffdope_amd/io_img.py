"""Image loading without OpenCV (diffdope/diffdope.py:1122-1153 uses cv2.imread / cvtColor / flip / resize).
PIL decodes; resizing re-implements cv2.resize's INTER_LINEAR (half-pixel centres, no antialiasing) and
INTER_NEAREST index rules so that the tensors match what the reference would hold."""
import numpy as np


def imread_rgb(path):
    """8-bit image -> float64 [H,W,3] in 0..1 (cv2.imread(path)[:,:,:3] -> BGR2RGB -> /255)."""
    from PIL import Image as PILImage

    im = PILImage.open(path)
    if im.mode in ("I;16", "I", "F"):
        arr = np.asarray(im).astype(np.float64)
        arr = np.repeat((arr / 257.0 if im.mode == "I;16" else arr)[..., None], 3, -1)  # cv2 would down-convert to 8 bit
        return np.floor(arr) / 255.0
    return np.asarray(im.convert("RGB")).astype(np.float64) / 255.0


def imread_depth(path):
    """cv2.imread(path, IMREAD_UNCHANGED): the raw integer depth, float64 [H,W]."""
    from PIL import Image as PILImage

    im = PILImage.open(path)
    arr = np.asarray(im)
    if arr.ndim == 3:
        arr = arr[..., 0]
    return arr.astype(np.float64)


def resize_linear(im, out_w, out_h):
    """cv2.resize(im, (out_w, out_h)) with the default INTER_LINEAR."""
    H, W = im.shape[:2]

    def axis(n_out, n_in):
        s = n_in / n_out
        x = (np.arange(n_out) + 0.5) * s - 0.5
        x0 = np.floor(x).astype(np.int64)
        f = x - x0
        lo = x0 < 0
        f[lo], x0[lo] = 0.0, 0
        hi = x0 >= n_in - 1
        f[hi], x0[hi] = 0.0, n_in - 1
        x1 = np.minimum(x0 + 1, n_in - 1)
        return x0, x1, f

    y0, y1, fy = axis(out_h, H)
    x0, x1, fx = axis(out_w, W)
    im = im.astype(np.float64)
    sh = (-1,) + (1,) * (im.ndim - 1)
    fxr = fx.reshape((1, -1) + (1,) * (im.ndim - 2))
    top = im[y0][:, x0] * (1 - fxr) + im[y0][:, x1] * fxr
    bot = im[y1][:, x0] * (1 - fxr) + im[y1][:, x1] * fxr
    fyr = fy.reshape(sh)
    return top * (1 - fyr) + bot * fyr


def resize_nearest(im, out_w, out_h):
    """cv2.resize(..., interpolation=INTER_NEAREST): src = min(floor(dst * scale), n-1)."""
    H, W = im.shape[:2]
    ys = np.minimum(np.floor(np.arange(out_h) * (H / out_h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * (W / out_w)).astype(np.int64), W - 1)
    return im[ys][:, xs]
