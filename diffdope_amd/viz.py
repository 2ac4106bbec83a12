"""Presentation helpers (diffdope/diffdope.py:242-528,1377-1616 are cv2/imageio/matplotlib code off the timed
path): numpy-only overlays and grids, GIF animation through PIL, loss curves through matplotlib (Agg)."""
import numpy as np


def find_crop(mask_hw, pad=0.1):
    """Square crop (row0, col0, size) around the non-zero part of a [H,W(,C)] mask tensor/array."""
    m = np.asarray(mask_hw)
    if m.ndim == 3:
        m = m[..., 0]
    ys, xs = np.nonzero(m > 0)
    if len(ys) == 0:
        return 0, 0, min(m.shape) - 1
    size = int(max(ys.max() - ys.min(), xs.max() - xs.min()) * (1 + 2 * pad)) + 1
    size = min(size, min(m.shape) - 1)
    r0 = int(np.clip((ys.min() + ys.max()) // 2 - size // 2, 0, m.shape[0] - 1 - size))
    c0 = int(np.clip((xs.min() + xs.max()) // 2 - size // 2, 0, m.shape[1] - 1 - size))
    return r0, c0, size


def overlay(background, foreground, alpha=0.7, add_background=True):
    """[H,W,3] float images in 0..1: the render where it is non-black, blended over the observed image."""
    fg = np.asarray(foreground, np.float32)
    if fg.ndim == 2:
        fg = np.repeat((fg / max(float(fg.max()), 1e-6))[..., None], 3, -1)
    if not add_background or background is None:
        return np.clip(fg, 0, 1)
    bg = np.asarray(background, np.float32)
    if bg.ndim == 2:
        bg = np.repeat((bg / max(float(bg.max()), 1e-6))[..., None], 3, -1)
    cover = (fg.sum(-1, keepdims=True) > 0).astype(np.float32)
    return np.clip(bg * (1 - cover * alpha) + fg * cover * alpha, 0, 1)


def make_grid(images, nrow=4, pad=2):
    """List of [H,W,3] float images -> one [rows*H.., cols*W.., 3] grid (row-major, `nrow` images per row)."""
    n = len(images)
    nrow = max(1, min(nrow, n))
    rows = (n + nrow - 1) // nrow
    H, W = images[0].shape[:2]
    out = np.zeros((rows * (H + pad) + pad, nrow * (W + pad) + pad, 3), np.float32)
    for i, im in enumerate(images):
        r, c = divmod(i, nrow)
        out[pad + r * (H + pad): pad + r * (H + pad) + H, pad + c * (W + pad): pad + c * (W + pad) + W] = im
    return out


def to_uint8(img):
    return (np.clip(img, 0, 1) * 255).round().astype(np.uint8)


def save_gif(frames_uint8, path, fps=10):
    from PIL import Image as PILImage

    ims = [PILImage.fromarray(f) for f in frames_uint8]
    ims[0].save(path, save_all=True, append_images=ims[1:], duration=int(1000 / fps), loop=0)
    return path


def plot_losses(losses_values, batch_index, keys=None):
    """Loss curves of one hypothesis as an RGB uint8 image (diffdope.py:1573-1616)."""
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    fig = plt.figure(figsize=(10, 6))
    for key, t in losses_values.items():
        if keys is None or key in keys:
            plt.plot(np.asarray(t)[:, batch_index], marker="o", label=key)
    plt.legend()
    fig.canvas.draw()
    img = np.asarray(fig.canvas.buffer_rgba())[..., :3].copy()
    plt.close(fig)
    return img


# ------------------------------------------------------------------------------------------------
# The reference's module-level helpers by name (diffdope/diffdope.py:243-528; `from .diffdope import *` exposes them as
# diffdope.<name>): same arguments, numpy/PIL inside instead of cv2/torchvision code.
def contour(mask_hw):
    """Boundary pixels of a boolean [H,W] mask (the pixels of the mask with a 4-neighbour outside it)."""
    m = np.asarray(mask_hw, bool)
    p = np.pad(m, 1)
    inner = p[:-2, 1:-1] & p[2:, 1:-1] & p[1:-1, :-2] & p[1:-1, 2:]
    return m & ~inner


def im_resize(image, width=None, height=None):
    """Aspect-preserving resize of a [H,W(,C)] uint8 image to the given width OR height (diffdope.py:313-334)."""
    from PIL import Image as PILImage

    h, w = image.shape[:2]
    if width is None and height is None:
        return image
    if width is None:
        width = max(1, int(round(w * height / float(h))))
    else:
        height = max(1, int(round(h * width / float(w))))
    return np.asarray(PILImage.fromarray(np.ascontiguousarray(image)).resize((int(width), int(height)), PILImage.BILINEAR))


def getimg_stack(color_imgs, depth=False, depth_max=3, w=1, h=1):
    """List of [H,W,3] (or [H,W] depth) float tensors/arrays -> one uint8 image of h rows x w columns (diffdope.py:277-310)."""
    ims = []
    for im in color_imgs:
        a = np.array(im.detach().cpu().numpy() if hasattr(im, "detach") else im, np.float32)
        if depth:
            a = np.where(a < 0, depth_max, a) / float(depth_max)
            a = np.repeat(a[..., None], 3, -1) if a.ndim == 2 else a
        ims.append(np.clip(a, 0, 1))
    rows = [np.concatenate(ims[r * w:(r + 1) * w], axis=1) for r in range(h) if ims[r * w:(r + 1) * w]]
    return to_uint8(np.concatenate(rows, axis=0))


def make_grid_image(img_batch, row, final_width, depth=False):
    """[B,H,W,3] float batch -> uint8 grid with `row` images per row, resized to final_width (diffdope.py:446-461)."""
    a = np.asarray(img_batch.detach().cpu().numpy() if hasattr(img_batch, "detach") else img_batch, np.float32)
    if depth:
        a = a / max(float(a.max()), 1e-6)
    grid = to_uint8(make_grid([a[i] for i in range(a.shape[0])], nrow=row))
    return im_resize(grid, width=final_width)


def make_grid_overlay_batch(foreground, background=None, alpha=0.5, row=2, final_width=2000, add_background=True, add_contour=True,
                            color_countour=[1, 0, 0], flip_result=True):
    """Grid of a batch of renders blended over the observed images, with the render's silhouette contour drawn
    (diffdope.py:464-528; argument names as in the reference, typo included)."""
    fg = np.asarray(foreground.detach().cpu().numpy() if hasattr(foreground, "detach") else foreground, np.float32)
    bg = None if background is None else np.asarray(background.detach().cpu().numpy() if hasattr(background, "detach") else background, np.float32)
    tiles = []
    for b in range(fg.shape[0]):
        im = overlay(None if bg is None else bg[b], fg[b], alpha=alpha, add_background=add_background and bg is not None)
        if add_contour:
            im = im.copy()
            im[contour(fg[b].sum(-1) > 0)] = np.asarray(color_countour, np.float32)
        tiles.append(im[::-1] if flip_result else im)
    return im_resize(to_uint8(make_grid(tiles, nrow=row)), width=final_width)


def make_grid_tensor(tensor, nrow=8, padding=2, normalize=False, value_range=None, scale_each=False, pad_value=0.0):
    """torchvision-style grid as the reference's module-level make_grid (diffdope.py:337-443): [B,C,H,W] tensor (or a
    list of [C,H,W]) -> [C, rows*(H+padding)+padding, cols*(W+padding)+padding] tensor, `nrow` images per row."""
    import torch

    t = torch.stack(list(tensor), 0) if isinstance(tensor, (list, tuple)) else tensor
    if t.dim() == 3:
        t = t[None]
    t = t.detach().float().clone()
    if normalize:
        def norm(x):
            lo, hi = (float(x.min()), float(x.max())) if value_range is None else value_range
            return (x.clamp(lo, hi) - lo) / max(hi - lo, 1e-5)
        t = torch.stack([norm(x) for x in t], 0) if scale_each else norm(t)
    B, C, H, W = t.shape
    cols = max(1, min(nrow, B))
    rows = (B + cols - 1) // cols
    out = torch.full((C, rows * (H + padding) + padding, cols * (W + padding) + padding), float(pad_value), dtype=t.dtype, device=t.device)
    for i in range(B):
        r, c = divmod(i, cols)
        out[:, padding + r * (H + padding): padding + r * (H + padding) + H, padding + c * (W + padding): padding + c * (W + padding) + W] = t[i]
    return out
