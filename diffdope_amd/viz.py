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
