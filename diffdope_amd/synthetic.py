"""Seeded synthetic scenes for tests and bench.py (SURVEY.md section 8d "Synthetic inputs").

Pure numpy, no I/O: a closed lat-long blob mesh with an exact triangle count T = 2*rows*cols,
per-vertex uv (seam vertices duplicated, like a PLY with texture_u/texture_v), a band-limited
procedural texture, the reference's example camera (configs/diffdope.yaml:3-8) scaled to the
requested resolution, and pose helpers.
"""
import math

import numpy as np

YAML_CAMERA = dict(fx=1390.53, fy=1386.99, cx=964.957, cy=522.586, im_width=1920, im_height=1080)


def blob_mesh(rows, cols, seed=0, radius=0.5, noise=0.18):
    """Returns pos [V,3] f32, tri [T,3] i32, uv [V,2] f32 with V=(rows+1)*(cols+1), T=2*rows*cols."""
    rng = np.random.RandomState(seed)
    th = np.linspace(0.0, math.pi, rows + 1)
    ph = np.linspace(0.0, 2.0 * math.pi, cols + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    r = np.ones_like(TH)
    for _ in range(6):
        k_t, k_p = rng.randint(1, 4), rng.randint(0, 4)
        a, ph0 = rng.uniform(-1, 1), rng.uniform(0, 2 * math.pi)
        r += noise / 3.0 * a * np.sin(TH) ** 2 * np.sin(k_t * TH) * np.cos(k_p * PH + ph0)
    ax = rng.uniform(0.7, 1.0, size=3)
    x = radius * ax[0] * r * np.sin(TH) * np.cos(PH)
    y = radius * ax[1] * r * np.sin(TH) * np.sin(PH)
    z = radius * ax[2] * r * np.cos(TH)
    # a CLOSED surface, bit for bit: the uv seam duplicates the first column of vertices (sin / cos of 2 pi are not exactly
    # those of 0) and every vertex of a pole row is the pole itself -- like the duplicated seam vertices of a textured scan
    for a in (x, y, z):
        a[:, -1] = a[:, 0]
    x[0, :] = 0.0; y[0, :] = 0.0; z[0, :] = z[0, 0]
    x[-1, :] = 0.0; y[-1, :] = 0.0; z[-1, :] = z[-1, 0]
    pos = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    uv = np.stack([PH / (2 * math.pi), TH / math.pi], axis=-1).reshape(-1, 2).astype(np.float32)
    idx = np.arange((rows + 1) * (cols + 1)).reshape(rows + 1, cols + 1)
    a, b = idx[:-1, :-1], idx[:-1, 1:]
    c, d = idx[1:, :-1], idx[1:, 1:]
    tri = np.stack([np.stack([a, c, b], -1), np.stack([b, c, d], -1)], axis=2).reshape(-1, 3).astype(np.int32)
    return pos, tri, uv


def vertex_colors(pos, seed=5):
    rng = np.random.RandomState(seed)
    f = rng.uniform(2.0, 6.0, size=(3, 3))
    p = rng.uniform(0, 2 * math.pi, size=3)
    col = 0.5 + 0.5 * np.sin(pos @ f.T * 2 * math.pi + p)
    return col.astype(np.float32)


def texture(size, seed=1, channels=3):
    """Band-limited noise in [0,1], tileable, [size,size,channels] f32."""
    rng = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.arange(size) / size, np.arange(size) / size, indexing="ij")
    img = np.zeros((size, size, channels), np.float64)
    for c in range(channels):
        for _ in range(8):
            kx, ky = rng.randint(1, 12, size=2)
            a, p = rng.uniform(0.3, 1.0), rng.uniform(0, 2 * math.pi)
            img[..., c] += a * np.sin(2 * math.pi * (kx * xx + ky * yy) + p)
    img -= img.min(axis=(0, 1), keepdims=True)
    img /= img.max(axis=(0, 1), keepdims=True)
    return img.astype(np.float32)


def camera_intrinsics(W, H):
    """The yaml camera scaled to W x H (same field of view)."""
    sx, sy = W / YAML_CAMERA["im_width"], H / YAML_CAMERA["im_height"]
    return dict(
        fx=YAML_CAMERA["fx"] * sx, fy=YAML_CAMERA["fy"] * sy, cx=YAML_CAMERA["cx"] * sx, cy=YAML_CAMERA["cy"] * sy,
        im_width=W, im_height=H,
    )


def quat_from_axis_angle(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    s = math.sin(angle / 2)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle / 2)])


def quat_mul(a, b):
    """Hamilton product, xyzw layout."""
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
        aw * bw - ax * bx - ay * by - az * bz,
    ])


def random_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def perturb_pose(q, t, rot_deg, trans_frac, rng):
    """Rotate by rot_deg about a random axis, translate by trans_frac*|t| in a random direction."""
    axis = rng.normal(size=3)
    q2 = quat_mul(quat_from_axis_angle(axis, math.radians(rot_deg)), q)
    dirv = rng.normal(size=3)
    dirv /= np.linalg.norm(dirv)
    return q2, np.asarray(t, np.float64) + dirv * trans_frac * np.linalg.norm(t)


def rotation_geodesic(q1, q2):
    """Angle (rad) between two xyzw quaternions."""
    q1 = np.asarray(q1, np.float64) / np.linalg.norm(q1)
    q2 = np.asarray(q2, np.float64) / np.linalg.norm(q2)
    return 2.0 * math.acos(min(1.0, abs(float(np.dot(q1, q2)))))


def matrix_rotation_geodesic(R1, R2):
    c = (np.trace(np.asarray(R1, np.float64).T @ np.asarray(R2, np.float64)) - 1.0) / 2.0
    return math.acos(max(-1.0, min(1.0, c)))
