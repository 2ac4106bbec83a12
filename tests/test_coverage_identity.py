"""The integer identities the rasteriser's coverage routine relies on (raster_dev.h small_mask_rel, round 6), checked on the host
against the plain form the oracle uses (oracle/ddx_oracle.c tri_covers / raster_math.h edge_inside): for a snapped triangle (a, b, c)
and a pixel centre P

  * the three edge functions of P sum to the area, exactly:  e(b->c) + e(c->a) + e(a->b) = (b - a) x (c - a);
  * a triangle of negative area is covered exactly where (a, c, b) is: its flipped edges are the same lines walked the other way,
    the ownership rule (dy > 0 || (dy == 0 && dx < 0)) of a reversed edge is the complement of the original's;
  * the ownership rule folds into the start value:  v > 0 || (v == 0 && own)  <=>  v + own - 1 >= 0.

Coordinates are drawn from a coarse grid so that centres ON an edge -- the cases the ownership rule exists for -- are common."""
import numpy as np

SUB = 256


def edge_inside(ax, ay, bx, by, px, py, flip):
    dx, dy = bx - ax, by - ay
    e = dx * (py - ay) - dy * (px - ax)
    if flip:
        e, dx, dy = -e, -dx, -dy
    if e > 0:
        return True
    if e < 0:
        return False
    return dy > 0 or (dy == 0 and dx < 0)


def covers_plain(a, b, c, px, py):
    """oracle form: the three edges with the flip of a negative area"""
    area = (b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1])
    if area == 0:
        return False
    flip = area < 0
    PX, PY = px * SUB + SUB // 2, py * SUB + SUB // 2
    return (edge_inside(b[0], b[1], c[0], c[1], PX, PY, flip) and edge_inside(c[0], c[1], a[0], a[1], PX, PY, flip)
            and edge_inside(a[0], a[1], b[0], b[1], PX, PY, flip))


def mask_rel(a, b, c, px0, py0, nxp, nyp):
    """small_mask_rel: corners relative to a, in the order of positive area; two edge functions multiplied out, the third from the area"""
    abx, aby, acx, acy = b[0] - a[0], b[1] - a[1], c[0] - a[0], c[1] - a[1]
    if abx * acy - acx * aby < 0:
        abx, aby, acx, acy = acx, acy, abx, aby
    area = abx * acy - acx * aby
    assert area > 0
    apx, apy = px0 * SUB + SUB // 2 - a[0], py0 * SUB + SUB // 2 - a[1]
    f2 = abx * apy - aby * apx
    f1 = acy * apx - acx * apy
    f0 = area - f1 - f2
    d0x, d0y = acx - abx, acy - aby
    b0 = f0 + int(d0y > 0 or (d0y == 0 and d0x < 0)) - 1
    b1 = f1 + int(acy < 0 or (acy == 0 and acx > 0)) - 1
    b2 = f2 + int(aby > 0 or (aby == 0 and abx < 0)) - 1
    sx2, sy2, sx1, sy1 = -aby * SUB, abx * SUB, acy * SUB, -acx * SUB
    sx0, sy0 = -(sx1 + sx2), -(sy1 + sy2)
    mask = 0
    for j in range(nyp):
        for i in range(nxp):
            v0, v1, v2 = b0 + i * sx0 + j * sy0, b1 + i * sx1 + j * sy1, b2 + i * sx2 + j * sy2
            if v0 >= 0 and v1 >= 0 and v2 >= 0:  # ((v0 | v1 | v2) >= 0 on two's-complement integers)
                mask |= 1 << (j * nxp + i)
    # every factor of the device's 24-bit multiplies stays inside 24 bits, every sum inside 32
    for f in (abx, aby, acx, acy, apx, apy):
        assert abs(f) < 1 << 23
    for v in (f0, f1, f2, b0 + (nxp - 1) * sx0 + (nyp - 1) * sy0, b1 + (nxp - 1) * sx1 + (nyp - 1) * sy1, b2 + (nxp - 1) * sx2 + (nyp - 1) * sy2):
        assert abs(v) < 1 << 31
    return mask


def test_relative_corner_coverage_equals_the_three_flipped_edge_functions():
    rng = np.random.RandomState(7)
    n_cases = n_tie = n_cov = 0
    for grid in (16, 32, 64, 128, 256, 1):  # coarse grids: vertices and edges through pixel centres are common
        for _ in range(2500):
            a = rng.randint(-2000, 6000, 2) // grid * grid + (SUB // 2 if rng.rand() < 0.5 else 0)
            ext = int(rng.choice([300, 600, 1500, 4000, 8191]))
            b = a + rng.randint(-ext, ext + 1, 2) // grid * grid
            c = a + rng.randint(-ext, ext + 1, 2) // grid * grid
            pts = np.stack([a, b, c])
            if (pts.max(0) - pts.min(0)).max() >= 8192:  # (a small triangle spans < 2^13 sub-pixels: scatter_one)
                continue
            area = int((b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1]))
            if area == 0:
                continue
            # the pixel box the rasteriser gives the routine: centres inside the snapped bounding box, clamped to the frame, <= 64 of them
            W = H = 48
            px0, px1 = max((int(pts[:, 0].min()) + 127) >> 8, 0), min((int(pts[:, 0].max()) - 128) >> 8, W - 1)
            py0, py1 = max((int(pts[:, 1].min()) + 127) >> 8, 0), min((int(pts[:, 1].max()) - 128) >> 8, H - 1)
            if px0 > px1 or py0 > py1 or (px1 - px0 + 1) * (py1 - py0 + 1) > 64:
                continue
            nxp, nyp = px1 - px0 + 1, py1 - py0 + 1
            A, B_, C = (int(a[0]), int(a[1])), (int(b[0]), int(b[1])), (int(c[0]), int(c[1]))
            mask = mask_rel(A, B_, C, px0, py0, nxp, nyp)
            for j in range(nyp):
                for i in range(nxp):
                    want = covers_plain(A, B_, C, px0 + i, py0 + j)
                    assert bool((mask >> (j * nxp + i)) & 1) == want, (A, B_, C, px0 + i, py0 + j)
                    n_cov += want
                    PX, PY = (px0 + i) * SUB + SUB // 2, (py0 + j) * SUB + SUB // 2
                    n_tie += any((q[0] - p[0]) * (PY - p[1]) - (q[1] - p[1]) * (PX - p[0]) == 0 for p, q in ((B_, C), (C, A), (A, B_)))
            n_cases += 1
    assert n_cases > 3000 and n_cov > 3000 and n_tie > 300, (n_cases, n_cov, n_tie)
