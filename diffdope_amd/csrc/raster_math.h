// raster_math.h -- per-pixel / per-triangle device math shared by the rasteriser, the renderer ops
// and the fused engine.  Semantics: SURVEY.md section 2.2 (nvdiffrast ops as used at
// diffdope/diffdope.py:198-231).  Everything that decides a triangle id (snapping, coverage,
// depth key) is exact integer arithmetic or explicitly ordered fp32 (this library is built with
// -ffp-contract=off), so visibility is reproducible bit for bit.
#pragma once
#include <limits.h>

#include "ddx_common.h"

#define DDX_SUBPIX 256  // 8 sub-pixel bits
#define DDX_TILE 16     // screen tile edge in pixels (one 256-thread workgroup per tile)

struct float4a { float x, y, z, w; };

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- snapped window coordinates ---------------------------------------------------------------
__device__ __forceinline__ int snap_coord(float ndc, int dim)
{
    float f = __fmaf_rn(ndc, (float)(dim * (DDX_SUBPIX / 2)), (float)(dim * (DDX_SUBPIX / 2)));
    const float lim = 16777216.0f;
    if (!(f > -lim)) f = -lim;
    if (f > lim) f = lim;
    return (int)rintf(f);
}

struct SnapTri {
    int X[3], Y[3];
    long long area;
    bool ok;
};

__device__ __forceinline__ void snap_triangle(const float4& p0, const float4& p1, const float4& p2, int H, int W,
                                              SnapTri& s)
{
    s.ok = false;
    if (!(p0.w > 0.f) || !(p1.w > 0.f) || !(p2.w > 0.f)) return;
    const float i0 = __fdiv_rn(1.0f, p0.w), i1 = __fdiv_rn(1.0f, p1.w), i2 = __fdiv_rn(1.0f, p2.w);
    s.X[0] = snap_coord(p0.x * i0, W); s.Y[0] = snap_coord(p0.y * i0, H);
    s.X[1] = snap_coord(p1.x * i1, W); s.Y[1] = snap_coord(p1.y * i1, H);
    s.X[2] = snap_coord(p2.x * i2, W); s.Y[2] = snap_coord(p2.y * i2, H);
    s.area = (long long)(s.X[1] - s.X[0]) * (long long)(s.Y[2] - s.Y[0]) -
             (long long)(s.X[2] - s.X[0]) * (long long)(s.Y[1] - s.Y[0]);
    s.ok = s.area != 0;
}

// per-vertex snap (x = INT_MIN marks a vertex with w <= 0) and triangle setup from snapped vertices
__device__ __forceinline__ int2 snap_vertex(const float4& p, int H, int W)
{
    if (!(p.w > 0.f)) return make_int2(INT_MIN, 0);
    const float iw = __fdiv_rn(1.0f, p.w);
    return make_int2(snap_coord(p.x * iw, W), snap_coord(p.y * iw, H));
}

__device__ __forceinline__ void snap_from_vertices(const int2& a, const int2& b, const int2& c, SnapTri& s)
{
    s.ok = false;
    if (a.x == INT_MIN || b.x == INT_MIN || c.x == INT_MIN) return;
    s.X[0] = a.x; s.Y[0] = a.y; s.X[1] = b.x; s.Y[1] = b.y; s.X[2] = c.x; s.Y[2] = c.y;
    s.area = (long long)(s.X[1] - s.X[0]) * (long long)(s.Y[2] - s.Y[0]) -
             (long long)(s.X[2] - s.X[0]) * (long long)(s.Y[1] - s.Y[0]);
    s.ok = s.area != 0;
}

// Near-plane clipping of a triangle with a vertex at w <= 0 (dr.rasterize clips against the view volume): the part with
// z + w >= 0 is a polygon of 3 or 4 corners (Sutherland-Hodgman over the edges 0->1->2->0; an intersection is always
// computed from the inside end of its edge towards the outside end, so two triangles sharing the edge get the same point),
// snapped like vertices and cut into a fan of one or two triangles.  Fragments take barycentrics and depth from the
// ORIGINAL triangle (the homogeneous form of pixel_bary holds for any w).  Same operations, in the same order, as the
// oracle's clip_near.  Returns the number of triangles written to out[] (0: nothing in front of the near plane).
__device__ __forceinline__ int clip_near(const float4& p0, const float4& p1, const float4& p2, int H, int W, SnapTri out[2])
{
    const float4 p[3] = {p0, p1, p2};
    float d[3];
    bool in[3];
    int nin = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d[i] = p[i].z + p[i].w;
        in[i] = d[i] >= 0.f;
        nin += in[i];
    }
    out[0].ok = false; out[1].ok = false;
    if (nin == 0) return 0;
    int QX[4] = {0, 0, 0, 0}, QY[4] = {0, 0, 0, 0};
    int n = 0;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3;
        if (in[i]) {
            if (!(p[i].w > 0.f)) bad = true;
            const float iw = __fdiv_rn(1.0f, p[i].w);
            const int qx = snap_coord(p[i].x * iw, W), qy = snap_coord(p[i].y * iw, H);
            if (n == 0) { QX[0] = qx; QY[0] = qy; } else if (n == 1) { QX[1] = qx; QY[1] = qy; } else if (n == 2) { QX[2] = qx; QY[2] = qy; } else { QX[3] = qx; QY[3] = qy; }
            ++n;
        }
        if (in[i] != in[j]) {
            const float4 a = in[i] ? p[i] : p[j], b = in[i] ? p[j] : p[i];  // a inside, b outside
            const float da = in[i] ? d[i] : d[j], db = in[i] ? d[j] : d[i];
            const float t = __fdiv_rn(da, da - db);
            const float x = __fmaf_rn(t, b.x - a.x, a.x), y = __fmaf_rn(t, b.y - a.y, a.y), w = __fmaf_rn(t, b.w - a.w, a.w);
            if (!(w > 0.f)) bad = true;
            const float iw = __fdiv_rn(1.0f, w);
            const int qx = snap_coord(x * iw, W), qy = snap_coord(y * iw, H);
            if (n == 0) { QX[0] = qx; QY[0] = qy; } else if (n == 1) { QX[1] = qx; QY[1] = qy; } else if (n == 2) { QX[2] = qx; QY[2] = qy; } else { QX[3] = qx; QY[3] = qy; }
            ++n;
        }
    }
    if (bad) return 0;
    int m = 0;
#pragma unroll
    for (int k = 1; k <= 2; ++k) {
        if (k + 1 >= n) break;
        SnapTri s;
        s.X[0] = QX[0]; s.Y[0] = QY[0];
        s.X[1] = k == 1 ? QX[1] : QX[2]; s.Y[1] = k == 1 ? QY[1] : QY[2];
        s.X[2] = k == 1 ? QX[2] : QX[3]; s.Y[2] = k == 1 ? QY[2] : QY[3];
        s.area = (long long)(s.X[1] - s.X[0]) * (long long)(s.Y[2] - s.Y[0]) - (long long)(s.X[2] - s.X[0]) * (long long)(s.Y[1] - s.Y[0]);
        s.ok = s.area != 0;
        if (s.ok) {
            if (m == 0) out[0] = s; else out[1] = s;
            ++m;
        }
    }
    return m;
}

// pixel-centre range [px0,px1] x [py0,py1] whose centres can lie inside the snapped bbox (unclamped)
__device__ __forceinline__ void snap_bbox(const SnapTri& s, int& px0, int& py0, int& px1, int& py1)
{
    const int xmin = min(s.X[0], min(s.X[1], s.X[2])), xmax = max(s.X[0], max(s.X[1], s.X[2]));
    const int ymin = min(s.Y[0], min(s.Y[1], s.Y[2])), ymax = max(s.Y[0], max(s.Y[1], s.Y[2]));
    px0 = (xmin - DDX_SUBPIX / 2 + (DDX_SUBPIX - 1)) >> 8;  // arithmetic shift = floor
    px1 = (xmax - DDX_SUBPIX / 2) >> 8;
    py0 = (ymin - DDX_SUBPIX / 2 + (DDX_SUBPIX - 1)) >> 8;
    py1 = (ymax - DDX_SUBPIX / 2) >> 8;
}

__device__ __forceinline__ bool edge_inside(int ax, int ay, int bx, int by, int px, int py, bool flip)
{
    int dx = bx - ax, dy = by - ay;
    long long e = (long long)dx * (long long)(py - ay) - (long long)dy * (long long)(px - ax);
    if (flip) { e = -e; dx = -dx; dy = -dy; }
    if (e > 0) return true;
    if (e < 0) return false;
    return (dy > 0) || (dy == 0 && dx < 0);
}

__device__ __forceinline__ bool tri_covers(const SnapTri& s, int px, int py)
{
    const int PX = px * DDX_SUBPIX + DDX_SUBPIX / 2, PY = py * DDX_SUBPIX + DDX_SUBPIX / 2;
    const bool flip = s.area < 0;
    return edge_inside(s.X[1], s.Y[1], s.X[2], s.Y[2], PX, PY, flip) &&
           edge_inside(s.X[2], s.Y[2], s.X[0], s.Y[0], PX, PY, flip) &&
           edge_inside(s.X[0], s.Y[0], s.X[1], s.Y[1], PX, PY, flip);
}

// ---- perspective-correct barycentrics -----------------------------------------------------------
struct Bary {
    float u, v, zw, a0, a1, a2, s;
    float p0x, p0y, p1x, p1y, p2x, p2y, fx, fy;
};

__device__ __forceinline__ bool pixel_bary(const float4& p0, const float4& p1, const float4& p2, int px, int py, int H,
                                           int W, Bary& o)
{
    const float xs = __fdiv_rn(2.0f, (float)W), xo = __fdiv_rn(1.0f, (float)W) - 1.0f;
    const float ys = __fdiv_rn(2.0f, (float)H), yo = __fdiv_rn(1.0f, (float)H) - 1.0f;
    const float fx = __fmaf_rn((float)px, xs, xo), fy = __fmaf_rn((float)py, ys, yo);
    o.fx = fx; o.fy = fy;
    o.p0x = __fmaf_rn(-fx, p0.w, p0.x); o.p0y = __fmaf_rn(-fy, p0.w, p0.y);
    o.p1x = __fmaf_rn(-fx, p1.w, p1.x); o.p1y = __fmaf_rn(-fy, p1.w, p1.y);
    o.p2x = __fmaf_rn(-fx, p2.w, p2.x); o.p2y = __fmaf_rn(-fy, p2.w, p2.y);
    o.a0 = __fmaf_rn(o.p1x, o.p2y, -(o.p1y * o.p2x));
    o.a1 = __fmaf_rn(o.p2x, o.p0y, -(o.p2y * o.p0x));
    o.a2 = __fmaf_rn(o.p0x, o.p1y, -(o.p0y * o.p1x));
    o.s = (o.a0 + o.a1) + o.a2;
    if (!(fabsf(o.s) > 0.f)) return false;
    const float is = __fdiv_rn(1.0f, o.s);
    o.u = o.a0 * is;
    o.v = o.a1 * is;
    const float zn = __fmaf_rn(o.a2, p2.z, __fmaf_rn(o.a1, p1.z, o.a0 * p0.z));
    const float wn = __fmaf_rn(o.a2, p2.w, __fmaf_rn(o.a1, p1.w, o.a0 * p0.w));
    o.zw = __fdiv_rn(zn, wn) + 0.0f;
    return true;
}

// z/w only (what the depth test needs): same operations, in the same order, as pixel_bary computes them,
// with the pixel's NDC centre (fx, fy) supplied by the caller; false if the triangle is degenerate there
// or the fragment falls outside the near/far planes
__device__ __forceinline__ bool pixel_depth(const float4& p0, const float4& p1, const float4& p2, float fx, float fy, float& zw)
{
    const float p0x = __fmaf_rn(-fx, p0.w, p0.x), p0y = __fmaf_rn(-fy, p0.w, p0.y);
    const float p1x = __fmaf_rn(-fx, p1.w, p1.x), p1y = __fmaf_rn(-fy, p1.w, p1.y);
    const float p2x = __fmaf_rn(-fx, p2.w, p2.x), p2y = __fmaf_rn(-fy, p2.w, p2.y);
    const float a0 = __fmaf_rn(p1x, p2y, -(p1y * p2x));
    const float a1 = __fmaf_rn(p2x, p0y, -(p2y * p0x));
    const float a2 = __fmaf_rn(p0x, p1y, -(p0y * p1x));
    const float s = (a0 + a1) + a2;
    if (!(fabsf(s) > 0.f)) return false;
    const float zn = __fmaf_rn(a2, p2.z, __fmaf_rn(a1, p1.z, a0 * p0.z));
    const float wn = __fmaf_rn(a2, p2.w, __fmaf_rn(a1, p1.w, a0 * p0.w));
    zw = __fdiv_rn(zn, wn) + 0.0f;
    return zw >= -1.0f && zw <= 1.0f;
}

// pixel index -> NDC centre: fx = fma(px, xs, xo) with xs = 2/W, xo = 1/W - 1 (as pixel_bary)
struct PixNdc { float xs, xo, ys, yo; };
__device__ __forceinline__ PixNdc make_pixndc(int H, int W)
{
    PixNdc n;
    n.xs = __fdiv_rn(2.0f, (float)W); n.xo = __fdiv_rn(1.0f, (float)W) - 1.0f;
    n.ys = __fdiv_rn(2.0f, (float)H); n.yo = __fdiv_rn(1.0f, (float)H) - 1.0f;
    return n;
}

__device__ __forceinline__ unsigned int depth_key(float zw)
{
    unsigned int bits = __float_as_uint(zw);
    return bits ^ ((bits >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

__device__ __forceinline__ float clamp01(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

// gradient of (u,v) w.r.t. the clip-space (x,y,w) of the three vertices, given upstream (gu,gv).
// True derivative of the forward: a clamped component passes no gradient (DESIGN.md deviation D2).  unclamped = true
// (DDX_COMPAT_UNCLAMPED_BARY_GRAD): the derivative of the UNCLAMPED expression, as nvdiffrast's rasterize backward takes it.
__device__ __forceinline__ void bary_backward(const Bary& bc, float gu, float gv, float gx[3], float gy[3], float gw[3], bool unclamped = false)
{
    if (!unclamped && (bc.u < 0.f || bc.u > 1.f)) gu = 0.f;
    if (!unclamped && (bc.v < 0.f || bc.v > 1.f)) gv = 0.f;
    const float is = __fdiv_rn(1.0f, bc.s);
    const float k = gu * bc.u + gv * bc.v;
    const float A0 = (gu - k) * is, A1 = (gv - k) * is, A2 = (-k) * is;
    gx[0] = A1 * (-bc.p2y) + A2 * bc.p1y;  gy[0] = A1 * bc.p2x + A2 * (-bc.p1x);
    gx[1] = A0 * bc.p2y + A2 * (-bc.p0y);  gy[1] = A0 * (-bc.p2x) + A2 * bc.p0x;
    gx[2] = A0 * (-bc.p1y) + A1 * bc.p0y;  gy[2] = A0 * bc.p1x + A1 * (-bc.p0x);
#pragma unroll
    for (int i = 0; i < 3; ++i) gw[i] = -bc.fx * gx[i] - bc.fy * gy[i];
}

// ---- bilinear texture, wrap boundary --------------------------------------------------------------
struct TexelSetup { int x0, x1, y0, y1; float fx, fy; };

__device__ __forceinline__ void tex_setup(float u, float v, int Th, int Tw, TexelSetup& s)
{
    u = u - floorf(u);
    v = v - floorf(v);
    const float x = __fmaf_rn(u, (float)Tw, -0.5f), y = __fmaf_rn(v, (float)Th, -0.5f);
    const float xf = floorf(x), yf = floorf(y);
    s.fx = x - xf; s.fy = y - yf;
    int x0 = (int)xf, y0 = (int)yf;
    int x1 = x0 + 1, y1 = y0 + 1;
    if (x0 < 0) x0 += Tw;
    if (y0 < 0) y0 += Th;
    if (x0 >= Tw) x0 -= Tw;
    if (y0 >= Th) y0 -= Th;
    if (x1 >= Tw) x1 -= Tw;
    if (y1 >= Th) y1 -= Th;
    s.x0 = x0; s.x1 = x1; s.y0 = y0; s.y1 = y1;
}

// ---- antialias pair analysis ------------------------------------------------------------------------
struct AAPair {
    bool valid;
    int tri, va, vb, d;
    bool chosen1, clamped;
    float ds, dc, alpha;
    float xa, ya, xb, yb, fx, fy;
};

__device__ __forceinline__ bool sign_bit(float x) { return (__float_as_uint(x) >> 31) != 0; }

// det of the (x, y, w) rows of three clip-space points, in the oracle's operation order (aa_det3_xyw there): its sign is the
// orientation of the projections wherever those exist
__device__ __forceinline__ float aa_det3_xyw(const float4& a, const float4& b, const float4& c)
{
    const float m0 = b.y * c.w - c.y * b.w;
    const float m1 = b.x * c.w - c.x * b.w;
    const float m2 = b.x * c.y - c.x * b.y;
    return (a.x * m0 - a.y * m1) + a.w * m2;
}

// Analyse the pixel pair (px,py)-(px+1,py) [d=0] or (px,py)-(px,py+1) [d=1] given the triangle ids
// (0-based, -1 = background) and z/w of both pixels.  P = clip positions [V,4] of this hypothesis.
__device__ __forceinline__ void aa_eval_pair(const float* __restrict__ P, const int* __restrict__ tri,
                                             const int* __restrict__ opp, int H, int W, int px, int py, int d, int t0,
                                             int t1, float z0, float z1, AAPair& o)
{
    o.valid = false;
    if (t0 == t1) return;
    bool chosen1;
    if (t0 >= 0 && t1 >= 0) chosen1 = !(z0 < z1);
    else chosen1 = t0 < 0;
    const int t = chosen1 ? t1 : t0;
    const int cx = chosen1 ? px + (d == 0) : px, cy = chosen1 ? py + (d == 1) : py;
    const float ds = chosen1 ? -1.0f : 1.0f;
    const int vi[3] = {tri[t * 3 + 0], tri[t * 3 + 1], tri[t * 3 + 2]};
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const float fx = (float)cx + 0.5f - hw, fy = (float)cy + 0.5f - hh;
    float x[3], y[3], ox[3], oy[3];
    float4 pv[3];
    bool behind[3];
    int nbehind = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pv[i] = ld4(P + (size_t)vi[i] * 4);
        behind[i] = !(pv[i].w > 0.f);
        nbehind += behind[i] ? 1 : 0;
        x[i] = 0.f; y[i] = 0.f;
        if (!behind[i]) {
            const float iw = __fdiv_rn(1.0f, pv[i].w);
            x[i] = __fmaf_rn(pv[i].x * iw, hw, -fx);
            y[i] = __fmaf_rn(pv[i].y * iw, hh, -fy);
        }
    }
    if (nbehind == 3) return;
    bool sil[3];
    if (nbehind == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int ov = opp[t * 3 + k];
            ox[k] = x[k]; oy[k] = y[k];
            if (ov >= 0) {
                const float4 p = ld4(P + (size_t)ov * 4);
                if (p.w > 0.f) {
                    const float iw = __fdiv_rn(1.0f, p.w);
                    ox[k] = __fmaf_rn(p.x * iw, hw, -fx);
                    oy[k] = __fmaf_rn(p.y * iw, hh, -fy);
                }
            }
        }
        const float bb = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
        float aw[3];
        aw[0] = (x[1] - ox[0]) * (y[2] - oy[0]) - (x[2] - ox[0]) * (y[1] - oy[0]);
        aw[1] = (x[2] - ox[1]) * (y[0] - oy[1]) - (x[0] - ox[1]) * (y[2] - oy[1]);
        aw[2] = (x[0] - ox[2]) * (y[1] - oy[2]) - (x[1] - ox[2]) * (y[0] - oy[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) sil[k] = sign_bit(aw[k]) == sign_bit(bb);
    } else {
        // a triangle cut by the eye plane (oracle aa_eval_pair): orientation tests in homogeneous form, and only the edges with both
        // endpoints in front of the eye plane can be the crossed edge (below)
        const float D = aa_det3_xyw(pv[0], pv[1], pv[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int ov = opp[t * 3 + k];
            sil[k] = true;
            if (ov >= 0) {
                const float4 q = ld4(P + (size_t)ov * 4);
                if (q.w > 0.f) sil[k] = sign_bit(aa_det3_xyw(q, pv[(k + 1) % 3], pv[(k + 2) % 3])) == sign_bit(D);
            }
        }
    }
    if (!(sil[0] || sil[1] || sil[2])) return;
    if (d) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float tmp = x[i]; x[i] = y[i]; y[i] = tmp; }
    }
    int best = -1;
    float rbest = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ia = (k + 1) % 3, ib = (k + 2) % 3;
        if (behind[ia] || behind[ib]) continue;
        if (sign_bit(y[ia]) == sign_bit(y[ib])) continue;
        const float dx = x[ib] - x[ia], dy = y[ib] - y[ia];
        const float r = ds * __fdiv_rn(x[ia] * dy - y[ia] * dx, dy);
        if (best < 0 || r > rbest) { best = k; rbest = r; }
    }
    if (best < 0) return;
    const int ia = (best + 1) % 3, ib = (best + 2) % 3;
    // select without dynamic register-array indexing
    const float xa = ia == 0 ? x[0] : (ia == 1 ? x[1] : x[2]), ya = ia == 0 ? y[0] : (ia == 1 ? y[1] : y[2]);
    const float xb = ib == 0 ? x[0] : (ib == 1 ? x[1] : x[2]), yb = ib == 0 ? y[0] : (ib == 1 ? y[1] : y[2]);
    const bool silb = best == 0 ? sil[0] : (best == 1 ? sil[1] : sil[2]);
    const float dx = xb - xa, dy = yb - ya;
    if (!(silb && fabsf(dy) >= fabsf(dx))) return;
    const float eps = 0.0625f;
    if (!(rbest > -eps && rbest < 1.0f + eps)) return;
    o.valid = true;
    o.tri = t; o.chosen1 = chosen1; o.d = d;
    o.va = ia == 0 ? vi[0] : (ia == 1 ? vi[1] : vi[2]);
    o.vb = ib == 0 ? vi[0] : (ib == 1 ? vi[1] : vi[2]);
    o.ds = ds; o.dc = rbest;
    o.clamped = !(rbest > 0.f && rbest < 1.f);
    const float dcc = rbest < 0.f ? 0.f : (rbest > 1.f ? 1.f : rbest);
    o.alpha = ds * (0.5f - dcc);
    o.xa = xa; o.ya = ya; o.xb = xb; o.yb = yb;
    o.fx = fx; o.fy = fy;
}

// gradient of alpha w.r.t. the clip (x,y,w) of the pair's two edge vertices, times galpha
__device__ __forceinline__ void aa_pair_backward(const AAPair& pr, const float* __restrict__ P, int H, int W,
                                                 float galpha, float g[2][3] /* [vertex a/b][x,y,w] */)
{
    const float gr = -galpha;
    const float D = pr.yb - pr.ya;
    const float r = __fdiv_rn(pr.xa * pr.yb - pr.ya * pr.xb, D);
    const float g_xa = gr * pr.yb / D, g_xb = gr * (-pr.ya) / D;
    const float g_ya = gr * (r - pr.xb) / D, g_yb = gr * (pr.xa - r) / D;
    const float gX[2] = {pr.d ? g_ya : g_xa, pr.d ? g_yb : g_xb};
    const float gY[2] = {pr.d ? g_xa : g_ya, pr.d ? g_xb : g_yb};
    const float ix[2] = {pr.d ? pr.ya : pr.xa, pr.d ? pr.yb : pr.xb};
    const float iy[2] = {pr.d ? pr.xa : pr.ya, pr.d ? pr.xb : pr.yb};
    const int vv[2] = {pr.va, pr.vb};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float pw = P[(size_t)vv[i] * 4 + 3];
        const float iw = __fdiv_rn(1.0f, pw);
        g[i][0] = gX[i] * 0.5f * (float)W * iw;
        g[i][1] = gY[i] * 0.5f * (float)H * iw;
        g[i][2] = -(gX[i] * (ix[i] + pr.fx) + gY[i] * (iy[i] + pr.fy)) * iw;
    }
}
