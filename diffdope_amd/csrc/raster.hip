// raster.hip -- software rasteriser for gfx950 (replaces dr.rasterize / RasterizeGLContext,
// diffdope/diffdope.py:198-200,1312).
//
// Meshes on this path are 20k-50k triangles landing on a few thousand pixels: most triangles own zero or
// one pixel centre.  The rasteriser is therefore split by triangle size:
//
//   scatter_kernel   one lane per (hypothesis, triangle), two triangles per lane.  Three 8-byte gathers of the
//                    per-vertex snapped window coordinates (1/256 px, produced once per vertex by the transform
//                    kernel), exact integer setup, pixel-centre bbox.  No centre inside -> dead.  Up to
//                    RASTER_SMALL_PX centres -> resolved right here in two passes: the coverage of the bbox centres
//                    as a bit mask (edge functions stepped in 32-bit integers relative to the bbox corner -- exact:
//                    a small triangle spans < 2^13 sub-pixels), then per set bit fp32 z/w and one non-returning
//                    64-bit atomicMin of (depth key, id) into the depth/visibility buffer zbuf[b,y,x].
//                    Larger -> appended (id, packed tile range) to the hypothesis' list of LARGE triangles, one
//                    atomic per wave that has any.  Either way the 16x16 tiles under bbox + 1 px (antialias apron)
//                    are flagged active with plain stores (no atomic on a hot word).
//   compact_big_kernel  workgroups [0,B): ordered per-hypothesis compaction of the flags (active lists); the other
//                    workgroups are the tile pass for LARGE triangles (none in the micro-polygon regime: they exit
//                    on one scalar load): per flagged tile, the hypothesis' list is range-tested in rounds of 256,
//                    refined with the exact edge predicate at the tile corners, and the survivors are staged in LDS
//                    (tile-local edge values + steps, clip vertices); then the 256 lanes are the 256 pixels,
//                    merging into zbuf with atomicMin.  No per-tile lists in memory, so nothing can overflow.
//   emit_kernel      (op-level API only) expands zbuf into nvdiffrast's rast tensor (u, v, z/w, id+1).
//
// zbuf invariant: all ones between passes.  The op-level entry memsets it; the fused engine re-arms only
// the active tiles at the end of each iteration (update_kernel), so a 640x480x64 frame set costs ~3 MB of
// stores per iteration instead of 157 MB.
#include "raster.h"

// ---------------------------------------------------------------------------------------------
// scratch carving
size_t raster_layout(RasterScratch& L, void* base, int B, int V, int T, int H, int W)
{
    const int ntx = ddx_cdiv(W, DDX_TILE), nty = ddx_cdiv(H, DDX_TILE);
    const long long NT = (long long)ntx * nty;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    char* p = (char*)base;
    const size_t o_counters = carve(16 * sizeof(int));
    const size_t o_flag = carve((size_t)B * NT * sizeof(int));
    const size_t o_big = carve((size_t)B * NT * sizeof(int));
    const size_t o_bigcount = carve((size_t)B * sizeof(int));
    L.zero_bytes = off;  // [counters | tile_flag | tile_big | bigcount] must be zero when a pass starts
    const size_t o_active = carve((size_t)B * NT * sizeof(int));
    const size_t o_bcount = carve((size_t)B * sizeof(int));
    const size_t o_snap = carve((size_t)B * V * sizeof(int2));
    const size_t o_range = carve((size_t)B * T * sizeof(uint2));
    const int zwb = (W + 3) / 4, zhb = (H + 3) / 4;
    const size_t zper = (size_t)zwb * zhb * 16;
    const size_t o_zbuf = carve((size_t)B * zper * sizeof(unsigned long long));
    L.counters = (int*)(p + o_counters);
    L.tile_flag = (int*)(p + o_flag);
    L.tile_big = (int*)(p + o_big);
    L.active = (int*)(p + o_active);
    L.b_count = (int*)(p + o_bcount);
    L.snap = (int2*)(p + o_snap);
    L.biglist = (uint2*)(p + o_range);
    L.bigcount = (int*)(p + o_bigcount);
    L.zbuf = (unsigned long long*)(p + o_zbuf);
    L.zbuf_bytes = (size_t)B * zper * sizeof(unsigned long long);
    L.zper = zper;
    L.zwb = zwb;
    L.trisort = nullptr;
    L.cull_sign = 0;
    L.cull_ok = nullptr;
    L.scatter_exchange = 0;
    L.ntx = ntx; L.nty = nty; L.NT = (int)NT;
    L.ndc.xs = 2.0f / (float)W; L.ndc.xo = 1.0f / (float)W - 1.0f;  // as make_pixndc (host float division is IEEE too)
    L.ndc.ys = 2.0f / (float)H; L.ndc.yo = 1.0f / (float)H - 1.0f;
#ifdef DDX_TRACE
    L.trace = nullptr;
#endif
    return off;
}

// ---------------------------------------------------------------------------------------------
// snap: clip -> 1/256-pixel window coordinates, once per vertex (each vertex is shared by ~6 triangles)
__global__ __launch_bounds__(256) void snap_kernel(const float* __restrict__ pos, long long n, int H, int W,
                                                   int2* __restrict__ snap)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = ld4(pos + i * 4);
    snap[i] = snap_vertex(p, H, W);
}

int raster_snap(const float* pos, int B, int V, int H, int W, const RasterScratch& L, hipStream_t s)
{
    const long long n = (long long)B * V;
    snap_kernel<<<ddx_cdiv(n, 256), 256, 0, s>>>(pos, n, H, W, L.snap);
    DDX_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long frag_key(const float4& p0, const float4& p1, const float4& p2, int px, int py,
                                                       int H, int W, int t)
{
    Bary bc;
    if (!pixel_bary(p0, p1, p2, px, py, H, W, bc)) return ~0ull;
    if (!(bc.zw >= -1.0f && bc.zw <= 1.0f)) return ~0ull;
    return ((unsigned long long)depth_key(bc.zw) << 32) | (unsigned)t;
}

// 32-bit edge function of a SMALL triangle relative to the bbox corner pixel centre (X0,Y0):
// e(i,j) = e00 + i*sx + j*sy for the pixel (px0+i, py0+j); `own` = ownership of the e == 0 line.
struct Edge32 { int e00, sx, sy; bool own; };

__device__ __forceinline__ Edge32 make_edge(int ax, int ay, int bx, int by, int X0, int Y0, bool flip)
{
    int dx = bx - ax, dy = by - ay;
    if (flip) { dx = -dx; dy = -dy; }
    Edge32 e;
    // e = dx*(PY-ay) - dy*(PX-ax); all factors < 2^14 for a small triangle near its own bbox: 24-bit multiplies
    // (full rate; a 32-bit v_mul_lo_u32 issues at quarter rate and this kernel is VALU-issue bound)
    e.e00 = __mul24(dx, Y0 - ay) - __mul24(dy, X0 - ax);
    e.sx = -dy * DDX_SUBPIX;
    e.sy = dx * DDX_SUBPIX;
    e.own = (dy > 0) || (dy == 0 && dx < 0);
    return e;
}

#if RASTER_SMALL_PX > 32
typedef unsigned long long scatter_mask_t;
#else
typedef unsigned scatter_mask_t;
#endif

// Coverage of one triangle (integer only, no memory reads): flags its tiles; for a SMALL triangle the bit mask of the covered
// pixel centres of its bbox (bit k = j * nxp + i  <=>  pixel (px0 + i, py0 + j)) goes to `cv`; returns the packed tile range
// of a LARGE triangle (resolved later by the tile pass), ~0u otherwise.
struct ScatterCov { scatter_mask_t mask; int px0, py0, nxp; int clipped; };  // clipped: a near-plane straddler, resolved by the tile pass

// WALK: the lane resolves its covered centres itself, right here (the plain variant of the kernel; kept inside this function,
// in the scope that computed the mask, because hoisting it out costs 2-3 % of the kernel in the compiler's schedule).
// WALK = 2: ... unless one of the lanes that reached this point owns more than SCATTER_DIRECT_MAX centres, in which case they
// all hand their masks to the wave's fragment exchange.
#ifndef SCATTER_DIRECT_MAX
#define SCATTER_DIRECT_MAX 3  // waves in which no lane owns more fragments than this resolve them lane by lane
#endif
// DEFER (the compacting variant of the kernel): a small triangle that is neither degenerate nor culled is only reported
// (cv.clipped = 2); scatter_small_deferred() resolves it after the wave has packed such triangles into consecutive lanes.
template <int WALK, bool DEFER = false>
__device__ __forceinline__ unsigned scatter_one(const float* __restrict__ pos, int V, int H, int W, const RasterScratch& L, int b, int t,
                                                int i0, int i1, int i2, const int2& a, const int2& bq, const int2& c, ScatterCov& cv, int cull)
{
    cv.mask = 0; cv.px0 = 0; cv.py0 = 0; cv.nxp = 1; cv.clipped = 0;
    unsigned range = ~0u;  // packed tile range of a LARGE triangle
    if (a.x != INT_MIN && bq.x != INT_MIN && c.x != INT_MIN) {
        const int xmin = min(a.x, min(bq.x, c.x)), xmax = max(a.x, max(bq.x, c.x));
        const int ymin = min(a.y, min(bq.y, c.y)), ymax = max(a.y, max(bq.y, c.y));
        int px0 = (xmin - DDX_SUBPIX / 2 + (DDX_SUBPIX - 1)) >> 8, px1 = (xmax - DDX_SUBPIX / 2) >> 8;
        int py0 = (ymin - DDX_SUBPIX / 2 + (DDX_SUBPIX - 1)) >> 8, py1 = (ymax - DDX_SUBPIX / 2) >> 8;
        px0 = max(px0, 0); py0 = max(py0, 0);
        px1 = min(px1, W - 1); py1 = min(py1, H - 1);
        if (px0 <= px1 && py0 <= py1) {
            const int nxp = px1 - px0 + 1, nyp = py1 - py0 + 1;
            const bool small = __mul24(nxp, nyp) <= RASTER_SMALL_PX && (xmax - xmin) < 8192 && (ymax - ymin) < 8192;
            bool alive = false;
            if (small) {
                // extents < 2^13 sub-pixels: the area and the edge functions fit 32 bits exactly
                const int area = __mul24(bq.x - a.x, c.y - a.y) - __mul24(c.x - a.x, bq.y - a.y);
                if (DEFER) {
                    if (area != 0 && !(cull != 0 && (area < 0) == (cull < 0))) cv.clipped = 2;
                } else
                if (area != 0 && !(cull != 0 && (area < 0) == (cull < 0))) {  // (non-degenerate and not a culled back face)
                    const bool flip = area < 0;
                    const int X0 = px0 * DDX_SUBPIX + DDX_SUBPIX / 2, Y0 = py0 * DDX_SUBPIX + DDX_SUBPIX / 2;
                    const Edge32 e0 = make_edge(bq.x, bq.y, c.x, c.y, X0, Y0, flip);
                    const Edge32 e1 = make_edge(c.x, c.y, a.x, a.y, X0, Y0, flip);
                    const Edge32 e2 = make_edge(a.x, a.y, bq.x, bq.y, X0, Y0, flip);
                    // pass 1: coverage of the <= RASTER_SMALL_PX bbox centres as a bit mask -- integer only; the ownership rule
                    // (v > 0 || (v == 0 && own)) is folded into the start value: v + own - 1 >= 0
                    const int b0 = e0.e00 + (int)e0.own - 1, b1 = e1.e00 + (int)e1.own - 1, b2 = e2.e00 + (int)e2.own - 1;
                    scatter_mask_t mask = 0;
                    {
                        // (a branch-free path for boxes of at most 2x2 centres -- nearly every triangle of the dense meshes -- that lets
                        // whole waves skip this loop was measured: +-0.5 % on cfg2 / cfg3 / cfg50k64: the kernel waits on its memory levels)
                        int idx = 0;
                        int r0 = b0, r1 = b1, r2 = b2;
                        for (int j = 0; j < nyp; ++j, r0 += e0.sy, r1 += e1.sy, r2 += e2.sy) {
                            int v0 = r0, v1 = r1, v2 = r2;
                            for (int i = 0; i < nxp; ++i, ++idx, v0 += e0.sx, v1 += e1.sx, v2 += e2.sx)
                                mask |= (scatter_mask_t)((v0 | v1 | v2) >= 0) << idx;
                        }
                    }
                    alive = mask != 0;  // a small triangle that covers no centre draws nothing: no tile to flag
                    bool walk = WALK == 1;
                    if (WALK == 2) {
                        const int cnt = RASTER_SMALL_PX > 32 ? __popcll(mask) : __popc((unsigned)mask);
                        walk = __ballot(cnt > SCATTER_DIRECT_MAX) == 0ull;  // (over the lanes active here)
                    }
                    if (walk) {
                        // one depth evaluation + one atomic per covered centre: the wave walks max(popcount) rounds instead
                        // of max(bbox area), and the clip-space vertices are loaded once, up front
                        if (mask) {
                            const float* P = pos + (size_t)b * V * 4;
                            const float4 p0 = ld4(P + (size_t)i0 * 4), p1 = ld4(P + (size_t)i1 * 4), p2 = ld4(P + (size_t)i2 * 4);
                            unsigned long long* Z = L.zbuf + (size_t)b * L.zper;
                            const PixNdc ndc = L.ndc;  // host-computed (IEEE divisions, same values as make_pixndc)
                            const float rn = __frcp_rn((float)nxp);
                            while (mask) {
                                const int k = RASTER_SMALL_PX > 32 ? __ffsll((long long)mask) - 1 : __ffs((unsigned)mask) - 1;
                                mask &= mask - 1;
                                const int j = (int)(((float)k + 0.5f) * rn), i = k - __mul24(j, nxp);  // k = j * nxp + i, exact for k < 64
                                float zw;
                                const float fx = __fmaf_rn((float)(px0 + i), ndc.xs, ndc.xo), fy = __fmaf_rn((float)(py0 + j), ndc.ys, ndc.yo);
                                if (pixel_depth(p0, p1, p2, fx, fy, zw))
                                    atomicMin(Z + zaddr(px0 + i, py0 + j, L.zwb), ((unsigned long long)depth_key(zw) << 32) | (unsigned)t);
                            }
                        }
                    } else {
                        cv.mask = mask; cv.px0 = px0; cv.py0 = py0; cv.nxp = nxp;
                    }
                }
            } else {
                const long long area = (long long)(bq.x - a.x) * (long long)(c.y - a.y) - (long long)(c.x - a.x) * (long long)(bq.y - a.y);
                alive = area != 0 && !(cull != 0 && (area < 0) == (cull < 0));
            }
            if (alive) {
                // tiles under bbox + 1 px: every pixel adjacent to a covered pixel lies in an active tile
                // (conservative: a centre inside the bbox need not be covered)
                const int tx0 = max(px0 - 1, 0) / DDX_TILE, tx1 = min(px1 + 1, W - 1) / DDX_TILE;
                const int ty0 = max(py0 - 1, 0) / DDX_TILE, ty1 = min(py1 + 1, H - 1) / DDX_TILE;
                // (the usual trip count is 1 x 1: keep the compiler from unrolling / vectorising these loops)
                int* flag = L.tile_flag + (size_t)b * L.NT;
#pragma clang loop unroll(disable) vectorize(disable)
                for (int ty = ty0; ty <= ty1; ++ty)
#pragma clang loop unroll(disable) vectorize(disable)
                    for (int tx = tx0; tx <= tx1; ++tx) flag[__mul24(ty, L.ntx) + tx] = 1;  // plain store, no atomics
                if (!small) {
                    int* big = L.tile_big + (size_t)b * L.NT;
#pragma clang loop unroll(disable) vectorize(disable)
                    for (int ty = ty0; ty <= ty1; ++ty)
#pragma clang loop unroll(disable) vectorize(disable)
                        for (int tx = tx0; tx <= tx1; ++tx) big[__mul24(ty, L.ntx) + tx] = 1;
                    range = (unsigned)tx0 | ((unsigned)ty0 << 8) | ((unsigned)(tx1 - tx0) << 16) | ((unsigned)(ty1 - ty0) << 24);
                    L.counters[3] = 1;  // plain store: "the batch has a large triangle"
                }
            }
        }
    } else {
        // a vertex at w <= 0 (rare: a hypothesis that dives through the camera).  If any corner lies in front of the near plane the
        // triangle is a straddler: the tile pass clips it (clip_near) and draws the visible part.  Where that part lands cannot
        // be bounded from the snapped corners, so it is listed for the whole frame.  (Kept to a few instructions on purpose:
        // with the clipping arithmetic inlined here the kernel's hot path lost 2x to instruction fetch, cfg2 12.5 -> 23-33 us.)
        const float* P = pos + (size_t)b * V * 4;
        const float2 zw0 = *reinterpret_cast<const float2*>(P + (size_t)i0 * 4 + 2), zw1 = *reinterpret_cast<const float2*>(P + (size_t)i1 * 4 + 2),
                     zw2 = *reinterpret_cast<const float2*>(P + (size_t)i2 * 4 + 2);
        if (zw0.x + zw0.y >= 0.f || zw1.x + zw1.y >= 0.f || zw2.x + zw2.y >= 0.f) {
            int* flag = L.tile_flag + (size_t)b * L.NT;
            int* big = L.tile_big + (size_t)b * L.NT;
#pragma clang loop unroll(disable) vectorize(disable)
            for (int i = 0; i < L.NT; ++i) { flag[i] = 1; big[i] = 1; }
            range = ((unsigned)(L.ntx - 1) << 16) | ((unsigned)(L.nty - 1) << 24);
            L.counters[3] = 1;
            cv.clipped = 1;
        }
    }
    return range;
}

// The body of scatter_one for a SMALL, non-degenerate, not culled triangle (see DEFER there): bbox, coverage mask of its
// <= RASTER_SMALL_PX centres, tile flags, one depth evaluation + atomicMin per covered centre.  Same arithmetic, same results.
__device__ __forceinline__ void scatter_small_deferred(const float* __restrict__ pos, int V, int H, int W, const RasterScratch& L, int b, int t,
                                                       int i0, int i1, int i2, const int2& a, const int2& bq, const int2& c)
{
    const int xmin = min(a.x, min(bq.x, c.x)), xmax = max(a.x, max(bq.x, c.x));
    const int ymin = min(a.y, min(bq.y, c.y)), ymax = max(a.y, max(bq.y, c.y));
    int px0 = (xmin - DDX_SUBPIX / 2 + (DDX_SUBPIX - 1)) >> 8, px1 = (xmax - DDX_SUBPIX / 2) >> 8;
    int py0 = (ymin - DDX_SUBPIX / 2 + (DDX_SUBPIX - 1)) >> 8, py1 = (ymax - DDX_SUBPIX / 2) >> 8;
    px0 = max(px0, 0); py0 = max(py0, 0);
    px1 = min(px1, W - 1); py1 = min(py1, H - 1);
    const int nxp = px1 - px0 + 1, nyp = py1 - py0 + 1;
    const int area = __mul24(bq.x - a.x, c.y - a.y) - __mul24(c.x - a.x, bq.y - a.y);
    const bool flip = area < 0;
    const int X0 = px0 * DDX_SUBPIX + DDX_SUBPIX / 2, Y0 = py0 * DDX_SUBPIX + DDX_SUBPIX / 2;
    const Edge32 e0 = make_edge(bq.x, bq.y, c.x, c.y, X0, Y0, flip);
    const Edge32 e1 = make_edge(c.x, c.y, a.x, a.y, X0, Y0, flip);
    const Edge32 e2 = make_edge(a.x, a.y, bq.x, bq.y, X0, Y0, flip);
    const int b0 = e0.e00 + (int)e0.own - 1, b1 = e1.e00 + (int)e1.own - 1, b2 = e2.e00 + (int)e2.own - 1;
    scatter_mask_t mask = 0;
    {
        int idx = 0;
        int r0 = b0, r1 = b1, r2 = b2;
        for (int j = 0; j < nyp; ++j, r0 += e0.sy, r1 += e1.sy, r2 += e2.sy) {
            int v0 = r0, v1 = r1, v2 = r2;
            for (int i = 0; i < nxp; ++i, ++idx, v0 += e0.sx, v1 += e1.sx, v2 += e2.sx)
                mask |= (scatter_mask_t)((v0 | v1 | v2) >= 0) << idx;
        }
    }
    if (!mask) return;  // covers no centre: draws nothing, no tile to flag
    {
        const int tx0 = max(px0 - 1, 0) / DDX_TILE, tx1 = min(px1 + 1, W - 1) / DDX_TILE;
        const int ty0 = max(py0 - 1, 0) / DDX_TILE, ty1 = min(py1 + 1, H - 1) / DDX_TILE;
        int* flag = L.tile_flag + (size_t)b * L.NT;
#pragma clang loop unroll(disable) vectorize(disable)
        for (int ty = ty0; ty <= ty1; ++ty)
#pragma clang loop unroll(disable) vectorize(disable)
            for (int tx = tx0; tx <= tx1; ++tx) flag[__mul24(ty, L.ntx) + tx] = 1;
    }
    const float* P = pos + (size_t)b * V * 4;
    const float4 p0 = ld4(P + (size_t)i0 * 4), p1 = ld4(P + (size_t)i1 * 4), p2 = ld4(P + (size_t)i2 * 4);
    unsigned long long* Z = L.zbuf + (size_t)b * L.zper;
    const PixNdc ndc = L.ndc;
    const float rn = __frcp_rn((float)nxp);
    while (mask) {
        const int k = RASTER_SMALL_PX > 32 ? __ffsll((long long)mask) - 1 : __ffs((unsigned)mask) - 1;
        mask &= mask - 1;
        const int j = (int)(((float)k + 0.5f) * rn), i = k - __mul24(j, nxp);
        float zw;
        const float fx = __fmaf_rn((float)(px0 + i), ndc.xs, ndc.xo), fy = __fmaf_rn((float)(py0 + j), ndc.ys, ndc.yo);
        if (pixel_depth(p0, p1, p2, fx, fy, zw))
            atomicMin(Z + zaddr(px0 + i, py0 + j, L.zwb), ((unsigned long long)depth_key(zw) << 32) | (unsigned)t);
    }
}

// every lane walks the fragments of its own triangle (clip-space vertices loaded only when it owns a centre)
__device__ __forceinline__ void scatter_walk(const ScatterCov& cv, const float* __restrict__ P, int i0, int i1, int i2, int t,
                                             unsigned long long* __restrict__ Z, const PixNdc& ndc, int zwb)
{
    if (!cv.mask) return;
    const float4 p0 = ld4(P + (size_t)i0 * 4), p1 = ld4(P + (size_t)i1 * 4), p2 = ld4(P + (size_t)i2 * 4);
    scatter_mask_t m = cv.mask;
    const float rn = __frcp_rn((float)cv.nxp);
    while (m) {
        const int kb = RASTER_SMALL_PX > 32 ? __ffsll((long long)m) - 1 : __ffs((unsigned)m) - 1;
        m &= m - 1;
        const int j = (int)(((float)kb + 0.5f) * rn), i = kb - __mul24(j, cv.nxp);  // kb = j * nxp + i, exact for kb < 64
        float zw;
        const float fx = __fmaf_rn((float)(cv.px0 + i), ndc.xs, ndc.xo), fy = __fmaf_rn((float)(cv.py0 + j), ndc.ys, ndc.yo);
        if (pixel_depth(p0, p1, p2, fx, fy, zw))
            atomicMin(Z + zaddr(cv.px0 + i, cv.py0 + j, zwb), ((unsigned long long)depth_key(zw) << 32) | (unsigned)t);
    }
}

// j-th (0-based) set bit of m; j < popcount(m)
__device__ __forceinline__ int select_bit(scatter_mask_t m, int j)
{
    int pos = 0;
    unsigned w = (unsigned)m;
#if RASTER_SMALL_PX > 32
    {
        const int c = __popc(w);
        if (j >= c) { j -= c; w = (unsigned)(m >> 32); pos = 32; }
    }
#endif
    int c = __popc(w & 0xFFFFu); if (j >= c) { j -= c; w >>= 16; pos += 16; } w &= 0xFFFFu;
    c = __popc(w & 0xFFu); if (j >= c) { j -= c; w >>= 8; pos += 8; } w &= 0xFFu;
    c = __popc(w & 0xFu); if (j >= c) { j -= c; w >>= 4; pos += 4; } w &= 0xFu;
    c = __popc(w & 3u); if (j >= c) { j -= c; w >>= 2; pos += 2; } w &= 3u;
    if (j >= (int)(w & 1u)) pos += 1;
    return pos;
}

// TPL triangles per lane, NT threads per workgroup: (2, 256) normally; (1, 64) for small meshes, where 512-triangle chunks
// would leave most of the chip without a workgroup (a 384-triangle CAD model x 64 hypotheses = 64 workgroups).
// EXCHANGE: redistribute the fragments over the lanes of the wave (see below) -- pays when triangles own many centres
// (the small-mesh variant); in the micro-polygon regime of the dense meshes the plain per-lane walk is ~10 % faster.
// MODE 0: plain; 1: exchange (small meshes); 2: plain unless a lane owns more than SCATTER_DIRECT_MAX centres
template <int SCATTER_TPL, int SCATTER_NT, int MODE>
__global__ __launch_bounds__(SCATTER_NT) void scatter_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int V,
                                                      int T, int H, int W, RasterScratch L)
{
    constexpr bool EXCHANGE = MODE == 1 || MODE == 2;
    constexpr bool COMPACT = MODE == 3;
    // COMPACT (MODE 3; dense meshes whose launch is several rounds of resident workgroups, where the kernel is VALU-bound): only
    // ~30 % of the triangles survive the bbox / area / back-face tests, and a wave pays the coverage + fragment code for its
    // 2 x 64 triangles whenever ONE lane survives.  The survivors of both triangles of the lanes are packed into consecutive
    // lanes through LDS (32 bytes each: first corner, the other two relative to it in 16 bits -- a small triangle spans < 2^13
    // sub-pixels --, vertex ids, triangle id) and resolved in ceil(n / 64) passes: one instead of two, with full lanes.
    __shared__ int4 s_q[COMPACT ? SCATTER_NT / 64 : 1][COMPACT ? 64 * SCATTER_TPL : 1][2];
    // fragment exchange of one wave: exclusive prefix of the lanes' fragment counts, their coverage masks and triangle records
    __shared__ int s_pref[EXCHANGE ? SCATTER_NT / 64 : 1][EXCHANGE ? 64 : 1];
    __shared__ scatter_mask_t s_mask[EXCHANGE ? SCATTER_NT / 64 : 1][EXCHANGE ? 64 : 1];
    __shared__ float4 s_rec[EXCHANGE ? SCATTER_NT / 64 : 1][EXCHANGE ? 64 : 1][4];  // p0, p1, p2, (px0, py0, nxp, id) as bits
    DDX_TRACE_BEGIN();
#if defined(DDX_TRACE) && defined(DDX_PHASES)
    unsigned long long sph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SPH(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); sph[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SPH(i)
#endif
    SPH(0);
    // (an XCD-aware mapping -- hypothesis b entirely on XCD b % 8 through a 1-D grid -- was measured: +-0 on cfg2,
    // -8 % on cfg3/cfg4; the plain 2-D grid stays)
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int2* S = L.snap + (size_t)b * V;
    int t[SCATTER_TPL], i0[SCATTER_TPL], i1[SCATTER_TPL], i2[SCATTER_TPL];
    bool ok[SCATTER_TPL];
    int2 va[SCATTER_TPL], vb[SCATTER_TPL], vc[SCATTER_TPL];
#pragma unroll
    for (int k = 0; k < SCATTER_TPL; ++k) {
        const int slot = (chunk * SCATTER_TPL + k) * SCATTER_NT + threadIdx.x;
        // engine: spatially sorted records {v0, v1, v2, original id} -- the lanes of a wave take triangles that are
        // neighbours in space, so their fragments share zbuf lines; one coalesced 16-byte load per triangle.
        // t[k] is the ORIGINAL triangle id (T marks a lane past the end)
        if (L.trisort) {
            const int4 rec = L.trisort[min(slot, T - 1)];
            i0[k] = rec.x; i1[k] = rec.y; i2[k] = rec.z;
            t[k] = slot < T ? rec.w : T;
        } else {
            const int tt = min(slot, T - 1);
            i0[k] = tri[tt * 3 + 0]; i1[k] = tri[tt * 3 + 1]; i2[k] = tri[tt * 3 + 2];
            t[k] = slot < T ? slot : T;
        }
    }
    SPH(1);
#pragma unroll
    for (int k = 0; k < SCATTER_TPL; ++k) {
        ok[k] = t[k] < T && (unsigned)i0[k] < (unsigned)V && (unsigned)i1[k] < (unsigned)V && (unsigned)i2[k] < (unsigned)V;
        const int j0 = ok[k] ? i0[k] : 0, j1 = ok[k] ? i1[k] : 0, j2 = ok[k] ? i2[k] : 0;
        va[k] = S[j0]; vb[k] = S[j1]; vc[k] = S[j2];
    }
    SPH(2);
    // back-face culling of this hypothesis in this pass (see RasterScratch::cull_sign): only while every vertex slice of the
    // transform reported the whole object inside the view volume (uniform: scalar loads)
    int cull = 0;
    if (L.cull_sign != 0 && L.cull_ok) {
        const int* ck = L.cull_ok + (size_t)b * 8;
        const bool all_ok = (ck[0] & ck[1] & ck[2] & ck[3] & ck[4] & ck[5] & ck[6] & ck[7]) != 0;
        cull = all_ok ? L.cull_sign : 0;
    }
    unsigned range[SCATTER_TPL];
    ScatterCov cv[SCATTER_TPL];
#pragma unroll
    for (int k = 0; k < SCATTER_TPL; ++k) {
        range[k] = ~0u;
        cv[k].mask = 0; cv[k].px0 = 0; cv[k].py0 = 0; cv[k].nxp = 1; cv[k].clipped = 0;
        if (t[k] >= T || !ok[k]) continue;
        // (plain variant: triangle by triangle -- coverage of both triangles first and all fragments afterwards measured
        // 2 us slower on cfg2: more atomics in flight at once make the atomicMin stream slower)
        range[k] = scatter_one<(MODE == 0 || MODE == 3) ? 1 : MODE == 2 ? 2 : 0, COMPACT>(pos, V, H, W, L, b, t[k], i0[k], i1[k], i2[k], va[k], vb[k], vc[k], cv[k], cull);
    }
    if (COMPACT) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int n_q = 0;
#pragma unroll
        for (int k = 0; k < SCATTER_TPL; ++k) {
            const bool push = cv[k].clipped == 2;
            if (push) cv[k].clipped = 0;
            const unsigned long long m = __ballot(push);
            if (push) {
                const int slot = n_q + __popcll(m & ((1ull << lane) - 1ull));
                const unsigned rb = ((unsigned)(vb[k].x - va[k].x) & 0xffffu) | ((unsigned)(vb[k].y - va[k].y) << 16);
                const unsigned rc = ((unsigned)(vc[k].x - va[k].x) & 0xffffu) | ((unsigned)(vc[k].y - va[k].y) << 16);
                s_q[wv][slot][0] = make_int4(va[k].x, va[k].y, (int)rb, (int)rc);
                s_q[wv][slot][1] = make_int4(i0[k], i1[k], i2[k], t[k]);
            }
            n_q += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < n_q; base += 64) {  // (wave-uniform)
            const int idx = base + lane;
            if (idx < n_q) {
                const int4 g = s_q[wv][idx][0], h = s_q[wv][idx][1];
                const int2 qa = make_int2(g.x, g.y);
                const int2 qb = make_int2(g.x + (int)(short)((unsigned)g.z & 0xffffu), g.y + ((int)g.z >> 16));
                const int2 qc = make_int2(g.x + (int)(short)((unsigned)g.w & 0xffffu), g.y + ((int)g.w >> 16));
                scatter_small_deferred(pos, V, H, W, L, b, h.w, h.x, h.y, h.z, qa, qb, qc);
            }
        }
    }
    SPH(3);
    // ---- fragments.  A lane owns 0..64 covered centres of its triangle, most lanes none: walked lane by lane the wave runs
    // max(count) rounds at ~15 % lane utilisation, and one atomic instruction touches one pixel of up to 64 different
    // triangles.  Instead the wave's fragments are numbered consecutively (prefix sum of the counts) and fragment f goes to
    // lane f % 64: every lane works, and consecutive lanes take consecutive pixels of the same triangle -- the same zbuf
    // line, which is what the atomicMin stream is bound by.
    if (EXCHANGE) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned long long* Z = L.zbuf + (size_t)b * L.zper;
        const float* P = pos + (size_t)b * V * 4;
        const PixNdc ndc = L.ndc;  // host-computed (IEEE divisions, same values as make_pixndc)
#pragma unroll
        for (int k = 0; k < SCATTER_TPL; ++k) {
            const int cnt = RASTER_SMALL_PX > 32 ? __popcll(cv[k].mask) : __popc((unsigned)cv[k].mask);
            if (MODE == 1 && __ballot(cnt > SCATTER_DIRECT_MAX) == 0ull) {
                // nobody owns more than a few centres: the exchange would cost more than the idle lanes do
                scatter_walk(cv[k], P, i0[k], i1[k], i2[k], t[k], Z, ndc, L.zwb);
                continue;
            }
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            const int F = __shfl(incl, 63, 64);
            if (F == 0) continue;  // (wave-uniform)
            if (cnt) {  // clip-space vertices only for triangles that own a pixel centre
                s_rec[wv][lane][0] = ld4(P + (size_t)i0[k] * 4);
                s_rec[wv][lane][1] = ld4(P + (size_t)i1[k] * 4);
                s_rec[wv][lane][2] = ld4(P + (size_t)i2[k] * 4);
                s_rec[wv][lane][3] = make_float4(__int_as_float(cv[k].px0), __int_as_float(cv[k].py0), __int_as_float(cv[k].nxp), __int_as_float(t[k]));
            }
            s_pref[wv][lane] = incl - cnt;
            s_mask[wv][lane] = cv[k].mask;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < F; base += 64) {
                const int f = base + lane;
                if (f < F) {
                    int Lo = 0;  // largest lane whose exclusive prefix is <= f: the owner of fragment f
#pragma unroll
                    for (int st = 32; st > 0; st >>= 1)
                        if (s_pref[wv][Lo + st] <= f) Lo += st;
                    const int kb = select_bit(s_mask[wv][Lo], f - s_pref[wv][Lo]);
                    const float4 p0 = s_rec[wv][Lo][0], p1 = s_rec[wv][Lo][1], p2 = s_rec[wv][Lo][2], q = s_rec[wv][Lo][3];
                    const int px0 = __float_as_int(q.x), py0 = __float_as_int(q.y), nxp = __float_as_int(q.z), tid_ = __float_as_int(q.w);
                    const int j = (int)(((float)kb + 0.5f) * __frcp_rn((float)nxp)), i = kb - __mul24(j, nxp);  // kb = j * nxp + i, exact for kb < 64
                    float zw;
                    const float fx = __fmaf_rn((float)(px0 + i), ndc.xs, ndc.xo), fy = __fmaf_rn((float)(py0 + j), ndc.ys, ndc.yo);
                    if (pixel_depth(p0, p1, p2, fx, fy, zw))
                        atomicMin(Z + zaddr(px0 + i, py0 + j, L.zwb), ((unsigned long long)depth_key(zw) << 32) | (unsigned)tid_);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the LDS arrays are reused by the next k)
        }
    }
    SPH(4);
    // LARGE triangles go to the hypothesis' list for the tile pass: one atomic per WAVE that has any (none in the
    // micro-polygon regime), the lanes take consecutive slots.  The order of the list does not matter (atomicMin).
#pragma unroll
    for (int k = 0; k < SCATTER_TPL; ++k) {
        const unsigned long long m = __ballot(range[k] != ~0u);
        if (m == 0ull) continue;
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(L.bigcount + b, __popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (range[k] != ~0u)  // (bit 31 of the id: a near-plane straddler, clipped again by the tile pass)
            L.biglist[(size_t)b * T + base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2((unsigned)t[k] | (cv[k].clipped ? 0x80000000u : 0u), range[k]);
    }
#if defined(DDX_TRACE) && defined(DDX_PHASES)
    if (threadIdx.x == 0 && L.trace) {
        const size_t wg = blockIdx.x + gridDim.x * (size_t)blockIdx.y;
        if (wg < 4096) {
            unsigned long long* q = L.trace + ((size_t)1 * 8192) * 4 + wg * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = sph[i];
        }
    }
#endif
    DDX_TRACE_END(L.trace, 0, 1ull);
}

// clip_near as a real call: inlined into the candidate loop of the tile pass its code (used by the rare near-plane straddlers
// only) sat in the middle of the loop body every large tile walks, and the large-triangle workloads paid for it in instruction
// fetch (hugetri's tile pass 24.7 -> 32.2 us).
__device__ __attribute__((noinline)) int clip_near_call(const float4& p0, const float4& p1, const float4& p2, int H, int W, SnapTri out[2])
{
    return clip_near(p0, p1, p2, H, W, out);
}

// One launch, two roles.
//   workgroups [0, B): compaction -- workgroup b turns hypothesis b's tile flags into its ordered active-tile
//     segment + count (ballot ranks, no atomics).
//   workgroups [B, B + RASTER_BIG_GRID): large triangles -- exit on one scalar load when the batch has none;
//     otherwise stride over all (b, tile) with tile_big set: sweep the hypothesis' packed ranges,
//     ballot-compact the triangles overlapping the tile into LDS, then lane = pixel (exact int64 coverage),
//     merging into zbuf with atomicMin.
#define BIG_SCAN 1024  // (hypothesis, tile) flags scanned per step of the large-triangle pass

#define CB_ROUNDS 16  // (CB_ROUNDS * 4 waves = 64 counts: one wave-wide prefix)
__global__ __launch_bounds__(256) void compact_big_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int B,
                                                          int V, int T, int H, int W, RasterScratch L)
{
    __shared__ int4 s_e0[4][64], s_e1[4][64], s_e2[4][64];  // staged LARGE triangles of each wave's tile: per edge (e.lo, e.hi, step x, step y) at the tile origin
    __shared__ int s_t[4][64];                              // ... their ids
    __shared__ int s_cand[4][256];                          // range-test survivors of 256 list entries (per wave)
    __shared__ float4 s_p0[4][64], s_p1[4][64], s_p2[4][64];  // ... and their clip-space vertices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DDX_TRACE_BEGIN();
#ifdef DDX_TRACE
    unsigned long long n_done = 0;
#endif
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        const int* flg = L.tile_flag + (size_t)b * L.NT;
        int* out = L.active + (size_t)b * L.NT;
        // 16 rounds of 256 flags at a time: all loads in flight at once, one ordered prefix over the (round, wave) counts --
        // a load / ballot / barrier loop per 256 flags paid one memory round trip per round (5 on 640x480, 15 on 1280x720)
        __shared__ int s_cnt[CB_ROUNDS * 4], s_off[CB_ROUNDS * 4 + 1];
        int carry = 0;
        for (int start = 0; start < L.NT; start += CB_ROUNDS * 256) {
            int fl[CB_ROUNDS];
#pragma unroll
            for (int c = 0; c < CB_ROUNDS; ++c) {
                const int i = start + c * 256 + tid;
                fl[c] = i < L.NT ? flg[i] : 0;
            }
            unsigned long long m[CB_ROUNDS];
#pragma unroll
            for (int c = 0; c < CB_ROUNDS; ++c) m[c] = __ballot(fl[c] != 0);
            __syncthreads();  // (s_cnt / s_off of the previous super-round are no longer read)
            if (lane < CB_ROUNDS) {
                unsigned long long mine = 0ull;
#pragma unroll
                for (int c = 0; c < CB_ROUNDS; ++c) mine = lane == c ? m[c] : mine;
                s_cnt[lane * 4 + wave] = __popcll(mine);
            }
            __syncthreads();
            if (wave == 0) {  // exclusive prefix of the CB_ROUNDS * 4 = 64 counts in (round, wave) order
                const int v = s_cnt[lane];
                int incl = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                s_off[lane] = incl - v;
                if (lane == 63) s_off[64] = incl;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CB_ROUNDS; ++c) {
                const int i = start + c * 256 + tid;
                if (fl[c] != 0) out[carry + s_off[c * 4 + wave] + __popcll(m[c] & ((1ull << lane) - 1ull))] = ((i / L.ntx) << 16) | (i % L.ntx);  // (ty, tx) packed
            }
            carry += s_off[64];
        }
        if (tid == 0) L.b_count[b] = carry;
        return;
    }
    if (L.counters[3] == 0) return;  // no large triangle in the whole batch
    // workgroup g owns the pairs (b, tile) with tile = (g - 13 b) mod G (+ multiples of G): the large tiles of one object
    // are neighbours, and the objects of all hypotheses sit at about the same place on screen -- both a contiguous and
    // a plain strided split pile them up on a few workgroups (measured: 4 tiles on some, none on most).  The flags of
    // BIG_SCAN pairs are read in parallel and compacted into LDS.
    __shared__ int s_big[BIG_SCAN];
    __shared__ int s_nbig;
    const int G = gridDim.x - B, g = blockIdx.x - B;
    const int per_b = (L.NT + G - 1) / G;          // candidate tiles per hypothesis for this workgroup
    const int n_cand = B * per_b;
    for (int base = 0; base < n_cand; base += BIG_SCAN) {
    __syncthreads();
    if (tid == 0) s_nbig = 0;
    __syncthreads();
    for (int j = base + tid; j < min(n_cand, base + BIG_SCAN); j += 256) {
        const int bb = j / per_b, kk = j - bb * per_b;
        int t0 = (g - 13 * bb) % G;
        if (t0 < 0) t0 += G;
        const int tile = t0 + kk * G;
        if (tile < L.NT) {
            const int i = bb * L.NT + tile;
            if (L.tile_big[i] != 0) s_big[atomicAdd(&s_nbig, 1)] = i;  // LDS atomic; the order does not matter (atomicMin below)
        }
    }
    __syncthreads();
    const int nbig = s_nbig;
    // ---- one WAVE per large tile (four tiles in flight per workgroup, no workgroup barrier inside): a tile is a chain of
    // four dependent gathers (list entry -> vertex ids -> snapped vertices -> clip vertices) before its pixel loop, and a
    // workgroup that walked its tiles one by one paid the chain once per tile (3-4.5 us each: 51 us for the 24-triangle
    // hugetri workload, 13 tiles per workgroup; 32 us now).  Lanes test 64 candidates per round and then shade 4 pixels
    // each.  (Dealing (tile, round of 64 candidates) items to the waves instead -- for workgroups with fewer than four
    // tiles -- was measured: no better on the 384-triangle lowpoly workload, 27 -> 28.5 us.)
    for (int e = wave; e < nbig; e += 4) {  // (wave-uniform)
        const int flat = s_big[e];
        const int b = flat / L.NT, tile = flat - b * L.NT;
        const int tcx = tile % L.ntx, tcy = tile / L.ntx;
        const float* P = pos + (size_t)b * V * 4;
        const int2* S = L.snap + (size_t)b * V;
        const uint2* BL = L.biglist + (size_t)b * T;
        const int n_big = min(L.bigcount[b], T);
        const int lx = lane % DDX_TILE, ly0 = lane / DDX_TILE;  // pixels (lx, ly0 + 4 q), q = 0..3
        const int px = tcx * DDX_TILE + lx;
        unsigned long long best[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        for (int c0 = 0; c0 < n_big; c0 += 256) {
            // ---- (1) range test of 256 list entries at once (4 coalesced loads in flight, nothing dependent): the ids of the
            // triangles whose packed tile range contains this tile, compacted into the wave's candidate list
            int ncand = 0;
            {
                uint2 en[4];
                bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = c0 + u * 64 + lane;
                    en[u] = idx < n_big ? BL[idx] : make_uint2(0u, 0u);
                    const unsigned r = en[u].y;
                    const int x0 = r & 255, y0 = (r >> 8) & 255, nx = (r >> 16) & 255, ny = r >> 24;
                    in[u] = idx < n_big && tcx >= x0 && tcx <= x0 + nx && tcy >= y0 && tcy <= y0 + ny;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long mu = __ballot(in[u]);
                    if (in[u]) s_cand[wave][ncand + __popcll(mu & ((1ull << lane) - 1ull))] = (int)en[u].x;
                    ncand += __popcll(mu);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        for (int r0 = 0; r0 < ncand; r0 += 64) {
            // ---- (2) 64 candidates per round: exact refinement, and everything the pixel loop needs (edge functions +
            // clip-space vertices) is staged in LDS by the lane that found the hit -- parallel gathers instead of one
            // dependent gather chain per triangle and pixel loop step
            const int idx = r0 + lane;
            const bool cand = idx < ncand;
            const unsigned ent_id = cand ? (unsigned)s_cand[wave][idx] : 0u;  // bit 31: a near-plane straddler (see scatter_one)
            const int t_id = (int)(ent_id & 0x7fffffffu);
            // the snapped triangle(s) of the candidate: its own three vertices, or -- for a triangle with a vertex at w <= 0 --
            // the one or two triangles of its near-clipped polygon (clip_near; fragments still come from the original triangle)
            int i0 = 0, i1 = 0, i2 = 0;
            SnapTri stv[2];
            stv[0].ok = false; stv[1].ok = false;
            int nst = 0;
            float4 cp0 = make_float4(0.f, 0.f, 0.f, 0.f), cp1 = cp0, cp2 = cp0;
            if (cand) {
                i0 = tri[t_id * 3 + 0]; i1 = tri[t_id * 3 + 1]; i2 = tri[t_id * 3 + 2];
                if (ent_id >> 31) {
                    cp0 = ld4(P + (size_t)i0 * 4); cp1 = ld4(P + (size_t)i1 * 4); cp2 = ld4(P + (size_t)i2 * 4);
                    SnapTri clipped[2];  // (lives in scratch: its address goes to a real call; only this rare path touches it)
                    nst = clip_near_call(cp0, cp1, cp2, H, W, clipped);
                    stv[0] = clipped[0]; stv[1] = clipped[1];
                } else {
                    const int2 sa = S[i0], sb = S[i1], sc = S[i2];
                    snap_from_vertices(sa, sb, sc, stv[0]);
                    nst = stv[0].ok ? 1 : 0;
                }
            }
            const bool any_second = __ballot(nst > 1) != 0ull;
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            if (sub == 1 && !any_second) break;  // (wave-uniform)
            bool hit = cand && sub < nst;
            const SnapTri& st = sub == 0 ? stv[0] : stv[1];
            // the packed range is the triangle's bbox in tiles: refine with the exact edge predicate at the four corner
            // pixel centres of the tile -- all four outside one edge => no centre of the tile can be covered
            int4 es[3] = {make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0), make_int4(0, 0, 0, 0)};
            if (hit) {
                {
                    const int cx0 = (tcx * DDX_TILE) * DDX_SUBPIX + DDX_SUBPIX / 2, cy0 = (tcy * DDX_TILE) * DDX_SUBPIX + DDX_SUBPIX / 2;
                    const int cx1 = (min(tcx * DDX_TILE + DDX_TILE, W) - 1) * DDX_SUBPIX + DDX_SUBPIX / 2;
                    const int cy1 = (min(tcy * DDX_TILE + DDX_TILE, H) - 1) * DDX_SUBPIX + DDX_SUBPIX / 2;
                    const bool flip = st.area < 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int ka = (k + 1) % 3, kb = (k + 2) % 3;
                        const bool any_in = edge_inside(st.X[ka], st.Y[ka], st.X[kb], st.Y[kb], cx0, cy0, flip) ||
                                            edge_inside(st.X[ka], st.Y[ka], st.X[kb], st.Y[kb], cx1, cy0, flip) ||
                                            edge_inside(st.X[ka], st.Y[ka], st.X[kb], st.Y[kb], cx0, cy1, flip) ||
                                            edge_inside(st.X[ka], st.Y[ka], st.X[kb], st.Y[kb], cx1, cy1, flip);
                        hit = hit && any_in;
                        // tile-local form of the same exact edge function for the pixel loop: value at the tile's first
                        // pixel centre (int64) with the ownership rule folded in (e + own - 1 >= 0), and the 32-bit
                        // steps per sub-pixel in x and y
                        int dx = st.X[kb] - st.X[ka], dy = st.Y[kb] - st.Y[ka];
                        long long ev = (long long)dx * (long long)(cy0 - st.Y[ka]) - (long long)dy * (long long)(cx0 - st.X[ka]);
                        if (flip) { ev = -ev; dx = -dx; dy = -dy; }
                        ev += ((dy > 0) || (dy == 0 && dx < 0)) ? 0 : -1;
                        // (steps kept per SUB-pixel: a corner clamped to the 2^24 guard band makes |dx|, |dy| reach 2^25, and the
                        // per-pixel step dx * 256 would leave 32 bits -- a triangle with a vertex just in front of the eye plane)
                        es[k] = make_int4((int)(unsigned)(ev & 0xffffffffll), (int)(ev >> 32), -dy, dx);
                    }
                }
            }
            const unsigned long long m = __ballot(hit);
            const int nh = __popcll(m);
            if (nh == 0) continue;  // (wave-uniform)
            if (hit) {
                const int slot = __popcll(m & ((1ull << lane) - 1ull));
                s_e0[wave][slot] = es[0]; s_e1[wave][slot] = es[1]; s_e2[wave][slot] = es[2];
                s_t[wave][slot] = t_id;
                const bool have = (ent_id >> 31) != 0u;  // (a straddler's clip-space vertices are already here)
                s_p0[wave][slot] = have ? cp0 : ld4(P + (size_t)i0 * 4);
                s_p1[wave][slot] = have ? cp1 : ld4(P + (size_t)i1 * 4);
                s_p2[wave][slot] = have ? cp2 : ld4(P + (size_t)i2 * 4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- lane = 4 pixels of the tile over the staged triangles (wave-uniform walk, LDS broadcast reads)
            if (px < W) {
                for (int j = 0; j < nh; ++j) {
                    const int4 q0 = s_e0[wave][j], q1 = s_e1[wave][j], q2 = s_e2[wave][j];
                    const long long c0 = (((long long)q0.y << 32) | (unsigned)q0.x) + (long long)(lx * DDX_SUBPIX) * q0.z;
                    const long long c1 = (((long long)q1.y << 32) | (unsigned)q1.x) + (long long)(lx * DDX_SUBPIX) * q1.z;
                    const long long c2 = (((long long)q2.y << 32) | (unsigned)q2.x) + (long long)(lx * DDX_SUBPIX) * q2.z;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ly = ly0 + 4 * q, py = tcy * DDX_TILE + ly;
                        const long long v0 = c0 + (long long)(ly * DDX_SUBPIX) * q0.w, v1 = c1 + (long long)(ly * DDX_SUBPIX) * q1.w,
                                        v2 = c2 + (long long)(ly * DDX_SUBPIX) * q2.w;
                        if ((v0 | v1 | v2) < 0 || py >= H) continue;
                        const unsigned long long key = frag_key(s_p0[wave][j], s_p1[wave][j], s_p2[wave][j], px, py, H, W, s_t[wave][j]);
                        best[q] = key < best[q] ? key : best[q];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the staging arrays are rewritten by the next round)
          }
        }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int py = tcy * DDX_TILE + ly0 + 4 * q;
            if (best[q] != ~0ull) atomicMin(L.zbuf + (size_t)b * L.zper + zaddr(px, py, L.zwb), best[q]);
        }
#ifdef DDX_TRACE
        if (wave == 0) n_done += 1 + ((unsigned long long)n_big << 32);
#endif
    }
    }
    DDX_TRACE_END(L.trace, 1, n_done);
}

// ---------------------------------------------------------------------------------------------
// emit nvdiffrast-style rast [B,H,W,4]
__global__ __launch_bounds__(256) void emit_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int V,
                                                   int B, int H, int W, RasterScratch L, float* __restrict__ rast)
{
    // blockIdx.y = hypothesis, 1024 consecutive pixels of it per workgroup: 32-bit pixel arithmetic (H, W <= 4096)
    const int b = blockIdx.y, HW = H * W;
    unsigned long long keys[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = blockIdx.x * 1024 + k * 256 + threadIdx.x;
        const int py = p / W, px = p - py * W;
        keys[k] = p < HW ? L.zbuf[(size_t)b * L.zper + zaddr(px, py, L.zwb)] : ~0ull;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (p >= HW) continue;
        const int py = p / W, px = p - py * W;
        const long long i = (long long)b * HW + p;
        const unsigned long long key = keys[k];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (key != ~0ull) {
            const int t = (int)(unsigned)(key & 0xffffffffull);
            const float* P = pos + (size_t)b * V * 4;
            const float4 p0 = ld4(P + (size_t)tri[t * 3 + 0] * 4), p1 = ld4(P + (size_t)tri[t * 3 + 1] * 4),
                         p2 = ld4(P + (size_t)tri[t * 3 + 2] * 4);
            Bary bc;
            pixel_bary(p0, p1, p2, px, py, H, W, bc);
            o = make_float4(clamp01(bc.u), clamp01(bc.v), bc.zw, (float)(t + 1));
        }
        *reinterpret_cast<float4*>(rast + i * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------
int raster_run(const float* pos, const int* tri, int B, int V, int T, int H, int W, const RasterScratch& L,
               hipStream_t s, bool clear, hipEvent_t* ev)
{
    if (ev) DDX_HIP(hipEventRecord(ev[0], s));
    if (clear) {
        DDX_HIP(hipMemsetAsync(L.counters, 0, L.zero_bytes, s));
        DDX_HIP(hipMemsetAsync(L.zbuf, 0xFF, L.zbuf_bytes, s));
    }
#ifndef SCATTER_TPL_DENSE
#define SCATTER_TPL_DENSE 2  // triangles per lane of the dense-mesh variants
#endif
    constexpr int TPLD = SCATTER_TPL_DENSE;
    if ((long long)ddx_cdiv(T, 512) * B >= 1024) {
        // dense meshes: the plain kernel in the micro-polygon regime (64 VGPRs, no LDS); the hybrid (72 VGPRs, 19 KB LDS: 4-20 %
        // slower there) when the caller expects triangles to own more than about one pixel centre each
        if (L.scatter_exchange == 2) scatter_kernel<TPLD, 256, 3><<<dim3(ddx_cdiv(T, 256 * TPLD), B), 256, 0, s>>>(pos, tri, V, T, H, W, L);
        else if (L.scatter_exchange) scatter_kernel<TPLD, 256, 2><<<dim3(ddx_cdiv(T, 256 * TPLD), B), 256, 0, s>>>(pos, tri, V, T, H, W, L);
        else scatter_kernel<TPLD, 256, 0><<<dim3(ddx_cdiv(T, 256 * TPLD), B), 256, 0, s>>>(pos, tri, V, T, H, W, L);
    }
    else scatter_kernel<1, 64, 1><<<dim3(ddx_cdiv(T, 64), B), 64, 0, s>>>(pos, tri, V, T, H, W, L);
    if (ev) DDX_HIP(hipEventRecord(ev[1], s));
    compact_big_kernel<<<B + RASTER_BIG_GRID, 256, 0, s>>>(pos, tri, B, V, T, H, W, L);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t ddx_rasterize_scratch_bytes(int B, int V, int T, int H, int W)
{
    if (B < 1 || V < 1 || T < 1 || H < 1 || W < 1 || H > 4096 || W > 4096) return 0;
    RasterScratch L;
    return raster_layout(L, nullptr, B, V, T, H, W);
}

extern "C" int ddx_rasterize_fwd(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W, void* scratch,
                                 size_t scratch_bytes, float* rast, void* stream)
{
    DDX_REQUIRE(pos && tri && scratch && rast, DDX_E_NULL, "rasterize_fwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && B <= 65535 && V >= 1 && T >= 1 && H >= 1 && W >= 1 && H <= 4096 && W <= 4096, DDX_E_SHAPE,
                "rasterize_fwd: bad shape B=%d V=%d T=%d H=%d W=%d", B, V, T, H, W);
    DDX_REQUIRE(((uintptr_t)pos & 15) == 0 && ((uintptr_t)rast & 15) == 0 && ((uintptr_t)scratch & 255) == 0, DDX_E_ALIGN,
                "rasterize_fwd: pos/rast must be 16-byte and scratch 256-byte aligned");
    RasterScratch L;
    const size_t need = raster_layout(L, scratch, B, V, T, H, W);
    DDX_REQUIRE(scratch_bytes >= need, DDX_E_SCRATCH, "rasterize_fwd: scratch %zu < required %zu bytes", scratch_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    if (int e = raster_snap(pos, B, V, H, W, L, s)) return e;
    if (int e = raster_run(pos, tri, B, V, T, H, W, L, s, true, nullptr)) return e;
    emit_kernel<<<dim3((unsigned)((H * W + 1023) / 1024), (unsigned)B), 256, 0, s>>>(pos, tri, V, B, H, W, L, rast);
    DDX_LAUNCH_CHECK();
    return 0;
}
