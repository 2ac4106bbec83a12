// raster_dev.h -- device code of the software rasteriser shared by the op-level kernels (raster.hip: dr.rasterize,
// diffdope/diffdope.py:198-200) and the fused engine (engine.hip): the per-triangle scatter path (exact integer setup and
// coverage, depth key, 64-bit atomicMin into zbuf) and the tile pass for large / near-clipped triangles.  A ScatterTarget says
// where ONE hypothesis' data lives; the callers differ in where the vertices come from (global gathers for the op-level
// kernel, the workgroup's LDS copy of its meshlet for the engine) and in which parity of the engine's double-buffered frame
// they write.
#pragma once
#include "raster.h"

// Measurement builds only (tools/ablate_step.sh): DDX_ABLATE = n leaves out the n innermost stages of the rasteriser -- 1: the
// fragments (depth + atomicMin), 2: + coverage mask, tile flags, clip-vertex loads, 3: + the triangle predicates and the compaction,
// 4 (engine.hip): + the meshlet loop (transform) -- so that counter passes over the builds attribute the instructions of a launch
// stage by stage.  The product is built without it; its frames are wrong with it.
#ifndef DDX_ABLATE
#define DDX_ABLATE 0
#endif

struct ScatterTarget {
    const float* P;            // clip-space vertices, 4 floats each, indexed by the i0 / i1 / i2 handed to the functions below
    unsigned long long* Z;     // zbuf of the hypothesis (zaddr() layout)
    unsigned char* flag;       // [NT] tile flags of the hypothesis (plain stores of 1, no atomic on a hot word)
    unsigned char* big;        // [NT] "a large triangle overlaps the tile"
    int* anybig;               // one word: "the batch has a large triangle" (plain store of 1)
    uint2* biglist;            // the hypothesis' list of large triangles (id | near-clipped << 31, packed tile range)
    int* bigcount;             // its length (one atomic per wave that appends)
    int ntx, nty, NT, zwb;
    PixNdc ndc;                // host-computed (IEEE divisions, same values as make_pixndc)
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long frag_key(const float4& p0, const float4& p1, const float4& p2, int px, int py,
                                                       int H, int W, int t)
{
    Bary bc;
    if (!pixel_bary(p0, p1, p2, px, py, H, W, bc)) return ~0ull;
    if (!(bc.zw >= -1.0f && bc.zw <= 1.0f)) return ~0ull;
    return ((unsigned long long)depth_key(bc.zw) << 32) | (unsigned)t;
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

// Coverage mask of the <= RASTER_SMALL_PX bbox centres of a small, non-degenerate triangle (integer only; bit k = j * nxp + i <=> pixel
// (px0 + i, py0 + j)): corner a, the other two RELATIVE to it (ab, ac) and already in the order that makes the area positive (the
// callers swap them for a triangle with negative area: the edges of (a, c, b) are the flipped edges of (a, b, c) -- the same lines
// walked the other way, the same integers, the same ownership rule -- rounds 1-5 negated dx, dy and the edge value instead).  All
// factors < 2^14 for a small triangle near its own bbox: 24-bit multiplies (full rate; a 32-bit v_mul_lo_u32 issues at quarter rate).  With p = first
// centre - a:   edge a->b: f2 = ab.x p.y - ab.y p.x;   edge c->a: f1 = ac.y p.x - ac.x p.y;   edge b->c: f0 = area - f1 - f2
// (the three edge functions of a point sum to the area, exactly, in integers) -- four 24-bit multiplies instead of six, no flip selects.
__device__ __forceinline__ scatter_mask_t small_mask_rel(int ax, int ay, int abx, int aby, int acx, int acy, int px0, int py0, int nxp, int nyp)
{
    const int area = __mul24(abx, acy) - __mul24(acx, aby);  // > 0
    const int apx = px0 * DDX_SUBPIX + DDX_SUBPIX / 2 - ax, apy = py0 * DDX_SUBPIX + DDX_SUBPIX / 2 - ay;
    const int f2 = __mul24(abx, apy) - __mul24(aby, apx);
    const int f1 = __mul24(acy, apx) - __mul24(acx, apy);
    const int f0 = area - f1 - f2;
    const int d0x = acx - abx, d0y = acy - aby;  // edge b->c
    // ownership of the e == 0 line (dy > 0 || (dy == 0 && dx < 0)) folded into the start value: v + own - 1 >= 0
    const int b0 = f0 + (int)((d0y > 0) || (d0y == 0 && d0x < 0)) - 1;
    const int b1 = f1 + (int)((acy < 0) || (acy == 0 && acx > 0)) - 1;  // edge c->a: (dx, dy) = -ac
    const int b2 = f2 + (int)((aby > 0) || (aby == 0 && abx < 0)) - 1;
    // steps per pixel: sx = -dy * 256, sy = dx * 256
    const int sx2 = -aby * DDX_SUBPIX, sy2 = abx * DDX_SUBPIX, sx1 = acy * DDX_SUBPIX, sy1 = -acx * DDX_SUBPIX;
    const int sx0 = -(sx1 + sx2), sy0 = -(sy1 + sy2);
    // (a branch-free path for boxes of at most 2x2 centres that lets whole waves skip the loop below: 6 of 7 waves of cfg2's survivors
    // qualify.  Rounds 3-5 measured such a path at +-0.5 % with back faces culled; with both faces drawn and this set-up it is worth 0.7 us
    // of cfg2's 41.8 per iteration at 64 hypotheses and 3 % of the instructions at saturation: profiles/r6q_ab_relall.log, r6h_*)
    if (__ballot(nxp > 2 || nyp > 2) == 0ull) {
        const bool c00 = (b0 | b1 | b2) >= 0;
        const bool c10 = ((b0 + sx0) | (b1 + sx1) | (b2 + sx2)) >= 0 && nxp > 1;
        const bool c01 = ((b0 + sy0) | (b1 + sy1) | (b2 + sy2)) >= 0 && nyp > 1;
        const bool c11 = ((b0 + sx0 + sy0) | (b1 + sx1 + sy1) | (b2 + sx2 + sy2)) >= 0 && nxp > 1 && nyp > 1;
        return (scatter_mask_t)((unsigned)c00 | ((unsigned)c10 << 1) | (((unsigned)c01 | ((unsigned)c11 << 1)) << nxp));  // bit k = j * nxp + i
    }
    scatter_mask_t mask = 0;
    int idx = 0;
    int r0 = b0, r1 = b1, r2 = b2;
    for (int j = 0; j < nyp; ++j, r0 += sy0, r1 += sy1, r2 += sy2) {
        int v0 = r0, v1 = r1, v2 = r2;
        for (int i = 0; i < nxp; ++i, ++idx, v0 += sx0, v1 += sx1, v2 += sx2)
            mask |= (scatter_mask_t)((v0 | v1 | v2) >= 0) << idx;
    }
    return mask;
}

// one depth evaluation + one atomic per covered centre of a small triangle: the wave walks max(popcount) rounds instead of
// max(bbox area), and the clip-space vertices are loaded once, up front
__device__ __forceinline__ void walk_mask(scatter_mask_t mask, int px0, int py0, int nxp, const float4& p0, const float4& p1, const float4& p2,
                                          int t, const ScatterTarget& S)
{
    // (k + 0.5) / nxp lies at least 0.5 / nxp >= 2^-7 away from every integer and is below 64: the hardware reciprocal (1 ulp) and
    // one rounded product cannot carry it across one -- the same j as with the correctly rounded reciprocal, eleven instructions less
    // per triangle; k from the two 32-bit halves: (float) of a 64-bit bit index was a six-instruction u64 -> f32 conversion)
    const float rn = __builtin_amdgcn_rcpf((float)nxp);
#if DDX_ABLATE >= 1
    if (mask) S.flag[0] = 1;
    return;
#endif
    while (mask) {
#if RASTER_SMALL_PX > 32
        const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
        const int k = mlo ? __ffs(mlo) - 1 : 31 + __ffs(mhi);
#else
        const int k = __ffs((unsigned)mask) - 1;
#endif
        mask &= mask - 1;
        const int j = (int)(((float)k + 0.5f) * rn), i = k - __mul24(j, nxp);  // k = j * nxp + i, exact for k < 64
        float zw;
        const float fx = __fmaf_rn((float)(px0 + i), S.ndc.xs, S.ndc.xo), fy = __fmaf_rn((float)(py0 + j), S.ndc.ys, S.ndc.yo);
        if (pixel_depth(p0, p1, p2, fx, fy, zw))
            atomicMin(S.Z + zaddr(px0 + i, py0 + j, S.zwb), ((unsigned long long)depth_key(zw) << 32) | (unsigned)t);
    }
}

// tiles under bbox + 1 px: every pixel adjacent to a covered pixel lies in an active tile (conservative: a centre inside the
// bbox need not be covered).  (The usual trip count is 1 x 1: keep the compiler from unrolling / vectorising these loops.)
__device__ __forceinline__ void flag_tiles(unsigned char* flag, int ntx, int tx0, int ty0, int tx1, int ty1)
{
    // (the usual range is ONE tile: its store goes out without loop bookkeeping -- the two nested loops cost the wave a dozen scalar
    // instructions for a single trip --, and the loops run only for a lane whose range is larger; a second store of the first tile is
    // the same byte)
    flag[__mul24(ty0, ntx) + tx0] = 1;  // plain store, no atomics
    if (tx1 > tx0 || ty1 > ty0) {
#pragma clang loop unroll(disable) vectorize(disable)
        for (int ty = ty0; ty <= ty1; ++ty)
#pragma clang loop unroll(disable) vectorize(disable)
            for (int tx = tx0; tx <= tx1; ++tx) flag[__mul24(ty, ntx) + tx] = 1;
    }
}

// WALK: the lane resolves its covered centres itself, right here (the plain variant of the kernel; kept inside this function,
// in the scope that computed the mask, because hoisting it out costs 2-3 % of the kernel in the compiler's schedule).
// WALK = 2: ... unless one of the lanes that reached this point owns more than SCATTER_DIRECT_MAX centres, in which case they
// all hand their masks to the wave's fragment exchange.
#define SCATTER_FQ 128  // fragments a wave of the compacting variant queues before it resolves them (frames of at most 4096 x 4096 pixels: ddx_engine_create's limit)
#ifndef SCATTER_DIRECT_MAX
#define SCATTER_DIRECT_MAX 3  // waves in which no lane owns more fragments than this resolve them lane by lane
#endif
// DEFER (the compacting variant of the kernel): a small triangle that is neither degenerate nor culled is only reported
// (cv.clipped = 2, its pixel box in cv); scatter_resolve packs such triangles into consecutive lanes and resolves them there.
template <int WALK, bool DEFER = false>
__device__ __forceinline__ unsigned scatter_one(const ScatterTarget& S, int H, int W, int t, int i0, int i1, int i2, const int2& a, const int2& bq,
                                                const int2& c, ScatterCov& cv, int cull)
{
    cv.mask = 0; cv.px0 = 0; cv.py0 = 0; cv.nxp = 1; cv.clipped = 0;
    unsigned range = ~0u;  // packed tile range of a LARGE triangle
    // ONE level of branching for the common case (round 5).  The tests a triangle has to pass -- every vertex in front of the eye
    // plane, a bounding box that holds a pixel centre, small, not degenerate, not a culled back face -- used to be five nested ifs:
    // each costs the wave a handful of scalar instructions for its exec mask whether or not a lane takes it, and the scalar count of
    // this part equalled its vector count.  The predicates are evaluated side by side (on garbage for a vertex at w <= 0: unsigned
    // arithmetic, nothing is read from memory) and combined; the rare cases -- a large triangle, a vertex behind the eye plane -- keep
    // their own branches behind the common one.  Same stores, same atomics.
    const bool inside = a.x != INT_MIN && bq.x != INT_MIN && c.x != INT_MIN;
    const int xmin = min(a.x, min(bq.x, c.x)), xmax = max(a.x, max(bq.x, c.x));
    const int ymin = min(a.y, min(bq.y, c.y)), ymax = max(a.y, max(bq.y, c.y));
    int px0 = (int)((unsigned)xmin + (unsigned)(DDX_SUBPIX - 1 - DDX_SUBPIX / 2)) >> 8, px1 = (int)((unsigned)xmax - (unsigned)(DDX_SUBPIX / 2)) >> 8;
    int py0 = (int)((unsigned)ymin + (unsigned)(DDX_SUBPIX - 1 - DDX_SUBPIX / 2)) >> 8, py1 = (int)((unsigned)ymax - (unsigned)(DDX_SUBPIX / 2)) >> 8;
    px0 = max(px0, 0); py0 = max(py0, 0);
    px1 = min(px1, W - 1); py1 = min(py1, H - 1);
    const bool cand = inside && px0 <= px1 && py0 <= py1;
    const int nxp = px1 - px0 + 1, nyp = py1 - py0 + 1;
    const bool small = __mul24(nxp, nyp) <= RASTER_SMALL_PX && ((unsigned)xmax - (unsigned)xmin) < 8192u && ((unsigned)ymax - (unsigned)ymin) < 8192u;
    // extents < 2^13 sub-pixels: the area and the edge functions fit 32 bits exactly (only looked at for a small triangle)
    // (differences in unsigned arithmetic: a vertex at w <= 0 carries INT_MIN, and this line is no longer behind the test for it)
    const int area = __mul24((int)((unsigned)bq.x - (unsigned)a.x), (int)((unsigned)c.y - (unsigned)a.y)) -
                     __mul24((int)((unsigned)c.x - (unsigned)a.x), (int)((unsigned)bq.y - (unsigned)a.y));
    const bool front = area != 0 && !(cull != 0 && (area < 0) == (cull < 0));  // (non-degenerate and not a culled back face)
    if (cand && small && front) {
        if (DEFER) {
            cv.clipped = 2;
            cv.px0 = px0; cv.py0 = py0; cv.nxp = nxp | (nyp << 8);  // (the box travels with the survivor's record)
            cv.mask = (scatter_mask_t)(area < 0);                  // (... which holds its corners in the order of positive area)
        } else if (DDX_ABLATE >= 2) {
            S.flag[0] = 1;
        } else {
            // pass 1: coverage of the <= RASTER_SMALL_PX bbox centres as a bit mask -- integer only
            // (relative corners in the order of positive area: see small_mask_rel)
            const int abx_ = bq.x - a.x, aby_ = bq.y - a.y, acx_ = c.x - a.x, acy_ = c.y - a.y;
            const bool neg_ = area < 0;
            scatter_mask_t mask = small_mask_rel(a.x, a.y, neg_ ? acx_ : abx_, neg_ ? acy_ : aby_, neg_ ? abx_ : acx_, neg_ ? aby_ : acy_, px0, py0, nxp, nyp);
            bool walk = WALK == 1;
            if (WALK == 2) {
                const int cnt = RASTER_SMALL_PX > 32 ? __popcll(mask) : __popc((unsigned)mask);
                walk = __ballot(cnt > SCATTER_DIRECT_MAX) == 0ull;  // (over the lanes active here)
            }
            if (mask != 0) {  // a small triangle that covers no centre draws nothing: no tile to flag
                flag_tiles(S.flag, S.ntx, max(px0 - 1, 0) / DDX_TILE, max(py0 - 1, 0) / DDX_TILE, min(px1 + 1, W - 1) / DDX_TILE, min(py1 + 1, H - 1) / DDX_TILE);
                if (walk) {
                    const float4 p0 = ld4(S.P + (size_t)i0 * 4), p1 = ld4(S.P + (size_t)i1 * 4), p2 = ld4(S.P + (size_t)i2 * 4);
                    walk_mask(mask, px0, py0, nxp, p0, p1, p2, t, S);
                } else {
                    cv.mask = mask; cv.px0 = px0; cv.py0 = py0; cv.nxp = nxp;
                }
            }
        }
    }
    if (cand && !small) {  // a LARGE triangle: listed for the tile pass
        const long long area64 = (long long)(bq.x - a.x) * (long long)(c.y - a.y) - (long long)(c.x - a.x) * (long long)(bq.y - a.y);
        if (area64 != 0 && !(cull != 0 && (area64 < 0) == (cull < 0))) {
            const int tx0 = max(px0 - 1, 0) / DDX_TILE, tx1 = min(px1 + 1, W - 1) / DDX_TILE;
            const int ty0 = max(py0 - 1, 0) / DDX_TILE, ty1 = min(py1 + 1, H - 1) / DDX_TILE;
            flag_tiles(S.flag, S.ntx, tx0, ty0, tx1, ty1);
            flag_tiles(S.big, S.ntx, tx0, ty0, tx1, ty1);
            range = (unsigned)tx0 | ((unsigned)ty0 << 8) | ((unsigned)(tx1 - tx0) << 16) | ((unsigned)(ty1 - ty0) << 24);
            *S.anybig = 1;  // plain store: "the batch has a large triangle"
        }
    }
    if (!inside) {
        // a vertex at w <= 0 (rare: a hypothesis that dives through the camera).  If any corner lies in front of the near plane the
        // triangle is a straddler: the tile pass clips it (clip_near) and draws the visible part.  Where that part lands cannot
        // be bounded from the snapped corners, so it is listed for the whole frame.  (Kept to a few instructions on purpose:
        // with the clipping arithmetic inlined here the kernel's hot path lost 2x to instruction fetch, cfg2 12.5 -> 23-33 us.)
        const float2 zw0 = *reinterpret_cast<const float2*>(S.P + (size_t)i0 * 4 + 2), zw1 = *reinterpret_cast<const float2*>(S.P + (size_t)i1 * 4 + 2),
                     zw2 = *reinterpret_cast<const float2*>(S.P + (size_t)i2 * 4 + 2);
        if (zw0.x + zw0.y >= 0.f || zw1.x + zw1.y >= 0.f || zw2.x + zw2.y >= 0.f) {
#pragma clang loop unroll(disable) vectorize(disable)
            for (int i = 0; i < S.NT; ++i) { S.flag[i] = 1; S.big[i] = 1; }
            range = ((unsigned)(S.ntx - 1) << 16) | ((unsigned)(S.nty - 1) << 24);
            *S.anybig = 1;
            cv.clipped = 1;
        }
    }
    return range;
}

// The compacting variant's second stage (round 6): coverage mask + tile flags of a survivor whose pixel box travels in its record;
// the covered centres are queued as FRAGMENTS (scatter_resolve) instead of being walked by the lane that owns the triangle.
__device__ __forceinline__ scatter_mask_t scatter_small_mask(const ScatterTarget& S, int H, int W, int ax, int ay, int abx, int aby, int acx, int acy, int px0,
                                                             int py0, int nxp, int nyp)
{
#if DDX_ABLATE >= 2
    if (abx == 12345 && nxp == 77 && ax == acy && ay == aby + acx) S.flag[0] = 1;
    return 0;
#endif
    const scatter_mask_t mask = small_mask_rel(ax, ay, abx, aby, acx, acy, px0, py0, nxp, nyp);
    if (mask)  // (covers no centre: draws nothing, no tile to flag)
        flag_tiles(S.flag, S.ntx, max(px0 - 1, 0) / DDX_TILE, max(py0 - 1, 0) / DDX_TILE, min(px0 + nxp, W - 1) / DDX_TILE, min(py0 + nyp, H - 1) / DDX_TILE);
    return mask;
}

// every lane walks the fragments of its own triangle (clip-space vertices loaded only when it owns a centre)
__device__ __forceinline__ void scatter_walk(const ScatterCov& cv, const ScatterTarget& S, int i0, int i1, int i2, int t)
{
    if (!cv.mask) return;
    const float4 p0 = ld4(S.P + (size_t)i0 * 4), p1 = ld4(S.P + (size_t)i1 * 4), p2 = ld4(S.P + (size_t)i2 * 4);
    walk_mask(cv.mask, cv.px0, cv.py0, cv.nxp, p0, p1, p2, t, S);
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

// ---------------------------------------------------------------------------------------------
// The per-thread body of the scatter pass once the thread holds its TPL triangles: t[k] (original id; >= T marks a lane past the
// end), ok[k], vertex handles i0/i1/i2 (indices into S.P) and snapped corners va/vb/vc.
// MODE 0: plain -- every lane walks the fragments of its own triangles; 1: exchange (small meshes) -- the wave's fragments are
// numbered consecutively by a prefix sum of the lanes' popcounts and fragment f goes to lane f % 64; 2: plain unless a lane owns
// more than SCATTER_DIRECT_MAX centres; 3: compacting (dense meshes whose launch is several rounds of resident workgroups, where
// the kernel is VALU-bound): only ~30 % of the triangles survive the bbox / area / back-face tests, and a wave pays the coverage +
// fragment code for its 2 x 64 triangles whenever ONE lane survives; the survivors of both triangles of the lanes are packed into
// consecutive lanes through LDS and resolved in ceil(n / 64) passes: one instead of two, with full lanes.
// All variants produce bit-identical frames (ids in zbuf are the original ones; atomicMin does not care about order).
// Returns whether this thread listed a triangle for the tile pass.
template <int TPL, int NTHREADS, int MODE>
__device__ __forceinline__ bool scatter_resolve(const ScatterTarget& S, int H, int W, int T, const int (&t)[TPL], const int (&i0)[TPL], const int (&i1)[TPL],
                                                const int (&i2)[TPL], const bool (&ok)[TPL], const int2 (&va)[TPL], const int2 (&vb)[TPL],
                                                const int2 (&vc)[TPL], int cull)
{
    constexpr bool EXCHANGE = MODE == 1 || MODE == 2;
    constexpr bool COMPACT = MODE == 3;
    // compacting variant: 32 bytes per survivor: first corner, the other two relative to it in 16 bits (a small triangle spans
    // < 2^13 sub-pixels), vertex handles, triangle id
    static_assert(!COMPACT || NTHREADS * TPL <= 1024, "a fragment names its survivor in 10 bits");
    __shared__ unsigned short s_fq[COMPACT ? NTHREADS / 64 : 1][COMPACT ? SCATTER_FQ : 1];  // the wave's fragment queue: bit of the centre in the survivor's mask | survivor << 6
    __shared__ int4 s_q[2][COMPACT ? NTHREADS * TPL : 1];  // (two planes of 16-byte records: a 32-byte record per lane made every 128-bit access a two-way bank conflict)
    // fragment exchange of one wave: exclusive prefix of the lanes' fragment counts, their coverage masks and triangle records
    __shared__ int s_pref[EXCHANGE ? NTHREADS / 64 : 1][EXCHANGE ? 64 : 1];
    __shared__ scatter_mask_t s_mask[EXCHANGE ? NTHREADS / 64 : 1][EXCHANGE ? 64 : 1];
    __shared__ float4 s_rec[EXCHANGE ? NTHREADS / 64 : 1][EXCHANGE ? 64 : 1][4];  // p0, p1, p2, (px0, py0, nxp, id) as bits
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool listed = false;
    unsigned range[TPL];
    ScatterCov cv[TPL];
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
        range[k] = ~0u;
        cv[k].mask = 0; cv[k].px0 = 0; cv[k].py0 = 0; cv[k].nxp = 1; cv[k].clipped = 0;
        if (t[k] >= T || !ok[k]) continue;
        // (plain variant: triangle by triangle -- coverage of both triangles first and all fragments afterwards measured
        // 2 us slower on cfg2 in round 2: more atomics in flight at once make the atomicMin stream slower; with the wave's fragments queued
        // and resolved 64 at a time, as the compacting variant does it, round 6: 41.08 -> 41.25 us one chain, 40.0 -> 39.7 two chains:
        // tools/experiments/plain_fragment_queue.patch)
        range[k] = scatter_one<(MODE == 0 || MODE == 3) ? 1 : MODE == 2 ? 2 : 0, COMPACT>(S, H, W, t[k], i0[k], i1[k], i2[k], va[k], vb[k], vc[k], cv[k], cull);
    }
    if (COMPACT) {
        // survivors: 32 bytes each in two planes -- (first corner, the other two relative to it in 16 bits, in the order of positive
        // area) and (vertex handles in 10 bits each, triangle id, first pixel of the box, its extent), in the wave's own part of the planes.
        // (Numbering the WORKGROUP's survivors through -- the four counts behind a barrier, batch i of 64 to wave i % 4: cfg2's ~92 of 128
        // per wave are a full pass and one at 44 % of the lanes, the workgroup's ~368 five full passes and one at 75 % -- was built and
        // measured, round 6: another 5 % fewer VALU instructions at saturation, and the barrier gave it back: cfg2 @512 214 = 214 us per
        // iteration, cfg4 @512 227 -> 222, cfg50k64 55.9 -> 56.9, cfg3 108.6 -> 109.0: profiles/r6m_*.)
        constexpr int QW = 64 * TPL;
        int n_q = 0;
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
            const bool push = cv[k].clipped == 2;
            if (push) cv[k].clipped = 0;
            const unsigned long long m = __ballot(push);
            if (push) {
                const int slot = wv * QW + n_q + __popcll(m & ((1ull << lane) - 1ull));
                const unsigned rb = ((unsigned)(vb[k].x - va[k].x) & 0xffffu) | ((unsigned)(vb[k].y - va[k].y) << 16);
                const unsigned rc = ((unsigned)(vc[k].x - va[k].x) & 0xffffu) | ((unsigned)(vc[k].y - va[k].y) << 16);
                const bool neg = cv[k].mask != 0;  // (negative area: the two relative corners swap places, see small_mask_rel; the vertex handles keep their order)
                s_q[0][slot] = make_int4(va[k].x, va[k].y, (int)(neg ? rc : rb), (int)(neg ? rb : rc));
                s_q[1][slot] = make_int4(i0[k] | (i1[k] << 10) | (i2[k] << 20), t[k], cv[k].px0 | (cv[k].py0 << 16), cv[k].nxp);
            }
            n_q += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- the survivors' covered centres become FRAGMENTS of the wave's queue (2 bytes: the centre's bit in the survivor's mask, the
        // survivor), resolved 64 at a time with one depth evaluation + atomicMin per lane.  (Until round 6 the lane that owned a survivor
        // walked its centres itself: 46 % of cfg2's survivors own a centre, hardly any more than two, and a wave ran max(count) = 1.5
        // rounds of the depth code at 30 % of its lanes for every 64 survivors.)
        int n_f = 0;
        auto drain = [&]() {  // (wave-uniform)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int fb = 0; fb < n_f; fb += 64) {
                const int fi = fb + lane;
                if (fi < n_f) {
                    const unsigned e = s_fq[wv][fi];
                    const int4 h = s_q[1][e >> 6];
                    const int kb = (int)(e & 63u), nxp = h.w & 255;
                    const int j = (int)(((float)kb + 0.5f) * __builtin_amdgcn_rcpf((float)nxp)), i = kb - __mul24(j, nxp);  // kb = j * nxp + i, exact for kb < 64 (walk_mask)
                    const int px = (h.z & 0xffff) + i, py = (int)((unsigned)h.z >> 16) + j;
                    const float4 p0 = ld4(S.P + (size_t)(h.x & 1023) * 4), p1 = ld4(S.P + (size_t)((h.x >> 10) & 1023) * 4), p2 = ld4(S.P + (size_t)((h.x >> 20) & 1023) * 4);
                    float zw;
                    const float fx = __fmaf_rn((float)px, S.ndc.xs, S.ndc.xo), fy = __fmaf_rn((float)py, S.ndc.ys, S.ndc.yo);
                    if (pixel_depth(p0, p1, p2, fx, fy, zw))
                        atomicMin(S.Z + zaddr(px, py, S.zwb), ((unsigned long long)depth_key(zw) << 32) | (unsigned)h.y);
                }
            }
            n_f = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        for (int base = 0; base < n_q; base += 64) {  // (wave-uniform)
            const int sw = wv * QW + base + lane;  // the survivor's place in the planes
            scatter_mask_t mask = 0;
            if (base + lane < n_q) {
                const int4 gg = s_q[0][sw], h = s_q[1][sw];
                mask = scatter_small_mask(S, H, W, gg.x, gg.y, (int)(short)((unsigned)gg.z & 0xffffu), (int)gg.z >> 16, (int)(short)((unsigned)gg.w & 0xffffu), (int)gg.w >> 16,
                                          h.z & 0xffff, (int)((unsigned)h.z >> 16), h.w & 255, h.w >> 8);
            }
#if DDX_ABLATE >= 1
            if (mask) S.flag[0] = 1;
            mask = 0;
#endif
            for (;;) {  // round r queues the r-th covered centre of every lane that has one
                const unsigned long long bal = __ballot(mask != 0);
                if (bal == 0ull) break;
                const int n = __popcll(bal);
                if (n_f + n > SCATTER_FQ) drain();
                if (mask) {
#if RASTER_SMALL_PX > 32
                    const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
                    const int kb = mlo ? __ffs(mlo) - 1 : 31 + __ffs(mhi);
#else
                    const int kb = __ffs((unsigned)mask) - 1;
#endif
                    mask &= mask - 1;
                    s_fq[wv][n_f + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(kb | (sw << 6));
                }
                n_f += n;
            }
        }
        if (n_f) drain();
    }
    // ---- fragments.  A lane owns 0..64 covered centres of its triangle, most lanes none: walked lane by lane the wave runs
    // max(count) rounds at ~15 % lane utilisation, and one atomic instruction touches one pixel of up to 64 different
    // triangles.  Instead the wave's fragments are numbered consecutively (prefix sum of the counts) and fragment f goes to
    // lane f % 64: every lane works, and consecutive lanes take consecutive pixels of the same triangle -- the same zbuf
    // line, which is what the atomicMin stream is bound by.
    if (EXCHANGE) {
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
            const int cnt = RASTER_SMALL_PX > 32 ? __popcll(cv[k].mask) : __popc((unsigned)cv[k].mask);
            if (MODE == 1 && __ballot(cnt > SCATTER_DIRECT_MAX) == 0ull) {
                // nobody owns more than a few centres: the exchange would cost more than the idle lanes do
                scatter_walk(cv[k], S, i0[k], i1[k], i2[k], t[k]);
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
                s_rec[wv][lane][0] = ld4(S.P + (size_t)i0[k] * 4);
                s_rec[wv][lane][1] = ld4(S.P + (size_t)i1[k] * 4);
                s_rec[wv][lane][2] = ld4(S.P + (size_t)i2[k] * 4);
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
                    const int j = (int)(((float)kb + 0.5f) * __builtin_amdgcn_rcpf((float)nxp)), i = kb - __mul24(j, nxp);  // kb = j * nxp + i, exact for kb < 64 (walk_mask)
                    float zw;
                    const float fx = __fmaf_rn((float)(px0 + i), S.ndc.xs, S.ndc.xo), fy = __fmaf_rn((float)(py0 + j), S.ndc.ys, S.ndc.yo);
                    if (pixel_depth(p0, p1, p2, fx, fy, zw))
                        atomicMin(S.Z + zaddr(px0 + i, py0 + j, S.zwb), ((unsigned long long)depth_key(zw) << 32) | (unsigned)tid_);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the LDS arrays are reused by the next k)
        }
    }
    // LARGE triangles go to the hypothesis' list for the tile pass: one atomic per WAVE that has any (none in the
    // micro-polygon regime), the lanes take consecutive slots.  The order of the list does not matter (atomicMin).
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
        const unsigned long long m = __ballot(range[k] != ~0u);
        if (m == 0ull) continue;
        int base = 0;
        if (lane == __ffsll((long long)m) - 1) base = atomicAdd(S.bigcount, __popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (range[k] != ~0u) {  // (bit 31 of the id: a near-plane straddler, clipped again by the tile pass)
            S.biglist[base + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2((unsigned)t[k] | (cv[k].clipped ? 0x80000000u : 0u), range[k]);
            listed = true;
        }
    }
    return listed;
}

// clip_near as a real call: inlined into the candidate loop of the tile pass its code (used by the rare near-plane straddlers
// only) sat in the middle of the loop body every large tile walks, and the large-triangle workloads paid for it in instruction
// fetch (hugetri's tile pass 24.7 -> 32.2 us).
__device__ __attribute__((noinline)) static int clip_near_call(const float4& p0, const float4& p1, const float4& p2, int H, int W, SnapTri out[2])
{
    return clip_near(p0, p1, p2, H, W, out);
}

// ---------------------------------------------------------------------------------------------
// The tile pass for LARGE triangles (more than RASTER_SMALL_PX pixel centres in the bbox, or clipped by the near plane): G
// workgroups of 256 threads, this one is number g.  Workgroup g owns the pairs (b, tile) with tile = (g - 13 b) mod G (+ multiples
// of G): the large tiles of one object are neighbours, and the objects of all hypotheses sit at about the same place on screen --
// both a contiguous and a plain strided split pile them up on a few workgroups (measured: 4 tiles on some, none on most).  The
// flags of BIG_SCAN pairs are read in parallel and compacted into LDS; then one WAVE per large tile (four tiles in flight per
// workgroup, no workgroup barrier inside): a tile is a chain of four dependent gathers (list entry -> vertex ids -> snapped
// vertices -> clip vertices) before its pixel loop, and a workgroup that walked its tiles one by one paid the chain once per tile
// (3-4.5 us each: 51 us for the 24-triangle hugetri workload, 13 tiles per workgroup; 32 us now).  Per large tile: candidates of
// the hypothesis' list in rounds of 256 -> range test -> exact edge predicate at the 4 corner centres of the tile ->
// ballot-compaction -> the hit thread stages the triangle in LDS (tile-local int64 edge values + 32-bit steps with the ownership
// rule folded in, clip vertices) -> every lane walks the staged triangles for 4 pixels of the tile -> one atomicMin per pixel.
// tile_big [B, NTp] bytes; snap [B,V]; pos [B,V,4]; biglist [B,T]; bigcount [B]; zbuf [B, zper].
#define BIG_SCAN 1024  // (hypothesis, tile) flags scanned per step of the large-triangle pass

// The staging arrays of one wave of the tile pass (LDS): NS staged triangles per round.
template <int NS>
struct BigStage {
    int4 e0[NS], e1[NS], e2[NS];  // per edge (e.lo, e.hi, step x, step y) at the tile origin
    float4 p0[NS], p1[NS], p2[NS];  // ... the triangles' clip-space vertices
    int t[NS];                      // ... their ids
    int cand[256];                  // range-test survivors of 256 list entries
};

// ONE large tile by one wave: P / S / BL / zb are the hypothesis' own rows (clip vertices, snapped vertices, list, zbuf).
// NS = 64: every round of 64 candidates is staged at once (the tile-pass kernels); NS = 16: in chunks of 16 hits (the pass run
// from inside shade_kernel on 3 KB of LDS per wave).  Same triangles, same keys, min() does not care about the order.
template <int NS>
__device__ __forceinline__ void big_tile_wave(BigStage<NS>& st_, const float* __restrict__ P, const int* __restrict__ tri, const int2* __restrict__ S,
                                              const uint2* __restrict__ BL, int n_big, unsigned long long* __restrict__ zb, int zwb, int ntx,
                                              int tile, int H, int W)
{
    const int lane = threadIdx.x & 63;
    const int tcx = tile % ntx, tcy = tile / ntx;
    {
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
                    if (in[u]) st_.cand[ncand + __popcll(mu & ((1ull << lane) - 1ull))] = (int)en[u].x;
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
            const unsigned ent_id = cand ? (unsigned)st_.cand[idx] : 0u;  // bit 31: a near-plane straddler (see scatter_one)
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
            const int slot = __popcll(m & ((1ull << lane) - 1ull));
            float4 q0v = cp0, q1v = cp1, q2v = cp2;
            if (hit && !(ent_id >> 31)) {  // (a straddler's clip-space vertices are already here)
                q0v = ld4(P + (size_t)i0 * 4); q1v = ld4(P + (size_t)i1 * 4); q2v = ld4(P + (size_t)i2 * 4);
            }
          for (int h0 = 0; h0 < nh; h0 += NS) {  // (one trip when NS = 64)
            if (hit && slot >= h0 && slot < h0 + NS) {
                const int sl_ = slot - h0;
                st_.e0[sl_] = es[0]; st_.e1[sl_] = es[1]; st_.e2[sl_] = es[2];
                st_.t[sl_] = t_id;
                st_.p0[sl_] = q0v; st_.p1[sl_] = q1v; st_.p2[sl_] = q2v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- lane = 4 pixels of the tile over the staged triangles (wave-uniform walk, LDS broadcast reads)
            if (px < W) {
                const int nj = min(NS, nh - h0);
                for (int j = 0; j < nj; ++j) {
                    const int4 q0 = st_.e0[j], q1 = st_.e1[j], q2 = st_.e2[j];
                    const long long c0 = (((long long)q0.y << 32) | (unsigned)q0.x) + (long long)(lx * DDX_SUBPIX) * q0.z;
                    const long long c1 = (((long long)q1.y << 32) | (unsigned)q1.x) + (long long)(lx * DDX_SUBPIX) * q1.z;
                    const long long c2 = (((long long)q2.y << 32) | (unsigned)q2.x) + (long long)(lx * DDX_SUBPIX) * q2.z;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ly = ly0 + 4 * q, py = tcy * DDX_TILE + ly;
                        const long long v0 = c0 + (long long)(ly * DDX_SUBPIX) * q0.w, v1 = c1 + (long long)(ly * DDX_SUBPIX) * q1.w,
                                        v2 = c2 + (long long)(ly * DDX_SUBPIX) * q2.w;
                        if ((v0 | v1 | v2) < 0 || py >= H) continue;
                        const unsigned long long key = frag_key(st_.p0[j], st_.p1[j], st_.p2[j], px, py, H, W, st_.t[j]);
                        best[q] = key < best[q] ? key : best[q];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the staging arrays are rewritten by the next round)
          }
          }
        }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int py = tcy * DDX_TILE + ly0 + 4 * q;
            if (best[q] != ~0ull) atomicMin(zb + zaddr(px, py, zwb), best[q]);
        }
    }
}

template <int NWB /* waves per workgroup: 4 (op-level, 1024 workgroups) or 16 (engine: 256 workgroups of 1024 threads -- the same number
                     of tiles in flight, and a launch that exits at once when the batch has no large triangle costs 256 dispatches) */>
__device__ __forceinline__ void big_pass_body(const float* __restrict__ pos, const int* __restrict__ tri, const int2* __restrict__ snap,
                                              const unsigned char* __restrict__ tile_big, const uint2* __restrict__ biglist,
                                              const int* __restrict__ bigcount, unsigned long long* __restrict__ zbuf, size_t zper, int zwb, int ntx,
                                              int NT, int NTp, int B, int V, int T, int H, int W, int g, int G, unsigned long long& n_done)
{
    __shared__ BigStage<64> s_stage[NWB];  // staged LARGE triangles of each wave's tile
    __shared__ int s_big[BIG_SCAN];
    __shared__ int s_nbig;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int per_b = (NT + G - 1) / G;          // candidate tiles per hypothesis for this workgroup
    const int n_cand = B * per_b;
    for (int base = 0; base < n_cand; base += BIG_SCAN) {
    __syncthreads();
    if (tid == 0) s_nbig = 0;
    __syncthreads();
    for (int j = base + tid; j < min(n_cand, base + BIG_SCAN); j += NWB * 64) {
        const int bb = j / per_b, kk = j - bb * per_b;
        int t0 = (g - 13 * bb) % G;
        if (t0 < 0) t0 += G;
        const int tile = t0 + kk * G;
        if (tile < NT) {
            if (tile_big[(size_t)bb * NTp + tile] != 0) s_big[atomicAdd(&s_nbig, 1)] = bb * NT + tile;  // LDS atomic; the order does not matter (atomicMin below)
        }
    }
    __syncthreads();
    const int nbig = s_nbig;
    for (int e = wave; e < nbig; e += NWB) {  // (wave-uniform)
        const int flat = s_big[e];
        const int b = flat / NT, tile = flat - b * NT;
        const int n_big = min(bigcount[b], T);
        big_tile_wave<64>(s_stage[wave], pos + (size_t)b * V * 4, tri, snap + (size_t)b * V, biglist + (size_t)b * T, n_big,
                          zbuf + (size_t)b * zper, zwb, ntx, tile, H, W);
        if (wave == 0) n_done += 1 + ((unsigned long long)n_big << 32);
    }
    }
}
