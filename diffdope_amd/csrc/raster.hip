// raster.hip -- software rasteriser for gfx950 (replaces dr.rasterize / RasterizeGLContext,
// diffdope/diffdope.py:198-200,1312): the op-level entry points.  The device code of the scatter path and of the tile pass
// lives in raster_dev.h, shared with the fused engine (engine.hip), which runs the same arithmetic from its own kernels.
//
// Meshes on this path are 20k-50k triangles landing on a few thousand pixels: most triangles own zero or
// one pixel centre.  The rasteriser is therefore split by triangle size:
//
//   scatter_kernel   one lane per (hypothesis, triangle), two triangles per lane.  Three 8-byte gathers of the
//                    per-vertex snapped window coordinates (1/256 px, produced once per vertex by snap_kernel),
//                    exact integer setup, pixel-centre bbox.  No centre inside -> dead.  Up to
//                    RASTER_SMALL_PX centres -> resolved right here in two passes: the coverage of the bbox centres
//                    as a bit mask (edge functions stepped in 32-bit integers relative to the bbox corner -- exact:
//                    a small triangle spans < 2^13 sub-pixels), then per set bit fp32 z/w and one non-returning
//                    64-bit atomicMin of (depth key, id) into the depth/visibility buffer zbuf[b,y,x].
//                    Larger -> appended (id, packed tile range) to the hypothesis' list of LARGE triangles, one
//                    atomic per wave that has any.  Either way the 16x16 tiles under bbox + 1 px (antialias apron)
//                    are flagged active with plain stores (no atomic on a hot word).
//   compact_big_kernel  workgroups [0,B): ordered per-hypothesis compaction of the flags (active lists); the other
//                    workgroups are the tile pass for LARGE triangles (none in the micro-polygon regime: they exit
//                    on one scalar load), raster_dev.h big_pass_body.  No per-tile lists in memory, so nothing can overflow.
//   emit_kernel      expands zbuf into nvdiffrast's rast tensor (u, v, z/w, id+1).
//
// zbuf invariant: all ones between passes.  The op-level entry memsets it; the fused engine keeps two copies and re-arms only
// the active tiles of the previous iteration while it draws the next one.
#include "raster_dev.h"

// ---------------------------------------------------------------------------------------------
// scratch carving
size_t raster_layout(RasterScratch& L, void* base, int B, int V, int T, int H, int W, int npar)
{
    const int ntx = ddx_cdiv(W, DDX_TILE), nty = ddx_cdiv(H, DDX_TILE);
    const long long NT = (long long)ntx * nty;
    const long long NTp = (NT + 255) & ~255ll;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    char* p = (char*)base;
    const size_t o_counters = carve(16 * sizeof(int));
    const size_t o_flag = carve((size_t)npar * B * NTp);
    const size_t o_big = carve((size_t)npar * B * NTp);
    const size_t o_bigcount = carve((size_t)npar * B * sizeof(int));
    const size_t o_bigarrive = carve((size_t)npar * B * sizeof(int));
    L.zero_bytes = off;  // [counters | tile_flag | tile_big | bigcount | bigarrive] must be zero when a pass starts
    const size_t o_active = carve((size_t)B * NT * sizeof(int));
    const size_t o_bcount = carve((size_t)B * sizeof(int));
    const size_t o_rows = carve((size_t)B * 2 * sizeof(int));
    const size_t o_snap = carve((size_t)B * V * sizeof(int2));
    const size_t o_range = carve((size_t)B * T * sizeof(uint2));
    const int zwb = (W + 3) / 4, zhb = (H + 3) / 4;
    const size_t zper = (size_t)zwb * zhb * 16;
    const size_t o_zbuf = carve((size_t)npar * B * zper * sizeof(unsigned long long));
    L.counters = (int*)(p + o_counters);
    L.tile_flag = (unsigned char*)(p + o_flag);
    L.tile_big = (unsigned char*)(p + o_big);
    L.active = (int*)(p + o_active);
    L.b_count = (int*)(p + o_bcount);
    L.row_range = (int*)(p + o_rows);
    L.snap = (int2*)(p + o_snap);
    L.biglist = (uint2*)(p + o_range);
    L.bigcount = (int*)(p + o_bigcount);
    L.bigarrive = (int*)(p + o_bigarrive);
    L.zbuf = (unsigned long long*)(p + o_zbuf);
    L.zbuf_bytes = (size_t)npar * B * zper * sizeof(unsigned long long);
    L.zper = zper;
    L.zwb = zwb;
    L.npar = npar;
    L.ntx = ntx; L.nty = nty; L.NT = (int)NT; L.NTp = (int)NTp;
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


// TPL triangles per lane, NT threads per workgroup: (2, 256) normally; (1, 64) for small meshes, where 512-triangle chunks
// would leave most of the chip without a workgroup (a 384-triangle CAD model x 64 hypotheses = 64 workgroups).
// MODE: see scatter_resolve (raster_dev.h).
template <int SCATTER_TPL, int SCATTER_NT, int MODE>
__global__ __launch_bounds__(SCATTER_NT) void scatter_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int V,
                                                      int T, int H, int W, RasterScratch L)
{
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int2* S = L.snap + (size_t)b * V;
    int t[SCATTER_TPL], i0[SCATTER_TPL], i1[SCATTER_TPL], i2[SCATTER_TPL];
    bool ok[SCATTER_TPL];
    int2 va[SCATTER_TPL], vb[SCATTER_TPL], vc[SCATTER_TPL];
#pragma unroll
    for (int k = 0; k < SCATTER_TPL; ++k) {
        const int slot = (chunk * SCATTER_TPL + k) * SCATTER_NT + threadIdx.x;
        const int tt = min(slot, T - 1);
        i0[k] = tri[tt * 3 + 0]; i1[k] = tri[tt * 3 + 1]; i2[k] = tri[tt * 3 + 2];
        t[k] = slot < T ? slot : T;  // (T marks a lane past the end)
    }
#pragma unroll
    for (int k = 0; k < SCATTER_TPL; ++k) {
        ok[k] = t[k] < T && (unsigned)i0[k] < (unsigned)V && (unsigned)i1[k] < (unsigned)V && (unsigned)i2[k] < (unsigned)V;
        const int j0 = ok[k] ? i0[k] : 0, j1 = ok[k] ? i1[k] : 0, j2 = ok[k] ? i2[k] : 0;
        va[k] = S[j0]; vb[k] = S[j1]; vc[k] = S[j2];
    }
    ScatterTarget tg;
    tg.P = pos + (size_t)b * V * 4;
    tg.Z = L.zbuf + (size_t)b * L.zper;
    tg.flag = L.tile_flag + (size_t)b * L.NTp;
    tg.big = L.tile_big + (size_t)b * L.NTp;
    tg.anybig = L.counters + 3;
    tg.biglist = L.biglist + (size_t)b * T;
    tg.bigcount = L.bigcount + b;
    tg.ntx = L.ntx; tg.nty = L.nty; tg.NT = L.NT; tg.zwb = L.zwb;
    tg.ndc = L.ndc;
    scatter_resolve<SCATTER_TPL, SCATTER_NT, MODE>(tg, H, W, T, t, i0, i1, i2, ok, va, vb, vc, 0 /* both faces, like dr.rasterize */);
}

// One launch, two roles.
//   workgroups [0, B): compaction -- workgroup b turns hypothesis b's tile flags into its ordered active-tile
//     segment + count (ballot ranks, no atomics).
//   workgroups [B, B + RASTER_BIG_GRID): large triangles -- exit on one scalar load when the batch has none;
//     otherwise big_pass_body.
#define CB_ROUNDS 16  // (CB_ROUNDS * 4 waves = 64 counts: one wave-wide prefix)
__global__ __launch_bounds__(256) void compact_big_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int B,
                                                          int V, int T, int H, int W, RasterScratch L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        const unsigned char* flg = L.tile_flag + (size_t)b * L.NTp;
        int* out = L.active + (size_t)b * L.NT;
        // 16 rounds of 256 flags at a time: all loads in flight at once, one ordered prefix over the (round, wave) counts --
        // a load / ballot / barrier loop per 256 flags paid one memory round trip per round (5 on 640x480, 15 on 1280x720)
        __shared__ int s_cnt[CB_ROUNDS * 4], s_off[CB_ROUNDS * 4 + 1];
        __shared__ int s_lo, s_hi;  // lowest / highest flagged tile index (the rows the hypothesis draws into)
        if (tid == 0) { s_lo = 0x7fffffff; s_hi = -1; }
        int my_lo = 0x7fffffff, my_hi = -1;
        int carry = 0;
        for (int start = 0; start < L.NT; start += CB_ROUNDS * 256) {
            int fl[CB_ROUNDS];
#pragma unroll
            for (int c = 0; c < CB_ROUNDS; ++c) {
                const int i = start + c * 256 + tid;
                fl[c] = i < L.NT ? (int)flg[i] : 0;
                if (fl[c] != 0) { my_lo = min(my_lo, i); my_hi = max(my_hi, i); }
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
        if (my_hi >= 0) { atomicMin(&s_lo, my_lo); atomicMax(&s_hi, my_hi); }
        __syncthreads();
        if (tid == 0) {
            L.b_count[b] = carry;
            const bool any = s_hi >= 0;
            L.row_range[b * 2 + 0] = any ? (s_lo / L.ntx) * DDX_TILE : 1;
            L.row_range[b * 2 + 1] = any ? min(H - 1, (s_hi / L.ntx) * DDX_TILE + DDX_TILE - 1) : 0;
        }
        return;
    }
    if (L.counters[3] == 0) return;  // no large triangle in the whole batch
    unsigned long long n_done = 0;
    big_pass_body<4>(pos, tri, L.snap, L.tile_big, L.biglist, L.bigcount, L.zbuf, L.zper, L.zwb, L.ntx, L.NT, L.NTp, B, V, T, H, W,
                  (int)blockIdx.x - B, (int)gridDim.x - B, n_done);
}

// ---------------------------------------------------------------------------------------------
// emit nvdiffrast-style rast [B,H,W,4]
__global__ __launch_bounds__(256) void emit_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int V,
                                                   int B, int H, int W, RasterScratch L, float* __restrict__ rast, int restrict_rows,
                                                   int restore)
{
    // blockIdx.y = hypothesis, 1024 consecutive pixels of it per workgroup: 32-bit pixel arithmetic (H, W <= 4096)
    const int b = blockIdx.y, HW = H * W;
    if (restrict_rows) {  // (the fused materialising path: rows far from the hypothesis' active tiles are never read by its consumers)
        const int lo = (L.row_range[b * 2] - EMIT_ROW_MARGIN) * W, hi = (L.row_range[b * 2 + 1] + EMIT_ROW_MARGIN + 1) * W;
        const int p0 = blockIdx.x * 1024;
        if (p0 + 1024 <= lo || p0 >= hi) return;  // (workgroup-uniform; a hypothesis that draws nothing has the range (1, 0): rows 0 .. EMIT_ROW_MARGIN are emitted, as background)
    }
    unsigned long long keys[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = blockIdx.x * 1024 + k * 256 + threadIdx.x;
        const int py = p / W, px = p - py * W;
        keys[k] = p < HW ? L.zbuf[(size_t)b * L.zper + zaddr(px, py, L.zwb)] : ~0ull;
        // (restore: every entry a triangle won is put back to "empty" once read -- a few per cent of the frame --, so the next call on
        // this scratch needs no 8-bytes-per-pixel clear of the whole buffer: ddx_rasterize_fwd_rows_clean)
        if (restore && keys[k] != ~0ull) L.zbuf[(size_t)b * L.zper + zaddr(px, py, L.zwb)] = ~0ull;
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
    // dense meshes: the plain kernel (64 VGPRs, no LDS); small ones: one triangle per lane, 64-thread workgroups, fragment exchange
    if ((long long)ddx_cdiv(T, 512) * B >= 1024) scatter_kernel<2, 256, 0><<<dim3(ddx_cdiv(T, 512), B), 256, 0, s>>>(pos, tri, V, T, H, W, L);
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
    return ddx_rasterize_fwd_rows(pos, tri, B, V, T, H, W, scratch, scratch_bytes, rast, nullptr, 1, stream);
}

extern "C" int ddx_rasterize_fwd_rows(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W, void* scratch,
                                      size_t scratch_bytes, float* rast, int32_t* row_range, int emit_all, void* stream)
{
    return ddx_rasterize_fwd_rows_clean(pos, tri, B, V, T, H, W, scratch, scratch_bytes, rast, row_range, emit_all, 0, stream);
}

// ... for a caller that keeps its scratch from call to call (a loop over iterations): zbuf_clean != 0 says that the depth buffer inside
// `scratch` is all-empty -- as THIS function leaves it: the pass that reads it puts every entry it finds back --, and the clear of
// 8 bytes per pixel and hypothesis (157 MB at 64 x 640x480, 27 us) is left out.  The first call on a scratch, a call after one of the
// other entry points, after a change of (B, V, T, H, W) or after an error passes 0.  The same rast and row_range either way.
extern "C" int ddx_rasterize_fwd_rows_clean(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W, void* scratch,
                                            size_t scratch_bytes, float* rast, int32_t* row_range, int emit_all, int zbuf_clean, void* stream)
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
    if (zbuf_clean) DDX_HIP(hipMemsetAsync(L.counters, 0, L.zero_bytes, s));
    if (int e = raster_run(pos, tri, B, V, T, H, W, L, s, !zbuf_clean, nullptr)) return e;
    const int restrict_rows = (row_range && !emit_all) ? 1 : 0;
    emit_kernel<<<dim3((unsigned)((H * W + 1023) / 1024), (unsigned)B), 256, 0, s>>>(pos, tri, V, B, H, W, L, rast, restrict_rows, 1);
    DDX_LAUNCH_CHECK();
    // the rows a hypothesis draws into, for the consumers of `rast` (the scratch is overwritten by the next call: the caller keeps a copy)
    if (row_range) DDX_HIP(hipMemcpyAsync(row_range, L.row_range, (size_t)B * 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
    return 0;
}
