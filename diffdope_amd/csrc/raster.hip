// raster.hip -- software tile rasteriser for gfx950 (replaces dr.rasterize / RasterizeGLContext,
// diffdope/diffdope.py:198-200,1312).
//
// Pipeline per call (all hypotheses b at once):
//   1. bin<COUNT>  one thread per (b, triangle): snap to 1/256 px, pixel-centre bbox (triangles whose
//                  bbox holds no pixel centre die here -- most of a 20k-50k triangle mesh at 640x480),
//                  bbox grown by 1 px (antialias apron) -> range of 16x16 tiles; per-tile counts are
//                  bumped with WAVE-AGGREGATED atomics (ballot of lanes hitting the same tile, one
//                  atomic per distinct tile per wave, prefix-popcount gives each lane its rank);
//   2. scan        one workgroup per hypothesis: exclusive scan of its tile counts, one global atomic
//                  to reserve the item range, compaction of the non-empty tiles into the active list;
//   3. bin<FILL>   same walk as 1, writes triangle ids into the per-tile lists;
//   4. raster      persistent workgroups stride over the ACTIVE tiles only.  256 lanes = 256
//                  triangles per step: gather the 3 clip vertices, set up, and for the (typically 1-4)
//                  pixel centres in bbox ^ tile do the exact int64 coverage test, fp32 z/w, and a
//                  64-bit (depth key, id) ds_min into a 2 KB LDS depth tile.  Triangles covering more
//                  than RASTER_BIG_PX pixels of the tile are deferred to a cooperative pass where the
//                  256 lanes are the 256 pixels.  Result: vis[b,y,x] = tri+1 (u32), 4 B per pixel,
//                  written for active tiles only.
//   5. emit        (op-level API only) full-frame expansion of vis into nvdiffrast's rast tensor
//                  (u, v, z/w, id+1); the fused engine never does this.
#include "raster.h"

// ---------------------------------------------------------------------------------------------
// scratch carving
size_t raster_layout(RasterScratch& L, void* base, int B, int T, int H, int W, long long pairs_hint)
{
    const int ntx = ddx_cdiv(W, DDX_TILE), nty = ddx_cdiv(H, DDX_TILE);
    const long long NT = (long long)ntx * nty;
    long long cap = pairs_hint > 0 ? pairs_hint : (4LL * B * T + 64LL * B * NT);
    if (cap > 0x7fffffffLL) cap = 0x7fffffffLL;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    char* p = (char*)base;
    const size_t o_counters = carve(16 * sizeof(int));
    const size_t o_count = carve((size_t)B * NT * sizeof(int));
    const size_t o_cursor = carve((size_t)B * NT * sizeof(int));
    L.zero_bytes = off;  // [counters | tile_count | tile_cursor] are zeroed with one memset per call
    const size_t o_offset = carve((size_t)B * NT * sizeof(int));
    const size_t o_active = carve((size_t)B * NT * sizeof(int));
    const size_t o_bbase = carve((size_t)B * 2 * sizeof(int));
    const size_t o_items = carve((size_t)cap * sizeof(int));
    const size_t o_vis = carve((size_t)B * H * W * sizeof(unsigned));
    L.counters = (int*)(p + o_counters);
    L.tile_count = (int*)(p + o_count);
    L.tile_cursor = (int*)(p + o_cursor);
    L.tile_offset = (int*)(p + o_offset);
    L.active = (int*)(p + o_active);
    L.b_active = (int*)(p + o_bbase);
    L.items = (int*)(p + o_items);
    L.vis = (unsigned*)(p + o_vis);
    L.capacity = (int)cap;
    L.ntx = ntx; L.nty = nty; L.NT = (int)NT;
    return off;
}

// ---------------------------------------------------------------------------------------------
// binning
template <bool FILL>
__global__ __launch_bounds__(256) void bin_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int V,
                                                  int T, int H, int W, RasterScratch L)
{
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
    if (t < T) {
        const int i0 = tri[t * 3 + 0], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
        if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
            const float* P = pos + (size_t)b * V * 4;
            const float4 p0 = ld4(P + (size_t)i0 * 4), p1 = ld4(P + (size_t)i1 * 4), p2 = ld4(P + (size_t)i2 * 4);
            SnapTri s;
            snap_triangle(p0, p1, p2, H, W, s);
            if (s.ok) {
                int px0, py0, px1, py1;
                snap_bbox(s, px0, py0, px1, py1);
                px0 = max(px0, 0); py0 = max(py0, 0);
                px1 = min(px1, W - 1); py1 = min(py1, H - 1);
                if (px0 <= px1 && py0 <= py1) {
                    // 1-pixel apron so that every pixel adjacent to a covered pixel lies in an active tile
                    tx0 = max(px0 - 1, 0) / DDX_TILE; tx1 = min(px1 + 1, W - 1) / DDX_TILE;
                    ty0 = max(py0 - 1, 0) / DDX_TILE; ty1 = min(py1 + 1, H - 1) / DDX_TILE;
                }
            }
        }
    }
    const int nx = tx1 - tx0 + 1, ny = ty1 - ty0 + 1;
    const int ntiles = (nx > 0 && ny > 0) ? nx * ny : 0;
    // every lane walks its own tile list; the wave iterates until the longest list is done
    int maxn = ntiles;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxn = max(maxn, __shfl_xor(maxn, o, 64));
    int* const counter = (FILL ? L.tile_cursor : L.tile_count) + (size_t)b * L.NT;
    for (int k = 0; k < maxn; ++k) {
        const bool valid = k < ntiles;
        const int key = valid ? (ty0 + k / nx) * L.ntx + (tx0 + k % nx) : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int key_l = __shfl(key, leader, 64);
            const unsigned long long same = __ballot(valid && key == key_l);
            const int rank = __popcll(same & ((1ull << lane) - 1ull));
            int base = 0;
            if (lane == leader) base = atomicAdd(counter + key_l, __popcll(same));
            base = __shfl(base, leader, 64);
            if (FILL && valid && key == key_l) {
                const long long slot = (long long)L.tile_offset[(size_t)b * L.NT + key_l] + base + rank;
                if (slot < L.capacity) L.items[slot] = t;
            }
            todo &= ~same;
        }
    }
}

// one workgroup per hypothesis: scan tile counts, reserve the item range, compact active tiles
__global__ __launch_bounds__(256) void scan_kernel(RasterScratch L)
{
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int wsum[4], wact[4];
    __shared__ int s_base, s_abase, s_carry, s_acarry;
    if (tid == 0) { s_carry = 0; s_acarry = 0; }
    __syncthreads();
    const int* cnt = L.tile_count + (size_t)b * L.NT;
    // pass 1: totals
    int tot = 0, act = 0;
    for (int i = tid; i < L.NT; i += 256) { const int c = cnt[i]; tot += c; act += c > 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tot += __shfl_xor(tot, o, 64); act += __shfl_xor(act, o, 64); }
    if (lane == 0) { wsum[wave] = tot; wact[wave] = act; }
    __syncthreads();
    if (tid == 0) {
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const int nact = wact[0] + wact[1] + wact[2] + wact[3];
        s_base = atomicAdd(&L.counters[1], total);
        s_abase = atomicAdd(&L.counters[2], nact);
        if ((long long)s_base + total > L.capacity) L.counters[0] = 1;
        L.b_active[b * 2 + 0] = s_abase;
        L.b_active[b * 2 + 1] = nact;
    }
    __syncthreads();
    // pass 2: exclusive scan in tile order (deterministic), 256 tiles per step
    for (int start = 0; start < L.NT; start += 256) {
        const int i = start + tid;
        const int c = i < L.NT ? cnt[i] : 0;
        const int a = c > 0;
        int incl = c, aincl = a;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o, 64), na = __shfl_up(aincl, o, 64);
            if (lane >= o) { incl += n; aincl += na; }
        }
        if (lane == 63) { wsum[wave] = incl; wact[wave] = aincl; }
        __syncthreads();
        int woff = 0, waoff = 0;
        for (int w = 0; w < wave; ++w) { woff += wsum[w]; waoff += wact[w]; }
        const int excl = s_carry + woff + incl - c;
        const int aexcl = s_acarry + waoff + aincl - a;
        if (i < L.NT) {
            L.tile_offset[(size_t)b * L.NT + i] = s_base + excl;
            if (a) L.active[s_abase + aexcl] = b * L.NT + i;
        }
        __syncthreads();
        if (tid == 255) { s_carry = excl + c; s_acarry = aexcl + a; }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// tile raster
#define RASTER_BIG_PX 8
#define RASTER_BIG_CAP 1024

__device__ __forceinline__ void load_tri(const float* __restrict__ P, const int* __restrict__ tri, int t, float4& p0,
                                         float4& p1, float4& p2)
{
    const int i0 = tri[t * 3 + 0], i1 = tri[t * 3 + 1], i2 = tri[t * 3 + 2];
    p0 = ld4(P + (size_t)i0 * 4); p1 = ld4(P + (size_t)i1 * 4); p2 = ld4(P + (size_t)i2 * 4);
}

__device__ __forceinline__ unsigned long long frag_key(const float4& p0, const float4& p1, const float4& p2,
                                                       const SnapTri& s, int px, int py, int H, int W, int t)
{
    if (!tri_covers(s, px, py)) return ~0ull;
    Bary bc;
    if (!pixel_bary(p0, p1, p2, px, py, H, W, bc)) return ~0ull;
    if (!(bc.zw >= -1.0f && bc.zw <= 1.0f)) return ~0ull;
    return ((unsigned long long)depth_key(bc.zw) << 32) | (unsigned)t;
}

__global__ __launch_bounds__(256) void raster_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int V,
                                                     int H, int W, RasterScratch L)
{
    __shared__ unsigned long long zbuf[DDX_TILE * DDX_TILE];
    __shared__ int big[RASTER_BIG_CAP];
    __shared__ int n_big;
    const int tid = threadIdx.x;
    const int n_active = L.counters[2];
    for (int work = blockIdx.x; work < n_active; work += gridDim.x) {
        const int flat = L.active[work];
        const int b = flat / L.NT, tile = flat - b * L.NT;
        const int ox = (tile % L.ntx) * DDX_TILE, oy = (tile / L.ntx) * DDX_TILE;
        const float* P = pos + (size_t)b * V * 4;
        zbuf[tid] = ~0ull;
        if (tid == 0) n_big = 0;
        __syncthreads();
        const int cnt = L.tile_count[flat];
        const int off = L.tile_offset[flat];
        const int avail = max(0, min(cnt, L.capacity - off));
        for (int i = tid; i < avail; i += 256) {
            const int t = L.items[off + i];
            float4 p0, p1, p2;
            load_tri(P, tri, t, p0, p1, p2);
            SnapTri s;
            snap_triangle(p0, p1, p2, H, W, s);
            if (!s.ok) continue;
            int px0, py0, px1, py1;
            snap_bbox(s, px0, py0, px1, py1);
            px0 = max(px0, ox); py0 = max(py0, oy);
            px1 = min(px1, min(ox + DDX_TILE, W) - 1); py1 = min(py1, min(oy + DDX_TILE, H) - 1);
            if (px0 > px1 || py0 > py1) continue;  // apron-only membership
            const int npx = (px1 - px0 + 1) * (py1 - py0 + 1);
            if (npx > RASTER_BIG_PX) {
                const int slot = atomicAdd(&n_big, 1);
                if (slot < RASTER_BIG_CAP) { big[slot] = t; continue; }
            }
            for (int py = py0; py <= py1; ++py)
                for (int px = px0; px <= px1; ++px) {
                    const unsigned long long key = frag_key(p0, p1, p2, s, px, py, H, W, t);
                    if (key != ~0ull) atomicMin(&zbuf[(py - oy) * DDX_TILE + (px - ox)], key);
                }
        }
        __syncthreads();
        // cooperative pass: lane = pixel
        const int lx = tid % DDX_TILE, ly = tid / DDX_TILE;
        const int px = ox + lx, py = oy + ly;
        unsigned long long best = zbuf[tid];
        const int nb = min(n_big, RASTER_BIG_CAP);
        if (px < W && py < H) {
            for (int j = 0; j < nb; ++j) {
                const int t = big[j];
                float4 p0, p1, p2;
                load_tri(P, tri, t, p0, p1, p2);
                SnapTri s;
                snap_triangle(p0, p1, p2, H, W, s);
                const unsigned long long key = frag_key(p0, p1, p2, s, px, py, H, W, t);
                best = key < best ? key : best;
            }
            L.vis[((size_t)b * H + py) * W + px] = best == ~0ull ? 0u : (unsigned)(best & 0xffffffffull) + 1u;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// emit nvdiffrast-style rast [B,H,W,4]
__global__ __launch_bounds__(256) void emit_kernel(const float* __restrict__ pos, const int* __restrict__ tri, int V,
                                                   int B, int H, int W, RasterScratch L, float* __restrict__ rast)
{
    const long long n = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int px = (int)(i % W);
        const int py = (int)((i / W) % H);
        const int b = (int)(i / ((long long)W * H));
        const int tile = (py / DDX_TILE) * L.ntx + (px / DDX_TILE);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (L.tile_count[(size_t)b * L.NT + tile] > 0) {
            const unsigned id = L.vis[i];
            if (id) {
                const int t = (int)id - 1;
                float4 p0, p1, p2;
                load_tri(pos + (size_t)b * V * 4, tri, t, p0, p1, p2);
                Bary bc;
                pixel_bary(p0, p1, p2, px, py, H, W, bc);
                o = make_float4(clamp01(bc.u), clamp01(bc.v), bc.zw, (float)id);
            }
        }
        *reinterpret_cast<float4*>(rast + i * 4) = o;
    }
}

// ---------------------------------------------------------------------------------------------
int raster_run(const float* pos, const int* tri, int B, int V, int T, int H, int W, const RasterScratch& L,
               hipStream_t s, hipEvent_t* ev)
{
    if (ev) DDX_HIP(hipEventRecord(ev[0], s));
    DDX_HIP(hipMemsetAsync(L.counters, 0, L.zero_bytes, s));
    dim3 gbin(ddx_cdiv(T, 256), B);
    bin_kernel<false><<<gbin, 256, 0, s>>>(pos, tri, V, T, H, W, L);
    if (ev) DDX_HIP(hipEventRecord(ev[1], s));
    scan_kernel<<<B, 256, 0, s>>>(L);
    if (ev) DDX_HIP(hipEventRecord(ev[2], s));
    bin_kernel<true><<<gbin, 256, 0, s>>>(pos, tri, V, T, H, W, L);
    if (ev) DDX_HIP(hipEventRecord(ev[3], s));
    raster_kernel<<<RASTER_GRID, 256, 0, s>>>(pos, tri, V, H, W, L);
    DDX_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t ddx_rasterize_scratch_bytes(int B, int T, int H, int W, long long pairs_hint)
{
    if (B < 1 || T < 1 || H < 1 || W < 1) return 0;
    RasterScratch L;
    return raster_layout(L, nullptr, B, T, H, W, pairs_hint);
}

extern "C" int ddx_rasterize_fwd(const float* pos, const int32_t* tri, int B, int V, int T, int H, int W, void* scratch,
                                 size_t scratch_bytes, float* rast, int32_t* status, void* stream)
{
    DDX_REQUIRE(pos && tri && scratch && rast && status, DDX_E_NULL, "rasterize_fwd: NULL pointer");
    DDX_REQUIRE(B >= 1 && B <= 65535 && V >= 1 && T >= 1 && H >= 1 && W >= 1 && H <= 16384 && W <= 16384, DDX_E_SHAPE,
                "rasterize_fwd: bad shape B=%d V=%d T=%d H=%d W=%d", B, V, T, H, W);
    DDX_REQUIRE(((uintptr_t)pos & 15) == 0 && ((uintptr_t)rast & 15) == 0 && ((uintptr_t)scratch & 255) == 0, DDX_E_ALIGN,
                "rasterize_fwd: pos/rast must be 16-byte and scratch 256-byte aligned");
    RasterScratch L;
    // infer the pairs capacity from the scratch size: everything but the item list is fixed
    const size_t fixed = raster_layout(L, scratch, B, T, H, W, 1);
    DDX_REQUIRE(scratch_bytes >= fixed, DDX_E_SCRATCH, "rasterize_fwd: scratch %zu < minimum %zu bytes", scratch_bytes, fixed);
    long long cap = (long long)((scratch_bytes - fixed) / sizeof(int)) + 1;
    // round down until the layout fits (256-byte rounding of the item section)
    while (cap > 1 && raster_layout(L, scratch, B, T, H, W, cap) > scratch_bytes) cap -= 64;
    raster_layout(L, scratch, B, T, H, W, cap < 1 ? 1 : cap);
    hipStream_t s = (hipStream_t)stream;
    if (int e = raster_run(pos, tri, B, V, T, H, W, L, s, nullptr)) return e;
    const long long n = (long long)B * H * W;
    emit_kernel<<<(n + 255) / 256 > 8192 ? 8192 : (int)((n + 255) / 256), 256, 0, s>>>(pos, tri, V, B, H, W, L, rast);
    DDX_LAUNCH_CHECK();
    DDX_HIP(hipMemcpyAsync(status, L.counters, 4 * sizeof(int), hipMemcpyDeviceToDevice, s));
    return 0;
}
